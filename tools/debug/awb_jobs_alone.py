"""Developer tool (GPU box): the two jobs of the mixed AWB bank (configs[4]) one after the other, five runs each, for a kernel trace
(rocprofv3 --kernel-trace --stats): what each kernel costs when nothing runs beside it."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import bench as B
from pycricodecs_amd.batch import Job
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
which = sys.argv[2] if len(sys.argv) > 2 else "both"
bank, uniq, order, subkey = B.build_awb_bank(n, 0, 1)
hj, aj = Job.awb_decode(bank, B.KEY)
d_in, ho, hscr, hst = hj.alloc("cuda:0")
_, ao, ascr, ast = aj.alloc("cuda:0", upload=False)
for job, bufs, name in ((aj, (d_in, ao, ascr, ast), "adx"), (hj, (d_in, ho, hscr, hst), "hca")):
    if which not in ("both", name):
        continue
    for _ in range(2):
        job.run(*bufs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        job.run(*bufs)
    e1.record(); torch.cuda.synchronize()
    print("%s job alone: %.3f ms per run, %d units" % (name, e0.elapsed_time(e1) / 5, job.units), flush=True)
