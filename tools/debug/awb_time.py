"""Developer tool (GPU box): the mixed AWB bank (configs[4] shape), kernel classes by HIP events, jobs alone and together."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import bench as B
from pycricodecs_amd.batch import Job
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
for fam in ("tonal", "sfx"):
    bank, uniq, order, subkey = B.build_awb_bank(n, 0, 1, family=fam)
    hj, aj = Job.awb_decode(bank, B.KEY)
    d_in, ho, hscr, hst = hj.alloc("cuda:0")
    _, ao, ascr, ast = aj.alloc("cuda:0", upload=False)
    hj.enable_events(True); aj.enable_events(True)
    for _ in range(2):
        aj.run(d_in, ao, ascr, ast); hj.run(d_in, ho, hscr, hst)
    torch.cuda.synchronize()
    ta = th = 0.0
    for _ in range(3):
        aj.run(d_in, ao, ascr, ast); torch.cuda.synchronize(); ta += sum(aj.event_ms().values()) / 3
        hj.run(d_in, ho, hscr, hst); torch.cuda.synchronize(); th += sum(hj.event_ms().values()) / 3
    print("%s bank of %d clips: ADX job alone %.3f ms (%d rows), HCA job alone %.3f ms (%d frames)" % (fam, n, ta, aj.units, th, hj.units), flush=True)
    del d_in, ho, hscr, ao, ascr
    torch.cuda.empty_cache()
