"""Developer tool (GPU box): the HCA job of the mixed AWB bank against the transform's run length."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import bench as B
from pycricodecs_amd import _capi
from pycricodecs_amd.batch import Job
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
bank, uniq, order, subkey = B.build_awb_bank(n, 0, 1)
for run in (8, 16, 32):
    with _capi.testing_knobs(hca_run=run):
        hj, aj = Job.awb_decode(bank, B.KEY)
        d_in, ho, hscr, hst = hj.alloc("cuda:0")
        hj.enable_events(True)
        for _ in range(2): hj.run(d_in, ho, hscr, hst)
        torch.cuda.synchronize()
        acc = {}
        for _ in range(4):
            hj.run(d_in, ho, hscr, hst); torch.cuda.synchronize()
            for k, v in hj.event_ms().items(): acc[k] = acc.get(k, 0.0) + v / 4
        print("run %2d: %s  (%d frames)" % (run, "  ".join("%s %.3f ms" % kv for kv in acc.items()), hj.units), flush=True)
        del d_in, ho, hscr, hst, hj, aj
        torch.cuda.empty_cache()
