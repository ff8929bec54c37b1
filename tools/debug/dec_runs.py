"""Developer tool (GPU box): HCA decode against the frames of a transform run (8 / 16 / 32 / 64), by batch size."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import bench as B
from pycricodecs_amd import _capi
from pycricodecs_amd.batch import Job
uniq = B.make_hca_streams(8, 10.0, 0, 1, "tonal")
for n in (64, 250, 500, 1000, 2000, 4000, 10000):
    line = []
    for run in (8, 16, 32, 64):
        with _capi.testing_knobs(hca_run=run):
            job = Job.hca_decode(B.tile(uniq, n), keys=[B.KEY] * n)
            bufs = job.alloc("cuda:0"); job.enable_events(True)
            for _ in range(2): job.run(*bufs)
            torch.cuda.synchronize()
            tr = 0.0
            for _ in range(5):
                job.run(*bufs); torch.cuda.synchronize(); tr += job.event_ms().get("k_hca_transform", 0.0) / 5
            line.append("%d: %.3f" % (run, tr))
            del bufs, job; torch.cuda.empty_cache()
    print("%6d streams (%8d frames): transform ms by run length  %s" % (n, n * 469, "   ".join(line)), flush=True)
