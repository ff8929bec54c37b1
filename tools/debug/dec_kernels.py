"""Developer tool (GPU box): the headline HCA decode's two kernels by HIP events (one launch each)."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import bench as B
from pycricodecs_amd.batch import Job
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
fam = sys.argv[2] if len(sys.argv) > 2 else "tonal"
q = int(sys.argv[3]) if len(sys.argv) > 3 else 1
uniq = B.make_hca_streams(8, 10.0, 0, q, fam)
job = Job.hca_decode(B.tile(uniq, n), keys=[B.KEY] * n)
bufs = job.alloc("cuda:0"); job.enable_events(True)
for _ in range(2): job.run(*bufs)
torch.cuda.synchronize()
acc = {}
for _ in range(5):
    job.run(*bufs); torch.cuda.synchronize()
    for k, v in job.event_ms().items(): acc[k] = acc.get(k, 0.0) + v / 5
print("%d streams, %s, quality %d: %s  total %.3f ms  %.1f M frames/s" % (n, fam, q, "  ".join("%s %.3f ms" % kv for kv in acc.items()),
      sum(acc.values()), job.units / sum(acc.values()) / 1e3), flush=True)
