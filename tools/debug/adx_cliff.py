"""Developer tool (GPU box): the lane-per-(file, channel, segment) ADX encoder on n files of 1 s -- the step between 24 000 and 26 000 files
(DESIGN section 2).  python tools/debug/adx_cliff.py N [seconds]"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from pycricodecs_amd import synth
from pycricodecs_amd.batch import Job
n = int(sys.argv[1]) if len(sys.argv) > 1 else 26000
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
uniq = [synth.wav(i, int(48000 * secs) // 32 * 32, 2, 48000) for i in range(8)]
job = Job.adx_encode([uniq[i % 8] for i in range(n)])
bufs = job.alloc("cuda:0"); job.enable_events(True)
job.run(*bufs); torch.cuda.synchronize()
ms = []
for _ in range(3):
    job.run(*bufs); ms.append(sum(job.event_ms().values()))
print("adx encode %s: %d files x %.1f s (%.2f GB in), %.2f ms (%s)" % (job.dominant_kernel, n, secs, job.input_bytes / 1e9, sorted(ms)[1], " ".join("%.2f" % m for m in ms)), flush=True)
