"""Developer tool (GPU box): ADX kernels on a batch large enough for the lane-per-chain mapping (20 000 x 2 s stereo files)."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import oracle_lib as O
from pycricodecs_amd import synth
from pycricodecs_amd.batch import Job
uniq = [synth.wav(i, 96000, 2, 48000) for i in range(8)]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
ws = [uniq[i % 8] for i in range(n)]
job = Job.adx_encode(ws)
bufs = job.alloc("cuda:0"); job.enable_events(True)
job.run(*bufs); torch.cuda.synchronize()
ms = 0
for _ in range(3):
    job.run(*bufs); ms += sum(job.event_ms().values()) / 3
adx0 = bytes(bufs[1][:int(job.output_offsets[1])].cpu().numpy()); ref = O.adx_encode(ws[0])
assert adx0[:len(ref)] == ref
print("adx encode %s: %d files, %d blocks in %.2f ms -> %.2f G blocks/s" % (job.dominant_kernel, n, job.units2, ms, job.units2 / ms / 1e6))
adx_u = [O.adx_encode(w) for w in uniq]
del bufs, job; torch.cuda.empty_cache()
job = Job.adx_decode([adx_u[i % 8] for i in range(n)])
bufs = job.alloc("cuda:0"); job.enable_events(True)
job.run(*bufs); torch.cuda.synchronize()
ms = 0
for _ in range(3):
    job.run(*bufs); ms += sum(job.event_ms().values()) / 3
print("adx decode %s: %d files, %d blocks in %.2f ms -> %.2f G blocks/s" % (job.dominant_kernel, n, job.units2, ms, job.units2 / ms / 1e6))
