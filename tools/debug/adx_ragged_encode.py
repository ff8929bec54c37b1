"""Developer tool (GPU box): ADX encode of a ragged bank (clips of 0.05-2 s, log-uniform, stereo) -- the lane-per-segment
encoder with and without length-sorted lanes is compared by running this at two commits."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import oracle_lib as O
from pycricodecs_amd import synth
from pycricodecs_amd.batch import Job
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = np.random.default_rng(7)
lens = np.exp(rng.uniform(np.log(0.05), np.log(2.0), 64))
uniq = [synth.wav(500 + k, int(48000 * lens[k]), 2, 48000) for k in range(64)]
order = rng.integers(0, 64, N)
items = [uniq[int(k)] for k in order]
job = Job.adx_encode(items)
bufs = job.alloc("cuda:0")
job.run(*bufs); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): job.run(*bufs)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
outs = job.split(memoryview(bufs[1].cpu().numpy()))
refs = {}
for i in list(range(0, N, max(1, N // 200))) + [N - 1]:
    k = int(order[i])
    if k not in refs: refs[k] = O.adx_encode(uniq[k])
    assert bytes(outs[i]) == refs[k], i
print("%d clips, %d block rows: %.3f ms, %.2f G rows/s (%s)" % (N, job.units, dt * 1e3, job.units / dt / 1e9, job.dominant_kernel))
