"""Developer tool (GPU box): decode kernel times on band-limited content (empty high bands) vs the full-band bench signal."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import oracle_lib as O
from pycricodecs_amd import synth
from pycricodecs_amd.batch import Job
n = 480000
t = np.arange(n) / 48000.0
def mk(seed, lowpass):
    rng = np.random.default_rng(seed)
    x = 0.4 * np.sin(2 * np.pi * (220 + 30 * seed) * t) + 0.2 * np.sin(2 * np.pi * (1500 + 100 * seed) * t)
    if not lowpass:
        x = x + 0.05 * rng.standard_normal(n)
    else:
        x = x + 0.02 * np.sin(2 * np.pi * 7000 * t)
    pcm = np.clip(np.round(x * 32767), -32768, 32767).astype("<i2")
    return synth.wav_bytes(np.stack([pcm, pcm[::-1]], 1), 48000)
for lowpass in (False, True):
    uniq = [O.hca_encode(mk(s, lowpass), 1) for s in range(4)]
    job = Job.hca_decode([uniq[i % 4] for i in range(1000)])
    bufs = job.alloc("cuda:0"); job.enable_events(True)
    job.run(*bufs); torch.cuda.synchronize()
    ms = {}
    for _ in range(3):
        job.run(*bufs)
        for k, v in job.event_ms().items(): ms[k] = ms.get(k, 0) + v / 3
    out = bytes(bufs[1][:int(job.output_offsets[1])].cpu().numpy()); ref = O.hca_decode(uniq[0])
    assert out[:len(ref)] == ref
    print("band-limited" if lowpass else "full band   ", {k: round(v, 3) for k, v in ms.items()})
    del bufs, job; torch.cuda.empty_cache()
