"""Developer tool (GPU box): HCA decode kernel times per channel count (1000 x 10 s streams, quality High)."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import oracle_lib as O
from pycricodecs_amd import synth
from pycricodecs_amd.batch import Job
for ch in (1, 2, 3, 4, 6, 8):
    uniq = [O.hca_encode(synth.wav(i, 480000 // (1 if ch <= 2 else 4), ch, 48000), 1) for i in range(4)]
    items = [uniq[i % 4] for i in range(1000 if ch <= 2 else 400)]
    job = Job.hca_decode(items)
    bufs = job.alloc("cuda:0")
    job.enable_events(True)
    job.run(*bufs); torch.cuda.synchronize()
    ms = {}
    for _ in range(3):
        job.run(*bufs)
        for k, v in job.event_ms().items():
            ms[k] = ms.get(k, 0) + v / 3
    tot = sum(ms.values())
    print("ch %d: %d frames, %s -> %.1f M frames/s (%.1f M channel-frames/s)" % (ch, job.units, {k: round(v, 3) for k, v in ms.items()}, job.units / tot / 1e3, job.units * ch / tot / 1e3))
    del bufs, job
    torch.cuda.empty_cache()
