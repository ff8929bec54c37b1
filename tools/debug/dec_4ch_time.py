"""Developer tool (GPU box): decode kernel times for 4000 x 2.5 s four-channel streams (quality High)."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import oracle_lib as O
from pycricodecs_amd import synth
from pycricodecs_amd.batch import Job
uniq = [O.hca_encode(synth.wav(i, 120000, 4, 48000), 1) for i in range(4)]
job = Job.hca_decode([uniq[i % 4] for i in range(4000)])
bufs = job.alloc("cuda:0"); job.enable_events(True)
job.run(*bufs); torch.cuda.synchronize()
ms = {}
for _ in range(3):
    job.run(*bufs)
    for k, v in job.event_ms().items(): ms[k] = ms.get(k, 0) + v / 3
print("4ch %d frames" % job.units, {k: round(v, 3) for k, v in ms.items()})
