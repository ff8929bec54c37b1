"""Developer tool (GPU box): wave-per-file ADX decode / encode kernel times, 1000 x 10 s stereo files and one file."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import oracle_lib as O
from pycricodecs_amd import synth
from pycricodecs_amd.batch import Job
w = [synth.wav(i, 480000, 2, 48000) for i in range(4)]
adx = [O.adx_encode(x) for x in w]
for n in (1000, 1):
    for name, job in (("decode", Job.adx_decode([adx[i % 4] for i in range(n)])), ("encode", Job.adx_encode([w[i % 4] for i in range(n)]))):
        bufs = job.alloc("cuda:0"); job.enable_events(True)
        job.run(*bufs); torch.cuda.synchronize()
        ms = 0.0
        for _ in range(3):
            job.run(*bufs); ms += sum(job.event_ms().values()) / 3
        print("%s %4d files: %.3f ms" % (name, n, ms))
        del bufs
