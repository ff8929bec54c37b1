import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import oracle_lib as O
from pycricodecs_amd import synth
from pycricodecs_amd.batch import Job
w = [synth.wav(i, 480000, 2, 48000) for i in range(4)]
adx = [O.adx_encode(x) for x in w]
short = [O.adx_encode(synth.wav(10 + i, 16000 + 3200 * i, 2, 48000)) for i in range(8)]
for label, items_d, items_e in (("1000 x 10 s", [adx[i % 4] for i in range(1000)], [w[i % 4] for i in range(1000)]),
                                ("8000 x ~0.5 s", [short[i % 8] for i in range(8000)], None), ("32000 x ~0.5 s", [short[i % 8] for i in range(32000)], None)):
    for mapping in ("file", "chain"):
        os.environ["CRICODECS_ADX_MAPPING"] = mapping
        for name, mk in (("decode", lambda: Job.adx_decode(items_d)), ("encode", (lambda: Job.adx_encode(items_e)) if items_e else None)):
            if mk is None: continue
            job = mk()
            bufs = job.alloc("cuda:0"); job.enable_events(True)
            job.run(*bufs); torch.cuda.synchronize()
            ms = 0.0
            for _ in range(3):
                job.run(*bufs); ms += sum(job.event_ms().values()) / 3
            print("%-16s %-6s %-6s %.3f ms (%s)" % (label, mapping, name, ms, job.dominant_kernel))
            del bufs, job
            torch.cuda.empty_cache()
