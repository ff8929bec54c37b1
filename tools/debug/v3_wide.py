"""Developer tool (GPU box): v3.0 noise fill on the wide layouts against the same layouts without it (250 x 10 s streams)."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import bench as B
import hca_forge
import oracle_lib as O
from pycricodecs_amd.batch import Job
for ch in (3, 5, 6, 8):
    for v3 in (False, True):
        plain = [O.hca_encode(B.family_wav(8000 + 10 * ch + u, 10.0, "tonal", ch=ch), 1) for u in range(4)]
        if v3:
            plain = [hca_forge.forge_v3(h, 0) for h in plain]
        items = [O.hca_crypt(h, 1, 56, B.KEY) for h in plain]
        job = Job.hca_decode(B.tile(items, 250), keys=[B.KEY] * 250)
        bufs = job.alloc("cuda:0"); job.enable_events(True)
        job.run(*bufs); torch.cuda.synchronize()
        ms = {}
        for _ in range(3):
            job.run(*bufs); torch.cuda.synchronize()
            for k, v in job.event_ms().items(): ms[k] = ms.get(k, 0) + v / 3
        outs = job.split(memoryview(bufs[1].cpu().numpy()))
        assert bytes(outs[1]) == O.hca_decode(items[1], B.KEY)
        tot = sum(ms.values())
        print("%d ch %s: forms %s  %.3f ms  %.1f M frames/s  %s" % (ch, "v3.0 noise fill" if v3 else "v2.0           ", job.transform_forms(), tot, job.units / tot / 1e3, {k: round(v, 3) for k, v in ms.items()}), flush=True)
        del bufs; torch.cuda.empty_cache()
