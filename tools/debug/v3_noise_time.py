"""Developer tool (GPU box): decode kernel times for forged v3.0 / min_resolution 0 streams (noise reconstruction path)."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import hca_forge, oracle_lib as O
from pycricodecs_amd import synth
from pycricodecs_amd.batch import Job
for q in (1, 2):
    uniq = [hca_forge.forge_v3(O.hca_encode(synth.wav(i, 480000, 2, 48000), q), 0) for i in range(4)]
    job = Job.hca_decode([uniq[i % 4] for i in range(1000)])
    bufs = job.alloc("cuda:0"); job.enable_events(True)
    job.run(*bufs); torch.cuda.synchronize()
    ms = {}
    for _ in range(3):
        job.run(*bufs)
        for k, v in job.event_ms().items(): ms[k] = ms.get(k, 0) + v / 3
    out = bytes(bufs[1][:int(job.output_offsets[1])].cpu().numpy()); ref = O.hca_decode(uniq[0])
    assert out[:len(ref)] == ref
    print("v3 min_res 0, quality %d:" % q, {k: round(v, 3) for k, v in ms.items()}, "-> %.1f M frames/s" % (job.units / sum(ms.values()) / 1e3))
    del bufs, job; torch.cuda.empty_cache()
