"""Developer tool (GPU box): decode kernel times per quality (1000 x 10 s encrypted stereo streams), checked against the oracle."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import oracle_lib as O
from pycricodecs_amd import synth
from pycricodecs_amd.batch import Job
ch = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for q in (1, 2, 3, 4):
    uniq = [O.hca_encode(synth.wav(i, 480000 if ch <= 2 else 240000, ch, 48000), q) for i in range(4)]
    refs = [O.hca_decode(u) for u in uniq]
    n = 1000
    job = Job.hca_decode([uniq[i % 4] for i in range(n)])
    bufs = job.alloc("cuda:0"); job.enable_events(True)
    job.run(*bufs); torch.cuda.synchronize()
    blob = bytes(bufs[1][: int(job.output_offsets[4])].cpu().numpy())
    for i in range(4):
        o = blob[int(job.output_offsets[i]): int(job.output_offsets[i]) + len(refs[i])]
        assert o == refs[i], ("mismatch", q, i)
    assert not bufs[3].cpu().numpy().any()
    ms = {}
    for _ in range(3):
        job.run(*bufs)
        for k, v in job.event_ms().items(): ms[k] = ms.get(k, 0) + v / 3
    print("q%d ch%d %d frames" % (q, ch, job.units), {k: round(v, 3) for k, v in ms.items()}, flush=True)
