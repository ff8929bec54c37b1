"""Developer tool (GPU box): decode rate of v3.0 noise-fill streams (min_resolution 0), plain and with HFR / intensity stereo."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import bench as B
import oracle_lib as O
import hca_forge
from pycricodecs_amd.batch import Job
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
for label, q, ch in (("High stereo", 1, 2), ("Middle stereo (HFR)", 2, 2), ("Low stereo", 3, 2), ("High mono", 1, 1), ("High 4 ch", 1, 4), ("Middle 4 ch", 2, 4)):
    try:
        O.hca_decode(hca_forge.forge_v3(O.hca_encode(B.family_wav(8100 + 10 * q, 1.0, "tonal", ch=ch), q), 0), 0)
    except O.OracleError:
        print("%-34s the reference rejects this layout under a v3.0 header" % label); continue
    uniq = [O.hca_crypt(hca_forge.forge_v3(O.hca_encode(B.family_wav(8100 + 10 * q + u, 10.0, "tonal", ch=ch), q), 0), 1, 56, B.KEY) for u in range(4)]
    job = Job.hca_decode(B.tile(uniq, N), keys=[B.KEY] * N)
    bufs = job.alloc("cuda:0")
    job.enable_events(True)
    job.run(*bufs); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): job.run(*bufs)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    outs = job.split(memoryview(bufs[1].cpu().numpy()))
    for i in (1, N - 2): assert bytes(outs[i]) == O.hca_decode(uniq[i % 4], B.KEY), "output differs (%s)" % label
    print("%-34s %7.3f ms  %6.1f M frames/s  %s" % (label, dt * 1e3, job.units / dt / 1e6, {k: round(v, 3) for k, v in job.event_ms().items()}), flush=True)
