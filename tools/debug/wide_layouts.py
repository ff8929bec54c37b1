"""Developer tool (GPU box): decode rate of the wide / odd channel layouts (3, 5, 6, 7, 8 channels, plain formats)."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import bench as B
import oracle_lib as O
from pycricodecs_amd.batch import Job
N = int(sys.argv[1]) if len(sys.argv) > 1 else 250
for ch in (1, 2, 4, 3, 5, 6, 7, 8):
    uniq = [O.hca_crypt(O.hca_encode(B.family_wav(8000 + 10 * ch + u, 10.0, "tonal", ch=ch), 1), 1, 56, B.KEY) for u in range(4)]
    job = Job.hca_decode(B.tile(uniq, N), keys=[B.KEY] * N)
    bufs = job.alloc("cuda:0")
    job.enable_events(True)
    job.run(*bufs); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): job.run(*bufs)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    ref = O.hca_decode(uniq[1], B.KEY)
    outs = job.split(memoryview(bufs[1].cpu().numpy()))
    import os
    assert os.environ.get("NOVERIFY") or bytes(outs[1]) == ref and bytes(outs[N - 3]) == O.hca_decode(uniq[(N - 3) % 4], B.KEY), "output differs (%d ch)" % ch
    print("%d ch: %6.3f ms  %6.1f M frames/s  %s  kernels %s" % (ch, dt * 1e3, job.units / dt / 1e6, job.dominant_kernel, {k: round(v, 3) for k, v in job.event_ms().items()}), flush=True)
