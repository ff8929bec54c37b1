"""Developer tool (GPU box): HCA encode rate by channel count (k_hca_encode: a wave per (frame, channel)) and by quality."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import bench as B
import oracle_lib as O
from pycricodecs_amd.batch import Job
N = int(sys.argv[1]) if len(sys.argv) > 1 else 250
for ch in (1, 2, 3, 4, 5, 6, 7, 8):
    uniq = [B.family_wav(8200 + 10 * ch + u, 10.0, "tonal", ch=ch) for u in range(4)]
    job = Job.hca_encode(B.tile(uniq, N), quality=1)
    bufs = job.alloc("cuda:0")
    job.run(*bufs); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): job.run(*bufs)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    outs = job.split(memoryview(bufs[1].cpu().numpy()))
    assert bytes(outs[1]) == O.hca_encode(uniq[1], 1), ch
    print("%d ch: %7.3f ms  %6.1f M frames/s  %6.1f M channel-frames/s" % (ch, dt * 1e3, job.units / dt / 1e6, job.units * ch / dt / 1e6), flush=True)
for q in (0, 2, 3, 4):
    uniq = [B.family_wav(8300 + u, 10.0, "tonal", ch=2) for u in range(4)]
    job = Job.hca_encode(B.tile(uniq, 1000), quality=q)
    bufs = job.alloc("cuda:0")
    job.run(*bufs); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): job.run(*bufs)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    outs = job.split(memoryview(bufs[1].cpu().numpy()))
    assert bytes(outs[1]) == O.hca_encode(uniq[1], q), q
    print("stereo, quality %d: %7.3f ms  %6.1f M frames/s" % (q, dt * 1e3, job.units / dt / 1e6), flush=True)
