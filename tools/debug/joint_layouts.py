"""Developer tool (GPU box): decode rate of joint-stereo / HFR formats (Middle, Low quality) by channel count."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import bench as B
import oracle_lib as O
from pycricodecs_amd.batch import Job
N = int(sys.argv[1]) if len(sys.argv) > 1 else 250
for q in (2, 3):
    for ch in (2, 3, 4, 5, 6, 7, 8):
        uniq = [O.hca_crypt(O.hca_encode(B.family_wav(8300 + 10 * ch + u, 10.0, "tonal", ch=ch), q), 1, 56, B.KEY) for u in range(4)]
        job = Job.hca_decode(B.tile(uniq, N), keys=[B.KEY] * N)
        bufs = job.alloc("cuda:0"); job.enable_events(True)
        job.run(*bufs); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): job.run(*bufs)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        outs = job.split(memoryview(bufs[1].cpu().numpy()))
        assert bytes(outs[1]) == O.hca_decode(uniq[1], B.KEY), (q, ch)
        print("quality %s, %d ch: %7.3f ms  %6.1f M frames/s  %6.1f M channel-frames/s  %s" % (B.QNAME[q], ch, dt * 1e3, job.units / dt / 1e6, job.units * ch / dt / 1e6, {k: round(v, 3) for k, v in job.event_ms().items()}), flush=True)
