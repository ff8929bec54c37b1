"""Developer tool (GPU box): time of one HCA encode job (1000 x 10 s for mono / stereo, 250 x 10 s beyond), no verification.
    python tools/debug/enc_time.py [channels [quality [files [timed runs]]]]"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from pycricodecs_amd import synth
from pycricodecs_amd.batch import Job
ch = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else (1000 if ch <= 2 else 250)
ws = [synth.wav(i, 480000, ch, 48000) for i in range(8)]
ws = (ws * ((n + 7) // 8))[:n]
job = Job.hca_encode(ws, quality=int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bufs = job.alloc("cuda:0")
for _ in range(2):
    job.run(*bufs)
torch.cuda.synchronize()
t = []
for _ in range(int(sys.argv[4]) if len(sys.argv) > 4 else 5):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); job.run(*bufs); b.record(); torch.cuda.synchronize(); t.append(a.elapsed_time(b))
ms = sorted(t)[len(t) // 2]
print("%d ch: %.3f ms, %.1f M frames/s, %.1f M channel-frames/s" % (ch, ms, job.units / ms / 1e3, job.units * ch / ms / 1e3))
