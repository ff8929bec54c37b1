"""Developer tool (GPU box): HCA encode / ADX encode of 1000 x 10 s stereo WAVs with the items packed back to back against items placed
so that the samples behind the 44-byte WAV header start a 128-byte line."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from pycricodecs_amd import synth
from pycricodecs_amd.batch import Job
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ws = [synth.wav(i, 480000, 2, 48000) for i in range(8)]
ws = (ws * ((n + 7) // 8))[:n]
def offsets(shift):
    o = np.zeros(n + 1, dtype=np.uint64); pos = 0
    for i, w in enumerate(ws):
        pos = (pos + 44 + shift + 127) // 128 * 128 - 44 - shift if shift is not None else pos
        o[i] = pos; pos += len(w)
    o[n] = pos
    return o
for kind in ("hca", "adx"):
    for label, offs in (("packed", None), ("samples on a line", offsets(0)), ("samples 64 B into a line", offsets(64))):
        job = (Job.hca_encode(ws, quality=1, offsets=offs) if kind == "hca" else Job.adx_encode(ws, offsets=offs))
        bufs = job.alloc("cuda:0")
        for _ in range(2): job.run(*bufs)
        torch.cuda.synchronize()
        t = []
        for _ in range(5):
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record(); job.run(*bufs); b.record(); torch.cuda.synchronize(); t.append(a.elapsed_time(b))
        print("%s encode, %-26s %.3f ms" % (kind, label + ":", sorted(t)[2]), flush=True)
        del bufs, job; torch.cuda.empty_cache()
