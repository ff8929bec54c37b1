import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
os.environ["CRICODECS_ADX_MAPPING"] = "lane"; os.environ["CRICODECS_ADX_WARM"] = sys.argv[1] if len(sys.argv) > 1 else "20"
import numpy as np, torch
import oracle_lib as O
from pycricodecs_amd import synth
from pycricodecs_amd.batch import Job
for (seed, n, ch) in [(1500, 32 * 900, 2), (1501, 32 * 1300 + 7, 1)]:
    w = synth.wav(seed, n, ch, 48000)
    ref = O.adx_encode(w)
    job = Job.adx_encode([w])
    print(job.dominant_kernel, job.scratch_bytes)
    bufs = job.alloc("cuda:0"); job.run(*bufs); torch.cuda.synchronize()
    out = bytes(bufs[1].cpu().numpy())[:len(ref)]
    hs = int.from_bytes(ref[2:4], "big") + 4
    rowb = 18 * ch
    bad = [(r, c) for r in range((len(ref) - hs) // rowb) for c in range(ch) if out[hs + r * rowb + c * 18: hs + r * rowb + c * 18 + 18] != ref[hs + r * rowb + c * 18: hs + r * rowb + c * 18 + 18]]
    segs = [sum(1 for (r, c) in bad if lo <= r < lo + 204) for lo in range(0, 1400, 204)]
    print("bad per segment of 204 rows:", segs)
    print("file", seed, "rows", (len(ref) - hs) // rowb, "bad blocks", len(bad), bad[:10], bad[-5:])
    print("header equal", out[:hs] == ref[:hs])
