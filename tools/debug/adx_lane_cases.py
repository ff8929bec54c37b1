"""Developer tool (GPU box): the lane-per-segment ADX encoder on the shapes that matter -- 10 s files (1000 / 4000), 1 s files
(20 000), a ragged bank (100 000 clips), and the three material families (merging well, merging late, not merging)."""
import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import bench as B
import oracle_lib as O
from pycricodecs_amd.batch import Job
os.environ["CRICODECS_ADX_MAPPING"] = "lane"


def run(label, items, check):
    job = Job.adx_encode(items)
    bufs = job.alloc("cuda:0"); job.enable_events(True)
    job.run(*bufs); torch.cuda.synchronize()
    ms = 0.0
    for _ in range(3):
        job.run(*bufs); torch.cuda.synchronize(); ms += sum(job.event_ms().values()) / 3
    outs = job.split(memoryview(bufs[1].cpu().numpy()))
    for i, ref in check: assert bytes(outs[i]) == ref, (label, i)
    print("%-44s %8.3f ms  %6.2f G rows/s  (%s)" % (label, ms, job.units / ms / 1e6, job.dominant_kernel), flush=True)
    del bufs; torch.cuda.empty_cache()


for fam in ("tonal", "noise", "sparse"):
    uniq = [B.family_wav(3000 + u, 10.0, fam) for u in range(8)]
    refs = [O.adx_encode(w) for w in uniq]
    run("1000 x 10 s, %s" % fam, B.tile(uniq, 1000), [(i, refs[i % 8]) for i in (0, 3, 501, 999)])
    if fam == "tonal":
        run("4000 x 10 s, tonal", B.tile(uniq, 4000), [(i, refs[i % 8]) for i in (1, 3999)])
uniq = [B.family_wav(3100 + u, 1.0, "tonal") for u in range(8)]
refs = [O.adx_encode(w) for w in uniq]
run("20000 x 1 s, tonal", B.tile(uniq, 20000), [(i, refs[i % 8]) for i in (2, 19999)])
rng = np.random.default_rng(7)
lens = np.exp(rng.uniform(np.log(0.05), np.log(2.0), 64))
from pycricodecs_amd import synth
uniq = [synth.wav(500 + k, int(48000 * lens[k]), 2, 48000) for k in range(64)]
order = rng.integers(0, 64, 100000)
refs = {}
chk = []
for i in (0, 17, 5000, 99999):
    k = int(order[i]); refs.setdefault(k, O.adx_encode(uniq[k])); chk.append((i, refs[k]))
run("100000 ragged clips (0.05-2 s)", [uniq[int(k)] for k in order], chk)
