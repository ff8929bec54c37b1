"""Developer tool (GPU box): ADX encode / decode kernel time against the number of files (10 s stereo), mappings forced in turn:
encode wave (k_adx_seg_encode), lane (k_adx_lane_encode), file (k_adx_encode_wpf, unsegmented); decode seg / file / chain."""
import os, sys, json
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import oracle_lib as O
from pycricodecs_amd import synth
from pycricodecs_amd.batch import Job
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
uniq = [synth.wav(40 + i, int(48000 * secs) // 32 * 32, 2, 48000) for i in range(8)]
adx = [O.adx_encode(w) for w in uniq]


def t(job):
    bufs = job.alloc("cuda:0"); job.enable_events(True)
    job.run(*bufs); torch.cuda.synchronize()
    ms = 0.0
    for _ in range(3):
        job.run(*bufs); ms += sum(job.event_ms().values()) / 3
    del bufs; torch.cuda.empty_cache()
    return round(ms, 3)


for n in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "1,4,16,64,128,256,500,1000,2000,4000".split(","))]:
    row = {"files": n, "seconds": secs}
    for m in ("wave", "lane", "file"):
        os.environ["CRICODECS_ADX_MAPPING"] = m
        j = Job.adx_encode([uniq[i % 8] for i in range(n)])
        row["enc_" + m] = (j.dominant_kernel.replace("k_adx_", ""), t(j))
    for m in ("seg", "file", "chain"):
        os.environ["CRICODECS_ADX_MAPPING"] = m
        j = Job.adx_decode([adx[i % 8] for i in range(n)])
        row["dec_" + m] = (j.dominant_kernel.replace("k_adx_", ""), t(j))
    print(json.dumps(row), flush=True)
