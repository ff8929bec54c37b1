"""Developer tool (GPU box): the headline HCA decode in a loop with the shader clock and the socket power sampled beside it (rocm-smi),
then its two kernels by HIP events -- for A/B builds whose difference is traffic (a power-capped part gives saved traffic back as clock).
    python tools/debug/dec_power.py [streams [seconds]]"""
import re, subprocess, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import bench as B
from pycricodecs_amd.batch import Job
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
job = Job.hca_decode(B.tile(B.make_hca_streams(8, 10.0, 0, 1, "tonal"), n), keys=[B.KEY] * n)
bufs = job.alloc("cuda:0"); job.enable_events(True)
job.run(*bufs); torch.cuda.synchronize()
t0 = time.time(); runs = 0; smi = []
while time.time() - t0 < secs:
    for _ in range(16): job.run(*bufs)
    if len(smi) < 3 and time.time() - t0 > secs * (len(smi) + 1) / 4:
        for _ in range(16): job.run(*bufs)
        runs += 16
        txt = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        m, w = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", txt), re.search(r"Power \(W\): ([0-9.]+)", txt)
        smi.append((int(m.group(1)) if m else None, float(w.group(1)) if w else None))
    torch.cuda.synchronize(); runs += 16
ms = (time.time() - t0) / runs * 1e3
acc = {}
for _ in range(5):
    job.run(*bufs); torch.cuda.synchronize()
    for k, v in job.event_ms().items(): acc[k] = acc.get(k, 0.0) + v / 5
print("%d streams: %.3f ms per run over %.1f s; %s; (sclk MHz, W) %s" % (n, ms, time.time() - t0, "  ".join("%s %.3f ms" % kv for kv in acc.items()), smi), flush=True)
