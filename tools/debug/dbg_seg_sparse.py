import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import bench as B, oracle_lib as O
from pycricodecs_amd.batch import Job
fam = sys.argv[1] if len(sys.argv) > 1 else "sparse"
adx = [O.adx_encode(B.family_wav(3000 + u, 10.0, fam)) for u in range(int(sys.argv[2]) if len(sys.argv) > 2 else 8)]
job = Job.adx_decode(adx)
print(job.dominant_kernel)
bufs = job.alloc("cuda:0"); job.run(*bufs); torch.cuda.synchronize()
outs = job.split(bytes(bufs[1].cpu().numpy()))
print([bytes(o) == O.adx_decode(a) for o, a in zip(outs, adx)])
