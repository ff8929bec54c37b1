"""Developer tool (GPU box): k_hca_crypt time for 1000 x 10 s stereo streams."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import oracle_lib as O
from pycricodecs_amd import synth
from pycricodecs_amd.batch import Job
KEY = 0xCF222F1FE0748978
uniq = [O.hca_encode(synth.wav(i, 480000, 2, 48000), 1) for i in range(4)]  # 10 s stereo
items = [uniq[i % 4] for i in range(1000)]
job = Job.hca_crypt(items, 1, 56, keys=[KEY] * len(items))
bufs = job.alloc("cuda:0")
job.enable_events(True)
job.run(*bufs); torch.cuda.synchronize()
ms = 0
for _ in range(3):
    job.run(*bufs); ms += sum(job.event_ms().values()) / 3
out = bytes(bufs[1][:len(uniq[0])].cpu().numpy())
assert out == O.hca_crypt(uniq[0], 1, 56, KEY), "crypt differs from the oracle"
print("hca_crypt: %d frames in %.3f ms -> %.1f M frames/s, %.1f GB/s (2 x frame bytes)" % (job.units, ms, job.units / ms / 1e3, job.algorithmic_bytes / ms / 1e6))
