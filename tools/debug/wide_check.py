import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import oracle_lib as O, hca_forge
from pycricodecs_amd import synth
from pycricodecs_amd.batch import Job
base = O.hca_encode(synth.wav(91, 90000, 2, 48000), 1)
def takes(s):
    try: O.hca_decode(s); return True
    except O.OracleError: return False
rnd = hca_forge.accepted_random_stream(base, 4242, 1.0, takes)
C = 2
job = Job.hca_decode([rnd])
bufs = job.alloc("cuda:0"); job.run(*bufs); torch.cuda.synchronize()
frames = job.units; fs = int.from_bytes(rnd[28:30], "big")
rec = ((((C * (2048 + 128 + 8) + 16) + 127) >> 7) | 1) << 7
R = (fs + 3) // 4; tiles = (frames + 63) // 64
off = (frames * rec + 255) // 256 * 256 + tiles * (R + 1) * 256 + (frames * 4 + 255) // 256 * 256
meta = bufs[2][off:off + tiles * C * 8 * 64 * 16].cpu().numpy().reshape(tiles, C, 8, 64, 16)
bits = (meta & 15).transpose(0, 3, 1, 2, 4).reshape(tiles * 64, C, 128)[:frames]
print("frames", frames, "frames with a band wider than 8 bits:", int((bits > 8).any(axis=(1, 2)).sum()))
tails = np.frombuffer(bufs[2][:frames * rec].cpu().numpy().tobytes(), dtype=np.uint8).reshape(frames, rec)[:, C * 2184:C * 2184 + 16].copy().view("<u4")
print("narrow flags:", ((tails[:, 2] & 0x40000000) != 0).astype(int).tolist()[:90])
