"""Developer tool (GPU box): how wide are the quantised lines?  Reads the per-band code descriptions k_hca_parse leaves in
scratch (max bits per band) for the bench streams and reports the share of frames / bands needing more than 8 bits."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import oracle_lib as O
from pycricodecs_amd import synth
from pycricodecs_amd.batch import Job
q = int(sys.argv[1]) if len(sys.argv) > 1 else 1
C = 2
items = [O.hca_encode(synth.wav(i, 480000, C, 48000), q) for i in range(8)]
job = Job.hca_decode(items)
bufs = job.alloc("cuda:0")
job.run(*bufs); torch.cuda.synchronize()
frames = job.units
fs = int.from_bytes(items[0][28:30], "big")
rec = ((((C * (2048 + 128 + 8) + 16) + 127) >> 7) | 1) << 7
R = (fs + 3) // 4
tiles = (frames + 63) // 64
off = frames * rec
off = (off + 255) // 256 * 256
off += tiles * (R + 1) * 256
off += (frames * 4 + 255) // 256 * 256
meta = bufs[2][off:off + tiles * C * 8 * 64 * 16].cpu().numpy().reshape(tiles, C, 8, 64, 16)   # [tile][c][blk][lane][band in block]
bits = (meta & 15).transpose(0, 3, 1, 2, 4).reshape(tiles * 64, C, 128)[:frames]           # [frame][c][band]
print("quality %d: %d frames; bands with max bits > 8 (resolution >= 12): %.2f %%; frames with any such band: %.1f %%" %
      (q, frames, 100.0 * (bits > 8).mean(), 100.0 * (bits > 8).any(axis=(1, 2)).mean()))
hist = np.bincount(bits.reshape(-1), minlength=13)
print("max-bits histogram (0..12):", (100.0 * hist / hist.sum()).round(1).tolist())
