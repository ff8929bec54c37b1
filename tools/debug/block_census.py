"""Developer tool (GPU box): what the parse kernel's 16-band blocks look like on the bench material.  Reads the band code
descriptions k_hca_parse leaves in scratch and classifies every (tile of 64 frames, channel, block) by the widest code any of the
tile's frames has in it: 0 bits (nothing to parse), <= 3 bits (resolutions 1-3), 4 bits (resolutions 4-7), more (8-15).
    python tools/debug/block_census.py [tonal|sparse|noise|mixed] [quality]"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import bench as B
from pycricodecs_amd import _capi
from pycricodecs_amd.batch import Job
fam = sys.argv[1] if len(sys.argv) > 1 else "tonal"
q = int(sys.argv[2]) if len(sys.argv) > 2 else 1
items = B.make_hca_streams(16, 10.0, 0, q, fam)
job = Job.hca_decode(items, keys=[B.KEY] * len(items))
bufs = job.alloc("cuda:0")
job.run(*bufs); torch.cuda.synchronize()
arr = (_capi.HcaGroupInfo * 8)()
n = _capi.lib().cri_job_hca_groups(job._h, arr, 8)
for g in arr[:n]:
    C, tiles = g.channels, (g.frames + 63) // 64
    meta = bufs[2][g.code_desc_offset:g.code_desc_offset + tiles * C * 8 * 64 * 16].cpu().numpy().reshape(tiles, C, 8, 64, 16)
    bits = meta & 15
    full = g.frames // 64                                         # whole tiles only
    bits = bits[:full]
    blockmax = bits.max(axis=(3, 4))                              # [tile][c][blk]
    lanemax = bits.max(axis=4)                                    # [tile][c][blk][frame]
    tot = blockmax.size
    print("%s q%d: %d frames, %d blocks of 16 bands x 64 frames" % (fam, q, g.frames, tot))
    for name, m in (("no bits", blockmax == 0), ("<= 2 bits", (blockmax > 0) & (blockmax <= 2)), ("3 bits", blockmax == 3), ("4 bits", blockmax == 4), ("> 4 bits", blockmax > 4)):
        print("  widest code %-9s %5.1f %% of blocks" % (name, 100.0 * m.sum() / tot))
    print("  by block index (share with widest code <= 3 bits / == 0):", [(round(100.0 * (blockmax[:, :, b] <= 3).mean()), round(100.0 * (blockmax[:, :, b] == 0).mean())) for b in range(8)])
    print("  per frame (lane) blocks with no bits: %.1f %%, <= 3 bits: %.1f %%" % (100.0 * (lanemax == 0).mean(), 100.0 * (lanemax <= 3).mean()))
    h = np.bincount(bits.reshape(-1), minlength=13)
    print("  band max-bits histogram 0..12:", (100.0 * h / h.sum()).round(1).tolist())
    # pairs of neighbouring bands (2k, 2k+1) by class
    b0, b1 = bits[..., 0::2], bits[..., 1::2]
    print("  pairs with both codes <= 3 bits: %.1f %%; equal resolution class: %.1f %%" % (100.0 * ((b0 <= 3) & (b1 <= 3)).mean(), 100.0 * (b0 == b1).mean()))
