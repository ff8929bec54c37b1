"""Developer tool (GPU box): how many FRAMES of a material family have a band that needs more than 8 bits (resolution >= 12), against how
many 64-frame tiles do -- what a per-frame choice of int8 / int16 lines could win back over the per-tile one."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import bench as B
from pycricodecs_amd import _capi
from pycricodecs_amd.batch import Job
for fam in sys.argv[1:] or ["sparse", "mixed", "tonal"]:
    uniq = B.make_hca_streams(8, 10.0, 0, 1, fam)
    n = 64
    job = Job.hca_decode(B.tile(uniq, n), keys=[B.KEY] * n)
    bufs = job.alloc("cuda:0")
    job.run(*bufs); torch.cuda.synchronize()
    arr = (_capi.HcaGroupInfo * 64)()
    k = job._L.cri_job_hca_groups(job._h, arr, 64)
    g = arr[0]
    C_, frames = g.channels, g.frames
    tiles = (frames + 63) // 64
    desc = bufs[2][g.code_desc_offset:g.code_desc_offset + tiles * C_ * 8 * 64 * 16].cpu().numpy().reshape(tiles, C_, 8, 64, 16)
    bits = desc & 0x0F
    wide_band = bits > 8                                           # [tile][ch][block][frame][band in block]
    per_frame = wide_band.any(axis=(1, 2, 4))                      # [tile][frame]
    per_frame_count = wide_band.sum(axis=(1, 2, 4))
    per_frame_ch = wide_band.sum(axis=(2, 4)).max(axis=1)          # most wide bands of a channel, per frame
    per_tile = per_frame.any(axis=1)
    blocks = wide_band.any(axis=(1, 3, 4))                         # [tile][block]: blocks that hold a wide band in some frame
    print("%-7s frames with a wide band %5.1f %%, tiles %5.1f %%; wide bands per frame (of those): mean %.1f, most in one channel %d; "
          "16-band blocks that hold one (by block index): %s" % (fam, 100 * per_frame.mean(), 100 * per_tile.mean(),
          per_frame_count[per_frame].mean() if per_frame.any() else 0, per_frame_ch.max(), np.round(blocks.mean(axis=0), 2).tolist()), flush=True)
    del bufs

# second part: the parse's two symbol paths (table: every code of a 16-band block at most four bits in all 64 frames of the tile; else generic)
for fam in sys.argv[1:] or ["tonal", "sparse", "mixed", "noise"]:
    uniq = B.make_hca_streams(8, 10.0, 0, 1, fam)
    n = 64
    job = Job.hca_decode(B.tile(uniq, n), keys=[B.KEY] * n)
    bufs = job.alloc("cuda:0")
    job.run(*bufs); torch.cuda.synchronize()
    arr = (_capi.HcaGroupInfo * 64)()
    job._L.cri_job_hca_groups(job._h, arr, 64)
    g = arr[0]
    C_, frames = g.channels, g.frames
    tiles = (frames + 63) // 64
    desc = bufs[2][g.code_desc_offset:g.code_desc_offset + tiles * C_ * 8 * 64 * 16].cpu().numpy().reshape(tiles, C_, 8, 64, 16)
    bits = desc & 0x0F
    coded_blk = (bits > 0).any(axis=(3, 4))                        # [tile][ch][block]: blocks with any coded band
    long_sym = (bits > 4).any(axis=3)                              # [tile][ch][block][band]: a long code in some frame of the tile
    generic_blk = long_sym.any(axis=3)
    nb = coded_blk.sum()
    gen = (generic_blk & coded_blk).sum()
    short_in_generic = (~long_sym[generic_blk & coded_blk]).mean() if gen else 0.0
    print("%-7s coded blocks %d: generic path %4.1f %%; symbols of generic blocks that are short in all 64 frames: %4.1f %%; generic share by block index %s"
          % (fam, nb, 100.0 * gen / nb, 100.0 * short_in_generic, np.round((generic_blk & coded_blk).sum(axis=(0, 1)) / np.maximum(coded_blk.sum(axis=(0, 1)), 1), 2).tolist()), flush=True)
    del bufs
