"""Developer tool (GPU box): how many quantised lines of a family's frames do not fit int8 (the tile-wide int16 form's reason)."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import bench as B
from pycricodecs_amd import _capi
from pycricodecs_amd.batch import Job
for fam in (sys.argv[1:] or ["sparse", "mixed", "tonal"]):
    items = B.make_hca_streams(16, 10.0, 0, 1, fam)
    job = Job.hca_decode(items, keys=[B.KEY] * len(items))
    bufs = job.alloc("cuda:0")
    job.run(*bufs); torch.cuda.synchronize()
    arr = (_capi.HcaGroupInfo * 8)()
    n = _capi.lib().cri_job_hca_groups(job._h, arr, 8)
    for g in arr[:n]:
        C, tiles = g.channels, (g.frames + 63) // 64
        tile_bytes = 8 * C * 4 * 4096
        raw = bufs[2][g.lines_offset:g.lines_offset + tiles * tile_bytes].cpu().numpy()
        flags = np.array([int(np.frombuffer(bufs[2][g.first_record_offset + f * g.record_bytes + g.flags_offset:g.first_record_offset + f * g.record_bytes + g.flags_offset + 4].cpu().numpy(), dtype=np.uint32)[0]) for f in range(0, g.frames, 64)])
        wide_tiles = [t for t in range(g.frames // 64) if not (flags[t] & g.narrow_flag)]
        tot = big = 0
        per_frame = []
        for t in wide_tiles[:40]:
            v = raw[t * tile_bytes:(t + 1) * tile_bytes].view(np.int16).reshape(8, C, 4, 64, 32)      # [sf][c][quarter][frame][line]
            ov = np.abs(v.astype(np.int32)) > 127
            tot += v.size; big += int(ov.sum())
            per_frame.extend(ov.sum(axis=(0, 1, 2, 4)).tolist())
        pf = np.array(per_frame) if per_frame else np.zeros(1)
        print("%s: %d of %d tiles wide; lines beyond int8: %.3f %%; per frame: mean %.1f, max %d, frames without any %.1f %%" % (
            fam, len(wide_tiles), g.frames // 64, 100.0 * big / max(tot, 1), pf.mean(), int(pf.max()), 100.0 * (pf == 0).mean()))
