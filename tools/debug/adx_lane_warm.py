"""Developer tool (GPU box): k_adx_lane_encode on 1000 x 10 s against the warm-up length (per cent of the planner's 640 rows) and the
segment length (rows): what the repairs cost when the warm-up is short, per material family."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import bench as B
import oracle_lib as O
from pycricodecs_amd import _capi
from pycricodecs_amd.batch import Job
fams = sys.argv[1].split(",") if len(sys.argv) > 1 else ["tonal", "noise", "sfx"]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
for fam in fams:
    uniq = [B.family_wav(3000 + u, 10.0, fam) for u in range(8)]
    refs = [O.adx_encode(w) for w in uniq]
    for warm_pct in (100, 50, 25, 12, 6):
        for rows in (480,):
            warm = (640 * warm_pct // 100 + 3) // 4 * 4
            with _capi.testing_knobs(adx_mapping="lane", adx_warm_pct=warm_pct, adx_seglen=max(1, rows * 100 // warm)):
                job = Job.adx_encode(B.tile(uniq, n))
                bufs = job.alloc("cuda:0"); job.enable_events(True)
                job.run(*bufs); torch.cuda.synchronize()
                ms = 0.0
                for _ in range(3):
                    job.run(*bufs); torch.cuda.synchronize(); ms += sum(job.event_ms().values()) / 3
                outs = job.split(memoryview(bufs[1].cpu().numpy()))
                for i in (0, 3, n // 2 + 1, n - 1): assert bytes(outs[i]) == refs[i % 8], (fam, warm_pct, i)
                print("%-6s warm %3d rows, segments of %4d: %.3f ms  %s" % (fam, warm, rows, ms,
                      " ".join("%s=%.3f" % kv for kv in job.event_ms().items())), flush=True)
                del bufs, job
                torch.cuda.empty_cache()
