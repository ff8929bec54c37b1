"""Developer tool (GPU box): k_adx_lane_encode on 1000 x 10 s against the least segment length (per cent of the warm-up): waves per SIMD."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import bench as B
import oracle_lib as O
from pycricodecs_amd import _capi
from pycricodecs_amd.batch import Job
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
uniq = [B.family_wav(3000 + u, secs, "tonal") for u in range(8)]
refs = [O.adx_encode(w) for w in uniq]
for pct in (50, 60, 75, 80, 100, 125, 150, 200, 300):
    with _capi.testing_knobs(adx_mapping="lane", adx_seglen=pct):
        job = Job.adx_encode(B.tile(uniq, n))
        bufs = job.alloc("cuda:0"); job.enable_events(True)
        job.run(*bufs); torch.cuda.synchronize()
        ms = 0.0
        for _ in range(3):
            job.run(*bufs); torch.cuda.synchronize(); ms += sum(job.event_ms().values()) / 3
        outs = job.split(memoryview(bufs[1].cpu().numpy()))
        assert bytes(outs[3]) == refs[3] and bytes(outs[n - 1]) == refs[(n - 1) % 8]
        print("seglen %3d %% of the warm-up: %.3f ms (%s)" % (pct, ms, job.dominant_kernel), flush=True)
        del bufs, job
        torch.cuda.empty_cache()
