"""Developer tool (GPU box): k_adx_lane_encode, files of 1 s left uncut, against the number of files: where the time steps up tells how
the launch's waves land on the SIMDs (391 waves at 12 500 files, 1024 SIMDs on the device)."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import bench as B
import oracle_lib as O
from pycricodecs_amd import _capi
from pycricodecs_amd.batch import Job
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
counts = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else (16000, 20000, 22000, 24000, 26000, 32000, 40000)
uniq = [B.family_wav(3000 + u, secs, "tonal") for u in range(8)]
refs = [O.adx_encode(w) for w in uniq]
for n in counts:
    with _capi.testing_knobs(adx_mapping="lane", adx_seglen=300):
        job = Job.adx_encode(B.tile(uniq, n))
        bufs = job.alloc("cuda:0"); job.enable_events(True)
        job.run(*bufs); torch.cuda.synchronize()
        ms = 0.0
        for _ in range(3):
            job.run(*bufs); torch.cuda.synchronize(); ms += sum(job.event_ms().values()) / 3
        outs = job.split(memoryview(bufs[1].cpu().numpy()))
        assert bytes(outs[3]) == refs[3] and bytes(outs[n - 1]) == refs[(n - 1) % 8]
        print("%6d files (%5d waves): %.3f ms  %.2f G rows/s  %s" % (n, n * 2 // 64, ms, job.units / ms / 1e6,
              " ".join("%s=%.3f" % kv for kv in job.event_ms().items())), flush=True)
        del bufs, job
        torch.cuda.empty_cache()
