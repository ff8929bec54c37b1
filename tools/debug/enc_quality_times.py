"""Developer tool (GPU box): k_hca_encode time per quality for 1000 x 10 s stereo WAVs."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from pycricodecs_amd import synth
from pycricodecs_amd.batch import Job
ch = int(sys.argv[1]) if len(sys.argv) > 1 else 2
uniq = [synth.wav(i, 480000, ch, 48000) for i in range(8)]
ws = [uniq[i % 8] for i in range(1000 if ch <= 2 else 300)]
for q in (0, 1, 2, 3, 4):
    job = Job.hca_encode(ws, quality=q)
    bufs = job.alloc("cuda:0")
    job.enable_events(True)
    job.run(*bufs); torch.cuda.synchronize()
    t = 0.0
    for _ in range(3):
        job.run(*bufs)
        t += sum(job.event_ms().values())
    print("channels %d quality %d: %.2f ms per %d frames -> %.1f M frames/s" % (ch, q, t / 3, job.units, job.units / (t / 3) / 1e3))
    del bufs, job
    torch.cuda.empty_cache()
