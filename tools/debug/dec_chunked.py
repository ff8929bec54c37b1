"""Developer tool (GPU box): does the headline HCA decode gain from running in chunks whose frame records (2.7 KB per frame between
k_hca_parse and the transform) stay in the 256 MB Infinity Cache?  N streams as P jobs of N / P streams each, run back to back on one
stream and alternating on two; wall time by events over the whole set, against one job of N.
    python tools/debug/dec_chunked.py [streams [parts ...]]"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import bench as B
from pycricodecs_amd.batch import Job
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
parts = [int(x) for x in sys.argv[2:]] or [1, 4, 10, 25, 50, 100]
uniq = B.make_hca_streams(8, 10.0, 0, 1, "tonal")
s2 = torch.cuda.Stream()


def timed(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = []
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best.append(e0.elapsed_time(e1))
    best.sort(); return best[len(best) // 2]


for p in parts:
    per = n // p
    jobs = [Job.hca_decode(B.tile(uniq, per), keys=[B.KEY] * per) for _ in range(p)]
    bufs = [j.alloc("cuda:0") for j in jobs]
    units = sum(j.units for j in jobs)

    def one():
        for j, b in zip(jobs, bufs): j.run(*b)

    def two():
        cur = torch.cuda.current_stream()
        s2.wait_stream(cur)
        for k, (j, b) in enumerate(zip(jobs, bufs)):
            if k & 1:
                with torch.cuda.stream(s2): j.run(*b)
            else:
                j.run(*b)
        cur.wait_stream(s2)

    t1 = timed(one); t2 = timed(two) if p > 1 else float("nan")
    print("%5d parts of %5d streams (%7d frames, records %6.1f MB each): one stream %.3f ms (%.1f M frames/s), two streams %.3f ms (%.1f M)" %
          (p, per, units // p, units / p * 2665 / 1e6, t1, units / t1 / 1e3, t2, units / t2 / 1e3), flush=True)
    del jobs, bufs; torch.cuda.empty_cache()
