"""Developer tool (GPU box): pipelined against one-piece host path over job sizes (where should the default switch over?)."""
import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import oracle_lib as O
from pycricodecs_amd import synth
from pycricodecs_amd.batch import Job
KEY = 0xCF222F1FE0748978
uniq = [O.hca_crypt(O.hca_encode(synth.wav(i, 480000, 2, 48000), 1), 1, 56, KEY) for i in range(4)]
for N in (4, 10, 20, 50, 100, 300, 1000):
    items = [bytes(bytearray(uniq[i % 4])) for i in range(N)]
    job = Job.hca_decode(items, keys=[KEY] * N)
    out = np.zeros(job.output_bytes, dtype=np.uint8)
    res = []
    for env in ("0", str(1 << 62)):
        os.environ["CRICODECS_HOST_SLICE_MIN"] = env
        job.run_host(out=out); best = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); job.run_host(out=out); best = min(best, time.perf_counter() - t0)
        res.append(best * 1e3)
    print("%5d streams (%6.1f MB in + out): pipelined %7.2f ms, one piece %7.2f ms" % (N, (job.input_bytes + job.output_bytes) / 1e6, res[0], res[1]), flush=True)
