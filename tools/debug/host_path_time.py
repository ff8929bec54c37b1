"""Developer tool (GPU box): PCIe-inclusive rate of the host-buffer batch calls (cri_job_run_host_items / _into) on N streams:
pageable and page-locked input and output, pipelined (the default for a job this size) and in one piece.
   python tools/debug/host_path_time.py [streams]"""
import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import ctypes as C
import numpy as np
import oracle_lib as O
from pycricodecs_amd import synth, _capi
from pycricodecs_amd.batch import Job, pinned_array
KEY = 0xCF222F1FE0748978
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
uniq = [O.hca_crypt(O.hca_encode(synth.wav(i, 480000, 2, 48000), 1), 1, 56, KEY) for i in range(8)]
ref = [O.hca_decode(u, KEY) for u in uniq]
items = [bytes(bytearray(uniq[i % 8])) for i in range(N)]          # N separate objects, as a caller that read N files has them
job = Job.hca_decode(items, keys=[KEY] * N)
L = _capi.lib()
status = (C.c_int32 * job.n)()


def timed(label, fn, reps=3):
    fn()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); outs, st = fn(); best = min(best, time.perf_counter() - t0)
    assert not np.any(st)
    for i in (0, 1, N // 2, N - 1): assert bytes(outs[i]) == ref[i % 8], (label, i)
    print("%-64s %8.1f ms -> %6.2f M frames/s (%.1f GB/s out)" % (label, best * 1e3, job.units / best / 1e6, job.output_bytes / best / 1e9), flush=True)


def run_blob(src, out):
    rc = L.cri_job_run_host_into(job._h, src, out.ctypes.data, status)
    assert rc == 0
    return job.split(memoryview(out)), np.array(status[:N])


print("%d streams: %d frames, %.2f GB in, %.2f GB out" % (N, job.units, job.input_bytes / 1e9, job.output_bytes / 1e9))
pin = pinned_array(job.output_bytes)
page = np.zeros(job.output_bytes, dtype=np.uint8)
blob = job.blob
pin_in = pinned_array(len(blob)); pin_in[:] = np.frombuffer(blob, dtype=np.uint8)
for env, tag in ((None, "pipelined"), (str(1 << 62), "one piece")):
    if env is None: os.environ.pop("CRICODECS_HOST_SLICE_MIN", None)
    else: os.environ["CRICODECS_HOST_SLICE_MIN"] = env
    timed("items (separate bytes objects) -> pageable out, " + tag, lambda: job.run_host(out=page))
    timed("items (separate bytes objects) -> page-locked out, " + tag, lambda: job.run_host(out=pin))
    timed("blob (pageable) -> pageable out, " + tag, lambda: run_blob(blob, page))
    timed("blob (pageable) -> page-locked out, " + tag, lambda: run_blob(blob, pin))
    timed("blob (page-locked) -> page-locked out, " + tag, lambda: run_blob(pin_in.ctypes.data, pin))
