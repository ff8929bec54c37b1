"""Developer tool (GPU box): PCIe-inclusive rate of the host-buffer batch calls (cri_job_run_host_items / _into) on N streams,
pageable and page-locked output, unsliced and pipelined.   python tools/debug/host_path_time.py [streams]"""
import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import oracle_lib as O
from pycricodecs_amd import synth, _capi
from pycricodecs_amd.batch import Job, pinned_array, pinned_release
KEY = 0xCF222F1FE0748978
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
uniq = [O.hca_crypt(O.hca_encode(synth.wav(i, 480000, 2, 48000), 1), 1, 56, KEY) for i in range(8)]
ref0 = O.hca_decode(uniq[0], KEY)
items = [uniq[i % 8] for i in range(N)]
job = Job.hca_decode(items, keys=[KEY] * len(items))


def timed(label, fn, reps=3):
    fn()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); outs, st = fn(); dt = time.perf_counter() - t0
        best = min(best, dt)
    assert not st.any() and bytes(outs[0]) == ref0 and bytes(outs[-8]) == ref0
    print("%-58s %8.1f ms -> %6.2f M frames/s (%.1f GB/s out)" % (label, best * 1e3, job.units / best / 1e6, job.output_bytes / best / 1e9), flush=True)


print("%d streams: %d frames, %.2f GB in, %.2f GB out" % (N, job.units, job.input_bytes / 1e9, job.output_bytes / 1e9))
pin = pinned_array(job.output_bytes)
page = np.empty(job.output_bytes, dtype=np.uint8); page[:] = 0
for env in ("0", str(1 << 62)):
    os.environ["CRICODECS_HOST_SLICE_MIN"] = env
    tag = "pipelined" if env == "0" else "one piece"
    timed("items (pageable bytes) -> pageable out, " + tag, lambda: job.run_host(out=page))
    timed("items (pageable bytes) -> pinned out,   " + tag, lambda: job.run_host(out=pin))
# blob forms
blob = job.blob
import ctypes as C
pin_in = pinned_array(len(blob)); pin_in[:] = np.frombuffer(blob, dtype=np.uint8)
status = (C.c_int32 * job.n)()
L = _capi.lib()
j2 = Job.hca_decode([uniq[i % 8] for i in range(N)], keys=[KEY] * N)      # same layout; used through the blob entry point
def run_blob(src_ptr, out):
    rc = L.cri_job_run_host_into(j2._h, src_ptr, out.ctypes.data, status)
    assert rc == 0
    return j2.split(memoryview(out)), np.array(status[:N])
for env in ("0", str(1 << 62)):
    os.environ["CRICODECS_HOST_SLICE_MIN"] = env
    tag = "pipelined" if env == "0" else "one piece"
    timed("blob (pageable) -> pinned out, " + tag, lambda: run_blob(blob, pin))
    timed("blob (pinned)   -> pinned out, " + tag, lambda: run_blob(pin_in.ctypes.data, pin))
os.environ["CRICODECS_HOST_SLICE_MIN"] = "0"
os.environ["CRICODECS_HOST_DOWN_ON_RUN"] = "1"
for k in ("8", "16", "32"):
    os.environ["CRICODECS_HOST_SLICES"] = k
    timed("blob (pinned) -> pinned out, %s slices, downloads on the run stream" % k, lambda: run_blob(pin_in.ctypes.data, pin))
    timed("blob (pageable) -> pinned out, %s slices, downloads on the run stream" % k, lambda: run_blob(blob, pin))
    timed("items -> pinned out, %s slices, downloads on the run stream" % k, lambda: job.run_host(out=pin))
