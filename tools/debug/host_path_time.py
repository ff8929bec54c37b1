"""Developer tool (GPU box): PCIe-inclusive rate of the host-buffer batch call (cri_job_run_host: malloc, H2D, kernels, D2H, free)."""
import ctypes as C
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import oracle_lib as O
from pycricodecs_amd import synth, _capi
from pycricodecs_amd.batch import Job
KEY = 0xCF222F1FE0748978
uniq = [O.hca_crypt(O.hca_encode(synth.wav(i, 480000, 2, 48000), 1), 1, 56, KEY) for i in range(4)]
items = [uniq[i % 4] for i in range(1000)]
job = Job.hca_decode(items, keys=[KEY] * len(items))
job.run_host()
t0 = time.perf_counter()
outs, st = job.run_host()
dt = time.perf_counter() - t0
assert not st.any() and bytes(outs[0]) == O.hca_decode(uniq[0], KEY)
print("Job.run_host: %d frames, %.2f GB in + %.2f GB out in %.1f ms -> %.2f M frames/s PCIe-inclusive" % (
    job.units, job.input_bytes / 1e9, job.output_bytes / 1e9, dt * 1e3, job.units / dt / 1e6))
# the C call alone
out = C.POINTER(C.c_uint8)(); status = (C.c_int32 * job.n)()
t0 = time.perf_counter()
rc = _capi.lib().cri_job_run_host(job._h, job.blob, C.byref(out), status)
dt = time.perf_counter() - t0
_capi.lib().cri_free(out)
print("cri_job_run_host alone: %.1f ms -> %.2f M frames/s" % (dt * 1e3, job.units / dt / 1e6))
