"""Developer tool (GPU box): ADX decode from host memory to host memory (Job.run_host) against the link rate."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import bench as B
import oracle_lib as O
from pycricodecs_amd import _capi
from pycricodecs_amd.batch import Job, pinned_array
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
wgs = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
uniq = [O.adx_encode(B.family_wav(3000 + u, 10.0, "tonal")) for u in range(8)]
refs = [O.adx_decode(a) for a in uniq]
items = B.tile(uniq, n)
for wg in wgs:
    ctx = _capi.testing_knobs(host_pull_wgs=wg) if wg else None
    if ctx: ctx.__enter__()
    job = Job.adx_decode(items)
    pin_in = pinned_array(job.input_bytes)
    for label, out, joined in (("items -> pageable", np.zeros(job.output_bytes, dtype=np.uint8), False), ("blob -> pinned", pinned_array(job.output_bytes), True)):
        job.run_host(out=out, joined=joined)
        best = None
        for _ in range(4):
            t0 = time.perf_counter(); outs, st = job.run_host(out=out, joined=joined); dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        assert not st.any() and bytes(outs[5]) == refs[5] and bytes(outs[n - 1]) == refs[(n - 1) % 8]
        print("%-20s %d files, pull workgroups %d: %.2f ms; in %.2f GB out %.2f GB; out / 57 GB/s = %.2f ms -> %.0f %% of the link rate; %.2f G frames/s" % (
            label, n, wg, best * 1e3, job.input_bytes / 1e9, job.output_bytes / 1e9, job.output_bytes / 57e9 * 1e3, 100 * job.output_bytes / 57e9 / best, job.units / best / 1e9), flush=True)
    del job
    if ctx: ctx.__exit__(None, None, None)
