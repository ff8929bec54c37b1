"""Developer tool (GPU box): PCIe rates with page-locked host memory -- H2D alone, D2H alone, both at once on two streams."""
import time, torch
dev = "cuda:0"
GB = 1 << 30
h_in = torch.empty(1 * GB, dtype=torch.uint8).pin_memory(); h_out = torch.empty(4 * GB, dtype=torch.uint8).pin_memory()
d_in = torch.empty(1 * GB, dtype=torch.uint8, device=dev); d_out = torch.empty(4 * GB, dtype=torch.uint8, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def t(fn):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return time.perf_counter() - t0
def h2d():
    with torch.cuda.stream(s1): d_in.copy_(h_in, non_blocking=True)
def d2h():
    with torch.cuda.stream(s2): h_out.copy_(d_out, non_blocking=True)
def d2h_chunks(n=16):
    c = h_out.numel() // n
    with torch.cuda.stream(s2):
        for k in range(n): h_out[k * c:(k + 1) * c].copy_(d_out[k * c:(k + 1) * c], non_blocking=True)
a = t(h2d); b = t(d2h); c = t(lambda: (h2d(), d2h())); d = t(d2h_chunks); e = t(lambda: (h2d(), d2h_chunks()))
print("H2D 1 GB: %.1f ms (%.1f GB/s); D2H 4 GB: %.1f ms (%.1f GB/s); both at once: %.1f ms; D2H in 16 chunks: %.1f ms; H2D + chunks: %.1f ms" % (a * 1e3, 1.074 / a, b * 1e3, 4.295 / b, c * 1e3, d * 1e3, e * 1e3))
# the pipelined host path's shape: upload chunks (events) -> a kernel per chunk on a third stream (events) -> download chunks
s3 = torch.cuda.Stream()
def pipeline(n=16, up_chunks=True):
    ci, co = h_in.numel() // n, h_out.numel() // n
    evs = []
    with torch.cuda.stream(s1):
        if not up_chunks: d_in.copy_(h_in, non_blocking=True)
        for k in range(n):
            if up_chunks: d_in[k * ci:(k + 1) * ci].copy_(h_in[k * ci:(k + 1) * ci], non_blocking=True)
            e = torch.cuda.Event(); e.record(s1); evs.append(e)
    for k in range(n):
        s3.wait_event(evs[k])
        with torch.cuda.stream(s3):
            d_out[k * co:(k + 1) * co].add_(1)
            e = torch.cuda.Event(); e.record(s3)
        s2.wait_event(e)
        with torch.cuda.stream(s2): h_out[k * co:(k + 1) * co].copy_(d_out[k * co:(k + 1) * co], non_blocking=True)
print("pipeline (16 up chunks, kernel, 16 down chunks): %.1f ms; one upload: %.1f ms; 64 chunks: %.1f ms" % (t(pipeline) * 1e3, t(lambda: pipeline(16, False)) * 1e3, t(lambda: pipeline(64)) * 1e3))
# the same pipeline with the library's page-locked memory (cri_pinned_alloc) instead of torch's
import sys
sys.path.insert(0, ".")
from pycricodecs_amd.batch import pinned_array
h_in = torch.from_numpy(pinned_array(1 * GB)); h_out = torch.from_numpy(pinned_array(4 * GB))
h_in.fill_(1); h_out.fill_(0)
print("cri_pinned_alloc memory: pipeline %.1f ms; H2D alone %.1f ms; D2H alone %.1f ms" % (t(pipeline) * 1e3, t(h2d) * 1e3, t(d2h) * 1e3))
