"""Developer tool (GPU box): cost of page-locking caller memory (hipHostRegister / hipHostUnregister) against copying it into
page-locked staging memory, for one blob and for many separate items."""
import ctypes as C, time, numpy as np, torch
hip = C.CDLL("libamdhip64.so")
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]; hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
torch.zeros(1, device="cuda:0")
def reg_cost(buf, n):
    p = buf.ctypes.data
    t0 = time.perf_counter(); r = hip.hipHostRegister(p, n, 0); t1 = time.perf_counter(); u = hip.hipHostUnregister(p); t2 = time.perf_counter()
    assert r == 0 and u == 0, (r, u)
    return t1 - t0, t2 - t1
for mb in (40, 640, 3200):
    a = np.ones(mb << 20, dtype=np.uint8)
    for rep in range(3):
        r, u = reg_cost(a, a.size)
        print("blob %5d MB: register %8.2f ms (%.1f GB/s), unregister %8.2f ms" % (mb, r * 1e3, a.size / r / 1e9, u * 1e3))
items = [np.ones(326000 + 64 * (i % 7), dtype=np.uint8) for i in range(2000)]
for rep in range(2):
    t0 = time.perf_counter()
    for it in items: assert hip.hipHostRegister(it.ctypes.data, it.size, 0) == 0
    t1 = time.perf_counter()
    for it in items: assert hip.hipHostUnregister(it.ctypes.data) == 0
    t2 = time.perf_counter()
    print("2000 items of 326 KB: register %.1f ms, unregister %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
stage = C.c_void_p(); assert hip.hipHostMalloc(C.byref(stage), 64 << 20, 0) == 0
st = np.ctypeslib.as_array(C.cast(stage, C.POINTER(C.c_uint8)), shape=(64 << 20,))
big = np.ones(640 << 20, dtype=np.uint8)
for rep in range(2):
    t0 = time.perf_counter()
    for k in range(10): C.memmove(stage, big.ctypes.data + (k * 64 << 20), 64 << 20)
    dt = time.perf_counter() - t0
    print("memcpy 640 MB blob -> page-locked staging: %.1f ms (%.1f GB/s)" % (dt * 1e3, 0.671 / dt))
    t0 = time.perf_counter(); pos = 0
    for it in items:
        if pos + it.size > (64 << 20): pos = 0
        C.memmove(stage.value + pos, it.ctypes.data, it.size); pos += it.size
    dt = time.perf_counter() - t0
    print("memcpy 2000 items -> staging: %.1f ms (%.1f GB/s)" % (dt * 1e3, sum(i.size for i in items) / dt / 1e9))
