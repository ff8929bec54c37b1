"""hostwave (TEST INFRASTRUCTURE): what a test process does when CRI_TEST_HOSTWAVE=1 points it at the EMULATED build of the library
(CRICODECS_LIB_DIR=tests/hostwave/lib: the product's kernel sources compiled for x86-64 under a lockstep wave64 emulator whose "device
memory" is host memory).  Device buffers become CPU tensors, the stream is the null stream, synchronize() has nothing to wait for.
Set by tests/test_hostwave.py for the pytest / soak processes it starts; never on a GPU box."""
import os

ON = os.environ.get("CRI_TEST_HOSTWAVE") == "1"


_keep = []


def guarded(nbytes, dtype):
    """uint8 tensor of nbytes whose storage is fenced by PROT_NONE pages (never unmapped: test processes are short-lived)."""
    import ctypes
    import mmap
    import torch
    page = mmap.PAGESIZE
    body = (nbytes + 63) // 64 * 64
    total = (body + page - 1) // page * page + 2 * page
    libc = ctypes.CDLL(None, use_errno=True)
    libc.mmap.restype = ctypes.c_void_p
    libc.mmap.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long]
    libc.mprotect.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    base = libc.mmap(None, total, mmap.PROT_READ | mmap.PROT_WRITE, mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS, -1, 0)
    assert base and base != ctypes.c_void_p(-1).value
    assert libc.mprotect(base, page, 0) == 0 and libc.mprotect(base + total - page, page, 0) == 0
    start = base + total - page - body
    arr = (ctypes.c_uint8 * nbytes).from_address(start)
    _keep.append(arr)
    return torch.frombuffer(arr, dtype=dtype)


def enable():
    if not ON:
        return False
    import torch
    from pycricodecs_amd import batch
    if getattr(batch.Job, "_hostwave", False):
        return True
    alloc = batch.Job.alloc

    def alloc_on_host(self, device="cuda:0", upload=True):
        if os.environ.get("HOSTWAVE_GUARD") != "1":
            # (256-byte aligned like a device allocation: the traffic census counts 128-byte lines)
            def al(n, dtype=torch.uint8, fill=None):
                raw = torch.empty(n * torch.empty((), dtype=dtype).element_size() + 256, dtype=torch.uint8)
                off = (-raw.data_ptr()) % 256
                t = raw[off:off + n * torch.empty((), dtype=dtype).element_size()].view(dtype)
                if fill is not None:
                    t.fill_(fill)
                return t
            d_in = al(max(self.input_bytes, 1))
            if upload and self.input_bytes:
                d_in.zero_()
                self.upload(d_in)
            # scratch is `torch.empty` on the device: nothing may depend on what it held
            return d_in, al(max(self.output_bytes, 1), fill=0), al(max(self.scratch_bytes, 1), fill=0xCD), al(max(self.n, 1), torch.int32, fill=0)
        # HOSTWAVE_GUARD=1: every device buffer of the job between two inaccessible pages, its END (rounded up to 64 bytes) on the
        # page boundary -- a kernel that reads or writes past a buffer (or before it) dies there, with the emulator's report
        d_in = guarded(max(self.input_bytes, 1), torch.uint8)
        if upload and self.input_bytes:
            d_in.zero_()
            self.upload(d_in)
        d_out = guarded(max(self.output_bytes, 1), torch.uint8); d_out.zero_()
        d_scratch = guarded(max(self.scratch_bytes, 1), torch.uint8); d_scratch.fill_(0xCD)
        d_status = guarded(max(self.n, 1) * 4, torch.uint8).view(torch.int32); d_status.zero_()
        return d_in, d_out, d_scratch, d_status
    batch.Job.alloc = alloc_on_host
    batch.Job._hostwave = True

    class _NullStream:
        cuda_stream = 0

        def __init__(self, *a, **k):
            pass

        def wait_stream(self, other):
            pass

        def synchronize(self):
            pass
    torch.cuda.current_stream = lambda *a, **k: _NullStream()
    torch.cuda.Stream = _NullStream                # (side streams of the Python layer, awb.py: everything is synchronous here)
    torch.cuda.synchronize = lambda *a, **k: None
    return True
