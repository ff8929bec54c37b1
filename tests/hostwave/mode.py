"""hostwave (TEST INFRASTRUCTURE): what a test process does when CRI_TEST_HOSTWAVE=1 points it at the EMULATED build of the library
(CRICODECS_LIB_DIR=tests/hostwave/lib: the product's kernel sources compiled for x86-64 under a lockstep wave64 emulator whose "device
memory" is host memory).  Device buffers become CPU tensors, the stream is the null stream, synchronize() has nothing to wait for.
Set by tests/test_hostwave.py for the pytest / soak processes it starts; never on a GPU box."""
import os

ON = os.environ.get("CRI_TEST_HOSTWAVE") == "1"


def enable():
    if not ON:
        return False
    import torch
    from pycricodecs_amd import batch
    if getattr(batch.Job, "_hostwave", False):
        return True
    alloc = batch.Job.alloc

    def alloc_on_host(self, device="cuda:0", upload=True):
        bufs = alloc(self, "cpu", upload)
        bufs[2].fill_(0xCD)                        # scratch is `torch.empty` on the device: nothing may depend on what it held
        return bufs
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
