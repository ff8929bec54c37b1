import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # CRI_TEST_HOSTWAVE=1 (tests/test_hostwave.py, never a GPU box): the `-m gpu` parity tests on the emulated build of the library
    sys.path.insert(0, os.path.join(ROOT, "tests", "hostwave"))
    import mode
    mode.enable()


@pytest.fixture
def knobs():
    """knobs(adx_mapping="seg", adx_warm_pct=1, ...): for the rest of the test the package runs on the TESTING build of the library
    (pycricodecs_amd/lib/libcricodecs_hip_testing.so: the same sources with -DCRI_TESTING) with those planner knobs set, so that
    the parity tests can push work onto the repair passes and the general kernels.  The shipped library has no such entry point."""
    from pycricodecs_amd import _capi
    ctx = _capi.testing_knobs()
    L = ctx.__enter__()

    def set_knobs(**kw):
        for k, v in kw.items():
            if k == "adx_mapping":
                v = ctx.ADX_MAPPING[v]
            assert L.cri_test_set(k.encode(), int(v)) == 0, "unknown knob %r" % k
    yield set_knobs
    ctx.__exit__(None, None, None)
