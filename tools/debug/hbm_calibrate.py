"""Developer tool (GPU box, under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE): streams 2 GiB through the testing library's read and write
kernels at 4 / 8 / 16 bytes per lane (cri_test_stream), so that the counters can be scaled per access width (tools/prof_r04.sh)."""
import ctypes as C, sys
sys.path.insert(0, ".")
from pycricodecs_amd import _capi
with _capi.testing_knobs() as L:
    L.cri_test_stream.argtypes = [C.c_uint64, C.c_int]
    assert L.cri_test_stream(2 << 30, 3) == 0
print("streamed", 2 << 30, "bytes x 3 per kernel")
