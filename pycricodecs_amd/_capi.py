"""ctypes binding of lib/libcricodecs_hip.so (C ABI: include/cricodecs_hip.h).

The library is the only implementation of the codec work; if it is missing or no gfx950 device is present the
calls fail loudly (CriCodecsError / OSError) -- there is no Python or CPU fallback.
"""
import ctypes as C
import os
import subprocess

try:  # load torch's bundled HIP runtime first so both sides share one libamdhip64 (same soname)
    if os.environ.get("CRICODECS_NO_TORCH") == "1":      # (the AddressSanitizer run: its allocator hooks need the system's HIP runtime)
        raise ImportError
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.environ.get("CRICODECS_LIB_DIR") or os.path.join(HERE, "lib")      # (the AddressSanitizer build lives in a directory of its own: tools/asan_gpu.sh)
LIB_PATH = os.path.join(LIB_DIR, "libcricodecs_hip.so")
TESTING_LIB_PATH = os.path.join(LIB_DIR, "libcricodecs_hip_testing.so")   # the same sources + the parity tests' knobs (cri_test_set); never shipped

SYMBOLS = [
    "cri_adx_decode", "cri_adx_encode", "cri_hca_decode", "cri_hca_encode", "cri_hca_crypt", "cri_free", "cri_strerror",
    "cri_device_available", "cri_job_create_hca_decode", "cri_job_create_adx_decode", "cri_job_create_adx_encode",
    "cri_job_create_hca_encode", "cri_job_create_hca_crypt", "cri_job_kind", "cri_job_items", "cri_job_input_bytes",
    "cri_job_output_bytes", "cri_job_output_offsets", "cri_job_host_status", "cri_job_scratch_bytes", "cri_job_units",
    "cri_job_units2", "cri_job_algorithmic_bytes", "cri_job_run", "cri_job_dominant_kernel", "cri_job_destroy", "cri_job_run_host", "cri_job_enable_events",
    "cri_job_event_ms", "cri_awb_index", "cri_job_create_awb_decode", "cri_job_run_host_into",
    "cri_job_create_hca_decode_items", "cri_job_create_adx_decode_items", "cri_job_create_adx_encode_items",
    "cri_job_create_hca_encode_items", "cri_job_create_hca_crypt_items", "cri_job_input_offsets", "cri_device_count", "cri_set_device", "cri_get_device", "cri_job_device", "cri_job_run_floats", "cri_job_float_count", "cri_job_float_offsets",
    "cri_usm_audio_mask", "cri_usm_index", "cri_job_create_usm_audio_demux", "cri_job_create_sfa_pack", "cri_job_item_tags", "cri_job_item_sizes",
    "cri_job_hca_groups", "cri_job_run_host_items", "cri_pinned_alloc", "cri_pinned_free", "cri_release_cache", "cri_build_id",
]


class UsmChunk(C.Structure):
    _fields_ = [("fourcc", C.c_char * 4), ("chno", C.c_uint32), ("type", C.c_uint32), ("padding", C.c_uint32),
                ("payload_offset", C.c_uint64), ("payload_len", C.c_uint32), ("frame_time", C.c_uint32),
                ("frame_rate", C.c_uint32), ("pad", C.c_uint32)]


class HcaGroupInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("channels", "frames", "record_bytes", "flags_offset", "narrow_flag", "narrow_capable", "plain", "transform_form")] + \
               [("first_record_offset", C.c_uint64), ("lines_offset", C.c_uint64), ("code_desc_offset", C.c_uint64)]


class AdxEncodeParams(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("bitdepth", "blocksize", "encoding_mode", "highpass_frequency", "filter",
                                           "adx_version", "force_no_looping")]


_lib = None
_product = None
_testing = None


def lib():
    """The library every call of this package goes through: the shipped one, unless a test has switched to the testing build."""
    global _lib, _product
    if _lib is not None:
        return _lib
    if _product is None:
        _product = _bind(LIB_PATH)
    _lib = _product
    return _lib


class testing_knobs:
    """Context manager for the parity tests: routes the package through libcricodecs_hip_testing.so (the same sources built with
    -DCRI_TESTING) and sets planner knobs that the shipped library does not expose (cri_capi.cpp, struct Knobs) -- e.g.
    testing_knobs(adx_mapping="seg", adx_warm_pct=1).  Everything is restored on exit."""
    ADX_MAPPING = {"auto": 0, "chain": 1, "file": 2, "seg": 3, "lane": 4, "wave": 5}

    def __init__(self, **knobs):
        self.knobs = knobs

    def __enter__(self):
        global _lib, _testing
        if _testing is None:
            _testing = _bind(TESTING_LIB_PATH)
            _testing.cri_test_set.argtypes = [C.c_char_p, C.c_longlong]
        self.prev = _lib
        _lib = _testing
        _testing.cri_test_set(None, 0)
        for k, v in self.knobs.items():
            if k == "adx_mapping":
                v = self.ADX_MAPPING[v]
            rc = _testing.cri_test_set(k.encode(), int(v))
            assert rc == 0, "unknown knob %r" % k
        return _testing

    def __exit__(self, *exc):
        global _lib
        _testing.cri_test_set(None, 0)
        _lib = self.prev
        return False


def _fresh(path):
    """A prebuilt library is only loaded if it was built from THIS tree's sources (build.source_id() against the id compiled into the
    file): prebuilt binaries travel with the tree, file times do not mean anything there.  A stale or missing one is rebuilt on the
    spot (hipcc cross-compiles anywhere); CRICODECS_NO_REBUILD=1 turns that into an error instead."""
    from . import build as B
    if not os.path.isdir(B.CSRC):                              # an installed copy without sources: nothing to hold it to
        return
    want = B.source_id()
    if os.path.exists(path) and B.embedded_id(path) == want:
        return
    if os.environ.get("CRICODECS_NO_REBUILD") == "1":
        raise OSError("%s is not built from this tree (library %s, sources %s): run `python -m pycricodecs_amd.build`" % (path, B.embedded_id(path), want))
    import sys
    have = B.embedded_id(path)
    print("pycricodecs_amd: %s is stale or missing (library %s, sources %s): rebuilding" % (os.path.basename(path), have, want), file=sys.stderr, flush=True)
    try:
        B.build(verbose=False)                                 # (serialised across processes by an flock on the library directory; outputs are renamed into place)
    except (OSError, subprocess.SubprocessError) as e:
        raise OSError("%s is not built from this tree (library %s, sources %s) and rebuilding it failed: %s -- run `python -m pycricodecs_amd.build` where hipcc is"
                      % (path, have, want, e)) from e


def _bind(path):
    _fresh(path)
    if not os.path.exists(path):
        raise OSError("%s not built: run `python -m pycricodecs_amd.build` (hipcc, gfx950)" % path)
    L = C.CDLL(path)
    if hasattr(L, "hostwave_divergent_completions") and os.environ.get("CRI_TEST_HOSTWAVE") != "1":
        # tests/hostwave builds the same sources for the CPU under a wave64 emulator, under the same file name, for the parity tests:
        # never a way to run the product (CRICODECS_LIB_DIR pointing there by accident must not turn into a silent CPU path)
        raise OSError("%s is the emulated TEST build of the library (tests/hostwave), not the HIP library: refusing to use it outside the test suite" % path)
    L.cri_build_id.argtypes = []
    L.cri_build_id.restype = C.c_char_p
    u8p, u64p, i32p, szp = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.POINTER(C.c_int32), C.POINTER(C.c_size_t)
    vp = C.c_void_p
    L.cri_adx_decode.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(u8p), szp]
    L.cri_adx_encode.argtypes = [C.c_char_p, C.c_size_t] + [C.c_uint32] * 6 + [C.c_int, C.POINTER(u8p), szp]
    L.cri_hca_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint64, C.c_uint16, C.POINTER(u8p), szp]
    L.cri_hca_encode.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint32, C.POINTER(u8p), szp]
    L.cri_hca_crypt.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint16]
    L.cri_free.argtypes = [vp]
    L.cri_free.restype = None
    L.cri_strerror.argtypes = [C.c_int]
    L.cri_strerror.restype = C.c_char_p
    L.cri_job_create_hca_decode.argtypes = [vp, u64p, C.c_uint32, u64p, C.POINTER(C.c_uint16), C.POINTER(vp)]
    L.cri_job_create_adx_decode.argtypes = [vp, u64p, C.c_uint32, C.POINTER(vp)]
    L.cri_job_create_adx_encode.argtypes = [vp, u64p, C.c_uint32, C.POINTER(AdxEncodeParams), C.POINTER(vp)]
    L.cri_job_create_hca_encode.argtypes = [vp, u64p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(vp)]
    L.cri_job_create_hca_crypt.argtypes = [vp, u64p, C.c_uint32, C.c_uint32, C.c_uint32, u64p, C.POINTER(C.c_uint16), C.POINTER(vp)]
    for name in ("cri_job_kind", "cri_job_items"):
        getattr(L, name).argtypes = [vp]
        getattr(L, name).restype = C.c_uint32
    for name in ("cri_job_input_bytes", "cri_job_output_bytes", "cri_job_scratch_bytes", "cri_job_units", "cri_job_units2",
                 "cri_job_algorithmic_bytes"):
        getattr(L, name).argtypes = [vp]
        getattr(L, name).restype = C.c_uint64
    L.cri_job_output_offsets.argtypes = [vp]
    L.cri_job_output_offsets.restype = u64p
    L.cri_job_host_status.argtypes = [vp]
    L.cri_job_host_status.restype = i32p
    L.cri_job_dominant_kernel.argtypes = [vp]
    L.cri_job_dominant_kernel.restype = C.c_char_p
    L.cri_job_run.argtypes = [vp, vp, vp, vp, vp, vp]
    L.cri_job_create_hca_decode_items.argtypes = [vp, u64p, C.POINTER(C.c_uint16), C.POINTER(vp)]
    L.cri_job_create_adx_decode_items.argtypes = [vp, C.POINTER(vp)]
    L.cri_job_create_adx_encode_items.argtypes = [vp, C.POINTER(AdxEncodeParams), C.POINTER(vp)]
    L.cri_job_create_hca_encode_items.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(vp)]
    L.cri_job_create_hca_crypt_items.argtypes = [vp, C.c_uint32, C.c_uint32, u64p, C.POINTER(C.c_uint16), C.POINTER(vp)]
    L.cri_job_hca_groups.argtypes = [vp, C.POINTER(HcaGroupInfo), C.c_int]
    L.cri_job_input_offsets.argtypes = [vp]
    L.cri_job_input_offsets.restype = u64p
    L.cri_set_device.argtypes = [C.c_int]
    L.cri_job_device.argtypes = [vp]
    L.cri_job_run_floats.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.cri_job_float_count.argtypes = [vp]
    L.cri_job_float_count.restype = C.c_uint64
    L.cri_job_float_offsets.argtypes = [vp]
    L.cri_job_float_offsets.restype = u64p
    L.cri_job_run_host.argtypes = [vp, vp, C.POINTER(u8p), i32p]
    L.cri_job_run_host_into.argtypes = [vp, vp, vp, i32p]
    L.cri_job_run_host_items.argtypes = [vp, vp, vp, i32p]
    L.cri_pinned_alloc.argtypes = [C.c_size_t]
    L.cri_pinned_alloc.restype = vp
    L.cri_pinned_free.argtypes = [vp]
    L.cri_pinned_free.restype = None
    L.cri_release_cache.argtypes = []
    L.cri_release_cache.restype = None
    L.cri_job_enable_events.argtypes = [vp, C.c_int]
    L.cri_job_event_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_char_p), C.c_int]
    L.cri_job_destroy.argtypes = [vp]
    L.cri_awb_index.argtypes = [vp, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint16), C.POINTER(C.c_uint32), u64p, u8p, C.c_uint32]
    L.cri_job_create_awb_decode.argtypes = [vp, C.c_size_t, C.c_uint64, C.POINTER(vp), C.POINTER(vp)]
    L.cri_job_destroy.restype = None
    L.cri_usm_audio_mask.argtypes = [C.c_uint64, u8p]
    L.cri_usm_index.argtypes = [vp, C.c_size_t, C.POINTER(UsmChunk), C.c_uint32, C.POINTER(C.c_uint32)]
    L.cri_job_create_usm_audio_demux.argtypes = [vp, C.c_size_t, C.c_uint64, C.c_uint32, C.POINTER(vp)]
    L.cri_job_create_sfa_pack.argtypes = [vp, u64p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.POINTER(vp)]
    L.cri_job_item_tags.argtypes = [vp]
    L.cri_job_item_tags.restype = C.POINTER(C.c_uint32)
    L.cri_job_item_sizes.argtypes = [vp]
    L.cri_job_item_sizes.restype = C.POINTER(C.c_uint64)
    return L


class CriCodecsError(Exception):
    """Library-domain failure (no device, HIP error, unsupported-on-device input)."""
    def __init__(self, code):
        self.code = code
        super().__init__("%s (code %d)" % (strerror(code), code))


def build_id():
    """The id of the sources the loaded library was built from (cri_build_id)."""
    return lib().cri_build_id().decode()


def strerror(code):
    return lib().cri_strerror(code).decode()


def raise_for(code):
    """Map a return code to the exception type + message the reference extension raises
    (adx.cpp:32-38, pcm.cpp:35-38, hca.cpp:3252-3268)."""
    if code == 0:
        return
    if code == -3 or code in (-411, -412):                 # usm.py:130, 189 raise NotImplementedError too
        raise NotImplementedError(strerror(code))
    if -18 <= code <= -1 or -110 <= code <= -101 or -216 <= code <= -201:
        raise ValueError(strerror(code))
    raise CriCodecsError(code)


def take(out, n):
    data = C.string_at(out, n.value)
    lib().cri_free(out)
    return data
