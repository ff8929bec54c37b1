"""The float shortcuts inside the encoders, enumerated over their WHOLE domains on the CPU.

`csrc/cri_adx_quant.h` (the ADX encoders' float quantisers) and `csrc/cri_hca_enc_cost.h` (the HCA encoder's class / rank band cost)
replace integer and table rules of the reference by a few float operations; they are shared, as headers, by the product kernels and
by the device-side exhaustive tests (tests/test_gpu_adx.py::test_adx_float_quantisers_exhaustive,
tests/test_gpu_hca_encode.py::test_hca_encoder_band_cost_rule_on_the_device).  The same headers compile for the host
(tests/shim/device_fn_host.cpp: plain IEEE binary32, explicit fma where one is meant, -ffp-contract=off), so the same enumeration runs
here without a GPU: every delta in [-2^18, 2^18) x every scale 1 .. 4096 and 8192 per bit depth (2.1 G cases each), and every
magnitude 0 .. 0.9999999f of both signs at all fifteen resolutions (2.1 G floats x 15)."""
import ctypes as C
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pycricodecs_amd", "csrc")


@pytest.fixture(scope="module")
def fn():
    src = os.path.join(ROOT, "tests", "shim", "device_fn_host.cpp")
    out = os.path.join(ROOT, "tests", "shim", "libdevice_fn_host.so")
    deps = [src] + [os.path.join(CSRC, h) for h in ("cri_adx_quant.h", "cri_hca_enc_cost.h", "cri_bits.h", "cri_types.h", "cri_tables.h")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-mfma", "-fPIC", "-shared", "-pthread", "-Wall", "-I/opt/rocm/include", src, "-o", out], check=True)
    L = C.CDLL(out)
    L.host_adx_quantisers.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_ulonglong)] * 2 + [C.POINTER(C.c_int32)]
    L.host_enc_band_cost.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.POINTER(C.c_uint32)]
    return L


@pytest.mark.parametrize("form,bitdepth", [(0, b) for b in range(2, 9)] + [(1, 4)], ids=lambda v: str(v))
def test_adx_float_quantisers_over_every_delta_and_scale(fn, form, bitdepth):
    """adx.cpp:256-261 (delta +- scale / 2, C division, clamp) against AdxQuantSmall (form 0) / AdxQuantLane (form 1)."""
    cases, bad, first = C.c_ulonglong(), C.c_ulonglong(), (C.c_int32 * 4)()
    assert fn.host_adx_quantisers(form, bitdepth, -(1 << 18), (1 << 18) - 1, C.byref(cases), C.byref(bad), first) == 0
    assert cases.value == 4097 * (1 << 19)
    assert bad.value == 0, "delta %d scale %d: got %d, the reference's rule gives %d" % tuple(first)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("part", range(4))
def test_hca_encoder_band_cost_over_every_float(fn, part):
    """hca.cpp:2771-2786 (quantise every spectrum, look its code length up; the dead zone from resolution 8 on) against the class / rank
    rule with the clamp anomaly, tables from the product's own builder (hca_enc_build_tables through the host shim)."""
    import test_host_logic as H
    shim = H.shim.__wrapped__() if hasattr(H.shim, "__wrapped__") else None
    assert shim is not None
    tab = (C.c_uint8 * 16384)()
    shim.shim_hca_enc_tables.argtypes = [C.c_void_p, C.c_size_t]
    assert shim.shim_hca_enc_tables(tab, 16384) > 0
    clamp = 0x3F7FFFFE
    bands = (clamp + 8) // 8
    lo, hi = bands * part // 4, bands * (part + 1) // 4
    cases, bad, first = C.c_ulonglong(), C.c_ulonglong(), (C.c_uint32 * 4)()
    assert fn.host_enc_band_cost(tab, 8 * lo, 1, hi - lo, C.byref(cases), C.byref(bad), first) == 0
    assert bad.value == 0, "first spectrum %08x, resolution %d: %d bits, the reference's rule gives %d" % tuple(first)
    assert cases.value == (hi - lo) * 2 * 15


def test_table_free_crc16_accepts_exactly_what_the_reference_checksum_accepts(fn):
    """csrc/cri_bits.h (crcq_word / crcq_byte / crcq_fold: remainder modulo x^15 + x + 1, 32 message bits per step, + the parity
    modulo x + 1) as k_hca_parse applies it, against the reference's CRC-16 (hca.cpp:186-211 through the oracle): a message with its
    checksum appended is accepted at every length 2 .. 2100 (word and byte tails), and any single flipped bit, any byte changed and
    any 16-bit burst is rejected -- what a CRC-16 guarantees."""
    import numpy as np
    import oracle_lib as O
    fn.host_crc_accepts.argtypes = [C.c_char_p, C.c_size_t]
    rng = np.random.default_rng(5)
    for n in list(range(0, 70)) + [680, 681, 682, 683, 1021, 2098]:
        body = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        msg = body + O.crc16(body).to_bytes(2, "big")
        assert fn.host_crc_accepts(msg, len(msg)) == 1, n
        assert O.crc16(msg) == 0
        for _ in range(24):
            m = bytearray(msg)
            kind = rng.integers(0, 3)
            pos = int(rng.integers(0, len(m)))
            if kind == 0:
                m[pos] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1:
                m[pos] = (m[pos] + int(rng.integers(1, 256))) & 0xFF
            else:
                burst = int(rng.integers(1, 1 << 16))
                m[pos] ^= burst >> 8
                if pos + 1 < len(m):
                    m[pos + 1] ^= burst & 0xFF
            if bytes(m) != msg:
                assert fn.host_crc_accepts(bytes(m), len(m)) == 0, (n, kind, pos)
    # and on random garbage the two agree about validity (the rule is "divisible by the polynomial", not "equal to a stored field")
    agree = 0
    for _ in range(20000):
        m = rng.integers(0, 256, int(rng.integers(2, 40)), dtype=np.uint8).tobytes()
        assert (fn.host_crc_accepts(m, len(m)) == 1) == (O.crc16(m) == 0)
        agree += 1
    assert agree == 20000


def test_noise_generator_jump_ahead_equals_single_steps(fn):
    """lcg_jump (cri_bits.h) against r' = 0x343FD r + 0x269EC3 applied n times (hca.cpp:1616), n = 0 .. 20 000 from 64 start values,
    and jumps compose (a then b = a + b, wrapping)."""
    fn.host_lcg_jump_mismatches.restype = C.c_ulonglong
    fn.host_lcg_jump_mismatches.argtypes = [C.c_uint32, C.c_uint32]
    assert fn.host_lcg_jump_mismatches(64, 20000) == 0
