"""Host-side header logic of the product (pycricodecs_amd/csrc/cri_host.cpp) driven on the CPU through a test-only shim
(tests/shim/host_shim.cpp, plain g++): the planner that calls it in the library needs a device to build a job.
Covers the round-1 advisor findings (a RIFF chunk length that wraps, header walks that stay inside the header) and a
mutation fuzz of the HCA header walk against the oracle's restatement."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

import hca_forge
import oracle_lib as O
from pycricodecs_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def shim():
    src = os.path.join(ROOT, "tests", "shim", "host_shim.cpp")
    host = os.path.join(ROOT, "pycricodecs_amd", "csrc", "cri_host.cpp")
    san = os.environ.get("CRI_TEST_SANITIZED") == "1"           # tests/test_sanitizers.py: the product's host logic under ASan + UBSan
    out = os.path.join(ROOT, "tests", "shim", "libhost_shim_san.so" if san else "libhost_shim.so")
    deps = [src, host, os.path.join(ROOT, "pycricodecs_amd", "csrc", "cri_host.h"), os.path.join(ROOT, "pycricodecs_amd", "csrc", "cri_types.h"), os.path.join(ROOT, "pycricodecs_amd", "csrc", "cri_tables.h")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        flags = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"] if san else []
        subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-Wall"] + flags + [src, host, "-o", out], check=True)
    L = C.CDLL(out)
    L.shim_wav_parse.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32)]
    L.shim_hca_parse_header.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.POINTER(C.c_uint32)]
    L.shim_hca_crypt_header.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
    L.shim_adx_parse_header.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32)]
    return L


def wav_parse(L, b):
    out = (C.c_uint32 * 10)()
    return L.shim_wav_parse(b, len(b), out), list(out)


@pytest.mark.timeout(30)
def test_wav_chunk_length_that_wraps_terminates(shim):
    """A chunk length of 0xFFFFFFF8 makes the reference's 32-bit `size = len + 8` zero: neither cursor advances
    (pcm.cpp:295-296, 322).  The product (and the oracle) reject every wrapped length with the header error."""
    w = bytearray(synth.wav(1, 64, 2, 48000))
    junk = bytearray(b"JUNK" + struct.pack("<I", 0xFFFFFFF8) + bytes(8))
    crafted = bytes(w[:12]) + bytes(junk) + bytes(w[12:])
    crafted = crafted[:4] + struct.pack("<I", len(crafted) - 8) + crafted[8:]
    for wrap in (0xFFFFFFF8, 0xFFFFFFF9, 0xFFFFFFFF):
        c = bytearray(crafted)
        c[16:20] = struct.pack("<I", wrap)
        rc, _ = wav_parse(shim, bytes(c))
        assert rc == -107
        with pytest.raises(O.OracleError) as e:
            O.adx_encode(bytes(c))
        assert e.value.code == -107
    # a well-formed unknown chunk in the same place is skipped
    c = bytearray(crafted)
    c[16:20] = struct.pack("<I", 8)
    rc, f = wav_parse(shim, bytes(c))
    assert rc == 0 and f[0] == 2 and f[6] == 64 * 4
    assert O.adx_encode(bytes(c)) == O.adx_encode(bytes(w))


FIELDS = ["version", "header_size", "channels", "rate", "frame_count", "delay", "padding", "frame_size", "min_res", "max_res",
          "track_count", "channel_config", "stereo_type", "total_bands", "base_bands", "stereo_bands", "bands_per_hfr_group",
          "ms_stereo", "ath_type", "loop_start_frame", "loop_end_frame", "loop_start_delay", "loop_end_padding", "loop_flag",
          "ciph_type", "comment_len", "hfr_group_count"]


def parse_header(L, b, size_arg=None):
    out = (C.c_uint32 * 32)()
    hs = int.from_bytes(b[6:8], "big") if size_arg is None else size_arg
    rc = L.shim_hca_parse_header(b, len(b), hs, out)
    return rc, dict(zip(FIELDS, list(out)))


def test_v1_header_forms_parse(shim):
    """dec / vbr / ath / rva / comm chunk forms (hca.cpp:710-830) through the product's header walk."""
    base = O.hca_encode(synth.wav(3, 3000, 2, 48000), 1)
    h = hca_forge.forge_header(base, version=0x0101, dec=dict(stereo_type=0), ath=None)
    rc, f = parse_header(shim, h)
    assert rc == 0 and f["ath_type"] == 1 and f["bands_per_hfr_group"] == 0 and f["stereo_bands"] == 0 and f["total_bands"] == f["base_bands"]
    h = hca_forge.forge_header(base, version=0x0102, dec=dict(stereo_type=1, base=60), ath=0, rva=1.5, comm=b"hello world")
    rc, f = parse_header(shim, h)
    assert rc == 0 and f["ath_type"] == 0 and f["stereo_type"] == 1 and f["base_bands"] == 60 and f["stereo_bands"] == f["total_bands"] - 60
    assert f["comment_len"] == 11
    h = hca_forge.forge_header(base, version=0x0200, vbr=(0x100, 3))
    assert parse_header(shim, h)[0] == -201                     # vbr needs frame_size == 0 (hca.cpp:738-739), and frame_size 0 is rejected later
    h = hca_forge.forge_header(base, version=0x0200, ath=2)
    assert parse_header(shim, h)[0] == -201                     # unknown ATH type (hca.cpp:451-485)


def test_crypt_header_stays_inside_the_header(shim):
    """a v1.1 header HCA+fmt+dec+ath+crc (44 bytes).  The `ath` chunk does not shrink `size` (hca.cpp:3203-3206), so
    the later chunk tests still pass their size check at the very end of the header; every magic compare and every write
    must nevertheless stay inside it."""
    base = O.hca_encode(synth.wav(3, 3000, 2, 48000), 1)
    h = hca_forge.forge_header(base, version=0x0101, dec=dict(stereo_type=0), ath=1, ciph=None, pad=False)
    hs = int.from_bytes(h[6:8], "big")
    assert hs == 44
    for tail in (b"ciph\x00\x38\x00\x00", b"\xe3\xe9\xf0\xe8\x00\x38", b"comm\x05abcde", b"rva\x00\x3f\x80\x00\x00", b"pad\x00"):
        guard = tail + bytes(64)
        buf = C.create_string_buffer(h[:hs] + guard, hs + len(guard))
        shim.shim_hca_crypt_header(buf, hs, 1, 56)
        assert buf.raw[hs:] == guard, tail                      # nothing outside the header was touched
        got = buf.raw[:hs]
        assert got[:3] == bytes(x ^ 0x80 for x in h[:3]) and O.crc16(got) == 0
        # and it agrees with the oracle's walk on the same bytes (one-frame stream so that the oracle's crypt accepts it)
    one = hca_forge.one_frame_stream(h)
    enc = O.hca_crypt(one, 1, 56, 0x1234)
    buf = C.create_string_buffer(one[:hs], hs)
    shim.shim_hca_crypt_header(buf, hs, 1, 56)
    assert buf.raw[:hs] == enc[:hs]


def test_header_walk_mutation_fuzz_vs_oracle(shim):
    """Random edits of valid headers of every chunk form: the product's walk and the oracle's accept / reject alike, and
    (through HcaCrypt on a one-frame stream) produce the same rewritten header."""
    rng = np.random.default_rng(5)
    base = O.hca_encode(synth.wav(3, 3000, 2, 48000), 1)
    forms = [hca_forge.forge_header(base, version=0x0200),
             hca_forge.forge_header(base, version=0x0101, dec=dict(stereo_type=0), ath=None),
             hca_forge.forge_header(base, version=0x0102, dec=dict(stereo_type=1, base=60), ath=1, rva=0.5, comm=b"c" * 7),
             hca_forge.forge_header(base, version=0x0103, dec=dict(stereo_type=0), ath=0, loop=(0, 1, 0, 0), comm=b""),
             hca_forge.forge_header(base, version=0x0300, ath=0, rva=2.0, ciph=56)]
    agree = accepted = 0
    for it in range(3000):
        h = bytearray(hca_forge.one_frame_stream(forms[it % len(forms)]))
        hs = int.from_bytes(h[6:8], "big")
        for _ in range(int(rng.integers(1, 4))):
            k = int(rng.integers(4, hs - 2))
            h[k] = int(rng.integers(0, 256)) if rng.random() < 0.5 else h[k] ^ (1 << int(rng.integers(0, 8)))
        if rng.random() < 0.8:
            hs2 = int.from_bytes(h[6:8], "big")
            if 8 <= hs2 <= len(h):
                h[hs2 - 2:hs2] = struct.pack(">H", hca_forge.crc16(bytes(h[:hs2 - 2])))
        b = bytes(h)
        rc, f = parse_header(shim, b)
        try:
            enc = O.hca_crypt(b, 1, 1, 0)
            ok = True
        except O.OracleError:
            ok = False
        frames_fit = rc == 0 and f["header_size"] + f["frame_count"] * f["frame_size"] <= len(b)
        assert ok == frames_fit, (it, rc, f)
        agree += 1
        if ok:
            accepted += 1
            hsx = f["header_size"]
            buf = C.create_string_buffer(b[:hsx], hsx)
            shim.shim_hca_crypt_header(buf, hsx, 1, 1)
            assert buf.raw[:hsx] == enc[:hsx], it
    assert accepted > 300


def test_encoder_threshold_tables_equal_the_quantiser_rule(shim):
    """k_hca_encode's rate loop never quantises (hca.cpp:2763-2790): a spectrum is classed once -- how many of the fifteen
    resolutions' length thresholds of its sign it reaches, from a table row per half-binade and sign -- and costs
    shortest[r] + (class + (16 - rank[r]) >= 16) at resolution r, minus the clamp-value anomaly.  The tables come from
    hca_enc_build_tables; here the rule is held against the reference's own expression -- (int)(x * inv + (inv + 1)) -
    (int)(inv + 0.5 - 8) into QuantizeSpectrumBits, |x| >= dead zone for resolutions 8..15 -- on every float within 4096 ulps of
    a threshold, the clamp value and its neighbours, the smallest magnitudes, and two million random values per resolution."""
    ET_CP, ET_CLS, ET_INV, ET_CLEN, ET_CODE, ET_WIN4, ET_BYTES = 1568, 3872, 2048, 2336, 2592, 2848, 10016          # cri_types.h, HCA_ET_*
    CLAMP = 0x3F7FFFFE
    buf = (C.c_uint8 * 16384)()
    shim.shim_hca_enc_tables.argtypes = [C.c_void_p, C.c_size_t]
    assert shim.shim_hca_enc_tables(buf, 16384) == ET_BYTES
    blob = bytes(buf)[:ET_BYTES]
    cp = np.frombuffer(blob, dtype=np.uint32, count=120, offset=ET_CP).reshape(60, 2)
    rows = np.frombuffer(blob, dtype=np.uint32, count=768 * 2, offset=ET_CLS).reshape(768, 2)
    cls = np.stack([rows[0:256], rows[512:768]], axis=1)           # [half-binade = top bits of |x|][sign]; rows 256 .. 511 are never read
    assert all((cls[h, s] == (0x7F800000, 0)).all() for h in range(2 * 114 + 2) for s in range(2))      # below 2^-12: under every threshold
    assert (cls[254:, :, 1] == 15).all()
    inv_tab = np.frombuffer(blob, dtype=np.float32, count=16, offset=ET_INV)
    clen_all = np.frombuffer(blob, dtype=np.uint8, count=256, offset=ET_CLEN).reshape(16, 16)
    code_all = np.frombuffer(blob, dtype=np.uint8, count=256, offset=ET_CODE).reshape(16, 16)
    clen = clen_all[:8]
    for r in range(8, 16):                                         # sign-magnitude resolutions: the length of a zero, no code
        assert (clen_all[r] == r - 4).all() and (code_all[r] == 0).all()
    # the window rows: point j = l8 + 8 r takes folded inputs k = 2 j and 127 - 2 j (hca.cpp:2532-2547), window times 2^-15
    win4 = np.frombuffer(blob, dtype=np.float32, count=256, offset=ET_WIN4).reshape(8, 8, 4)
    import os
    import re
    text = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "cri_tables.h")).read()       # (generated from the reference: tools/gen_tables.py)
    w = [float.fromhex(t) for t in re.findall(r"[-0-9a-fx.]+p[-+]?\d+(?=f)", text[text.index("HCA_WINDOW[128]"):].split("};")[0])]
    assert len(w) == 128
    for r in range(8):
        for l8 in range(8):
            for odd in range(2):
                k = 127 - 2 * l8 - 16 * r if odd else 2 * l8 + 16 * r
                low = k < 64
                wa = np.float32(w[63 - k if low else k - 64]) * np.float32(1.0 / 32768.0)
                wb = np.float32(w[64 + k if low else 191 - k]) * np.float32(1.0 / 32768.0)
                assert win4[r, l8, odd] == (-wa if low else wa) and win4[r, l8, 2 + odd] == wb
    curve = [15, 14, 14, 14, 14, 14, 14, 13, 13, 13, 13, 13, 13, 12, 12, 12, 12, 12, 12, 11, 11, 11, 11, 11, 11, 10, 10, 10, 10, 10, 10, 10,
             9, 9, 9, 9, 9, 9, 8, 8, 8, 8, 8, 8, 7, 6, 6, 5, 4, 4, 4, 3, 3, 3, 2, 2, 2, 2, 1]
    dead = {8: 0x3D042108, 9: 0x3C820821, 10: 0x3C010204, 11: 0x3B808081, 12: 0x3B004020, 13: 0x3A802008, 14: 0x3A001002, 15: 0x39800801}
    thresholds = sorted(set(int(t) for t in cls[:, :, 0].reshape(-1) if t != 0x7F800000))
    assert len(thresholds) == 22                                   # fifteen per sign; the eight dead zones serve both signs

    def classes(x):                                                # what the kernel does per spectrum
        u = x.view(np.uint32)
        h = ((u >> 22) & 0x1FF).astype(np.int64)
        assert h.max() <= 253
        row = cls[h, (u >> 31).astype(np.int64)]
        ax = np.abs(x)
        return row[:, 1].astype(np.int64) + (ax >= row[:, 0].copy().view(np.float32))

    rng = np.random.default_rng(5)
    seen = set()
    for pos in range(59):
        r = curve[pos]
        assert (cp[pos, 1] >> 20) & 15 == r
        if r in seen:
            continue
        seen.add(r)
        k = int(cp[pos, 0] & 0xFF)
        assert cp[pos, 0] == k * 0x01010101 and 1 <= k <= 15
        shortest, anomaly = int(cp[pos, 1] & 0xFF) // 8, int(cp[pos, 1] >> 28) & 1
        mags = [np.arange(max(0, t - 4096), min(CLAMP, t + 4096) + 1, dtype=np.uint32) for t in thresholds]
        mags += [np.arange(CLAMP - 64, CLAMP + 1, dtype=np.uint32), np.arange(0, 64, dtype=np.uint32),
                 rng.integers(0, CLAMP + 1, 2_000_000, dtype=np.uint32)]
        mag = np.concatenate(mags)
        x = np.concatenate([mag, mag | np.uint32(0x80000000)]).view(np.float32)
        if r >= 8:
            want = (r - 4) + (np.abs(x) >= np.uint32(dead[r]).view(np.float32)).astype(np.int64)
        else:
            inv = np.float32(inv_tab[r]); up = np.float32(inv + np.float32(1))
            down = int(float(inv) + 0.5 - 8)
            t = (x * inv).astype(np.float32) + up
            want = clen[r][t.astype(np.int32) - down].astype(np.int64)
        got = shortest + (((classes(x) + k) & 0x10) != 0).astype(np.int64)
        if anomaly:
            got = np.where(x.view(np.uint32) == CLAMP, 0, got)
        assert np.array_equal(got, want), f"resolution {r}"
    assert seen == set(range(1, 16))
    assert cp[59, 0] == 0 and cp[59, 1] == 0
