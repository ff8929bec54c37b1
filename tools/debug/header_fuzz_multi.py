"""Developer tool (GPU box): the header-mutation fuzz of tests/test_gpu_boundary.py over more base streams (channel counts,
qualities, longer than one run of 8 frames) and ADX encodings.  usage: python tools/debug/header_fuzz_multi.py [iterations]
(the ADX half is slow: edits of the sample-count field make both sides decode gigabytes of silence)"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import oracle_lib as O, hca_forge
from pycricodecs_amd import synth, CriCodecs as cc
N = int(sys.argv[1]) if len(sys.argv) > 1 else 800
bad = tot = ok = 0
for ch in (1, 2, 4, 6, 8):
    for q in (1, 3, 11, 13):                               # 1x: the same stream forged as v3.0 with noise reconstruction
        rng = np.random.default_rng(100 * ch + q)
        base = O.hca_encode(synth.wav(70 + ch, 9500, ch, 48000), quality=q % 10)
        if q >= 10:
            base = hca_forge.forge_v3(base, 0)
        hs0 = int.from_bytes(base[6:8], "big")
        for it in range(N):
            b = bytearray(base)
            for _ in range(int(rng.integers(1, 4))):
                p = int(rng.integers(0, hs0))
                b[p] = int(rng.integers(0, 256)) if rng.random() < 0.5 else b[p] ^ (1 << int(rng.integers(0, 8)))
            if it % 4:
                b[6:8] = base[6:8]
                b[hs0 - 2:hs0] = hca_forge.crc16(bytes(b[:hs0 - 2])).to_bytes(2, "big")
            data = bytes(b)
            hs = int.from_bytes(data[6:8], "big")
            try:
                ref = O.hca_decode(data)
            except O.OracleError:
                ref = None
            try:
                got = cc.HcaDecode(data, hs, 0, 0)
            except (ValueError, NotImplementedError, RuntimeError):
                got = None
            tot += 1
            if (got is None) != (ref is None) or (ref is not None and got != ref):
                bad += 1
                print("MISMATCH ch %d q %d it %d: device %s oracle %s comp %s" % (ch, q, it, "rejects" if got is None else len(got), "rejects" if ref is None else len(ref), data[0x18:0x28].hex()))
            elif ref is not None:
                ok += 1
print("%d cases, %d decoded on both sides, %d mismatches" % (tot, ok, bad))

# ---- ADX: the same over several encodings (skipped with a second argument "hca")
if len(sys.argv) > 2 and sys.argv[2] == "hca":
    sys.exit(0)
bad = tot = ok = 0
for ch in (1, 2):
    for (bd, bs, mode, ver) in ((4, 18, 3, 4), (4, 18, 2, 3), (4, 18, 4, 5), (8, 34, 3, 4), (6, 26, 3, 4)):
        rng = np.random.default_rng(1000 + 10 * ch + bd + mode)
        base = O.adx_encode(synth.wav(90 + ch, 4000, ch, 44100), bd, bs, mode, 500, 0, ver)
        hs0 = int.from_bytes(base[2:4], "big") + 4
        for it in range(N):
            b = bytearray(base)
            for _ in range(int(rng.integers(1, 4))):
                p = int(rng.integers(0, hs0))
                b[p] = int(rng.integers(0, 256)) if rng.random() < 0.5 else b[p] ^ (1 << int(rng.integers(0, 8)))
            data = bytes(b)
            try:
                ref = O.adx_decode(data)
            except O.OracleError:
                ref = None
            unsupported = False
            try:
                got = cc.AdxDecode(data)
            except (ValueError, NotImplementedError):
                got = None
            except Exception as e:                         # CriCodecsError -304: valid but not on the device path (> 64 channels)
                unsupported = getattr(e, "code", 0) == -304
                got = None
            tot += 1
            if unsupported:
                continue
            if (got is None) != (ref is None) or (ref is not None and got != ref):
                bad += 1
                print("ADX MISMATCH ch %d %s it %d: device %s oracle %s header %s" % (ch, (bd, bs, mode, ver), it, "rejects" if got is None else len(got), "rejects" if ref is None else len(ref), data[:24].hex()))
            elif ref is not None:
                ok += 1
print("ADX: %d cases, %d decoded on both sides, %d mismatches" % (tot, ok, bad))
