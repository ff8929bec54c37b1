"""Developer tool (GPU box): the header-mutation fuzz of tests/test_gpu_parity.py over more base streams (channel counts,
qualities, longer than one run of 8 frames).  usage: python tools/debug/header_fuzz_multi.py [iterations]"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import oracle_lib as O, hca_forge
from pycricodecs_amd import synth, CriCodecs as cc
N = int(sys.argv[1]) if len(sys.argv) > 1 else 800
bad = tot = ok = 0
for ch in (1, 2, 4, 6, 8):
    for q in (1, 3):
        rng = np.random.default_rng(100 * ch + q)
        base = O.hca_encode(synth.wav(70 + ch, 9500, ch, 48000), quality=q)
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
