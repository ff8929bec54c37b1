"""Developer tool: rebuild iteration N of test_header_mutation_fuzz[hca] (same generator) and, with a second argument, compare device and oracle."""
import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import oracle_lib as O, hca_forge
from pycricodecs_amd import synth
rng = np.random.default_rng(1)
w = synth.wav(77, 3008, 2, 48000)
base = O.hca_encode(w, quality=2)
region = 96
target = int(sys.argv[1]) if len(sys.argv) > 1 else 2712
for it in range(target + 1):
    b = bytearray(base)
    if it % 8 == 7:
        b = b[:int(rng.integers(0, len(b)))]
    else:
        for _ in range(int(rng.integers(1, 4))):
            p = int(rng.integers(0, min(region, len(b))))
            b[p] = int(rng.integers(0, 256)) if rng.random() < 0.5 else b[p] ^ (1 << int(rng.integers(0, 8)))
        if it % 2 == 0:
            hs0 = int.from_bytes(base[6:8], "big")
            b[6:8] = base[6:8]
            b[hs0 - 2:hs0] = hca_forge.crc16(bytes(b[:hs0 - 2])).to_bytes(2, "big")
data = bytes(b)
print("diff bytes:", [(i, hex(base[i]), hex(data[i])) for i in range(min(len(base), len(data))) if base[i] != data[i]][:10], len(data))
print("base comp:", base[0x18:0x28].hex(), " new comp:", data[0x18:0x28].hex(), "fmt:", data[8:0x18].hex())

if len(sys.argv) > 2:
    from pycricodecs_amd import CriCodecs as cc
    hs = int.from_bytes(data[6:8], "big")
    got = cc.HcaDecode(data, hs, 0, 0); ref = O.hca_decode(data)
    g = np.frombuffer(got[44:], dtype="<i2"); r = np.frombuffer(ref[44:], dtype="<i2")
    bad = np.nonzero(g != r)[0]
    print("mismatching samples:", len(bad), "of", len(g), "first", bad[:10], "last", bad[-5:], "max abs diff", int(np.abs(g.astype(int) - r.astype(int)).max()))
    ch = 2
    print("frames touched:", sorted(set((bad // ch // 1024).tolist()))[:20])
