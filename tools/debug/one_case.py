import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import hca_forge, oracle_lib as O
from pycricodecs_amd import synth, CriCodecs as cc
ch, q, seed = 2, 4, 59
base = hca_forge.forge_v3(O.hca_encode(synth.wav(3, 4000, ch, 48000), q), 0)
hs = int.from_bytes(base[6:8], "big")
f = hca_forge.random_frames(base, seed, density=1.0 if seed % 2 else 0.35)
ref = np.frombuffer(O.hca_decode(f)[44:], dtype="<i2").reshape(-1, ch)
got = np.frombuffer(cc.HcaDecode(f, hs, 0, 0)[44:], dtype="<i2").reshape(-1, ch)
d = np.argwhere(ref != got)
print("header comp", base[0x18:0x28].hex(), "frames", int.from_bytes(base[16:20], "big"))
print("ndiff", len(d), "first", d[:6].tolist(), "last", d[-3:].tolist())
for n, c in d[:6]:
    print(n, c, "frame", (n + 128) // 1024, "sf", ((n + 128) % 1024) // 128, "ref", ref[n, c], "got", got[n, c])
fl = O.hca_decode_float(f).reshape(-1, ch)
print("float around first diff:", fl[d[0][0] + 128 - 2: d[0][0] + 128 + 3, d[0][1]])
