"""Developer tool (GPU box): encode one WAV with the library and with the oracle, and take the first differing frame apart
(noise level, evaluation boundary, per channel: delta width, scalefactors, intensity / HFR words; whether our checksum holds)."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import golden_util as G, oracle_lib as O
from pycricodecs_amd import CriCodecs, synth, hca as H

name = sys.argv[1] if len(sys.argv) > 1 else "s0_3008_2_48000.wav"
q = int(sys.argv[2]) if len(sys.argv) > 2 else 0
w = G.load(name) if name.endswith(".wav") else synth.wav(int(name), 48000, 2, 48000)
ours, ref = CriCodecs.HcaEncode(w, 0, q), O.hca_encode(w, q)
hs = int.from_bytes(ref[6:8], "big")
plain = bytes(x & 0x7F for x in ref[:hs])
i = plain.index(b"comp")
comp = ref[i:i + 16]
fs = int.from_bytes(comp[4:6], "big")
ch = ref[12]
total, base, stereo = comp[10], comp[11], comp[12]
print("len", len(ours), len(ref), "header", hs, "frame", fs, "channels", ch, "total / base / stereo", total, base, stereo)
class BR:
    def __init__(s, b): s.b, s.p = b, 0
    def get(s, n):
        v = 0
        for _ in range(n):
            v = (v << 1) | ((s.b[s.p >> 3] >> (7 - (s.p & 7))) & 1); s.p += 1
        return v
def parse(fr, coded):
    r = BR(fr); out = {"sync": r.get(16), "noise": r.get(9), "eb": r.get(7), "ch": []}
    for c in range(ch):
        db = r.get(3); sf = []
        if db == 6:
            sf = [r.get(6) for _ in range(coded[c])]
        elif db:
            sf = [r.get(6)]; esc = (1 << db) - 1; mid = esc >> 1
            for _ in range(1, coded[c]):
                d = r.get(db)
                sf.append(r.get(6) if d == esc else sf[-1] + d - mid)
        out["ch"].append({"db": db, "sf": sf, "at": r.p})
    out["spectra_at"] = r.p
    return out
nf = (len(ref) - hs) // fs
for f in range(nf):
    a, b = ours[hs + f * fs: hs + (f + 1) * fs], ref[hs + f * fs: hs + (f + 1) * fs]
    if a == b:
        continue
    d = [k for k in range(fs) if a[k] != b[k]]
    print("frame", f, "differs at", len(d), "bytes, first", d[:12], "crc of ours", O.crc16(a), "crc of ref", O.crc16(b))
    coded = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [base + stereo] * ch
    try:
        pa, pb = parse(a, coded), parse(b, coded)
        for k in ("sync", "noise", "eb", "spectra_at"):
            print(" ", k, pa[k], pb[k])
        for c in range(ch):
            print("  ch", c, "db", pa["ch"][c]["db"], pb["ch"][c]["db"], "at", pa["ch"][c]["at"], pb["ch"][c]["at"])
            if pa["ch"][c]["sf"] != pb["ch"][c]["sf"]:
                print("   ours", pa["ch"][c]["sf"]); print("   ref ", pb["ch"][c]["sf"])
    except Exception as e:
        print("  parse failed", e)
    print("  ours", a[:24].hex(), "...", a[-4:].hex()); print("  ref ", b[:24].hex(), "...", b[-4:].hex())
    break
else:
    print("identical")
