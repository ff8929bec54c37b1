"""Developer tool (GPU box): long random-frame fuzz of the HCA decoder against the oracle (all qualities, 1/2/4/6 channels,
v2.0 and forged v3.0 headers).  usage: python tools/debug/frame_fuzz.py [seeds]"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import hca_forge, oracle_lib as O
from pycricodecs_amd import synth, CriCodecs as cc
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = acc = tot = 0
for ch in (1, 2, 4, 6):
    for q in (0, 1, 2, 3, 4):
        base2 = O.hca_encode(synth.wav(3, 4000, ch, 48000), q)
        for v3 in (False, True):
            base = hca_forge.forge_v3(base2, 0) if v3 else base2
            hs = int.from_bytes(base[6:8], "big")
            for seed in range(seeds):
                f = hca_forge.random_frames(base, seed, density=1.0 if seed % 2 else 0.35)
                tot += 1
                try:
                    ref = O.hca_decode(f)
                except O.OracleError:
                    ref = None
                try:
                    got = cc.HcaDecode(f, hs, 0, 0)
                except ValueError:
                    got = None
                if ref is not None:
                    acc += 1
                if got != ref:
                    bad += 1
                    print("MISMATCH ch %d q %d v3 %s seed %d: gpu %s oracle %s" % (ch, q, v3, seed, "reject" if got is None else len(got), "reject" if ref is None else len(ref)))
print("%d cases, %d accepted by the oracle, %d mismatches" % (tot, acc, bad))
