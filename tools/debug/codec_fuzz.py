"""Developer tool (GPU box): randomised parity of ADX decode/encode and HCA encode against the oracle.
usage: python tools/debug/codec_fuzz.py [cases]"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import oracle_lib as O
from pycricodecs_amd import synth, CriCodecs as cc
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(2024)
bad = tot = 0


def both(g, o):
    try:
        r = o()
    except O.OracleError:
        r = None
    try:
        x = g()
    except (ValueError, NotImplementedError):
        x = None
    return x, r


def rand_pcm(n, ch, kind):
    if kind == 0:
        return rng.integers(-32768, 32768, (n, ch)).astype("<i2")                    # white, full scale
    if kind == 1:
        return (rng.integers(0, 2, (n, ch)) * 65535 - 32768).astype("<i2")           # square extremes
    if kind == 2:
        x = np.zeros((n, ch), dtype="<i2"); x[rng.integers(0, n, max(1, n // 50))] = rng.integers(-32768, 32768); return x   # sparse clicks
    t = np.arange(n)[:, None]
    return (np.sin(t * rng.uniform(0.001, 3.0)) * rng.uniform(1, 32767)).astype("<i2").repeat(ch, 1)

for case in range(N):
    ch = int(rng.integers(1, 3)); n = int(rng.integers(1, 300)) * 32; sr = int(rng.choice([8000, 22050, 44100, 48000]))
    w = synth.wav_bytes(rand_pcm(n, ch, case % 4), sr)
    # ADX encode variants
    for (bd, bs, mode, ver) in [(4, 18, 3, 4), (4, 18, 4, 4), (4, 18, 2, 3), (8, 34, 3, 5), (2, 10, 3, 4), (6, 26, 4, 4), (12, 20, 3, 4), (15, 32, 3, 4)]:
        filt = int(rng.integers(0, 4)) if mode == 2 else 0
        hp = int(rng.choice([0, 500, 2000]))
        g, r = both(lambda: cc.AdxEncode(w, bd, bs, mode, hp, filt, ver, False), lambda: O.adx_encode(w, bd, bs, mode, hp, filt, ver))
        tot += 1
        if g != r:
            bad += 1; print("ADX ENC MISMATCH", case, ch, n, sr, bd, bs, mode, ver, filt, hp, None if g is None else len(g), None if r is None else len(r))
        if r is not None:
            # decode the encoded file, and a copy with random block bytes
            for variant in range(2):
                a = bytearray(r)
                if variant:
                    hs = int.from_bytes(a[2:4], "big") + 4
                    idx = rng.integers(hs, len(a), max(1, (len(a) - hs) // 3))
                    for i in idx:
                        a[int(i)] = int(rng.integers(0, 256))
                g2, r2 = both(lambda: cc.AdxDecode(bytes(a)), lambda: O.adx_decode(bytes(a)))
                tot += 1
                if g2 != r2:
                    bad += 1; print("ADX DEC MISMATCH", case, variant, ch, n, bd, bs, mode, ver, None if g2 is None else len(g2), None if r2 is None else len(r2))
    for q in (0, 1, 2, 3, 4):
        g, r = both(lambda: cc.HcaEncode(w, False, q), lambda: O.hca_encode(w, quality=q))
        tot += 1
        if g != r:
            bad += 1; print("HCA ENC MISMATCH", case, ch, n, sr, q, None if g is None else len(g), None if r is None else len(r))
print("%d comparisons, %d mismatches" % (tot, bad))
