import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import hca_forge, oracle_lib as O
from pycricodecs_amd import synth
from pycricodecs_amd.batch import Job
for q, ch, n in [(3, 2, 5000), (3, 1, 5000), (4, 2, 5000), (5, 2, 3000)]:
    base = hca_forge.forge_v3(O.hca_encode(synth.wav(30 + q, n, ch, 48000), q), 0)
    hdr = base[:int.from_bytes(base[6:8], "big")]
    i = hdr.find(b"comp") if b"comp" in hdr else -1
    print("q", q, "ch", ch, "comp", hdr[0x18:0x28].hex())
    items = [base] + [hca_forge.random_frames(base, s, density=1.0 if s % 2 else 0.35) for s in range(6)]
    outs, st = Job.hca_decode(items).run_host()
    for k, it in enumerate(items):
        try:
            ref = O.hca_decode(it); rs = 0
        except O.OracleError as e:
            ref = None; rs = str(e)
        same = (ref is not None and bytes(outs[k]) == ref)
        print("  item", k, "gpu status", st[k], "oracle", rs, "same", same)
        if ref is not None and st[k] == 0 and not same:
            import numpy as np
            a = np.frombuffer(bytes(outs[k])[44:], dtype="<i2"); b = np.frombuffer(ref[44:], dtype="<i2")
            d = np.nonzero(a != b)[0]
            print("    ndiff", len(d), "first", d[:10], "frame", d[0] // (1024 * ch) if len(d) else None)
