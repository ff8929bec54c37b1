#!/usr/bin/env python3
"""Adds the f2 header-form fixtures to tests/golden/ from the REAL reference (oracle/_ref/criref): streams with the v1.x
`dec` chunk and the optional `vbr` / `ath` / `rva` / `comm` chunks (hca.cpp:710-830), the reference's decode of each (PCM16
and pre-clamp float digests) and its HcaCrypt outputs (CryptHeader walks the same chunks, hca.cpp:3166-3250).
Run in the build container only:  python tests/golden/make_golden_headers.py   (rewrites manifest.json's "header_forms")."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import hca_forge  # noqa: E402
import ref_tool as R  # noqa: E402
from make_golden import KEY, sha  # noqa: E402
from pycricodecs_amd import synth  # noqa: E402


def main():
    assert R.available(), "build oracle/_ref/criref first (make -C oracle ref)"
    forms = hca_forge.header_form_streams(R.hca_encode, synth.wav)
    out = []
    for name in sorted(forms):
        h = forms[name]
        fn = "hdr_%s.hca" % name
        with open(os.path.join(HERE, fn), "wb") as f:
            f.write(h)
        ent = {"file": fn, "sha": sha(h)}
        try:
            ent["decoded_sha"] = sha(R.hca_decode(h))
            ent["float_sha"] = sha(R.hca_decode_float(h).tobytes())
        except R.RefError:
            ent["decoded_sha"] = ent["float_sha"] = None       # the reference rejects it (header or first bad frame)
        for label, ctype, key, sub in (("enc56", 56, KEY, 0), ("enc1", 1, 0, 0), ("enc56_sub", 56, 0x7654321, 0x1111)):
            try:
                enc = R.hca_crypt(h, 1, ctype, key, sub)
            except R.RefError:
                ent[label] = None
                continue
            e = {"type": ctype, "key": hex(key), "subkey": sub, "sha": sha(enc), "decrypted_sha": sha(R.hca_crypt(enc, 0, 0, key, sub))}
            try:
                e["decoded_sha"] = sha(R.hca_decode(enc, key, sub))
            except R.RefError:
                e["decoded_sha"] = None
            ent[label] = e
        out.append(ent)
    mp = os.path.join(HERE, "manifest.json")
    with open(mp) as f:
        man = json.load(f)
    man["header_forms"] = out
    with open(mp, "w") as f:
        json.dump(man, f, indent=1, sort_keys=True)
    print("wrote %d header-form fixtures" % len(out))


if __name__ == "__main__":
    main()
