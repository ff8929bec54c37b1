#!/usr/bin/env python3
"""Device code of csrc/*.hip at a git revision against the working tree (no GPU needed): per translation unit, whether the ISA is
identical (only the compilation-unit id differs), else the static instruction counts of every kernel that changed.
    python tools/isa_diff.py <rev> > profiles/r06_isa_diff.txt
Round 6 used it to show that taking the EXP_* experiment bodies out of the decoder changed no instruction of it."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pycricodecs_amd import build as B  # noqa: E402


def isa(csrc, src, out):
    subprocess.run([B._hipcc()] + B.FLAGS + ["-x", "hip", "--offload-device-only", "-S", os.path.join(csrc, src), "-o", out], check=True, capture_output=True)
    text = open(out).read()
    text = re.sub(r"__hip_cuid_[0-9a-f]+", "__hip_cuid_X", text)
    return "\n".join(ln for ln in text.split("\n") if not re.match(r"\s*\.(file|ident)|^;", ln))


def kernels(text):
    out = {}
    for m in re.finditer(r"\n(_ZN3cri\w+):\s*;[^\n]*\n(.*?)s_endpgm", text, re.S):
        ins = [ln.split()[0] for ln in m.group(2).split("\n") if ln.startswith("\t") and ln.strip() and not ln.strip().startswith((".", ";"))]
        out[m.group(1)] = (len(ins), sum(x.startswith("v_") for x in ins), sum(x.startswith("s_") for x in ins), sum(x.startswith("ds_") for x in ins),
                           sum(x.startswith(("global_", "buffer_", "flat_", "scratch_")) for x in ins), hash(m.group(2)))
    return out


def main(rev):
    with tempfile.TemporaryDirectory() as td:
        subprocess.run("git -C %s archive %s pycricodecs_amd/csrc include | tar -x -C %s" % (ROOT, rev, td), shell=True, check=True)
        old_csrc = os.path.join(td, "pycricodecs_amd", "csrc")
        print("# device ISA (gfx950, product flags): %s against the working tree (source id %s)" % (subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", rev], capture_output=True, text=True).stdout.strip(), B.source_id()))
        for src in [s for s in B.SOURCES if s.endswith(".hip")]:
            a, b = isa(old_csrc, src, os.path.join(td, "a.s")), isa(B.CSRC, src, os.path.join(td, "b.s"))
            if a == b:
                print("%-18s identical (every instruction, every kernel descriptor)" % src)
                continue
            ka, kb = kernels(a), kernels(b)
            names = subprocess.run(["c++filt"] + list(kb), capture_output=True, text=True).stdout.split("\n")
            changed = [(n, k) for n, k in zip(names, kb) if ka.get(k, (0,) * 6)[5] != kb[k][5]]
            print("%-18s %d of %d kernels differ:" % (src, len(changed), len(kb)))
            for n, k in changed:
                o, w = ka.get(k, (0,) * 6), kb[k]
                print("    %-70s total %5d -> %5d   VALU %5d -> %5d   SALU %5d -> %5d   LDS %4d -> %4d   memory %4d -> %4d"
                      % (re.sub(r"^void cri::|\(.*$", "", n)[:70], o[0], w[0], o[1], w[1], o[2], w[2], o[3], w[3], o[4], w[4]))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "HEAD")
