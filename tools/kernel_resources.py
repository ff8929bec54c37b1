#!/usr/bin/env python3
"""What the compiler says about every kernel of the library (no GPU needed): registers, spills, scratch, waves per SIMD -- from
`hipcc -Rpass-analysis=kernel-resource-usage` over csrc/*.hip with the product's flags -- and, from the ISA, static instruction
counts per kernel (VALU / SALU / LDS / memory / lane reads).
    python tools/kernel_resources.py [--json] > profiles/r06_kernel_resources.txt
DESIGN.md quotes these figures; tests/test_design_facts.py holds the quoted ones to this tool's output."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pycricodecs_amd import build as B  # noqa: E402

FIELDS = {"VGPRs": "vgpr", "TotalSGPRs": "sgpr", "ScratchSize [bytes/lane]": "scratch_bytes", "Occupancy [waves/SIMD]": "waves_per_simd",
          "SGPRs Spill": "sgpr_spills", "VGPRs Spill": "vgpr_spills"}


def demangle(names):
    out = subprocess.run(["c++filt"] + names, capture_output=True, text=True, check=True).stdout.split("\n")
    return [re.sub(r"^void ", "", re.sub(r"\(.*$", "", x)).replace("cri::", "") for x in out[:len(names)]]


def kernel_resources(sources=None):
    """{demangled kernel name: {vgpr, sgpr, waves_per_simd, vgpr_spills, sgpr_spills, scratch_bytes, static: {...}, source}}; the
    translation units are compiled side by side."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = sources or [s for s in B.SOURCES if s.endswith(".hip")]
    res = {}
    with ThreadPoolExecutor(len(srcs)) as ex:
        for part in ex.map(_one_source, srcs):
            res.update(part)
    return res


def _one_source(src):
    res = {}
    for src in [src]:
        with tempfile.TemporaryDirectory() as td:
            asm = os.path.join(td, "k.s")
            cmd = [B._hipcc()] + B.FLAGS + ["-x", "hip", "--offload-device-only", "-S", os.path.join(B.CSRC, src), "-o", asm, "-Rpass-analysis=kernel-resource-usage"]
            err = subprocess.run(cmd, capture_output=True, text=True, check=True).stderr
            cur, raw = None, {}
            for line in err.split("\n"):
                m = re.search(r"remark: Function Name: (\S+)", line)
                if m:
                    cur = m.group(1)
                    raw[cur] = {}
                    continue
                m = re.search(r"remark:\s+([^:]+): (\S+) \[-Rpass", line)
                if m and cur and m.group(1).strip() in FIELDS:
                    raw[cur][FIELDS[m.group(1).strip()]] = int(m.group(2))
            text = open(asm).read()
            for mangled in raw:
                i = text.find("\n%s:" % mangled)
                j = text.find("s_endpgm", i)
                body = text[i:j] if i >= 0 and j >= 0 else ""
                ins = [ln.strip().split()[0] for ln in body.split("\n") if ln.startswith("\t") and ln.strip() and not ln.strip().startswith((".", ";"))]
                raw[mangled]["static"] = {"valu": sum(x.startswith("v_") for x in ins), "salu": sum(x.startswith("s_") for x in ins), "lds": sum(x.startswith("ds_") for x in ins),
                                          "vmem": sum(x.startswith(("global_", "buffer_", "flat_")) for x in ins), "scratch": sum(x.startswith("scratch_") for x in ins),
                                          "lane_reads_writes": sum(x.startswith(("v_readlane", "v_writelane", "v_readfirstlane")) for x in ins)}
            names = list(raw)
            for n, d in zip(names, demangle(names)):
                res[d] = dict(raw[n], source=src)
    return res


def main():
    res = kernel_resources()
    if "--json" in sys.argv:
        print(json.dumps(res, indent=1, sort_keys=True))
        return
    print("# hipcc -Rpass-analysis=kernel-resource-usage, %s; flags: %s" % (subprocess.run([B._hipcc(), "--version"], capture_output=True, text=True).stdout.split("\n")[0], " ".join(B.FLAGS)))
    print("# source id %s" % B.source_id())
    print("%-78s %5s %5s %6s %6s %7s %8s | %6s %6s %5s %5s" % ("kernel", "VGPR", "SGPR", "waves", "vspill", "sspill", "scratch", "VALU", "SALU", "LDS", "VMEM"))
    for k in sorted(res):
        r = res[k]
        s = r["static"]
        print("%-78s %5d %5d %6d %6d %7d %8d | %6d %6d %5d %5d" % (k[:78], r["vgpr"], r["sgpr"], r["waves_per_simd"], r["vgpr_spills"], r["sgpr_spills"], r["scratch_bytes"],
                                                                    s["valu"], s["salu"], s["lds"], s["vmem"]))


if __name__ == "__main__":
    main()
