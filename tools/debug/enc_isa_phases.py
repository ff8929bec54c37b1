"""Developer tool (no GPU): static instruction counts of k_hca_encode<CT> between its phase marks.

    python tools/debug/enc_isa_phases.py [CT]

Compiles csrc/cri_hca_enc.hip to ISA with -DCRI_ENC_ASM_MARKS (every ENC_MARK(k) becomes a comment line) and counts the VALU / SALU /
LDS / VMEM instructions between consecutive marks in program order.  Loop bodies are counted ONCE (the search step runs eight times,
see the per-label listing with -v); spills show as scratch_ instructions."""
import collections, os, re, subprocess, sys
ct = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 2
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from pycricodecs_amd import build as B
out = "/tmp/enc_marks.s"
subprocess.run([B._hipcc()] + B.FLAGS + ["-DCRI_ENC_ASM_MARKS"] + B._extra() + ["-x", "hip", "--cuda-device-only", "-S", os.path.join(B.CSRC, "cri_hca_enc.hip"), "-o", out],
               check=True, stderr=subprocess.DEVNULL)
name = "_ZN3cri12k_hca_encodeILi%dEEEvNS_10HcaEncArgsE" % ct
lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i])
phase, counts, order = "start", collections.OrderedDict(), []
def kind(op):
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_")): return "vmem"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "wait"
    if op.startswith("s_"): return "salu"
    return None
label = None
per_label = collections.OrderedDict()
for l in lines[start:end]:
    t = l.strip()
    m = re.match(r"; ENC_PHASE_END (\d+)", t)
    if m:
        phase = "after mark %s" % m.group(1)
        continue
    m = re.match(r"(\.LBB\d+_\d+):", t)
    if m:
        label = m.group(1)
        continue
    if not t or t.startswith((";", ".", "//")):
        continue
    k = kind(t.split()[0])
    if not k:
        continue
    counts.setdefault(phase, collections.Counter())[k] += 1
    per_label.setdefault((phase, label), collections.Counter())[k] += 1
tot = collections.Counter()
for ph, c in counts.items():
    print("%-16s %s" % (ph, "  ".join("%s %4d" % (k, c[k]) for k in ("valu", "salu", "lds", "vmem", "smem", "scratch", "barrier", "wait"))))
    tot.update(c)
print("%-16s %s" % ("total", "  ".join("%s %4d" % (k, tot[k]) for k in ("valu", "salu", "lds", "vmem", "smem", "scratch", "barrier", "wait"))))
for l in lines[start:end + 40]:
    if any(k in l for k in ("vgpr_count", "vgpr_spill", "sgpr_spill", "Occupancy", "ScratchSize", "NumVgprs", "; LDSByteSize")):
        print(l.strip())
if "-v" in sys.argv:
    for (ph, lb), c in per_label.items():
        print("  %-14s %-12s valu %4d salu %4d lds %3d" % (ph, lb, c["valu"], c["salu"], c["lds"]))
