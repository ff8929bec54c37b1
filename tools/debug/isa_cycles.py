"""Developer tool (no GPU): VALU issue cycles of a kernel from its ISA, weighted by the rates tools/probes/valu_rate*.hip measured on gfx950:
~2 cycles per wave64 instruction for the plain add / sub / logic / right-shift / fp32 add-mul-fma forms with VGPR or constant operands,
~4 for everything else (any SGPR operand, v_lshlrev, min / max / med3, converts, DPP, SDWA, packed, 24-bit and 32-bit multiplies, bit-field
and permute ops, three-operand integer forms, carries, compares, lane reads).  Static counts (a loop body counts once).
    python tools/debug/isa_cycles.py file.s kernel_symbol_substring [label_from label_to]"""
import collections, re, sys
FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_mov_b32", "v_lshrrev_b32", "v_ashrrev_i32",
        "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_add_u16", "v_sub_u16", "v_cndmask_b32"}
def cost(line):
    t = line.split(";")[0].strip()
    op = t.split()[0]
    base = re.sub(r"_e32$|_e64$", "", op)
    if base.endswith("_dpp") or base.endswith("_sdwa") or " row_" in t or "quad_perm" in t:
        return base, 4
    if base in FAST:
        ops = t[len(op):]
        if re.search(r"\bs\d+\b|\bs\[\d+:\d+\]|\bvcc\b|\bexec\b|\bm0\b", ops) and base != "v_cndmask_b32":
            return base + " (sgpr)", 4
        return base, 2
    return base, 4
def main():
    lines = open(sys.argv[1]).read().split("\n")
    sym = sys.argv[2]
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sym in l and l.rstrip().endswith((":", "E")) or (l.startswith("_Z") and sym in l and ":" in l))
    end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i] or lines[i].startswith(".Lfunc_end"))
    lo, hi = start, end
    if len(sys.argv) > 4:
        lo = next(i for i in range(start, end) if lines[i].startswith(sys.argv[3] + ":"))
        hi = next(i for i in range(lo + 1, end) if lines[i].startswith(sys.argv[4] + ":"))
    n = collections.Counter(); cyc = collections.Counter()
    for l in lines[lo:hi]:
        t = l.strip()
        if not t.startswith("v_"):
            continue
        b, c = cost(t)
        n[b] += 1; cyc[b] += c
    tot_n, tot_c = sum(n.values()), sum(cyc.values())
    print("%d VALU instructions, %d issue cycles (%.2f per instruction; all-4 model: %d)" % (tot_n, tot_c, tot_c / max(tot_n, 1), 4 * tot_n))
    for b, c in cyc.most_common(28):
        print("  %-28s %5d x  %6d cycles  %4.1f %%" % (b, n[b], c, 100.0 * c / tot_c))
if __name__ == "__main__":
    main()
