#!/usr/bin/env python3
"""Copies the summaries tools/prof_r03.sh left under gpurun_out/<tag>/ into profiles/<tag>_* (tracked) and derives the files
bench.py reads: profiles/<tag>_traffic.json (HBM-side bytes per frame of the decode kernels) and
profiles/<tag>_pmc_1000streams.json (SQ counters per dispatch, VALU instructions per frame).   python tools/make_profiles_r03.py r03_a
A counter whose pass did not finish is null -- nothing is carried over from an earlier commit's files."""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FRAMES_1000 = 469000          # 1000 x 10 s stereo High streams / WAVs


def short(k):
    return k.replace("void cri::", "").replace("cri::", "")


def main(tag):
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    for w in ("hca_decode", "hca_encode", "adx_roundtrip", "awb_mixed", "hca_crypt", "secondaries_1000", "wide_layouts"):
        p = os.path.join(src, w + "_kernel_stats.csv")
        if os.path.exists(p):
            rows = list(csv.reader(open(p)))
            keep = [rows[0]] + [r for r in rows[1:] if "cri::" in r[0]]
            with open(os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, w)), "w", newline="") as f:
                csv.writer(f, quoting=csv.QUOTE_NONNUMERIC).writerows(keep)
    for name in ("bench.json", "bench_hca_encode.json", "bench_adx_roundtrip.json", "bench_awb_mixed.json"):
        p = os.path.join(src, name)
        if os.path.exists(p) and os.path.getsize(p):
            shutil.copy(p, os.path.join(dst, "%s_%s" % (tag, name.replace("bench.json", "bench_hca_decode.json"))))
    for name in ("hca_encode_phases.txt", "commit.txt"):
        p = os.path.join(src, name)
        if os.path.exists(p):
            shutil.copy(p, os.path.join(dst, "%s_%s" % (tag, name)))
    raw = json.load(open(os.path.join(src, "counters_raw.json")))
    commit = open(os.path.join(src, "commit.txt")).read().strip() if os.path.exists(os.path.join(src, "commit.txt")) else "?"
    # ---- HBM-side traffic of the decode kernels (tonal 1000 streams)
    out = {"_about": "HBM-side traffic, MI355X, commit %s.  tools/prof_r03.sh: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate counters-only "
                     "passes over 1000-stream batches (469 000 frames per dispatch; the full-size passes do not finish inside their timeout).  Counters are "
                     "in KB (1024 B) per dispatch, averaged over the run's dispatches; fetch correction 1.0 (see r02_*_traffic.json for the calibration "
                     "by known byte counts).  null = that counter's pass did not finish." % commit,
           "frames_per_dispatch": FRAMES_1000, "kernels": {}, "algorithmic_bytes_per_frame": 682 + 4096}
    total, complete = 0.0, True
    for k, v in raw.get("hca_decode", {}).items():
        if "k_hca_" not in k:
            continue
        name = "k_hca_parse" if "parse" in k else "k_hca_transform"
        fb = v["FETCH_SIZE"] * 1024 / FRAMES_1000 if "FETCH_SIZE" in v else None
        wb = v["WRITE_SIZE"] * 1024 / FRAMES_1000 if "WRITE_SIZE" in v else None
        ent = {"kernel_symbol": short(k), "FETCH_SIZE_KB": v.get("FETCH_SIZE"), "WRITE_SIZE_KB": v.get("WRITE_SIZE"), "fetch_correction": 1.0,
               "fetch_bytes_per_frame": None if fb is None else round(fb, 1), "write_bytes_per_frame": None if wb is None else round(wb, 1)}
        if fb is None or wb is None:
            complete = False
        else:
            ent["hbm_bytes_per_frame"] = round(fb + wb, 1)
            total += fb + wb
        out["kernels"][name] = ent
    out["total_hbm_bytes_per_frame"] = round(total, 1) if complete and total else None
    others = {}
    for w in ("hca_decode_sparse", "hca_encode", "adx_roundtrip"):
        for k, v in raw.get(w, {}).items():
            if "k_fill" in k or "scatter" in k:
                continue
            e = {c + "_KB": (round(v[c], 1) if c in v else None) for c in ("FETCH_SIZE", "WRITE_SIZE")}
            if w != "adx_roundtrip" and all(e.values()):
                e["hbm_bytes_per_frame"] = round((v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024 / FRAMES_1000, 1)
            others.setdefault(w, {})[short(k)] = e
    out["other_workloads_per_dispatch"] = others
    if out["total_hbm_bytes_per_frame"]:
        with open(os.path.join(dst, "%s_traffic.json" % tag), "w") as f:
            json.dump(out, f, indent=1)
    else:
        with open(os.path.join(dst, "%s_traffic_incomplete.json" % tag), "w") as f:
            json.dump(out, f, indent=1)
    # ---- SQ counters
    sq = {"_about": "rocprofv3 --pmc passes (counters only, no trace domains; tools/prof_r03.sh) at commit %s: average per dispatch over 1000-stream batches "
                    "(469 000 frames).  SQ_ACTIVE_INST_VALU is in quad-cycles (= SQ_INSTS_VALU: a wave64 VALU instruction holds its SIMD for 4 cycles).  "
                    "VALU_per_frame = SQ_INSTS_VALU / frames; valu_busy = 4 * SQ_INSTS_VALU / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs): the share of the "
                    "kernel's cycles its SIMDs spend issuing VALU instructions (clock-independent).  bench.py quotes both in the HCA decode line." % commit,
          "frames_per_dispatch": FRAMES_1000, "kernels": {}, "workloads": {}}
    for w in ("hca_decode", "hca_decode_sparse", "hca_encode", "adx_roundtrip"):
        for k, v in raw.get(w, {}).items():
            if "k_fill" in k or "scatter" in k:
                continue
            ent = {c: v[c] for c in sorted(v) if c.startswith("SQ_") or c.startswith("GRBM")}
            if "SQ_INSTS_VALU" in v and w != "adx_roundtrip":
                ent["VALU_per_frame"] = round(v["SQ_INSTS_VALU"] / FRAMES_1000, 1)
            if "SQ_INSTS_VALU" in v and v.get("GRBM_GUI_ACTIVE"):
                # a wave64 VALU instruction holds one of the chip's 1024 SIMDs for 4 cycles; GRBM_GUI_ACTIVE sums the 8 XCDs' busy cycles
                ent["valu_busy"] = round(v["SQ_INSTS_VALU"] * 4 / 1024 / (v["GRBM_GUI_ACTIVE"] / 8), 3)
            if w == "hca_decode":
                sq["kernels"][short(k)] = ent
            sq["workloads"].setdefault(w, {})[short(k)] = ent
    with open(os.path.join(dst, "%s_pmc_1000streams.json" % tag), "w") as f:
        json.dump(sq, f, indent=1)
    print(json.dumps(out["kernels"], indent=1), out["total_hbm_bytes_per_frame"])
    print(json.dumps({k: v.get("VALU_per_frame") for k, v in sq["kernels"].items()}))


if __name__ == "__main__":
    main(sys.argv[1])
