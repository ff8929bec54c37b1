#!/usr/bin/env python3
"""Copies the summaries tools/prof_r02.sh left under gpurun_out/<tag>/ into profiles/<tag>_* (tracked) and derives the HBM
traffic file bench.py reads (profiles/<tag>_traffic.json).   python tools/make_profiles.py r02_a"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag):
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    for w in ("hca_decode", "hca_encode", "adx_roundtrip", "awb_mixed", "hca_crypt"):
        p = os.path.join(src, w + "_kernel_stats.csv")
        if os.path.exists(p):
            rows = list(csv.reader(open(p)))
            keep = [rows[0]] + [r for r in rows[1:] if "cri::" in r[0]]
            with open(os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, w)), "w", newline="") as f:
                csv.writer(f, quoting=csv.QUOTE_NONNUMERIC).writerows(keep)
    for name in ("bench.json", "bench_hca_encode.json", "bench_adx_roundtrip.json", "bench_awb_mixed.json"):
        p = os.path.join(src, name)
        if os.path.exists(p):
            shutil.copy(p, os.path.join(dst, "%s_%s" % (tag, name.replace("bench.json", "bench_hca_decode.json"))))
    raw = json.load(open(os.path.join(src, "traffic_raw.json")))
    bench = json.load(open(os.path.join(src, "bench.json")))
    frames = bench["config"]["streams_per_gpu"] * bench["config"]["frames_per_stream"]
    out = {"_about": "HBM-side traffic of the HCA decode path (headline workload: %d frames per dispatch), MI355X.  Collected by tools/prof_r02.sh: "
                     "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, no trace domains (MI355X_MICROARCH.md, HBM section).  Counters "
                     "are in KB (1024 B) per dispatch, averaged over the run's dispatches.  Calibration by known byte counts: k_hca_transform_plain<2> has "
                     "to read the int8 lines (16 rows x 128 B), the scalefactors (256 B) and the record tails (~32 B) = 2.33 KB per frame and the counter "
                     "says 2.34-2.40 KB: since the quantised lines are laid out tile-major a lane's 8 bytes come from 64-byte pieces and FETCH_SIZE no longer "
                     "under-reports them (the gfx950 x2 rule applies to wide coalesced streaming reads: it was applied in round 1, when a row was one 128-byte "
                     "piece), so every fetch correction is 1.0 here.  k_hca_parse has to write lines 2048 + code descriptions 256 + scalefactors 256 + "
                     "intensity / tail 32 = 2592 B and WRITE_SIZE says 2.6-2.7 KB; it has to read 682 B of input and 256 B of code descriptions ONCE, but the "
                     "code descriptions are re-read for each of the 8 subframes (2 KB) and a parse wave's 16 KB of them do not survive in the 4 MB L2 of an "
                     "XCD under the record stream of 16 waves per CU, so nearly all of those re-reads reach the fabric (the counters include Infinity-Cache hits)." % frames,
           "frames_per_dispatch": frames, "kernels": {}, "algorithmic_bytes_per_frame": bench["roofline"]["algorithmic_bytes_per_launch"] // frames}
    total = 0.0
    # a counter pass of the full-size batch sometimes does not finish inside its timeout (tools/prof_r02.sh): that counter is then
    # carried over from the newest earlier traffic file -- named in the kernel's entry -- whose kernels moved the same bytes
    earlier = sorted(f for f in os.listdir(dst) if f.endswith("_traffic.json") and f[:5] < tag)
    prev = json.load(open(os.path.join(dst, earlier[-1]))) if earlier else None
    for k, v in raw["hca_decode"].items():
        if "k_hca_" not in k:
            continue
        name = "k_hca_parse" if "parse" in k else "k_hca_transform"
        carried = []
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            if c not in v:
                v[c] = prev["kernels"][name][c + "_KB"]; carried.append("%s from %s" % (c, earlier[-1]))
        fb, wb = v["FETCH_SIZE"] * 1024 / frames, v["WRITE_SIZE"] * 1024 / frames
        out["kernels"][name] = {"kernel_symbol": k.replace("void cri::", "").replace("cri::", ""), "FETCH_SIZE_KB": v["FETCH_SIZE"], "WRITE_SIZE_KB": v["WRITE_SIZE"],
                                "fetch_correction": 1.0, "fetch_bytes_per_frame": round(fb, 1), "write_bytes_per_frame": round(wb, 1),
                                "hbm_bytes_per_frame": round(fb + wb, 1), "carried_over": carried}
        total += fb + wb
    out["total_hbm_bytes_per_frame"] = round(total, 1)
    others = {}
    for w in ("hca_encode", "adx_roundtrip", "awb_mixed", "hca_crypt"):
        for k, v in raw.get(w, {}).items():
            if "k_fill" in k or "scatter" in k:
                continue
            others.setdefault(w, {})[k.replace("void cri::", "").replace("cri::", "")] = {c + "_KB": (round(v[c], 1) if c in v else None) for c in ("FETCH_SIZE", "WRITE_SIZE")}   # None: that counter pass did not finish (timeout)
    out["other_workloads_raw_counters_per_dispatch"] = others
    with open(os.path.join(dst, "%s_traffic.json" % tag), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out["kernels"], indent=1), out["total_hbm_bytes_per_frame"])


if __name__ == "__main__":
    main(sys.argv[1])
