#!/usr/bin/env python3
"""Copies the summaries tools/prof_r05.sh left under gpurun_out/<tag>/ into profiles/<tag>_* (tracked) and derives the files bench.py
reads:  profiles/<tag>_traffic.json  (HBM-side bytes per unit of EVERY workload's kernels, calibrated per access width, summed over
the dispatches of a step)  and  profiles/<tag>_pmc_1000streams.json  (SQ counters per step, VALU instructions per frame).
    python tools/make_profiles_r05.py r04_a
A counter whose pass did not finish is null -- nothing is carried over from an earlier commit's files."""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FRAMES_1000 = 469000          # 1000 x 10 s stereo High streams / WAVs
ADX_ROWS_1000 = 15000 * 1000  # 1000 x 10 s stereo files: block rows (a row = one 18-byte block per channel)
FRAMES_FULL = 4690000         # 10 000 x 10 s stereo High streams (BASELINE configs[2])
FRAMES_ENC_FULL = 14070000    # 10 000 x 30 s stereo WAVs (BASELINE configs[3])
STEPS = {"hca_decode_full": 1, "hca_decode_sparse_full": 1, "hca_encode_full": 1, "hca_decode": 4, "hca_decode_sparse": 4, "hca_encode": 4, "adx_roundtrip": 4, "adx_roundtrip_sfx": 4}      # warm-up + timed steps of the counter commands
UNITS = {"hca_decode_full": FRAMES_FULL, "hca_decode_sparse_full": FRAMES_FULL, "hca_encode_full": FRAMES_ENC_FULL}      # (everything else: the 1000-item figures below)
# dominant global access width (bytes per lane) of each kernel's reads / writes, from the ISA (global_load_dwordx4 = 16 ...): which
# calibration stream scales its FETCH_SIZE / WRITE_SIZE.  k_hca_transform_plain reads its int8 lines 8 bytes at a time, its int16
# lines (sparse material) 16.
WIDTH = {"k_hca_parse": (16, 16), "k_hca_transform": (8, 16), "k_hca_encode": (4, 4), "k_adx_lane_encode": (16, 4), "k_adx_seg_decode": (16, 16),
         "k_adx_seg_fix": (16, 2), "k_adx_decode_wpf": (4, 4), "k_adx_seg_serial": (16, 2), "k_adx_lane_encode_serial": (16, 4)}
# algorithmic bytes per unit: what the kernel must read / write at least
ALG = {"hca_decode_full": {"unit": "frame", "units": FRAMES_FULL, "read": 682, "write": 4096},
       "hca_decode_sparse_full": {"unit": "frame", "units": FRAMES_FULL, "read": 682, "write": 4096},
       "hca_encode_full": {"unit": "frame", "units": FRAMES_ENC_FULL, "read": 4096, "write": 682},
       "hca_decode": {"unit": "frame", "units": FRAMES_1000, "read": 682, "write": 4096},
       "hca_decode_sparse": {"unit": "frame", "units": FRAMES_1000, "read": 682, "write": 4096},
       "hca_encode": {"unit": "frame", "units": FRAMES_1000, "read": 4096, "write": 682},
       "adx_roundtrip": {"unit": "block row (stereo), encode + decode", "units": ADX_ROWS_1000, "read": 128 + 36, "write": 36 + 128},
       "adx_roundtrip_sfx": {"unit": "block row (stereo), encode + decode", "units": ADX_ROWS_1000, "read": 128 + 36, "write": 36 + 128}}


def short(k):
    return k.replace("void cri::", "").replace("cri::", "")


def klass(k):
    k = short(k)
    for name in ("k_adx_lane_encode_serial", "k_adx_lane_encode", "k_adx_seg_decode", "k_adx_seg_fix", "k_adx_seg_serial", "k_adx_decode_wpf", "k_hca_parse", "k_hca_transform", "k_hca_encode"):
        if k.startswith(name):
            return name
    return k.split("<")[0]


def calibration(raw):
    """{("read" | "write", width): true bytes / counted bytes} from the known streams of tools/debug/hbm_calibrate.py (2 GiB per dispatch)."""
    cal, detail = {}, {}
    for k, v in raw.get("calibration", {}).items():
        kind = "read" if "stream_read" in k else ("write" if "stream_write" in k else None)
        if not kind:
            continue
        width = 16 if ", 4" in k else (8 if ", 2" in k else 4)
        c = "FETCH_SIZE" if kind == "read" else "WRITE_SIZE"
        if c not in v or not v[c]:
            continue
        counted = v[c] * 1024 / v["dispatches"]
        cal[(kind, width)] = (2 << 30) / counted
        detail["%s %d B/lane" % (kind, width)] = {"kernel": short(k), "known_bytes_per_dispatch": 2 << 30, "counted_bytes_per_dispatch": round(counted), "scale": round((2 << 30) / counted, 4)}
    return cal, detail


def main(tag):
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    for w in ("hca_decode", "hca_encode", "adx_roundtrip", "awb_mixed", "hca_crypt", "secondaries_1000", "wide_layouts", "enc_layouts"):
        p = os.path.join(src, w + "_kernel_stats.csv")
        if os.path.exists(p):
            rows = list(csv.reader(open(p)))
            keep = [rows[0]] + [r for r in rows[1:] if "cri::" in r[0]]
            with open(os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, w)), "w", newline="") as f:
                csv.writer(f, quoting=csv.QUOTE_NONNUMERIC).writerows(keep)
    for name in ("bench.json", "bench_hca_encode.json", "bench_adx_roundtrip.json", "bench_awb_mixed.json"):
        p = os.path.join(src, name)
        if os.path.exists(p) and os.path.getsize(p):
            shutil.copy(p, os.path.join(dst, "%s_%s" % (tag, name.replace("bench.json", "bench_default.json"))))
    for name in ("hca_encode_phases.txt", "hca_encode_phases_low.txt", "hca_encode_phases_8ch.txt", "commit.txt"):
        p = os.path.join(src, name)
        if os.path.exists(p):
            shutil.copy(p, os.path.join(dst, "%s_%s" % (tag, name)))
    raw = json.load(open(os.path.join(src, "counters_raw.json")))
    commit = open(os.path.join(src, "commit.txt")).read().strip() if os.path.exists(os.path.join(src, "commit.txt")) else "?"
    cal, cal_detail = calibration(raw)
    out = {"_about": "HBM-side traffic, MI355X, commit %s.  tools/prof_r05.sh: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate counters-only passes "
                     "(--kernel-include-regex on the library's kernels).  Workloads named *_full ran at the size the bench quotes (10 000 streams x 10 s for the decode, 10 000 x 30 s for the encode: units_per_step says); the others on 1000-item batches.  Counters are in KB (1024 B); every dispatch of a "
                     "run is SUMMED and divided by the run's steps (kernels that run several times per step -- k_adx_lane_encode's repair rounds, "
                     "k_adx_seg_fix -- count in full).  Each kernel's counts are scaled by the calibration stream of its access width ('calibration': known "
                     "2 GiB streams through the testing library's read / write kernels at 4, 8, 16 bytes per lane, same profiler, same passes).  "
                     "null = that counter's pass did not finish." % commit,
           "calibration": cal_detail, "workloads": {}}
    for w, alg in ALG.items():
        ks, tot, complete = {}, 0.0, True
        for k, v in raw.get(w, {}).items():
            if "k_fill" in k or "scatter" in k or "k_test" in k:
                continue
            name = klass(k)
            rw, ww = WIDTH.get(name, (16, 16))
            if w.startswith("hca_decode_sparse") and name == "k_hca_transform":
                rw = 16
            fr = v["FETCH_SIZE"] * 1024 / STEPS[w] if "FETCH_SIZE" in v else None
            wr = v["WRITE_SIZE"] * 1024 / STEPS[w] if "WRITE_SIZE" in v else None
            fs, ws = cal.get(("read", rw)), cal.get(("write", ww if ww >= 4 else 4))
            ent = {"kernel_symbol": short(k), "dispatches_per_step": v["dispatches"] / STEPS[w], "read_width": rw, "write_width": ww, "fetch_scale": None if fs is None else round(fs, 4),
                   "write_scale": None if ws is None else round(ws, 4), "FETCH_SIZE_KB_per_step": None if fr is None else round(fr / 1024, 1), "WRITE_SIZE_KB_per_step": None if wr is None else round(wr / 1024, 1)}
            if fr is None or wr is None or fs is None or ws is None:
                complete = False
            else:
                ent["read_bytes_per_unit"] = round(fr * fs / alg["units"], 1)
                ent["write_bytes_per_unit"] = round(wr * ws / alg["units"], 1)
                ent["hbm_bytes_per_unit"] = round((fr * fs + wr * ws) / alg["units"], 1)
                tot += fr * fs + wr * ws
            ks[name] = ent
        if ks:
            out["workloads"][w] = {"unit": alg["unit"], "units_per_step": alg["units"], "algorithmic_bytes_per_unit": alg["read"] + alg["write"],
                                   "kernels": ks, "total_hbm_bytes_per_unit": round(tot / alg["units"], 1) if complete and tot else None}
            if complete and tot:
                out["workloads"][w]["traffic_over_algorithmic"] = round(tot / alg["units"] / (alg["read"] + alg["write"]), 3)
    # (the form bench.py reads: the full-size passes under the plain names, the 1000-item ones kept as *_1000)
    for w in ("hca_decode", "hca_decode_sparse", "hca_encode"):
        if out["workloads"].get(w + "_full", {}).get("total_hbm_bytes_per_unit"):
            if w in out["workloads"]:
                out["workloads"][w + "_1000"] = out["workloads"][w]
            out["workloads"][w] = out["workloads"].pop(w + "_full")
    hd = out["workloads"].get("hca_decode", {})
    out["frames_per_dispatch"] = hd.get("units_per_step", FRAMES_1000)
    out["kernels"] = {k: dict(v, hbm_bytes_per_frame=v.get("hbm_bytes_per_unit")) for k, v in hd.get("kernels", {}).items()}
    out["total_hbm_bytes_per_frame"] = hd.get("total_hbm_bytes_per_unit")
    out["algorithmic_bytes_per_frame"] = 682 + 4096
    with open(os.path.join(dst, "%s_traffic%s.json" % (tag, "" if out["total_hbm_bytes_per_frame"] else "_incomplete")), "w") as f:
        json.dump(out, f, indent=1)
    # ---- SQ counters
    sq = {"_about": "rocprofv3 --pmc passes (counters only, no trace domains; tools/prof_r05.sh) at commit %s: per STEP (full-size batches for the HCA decode / encode -- frames_per_step says -- 1000-item ones under *_1000 and for the ADX round trip; every dispatch "
                    "of a run summed, divided by its steps).  SQ_ACTIVE_INST_VALU is in quad-cycles (= SQ_INSTS_VALU: a wave64 VALU instruction holds its "
                    "SIMD for 4 cycles).  VALU_per_frame = SQ_INSTS_VALU / frames; valu_busy = 4 * SQ_INSTS_VALU / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 "
                    "XCDs): the share of the kernel's cycles its SIMDs spend issuing VALU instructions (clock-independent).  bench.py quotes both." % commit,
          "frames_per_dispatch": FRAMES_1000, "kernels": {}, "workloads": {}}
    for w in STEPS:
        for k, v in raw.get(w, {}).items():
            if "k_fill" in k or "scatter" in k or "k_test" in k:
                continue
            ent = {c: v[c] / STEPS[w] for c in sorted(v) if c.startswith("SQ_") or c.startswith("GRBM")}
            ent["dispatches_per_step"] = v["dispatches"] / STEPS[w]
            if "SQ_INSTS_VALU" in v and not w.startswith("adx"):
                ent["VALU_per_frame"] = round(v["SQ_INSTS_VALU"] / STEPS[w] / UNITS.get(w, FRAMES_1000), 1)
                ent["frames_per_step"] = UNITS.get(w, FRAMES_1000)
            if "SQ_INSTS_VALU" in v and w.startswith("adx"):
                ent["VALU_per_block_row"] = round(v["SQ_INSTS_VALU"] / STEPS[w] / ADX_ROWS_1000, 2)
            if "SQ_INSTS_VALU" in v and v.get("GRBM_GUI_ACTIVE"):
                ent["valu_busy"] = round(v["SQ_INSTS_VALU"] * 4 / 1024 / (v["GRBM_GUI_ACTIVE"] / 8), 3)
            sq["workloads"].setdefault(w, {})[short(k)] = ent
    for w in ("hca_decode", "hca_decode_sparse", "hca_encode"):      # the full-size passes under the plain names (what bench.py quotes)
        if w + "_full" in sq["workloads"]:
            if w in sq["workloads"]:
                sq["workloads"][w + "_1000"] = sq["workloads"][w]
            sq["workloads"][w] = sq["workloads"].pop(w + "_full")
    sq["kernels"] = sq["workloads"].get("hca_decode", {})
    sq["frames_per_dispatch"] = next(iter(sq["kernels"].values()), {}).get("frames_per_step", FRAMES_1000) if sq["kernels"] else FRAMES_1000
    with open(os.path.join(dst, "%s_pmc.json" % tag), "w") as f:
        json.dump(sq, f, indent=1)
    print(json.dumps(cal_detail, indent=1))
    for w, d in out["workloads"].items():
        print(w, d["total_hbm_bytes_per_unit"], "B per", d["unit"], "vs algorithmic", d["algorithmic_bytes_per_unit"], {k: (v.get("read_bytes_per_unit"), v.get("write_bytes_per_unit")) for k, v in d["kernels"].items()})
    print(json.dumps({k: (v.get("VALU_per_frame"), v.get("valu_busy")) for w in sq["workloads"].values() for k, v in w.items()}))


if __name__ == "__main__":
    main(sys.argv[1])
