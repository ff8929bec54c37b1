#!/usr/bin/env python3
"""Round 6: copies what tools/prof_r06.sh left under gpurun_out/<tag>/ into profiles/<tag>_* exactly as round 5's tool does (kernel
stats, the calibrated traffic file, the SQ counter file: tools/make_profiles_r05.py) and adds what is new this round:
  * profiles/<tag>_bench_detail_<workload>.json: the full result behind each (now compact) bench line;
  * profiles/<tag>_secondaries.md: ONE table of the decode secondaries at 1000 streams and at the headline's size (frames/s,
    channel-frames/s, frac_end_to_end, kernel ms, transform instance) -- "is joint stereo slower than plain" without the
    1.8-fills caveat.
    python tools/make_profiles_r06.py r06_a"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def secondaries_table(detail):
    sec = detail.get("secondary", {})
    rows = ["| row | streams | M frames/s | M channel-frames/s | `frac_end_to_end` | kernel ms (parse / transform) | transform instance |", "|---|---|---|---|---|---|---|"]
    head = {"frames_per_s": detail["value"], "channel_frames_per_s": detail["value"] * 2, "frac_end_to_end": detail["roofline"]["frac_end_to_end"],
            "kernel_ms": detail["roofline"]["kernel_ms_per_step"], "transform_kernel": "k_hca_transform_plain<2>", "workload": "%d x" % detail["config"].get("streams_per_gpu", 0)}
    for name, e in [("headline (High, tonal, stereo)", head)] + sorted((k, v) for k, v in sec.items() if k.startswith("hca_decode_") and isinstance(v, dict) and "frames_per_s" in v and "channel_frames_per_s" in v):
        km = e.get("kernel_ms", {})
        rows.append("| %s | %s | %.1f | %.1f | %.4f | %s / %s | `%s` |" % (name, e["workload"].split(" x")[0].split(", ")[-1], e["frames_per_s"] / 1e6, e["channel_frames_per_s"] / 1e6, e["frac_end_to_end"],
                                                                       km.get("k_hca_parse", "-"), km.get("k_hca_transform", "-"), "%s, forms %s" % (e.get("transform_kernel", ""), e.get("transform_forms", ""))))
    return "\n".join(rows) + "\n"


def main(tag):
    import make_profiles_r05 as P5
    P5.main(tag)
    src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
    for w in ("default", "hca_encode", "adx_roundtrip", "awb_mixed"):
        p = os.path.join(src, "bench_detail_%s.json" % w)
        if os.path.exists(p):
            shutil.copy(p, os.path.join(dst, "%s_bench_detail_%s.json" % (tag, w)))
    p = os.path.join(src, "bench_detail_default.json")
    if os.path.exists(p):
        with open(p) as fh:
            detail = json.load(fh)
        with open(os.path.join(dst, "%s_secondaries.md" % tag), "w") as fh:
            fh.write("# HCA decode: every row at 1000 streams and at the headline's size (%s, `python bench.py`)\n\n" % tag + secondaries_table(detail))
        print(secondaries_table(detail))


if __name__ == "__main__":
    main(sys.argv[1])
