#!/usr/bin/env python3
"""Where a kernel's memory traffic comes from, WITHOUT a GPU: a census of every global-memory instruction of the product's kernels, taken
while they run on the wave64 emulator (tests/hostwave/, census build: the kernel files compiled with -fsanitize=thread, whose hooks the
emulator implements itself -- see hostwave.cpp "memory-traffic census").

For every wave-level memory instruction: the bytes its lanes asked for (`useful`) and the distinct 128-byte lines / 32-byte sectors they lie
in.  Per kernel and source line, per unit (frame / block row):

    useful      bytes requested
    line        128 B x distinct lines per instruction, summed     = traffic if NO line survives in a cache between two instructions
    footprint   128 B x distinct lines of the whole launch          = traffic if EVERY line is fetched / written exactly once

The measured HBM-side traffic (profiles/r05_a_traffic.json: FETCH_SIZE / WRITE_SIZE, calibrated) must lie between footprint and line; where
it sits says how much re-use the caches actually deliver.  This is a MODEL of the access pattern -- exact about what the code asks for, silent
about timing and about what the hardware's caches do with it -- and it is what a layout change can be priced with before a GPU sees it.

    python tools/traffic_census.py [hca_decode|hca_encode|adx_roundtrip] [--streams N] [--seconds S] [--json out.json]
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HW = os.path.join(ROOT, "tests", "hostwave")
SYMBOLIZER = "/opt/rocm/lib/llvm/bin/llvm-symbolizer"

WORKLOAD = r'''
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests"); sys.path.insert(0, %(root)r + "/tests/hostwave")
import mode; mode.enable()
import numpy as np, oracle_lib as O
from pycricodecs_amd import synth
from pycricodecs_amd.batch import Job
KEY = 0xCF222F1FE0748978
n, secs, wl, ch, q, v3, fam = %(streams)d, %(seconds)f, %(workload)r, %(channels)d, %(quality)d, %(v3)d, %(family)r
if fam == "tonal":
    wavs = [synth.wav(500 + i, int(48000 * secs), ch, 48000) for i in range(n)]
else:                                          # bench.py's other signal families (sparse: int16 lines; noise; mixed; sfx)
    import bench
    wavs = [bench.family_wav(500 + i, secs, fam, ch=ch) for i in range(n)]
def run(job):
    bufs = job.alloc("cpu"); job.run(*bufs)
    st = bufs[3].numpy()[:job.n]
    assert (st == 0).all(), st
    return job, bufs
if wl == "hca_decode":
    import hca_forge
    plain = [O.hca_encode(w, q) for w in wavs]
    if v3:
        plain = [hca_forge.forge_v3(h, 0) for h in plain]
    items = [O.hca_crypt(h, 1, 56, KEY, 0) for h in plain]
    job, bufs = run(Job.hca_decode(items, keys=[KEY] * n))
    outs = job.split(bytes(bufs[1].numpy()))
    assert all(bytes(o) == O.hca_decode(h, KEY, 0) for o, h in zip(outs, items))
    print("FORMS", job.transform_forms())
    print("UNITS", job.units, "frames")
elif wl == "hca_encode":
    job, bufs = run(Job.hca_encode(wavs, quality=q))
    print("UNITS", job.units, "frames")
else:
    from pycricodecs_amd import _capi
    import contextlib
    knobs = _capi.testing_knobs(adx_mapping=%(adx_mapping)r) if %(adx_mapping)r != "auto" else contextlib.nullcontext()
    with knobs:                                # (the testing build of the census library: the planner's mapping forced, as the parity tests do)
        job, bufs = run(Job.adx_encode(wavs))
        enc = [bytes(x) for x in job.split(bytes(bufs[1].numpy()))]
        assert all(e == O.adx_encode(w) for e, w in zip(enc, wavs))
        j2, b2 = run(Job.adx_decode(enc))
    print("UNITS", job.units, "block rows (each counted once: encode and decode both run)")
'''


def run_census(workload, streams, seconds, channels=2, quality=1, v3=0, adx_mapping="auto", trace=None, lib_dir=None, family="tonal"):
    if not lib_dir:
        subprocess.run([sys.executable, os.path.join(HW, "build.py"), "--traffic"], check=True, stdout=subprocess.DEVNULL)
    lib = lib_dir or os.path.join(HW, "lib_traffic")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "census.jsonl")
        env = dict(os.environ, CRI_TEST_HOSTWAVE="1", CRICODECS_LIB_DIR=lib, CRICODECS_NO_REBUILD="1", HOSTWAVE_TRAFFIC_OUT=out, HOSTWAVE_THREADS="8")
        if trace:                                              # every workgroup's memory instructions in order, for tools/l2_replay.py
            env["HOSTWAVE_TRACE_OUT"] = trace
        root = os.path.abspath(os.path.join(lib, "..", "..", "..")) if lib_dir else ROOT      # (a census build of ANOTHER tree, e.g. round 5's decoder: its own package and build id)
        r = subprocess.run([sys.executable, "-c", WORKLOAD % dict(root=root, streams=streams, seconds=seconds, workload=workload, channels=channels, quality=quality, v3=v3, adx_mapping=adx_mapping, family=family)], env=env, capture_output=True, text=True, cwd=root)
        if r.returncode:
            raise SystemExit(r.stdout[-2000:] + r.stderr[-4000:])
        units = int(re.search(r"UNITS (\d+)", r.stdout).group(1))
        with open(out) as f:
            launches = [json.loads(l) for l in f if l.strip()]
    return units, launches, os.path.join(lib, "libcricodecs_hip.so")


def symbolize(lib, offsets):
    """offset -> "file:line" of the innermost frame that lies in the product's sources (csrc), else the innermost frame"""
    offs = sorted(offsets)
    p = subprocess.run([SYMBOLIZER, "-e", lib, "-i", "-f", "-C"] + [hex(o - 1) for o in offs], capture_output=True, text=True, check=True)
    res, blocks = {}, p.stdout.strip().split("\n\n")
    for o, b in zip(offs, blocks):
        lines = b.strip().split("\n")
        frames = [(lines[i], lines[i + 1]) for i in range(0, len(lines) - 1, 2)]
        pick = next((f for f in frames if "/csrc/" in f[1]), frames[0] if frames else ("?", "?"))
        m = re.search(r"([^/]+):(\d+):\d+$", pick[1])
        res[o] = ("%s:%s" % (m.group(1), m.group(2))) if m else pick[1]
    return res


def kernel_class(name):
    name = name.strip("() ")
    return re.sub(r"<.*$", "", name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", nargs="?", default="hca_decode", choices=["hca_decode", "hca_encode", "adx_roundtrip"])
    ap.add_argument("--streams", type=int, default=96)
    ap.add_argument("--seconds", type=float, default=1.0)
    ap.add_argument("--json", default=None)
    ap.add_argument("--channels", type=int, default=2)
    ap.add_argument("--quality", type=int, default=1, help="HCA quality 0..4 (1 = High, the headline)")
    ap.add_argument("--v3", action="store_true", help="hca_decode: the streams re-headed as v3.0 with min_resolution 0 (noise fill)")
    ap.add_argument("--data", default="tonal", choices=["tonal", "sparse", "noise", "mixed", "sfx"], help="bench.py's signal families (sparse: int16 lines)")
    ap.add_argument("--adx-mapping", default="auto", choices=["auto", "chain", "file", "seg", "lane", "wave"], help="adx_roundtrip: force the planner's mapping (the lane-per-segment kernels of large batches on a small one)")
    ap.add_argument("--min-share", type=float, default=0.02, help="source lines below this share of a kernel's line bytes are folded into 'other'")
    args = ap.parse_args()
    units, launches, lib = run_census(args.workload, args.streams, args.seconds, args.channels, args.quality, int(args.v3), args.adx_mapping, family=args.data)
    sym = symbolize(lib, {s["site"] for l in launches for s in l["sites"]})
    kernels = collections.OrderedDict()
    for l in launches:
        k = kernels.setdefault(kernel_class(l["kernel"]), {"instances": set(), "launches": 0, "fp_r": 0, "fp_w": 0, "lines": collections.defaultdict(lambda: collections.Counter()), "lds_bytes": 0})
        k["instances"].add(l["kernel"].strip("() ")); k["launches"] += 1
        k["fp_r"] += l["footprint_read_bytes"]; k["fp_w"] += l["footprint_write_bytes"]; k["lds_bytes"] += l["lds_bytes"]
        for s in l["sites"]:
            c = k["lines"][(sym[s["site"]], s["rw"])]
            for f in ("instrs", "useful", "line", "s64", "s32"):
                c[f] += s[f]
    report = {"workload": args.workload, "streams": args.streams, "seconds": args.seconds, "units": units, "kernels": {}}
    print("%s: %d streams x %.2f s (%d channels, quality %d%s), %d units; bytes PER UNIT" % (args.workload, args.streams, args.seconds, args.channels, args.quality, ", v3.0 noise fill" if args.v3 else "", units))
    for name, k in kernels.items():
        tot = {rw: collections.Counter() for rw in "rw"}
        for (where, rw), c in k["lines"].items():
            tot[rw].update(c)
        if not (tot["r"]["useful"] + tot["w"]["useful"]):
            continue
        print("\n%s  (%s; %d launch%s)" % (name, ", ".join(sorted(k["instances"])), k["launches"], "" if k["launches"] == 1 else "es"))
        print("  %-6s %10s %10s %10s %10s   %s" % ("", "useful", "line", "sector32", "footprint", "line / useful"))
        kr = report["kernels"][name] = {"instances": sorted(k["instances"]), "by_source_line": []}
        for rw, label, fp in (("r", "read", k["fp_r"]), ("w", "write", k["fp_w"])):
            t = tot[rw]
            if t["useful"]:
                print("  %-6s %10.1f %10.1f %10.1f %10.1f   %.2f" % (label, t["useful"] / units, t["line"] / units, t["s32"] / units, fp / units, t["line"] / t["useful"]))
            kr[label] = {"useful": t["useful"] / units, "line": t["line"] / units, "sector32": t["s32"] / units, "sector64": t["s64"] / units, "footprint": fp / units}
        kr["lds_bytes"] = k["lds_bytes"] / units
        rows = sorted(k["lines"].items(), key=lambda kv: -kv[1]["line"])
        whole = tot["r"]["line"] + tot["w"]["line"]
        other = collections.Counter()
        for (where, rw), c in rows:
            kr["by_source_line"].append({"where": where, "rw": rw, "wave_instrs_per_unit": c["instrs"] / units, "useful": c["useful"] / units, "line": c["line"] / units, "sector32": c["s32"] / units})
            if c["line"] < args.min_share * whole:
                other.update(c)
                continue
            print("    %-28s %s %9.1f useful %9.1f line (%.2fx)  %7.2f B useful per lane-access, %.3f wave-instrs" % (where, rw, c["useful"] / units, c["line"] / units, c["line"] / max(c["useful"], 1),
                                                                                                                  c["useful"] / max(c["instrs"], 1) / 64, c["instrs"] / units))
        if other["line"]:
            print("    %-28s   %9.1f useful %9.1f line" % ("(other lines)", other["useful"] / units, other["line"] / units))
    if args.json:
        with open(args.json, "w") as f:
            json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
