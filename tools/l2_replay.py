#!/usr/bin/env python3
"""What one XCD's L2 would let through, WITHOUT a GPU: the memory instructions the kernels issue under the wave64 emulator
(tools/traffic_census.py's census build with HOSTWAVE_TRACE_OUT: every workgroup's global loads and stores in program order, each with
the 128-byte lines it touches), interleaved the way a chip keeps them in flight and run through a model of the L2.

Model (deliberately small): one XCD = 32 CUs x `--waves-per-cu` resident waves (16 for the decode kernels: four per SIMD); workgroups are
taken in launch order as slots free up; per round every resident workgroup issues its next memory instruction (equal progress); the
L2 is 4 MiB of 128-byte lines, 16-way set associative, LRU, write-back with write-allocate-without-fetch (streaming stores -- the
parse's quantised lines -- are written through without allocating); there is no L1 in front of
it and no Infinity Cache behind it (FETCH_SIZE / WRITE_SIZE count what crosses between L2 and the fabric either way).  Read misses x 128 B
= modelled fetch traffic, the dirty 32-byte sectors of evicted lines (+ what is dirty at the end) = modelled write traffic, both PER UNIT.

It is calibrated, not validated: `profiles/r05_a_traffic.json` has the counters of round 5's kernels at the headline's size (parse read
5427 / written 2664, transform read 6565 / written 4094 B per frame); `--lib-dir` runs any census build of the library (e.g. one made from
round 5's sources) so that before and after go through the same model.

    python tools/l2_replay.py [--streams 160 --seconds 10] [--lib-dir DIR] [--json out.json]
"""
import argparse
import array
import collections
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import traffic_census as T  # noqa: E402

LINES, WAYS = 4 * 1024 * 1024 // 128, 16
SETS = LINES // WAYS


def read_trace(path):
    """{launch number: [workgroup = [(write, (lines...)), ...] in program order]} (workgroups in block order)"""
    a = array.array("Q")
    with open(path, "rb") as f:
        a.frombytes(f.read())
    launches = collections.defaultdict(dict)
    i, n = 0, len(a)
    while i < n:
        assert a[i] == 0xB10C, "trace out of step"
        launch, block, cnt = a[i + 1], a[i + 2], a[i + 3]
        i += 4
        instrs = []
        for _ in range(cnt):
            h = a[i]
            k = h & 0xFFFF
            instrs.append(((h >> 48) & 3, tuple(a[i + 1:i + 1 + k])))
            i += 1 + k
        launches[launch][block] = instrs
    return {l: [b[k] for k in sorted(b)] for l, b in launches.items()}


def replay(blocks, slots):
    """(read-miss lines, written-back 32-byte sectors, line requests) of one launch"""
    sets = [collections.OrderedDict() for _ in range(SETS)]       # line -> mask of dirty sectors
    pop = [bin(i).count("1") for i in range(16)]
    miss = wb = req = 0
    pending = collections.deque(range(len(blocks)))
    active = []                                                    # [block index, position]
    while pending and len(active) < slots:
        active.append([pending.popleft(), 0])
    while active:
        nxt = []
        for st in active:
            instrs = blocks[st[0]]
            if st[1] >= len(instrs):
                if pending:
                    nxt.append([pending.popleft(), 0])
                continue
            write, lines = instrs[st[1]]
            st[1] += 1
            nxt.append(st)
            if write == 3:                                     # a streaming store (global_store ... nt): written through, nothing allocated
                for lm in lines:
                    req += 1
                    wb += pop[lm & 15]
                    d = sets[(lm >> 4) % SETS].pop(lm >> 4, None)
                    if d:
                        wb += pop[d & ~(lm & 15)]
                continue
            for lm in lines:
                ln, m = lm >> 4, (lm & 15) if write else 0
                req += 1
                s = sets[ln % SETS]
                d = s.pop(ln, None)
                if d is None:
                    if not write:
                        miss += 1
                    if len(s) >= WAYS:
                        _, dirty = s.popitem(last=False)
                        wb += pop[dirty]
                    s[ln] = m
                else:
                    s[ln] = d | m
        active = nxt
    wb += sum(pop[d] for s in sets for d in s.values())
    return miss, wb, req


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=160)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--waves-per-cu", type=int, default=16)
    ap.add_argument("--lib-dir", default=None, help="a census build of the library (default: this tree's, tests/hostwave/lib_traffic)")
    ap.add_argument("--json", default=None)
    ap.add_argument("--data", default="tonal", choices=["tonal", "sparse", "noise", "mixed", "sfx"])
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as td:
        trace = os.path.join(td, "trace.bin")
        units, launches, _lib = T.run_census("hca_decode", args.streams, args.seconds, trace=trace, lib_dir=args.lib_dir, family=args.data)
        tr = read_trace(trace)
    names = {l["launch"]: (T.kernel_class(l["kernel"]), l["block"]) for l in launches}
    out = {"workload": "hca_decode", "data": args.data, "streams": args.streams, "seconds": args.seconds, "frames": units, "model": "one XCD: 32 CUs x %d waves, 4 MiB 16-way LRU L2, no L1, equal progress" % args.waves_per_cu, "kernels": {}}
    print("hca_decode: %d streams x %.0f s = %d frames; bytes per frame through the modelled L2" % (args.streams, args.seconds, units))
    for launch in sorted(tr):
        name, threads = names.get(launch, ("?", 64))
        if name not in ("k_hca_parse", "k_hca_transform_plain"):
            continue
        slots = 32 * max(1, args.waves_per_cu // max(1, threads // 64))
        miss, wb, req = replay(tr[launch], slots)
        r = {"workgroups": len(tr[launch]), "resident": slots, "fetched": miss * 128 / units, "written_back": wb * 32 / units, "line_requests": req * 128 / units}
        out["kernels"][name] = r
        print("  %-24s %5d workgroups (%d resident): fetched %7.1f  written back %7.1f  (line requests %8.1f)" % (name, r["workgroups"], slots, r["fetched"], r["written_back"], r["line_requests"]))
    tot = sum(k["fetched"] + k["written_back"] for k in out["kernels"].values())
    out["total"] = tot
    print("  total %.1f B per frame" % tot)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
