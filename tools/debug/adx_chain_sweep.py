"""ON THE GPU BOX: ADX encode / decode throughput against the number of chains, for both kernel mappings (lane per chain,
wave per file), bs 18 / bd 4 stereo files of `--seconds` each.  Prints one JSON line per (files, mapping)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=1.0)
    ap.add_argument("--files", type=int, nargs="+", default=[1000, 4000, 8192, 16384, 32768, 100000])
    args = ap.parse_args()
    import torch
    import bench
    import oracle_lib as O
    from pycricodecs_amd.batch import Job
    uniq = [bench.family_wav(900 + u, args.seconds, "tonal") for u in range(16)]
    adx_u = [O.adx_encode(w) for w in uniq]
    for n in args.files:
        for mapping in ("chain", "file"):
            if mapping == "file" and n > 8192 * 4:
                continue
            os.environ["CRICODECS_ADX_MAPPING"] = mapping
            res = {"files": n, "chains": 2 * n, "mapping": mapping, "seconds_per_file": args.seconds}
            for what in ("encode", "decode"):
                items = bench.tile(uniq if what == "encode" else adx_u, n)
                job = Job.adx_encode(items) if what == "encode" else Job.adx_decode(items)
                bufs = job.alloc("cuda:0")
                job.enable_events(True)
                for _ in range(2):
                    job.run(*bufs)
                torch.cuda.synchronize()
                ms = 0.0
                steps = 3
                for _ in range(steps):
                    job.run(*bufs)
                    ms += sum(job.event_ms().values())
                ms /= steps
                refs = adx_u if what == "encode" else [O.adx_decode(a) for a in adx_u]
                bench.verify_items(bufs[1], job.output_offsets, [i % len(uniq) for i in range(n)], refs, "adx " + what)
                res[what] = {"ms": round(ms, 3), "blocks_per_s": round(job.units2 / ms * 1e3, 1), "GBps": round(job.algorithmic_bytes / ms / 1e6, 2),
                             "kernel": job.dominant_kernel}
                del bufs, job
                torch.cuda.empty_cache()
            print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
