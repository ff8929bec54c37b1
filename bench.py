#!/usr/bin/env python3
"""bench.py -- headline benchmark of the ADX / HCA hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--streams S] [--workload hca_decode|adx_roundtrip]

Default workload = BASELINE.json configs[2], the one north_star quotes its target on: HCA v2.0 decode of 10 000
encrypted (key 0xCF222F1FE0748978) 48 kHz stereo streams of 10 s each (469 frames x 682 B per stream), inputs
resident in HBM before the timed region.  A "step" is one pass of the decode path over the whole batch.  With
N > 1 every rank decodes its own 10 000 streams (file-sharded, no data-path collective: weak scaling).

The 10 000 streams are a tiling of `--unique` (default 64) distinct seeded streams produced by the CPU oracle
(encode + encrypt); every copy occupies its own HBM, so the traffic is real.  Says so in "data".

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel, timed with HIP events on the launch stream
inside the timed steps; `cpu_baseline` is the real reference (oracle/_ref/criref, single thread) when that binary
travelled with the repo, else the C restatement ("port").
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

KEY = 0xCF222F1FE0748978
HBM_PEAK_GBPS = 8000.0


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


def make_hca_streams(unique, seconds, rank, quality=1):
    import oracle_lib as O
    from pycricodecs_amd import synth
    out = []
    for u in range(unique):
        w = synth.wav(1000 * rank + u, int(48000 * seconds) // 32 * 32, 2, 48000)
        out.append(O.hca_crypt(O.hca_encode(w, quality), 1, 56, KEY))
    return out


def cpu_baseline_hca_decode(stream, seconds=10.0):
    """Single-thread CPU decode of one of the workload's streams, repeated for ~`seconds`."""
    frames = int.from_bytes(stream[16:20], "big")
    tool = os.path.join(ROOT, "oracle", "_ref", "criref")
    if os.path.exists(tool) and os.access(tool, os.X_OK):
        with tempfile.NamedTemporaryFile(suffix=".hca", delete=False) as f:
            f.write(stream)
            path = f.name
        try:
            p = subprocess.run([tool, "bench", "hcadec", path, str(seconds), hex(KEY)], capture_output=True, text=True, timeout=seconds * 6 + 60)
            reps, secs = p.stdout.split()
            return {"value": round(int(reps) * frames / float(secs), 1), "unit": "frames/s", "cores": 1, "kind": "reference",
                    "sample": "%d x decode of one 10 s stream (%d frames) of this workload, reference C++ built from /root/reference (oracle/_ref/criref), single thread" % (int(reps), frames)}
        except Exception as e:  # fall through to the port
            log("criref bench failed:", e)
        finally:
            os.unlink(path)
    import oracle_lib as O
    t0 = time.time()
    reps = 0
    while time.time() - t0 < seconds:
        O.hca_decode(stream, KEY)
        reps += 1
    secs = time.time() - t0
    return {"value": round(reps * frames / secs, 1), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d x decode of one 10 s stream (%d frames), oracle/cri_oracle.c, single thread" % (reps, frames)}


def measure(job, dev, steps=3, warmup=1):
    import torch
    bufs = job.alloc(dev)
    job.enable_events(True)
    for _ in range(warmup):
        job.run(*bufs)
    torch.cuda.synchronize()
    kms = {}
    t0 = time.perf_counter()
    for _ in range(steps):
        job.run(*bufs)
        for k, v in job.event_ms().items():
            kms[k] = kms.get(k, 0.0) + v
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    assert int((bufs[3] != 0).sum().item()) == 0
    dom = max(kms, key=kms.get)
    return bufs, {"ms_per_step": round(dt * 1e3, 3), "units_per_s": round(job.units / dt, 1), "units2_per_s": round(job.units2 / dt, 1),
                  "kernel_ms": {k: round(v / steps, 3) for k, v in kms.items()},
                  "achieved_GBps": round(job.algorithmic_bytes / (kms[dom] / steps * 1e-3) / 1e9, 2)}


def secondary_measurements(args, dev, rank):
    """HCA encode and ADX encode/decode on `--secondary-streams` 10 s stereo WAVs (tiled from the unique set)."""
    import torch
    import oracle_lib as O
    from pycricodecs_amd import synth
    from pycricodecs_amd.batch import Job
    n = args.secondary_streams
    uniq = [synth.wav(5000 + 1000 * rank + u, int(48000 * args.seconds) // 32 * 32, 2, 48000) for u in range(min(args.unique, 16))]
    wavs = [uniq[i % len(uniq)] for i in range(n)]
    res = {}
    job = Job.hca_encode(wavs, quality=1)
    bufs, r = measure(job, dev)
    blob = bytes(bufs[1][:int(job.output_offsets[1])].cpu().numpy())
    ref = O.hca_encode(wavs[0], 1)
    assert blob[:len(ref)] == ref, "GPU HCA encode differs from the oracle"
    r.update(workload="HCA encode (quality High), %d x %.0f s 48 kHz stereo WAVs" % (n, args.seconds), unit="frames/s", frames=job.units)
    res["hca_encode"] = r
    del bufs
    torch.cuda.empty_cache()
    job = Job.adx_encode(wavs)
    bufs, r = measure(job, dev)
    adx0 = bytes(bufs[1][:int(job.output_offsets[1])].cpu().numpy())
    ref = O.adx_encode(wavs[0])
    assert adx0[:len(ref)] == ref, "GPU ADX encode differs from the oracle"
    r.update(workload="ADX encode bs18/bd4/mode3/v4, %d x %.0f s 48 kHz stereo WAVs" % (n, args.seconds), unit="frames/s (units2 = blocks/s)",
             chains=2 * n, blocks=job.units2)
    res["adx_encode"] = r
    del bufs
    torch.cuda.empty_cache()
    adx_u = [O.adx_encode(w) for w in uniq]
    job = Job.adx_decode([adx_u[i % len(adx_u)] for i in range(n)])
    bufs, r = measure(job, dev)
    wav0 = bytes(bufs[1][:int(job.output_offsets[1])].cpu().numpy())
    ref = O.adx_decode(adx_u[0])
    assert wav0[:len(ref)] == ref, "GPU ADX decode differs from the oracle"
    r.update(workload="ADX decode of the same files", unit="frames/s (units2 = blocks/s)", chains=2 * n, blocks=job.units2)
    res["adx_decode"] = r
    del bufs
    torch.cuda.empty_cache()
    # USM audio layer: the ADX files as masked @SFA chunk streams, then the demux of a container made of them (HBM-bound copies)
    from pycricodecs_amd import usm
    adx_items = [adx_u[i % len(adx_u)] for i in range(n)]
    # (a container carries at most 256 audio channels: the chunk header's channel number is one byte)
    key = 0x0123456789ABCDEF
    job = Job.sfa_pack(adx_items[:250], usm.CODEC_ADX, key, True)
    bufs, r = measure(job, dev)
    packed = bufs[1].cpu().numpy().tobytes()
    r.update(workload="ADX files -> masked @SFA chunk streams (usm.py:584-657), %d x %.0f s" % (len(adx_items[:250]), args.seconds), unit="chunks/s", chunks=job.units)
    res["sfa_pack"] = r
    del bufs
    torch.cuda.empty_cache()
    crid = b"CRID" + (0x18).to_bytes(4, "big") + bytes([0, 0x18]) + bytes(22)      # an empty CRID chunk: only the signature is read here
    job = Job.usm_audio_demux(crid + packed, key, True)
    bufs, r = measure(job, dev)
    first = bytes(bufs[1][:len(adx_items[0])].cpu().numpy())
    assert first == adx_items[0], "USM demux does not give the ADX stream back"
    r.update(workload="demux + AudioMask of that container (%d channels, %d chunks)" % (job.n, job.units), unit="chunks/s", chunks=job.units)
    res["usm_demux"] = r
    return res


def build_awb_bank(n_total, rank, world, seed=77):
    """AFS2 bank of this rank's share of `n_total` short clips (BASELINE configs[4] shape: log-uniform 0.05-2 s, 48 kHz stereo,
    half ADX bs18/bd4, half HCA High encrypted with the bank's subkey).  The global clip list is the same on every rank;
    shares are longest-processing-time balanced by frame count (pycricodecs_amd.shard).  Returns (bank, uniq, order, subkey)."""
    import struct
    import numpy as np
    import oracle_lib as O
    from pycricodecs_amd import shard, synth
    subkey, align = 0x2468, 0x20
    rng = np.random.default_rng(seed)
    durs = np.exp(rng.uniform(np.log(0.05), np.log(2.0), 24))
    uniq = []
    for u, d in enumerate(durs):
        w = synth.wav(7000 + u, max(32, int(48000 * d) // 32 * 32), 2, 48000)
        uniq.append(("hca", O.hca_crypt(O.hca_encode(w, 1), 1, 56, KEY, subkey)))
        uniq.append(("adx", O.adx_encode(w)))
    order_all = rng.integers(0, len(uniq), n_total)
    wts = [shard.hca_weight(u[1]) if u[0] == "hca" else shard.adx_weight(u[1]) // 2 for u in uniq]
    mine = shard.my_items([wts[i] for i in order_all], rank, world) if world > 1 else range(n_total)
    order = [int(order_all[i]) for i in mine]
    n = len(order)
    hs0 = 16 + 2 * n + 4 * (n + 1)
    hs = hs0 + (-hs0 % align)
    offs, pos, parts = [hs0], hs, []
    for i in order:
        cb = uniq[i][1]
        cb = cb + b"\0" * (-len(cb) % align)
        parts.append(cb); pos += len(cb); offs.append(pos)
    head = struct.pack("<4sBBHIHH", b"AFS2", 2, 4, 2, n, align, subkey) + (np.arange(n) & 0xFFFF).astype("<u2").tobytes() + np.array(offs, dtype="<u4").tobytes()
    return head.ljust(hs, b"\0") + b"".join(parts), uniq, order, subkey


def awb_mixed_measurement(args, dev, rank, world=1, gather=False):
    """Decode of a mixed AFS2 bank through the AWB front door: one HCA job + one ADX job over the bank as it sits in HBM.
    With world > 1 every rank decodes its LPT share of awb_clips x world clips and (gather=True) the decoded PCM of all
    ranks is collected on rank 0 inside the timed region -- the only collective-like step of the whole path."""
    import torch
    import torch.distributed as dist
    import oracle_lib as O
    from pycricodecs_amd import shard
    from pycricodecs_amd.batch import Job
    bank, uniq, order, subkey = build_awb_bank(args.awb_clips * world, rank, world)
    n = len(order)
    hj, aj = Job.awb_decode(bank, KEY)
    d_in, ho, hscr, hst = hj.alloc(dev)
    _, ao, ascr, ast = aj.alloc(dev, upload=False)

    def step():
        hj.run(d_in, ho, hscr, hst); aj.run(d_in, ao, ascr, ast)
        if gather and world > 1:
            shard.gather_bytes_to_root(ho[:hj.output_bytes]); shard.gather_bytes_to_root(ao[:aj.output_bytes])

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(max(args.warmup, 1)):
        step()
    barrier()
    steps = max(args.steps, 1)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    dt = (time.perf_counter() - t0) / steps
    hca_units, adx_units = float(hj.units), float(aj.units)
    if world > 1:
        t = torch.tensor([dt, 0.0, 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(t[:1], op=dist.ReduceOp.MAX)
        u = torch.tensor([hca_units, adx_units, float(n)], dtype=torch.float64, device=dev)
        dist.all_reduce(u)
        dt, hca_units, adx_units, n_all = float(t[0].item()), float(u[0].item()), float(u[1].item()), int(u[2].item())
    else:
        n_all = n
    assert int((hst < 0).sum().item()) == 0 and int((ast < 0).sum().item()) == 0
    k_h = next(i for i in range(n) if uniq[order[i]][0] == "hca"); k_a = next(i for i in range(n) if uniq[order[i]][0] == "adx")
    ref = O.hca_decode(uniq[order[k_h]][1], KEY, subkey)
    assert bytes(ho[int(hj.output_offsets[k_h]):int(hj.output_offsets[k_h]) + len(ref)].cpu().numpy()) == ref, "AWB HCA item differs from the oracle"
    ref = O.adx_decode(uniq[order[k_a]][1])
    assert bytes(ao[int(aj.output_offsets[k_a]):int(aj.output_offsets[k_a]) + len(ref)].cpu().numpy()) == ref, "AWB ADX item differs from the oracle"
    return {"workload": "AFS2 bank(s) of %d clips%s (0.05-2 s log-uniform, 48 kHz stereo, 50 %% ADX bs18/bd4 + 50 %% HCA High encrypted with subkey), decode to WAV%s"
                        % (n_all, " over %d GPUs, LPT-sharded" % world if world > 1 else "", ", PCM gathered on rank 0 (RCCL send/recv)" if gather and world > 1 else ""),
            "bank_bytes_rank0": len(bank), "pcm_bytes_rank0": int(hj.output_bytes + aj.output_bytes), "ms_per_step": round(dt * 1e3, 3),
            "hca_frames": int(hca_units), "adx_frames": int(adx_units), "frames_per_s": round((hca_units + adx_units) / dt, 1),
            "clips_per_s": round(n_all / dt, 1), "unit": "frames/s (HCA frames + ADX block rows)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=10000)
    ap.add_argument("--unique", type=int, default=64)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--workload", default="hca_decode", choices=["hca_decode", "adx_roundtrip", "awb_mixed"])
    ap.add_argument("--no-gather", action="store_true", help="awb_mixed with --gpus > 1: leave the decoded PCM on the ranks")
    ap.add_argument("--quality", type=int, default=1, help="HCA quality of the decode workload: 1 = High (the headline), 2 Middle (intensity stereo), 3 Low (HFR)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--secondary-streams", type=int, default=1000)
    ap.add_argument("--awb-clips", type=int, default=12500, help="clips of the mixed AWB secondary figure (100 000 / 8 GPUs)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback)"
    torch.cuda.set_device(local)
    dev = "cuda:%d" % local
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device(dev))

    from pycricodecs_amd.batch import Job
    from pycricodecs_amd import synth

    if args.workload == "awb_mixed":                          # BASELINE configs[4]; its own line, not the headline metric
        r = awb_mixed_measurement(args, dev, rank, world, gather=not args.no_gather)
        if rank == 0:
            print(json.dumps({"metric": "audio frames/sec, mixed AWB bank decode (BASELINE configs[4])", "value": r["frames_per_s"], "unit": "frames/s",
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_step"], "higher_is_better": True,
                              "scaling": "weak", "vs_baseline": None, "dtype": "f32+int32", "data": "synthetic (24 unique durations x 2 codecs, tiled)",
                              "config": r}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    t_setup = time.time()
    extra = {}
    if args.workload == "hca_decode":
        uniq = make_hca_streams(args.unique, args.seconds, rank, args.quality)
        items = [uniq[i % len(uniq)] for i in range(args.streams)]
        job = Job.hca_decode(items, keys=[KEY] * len(items))
        metric_cfg = {"workload": "BASELINE configs[2]: HCA v2.0 decode, %d encrypted 48 kHz stereo streams x %.0f s (quality %s, frame %d B, key 0xCF222F1FE0748978) per GPU"
                      % (args.streams, args.seconds, {0: "Highest", 1: "High", 2: "Middle", 3: "Low", 4: "Lowest"}.get(args.quality, "?"), int.from_bytes(uniq[0][0x1C:0x1E], "big")), "streams_per_gpu": args.streams, "frames_per_stream": int.from_bytes(uniq[0][16:20], "big"),
                      "unique_streams": len(uniq), "parallelism": "file-sharded x%d, no collective" % world}
        unit_bytes = "frame_size + 2*1024*channels = 682 + 4096 = 4778 B per frame"
    else:
        import oracle_lib as O
        wavs_u = [synth.wav(1000 * rank + u, int(48000 * args.seconds) // 32 * 32, 2, 48000) for u in range(args.unique)]
        wavs = [wavs_u[i % len(wavs_u)] for i in range(args.streams)]
        job = Job.adx_encode(wavs)
        metric_cfg = {"workload": "BASELINE configs[1]: ADX encode (bs18/bd4/mode3/v4), %d 48 kHz stereo WAVs x %.0f s per GPU" % (args.streams, args.seconds),
                      "chains": 2 * args.streams, "parallelism": "file-sharded x%d, no collective" % world}
        unit_bytes = "blocksize + 2*samples_per_block = 18 + 64 = 82 B per block"
    assert not job.host_status.any(), "synthetic inputs rejected at the header stage"
    d_in, d_out, d_scratch, d_status = job.alloc(dev)
    job.enable_events(True)
    torch.cuda.synchronize()
    log("setup %.1fs: %d items, %.2f GB in, %.2f GB out, %.2f GB scratch, %d units" %
        (time.time() - t_setup, job.n, job.input_bytes / 1e9, job.output_bytes / 1e9, job.scratch_bytes / 1e9, job.units))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        job.run(d_in, d_out, d_scratch, d_status)
    barrier()
    kernel_ms = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        job.run(d_in, d_out, d_scratch, d_status)
        # event read-out waits only for this step's own kernels; it is part of the measured time
        for k, v in job.event_ms().items():
            kernel_ms[k] = kernel_ms.get(k, 0.0) + v
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    bad = int((d_status != 0).sum().item())
    assert bad == 0, "%d items failed on the device" % bad

    # cheap end-to-end check outside the timed region: item 0 equals the oracle's decode
    if args.workload == "hca_decode":
        import oracle_lib as O
        n0 = int(job.output_offsets[1])
        got = bytes(d_out[:n0].cpu().numpy())
        ref = O.hca_decode(items[0], KEY)
        assert got[:len(ref)] == ref, "GPU output differs from the oracle"

    units = job.units
    ms_per_step = elapsed * 1e3 / args.steps
    value = units * world / (elapsed / args.steps)
    dom = max(kernel_ms, key=kernel_ms.get)
    dom_ms = kernel_ms[dom] / args.steps
    achieved = job.algorithmic_bytes / (dom_ms * 1e-3) / 1e9
    # HBM bytes of the dominant kernel per launch: PMC counters cannot be collected from inside this process, so the
    # per-frame figure comes from the committed rocprofv3 counter passes of this same command (profiles/, tools/prof_traffic.sh)
    traffic, traffic_src = None, None
    import glob
    tfiles = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_traffic.json")))
    tpath = tfiles[-1] if tfiles else ""
    if args.workload == "hca_decode" and args.quality == 1 and tpath:
        with open(tpath) as fh:
            tj = json.load(fh)
        if dom in tj.get("kernels", {}):
            traffic = int(round(tj["kernels"][dom]["hbm_bytes_per_frame"] * units))
            traffic_src = "profiles/%s: %.0f B/frame (FETCH_SIZE x%.0f + WRITE_SIZE, separate --pmc passes) x %d frames" % (
                os.path.basename(tpath), tj["kernels"][dom]["hbm_bytes_per_frame"], tj["kernels"][dom]["fetch_correction"], units)
    out = {
        "metric": "audio frames/sec (decode+encode) at 1/2/4/8 GPU; HBM GB/s vs roofline",
        "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.workload == "hca_decode" else "int32", "data": "synthetic (seeded sines+noise; %d unique streams tiled to %d, each copy in its own HBM)" % (args.unique, args.streams),
        "config": metric_cfg,
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": job.algorithmic_bytes, "bytes_per_unit": unit_bytes,
                     "kernel_ms_per_step": {k: round(v / args.steps, 3) for k, v in kernel_ms.items()}},
    }
    # ---- secondary figures of the same hot path (smaller batches, few steps): HCA encode, ADX encode + decode (configs[1], [3])
    if rank == 0 and world == 1 and not args.no_secondary:     # (single-GPU run only: the other ranks of a scaling run must not wait)
        del d_in, d_out, d_scratch, d_status
        torch.cuda.empty_cache()
        out["secondary"] = secondary_measurements(args, dev, rank)
        torch.cuda.empty_cache()
        out["secondary"]["awb_mixed_decode"] = awb_mixed_measurement(args, dev, rank)
    if rank == 0 and world == 1 and not args.no_cpu:
        if args.workload == "hca_decode":
            out["cpu_baseline"] = cpu_baseline_hca_decode(items[0])
        else:
            out["cpu_baseline"] = None
    if rank == 0:
        out.update(extra)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
