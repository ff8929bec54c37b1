#!/usr/bin/env python3
"""bench.py -- benchmark of the ADX / HCA hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload hca_decode|hca_encode|adx_roundtrip|awb_mixed] ...

Workloads = BASELINE.json configs (a "step" is one pass of the path over the whole batch, inputs resident in HBM):
  hca_decode     configs[2] (default, the one north_star quotes its target on): HCA v2.0 decode of 10 000 encrypted
                 (key 0xCF222F1FE0748978) 48 kHz stereo streams x 10 s (469 frames x 682 B each)
  hca_encode     configs[3]: HCA encode (v2.0, quality High) of 10 000 x 30 s 48 kHz stereo WAVs (1407 frames each)
  adx_roundtrip  configs[1]: ADX encode + decode of 1 000 x 10 s 48 kHz stereo WAVs (bs 18 / bd 4), the decode job reading the
                 encode job's output buffer
  awb_mixed      configs[4]: AFS2 bank(s) of short ADX + HCA clips through the AWB front door, decoded PCM gathered on rank 0

--gpus N > 1: one process per GPU.  Without torchrun's environment this script launches itself under
`python -m torch.distributed.run --nproc-per-node N` (rendezvous on 127.0.0.1); under torchrun it reads RANK / WORLD_SIZE /
LOCAL_RANK.  Every rank works on its own batch (file-sharded, weak scaling, no data-path collective; awb_mixed adds the
RCCL gather of the PCM to rank 0 inside the timed region), timing is barrier + synchronize on both sides, MAX over ranks.

Every output of every timed batch is verified after the timed region, not item 0: each item is compared on the device, byte
for byte, with the CPU oracle's result for the unique input it is a copy of (so offsets past 4 GiB and 16 GiB are covered).

The batches are tilings of `--unique` distinct seeded inputs produced with the CPU oracle; every copy occupies its own HBM.
--data picks the signal family (family_pcm): tonal (sines + noise floor: every HCA frame qualifies for the int8 record form),
sparse (clean narrow-band material: bands of resolution 12..15, int16 records), noise (full-scale broadband material) or
mixed (tonal and sparse alternating).  "data" in the line says which, and the record-form census of the run is in
config.record_forms.

Prints ONE JSON line of at most 4 KB on stdout (rank 0; compact_line / emit): the contract's keys, `config`, `roofline`, `cpu_baseline`.
`roofline.frac` = algorithmic bytes of the whole step / wall time of a step / HBM peak (the end-to-end figure); `dominant_kernel`
inside it is the longest kernel with its OWN algorithmic bytes over its HIP-event time (events on the launch stream, inside the
timed steps).  `cpu_baseline` is the real reference (oracle/_ref/criref, single thread) when that binary travelled with the repo,
else the C restatement ("port").  Everything else of the default run -- the other BASELINE configurations at their written sizes,
the secondaries at 1000 streams and at full size, host-memory lines, single-call latencies -- goes to bench_detail.json (repo
root; also $BENCH_DETAIL_DIR) and to stderr.
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

KEY = 0xCF222F1FE0748978
HBM_PEAK_GBPS = 8000.0
QNAME = {0: "Highest", 1: "High", 2: "Middle", 3: "Low", 4: "Lowest"}


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


# ------------------------------------------------------------------------------------------------ synthetic inputs
def family_pcm(seed, n, ch, sr, family):
    """(n, ch) int16 of one of the signal families:
      tonal   pycricodecs_amd.synth: two sines + a noise floor per channel.  The encoder spends its bits on
              many bands: resolutions stay <= 11, i.e. every symbol fits 8 bits -> int8 frame records.
      sparse  sparse spectra (one loud sine, two sines, a quiet sine, low-passed noise; by seed): few bands get all the
              bits, resolutions 12..15 (up to 12-bit symbols) -> int16 frame records.  What clean tonal material does.
      noise   full-scale material (white noise, +-full-scale square noise, clicks on a noise floor, loud tones in noise):
              every band is loud, the noise level is high, resolutions are low -> int8 records, maximal symbol count.
      mixed   tonal and sparse alternating by seed (tiles hold both record forms).
      sfx     game-SFX shape: the tonal family with 0.05-0.5 s of DIGITAL silence (exact zeros, no noise floor) before and after the
              sound -- at most a third of the clip each.  Through the zeros an ADX decoder's state sits at a history-dependent fixed
              point, which is what the segmented ADX kernels' silent-run repair is for."""
    import numpy as np
    from pycricodecs_amd import synth
    if family == "mixed":
        family = "tonal" if seed % 2 == 0 else "sparse"
        seed //= 2
    if family == "tonal":
        return synth.pcm16(seed, n, ch, sr)
    if family == "sfx":
        rng = np.random.default_rng(20_000 + seed)
        head = min(int(rng.uniform(0.05, 0.5) * sr), n // 3)
        tail = min(int(rng.uniform(0.05, 0.5) * sr), n // 3)
        x = np.zeros((n, ch), dtype=np.int16)
        if n - head - tail > 0:
            x[head:n - tail] = synth.pcm16(seed, n - head - tail, ch, sr)      # (its own fade-in: the sound starts softly, as the family does)
        return x
    rng = np.random.default_rng(10_000 + seed)
    kind = seed % 4
    t = np.arange(n)[:, None] / sr
    if family == "sparse":
        f1, f2 = rng.uniform(200, 4000), rng.uniform(4000, 12000)
        ph = np.arange(ch)[None, :] * 0.7
        if kind == 0:
            x = 0.9 * 32767 * np.sin(2 * np.pi * f1 * t + ph)
        elif kind == 1:
            x = 0.45 * 32767 * (np.sin(2 * np.pi * f1 * t + ph) + np.sin(2 * np.pi * f2 * t))
        elif kind == 2:
            x = 0.05 * 32767 * np.sin(2 * np.pi * f1 * t + ph)
        else:
            X = np.fft.rfft(rng.normal(0, 1, (n, ch)), axis=0)
            X[len(X) // 20:] = 0
            x = np.fft.irfft(X, n, axis=0)
            x = x / np.abs(x).max() * 30000
    else:
        assert family == "noise", family
        if kind == 0:
            x = rng.integers(-32768, 32768, (n, ch))
        elif kind == 1:
            x = rng.integers(0, 2, (n, ch)) * 65535 - 32768
        elif kind == 2:
            x = rng.normal(0, 300, (n, ch))
            idx = rng.integers(0, n, max(1, n // 40))
            x[idx] = rng.integers(-32768, 32768, (len(idx), ch))
        else:
            x = sum(9000.0 * np.sin(2 * np.pi * f * t + c) for c, f in enumerate(rng.uniform(200, 18000, 6))) + rng.normal(0, 4000, (n, ch))
    # quadratic fade-in over 512 samples, as in synth.pcm16: an ADX file whose FIRST block needs a scale >= 0x100 is one the
    # reference's own decoder rejects (its 7-byte "(c)CRI" compare runs into the scale word, adx.cpp:345-348)
    m = min(512, n)
    x = np.asarray(x, dtype=np.float64)
    x[:m] *= ((np.arange(m) / 512.0) ** 2)[:, None]
    return np.clip(np.round(x), -32768, 32767).astype(np.int16)


def family_wav(seed, seconds, family, ch=2, sr=48000):
    from pycricodecs_amd import synth
    return synth.wav_bytes(family_pcm(seed, int(sr * seconds) // 32 * 32, ch, sr, family), sr)


def tile(uniq, n):
    return [uniq[i % len(uniq)] for i in range(n)]


# ------------------------------------------------------------------------------------------------ CPU baselines
def criref():
    tool = os.path.join(ROOT, "oracle", "_ref", "criref")
    return tool if os.path.exists(tool) and os.access(tool, os.X_OK) else None


def criref_bench(what, data, seconds, key=0):
    """(repetitions, elapsed seconds) of the real reference on one file, single thread."""
    with tempfile.NamedTemporaryFile(delete=False) as f:
        f.write(data)
        path = f.name
    try:
        p = subprocess.run([criref(), "bench", what, path, str(seconds), hex(key)], capture_output=True, text=True, timeout=seconds * 6 + 120)
        reps, secs = p.stdout.split()
        return int(reps), float(secs)
    finally:
        os.unlink(path)


def timed_loop(fn, seconds):
    t0, reps = time.time(), 0
    while time.time() - t0 < seconds or reps == 0:
        fn()
        reps += 1
    return reps, time.time() - t0


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def criref_all_cores(what, data, seconds, key, frames):
    """The same reference loop in one process per host core at once (single thread AND all cores from the same run)."""
    from concurrent.futures import ThreadPoolExecutor
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    with ThreadPoolExecutor(n) as ex:
        res = list(ex.map(lambda _: criref_bench(what, data, seconds, key), range(n)))
    return {"value": round(sum(r * frames / s for r, s in res), 1), "unit": "frames/s", "cores": n, "cpu_model": cpu_model(),
            "sample": "%d processes x %.0f s of the single-thread loop, all at once" % (n, seconds)}


def cpu_baseline(kind, sample, frames, seconds=10.0):  # noqa: C901
    """kind: hcadec (sample = encrypted stream), hcaenc (sample = WAV), adxrt (sample = WAV: encode, then decode of the result).
    `frames` = the metric's units one repetition processes.  value / cores = ONE host thread; all_cores = every core at once."""
    import oracle_lib as O
    what = {"hcadec": "HCA decode of one stream of this workload", "hcaenc": "HCA encode (High) of one WAV of this workload",
            "adxrt": "ADX encode + decode of one WAV of this workload"}[kind]
    if criref():
        try:
            if kind == "adxrt":
                r1, s1 = criref_bench("adxenc", sample, seconds / 2)
                r2, s2 = criref_bench("adxdec", O.adx_encode(sample), seconds / 2)
                value, reps = frames / (s1 / r1 + s2 / r2), r1 + r2
            else:
                reps, secs = criref_bench(kind, sample, seconds, KEY if kind == "hcadec" else 0)
                value = reps * frames / secs
            out = {"value": round(value, 1), "unit": "frames/s", "cores": 1, "kind": "reference", "cpu_model": cpu_model(),
                   "sample": "%d x %s (%d frames each), reference C++ built from /root/reference (oracle/_ref/criref), single thread" % (reps, what, frames)}
            try:
                if kind == "adxrt":
                    a = criref_all_cores("adxenc", sample, seconds / 2, 0, frames / 2)
                    b = criref_all_cores("adxdec", O.adx_encode(sample), seconds / 2, 0, frames / 2)
                    a["value"] = round(frames / (frames / 2 / a["value"] + frames / 2 / b["value"]), 1)
                    out["all_cores"] = a
                else:
                    out["all_cores"] = criref_all_cores(kind, sample, seconds, KEY if kind == "hcadec" else 0, frames)
            except Exception as e:
                log("all-cores baseline failed:", e)
            return out
        except Exception as e:  # fall through to the port
            log("criref bench failed:", e)
    if kind == "hcadec":
        reps, secs = timed_loop(lambda: O.hca_decode(sample, KEY), seconds)
    elif kind == "hcaenc":
        reps, secs = timed_loop(lambda: O.hca_encode(sample, 1), seconds)
    else:
        reps, secs = timed_loop(lambda: O.adx_decode(O.adx_encode(sample)), seconds)
    return {"value": round(reps * frames / secs, 1), "unit": "frames/s", "cores": 1, "kind": "port", "cpu_model": cpu_model(),
            "sample": "%d x %s (%d frames each), oracle/cri_oracle.c, single thread" % (reps, what, frames)}


def cpu_baseline_awb(parts, seconds=4.0):
    """Host-core baseline of the mixed AWB bank: the reference decodes one HCA clip and one ADX clip of the bank in a loop (single
    thread, then every core at once); the bank's rate follows from its own mix of HCA frames and ADX block rows."""
    import oracle_lib as O
    if not criref() or not parts.get("hca") or not parts.get("adx"):
        return None
    hca = O.hca_crypt(O.hca_crypt(parts["hca"], 0, 56, KEY, parts["subkey"]), 1, 56, KEY)     # the same frames under the plain key (the tool takes no subkey)
    adx = parts["adx"]
    hf = int.from_bytes(hca[16:20], "big")
    ar = (len(adx) - int.from_bytes(adx[2:4], "big") - 4) // (18 * adx[7]) - 1
    H, A = parts["hca_frames"], parts["adx_rows"]

    def rate(fn):
        r_h, r_a = fn("hcadec", hca, hf, KEY), fn("adxdec", adx, ar, 0)
        return (H + A) / (H / r_h + A / r_a)
    one = lambda what, data, units, key: (lambda r, s_: r * units / s_)(*criref_bench(what, data, seconds / 2, key))
    out = {"value": round(rate(one), 1), "unit": "frames/s (HCA frames + ADX block rows)", "cores": 1, "kind": "reference", "cpu_model": cpu_model(),
           "sample": "reference C++ (oracle/_ref/criref), single thread: %.0f s of HCA decode of one %d-frame clip and %.0f s of ADX decode of one %d-row clip of the bank, weighted by the bank's %d HCA frames / %d ADX rows"
                     % (seconds / 2, hf, seconds / 2, ar, H, A)}
    try:
        allc = lambda what, data, units, key: criref_all_cores(what, data, seconds / 2, key, units)["value"]
        n = len(os.sched_getaffinity(0))
        out["all_cores"] = {"value": round(rate(allc), 1), "unit": "frames/s", "cores": n, "sample": "%d processes of the same loops at once" % n}
    except Exception as e:
        log("all-cores baseline failed:", e)
    return out


# ------------------------------------------------------------------------------------------------ device helpers
def near_the_gpu(dev_index):
    """Runs this rank on the cores of the socket its GPU hangs off (and so puts what it allocates from here on into that
    socket's memory): on a two-socket host the PCIe-inclusive lines lose 10-15 % when the buffers sit across the inter-socket
    link.  Returns what it did, for the bench line; a no-op where sysfs does not say."""
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(dev_index)) != 0:
            return None
        bdf = buf.value.decode().lower()
        base = "/sys/bus/pci/devices/%s/" % bdf
        node = int(open(base + "numa_node").read())
        cpus = set()
        for part in open(base + "local_cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if node < 0 or not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return {"gpu": bdf, "numa_node": node, "cpus": len(cpus)}
    except Exception:
        return None


class Dist:
    """torch.distributed plumbing of one rank (world 1: no process group)."""
    def __init__(self):
        import torch
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback)"
        # Launcher smoke test on a box with fewer GPUs than ranks (CRICODECS_BENCH_SHARE_GPU=1): ranks share device 0 and
        # rendezvous over gloo, since RCCL refuses two ranks on one device.  Never the measured configuration.
        self.shared = os.environ.get("CRICODECS_BENCH_SHARE_GPU") == "1"
        dev_index = self.local % torch.cuda.device_count() if self.shared else self.local
        assert dev_index < torch.cuda.device_count(), "rank %d has no GPU: %d visible" % (self.local, torch.cuda.device_count())
        torch.cuda.set_device(dev_index)
        self.dev = "cuda:%d" % dev_index
        self.affinity0 = os.sched_getaffinity(0)
        self.numa = near_the_gpu(dev_index)
        if self.world > 1:
            import torch.distributed as dist
            if self.shared:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=torch.device(self.dev))

    def barrier(self):
        import torch
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def reduce(self, values, op="max"):
        """list of floats -> list reduced over ranks"""
        if self.world == 1:
            return list(values)
        import torch
        import torch.distributed as dist
        t = torch.tensor(list(values), dtype=torch.float64, device="cpu" if self.shared else self.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
        return [float(x) for x in t.tolist()]

    def close(self):
        if self.world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()


def verify_items(d_out, offsets, which, refs, what):
    """Every output item i equals refs[which[i]] byte for byte (compared on the device).  Returns a summary dict; raises on the
    first batch of mismatches."""
    import torch
    dev = d_out.device
    d_refs = [None if r is None else torch.frombuffer(bytearray(r), dtype=torch.uint8).to(dev) for r in refs]      # (None: a unique input no item is a copy of)
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    total = 0
    for i, u in enumerate(which):
        o, n = int(offsets[i]), d_refs[u].numel()
        bad += (d_out[o:o + n] != d_refs[u]).any()
        total += n
    nbad = int(bad.item())
    if nbad:
        first = next(i for i, u in enumerate(which) if not torch.equal(d_out[int(offsets[i]):int(offsets[i]) + d_refs[u].numel()], d_refs[u]))
        raise AssertionError("%s: %d of %d outputs differ from the oracle (first: item %d at offset %d)" % (what, nbad, len(which), first, int(offsets[first])))
    return {"items": len(which), "bytes": total, "max_offset": int(offsets[len(which) - 1]) if len(which) else 0,
            "how": "every item compared on the device, byte for byte, with the CPU oracle's output for its unique input"}


def run_timed(D, step, steps, warmup, jobs):
    """warmup, barrier, `steps` timed steps (HIP-event read-out included), barrier; returns (seconds per step MAX over ranks,
    {kernel class: ms per step})."""
    for _ in range(warmup):
        step()
    D.barrier()
    kms = {}
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
        for job in jobs:
            for k, v in job.event_ms().items():                # waits only for this step's own kernels; part of the measured time
                kms[k] = kms.get(k, 0.0) + v
    D.barrier()
    dt = (time.perf_counter() - t0) / steps
    dt = D.reduce([dt], "max")[0]
    return dt, {k: v / steps for k, v in kms.items()}


def sustained_run(job, bufs, seconds):
    """The same job in a loop for `seconds` (after the timed steps): what the clocks settle at under this kernel family.  The shader
    clock and the socket power are sampled once, halfway, from rocm-smi (None when it is not there).  The default steps of the bench
    are a burst of ~0.15 s; a power-capped part runs its long jobs at the rate reported here."""
    import subprocess
    import torch
    import re
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0; smi = None
    while True:
        for _ in range(8):
            job.run(*bufs)
        torch.cuda.synchronize(); n += 8
        el = time.perf_counter() - t0
        if smi is None and el > seconds / 2:
            for _ in range(8):
                job.run(*bufs)                                 # (sampled while the queue is full)
            n += 8
            try:
                txt = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
                m, w = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", txt), re.search(r"Power \(W\): ([0-9.]+)", txt)
                smi = {"sclk_mhz": int(m.group(1)) if m else None, "socket_power_w": float(w.group(1)) if w else None}
            except Exception:
                smi = {}
            torch.cuda.synchronize()
        if el >= seconds:
            break
    dt = (time.perf_counter() - t0) / n
    return dict({"seconds": round(time.perf_counter() - t0, 1), "ms_per_step": round(dt * 1e3, 3), "frames_per_s": round(job.units / dt, 1)}, **(smi or {}))


def roofline_of(alg_bytes, alg_bytes_path, kernel_ms, dt, traffic=None, extra=None):
    """`achieved` / `frac`: algorithmic bytes of the WHOLE step (SURVEY 8(d)'s per-unit figure x units) / wall time of a step -- all kernels
    of the path, launch gaps included -- against the HBM peak (round 6: until round 5 `frac` divided the whole path's bytes by ONE kernel's
    time, which credited that kernel with bytes another one moves).  `dominant_kernel`: the longest kernel of the step with the algorithmic
    bytes that are its OWN (alg_bytes: e.g. the HCA transform writes the PCM, 2 x 1024 x channels per frame; the parse reads the frame) over
    its HIP-event time.  alg_bytes_path: of the whole step."""
    dom = max(kernel_ms, key=kernel_ms.get)
    own = alg_bytes / (kernel_ms[dom] * 1e-3) / 1e9
    e2e = alg_bytes_path / dt / 1e9
    r = {"bound": "valu" if extra and "valu" in extra else "hbm", "kernel": dom, "achieved": round(e2e, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
         "frac": round(e2e / HBM_PEAK_GBPS, 5), "frac_end_to_end": round(e2e / HBM_PEAK_GBPS, 5), "traffic": None,
         "algorithmic_bytes_per_launch": int(alg_bytes_path),
         "kernel_ms_per_step": {k: round(v, 3) for k, v in kernel_ms.items()},
         "dominant_kernel": {"name": dom, "ms": round(kernel_ms[dom], 3), "own_algorithmic_bytes": int(alg_bytes), "achieved": round(own, 2), "frac": round(own / HBM_PEAK_GBPS, 5)}}
    if traffic:
        r.update(traffic)
        if r.get("traffic"):
            r["traffic_over_algorithmic"] = round(r["traffic"] / alg_bytes_path, 3)
    if extra:
        r.update(extra)
    return r


def committed_traffic(units, dom, workload="hca_decode"):
    """HBM bytes per step from the committed rocprofv3 counter passes of this same path (PMC counters cannot be collected from inside
    the process): the newest profiles/r*_traffic.json -- per-unit bytes of the workload's kernels (FETCH_SIZE + WRITE_SIZE, separate
    --pmc passes, scaled per access width by known calibration streams, every dispatch of a step summed) x this run's units.
    workload: hca_decode | hca_decode_sparse | hca_encode | adx_roundtrip | adx_roundtrip_sfx (units = block rows, each counted once)."""
    tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not tfiles:
        return None
    with open(tfiles[-1]) as fh:
        tj = json.load(fh)
    wl = tj.get("workloads", {}).get(workload)
    if wl is None and workload == "hca_decode":                 # (round-3 files: the headline's kernels only)
        wl = {"kernels": {k: dict(v, hbm_bytes_per_unit=v.get("hbm_bytes_per_frame")) for k, v in tj.get("kernels", {}).items()},
              "total_hbm_bytes_per_unit": tj.get("total_hbm_bytes_per_frame"), "unit": "frame"}
    if not wl or not wl.get("total_hbm_bytes_per_unit"):
        return None
    ks, total = wl["kernels"], wl["total_hbm_bytes_per_unit"]
    out = {"traffic": int(round(total * units)),
           "traffic_source": "profiles/%s %s: %.1f B per %s (FETCH_SIZE + WRITE_SIZE, separate --pmc passes, calibrated) x %d"
                             % (os.path.basename(tfiles[-1]), workload, total, wl.get("unit", "unit"), units)}
    if workload.startswith("hca_decode") and os.path.basename(tfiles[-1]) < "r06":
        # (the pass predates round 6's change of where the transform reads the code descriptions: tools/traffic_census.py models -2.1 KB read, +0.5 KB moved by the parse)
        out["traffic_source"] += "; counted before round 6 moved the code descriptions into the frame records"
    key = dom if dom in ks else next((k for k in ks if k.startswith(dom) or dom.startswith(k)), None)
    if key and ks[key].get("hbm_bytes_per_unit"):
        out["traffic_dominant_kernel"] = int(round(ks[key]["hbm_bytes_per_unit"] * units))
    return out


def committed_traffic_awb(hca_frames, adx_rows):
    """The mixed bank has no counter pass of its own: its traffic is composed from the per-unit figures of the same kernels in the HCA
    decode pass (per frame) and the ADX round trip's decode kernels (per block row)."""
    tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not tfiles:
        return None
    with open(tfiles[-1]) as fh:
        w = json.load(fh).get("workloads", {})
    h = (w.get("hca_decode") or {}).get("total_hbm_bytes_per_unit")
    a = sum(v.get("hbm_bytes_per_unit") or 0.0 for k, v in (w.get("adx_roundtrip") or {}).get("kernels", {}).items() if "decode" in k or "seg_fix" in k or "seg_serial" in k)
    if not h or not a:
        return None
    return {"traffic": int(round(h * hca_frames + a * adx_rows)),
            "traffic_source": "profiles/%s: %.1f B per HCA frame x %d + %.1f B per ADX block row x %d"
                              % (os.path.basename(tfiles[-1]), h, hca_frames, a, adx_rows)}


def committed_valu(units, kernel_ms, workload="hca_decode"):
    """What bounds the HCA kernels is VALU issue, not HBM: from the committed SQ counter passes of this path
    (profiles/r*_pmc.json: full-size batches from round 5 on) the wave64 VALU instructions per frame and `busy` = the share of a kernel's cycles its SIMDs
    spend issuing them (4 cycles each on one of 1024 SIMDs, against GRBM_GUI_ACTIVE: clock-independent).  floor_ms = the time this
    run's kernels would take if VALU issue were all they did (busy x measured time)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json"))) or sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_1000streams.json")))
    if not files:
        return None
    with open(files[-1]) as fh:
        pj = json.load(fh)
    per, busy, floor, batch = {}, {}, 0.0, None
    for name, k in (pj.get("workloads", {}).get(workload) or (pj.get("kernels", {}) if workload == "hca_decode" else {})).items():
        cls = "k_hca_parse" if "parse" in name else ("k_hca_transform" if "transform" in name else ("k_hca_encode" if "k_hca_encode" in name else None))
        if not cls or "VALU_per_frame" not in k:
            continue
        per[cls] = k["VALU_per_frame"]
        batch = k.get("frames_per_step", batch)
        if "valu_busy" in k:
            busy[cls] = k["valu_busy"]
            floor += k["valu_busy"] * kernel_ms.get(cls, 0.0)
    if not per:
        return None
    return {"valu": {"insts_per_frame": per, "busy": busy, "floor_ms": round(floor, 3) if busy else None,
                     "source": "profiles/%s (SQ_INSTS_VALU / GRBM_GUI_ACTIVE, %s)" % (os.path.basename(files[-1]), "%d frames per step" % batch if batch else "1000-stream batch")}}


# ------------------------------------------------------------------------------------------------ workloads
def make_hca_streams(unique, seconds, rank, quality, family):
    import oracle_lib as O
    return [O.hca_crypt(O.hca_encode(family_wav(1000 * rank + u, seconds, family), quality), 1, 56, KEY) for u in range(unique)]


def make_hca_streams_many(D, unique, seconds, quality, family, check=64):
    """`unique` DISTINCT encrypted streams for the headline at its written size ("10 000 synthetic encrypted stereo streams") without
    10 000 runs of the signal generator and of the CPU encoder: 64 base signals of the family; stream u is base u % 64 rotated by
    997 * (u // 64) samples at gain 1 - (u // 64) / 500 -- distinct PCM, the family's spectrum -- and the WAVs are encoded and enciphered
    ON THE DEVICE, a thousand at a time, by the library's own encoder, whose bytes are held to the oracle here on `check` of them spread
    over the batch (and on every item of every encode test and bench line)."""
    import numpy as np
    import torch
    import oracle_lib as O
    from pycricodecs_amd import synth
    from pycricodecs_amd.batch import Job
    n = int(48000 * seconds) // 32 * 32
    bases = [family_pcm(1000 * D.rank + b, n, 2, 48000, family).astype(np.int32) for b in range(min(64, unique))]

    def wav_of(u):
        k = u // len(bases)
        x = np.roll(bases[u % len(bases)], 997 * k, axis=0)
        return synth.wav_bytes((x * (500 - k % 400) // 500).astype(np.int16), 48000)
    out = []
    picks = set(range(0, unique, max(1, unique // check)))
    for lo in range(0, unique, 1000):
        wavs = oracle_many(wav_of, range(lo, min(unique, lo + 1000)), 32)
        ej = Job.hca_encode(wavs, quality=quality)
        eb = ej.alloc(D.dev)
        ej.run(*eb)
        torch.cuda.synchronize()
        assert int((eb[3] != 0).sum().item()) == 0
        plain = ej.split(bytes(eb[1].cpu().numpy()))
        for u in range(lo, lo + len(wavs)):
            if u in picks:
                assert bytes(plain[u - lo]) == O.hca_encode(wavs[u - lo], quality), "device encode of stream %d differs from the oracle's" % u
        cj = Job.hca_crypt([bytes(p) for p in plain], True, 56, keys=[KEY] * len(plain))
        cb = cj.alloc(D.dev)
        cj.run(*cb)
        torch.cuda.synchronize()
        out += [bytes(x) for x in cj.split(bytes(cb[1].cpu().numpy()))]
        del eb, cb, ej, cj
        torch.cuda.empty_cache()
    assert len(set(hash(x) for x in out)) == unique
    return out


def hca_decode_run(D, streams, unique, seconds, quality, family, steps, warmup, verify=True, uniq=None, sustain=0.0):
    """Decode of `streams` copies of `unique` streams (or of the prepared encrypted streams `uniq`).  Returns a result dict
    (rank-local verification, rank-reduced timing)."""
    import torch
    import oracle_lib as O
    from pycricodecs_amd.batch import Job
    if uniq is None:
        uniq = make_hca_streams(unique, seconds, D.rank, quality, family) if unique <= 256 else make_hca_streams_many(D, unique, seconds, quality, family)
    items = tile(uniq, streams)
    job = Job.hca_decode(items, keys=[KEY] * len(items))
    assert not job.host_status.any(), "synthetic inputs rejected at the header stage"
    bufs = job.alloc(D.dev)
    job.enable_events(True)
    dt, kms = run_timed(D, lambda: job.run(*bufs), steps, warmup, [job])
    assert not verify or int((bufs[3] != 0).sum().item()) == 0, "items failed on the device"
    fsz = int.from_bytes(uniq[0][0x1C:0x1E], "big")
    res = {"job": job, "dt": dt, "kernel_ms": kms, "units": job.units, "alg_bytes": job.algorithmic_bytes,
           # what each kernel of the path owns of a frame's algorithmic bytes: the parse reads the frame, the transform writes the PCM
           "alg_bytes_by_kernel": {"k_hca_parse": fsz * job.units, "k_hca_transform": job.algorithmic_bytes - fsz * job.units},
           "frame_size": fsz, "frames_per_stream": int.from_bytes(uniq[0][16:20], "big"), "channels": uniq[0][12],
           "census": job.record_census(bufs[2]), "sample": uniq[0],
           "bytes": {"in": job.input_bytes, "out": job.output_bytes, "scratch": job.scratch_bytes}}
    if sustain > 0:
        res["sustained"] = sustained_run(job, bufs, sustain)
    if verify:
        refs = oracle_many(lambda h: O.hca_decode(h, KEY), uniq)
        res["verified"] = verify_items(bufs[1], job.output_offsets, [i % len(uniq) for i in range(streams)], refs, "HCA decode")
        del refs
    del bufs
    torch.cuda.empty_cache()
    return res


def census_text(c):
    if not c["frames"]:
        return "n/a"
    return "%d of %d frames crossed scratch as int8 records, %d as int16" % (c["narrow"], c["frames"], c["frames"] - c["narrow"])


def hca_encode_run(D, streams, unique, seconds, quality, family, steps, warmup, verify=True, sustain=0.0):
    import torch
    import oracle_lib as O
    from pycricodecs_amd.batch import Job
    uniq = [family_wav(5000 + 1000 * D.rank + u, seconds, family) for u in range(unique)]
    job = Job.hca_encode(tile(uniq, streams), quality=quality)
    assert not job.host_status.any()
    bufs = job.alloc(D.dev)
    job.enable_events(True)
    dt, kms = run_timed(D, lambda: job.run(*bufs), steps, warmup, [job])
    assert int((bufs[3] != 0).sum().item()) == 0
    head = bytes(bufs[1][:96].cpu().numpy())                   # the first item's HCA header
    res = {"job": job, "dt": dt, "kernel_ms": kms, "units": job.units, "alg_bytes": job.algorithmic_bytes, "sample": uniq[0],
           "frames_per_stream": int.from_bytes(head[16:20], "big"), "frame_size": int.from_bytes(head[0x1C:0x1E], "big"),
           "bytes": {"in": job.input_bytes, "out": job.output_bytes, "scratch": job.scratch_bytes}}
    if sustain > 0:
        res["sustained"] = sustained_run(job, bufs, sustain)
    if verify:
        refs = [O.hca_encode(w, quality) for w in uniq]
        res["verified"] = verify_items(bufs[1], job.output_offsets, [i % len(uniq) for i in range(streams)], refs, "HCA encode")
    del bufs
    torch.cuda.empty_cache()
    return res


def adx_roundtrip_run(D, streams, unique, seconds, family, steps, warmup, verify=True):
    """ADX encode of the WAVs, then decode of the encoder's output where it lies (the decode job's input buffer IS the encode
    job's output buffer).  Both results are checked against the oracle for every file."""
    import torch
    import oracle_lib as O
    from pycricodecs_amd.batch import Job
    uniq = [family_wav(3000 + 1000 * D.rank + u, seconds, family) for u in range(unique)]
    adx_u = [O.adx_encode(w) for w in uniq]
    which = [i % len(uniq) for i in range(streams)]
    enc = Job.adx_encode(tile(uniq, streams))
    dec = Job.adx_decode(tile(adx_u, streams), offsets=enc.output_offsets)       # planned from the oracle's files: same headers, same layout
    assert not enc.host_status.any() and not dec.host_status.any()
    e_in, e_out, e_scr, e_st = enc.alloc(D.dev)
    _, d_out, d_scr, d_st = dec.alloc(D.dev, upload=False)
    enc.enable_events(True); dec.enable_events(True)

    def step():
        enc.run(e_in, e_out, e_scr, e_st)
        dec.run(e_out, d_out, d_scr, d_st)
    dt, kms = run_timed(D, step, steps, warmup, [enc, dec])
    assert int((e_st != 0).sum().item()) == 0 and int((d_st != 0).sum().item()) == 0
    res = {"dt": dt, "kernel_ms": kms, "units": enc.units + dec.units, "units2": enc.units2 + dec.units2,
           "alg_bytes": enc.algorithmic_bytes + dec.algorithmic_bytes, "alg_bytes_by_kernel": {enc.dominant_kernel: enc.algorithmic_bytes, dec.dominant_kernel: dec.algorithmic_bytes},
           "sample": uniq[0], "frames_per_stream": enc.units // streams,
           "bytes": {"in": enc.input_bytes, "adx": enc.output_bytes, "out": dec.output_bytes}}
    if verify:
        v1 = verify_items(e_out, enc.output_offsets, which, adx_u, "ADX encode")
        v2 = verify_items(d_out, dec.output_offsets, which, [O.adx_decode(a) for a in adx_u], "ADX decode of the encoder's output")
        res["verified"] = {"items": v1["items"] + v2["items"], "bytes": v1["bytes"] + v2["bytes"], "max_offset": max(v1["max_offset"], v2["max_offset"]),
                           "how": "every ADX file and every decoded WAV compared on the device, byte for byte, with the CPU oracle's (bit-exact check of configs[1])"}
    del e_in, e_out, e_scr, d_out, d_scr
    torch.cuda.empty_cache()
    return res


def hca_clip(h, nsamples):
    """An HCA file cut to its first `nsamples` samples as an encoder would have written the shorter clip: the frames that cover them
    (frames are independent: the frame bytes are the long file's, cipher included), the `fmt` chunk's frame count and appended-sample
    count rewritten (hca.cpp:693-707: samples = frames * 1024 - inserted - appended), the header checksum renewed."""
    import oracle_lib as O
    hs, fs = int.from_bytes(h[6:8], "big"), int.from_bytes(h[0x1C:0x1E], "big")
    delay = int.from_bytes(h[20:22], "big")
    nf = min(int.from_bytes(h[16:20], "big"), -(-(nsamples + delay) // 1024))
    nsamples = min(nsamples, nf * 1024 - delay)
    head = bytearray(h[:hs])
    head[16:20] = nf.to_bytes(4, "big")
    head[22:24] = (nf * 1024 - delay - nsamples).to_bytes(2, "big")
    head[hs - 2:hs] = O.crc16(bytes(head[:hs - 2])).to_bytes(2, "big")
    return bytes(head) + h[hs:hs + nf * fs]


def adx_clip(a, nsamples):
    """An ADX file cut to its first `nsamples` samples: the block rows that cover them (the encoder's state runs forward only, so the
    shorter clip's blocks are the long file's), the header's sample count rewritten (adx.cpp:300-340), the end-of-stream footer kept."""
    off, bs, ch = int.from_bytes(a[2:4], "big") + 4, a[5], a[7]
    spb = (bs - 2) * 8 // a[6]
    total = int.from_bytes(a[12:16], "big")
    rows_total = -(-total // spb)
    nsamples = min(nsamples, total)
    rows = -(-nsamples // spb)
    head = bytearray(a[:off])
    head[12:16] = nsamples.to_bytes(4, "big")
    return bytes(head) + a[off:off + rows * bs * ch] + a[off + rows_total * bs * ch:]


AWB_BASES = 24


def awb_clip_plan(n_total, rank, world, seed=77, family="tonal", n_durations=4096):
    """The mixed bank's clips and who decodes which (BASELINE configs[4]: "duration log-uniform ... per clip").  `n_durations` distinct
    clip lengths, log-uniform in 0.05-2 s and distinct to the sample (at most n_total), each as an HCA clip (High, encrypted with the
    bank's subkey) and as an ADX clip (bs18/bd4; its length rounded up to whole blocks of 32 samples): 2 x n_durations unique clips.  A clip is the head of one of 24 base signals of 2.1 s --
    the oracle encodes the 24 once per codec, hca_clip / adx_clip cut them to length the way an encoder would have written the shorter
    file (frame / block counts and sample counts in the headers, the HCA header checksum).  The global list is n_total draws from the
    unique clips (the same on every rank); this rank's share of it is longest-processing-time balanced by frame count
    (pycricodecs_amd.shard).  Returns (uniq, order_all, mine, subkey)."""
    import numpy as np
    import oracle_lib as O
    from pycricodecs_amd import shard, synth
    subkey = 0x2468
    rng = np.random.default_rng(seed)
    n_durations = max(1, min(n_durations, n_total))
    lens = set()
    while len(lens) < n_durations:                             # distinct sample counts
        lens.update(int(x) for x in np.exp(rng.uniform(np.log(0.05), np.log(2.0), n_durations - len(lens))) * 48000)
    lens = sorted(lens)
    rng.shuffle(lens)
    base_n = int(48000 * 2.1) // 32 * 32
    bases = []
    for u in range(AWB_BASES):
        w = synth.wav_bytes(family_pcm(7000 + u, base_n, 2, 48000, family), 48000)
        bases.append((O.hca_crypt(O.hca_encode(w, 1), 1, 56, KEY, subkey), O.adx_encode(w)))
    uniq = []
    for u, n in enumerate(lens):
        h, a = bases[u % AWB_BASES]
        uniq.append(("hca", hca_clip(h, n)))
        uniq.append(("adx", adx_clip(a, -(-n // 32) * 32)))      # whole blocks: the reference's own ADX decoder writes past its buffer for any other count (adx.cpp:392-415)
    order_all = rng.integers(0, len(uniq), n_total)
    wts = [shard.hca_weight(u[1]) if u[0] == "hca" else shard.adx_weight(u[1]) // 2 for u in uniq]
    mine = shard.my_items([wts[i] for i in order_all], rank, world) if world > 1 else list(range(n_total))
    return uniq, order_all, mine, subkey


def oracle_many(fn, items, threads=None):
    """fn over items on the host's cores (the oracle is a C library: its calls release the interpreter lock)."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(threads or min(64, os.cpu_count() or 8)) as ex:
        return list(ex.map(fn, items))


def build_awb_bank(n_total, rank, world, seed=77, family="tonal", n_durations=4096):
    """AFS2 bank of this rank's share of `n_total` short clips (BASELINE configs[4] shape, awb_clip_plan).  Returns (bank, uniq, order,
    subkey): order[i] = the unique clip behind the bank's item i."""
    import struct
    import numpy as np
    uniq, order_all, mine, subkey = awb_clip_plan(n_total, rank, world, seed, family, n_durations)
    align = 0x20
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


def awb_mixed_run(D, clips, steps, warmup, gather=False, verify=True, strong=False, family="tonal", n_durations=4096):
    """Decode of a mixed AFS2 bank through the AWB front door: one HCA job + one ADX job over the bank as it sits in HBM.
    With world > 1 every rank decodes its LPT share of clips x world clips and (gather=True) the decoded PCM of all
    ranks is collected on rank 0 inside the timed region -- the only collective-like step of the whole path."""
    import torch
    import oracle_lib as O
    from pycricodecs_amd import shard
    from pycricodecs_amd.batch import Job
    world = D.world
    bank, uniq, order, subkey = build_awb_bank(clips if strong else clips * world, D.rank, world, family=family, n_durations=n_durations)      # strong: `clips` is the whole job
    n = len(order)
    hj, aj = Job.awb_decode(bank, KEY)
    d_in, ho, hscr, hst = hj.alloc(D.dev)
    _, ao, ascr, ast = aj.alloc(D.dev, upload=False)
    gathered = {}
    aj_out = lambda t: t[:aj.output_bytes]

    # the two jobs are independent: the ADX one (one wave per file, as long as its longest clip) runs on a stream of its own and the
    # HCA kernels fill the rest of the chip meanwhile; both are inside the timed region (the main stream waits for the side one)
    side = torch.cuda.Stream(D.dev)
    hj.enable_events(True); aj.enable_events(True)

    def step():
        main = torch.cuda.current_stream(D.dev)
        side.wait_stream(main)
        aj.run(d_in, ao, ascr, ast, stream=side)
        hj.run(d_in, ho, hscr, hst)
        main.wait_stream(side)
        if gather and world > 1:
            # (launcher smoke test, ranks sharing one GPU over gloo: gloo has no device-tensor send / recv -- the parts travel as host copies)
            part = (lambda t: t.cpu()) if D.shared else (lambda t: t)
            gathered["hca"] = shard.gather_bytes_to_root(part(ho[:hj.output_bytes])); gathered["adx"] = shard.gather_bytes_to_root(part(aj_out(ao)))
    dt, kms = run_timed(D, step, max(steps, 1), max(warmup, 1), [hj, aj])
    hca_units, adx_units, n_all = D.reduce([float(hj.units), float(aj.units), float(n)], "sum")
    assert int((hst < 0).sum().item()) == 0 and int((ast < 0).sum().item()) == 0
    res = {"workload": "AFS2 bank(s) of %d clips%s (0.05-2 s log-uniform, %d distinct lengths, 48 kHz stereo, 50 %% ADX bs18/bd4 + 50 %% HCA High encrypted with subkey), decode to WAV%s"
                       % (int(n_all), " over %d GPUs, LPT-sharded" % world if world > 1 else "", len(uniq) // 2, ", PCM gathered on rank 0 (RCCL send/recv)" if gather and world > 1 else ""),
           "distinct_clips": len(set(order)), "distinct_lengths": len(uniq) // 2,
           "bank_bytes_rank0": len(bank), "pcm_bytes_rank0": int(hj.output_bytes + aj.output_bytes), "ms_per_step": round(dt * 1e3, 3),
           "hca_frames": int(hca_units), "adx_frames": int(adx_units), "frames_per_s": round((hca_units + adx_units) / dt, 1),
           "clips_per_s": round(n_all / dt, 1), "unit": "frames/s (HCA frames + ADX block rows)", "material": family}
    # the pieces of the line's roofline / cpu_baseline objects (this rank's jobs)
    dom_hca = max((k for k in kms if "hca" in k), key=lambda k: kms[k], default=None)
    dom_adx = max((k for k in kms if "adx" in k), key=lambda k: kms[k], default=None)
    res["_roofline_parts"] = {"kernel_ms": kms, "alg_bytes": hj.algorithmic_bytes + aj.algorithmic_bytes,
                              "alg_bytes_by_kernel": {k: v for k, v in ((dom_hca, hj.algorithmic_bytes), (dom_adx, aj.algorithmic_bytes)) if k}, "dt": dt}
    first_of = lambda kind: next((uniq[i][1] for i in order if uniq[i][0] == kind), None)
    res["_cpu_parts"] = {"hca": first_of("hca"), "adx": first_of("adx"), "hca_frames": int(hj.units), "adx_rows": int(aj.units), "subkey": subkey}
    if verify:
        used = set(order)
        refs = oracle_many(lambda t: None if t[0] not in used else (O.hca_decode(t[1][1], KEY, subkey) if t[1][0] == "hca" else O.adx_decode(t[1][1])), list(enumerate(uniq)))
        hi = [i for i in range(n) if uniq[order[i]][0] == "hca"]
        ai = [i for i in range(n) if uniq[order[i]][0] == "adx"]
        v1 = verify_items(ho, [hj.output_offsets[i] for i in hi], [order[i] for i in hi], refs, "AWB HCA items")
        v2 = verify_items(ao, [aj.output_offsets[i] for i in ai], [order[i] for i in ai], refs, "AWB ADX items")
        res["verified"] = {"items": v1["items"] + v2["items"], "bytes": v1["bytes"] + v2["bytes"], "how": v1["how"]}
        if gather and world > 1:
            # what arrived on the root: EVERY rank's items, each compared with the oracle's output for the clip it is a copy of (the ranks
            # send the root which clip sits where in their part), and every part's length and byte sum against its sender's own
            import torch.distributed as dist
            mine = [[order[i] for i in hi], [int(hj.output_offsets[i]) for i in hi], [order[i] for i in ai], [int(aj.output_offsets[i]) for i in ai],
                    int(ho[:hj.output_bytes].sum(dtype=torch.int64).item()), int(ao[:aj.output_bytes].sum(dtype=torch.int64).item()), int(hj.output_bytes), int(aj.output_bytes)]
            parts = [None] * world
            dist.all_gather_object(parts, mine)
            if D.rank == 0:
                need = set()
                for pr in parts:
                    need.update(pr[0]); need.update(pr[2])
                refs_all = oracle_many(lambda t: None if t[0] not in need else (refs[t[0]] if refs[t[0]] is not None else
                                                                                  (O.hca_decode(t[1][1], KEY, subkey) if t[1][0] == "hca" else O.adx_decode(t[1][1]))), list(enumerate(uniq)))
                items_root = 0
                for k, key in enumerate(("hca", "adx")):
                    got, offs = gathered[key]
                    got = got.to(D.dev)
                    assert offs[-1] == got.numel()
                    for rr in range(world):
                        part = got[offs[rr]:offs[rr + 1]]
                        assert part.numel() == parts[rr][6 + k] and int(part.sum(dtype=torch.int64).item()) == parts[rr][4 + k], "gathered PCM of rank %d differs" % rr
                        vv = verify_items(part, parts[rr][2 * k + 1], parts[rr][2 * k], refs_all, "gathered %s items of rank %d" % (key, rr))
                        items_root += vv["items"]
                res["gathered_bytes_on_root"] = int(gathered["hca"][0].numel() + gathered["adx"][0].numel())
                res["gathered_items_verified_on_root"] = items_root
                assert items_root == int(n_all)
    del d_in, ho, hscr, ao, ascr
    torch.cuda.empty_cache()
    return res


PROCESS_T0 = time.time()


def _full_size_in_budget(args, label):
    """The full-size form of a secondary is taken while the run is inside --full-secondary-budget seconds of wall time: the default run
    must end within minutes whatever the box (its one JSON line is printed last).  What is skipped says so in bench_detail.json."""
    spent = time.time() - PROCESS_T0
    if spent <= args.full_secondary_budget:
        return True
    log("%s at the headline's size: skipped (%.0f s of wall time spent, budget %.0f s)" % (label, spent, args.full_secondary_budget))
    return False


def _sec_sizes(args):
    n = args.secondary_streams
    return n, min(args.unique, 16), (args.streams if (args.streams > n and not args.no_full_secondary) else 0)      # (.., .., the headline's size: the chip filled ~18 times)


def _dec_entry(r, workload, extra=None):
    e = {"workload": workload, "frames_per_s": round(r["units"] / r["dt"], 1), "ms_per_step": round(r["dt"] * 1e3, 3), "frames": r["units"],
         "channel_frames_per_s": round(r["units"] * r["channels"] / r["dt"], 1),
         "frac_end_to_end": round(r["alg_bytes"] / r["dt"] / 1e9 / HBM_PEAK_GBPS, 5),
         "kernel_ms": {k: round(v, 3) for k, v in r["kernel_ms"].items()}, "transform_kernel": r["job"].dominant_kernel,
         # the transform instance of every format group (cri_hca_group_info.transform_form): 0 generic, 1 general, 2 / 3 / 4 in-lane plain / joint / noise fill, | 8 wide
         "transform_forms": r["job"].transform_forms(), "record_forms": census_text(r["census"]),
         "verified_items": r["verified"]["items"]}
    e.update(extra or {})
    r.pop("job", None)
    return e


def sec_host_paths(args, D, out):
    n, uq, full = _sec_sizes(args)
    # (first: the 44 GB of device buffers it needs are allocated while the device's memory is still in large pieces -- at the end of
    #  this list, after dozens of jobs' buffers have come and gone, the same downloads ran 10 % slower)
    out["hca_decode_host"] = host_path_run(min(args.streams, args.host_streams), uq, args.seconds)
    out["adx_decode_host"] = host_path_run(min(1000, args.streams), uq, args.seconds, codec="adx")


def sec_decode_families(args, D, out):
    """HCA decode of the other material families and qualities, stereo: at --secondary-streams and at the headline's size."""
    n, uq, full = _sec_sizes(args)
    for label, q, fam in (("hca_decode_sparse_spectra", 1, "sparse"), ("hca_decode_mixed", 1, "mixed"), ("hca_decode_noise", 1, "noise"), ("hca_decode_middle", 2, "tonal"),
                          ("hca_decode_low", 3, "tonal"), ("hca_decode_lowest", 4, "tonal")):
        uniq = make_hca_streams(uq, args.seconds, D.rank, q, fam)
        for tag, size in (("", n),) + ((("_full", full),) if full else ()):
            if tag and not _full_size_in_budget(args, label):
                out[label + tag] = {"skipped": "wall-time budget of the default run (--full-secondary-budget)"}
                continue
            r = hca_decode_run(D, size, uq, args.seconds, q, fam, 3, 1, uniq=uniq)
            out[label + tag] = _dec_entry(r, "HCA decode, %d x %.0f s encrypted stereo streams, quality %s, %s material" % (size, args.seconds, QNAME[q], fam))


def sec_decode_layouts(args, D, out):
    import oracle_lib as O
    n, uq, full = _sec_sizes(args)
    # the other layouts of the decode path: 6, 8 and 3 channels (plain formats: k_hca_transform_plain's wide form, a wave per four
    # channels), and k_hca_transform<false, 2> with the v3.0 noise fill (a v2.0 stream re-headed as v3.0 with min_resolution 0: every
    # band below the noise level is reconstructed).  Full size = as many channel-frames as the headline (streams x 2 / channels).
    import hca_forge
    nw = max(1, n // 4)
    for label, ch, v3, q in (("hca_decode_6ch", 6, False, 1), ("hca_decode_8ch", 8, False, 1), ("hca_decode_v3_noise_fill", 2, True, 1), ("hca_decode_3ch", 3, False, 1),
                             ("hca_decode_6ch_middle", 6, False, 2), ("hca_decode_5ch_middle", 5, False, 2),
                             ("hca_decode_6ch_v3_noise_fill", 6, True, 1)):      # (Middle: joint stereo + HFR, the wide joint form)
        plain = [O.hca_encode(family_wav(8000 + 10 * ch + u, args.seconds, "tonal", ch=ch), q) for u in range(4)]
        if v3:
            plain = [hca_forge.forge_v3(h, 0) for h in plain]
        uniq = [O.hca_crypt(h, 1, 56, KEY) for h in plain]
        for tag, size in (("", nw),) + ((("_full", max(1, full * 2 // ch)),) if full else ()):
            if tag and not _full_size_in_budget(args, label):
                out[label + tag] = {"skipped": "wall-time budget of the default run (--full-secondary-budget)"}
                continue
            r = hca_decode_run(D, size, 4, args.seconds, q, "tonal", 3, 1, uniq=uniq)
            out[label + tag] = _dec_entry(
                r, "HCA decode, %d x %.0f s encrypted %d-channel streams, quality %s%s" % (size, args.seconds, ch, QNAME[q], ", v3.0 header with min_resolution 0 (noise fill)" if v3 else ""))


def sec_encode_and_adx(args, D, out):
    n, uq, full = _sec_sizes(args)
    r = hca_encode_run(D, n, uq, args.seconds, 1, "tonal", 3, 1)
    out["hca_encode"] = {"workload": "HCA encode (quality High), %d x %.0f s 48 kHz stereo WAVs" % (n, args.seconds), "frames_per_s": round(r["units"] / r["dt"], 1),
                         "ms_per_step": round(r["dt"] * 1e3, 3), "frames": r["units"], "kernel_ms": {k: round(v, 3) for k, v in r["kernel_ms"].items()},
                         "achieved_GBps": round(r["alg_bytes"] / (max(r["kernel_ms"].values()) * 1e-3) / 1e9, 2), "verified_items": r["verified"]["items"]}
    r = adx_roundtrip_run(D, n, uq, args.seconds, "tonal", 3, 1)
    out["adx_roundtrip"] = {"workload": "ADX encode + decode (bs18/bd4/mode3/v4), %d x %.0f s 48 kHz stereo WAVs, decode reads the encoder's output buffer" % (n, args.seconds),
                            "frames_per_s": round(r["units"] / r["dt"], 1), "blocks_per_s": round(r["units2"] / r["dt"], 1), "ms_per_step": round(r["dt"] * 1e3, 3),
                            "chains": 2 * n, "kernel_ms": {k: round(v, 3) for k, v in r["kernel_ms"].items()},
                            "achieved_GBps": {k: round(r["alg_bytes_by_kernel"][k] / (v * 1e-3) / 1e9, 2) for k, v in r["kernel_ms"].items() if k in r["alg_bytes_by_kernel"]},
                            "verified_items": r["verified"]["items"]}


def sec_usm(args, D, out):
    import torch
    from pycricodecs_amd import usm
    from pycricodecs_amd.batch import Job
    import oracle_lib as O
    n, uq, full = _sec_sizes(args)
    # USM audio layer: ADX files as masked @SFA chunk streams, then the demux of a container made of them (HBM-bound copies)
    adx_u = [O.adx_encode(family_wav(3000 + u, args.seconds, "tonal")) for u in range(uq)]
    adx_items = tile(adx_u, 250)                               # (a container carries at most 256 audio channels)
    key = 0x0123456789ABCDEF
    for label, mk in (("sfa_pack", lambda: Job.sfa_pack(adx_items, usm.CODEC_ADX, key, True)),):
        job = mk()
        bufs = job.alloc(D.dev)
        job.enable_events(True)
        dt, kms = run_timed(D, lambda: job.run(*bufs), 3, 1, [job])
        packed = bufs[1].cpu().numpy().tobytes()
        out[label] = {"workload": "ADX files -> masked @SFA chunk streams (usm.py:584-657), %d x %.0f s" % (len(adx_items), args.seconds),
                      "chunks_per_s": round(job.units / dt, 1), "ms_per_step": round(dt * 1e3, 3), "chunks": job.units}
        del bufs
    torch.cuda.empty_cache()
    crid = b"CRID" + (0x18).to_bytes(4, "big") + bytes([0, 0x18]) + bytes(22)      # an empty CRID chunk: only the signature is read here
    job = Job.usm_audio_demux(crid + packed, key, True)
    bufs = job.alloc(D.dev)
    job.enable_events(True)
    dt, kms = run_timed(D, lambda: job.run(*bufs), 3, 1, [job])
    v = verify_items(bufs[1], job.output_offsets, [i % len(adx_u) for i in range(job.n)], adx_u, "USM demux")
    out["usm_demux"] = {"workload": "demux + AudioMask of that container (%d channels, %d chunks): gives the ADX files back" % (job.n, job.units),
                        "chunks_per_s": round(job.units / dt, 1), "ms_per_step": round(dt * 1e3, 3), "verified_items": v["items"]}
    del bufs
    torch.cuda.empty_cache()


def sec_awb(args, D, out):
    r = awb_mixed_run(D, args.awb_clips, 3, 1, n_durations=args.awb_durations)
    r.pop("_roofline_parts"); r.pop("_cpu_parts")
    out["awb_mixed_decode"] = r


SECONDARY_PARTS = ["sec_host_paths", "sec_decode_families", "sec_decode_layouts", "sec_encode_and_adx", "sec_usm", "sec_awb"]


def secondary_measurements(args, D):
    """Other rows of the path (N = 1 only), every output verified: HCA decode of the other qualities, material families (int16 records)
    and channel layouts -- at 1000 streams and at the headline's size --, HCA encode, ADX encode + decode, the USM @SFA layer, the mixed
    AWB bank, the host-memory paths, the single-file calls, and every other BASELINE configuration at its written size."""
    out = {}
    for name in SECONDARY_PARTS:                               # (host paths first: their 44 GB of device buffers are allocated while the device's memory is still in large pieces)
        globals()[name](args, D, out)
    out["single_call_ms"] = single_call_latency(args.seconds)
    lines = baseline_config_lines(args, D)
    # one compact object of everything above, next to the BASELINE configurations at the END of the line (the driver keeps the tail of stdout)
    out["summary_M_per_s"] = {k: round((v.get("frames_per_s") or v.get("chunks_per_s") or 0) / 1e6, 2) for k, v in out.items() if isinstance(v, dict) and ("frames_per_s" in v or "chunks_per_s" in v)}
    out["summary_M_per_s"]["single_call_ms [ours, reference]"] = out["single_call_ms"]
    out["baseline_configs"] = lines
    return out


def baseline_config_lines(args, D):
    """Every BASELINE.json configuration that is not the headline, at the size it is written, each as a line of its own inside the
    default run: value, ms_per_step, its own `roofline` (dominant kernel timed with HIP events in the timed steps) and `cpu_baseline`
    (the real reference on the host's cores, a few seconds each), every output verified on the device.
      configs[1]  ADX encode + decode round trip, 1 000 x 10 s stereo WAVs
      configs[3]  HCA encode (High), 10 000 x 30 s stereo WAVs (57.6 GB of PCM in HBM)
      configs[4]  mixed AWB bank, 100 000 short ADX + HCA clips on ONE GPU (the 8-GPU form is this batch dealt out: --scaling strong),
                  tonal material and the silence-padded SFX family"""
    import torch
    lines = {}
    cs = args.config_cpu_seconds
    ni = lambda n: max(4, int(round(n * args.config_items_scale)))
    sec = lambda x: max(0.3, x * args.config_seconds_scale)
    scaled = "" if (args.config_items_scale == 1.0 and args.config_seconds_scale == 1.0) else " [NOT the written size: items x %g, seconds x %g]" % (args.config_items_scale, args.config_seconds_scale)

    def line(r, workload, dtype, unit_bytes, cpu, traffic_key=None):
        kms = r["kernel_ms"]
        dom = max(kms, key=kms.get)
        alg_dom = r.get("alg_bytes_by_kernel", {}).get(dom, r["alg_bytes"])
        traffic = committed_traffic(int(r["units"]) // (2 if traffic_key.startswith("adx") else 1), dom, traffic_key) if traffic_key else None
        extra = {"bytes_per_unit": unit_bytes}
        if traffic_key in ("hca_encode", "hca_decode"):
            extra.update(committed_valu(int(r["units"]), kms, traffic_key) or {})
        out = {"workload": workload, "value": round(r["units"] / r["dt"], 1), "unit": "frames/s", "ms_per_step": round(r["dt"] * 1e3, 3), "dtype": dtype,
               "frames_per_step": int(r["units"]), "roofline": roofline_of(alg_dom, r["alg_bytes"], kms, r["dt"], traffic, extra),
               "verified": r.get("verified"), "cpu_baseline": cpu}
        return out
    # configs[2] again, as written: 10 000 DISTINCT streams (the headline tiles 64; make_hca_streams_many says how these are made)
    if args.streams >= 10000 and not args.no_distinct:
        r = hca_decode_run(D, args.streams, args.streams, args.seconds, 1, "tonal", 3, 1)
        lines["configs[2] hca_decode, %d distinct streams" % args.streams] = line(
            r, "BASELINE configs[2] with every stream distinct: HCA v2.0 decode, %d encrypted 48 kHz stereo streams x %.0f s, no two alike (64 base signals, each rotated and scaled per stream, encoded and enciphered on the device)" % (args.streams, args.seconds),
            "f32", "frame_size + 2*1024*channels = %d + 4096 B per frame" % r["frame_size"], None, "hca_decode")
        lines["configs[2] hca_decode, %d distinct streams" % args.streams]["record_forms"] = census_text(r["census"])
        r.pop("job", None)
        del r
        torch.cuda.empty_cache()
    # configs[1]
    r = adx_roundtrip_run(D, ni(1000), min(16, ni(1000)), sec(10.0), "tonal", 3, 1)
    lines["configs[1] adx_roundtrip"] = line(r, "BASELINE configs[1]: ADX encode + decode round trip (bs18/bd4/mode3/v4), 1000 48 kHz stereo WAVs x 10 s; a frame = one block row, counted for the encode and for the decode" + scaled,
                                             "int32", "blocksize + 2*samples_per_block = 82 B per block, encode and decode each", None if args.no_cpu else cpu_baseline("adxrt", r["sample"], 2 * r["frames_per_stream"], cs), "adx_roundtrip")
    r = adx_roundtrip_run(D, ni(1000), min(16, ni(1000)), sec(10.0), "sfx", 3, 1)
    lines["configs[1] adx_roundtrip, sfx material"] = line(r, "the same on the SFX family (0.05-0.5 s of digital silence before and after the sound)" + scaled, "int32", "82 B per block, encode and decode each", None, "adx_roundtrip_sfx")
    # configs[3]
    r = hca_encode_run(D, ni(10000), min(16, ni(10000)), sec(30.0), 1, "tonal", 3, 1)
    lines["configs[3] hca_encode"] = line(r, "BASELINE configs[3]: HCA encode (v2.0, quality High), 10000 48 kHz stereo WAVs x 30 s" + scaled, "f32",
                                          "frame_size + 2*1024*channels = %d + 4096 B per frame" % r["frame_size"], None if args.no_cpu else cpu_baseline("hcaenc", r["sample"], r["frames_per_stream"], cs), "hca_encode")
    r.pop("job", None)
    torch.cuda.empty_cache()
    # configs[4]
    for fam in ("tonal", "sfx"):
        r = awb_mixed_run(D, ni(args.config_awb_clips), 3, 1, family=fam, n_durations=args.awb_durations)
        rp, cp = r.pop("_roofline_parts"), r.pop("_cpu_parts")
        kms = rp["kernel_ms"]
        dom = max(kms, key=kms.get)
        lines["configs[4] awb_mixed" + ("" if fam == "tonal" else ", sfx material")] = {
            "workload": "BASELINE configs[4] on one GPU: " + r["workload"] + scaled, "value": r["frames_per_s"], "unit": r["unit"], "ms_per_step": r["ms_per_step"], "dtype": "f32+int32",
            "frames_per_step": r["hca_frames"] + r["adx_frames"], "clips_per_s": r["clips_per_s"], "bank_bytes": r["bank_bytes_rank0"], "pcm_bytes": r["pcm_bytes_rank0"],
            "roofline": roofline_of(rp["alg_bytes_by_kernel"].get(dom, rp["alg_bytes"]), rp["alg_bytes"], kms, rp["dt"], committed_traffic_awb(r["hca_frames"], r["adx_frames"]),
                                    {"bytes_per_unit": "HCA frame: frame_size + 4096 B; ADX block row: 2 x 82 B (stereo)"}),
            "verified": r.get("verified"), "cpu_baseline": None if (args.no_cpu or fam != "tonal") else cpu_baseline_awb(cp, cs)}
    return lines


def single_call_latency(seconds):
    """The five drop-in single-file calls (host bytes in, host bytes out: what PyCriCodecs' ADX / HCA classes call, hca.py:250, adx.py)
    on one stereo file of `seconds`, beside the reference's own time for the same call on one host core of this box."""
    import oracle_lib as O
    from pycricodecs_amd import CriCodecs as cc
    w = family_wav(9000, seconds, "tonal")
    hca = O.hca_crypt(O.hca_encode(w, 1), 1, 56, KEY)
    adx = O.adx_encode(w)
    hs = int.from_bytes(hca[6:8], "big")

    def ms(fn, n=10):
        fn()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        return round((time.perf_counter() - t0) / n * 1e3, 3)

    def ref(what, data, key=0):
        if not criref():
            return None
        reps, secs = criref_bench(what, data, 1.0, key)
        return round(secs / reps * 1e3, 3)
    assert cc.AdxDecode(adx) == O.adx_decode(adx) and cc.HcaDecode(hca, hs, KEY, 0) == O.hca_decode(hca, KEY)
    return {"file": "%.0f s 48 kHz stereo" % seconds, "unit": "ms per call, device path | reference C++ on one host core",
            "AdxDecode": [ms(lambda: cc.AdxDecode(adx)), ref("adxdec", adx)], "AdxEncode": [ms(lambda: cc.AdxEncode(w, 4, 18, 3, 500, 0, 4, False)), ref("adxenc", w)],
            "HcaDecode": [ms(lambda: cc.HcaDecode(hca, hs, KEY, 0)), ref("hcadec", hca, KEY)], "HcaEncode": [ms(lambda: cc.HcaEncode(w, False, 1)), ref("hcaenc", w)],
            "HcaCrypt": [ms(lambda: cc.HcaCrypt(hca, 0, hs, 0, KEY, 0)), None]}


def host_path_run(streams, unique, seconds, codec="hca"):
    """PCIe-inclusive: HCA (or ADX) decode of `streams` items from HOST memory to host memory -- what a caller without device
    buffers sees.  Two forms: the items' own `bytes` objects up (cri_job_run_host_items), WAVs down into a pageable numpy buffer
    (what Job.run_host() does for a Python caller); and one host blob up, WAVs down into page-locked memory (cri_job_run_host_into).
    Job planning (header parse of every item) is outside the timed call, as for the device-resident line."""
    import numpy as np
    import oracle_lib as O
    from pycricodecs_amd.batch import Job, pinned_array
    from pycricodecs_amd import _capi
    if codec == "adx":
        uniq = [O.adx_encode(family_wav(3000 + u, seconds, "tonal")) for u in range(unique)]
        items = tile(uniq, streams)
        job = Job.adx_decode(items)
        refs = [O.adx_decode(a) for a in uniq]
    else:
        uniq = make_hca_streams(unique, seconds, 0, 1, "tonal")
        items = tile(uniq, streams)
        job = Job.hca_decode(items, keys=[KEY] * len(items))
        refs = [O.hca_decode(h, KEY) for h in uniq]
    step = max(1, streams // 97)

    def timed(out, joined):
        job.run_host(out=out, joined=joined)
        best, st, outs = None, None, None
        for _ in range(3):
            t0 = time.perf_counter()
            outs, st = job.run_host(out=out, joined=joined)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        assert not st.any()
        checked = 0
        for i in list(range(0, streams, step)) + [streams - 1]:
            assert bytes(outs[i]) == refs[i % len(uniq)], "host path: item %d differs from the oracle" % i
            checked += 1
        return best, checked

    page = np.zeros(max(job.output_bytes, 1), dtype=np.uint8)
    t_items, checked = timed(page, False)
    del page
    pin = pinned_array(job.output_bytes)
    job.blob                                                   # the batch as one host blob (built once)
    t_blob, _ = timed(pin, True)
    del pin
    res = {"workload": ("ADX decode of %d x %.0f s stereo files from host memory to host memory (pipelined as parts over item ranges)" if codec == "adx" else
                        "HCA decode of %d x %.0f s encrypted stereo streams from host memory to host memory") % (streams, seconds),
           "share_of_link_rate": round(job.output_bytes / 57e9 / t_items, 3),
           "frames_per_s": round(job.units / t_items, 1), "ms": round(t_items * 1e3, 2),
           "form": "the items' own bytes objects (pageable) in, WAVs into a pageable numpy buffer: cri_job_run_host_items",
           "blob_form": {"frames_per_s": round(job.units / t_blob, 1), "ms": round(t_blob * 1e3, 2),
                         "form": "one pageable host blob in, WAVs into page-locked memory: cri_job_run_host_into"},
           "frames": job.units, "GBps_in_plus_out": round((job.input_bytes + job.output_bytes) / t_items / 1e9, 2),
           "bytes": {"in": job.input_bytes, "out": job.output_bytes}, "link": "PCIe: 57 GB/s each way measured (tools/debug/pcie_duplex.py); the download alone is %.1f ms" % (job.output_bytes / 57e9 * 1e3),
           "verified_items": checked, "verified_how": "every %d-th output of both forms compared on the host with the oracle's bytes; all statuses zero" % step}
    _capi.lib().cri_release_cache()
    return res


# ------------------------------------------------------------------------------------------------ main
def relaunch_under_torchrun(args):
    """--gpus N without torchrun's environment: start N ranks of this script on this node and pass their line through."""
    import torch
    have = torch.cuda.device_count()
    assert have >= args.gpus or os.environ.get("CRICODECS_BENCH_SHARE_GPU") == "1", "--gpus %d but only %d GPU(s) visible" % (args.gpus, have)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    log("launching %d ranks: %s" % (args.gpus, " ".join(cmd)))
    sys.exit(subprocess.run(cmd, env=env).returncode)


LINE_LIMIT = 4096                                              # bytes of the ONE stdout line (the driver's parser dropped round 5's 22 KB line)
DETAIL_NAME = "bench_detail.json"
DETAIL_ROOT = ROOT                                         # where emit() writes it (and $BENCH_DETAIL_DIR)


def _short(x, n):
    return x if not isinstance(x, str) or len(x) <= n else x[:n - 3] + "..."


def compact_line(full):
    """The ONE line the driver parses, from the full result: the contract's keys, `config` / `roofline` / `cpu_baseline` with their
    figures and short provenance strings, and where the rest is.  Everything that is left out here (secondaries, the other BASELINE
    configurations, long descriptions) is in bench_detail.json (emit)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "launcher_smoke_test")
    line = {k: _short(full[k], 120) for k in keep if k in full}
    cfg = full.get("config", {})
    c = {"workload": _short(cfg.get("workload", ""), 230)}
    for k in ("streams_per_gpu", "files_per_gpu", "frames_per_stream", "unique_streams", "unique_wavs", "chains", "batch_total", "hca_frames", "adx_frames", "distinct_lengths",
              "clips_per_s", "gathered_bytes_on_root", "gathered_items_verified_on_root", "record_forms"):
        if k in cfg:
            c[k] = _short(cfg[k], 90)
    if "parallelism" in cfg:
        c["parallelism"] = _short(cfg["parallelism"], 100)
    v = cfg.get("verified")
    if v:
        c["verified"] = {"items": v.get("items"), **({"items_all_ranks": v["items_all_ranks"]} if "items_all_ranks" in v else {}), "bytes": v.get("bytes"),
                         "how": "every item, on the device, byte for byte vs the CPU oracle"}
    if cfg.get("sustained"):
        c["sustained"] = cfg["sustained"]
    line["config"] = c
    r = full.get("roofline")
    if r:
        rr = {k: r[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_end_to_end", "traffic", "traffic_over_algorithmic", "algorithmic_bytes_per_launch",
                                "kernel_ms_per_step", "dominant_kernel") if k in r}
        if "valu" in r:
            rr["valu"] = {k: r["valu"][k] for k in ("insts_per_frame", "busy", "floor_ms") if k in r["valu"]}
        for k, n in (("bytes_per_unit", 80), ("traffic_source", 150)):
            if r.get(k):
                rr[k] = _short(r[k], n)
        line["roofline"] = rr
    b = full.get("cpu_baseline")
    if b:
        bb = {k: _short(b[k], 170) for k in ("value", "unit", "cores", "kind", "cpu_model", "sample") if k in b}
        if b.get("all_cores"):
            bb["all_cores"] = {k: b["all_cores"][k] for k in ("value", "cores") if k in b["all_cores"]}
        line["cpu_baseline"] = bb
    s = full.get("secondary") or {}
    if s.get("baseline_configs"):                              # the other BASELINE configurations of the default run: value only (all of it in the detail file)
        line["other_configs_M_per_s"] = {k.split(" ")[0] + ("/sfx" if "sfx" in k else "/distinct" if "distinct" in k else ""): round(v["value"] / 1e6, 1) for k, v in s["baseline_configs"].items()}
    line["detail"] = DETAIL_NAME
    return line


def emit(full):
    """Rank 0: the full result to bench_detail.json (repo root, and $BENCH_DETAIL_DIR when set -- gpurun_out/<tag> on a GPU box) and to
    stderr; ONE compact JSON line (<= LINE_LIMIT bytes) on stdout."""
    text = json.dumps(full, indent=1, default=str)
    for d in [DETAIL_ROOT] + ([os.environ["BENCH_DETAIL_DIR"]] if os.environ.get("BENCH_DETAIL_DIR") else []):
        try:
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, DETAIL_NAME), "w") as fh:
                fh.write(text + "\n")
        except OSError as e:
            log("could not write %s: %s" % (os.path.join(d, DETAIL_NAME), e))
    log("---- detail (also in %s) ----" % DETAIL_NAME)
    log(json.dumps(full, default=str))
    line = json.dumps(compact_line(full))
    if len(line) > LINE_LIMIT:                                 # never again a line the driver cannot read: drop the optional parts
        c = compact_line(full)
        for k in ("other_configs_M_per_s",):
            c.pop(k, None)
        for k in ("traffic_source", "bytes_per_unit", "valu", "dominant_kernel"):
            c.get("roofline", {}).pop(k, None)
        c.get("cpu_baseline", {}).pop("sample", None)
        line = json.dumps(c)
    assert len(line) <= LINE_LIMIT, len(line)
    print(line, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="hca_decode", choices=["hca_decode", "hca_encode", "adx_roundtrip", "awb_mixed"])
    ap.add_argument("--streams", type=int, default=None, help="items per GPU (default: 10000; adx_roundtrip 1000)")
    ap.add_argument("--unique", type=int, default=None, help="distinct inputs tiled to --streams (default 64; hca_encode 16)")
    ap.add_argument("--seconds", type=float, default=None, help="seconds per item (default 10; hca_encode 30)")
    ap.add_argument("--data", default="tonal", choices=["tonal", "sparse", "noise", "mixed", "sfx"], help="signal family of the synthetic inputs (see family_pcm)")
    ap.add_argument("--quality", type=int, default=1, help="HCA quality: 1 = High (the headline), 2 Middle (intensity stereo), 3 Low (HFR), 4 Lowest")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --streams (--awb-clips) items PER GPU; strong: that many in total, dealt out to the ranks (BASELINE configs[3] / [4]: a fixed batch, file-sharded 1 -> 8)")
    ap.add_argument("--no-gather", action="store_true", help="awb_mixed with --gpus > 1: leave the decoded PCM on the ranks")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--sustain", type=float, default=4.0, help="seconds of the same job in a loop after the timed steps: config.sustained (single GPU; not with --no-secondary / --no-verify / --no-cpu, the profiling runs' flags)")
    ap.add_argument("--no-distinct", action="store_true", help="skip the all-streams-distinct form of the headline inside the default run")
    ap.add_argument("--no-verify", action="store_true", help="skip the all-items check against the oracle (profiling runs)")
    ap.add_argument("--secondary-streams", type=int, default=1000)
    ap.add_argument("--no-full-secondary", action="store_true", help="secondaries at --secondary-streams only, not also at the headline's size")
    ap.add_argument("--full-secondary-budget", type=float, default=150.0, help="the full-size forms of the secondaries are taken while the run has spent less wall time than this (seconds); the rest of the default run (round 5: 142 s in all) follows whatever it says")
    ap.add_argument("--host-streams", type=int, default=10000, help="streams of the host-memory secondary (host output buffers: 1.92 MB each)")
    ap.add_argument("--awb-clips", type=int, default=12500, help="clips per GPU of the mixed AWB bank (100 000 / 8 GPUs)")
    ap.add_argument("--awb-durations", type=int, default=4096, help="distinct clip lengths of the mixed AWB bank (log-uniform 0.05-2 s; each as an HCA and as an ADX clip)")
    ap.add_argument("--config-awb-clips", type=int, default=100000, help="clips of the configs[4] line inside the default run (one GPU)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="seconds of reference CPU work of the headline's cpu_baseline (single thread; again on all cores)")
    ap.add_argument("--config-items-scale", type=float, default=1.0, help="dry runs / quick checks ONLY: the other BASELINE configurations inside the default run at this fraction of their written item counts (anything but 1 is labelled in their workload strings)")
    ap.add_argument("--config-seconds-scale", type=float, default=1.0, help="the same for their seconds per item")
    ap.add_argument("--config-cpu-seconds", type=float, default=4.0, help="seconds of reference CPU work per baseline of the configs lines (single thread; again on all cores)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_under_torchrun(args)
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        log("note: --gpus %d but WORLD_SIZE=%s; the launcher's world size is used" % (args.gpus, os.environ["WORLD_SIZE"]))
    D = Dist()
    wl = args.workload
    streams = args.streams or (1000 if wl == "adx_roundtrip" else 10000)
    unique = args.unique or (16 if wl == "hca_encode" else 64)
    seconds = args.seconds or (30.0 if wl == "hca_encode" else 10.0)
    verify = not args.no_verify
    args.streams, args.unique, args.seconds = streams, unique, seconds
    strong = args.scaling == "strong"
    total_streams = streams
    if strong:                                                 # this rank's share of the fixed batch (equal items: dealt round robin; the AWB bank: LPT by frames)
        streams = len(range(D.rank, total_streams, D.world))
    common = {"n_gpus": D.world, **({"launcher_smoke_test": "ranks share one GPU over gloo; not a measurement"} if D.shared and D.world > 1 else {}), "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None}
    par = "file-sharded x%d, one process per GPU, no data-path collective%s" % (D.world, " (a fixed batch of %d dealt out to the ranks)" % total_streams if strong else "")
    t_setup = time.time()

    if wl == "awb_mixed":                                      # BASELINE configs[4]; its own metric line
        r = awb_mixed_run(D, args.awb_clips, args.steps, args.warmup, gather=not args.no_gather, verify=verify, strong=strong, family=args.data if args.data in ("tonal", "sfx") else "tonal", n_durations=args.awb_durations)
        rp, cp = r.pop("_roofline_parts"), r.pop("_cpu_parts")
        if D.rank == 0:
            kms = rp["kernel_ms"]
            dom = max(kms, key=kms.get)
            common = dict(common, roofline=roofline_of(rp["alg_bytes_by_kernel"].get(dom, rp["alg_bytes"]), rp["alg_bytes"], kms, rp["dt"], committed_traffic_awb(r["hca_frames"], r["adx_frames"]),
                                                       {"bytes_per_unit": "HCA frame: frame_size + 4096 B; ADX block row: 2 x 82 B (stereo); rank 0's jobs"}))
            if not args.no_cpu:
                common["cpu_baseline"] = cpu_baseline_awb(cp)
            emit(dict(common, metric="audio frames/sec, mixed AWB bank decode (BASELINE configs[4])", value=r["frames_per_s"], unit="frames/s",
                      ms_per_step=r["ms_per_step"], dtype="f32+int32", data="synthetic (%d distinct clip lengths x 2 codecs, heads of 24 base signals; %s family)" % (r["distinct_lengths"], r["material"]),
                      config=r))
        D.close()
        return

    sustain = args.sustain if (D.world == 1 and not (args.no_secondary or args.no_verify or args.no_cpu)) else 0.0      # (profiling runs count dispatches: nothing extra in them)
    if wl == "hca_decode":
        r = hca_decode_run(D, streams, unique, seconds, args.quality, args.data, args.steps, args.warmup, verify, sustain=sustain)
        cfg = {"workload": "BASELINE configs[2]: HCA v2.0 decode, %d encrypted 48 kHz stereo streams x %.0f s (quality %s, frame %d B, key 0xCF222F1FE0748978) per GPU"
                           % (streams, seconds, QNAME.get(args.quality, "?"), r["frame_size"]),
               "streams_per_gpu": streams, "frames_per_stream": r["frames_per_stream"], "unique_streams": unique, "parallelism": par,
               "record_forms": census_text(r["census"])}
        if strong:
            cfg["batch_total"] = total_streams
        dtype, unit_bytes = "f32", "frame_size + 2*1024*channels = %d + 4096 B per frame" % r["frame_size"]
        cpu = ("hcadec", r["sample"], r["frames_per_stream"])
    elif wl == "hca_encode":
        r = hca_encode_run(D, streams, unique, seconds, args.quality, args.data, args.steps, args.warmup, verify, sustain=sustain)
        cfg = {"workload": "BASELINE configs[3]: HCA encode (v2.0, quality %s), %d 48 kHz stereo WAVs x %.0f s per GPU" % (QNAME.get(args.quality, "?"), streams, seconds),
               "streams_per_gpu": streams, "frames_per_stream": r["frames_per_stream"], "unique_wavs": unique, "parallelism": par}
        if strong:
            cfg["batch_total"] = total_streams
        dtype, unit_bytes = "f32", "frame_size + 2*1024*channels = %d + 4096 B per frame" % r["frame_size"]
        cpu = ("hcaenc", r["sample"], r["frames_per_stream"])
    else:
        r = adx_roundtrip_run(D, streams, unique, seconds, args.data, args.steps, args.warmup, verify)
        cfg = {"workload": "BASELINE configs[1]: ADX encode + decode round trip (bs18/bd4/mode3/v4), %d 48 kHz stereo WAVs x %.0f s per GPU; a frame = one block row (32 samples x 2 channels), counted once for the encode and once for the decode"
                           % (streams, seconds), "files_per_gpu": streams, "chains": 2 * streams, "unique_wavs": unique, "parallelism": par}
        dtype, unit_bytes = "int32", "blocksize + 2*samples_per_block = 18 + 64 = 82 B per block, encode and decode each"
        cpu = ("adxrt", r["sample"], 2 * r["frames_per_stream"])
    log("%s: setup + run + verify %.1fs; %s" % (wl, time.time() - t_setup, {k: "%.2f GB" % (v / 1e9) for k, v in r["bytes"].items()}))
    if r.get("sustained"):
        cfg["sustained"] = r["sustained"]
    if verify:
        cfg["verified"] = r["verified"]
        ok = D.reduce([float(r["verified"]["items"])], "sum")[0]
        cfg["verified"]["items_all_ranks"] = int(ok)
    units, dt, kms = r["units"], r["dt"], r["kernel_ms"]
    units_all = D.reduce([float(units)], "sum")[0]             # the whole job's units (weak: world x this rank's; strong: the fixed batch)
    dom = max(kms, key=kms.get)
    alg_dom = r.get("alg_bytes_by_kernel", {}).get(dom, r["alg_bytes"])
    traffic = None
    if args.quality == 1:
        if wl == "hca_decode" and args.data in ("tonal", "sparse"):
            traffic = committed_traffic(units, dom, "hca_decode" if args.data == "tonal" else "hca_decode_sparse")
        elif wl == "hca_encode" and args.data == "tonal":
            traffic = committed_traffic(units, dom, "hca_encode")
        elif wl == "adx_roundtrip" and args.data in ("tonal", "sfx"):
            traffic = committed_traffic(units // 2, dom, "adx_roundtrip" if args.data == "tonal" else "adx_roundtrip_sfx")
    extra = {"bytes_per_unit": unit_bytes}
    if wl in ("hca_decode", "hca_encode"):
        extra.update(committed_valu(units, kms, wl) or {})
    roof = roofline_of(alg_dom, r["alg_bytes"], kms, dt, traffic, extra)
    out = dict(common, metric="audio frames/sec (decode+encode) at 1/2/4/8 GPU; HBM GB/s vs roofline", value=round(units_all / dt, 1), unit="frames/s",
               ms_per_step=round(dt * 1e3, 3), dtype=dtype,
               data="synthetic, %s family (%s); %d unique tiled to %d, each copy in its own HBM" % (
                   args.data, {"tonal": "seeded sines + noise floor", "sparse": "sparse tones, low-passed noise", "noise": "full-scale noise / square / clicks",
                               "mixed": "tonal and sparse alternating", "sfx": "tonal between digital silence"}[args.data], unique, streams),
               config=cfg, roofline=roof)
    # other rows of the same hot path and the host-core baseline: single-GPU run only (ranks of a scaling run must not wait)
    if D.rank == 0 and D.world == 1 and wl == "hca_decode" and not args.no_secondary:
        out["secondary"] = secondary_measurements(args, D)
    rank, world = D.rank, D.world
    if world > 1:
        D.close()                                              # (the other ranks are done; the host-core baseline is rank 0's alone)
    if D.numa:
        out["config"]["host"] = dict(D.numa, note="the rank runs on the cores of its GPU's socket (matters to the host-memory lines only)")
    if rank == 0 and not args.no_cpu:
        os.sched_setaffinity(0, D.affinity0)                   # the host-core baseline gets every core of the box again
        out["cpu_baseline"] = cpu_baseline(*cpu, seconds=args.cpu_seconds)
    if rank == 0:
        emit(out)
    if world == 1:
        D.close()


if __name__ == "__main__":
    main()
