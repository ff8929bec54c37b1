"""GPU parity, round-4 additions (through the C ABI, against the pinned oracle): exhaustive checks of the float shortcuts inside the
encoders (the ADX quantisers, the HCA encoder's band-cost rule) run on the device over their whole domains, a seeded differential
fuzz of the segmented ADX kernels, the hipGraph capture the header promises, decode layouts of 9 .. 16 channels, job lifetime."""
import ctypes as C
import os

import numpy as np
import pytest

import golden_util as G
import oracle_lib as O
from pycricodecs_amd import synth

pytestmark = pytest.mark.gpu
KEY = G.KEY


@pytest.fixture(scope="module")
def cc():
    from pycricodecs_amd import CriCodecs, _capi
    assert _capi.lib().cri_device_available() == 1, "no HIP device: the GPU tests must run on the HIP path"
    return CriCodecs


def run_job(job, stream=None):
    if os.environ.get("CRI_TEST_HOST_RUN") == "1":             # tools/asan_gpu.sh: no torch in the process -- through the library's own host path
        outs, st = job.run_host()
        return [bytes(o) for o in outs], st
    import torch
    bufs = job.alloc("cuda:0")
    job.run(*bufs, stream=stream)
    torch.cuda.synchronize()
    blob = bytes(bufs[1].cpu().numpy())
    status = bufs[3].cpu().numpy()[:job.n]
    return job.split(blob), status


# ------------------------------------------------------------------------------------------------ a5: the float quantisers, every case
@pytest.mark.parametrize("form,bitdepth", [(0, b) for b in range(2, 9)] + [(1, 4)], ids=lambda v: str(v))
def test_adx_float_quantisers_exhaustive(cc, form, bitdepth):
    """The ADX encoders quantise in float (csrc/cri_adx_quant.h: AdxQuantSmall in k_adx_encode for bit depths <= 8, AdxQuantLane in
    k_adx_lane_encode) where the reference divides integers (adx.cpp:256-261).  The comments argue an error bound; here the same
    device functions are held against the integer rule for EVERY delta in [-2^18, 2^18) -- more than ((sample << 12) - prediction)
    >> 12 can reach -- times every scale a block can carry (1 .. 4096, and mode 4's 8192): 2.1 G cases per bit depth."""
    from pycricodecs_amd import _capi
    with _capi.testing_knobs() as L:
        L.cri_test_adx_quantisers.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.POINTER(C.c_int32)]
        cases, bad, first = C.c_ulonglong(), C.c_ulonglong(), (C.c_int32 * 4)()
        assert L.cri_test_adx_quantisers(form, bitdepth, -(1 << 18), (1 << 18) - 1, C.byref(cases), C.byref(bad), first) == 0
        assert cases.value == 4097 * (1 << 19)
        assert bad.value == 0, "delta %d scale %d: got %d, the reference's rule gives %d" % tuple(first)


# ------------------------------------------------------------------------------------------------ a32: the encoder's band cost, every float
def test_hca_encoder_band_cost_rule_on_the_device(cc):
    """k_hca_encode never quantises in its rate loop: a spectrum is classed once and a band costs 8 * shortest + (classes that reach
    the resolution's rank) - (the clamp-value anomaly) (csrc/cri_hca_enc_cost.h).  The same device functions against
    CalculateUsedBits' inner loop (hca.cpp:2771-2786) at all fifteen resolutions: every magnitude 0 .. 0.9999999f (ScaleSpectra's
    clamp) of both signs, in bands of eight consecutive bit patterns -- 1.07 G floats x 2 signs, nothing sampled."""
    from pycricodecs_amd import _capi
    with _capi.testing_knobs() as L:
        tab = (C.c_uint8 * 16384)()
        L.cri_test_enc_tables.argtypes = [C.c_void_p, C.c_size_t]
        assert L.cri_test_enc_tables(tab, 16384) > 0
        L.cri_test_enc_band_cost.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.POINTER(C.c_uint32)]
        cases, bad, first = C.c_ulonglong(), C.c_ulonglong(), (C.c_uint32 * 4)()
        clamp = 0x3F7FFFFE
        total = 0
        chunk = 1 << 24                                              # bands per launch
        bands = (clamp + 8) // 8
        for b0 in range(0, bands, chunk):
            n = min(chunk, bands - b0)
            assert L.cri_test_enc_band_cost(tab, 8 * b0, 1, n, C.byref(cases), C.byref(bad), first) == 0
            assert bad.value == 0, "first spectrum %08x, resolution %d: %d bits, the reference's rule gives %d" % tuple(first)
            total += cases.value
        assert total == bands * 2 * 15


# ------------------------------------------------------------------------------------------------ a3 / a5: differential fuzz of the segmented ADX kernels
def _material(rng, n, ch):
    """(n, ch) int16 of a randomly chosen family: tonal + noise floor, full-scale noise, pure tones (limit cycles), digital silence
    with bursts (game-SFX shape), square waves, a constant."""
    kind = int(rng.integers(0, 6))
    t = np.arange(n)[:, None]
    if kind == 0:
        x = sum(rng.uniform(500, 9000) * np.sin(2 * np.pi * rng.uniform(50, 12000) / 48000 * t + c) for c in range(3)) + rng.normal(0, rng.uniform(1, 300), (n, ch))
    elif kind == 1:
        x = rng.integers(-32768, 32768, (n, ch)).astype(np.float64)
    elif kind == 2:
        x = rng.uniform(1000, 32000) * np.sin(2 * np.pi * rng.uniform(100, 8000) / 48000 * t + np.arange(ch)[None, :])
    elif kind == 3:
        x = np.zeros((n, ch))
        for _ in range(int(rng.integers(1, 5))):
            a = int(rng.integers(0, max(1, n - 1))); b = min(n, a + int(rng.integers(16, 4000)))
            x[a:b] = rng.normal(0, rng.uniform(50, 9000), (b - a, ch))
    elif kind == 4:
        x = rng.uniform(2000, 30000) * np.sign(np.sin(2 * np.pi * rng.uniform(30, 3000) / 48000 * t + 0.1))
        x = np.repeat(x, ch, axis=1) if x.shape[1] == 1 else x
    else:
        x = np.full((n, ch), float(rng.integers(-3000, 3000)))
    x = np.asarray(x, dtype=np.float64) * np.ones((1, ch))
    m = min(512, n)
    x[:m] *= ((np.arange(m) / 512.0) ** 2)[:, None]                       # (a first scale word >= 0x100 is rejected by the reference's own decoder)
    return np.clip(np.round(x), -32768, 32767).astype(np.int16)


@pytest.mark.parametrize("batch", range(16))
def test_adx_segmented_kernels_differential_fuzz(cc, knobs, batch):
    """2048 decode cases and 2048 encode cases in 16 seeded batches (tools/debug/adx_lane_cases.py's shapes, promoted): every batch
    draws a mapping (segmented decode; wave-per-segment or lane-per-segment encode), a warm-up of 100 / 30 / 5 / 1 % and a least
    segment length, then 128 files of random length (1 .. 2500 rows), channel count, mode 2 / 3 / 4, high-pass 0 .. 65535 and
    material (tonal, full-scale noise, pure tones, silence with bursts, squares, DC); a quarter of the decode inputs carry an
    `80 01` end marker at a random row or are cut short.  Every output byte is the oracle's."""
    from pycricodecs_amd.batch import Job
    rng = np.random.default_rng(9000 + batch)
    warm = [100, 30, 5, 1][batch % 4]
    # ---- encode
    mode = [3, 3, 2, 4][(batch // 4) % 4]
    hp = int([500, 0, 65535, int(rng.integers(1, 20000))][batch % 4]) if mode != 2 else 500
    enc_map = "lane" if batch % 2 else "seg"
    knobs(adx_mapping=enc_map, adx_warm_pct=warm, adx_seglen=[0, 10, 3, 1][(batch // 2) % 4] if enc_map == "lane" else 0)
    wavs = []
    for k in range(128):
        rows = int(np.exp(rng.uniform(0, np.log(2500))))
        ch = int(rng.integers(1, 3))
        n = 32 * rows - int(rng.integers(0, 32)) * int(rng.integers(0, 2))
        wavs.append(synth.wav_bytes(_material(rng, max(n, 1), ch), int(rng.choice([48000, 44100, 22050]))))
    job = Job.adx_encode(wavs, mode=mode, highpass=hp)
    outs, st = run_job(job)
    refs = [O.adx_encode(w, 4, 18, mode, hp, 0, 4) for w in wavs]
    assert not st.any() and not job.host_status.any()
    for i, (o, r) in enumerate(zip(outs, refs)):
        assert bytes(o) == r, ("encode", batch, i, job.dominant_kernel)
    # ---- decode (of those files, some damaged; modes 2 / 3 segment, mode 4 takes the unsegmented kernels)
    knobs(adx_mapping="seg", adx_warm_pct=warm, adx_seglen=[0, 1, 2, 5][(batch // 2) % 4])
    files = []
    for k, r in enumerate(refs):
        b = bytearray(r)
        do = int.from_bytes(b[2:4], "big") + 4
        ch = b[7]
        rows = (len(b) - do) // (18 * ch)
        what = int(rng.integers(0, 8))
        if what == 0 and rows > 1:                                   # end marker on a row's first block
            row = int(rng.integers(0, rows))
            b[do + row * 18 * ch:do + row * 18 * ch + 2] = b"\x80\x01"
        elif what == 1 and rows > 1:                                 # cut inside a row
            b = b[:do + int(rng.integers(1, rows * 18 * ch))]
        files.append(bytes(b))
    job = Job.adx_decode(files)
    outs, st = run_job(job)
    for i, f in enumerate(files):
        try:
            want = O.adx_decode(f)
        except O.OracleError as e:
            assert job.host_status[i] == e.code or st[i] == e.code, ("decode status", batch, i)
            continue
        assert not job.host_status[i] and not st[i], ("decode", batch, i)
        assert bytes(outs[i]) == want, ("decode", batch, i, job.dominant_kernel)


# ------------------------------------------------------------------------------------------------ b: cri_job_run inside a hipGraph
@pytest.mark.parametrize("kind", ["hca_decode", "hca_encode", "adx_decode", "adx_encode", "hca_crypt"])
def test_job_run_captured_in_a_hip_graph(cc, kind):
    """include/cricodecs_hip.h: "cri_job_run only enqueues kernels on the caller's stream and can be captured in a hipGraph".  One
    capture, three replays over zeroed output buffers, every replay's bytes are the oracle's (small banks are launch-bound: this is
    how a caller amortises the launches)."""
    import torch
    from pycricodecs_amd.batch import Job
    wavs = [synth.wav(4100 + k, 32 * (40 + 300 * k), 1 + k % 2, 48000) for k in range(6)]
    if kind == "hca_decode":
        items = [O.hca_crypt(O.hca_encode(w, 1 + k % 3), 1, 56, KEY) for k, w in enumerate(wavs)]
        job, refs = Job.hca_decode(items, keys=[KEY] * len(items)), [O.hca_decode(h, KEY) for h in items]
    elif kind == "hca_encode":
        job, refs = Job.hca_encode(wavs, quality=2), [O.hca_encode(w, 2) for w in wavs]
    elif kind == "adx_decode":
        items = [O.adx_encode(w) for w in wavs]
        job, refs = Job.adx_decode(items), [O.adx_decode(a) for a in items]
    elif kind == "adx_encode":
        job, refs = Job.adx_encode(wavs), [O.adx_encode(w) for w in wavs]
    else:
        items = [O.hca_encode(w, 1) for w in wavs]
        job, refs = Job.hca_crypt(items, True, 56, keys=[KEY] * len(items)), [O.hca_crypt(h, 1, 56, KEY) for h in items]
    bufs = job.alloc("cuda:0")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        job.run(*bufs)                                             # (uncaptured once: modules loaded, arena-free path)
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        job.run(*bufs)
    for _ in range(3):
        bufs[1].zero_(); bufs[3].fill_(-1)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        assert int(bufs[3][:job.n].abs().sum().item()) == 0
        outs = job.split(bytes(bufs[1].cpu().numpy()))
        for i, (o, r) in enumerate(zip(outs, refs)):
            assert bytes(o) == r, (kind, i)
    del g


# ------------------------------------------------------------------------------------------------ a10 / a23: 9 .. 16 channels
@pytest.mark.parametrize("ch", [9, 12, 16])
def test_hca_decode_of_nine_to_sixteen_channels(cc, ch):
    """clHCA_DecodeHeader takes up to 16 channels (hca.cpp:662-687) and the reference decodes them; the wide forms of the in-lane
    transform are built for eight (two waves of four), so 9 .. 16 go to k_hca_transform_generic.  Forged streams (an 8-channel
    header re-written to `ch` channels with a frame size that fits them, seeded sparse random frames the oracle accepts), plain
    and with joint-stereo bands, v2.0 and v3.0: PCM equal to the oracle's."""
    import hca_forge
    from pycricodecs_amd.batch import Job
    base = O.hca_encode(synth.wav(77, 1024 * 12, 8, 48000), 1)

    def takes(stream):
        try:
            O.hca_decode(stream)
            return True
        except O.OracleError:
            return False
    items = []
    for k, (stereo, v3) in enumerate([(0, False), (8, False), (0, True)]):
        b = bytearray(hca_forge.forge_header(base, frame_size=4000))
        assert bytes(x & 0x7F for x in b[8:12]) == b"fmt\0"
        b[0x0C] = ch
        hca_forge.fix_header_crc(b)
        h = bytes(b)
        hs = int.from_bytes(h[6:8], "big")
        h = h[:hs] + bytes(4000 * int.from_bytes(h[0x10:0x14], "big"))
        bb = h[0x23]
        h = hca_forge.forge_comp(h, track_count=1, channel_config=0, total=bb, base=bb - stereo, stereo=stereo, hfr=0)
        if v3:
            h = hca_forge.forge_v3(h, 0)
        f = hca_forge.accepted_random_stream(h, 500 + 10 * ch + k, 0.04, takes)
        assert f is not None, (ch, k)
        items.append(f)
    job = Job.hca_decode(items)
    assert job.dominant_kernel.startswith("k_hca")
    outs, st = run_job(job)
    assert not st.any() and not job.host_status.any()
    for i, (o, h) in enumerate(zip(outs, items)):
        assert bytes(o) == O.hca_decode(h), (ch, i)


# ------------------------------------------------------------------------------------------------ b: a job may die while its kernels run
def test_job_destroyed_with_work_in_flight(cc):
    """cri_job_destroy waits for the job's last enqueued run before its metadata allocation is recycled into the next job
    (ADVICE r3): a job is run on a side stream and dropped at once, a second job of the same size is planned and run right
    behind it; both outputs are the oracle's."""
    import gc
    import torch
    from pycricodecs_amd.batch import Job
    a_items = [O.hca_crypt(O.hca_encode(synth.wav(4300 + k, 48000 * 2, 2, 48000), 1), 1, 56, KEY) for k in range(4)] * 60
    b_items = [O.hca_encode(synth.wav(4400 + k, 48000 * 2, 1, 48000), 3) for k in range(4)] * 60
    a_refs = [O.hca_decode(h, KEY) for h in a_items[:4]]
    b_refs = [O.hca_decode(h) for h in b_items[:4]]
    side = torch.cuda.Stream()
    for _ in range(3):
        ja = Job.hca_decode(a_items, keys=[KEY] * len(a_items))
        bufs_a = ja.alloc("cuda:0")
        offs_a = ja.output_offsets.copy()
        torch.cuda.synchronize()
        ja.run(*bufs_a, stream=side)
        del ja
        gc.collect()                                               # (the handle is destroyed here, kernels possibly still running)
        jb = Job.hca_decode(b_items)
        outs_b, st_b = run_job(jb)
        torch.cuda.synchronize()
        assert not st_b.any()
        for i in range(len(b_items)):
            assert bytes(outs_b[i]) == b_refs[i % 4], i
        assert int(bufs_a[3].abs().sum().item()) == 0
        blob = bytes(bufs_a[1].cpu().numpy())
        for i in range(len(a_items)):
            o = int(offs_a[i])
            assert blob[o:o + len(a_refs[i % 4])] == a_refs[i % 4], i


# ------------------------------------------------------------------------------------------------ a3: digital silence inside segmented files
@pytest.mark.parametrize("warm", [100, 1])
def test_adx_segmented_decode_through_runs_of_silence(cc, knobs, warm):
    """Digital silence parks the decoder at a history-dependent fixed point of its predictor (adx.cpp:208-212 with zero codes), so
    speculative segments inside it never merge; k_adx_seg_fix derives the state behind a RUN of silent segments from the last segment
    with sound (cycle-detecting walk of the recurrence) instead of repairing one segment per round.  Clips with silent heads, tails
    and gaps many segments long, mono / stereo / four channels, three coefficient sets, mode 2 (static filter 0 in silent blocks) --
    every byte the oracle's, on the segmented kernels."""
    from pycricodecs_amd.batch import Job
    knobs(adx_mapping="seg", adx_warm_pct=warm)
    rng = np.random.default_rng(77)
    files = []
    for k, (n, ch, mode, hp) in enumerate([(48000 * 3, 2, 3, 500), (48000 * 2, 1, 3, 500), (48000 * 3, 2, 3, 4000), (48000 * 2, 2, 2, 500), (48000 * 2, 4, 3, 100),
                                            (48000 * 4, 2, 3, 500), (48000 * 1, 2, 3, 500), (32 * 9000, 1, 3, 20000)]):
        x = synth.pcm16(6000 + k, n, ch, 48000).astype(np.int32)
        cuts = sorted(int(c) for c in rng.integers(0, n, 6))
        x[:cuts[0]] = 0                                            # head
        x[cuts[1]:cuts[2]] = 0                                     # a gap
        x[cuts[3]:cuts[4]] = 0                                     # another one
        x[cuts[5]:] = 0                                            # tail
        if k == 5:
            x[:] = 0; x[48000:48000 + 4000] = 12000                # a click in an otherwise silent file
        if k == 6:
            x[:] = 0                                               # nothing but silence
        files.append(O.adx_encode(synth.wav_bytes(x.astype(np.int16), 48000), 4, 18, mode, hp, 0, 4))
    job = Job.adx_decode(files)
    assert job.dominant_kernel == "k_adx_seg_decode"
    outs, st = run_job(job)
    assert not st.any() and not job.host_status.any()
    for i, (o, f) in enumerate(zip(outs, files)):
        assert bytes(o) == O.adx_decode(f), i


# ------------------------------------------------------------------------------------------------ a20: v3.0 noise fill on the wide in-lane form
@pytest.mark.parametrize("ch,q", [(3, 1), (5, 1), (6, 1), (6, 2), (7, 1), (8, 1), (8, 2)])
def test_hca_v3_noise_fill_on_wide_layouts(cc, ch, q):
    """reconstruct_noise (hca.cpp:1602-1635) for 3 and 5 .. 8 channels runs on the wide instance of the in-lane transform: the
    generator's draws run on through a frame's channels, so the workgroup's waves (four channels each) trade their draw counts per
    step.  Streams re-headed as v3.0 with min_resolution 0 (every band under the noise level is reconstructed), longer than a run
    of eight frames, encrypted and plain: PCM and the floats before the int16 conversion equal to the oracle's, bit for bit."""
    import hca_forge
    import torch
    from pycricodecs_amd.batch import Job
    items, keys = [], []
    for k, n in enumerate((1024 * 19 + 300, 5000, 1024 * 9)):
        h = hca_forge.forge_v3(O.hca_encode(synth.wav(8800 + 10 * ch + k, n, ch, 48000), q), 0)
        try:
            O.hca_decode(h)
        except O.OracleError:
            continue                                               # (a layout the reference rejects under a v3.0 header)
        if k == 1:
            h = O.hca_crypt(h, 1, 56, KEY)
        items.append(h); keys.append(KEY if k == 1 else 0)
    if not items:
        pytest.skip("the reference rejects this layout under a v3.0 header")
    job = Job.hca_decode(items, keys=keys)
    assert all(f == (4 | 8) for f in job.transform_forms()), job.transform_forms()
    bufs = job.alloc("cuda:0")
    d_f, offs = job.run_floats(*bufs)
    torch.cuda.synchronize()
    assert int(bufs[3].abs().sum().item()) == 0
    outs = job.split(bytes(bufs[1].cpu().numpy()))
    fl = d_f.cpu().numpy()
    for i, (h, key) in enumerate(zip(items, keys)):
        assert bytes(outs[i]) == O.hca_decode(h, key), (ch, q, i)
        want = O.hca_decode_float(h, key)
        got = fl[int(offs[i]):int(offs[i + 1])]
        assert got.size == want.size and np.array_equal(got.view(np.uint32), np.asarray(want, dtype=np.float32).reshape(-1).view(np.uint32)), (ch, q, i)


# ------------------------------------------------------------------------------------------------ the pipelined host path, ADX decode
@pytest.mark.parametrize("order", ["pipelined", "pipelined-small-pieces", "one-piece"])
def test_adx_decode_run_host_in_parts(cc, knobs, order):
    """An ADX decode job cannot be cut inside (its lanes are laid out by length, not by item), so its pipelined host path plans the
    items again as a few jobs over consecutive item ranges (run_host_core, host_parts_ready) that write where the whole job would:
    bytes and statuses equal the device-resident run's and the oracle's -- from separate items (staged in small pieces too), from a
    pageable blob and from a page-locked one, into pageable and page-locked memory, with items the host rejects among them."""
    from pycricodecs_amd import _capi
    from pycricodecs_amd.batch import Job, pinned_array
    knobs(host_slice_min=(1 << 62) if order == "one-piece" else 0)
    if order == "pipelined-small-pieces":
        knobs(host_stage_piece=1000)
    rng = np.random.default_rng(91)
    wavs = [synth.wav(700 + k, int(rng.integers(40, 60000)), 1 + k % 2, 48000) for k in range(10)]
    uniq = [O.adx_encode(w) for w in wavs] + [O.adx_encode(wavs[2], bitdepth=8), O.adx_encode(wavs[3], mode=4)]
    items = [uniq[int(k)] for k in rng.integers(0, len(uniq), 90)]
    items[11] = b"\x80\x00" + bytes(64)                          # rejected on the host
    items[57] = uniq[1][:40]                                    # a header cut short
    job = Job.adx_decode(items)
    assert job.host_status[11] != 0 and job.host_status[57] != 0
    want, st_dev = run_job(job)
    st_want = np.where(job.host_status != 0, job.host_status, st_dev)
    refs = {id(u): O.adx_decode(u) for u in uniq}
    for i, it in enumerate(items):
        if id(it) in refs:
            assert bytes(want[i]) == refs[id(it)], i
    for rep in range(2):                                        # (the second call finds the parts made by the first)
        outs, st = job.run_host()
        assert (st == st_want).all()
        for i, (a, b) in enumerate(zip(outs, want)):
            assert bytes(a) == bytes(b), (rep, i)
    buf = pinned_array(job.output_bytes)
    outs, st = job.run_host(out=buf)
    for i, (a, b) in enumerate(zip(outs, want)):
        assert bytes(a) == bytes(b), i
    blob = job.blob
    status = (C.c_int32 * job.n)()
    buf[:] = 0xEE
    assert _capi.lib().cri_job_run_host_into(job._h, blob, buf.ctypes.data, status) == 0
    assert (np.array(status[:job.n]) == st_want).all()
    for i, (a, b) in enumerate(zip(job.split(memoryview(buf)), want)):
        assert bytes(a) == bytes(b), i
    pin_in = pinned_array(len(blob))
    pin_in[:] = np.frombuffer(blob, dtype=np.uint8)
    page_out = np.full(job.output_bytes, 0xEE, dtype=np.uint8)
    assert _capi.lib().cri_job_run_host_into(job._h, pin_in.ctypes.data, page_out.ctypes.data, status) == 0
    for i, (a, b) in enumerate(zip(job.split(memoryview(page_out)), want)):
        assert bytes(a) == bytes(b), i
    o = job.output_offsets                                      # bytes no kernel writes are zero
    for i in range(job.n - 1):
        end = int(o[i]) + len(want[i])
        assert not page_out[end:int(o[i + 1])].any(), i
    del outs
    del buf, pin_in


@pytest.mark.parametrize("order", ["pipelined", "pipelined-small-pieces"])
def test_hca_decode_of_mixed_formats_run_host_in_parts(cc, knobs, order):
    """An HCA decode job of several format groups (channel counts, qualities, keys) runs its groups one after the other, each over
    items from anywhere in the batch: the pipelined host path plans it again as parts over item ranges, like an ADX decode job."""
    from pycricodecs_amd.batch import Job
    knobs(host_slice_min=0)
    if order == "pipelined-small-pieces":
        knobs(host_stage_piece=1500)
    rng = np.random.default_rng(17)
    uniq, keys = [], []
    for k, (ch, q, key) in enumerate([(2, 1, KEY), (1, 1, 0), (2, 4, KEY), (6, 1, 0), (2, 2, 12345), (4, 3, KEY), (2, 1, 0)]):
        h = O.hca_encode(synth.wav(900 + k, int(rng.integers(3000, 40000)), ch, 48000), q)
        uniq.append(O.hca_crypt(h, 1, 56, key) if key else h); keys.append(key)
    pick = rng.integers(0, len(uniq), 60)
    items = [uniq[int(k)] for k in pick]
    item_keys = [keys[int(k)] for k in pick]
    items[9] = b"HCA\0" + bytes(100)
    job = Job.hca_decode(items, keys=item_keys)
    assert len(job.transform_forms()) > 3
    want, st_dev = run_job(job)
    st_want = np.where(job.host_status != 0, job.host_status, st_dev)
    refs = [O.hca_decode(u, k) for u, k in zip(uniq, keys)]
    for i, k in enumerate(pick):
        if i != 9:
            assert bytes(want[i]) == refs[int(k)], i
    for rep in range(2):
        outs, st = job.run_host()
        assert (st == st_want).all()
        for i, (a, b) in enumerate(zip(outs, want)):
            assert bytes(a) == bytes(b), (rep, i)


# ------------------------------------------------------------------------------------------------ transform runs of 16 and 32 frames
@pytest.mark.parametrize("run", [16, 32])
def test_hca_decode_with_long_transform_runs(cc, knobs, run):
    """The planner gives large format groups transform runs of 16 or 32 frames instead of 8 (cri_capi.cpp: fewer halo passes); the
    parity batches are far too small for that, so the knob forces it: every transform form -- in-lane plain / joint / noise fill, the
    wide instances, the general kernels -- over streams shorter than, equal to and several times a
    run, encrypted and plain, PCM and the floats before the int16 conversion bit for bit the oracle's."""
    import hca_forge
    import torch
    from pycricodecs_amd.batch import Job
    knobs(hca_run=run)
    items, keys = [], []
    specs = [(2, 1, False), (1, 1, False), (2, 3, False), (2, 4, False), (4, 1, False), (6, 1, False), (6, 2, False), (8, 1, False),
             (2, 1, True), (6, 1, True), (3, 2, True)]
    for k, (ch, q, v3) in enumerate(specs):
        for m, n in enumerate((1024 * (run - 1) + 17, 1024 * run, 1024 * (2 * run + 3) + 500, 3000)):
            h = O.hca_encode(synth.wav(9100 + 10 * k + m, n, ch, 48000), q)
            if v3:
                h = hca_forge.forge_v3(h, 0)
                try:
                    O.hca_decode(h)
                except O.OracleError:
                    continue
            key = KEY if (k + m) % 2 else 0
            items.append(O.hca_crypt(h, 1, 56, key) if key else h); keys.append(key)
    job = Job.hca_decode(items, keys=keys)
    assert len(set(job.transform_forms())) >= 4, job.transform_forms()
    bufs = job.alloc("cuda:0")
    d_f, offs = job.run_floats(*bufs)
    torch.cuda.synchronize()
    assert int(bufs[3].abs().sum().item()) == 0
    outs = job.split(bytes(bufs[1].cpu().numpy()))
    fl = d_f.cpu().numpy()
    for i, (h, key) in enumerate(zip(items, keys)):
        assert bytes(outs[i]) == O.hca_decode(h, key), (run, i)
        want = O.hca_decode_float(h, key)
        mine = fl[int(offs[i]):int(offs[i + 1])]
        assert mine.size == want.size and np.array_equal(mine.view(np.uint32), np.asarray(want, dtype=np.float32).reshape(-1).view(np.uint32)), (run, i)


# ------------------------------------------------------------------------------------------------ where decoded WAVs are placed
def test_decoded_wavs_start_their_samples_on_a_line(cc):
    """cri_job_output_offsets of the decode jobs: every WAV is placed so that the samples behind its header (44 bytes, 112 with a
    smpl chunk) start a 128-byte line -- the decoders store PCM in whole sample rows, which are then whole lines -- items do not overlap,
    the bytes between them are zero after a run, and the items are the oracle's."""
    from pycricodecs_amd.batch import Job
    rng = np.random.default_rng(5)
    wavs = [synth.wav(40 + k, int(rng.integers(100, 9000)), 1 + k % 2, 48000) for k in range(6)]
    wavs.append(synth.wav_bytes(synth.pcm16(99, 6000, 2, 48000), 48000, loop=(1000, 5000)))
    adx = [O.adx_encode(w) for w in wavs]
    hca = [O.hca_encode(w, 1) for w in wavs]
    for job, refs in ((Job.adx_decode(adx), [O.adx_decode(a) for a in adx]), (Job.hca_decode(hca), [O.hca_decode(h) for h in hca])):
        outs, st = run_job(job)
        import torch
        bufs = job.alloc("cuda:0"); job.run(*bufs); torch.cuda.synchronize()
        blob = bufs[1].cpu().numpy()
        o = [int(v) for v in job.output_offsets]
        assert o[-1] == job.output_bytes and o[-1] % 64 == 0
        end = 0
        for i, ref in enumerate(refs):
            hdr = 0x70 if ref[0x24:0x28] == b"smpl" else 0x2C
            assert (o[i] + hdr) % 128 == 0 and o[i] >= end, (i, o[i], hdr)
            assert not blob[end:o[i]].any(), i
            assert bytes(blob[o[i]:o[i] + len(ref)]) == ref == bytes(outs[i]), i
            end = o[i] + len(ref)
        assert not blob[end:].any()
    assert any(r[0x24:0x28] == b"smpl" for r in refs)
