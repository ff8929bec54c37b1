"""GPU parity, round-2 additions (all through the C ABI, checked against the oracle and the reference-generated vectors):
header forms (f2), the pre-clamp floats north_star states the HCA tolerance on, the ADX parameter space, long streams,
shard-vs-unsharded equality, threads, device selection, and the planner regressions the round-1 advisor found."""
import struct
import threading

import numpy as np
import pytest

import golden_util as G
import hca_forge
import oracle_lib as O
from pycricodecs_amd import synth

pytestmark = pytest.mark.gpu
KEY = G.KEY
MAN = G.manifest()


@pytest.fixture(scope="module")
def cc():
    from pycricodecs_amd import CriCodecs, _capi
    assert _capi.lib().cri_device_available() == 1, "no HIP device: the GPU tests must run on the HIP path"
    return CriCodecs


def run_job(job, floats=False):
    import torch
    bufs = job.alloc("cuda:0")
    fl = None
    if floats:
        fl = job.run_floats(*bufs)
    else:
        job.run(*bufs)
    torch.cuda.synchronize()
    blob = bytes(bufs[1].cpu().numpy())
    status = bufs[3].cpu().numpy()[:job.n]
    return job.split(blob), status, fl


# ------------------------------------------------------------------------------------------------ f2: header forms
@pytest.mark.parametrize("f", MAN["header_forms"], ids=lambda f: f["file"])
def test_header_forms_golden(cc, f):
    """v1.x `dec` chunk, `vbr` / `ath` / `rva` / `comm` (hca.cpp:710-830): decode digests of the reference, its rejections, and
    its HcaCrypt output bytes (CryptHeader, hca.cpp:3166-3250) in both directions."""
    h = G.load(f["file"])
    hs = int.from_bytes(h[6:8], "big")
    if f["decoded_sha"] is None:
        with pytest.raises(ValueError):
            cc.HcaDecode(h, hs, 0, 0)
    else:
        assert G.sha(cc.HcaDecode(h, hs, 0, 0)) == f["decoded_sha"]
    for label in ("enc56", "enc1", "enc56_sub"):
        e = f[label]
        if e is None:
            with pytest.raises(ValueError):
                cc.HcaCrypt(h, 1, hs, 56 if label != "enc1" else 1, 1, 0)
            continue
        key = int(e["key"], 16)
        enc = cc.HcaCrypt(h, 1, hs, e["type"], key, e["subkey"])
        assert G.sha(enc) == e["sha"]
        assert G.sha(cc.HcaCrypt(enc, 0, hs, 0, key, e["subkey"])) == e["decrypted_sha"]
        if e["decoded_sha"] is None:
            with pytest.raises(ValueError):
                cc.HcaDecode(enc, hs, key, e["subkey"])
        else:
            assert G.sha(cc.HcaDecode(enc, hs, key, e["subkey"])) == e["decoded_sha"]


def test_header_forms_batch(cc):
    """the same streams as one batch job (several formats, ATH tables and cipher tables in one launch set)"""
    from pycricodecs_amd.batch import Job
    ents = [f for f in MAN["header_forms"]]
    items = [G.load(f["file"]) for f in ents]
    job = Job.hca_decode(items)
    outs, status, _ = run_job(job)
    for f, o, st, hst in zip(ents, outs, status, job.host_status):
        if f["decoded_sha"] is None:
            assert hst != 0 or st != 0, f["file"]
        else:
            assert hst == 0 and st == 0 and G.sha(o) == f["decoded_sha"], f["file"]


# ------------------------------------------------------------------------------------------------ pre-clamp floats
def test_device_floats_match_reference_digests(cc):
    """north_star: HCA within 1 ULP on the PCM floats before the int16 clamp.  The device's wave[][] (validation run of the
    decode job) is bit-identical (0 ULP) to the reference's: sha256 over the float bytes of every golden, forged, fuzz and
    header-form stream equals the digest the real reference produced (tests/golden/make_golden*.py)."""
    from pycricodecs_amd.batch import Job
    ents = []
    for case in MAN["cases"]:
        ents += [(h["file"], h["float_sha"], h["decoded_sha"]) for h in case["hca"]]
    ents += [(f["file"], f["float_sha"], f["decoded_sha"]) for f in MAN["forged"]]
    ents += [(f["file"], f["float_sha"], f["decoded_sha"]) for f in MAN["header_forms"] if f["float_sha"]]
    items = [G.load(e[0]) for e in ents]
    job = Job.hca_decode(items)
    outs, status, (d_f, offs) = run_job(job, floats=True)
    fl = d_f.cpu().numpy()
    assert not status.any() and not job.host_status.any()
    for i, (name, fsha, dsha) in enumerate(ents):
        mine = fl[int(offs[i]):int(offs[i + 1])]
        assert G.sha(mine.tobytes()) == fsha, name
        assert G.sha(outs[i]) == dsha, name


@pytest.mark.parametrize("ch,q,v3", [(1, 1, False), (2, 1, False), (2, 2, False), (2, 3, False), (2, 4, False), (4, 1, False), (4, 2, False),
                                      (6, 1, False), (8, 3, False), (3, 1, False), (5, 2, False), (2, 2, True), (2, 1, True), (6, 2, True)])
def test_device_floats_vs_oracle(cc, ch, q, v3):
    """every transform instance (plain 1/2/4, general 1..8 channels, generic odd layouts, v3.0 noise fill): floats equal to
    the oracle's bit patterns, whole streams, and the PCM16 of the same run equals the normal run's"""
    from pycricodecs_amd.batch import Job
    items = []
    for seed, n in ((40, 5000), (41, 12000), (42, 1024), (43, 300)):
        h = O.hca_encode(synth.wav(seed + ch, n, ch, 48000), q)
        items.append(hca_forge.forge_v3(h, 0) if v3 else h)
    job = Job.hca_decode(items)
    outs, status, (d_f, offs) = run_job(job, floats=True)
    fl = d_f.cpu().numpy()
    good = 0
    for i, h in enumerate(items):
        try:
            ref = O.hca_decode_float(h)
        except O.OracleError:                                  # (a forged v3.0 header on frames of another layout: both sides reject it)
            assert status[i] != 0, i
            continue
        assert status[i] == 0, i
        good += 1
        mine = fl[int(offs[i]):int(offs[i + 1])]
        assert mine.size == ref.size
        assert np.array_equal(mine.view(np.uint32), ref.view(np.uint32)), (i, int(np.argmax(mine.view(np.uint32) != ref.view(np.uint32))))
        assert outs[i] == O.hca_decode(h)
    assert good >= 2


def test_device_floats_random_frames(cc):
    """random-byte frames (saturating samples, escape codes, reads past the frame end): the floats agree bit for bit too"""
    from pycricodecs_amd.batch import Job
    items = []
    for q, ch in ((1, 2), (2, 2), (3, 2), (1, 1)):
        base = O.hca_encode(synth.wav(0, 800, ch, 48000), q)
        for seed in range(16):
            f = hca_forge.random_frames(base, seed, density=1.0 if seed % 2 else 0.35)
            try:
                O.hca_decode(f)
            except O.OracleError:
                continue
            items.append(f)
    assert len(items) > 8
    job = Job.hca_decode(items)
    outs, status, (d_f, offs) = run_job(job, floats=True)
    fl = d_f.cpu().numpy()
    for i, h in enumerate(items):
        ref = O.hca_decode_float(h)
        assert np.array_equal(fl[int(offs[i]):int(offs[i + 1])].view(np.uint32), ref.view(np.uint32)), i


# ------------------------------------------------------------------------------------------------ ADX parameter space
@pytest.mark.parametrize("hp", [0, 100, 4000, 20000, 65535])
@pytest.mark.parametrize("mapping", ["chain", "file"])
def test_adx_highpass_frequencies(cc, hp, mapping, knobs):
    """Highpass_Frequency != 500 (CalculateCoefficients, adx.cpp:58-64) through encode and decode, both kernel mappings"""
    knobs(adx_mapping=mapping)
    for seed, n, ch, sr in ((7, 4800, 2, 48000), (8, 3008, 1, 22050)):
        w = synth.wav(seed, n, ch, sr)
        for mode in (3, 4):
            ref = O.adx_encode(w, 4, 18, mode, hp, 0, 4)
            assert cc.AdxEncode(w, 4, 18, mode, hp, 0, 4, False) == ref, (hp, mode)
            assert cc.AdxDecode(ref) == O.adx_decode(ref), (hp, mode)


@pytest.mark.parametrize("filt", [0, 1, 2, 3])
@pytest.mark.parametrize("mapping", ["chain", "file"])
def test_adx_static_filters(cc, filt, mapping, knobs):
    """EncodingMode 2 with Filter 0..3 (static coefficient sets, adx.cpp:434, 463-468; the filter rides in the top bits of
    every block's scale word, 247) and the decoder's per-block predictor select"""
    knobs(adx_mapping=mapping)
    from pycricodecs_amd.batch import Job
    wavs = [synth.wav(50 + i, 3200 + 640 * i, 1 + i % 2, [48000, 44100, 32000][i % 3]) for i in range(5)]
    for bd, bs in ((4, 18), (8, 18), (6, 26)):
        refs = [O.adx_encode(w, bd, bs, 2, 500, filt, 4) for w in wavs]
        for w, r in zip(wavs, refs):
            assert cc.AdxEncode(w, bd, bs, 2, 500, filt, 4, False) == r, (filt, bd, bs)
            if filt == 0:
                assert cc.AdxDecode(r) == O.adx_decode(r), (filt, bd, bs)
            else:
                # the reference rejects its own filter >= 1 files: the first scale word's high byte (filter << 5) sits where it
                # expects the NUL that ends "(c)CRI" (adx.cpp:345-348, SURVEY 8(c) caveat 4); same error here
                with pytest.raises(O.OracleError):
                    O.adx_decode(r)
                with pytest.raises(ValueError, match="copyright"):
                    cc.AdxDecode(r)
        outs, status, _ = run_job(Job.adx_encode(wavs, bitdepth=bd, blocksize=bs, mode=2, filt=filt))
        assert not status.any() and [bytes(o) for o in outs] == refs
        # the decoder's per-block predictor select (adx.cpp:196-203) on files that pass the header check: blocks with filter
        # bits set anywhere but in the very first scale word
        if filt:
            spliced = []
            for w, r in zip(wavs, refs):
                r0 = O.adx_encode(w, bd, bs, 2, 500, 0, 4)
                hs = int.from_bytes(r0[2:4], "big") + 4
                spliced.append(r0[:hs + bs * r0[7]] + r[hs + bs * r0[7]:])     # first block row from the filter-0 file
            outs, status, _ = run_job(Job.adx_decode(spliced))
            assert not status.any() and [bytes(o) for o in outs] == [O.adx_decode(r) for r in spliced]
        else:
            outs, status, _ = run_job(Job.adx_decode(refs))
            assert not status.any() and [bytes(o) for o in outs] == [O.adx_decode(r) for r in refs]
    with pytest.raises(ValueError, match="Filter"):
        cc.AdxEncode(wavs[0], 4, 18, 2, 500, 4, 4, False)


def test_adx_mixed_filters_in_one_decode_batch(cc):
    """files made with different filters / highpass frequencies / modes in one decode job (per-stream coefficients)"""
    from pycricodecs_amd.batch import Job
    items = []
    for i in range(24):
        w = synth.wav(70 + i, 1600 + 320 * (i % 7), 1 + i % 2, [48000, 44100][i % 2])
        mode = [2, 3, 4][i % 3]
        a = O.adx_encode(w, 4, 18, mode, [0, 100, 500, 4000, 20000][i % 5], i % 4 if mode == 2 else 0, [3, 4, 5][i % 3])
        if mode == 2 and i % 4:                                # keep the filter bits out of the first scale word (see test_adx_static_filters)
            a0 = O.adx_encode(w, 4, 18, 2, 500, 0, [3, 4, 5][i % 3])
            hs = int.from_bytes(a0[2:4], "big") + 4
            a = a0[:hs + 18 * a0[7]] + a[hs + 18 * a0[7]:]
        items.append(a)
    outs, status, _ = run_job(Job.adx_decode(items))
    assert not status.any()
    for i, (o, a) in enumerate(zip(outs, items)):
        assert bytes(o) == O.adx_decode(a), i


# ------------------------------------------------------------------------------------------------ long streams
def test_ten_second_streams(cc):
    """>= 10 s per stream (469 HCA frames, 15 000 ADX block rows): single-file calls and a batch, whole-file byte equality"""
    from pycricodecs_amd.batch import Job
    n = 48000 * 10 + 352
    w = synth.wav(91, n, 2, 48000)
    hca = O.hca_crypt(O.hca_encode(w, 1), 1, 56, KEY)
    assert int.from_bytes(hca[16:20], "big") >= 469
    ref = O.hca_decode(hca, KEY)
    assert cc.HcaDecode(hca, 96, KEY, 0) == ref
    assert cc.HcaEncode(w, 0, 1) == O.hca_encode(w, 1)
    adx = O.adx_encode(w)
    assert cc.AdxEncode(w, 4, 18, 3, 500, 0, 4, False) == adx
    assert cc.AdxDecode(adx) == O.adx_decode(adx)
    hq = [O.hca_encode(w, q) for q in (2, 3)]
    outs, status, _ = run_job(Job.hca_decode([hca] + hq, keys=[KEY, 0, 0]))
    assert not status.any() and bytes(outs[0]) == ref and [bytes(o) for o in outs[1:]] == [O.hca_decode(h) for h in hq]
    for mapping_items in (3, 40):                              # wave-per-file and (padded with short clips) still correct
        items = [adx] + [O.adx_encode(synth.wav(92 + i, 640, 2, 48000)) for i in range(mapping_items - 1)]
        outs, status, _ = run_job(Job.adx_decode(items))
        assert not status.any() and bytes(outs[0]) == O.adx_decode(adx)


# ------------------------------------------------------------------------------------------------ shards
@pytest.mark.parametrize("world", [2, 3, 8])
def test_shard_jobs_equal_the_unsharded_batch(cc, world):
    """pycricodecs_amd.shard (what bench.py --gpus N and a multi-GPU caller use): every rank's shard, decoded by its own
    Job, byte-equals the items of the unsharded batch -- the results do not depend on the world size."""
    from pycricodecs_amd import shard
    from pycricodecs_amd.batch import Job
    items, keys = [], []
    for i in range(41):
        w = synth.wav(120 + i, 1024 * (1 + (i * 5) % 9) + 32 * (i % 4), 2, 48000)
        items.append(O.hca_crypt(O.hca_encode(w, 1 + i % 3), 1, 56, KEY))
        keys.append(KEY)
    whole, status, _ = run_job(Job.hca_decode(items, keys=keys))
    assert not status.any()
    weights = [shard.hca_weight(h) for h in items]
    seen = set()
    for r in range(world):
        mine = shard.my_items(weights, r, world)
        seen.update(mine)
        if not mine:
            continue
        outs, status, _ = run_job(Job.hca_decode([items[i] for i in mine], keys=[keys[i] for i in mine]))
        assert not status.any()
        for o, i in zip(outs, mine):
            assert bytes(o) == bytes(whole[i]), (r, i)
    assert seen == set(range(len(items)))


# ------------------------------------------------------------------------------------------------ threads / devices
def test_two_threads_through_the_extension_module(cc):
    """The CPython module releases the GIL around the library calls: two threads decode / encode concurrently and both get
    the oracle's bytes (the library keeps no mutable global state besides the one-time device probe)."""
    import importlib.util
    import os
    import sysconfig
    from pycricodecs_amd import build
    path = os.path.join(build.LIBDIR, "CriCodecs" + sysconfig.get_config_var("EXT_SUFFIX"))
    spec = importlib.util.spec_from_file_location("CriCodecs", path)
    ext = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ext)
    wavs = [synth.wav(200 + i, 9000 + 1000 * i, 2, 48000) for i in range(4)]
    hcas = [O.hca_crypt(O.hca_encode(w, 1), 1, 56, KEY) for w in wavs]
    exp_dec = [O.hca_decode(h, KEY) for h in hcas]
    exp_adx = [O.adx_encode(w) for w in wavs]
    exp_hca = [O.hca_encode(w, 2) for w in wavs]
    errors = []

    def worker(tid):
        try:
            for rep in range(6):
                for i in range(len(wavs)):
                    k = (i + tid) % len(wavs)
                    if ext.HcaDecode(hcas[k], 96, KEY, 0) != exp_dec[k]:
                        errors.append(("dec", tid, rep, k))
                    if ext.AdxEncode(wavs[k], 4, 18, 3, 500, 0, 4, False) != exp_adx[k]:
                        errors.append(("adx", tid, rep, k))
                    if ext.HcaEncode(wavs[k], 0, 2) != exp_hca[k]:
                        errors.append(("enc", tid, rep, k))
        except Exception as e:  # noqa: BLE001
            errors.append((tid, repr(e)))

    ts = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:4]


def test_device_selection_entry_points(cc):
    from pycricodecs_amd import _capi
    from pycricodecs_amd.batch import Job
    L = _capi.lib()
    n = L.cri_device_count()
    assert n >= 1 and L.cri_get_device() == 0
    assert L.cri_set_device(0) == 0
    assert L.cri_set_device(n) == -301 and L.cri_set_device(-1) == -301
    job = Job.hca_decode([G.load("s0_3008_2_48000_q1.hca")])
    assert L.cri_job_device(job._h) == 0
    # a job created in one thread runs (host wrapper) from another
    res = {}

    def other():
        outs, st = job.run_host()
        res["out"] = bytes(outs[0]); res["st"] = int(st[0])
    t = threading.Thread(target=other)
    t.start(); t.join()
    assert res["st"] == 0 and res["out"] == G.load("s0_3008_2_48000_q1.decoded.wav")


# ------------------------------------------------------------------------------------------------ planner regressions
def forge_adx_bitdepth(bs, bd, channels, rows, seed):
    """An ADX file the encoder cannot write (bitdepth 1, or any blocksize / bitdepth pair): header by hand, random blocks."""
    rng = np.random.default_rng(seed)
    spb = (bs - 2) * 8 // bd
    n = rows * spb
    base = 20 + 4 + 4 * max(channels, 2)
    hs = base + 6
    hs += -hs % 4
    head = bytearray(hs)
    head[0:2] = b"\x80\x00"
    head[2:4] = struct.pack(">H", hs - 4)
    head[4], head[5], head[6], head[7] = 3, bs, bd, channels
    head[8:12] = struct.pack(">I", 32000)
    head[12:16] = struct.pack(">I", n)
    head[16:18] = struct.pack(">H", 500)
    head[18], head[19] = 4, 0
    head[hs - 6:hs] = b"(c)CRI"
    blocks = bytearray()
    for k in range(rows * channels):                           # (the byte after "(c)CRI" -- the first scale's high byte -- must be 0, adx.cpp:345-348)
        blocks += struct.pack(">H", int(rng.integers(0, 0x100 if k == 0 else 0x400))) + rng.integers(0, 256, bs - 2, dtype=np.uint8).tobytes()
    return bytes(head) + bytes(blocks) + b"\x80\x01" + struct.pack(">H", bs - 4) + bytes(bs - 4)


def test_adx_bitdepth_1_big_blocks_do_not_fail_the_batch(cc):
    """one item with bitdepth 1 and blocksize 255 (2024 samples per block, 4.3 KB of LDS per chain and row) sized the
    whole launch past the 160 KB of LDS and failed every item.  The planner now sizes LDS per wave: such items decode, next
    to ordinary ones, and an item that cannot fit a wave by itself is the only one refused."""
    from pycricodecs_amd.batch import Job
    big2 = forge_adx_bitdepth(255, 1, 2, 3, 1)
    big24 = forge_adx_bitdepth(255, 1, 24, 2, 2)               # 24 channels x 4.3 KB: most of a wave's LDS
    mid = forge_adx_bitdepth(160, 1, 8, 2, 3)
    too_big = forge_adx_bitdepth(255, 1, 40, 1, 4)             # 40 x 4.3 KB > 150 KB: refused, alone
    normal = [O.adx_encode(synth.wav(300 + i, 3200, 2, 48000)) for i in range(6)]
    items = normal[:3] + [big2, big24] + normal[3:] + [mid, too_big]
    for it in (big2, big24, mid):
        assert cc.AdxDecode(it) == O.adx_decode(it)
    job = Job.adx_decode(items)
    outs, status, _ = run_job(job)
    assert job.host_status[-1] == -304 and not job.host_status[:-1].any() and not status.any()
    for o, a in zip(outs[:-1], items[:-1]):
        assert bytes(o) == O.adx_decode(a)


def test_sfa_pack_adx_shorter_than_one_chunk(cc):
    """usm.py:598 sizes the chunk before the last with Python's floor modulo; for a stream shorter than one chunk
    the operand is negative.  Chunk sizes against a direct statement of usm.py:584-640."""
    from pycricodecs_amd import usm

    def model_sizes(adx):
        rate, ch, bs = int.from_bytes(adx[8:12], "big"), adx[7], adx[5]
        first = int.from_bytes(adx[2:4], "big") + 4
        chunk = int(rate // 29.97 // 32) * (bs * ch)
        stream_size = len(adx) - bs
        tell, sizes = 0, []
        while tell < stream_size:
            if tell == 0:
                do = first
            else:
                do = (stream_size - first - chunk) % chunk if tell + chunk > stream_size else chunk
            do = min(do, len(adx) - tell)
            if do == 0:
                break                                          # (the reference would spin here: read(0) never advances)
            sizes.append(do)
            tell += do
        sizes.append(min(bs, len(adx) - tell))
        return sizes

    for n, ch, sr in ((320, 2, 48000), (640, 1, 48000), (960, 2, 48000), (1600, 2, 44100), (3200, 1, 22050), (4800, 2, 48000)):
        adx = O.adx_encode(synth.wav(400 + n, n, ch, sr))
        (chunks,) = usm.sfa_chunks([adx], "adx")
        sizes = [int.from_bytes(c[4:8], "big") - 0x18 - int.from_bytes(c[10:12], "big") for c in chunks]
        assert sizes == model_sizes(adx), (n, ch, sr)
        assert chunks[-1].endswith(b"#CONTENTS END   ===============\x00")


@pytest.mark.parametrize("mapping", ["chain", "file"])
def test_adx_decode_many_lengths_both_mappings(cc, mapping, knobs):
    """The lane-per-chain planner lays the files out by length (a wave lasts as long as its longest chain) and the wave-per-file
    kernels take them longest first; outputs stay in item order.  A few hundred clips of shuffled lengths, mono and stereo, cross
    wave boundaries in both."""
    from pycricodecs_amd.batch import Job
    knobs(adx_mapping=mapping)
    rng = np.random.default_rng(31)
    uniq = [O.adx_encode(synth.wav(700 + k, 32 * int(rng.integers(1, 60)), 1 + k % 2, 48000)) for k in range(24)]
    pick = rng.integers(0, len(uniq), 300)
    items = [uniq[k] for k in pick]
    job = Job.adx_decode(items)
    assert job.dominant_kernel == ("k_adx_decode_wpf" if mapping == "file" else "k_adx_decode")
    outs, st = job.run_host()
    assert not st.any()
    refs = [O.adx_decode(u) for u in uniq]
    for i, k in enumerate(pick):
        assert bytes(outs[i]) == refs[k], i
