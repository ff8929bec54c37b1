"""GPU parity, through the C ABI, against the pinned CPU oracle and the committed golden vectors.
ADX: block decode / encode kernels -- chain, wave-per-file, segmented and lane mappings -- against the oracle and the golden vectors.  Bit-exact."""
import ctypes as C
import struct

import numpy as np
import pytest

import golden_util as G
import oracle_lib as O
from gpu_common import KEY, MAN, cc, diff, run_job, run_job_floats  # noqa: F401
from pycricodecs_amd import synth

pytestmark = pytest.mark.gpu


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


# ------------------------------------------------------------------------------------------------ a3 / a4: segmented ADX decode
def _adx_files():
    rng = np.random.default_rng(123)
    files = []
    for k, (n, ch, sr, mode, hp) in enumerate([(32 * 400, 2, 48000, 3, 500), (32 * 1000 + 17, 1, 48000, 3, 500), (32 * 700, 2, 44100, 3, 500),
                                               (32 * 900, 2, 48000, 2, 500), (32 * 333, 4, 48000, 3, 2000), (32 * 1500, 2, 48000, 3, 100),
                                               (32 * 64, 2, 22050, 3, 500), (32 * 5, 1, 48000, 3, 500), (32 * 2100, 2, 48000, 3, 500)]):
        w = synth.wav(1200 + k, n, ch, sr)
        files.append(O.adx_encode(w, 4, 18, mode, hp, 0, 4))
    loud = (rng.integers(-32768, 32768, (32 * 600, 2))).astype(np.int16)          # full-scale noise: the clamp is hit all the time
    loud[:64] = 0                                                                 # (a first scale word >= 0x100 is rejected, adx.cpp:345-348)
    files.append(O.adx_encode(synth.wav_bytes(loud, 48000)))
    return files


# ------------------------------------------------------------------------------------------------ a5 / a6: segmented ADX encode
def _enc_wavs():
    rng = np.random.default_rng(321)
    wavs = [synth.wav(1500, 32 * 900, 2, 48000), synth.wav(1501, 32 * 1300 + 7, 1, 48000), synth.wav(1502, 32 * 640, 2, 44100),
            synth.wav(1503, 32 * 50, 2, 48000), synth.wav(1504, 31, 1, 48000)]
    quiet = synth.pcm16(1505, 32 * 800, 2, 48000)
    quiet[32 * 200:32 * 330] = 0                                # digital silence in the middle: silent blocks keep the RAW history (adx.cpp:231-234)
    quiet[32 * 500:32 * 501] = 0
    wavs.append(synth.wav_bytes(quiet, 48000))
    loud = rng.integers(-32768, 32768, (32 * 700, 2)).astype(np.int16)          # full-scale noise: clamps everywhere
    loud[:64] = 0
    wavs.append(synth.wav_bytes(loud, 48000))
    return wavs


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


# ------------------------------------------------------------------------------------------------ golden
@pytest.mark.parametrize("case", MAN["cases"], ids=lambda c: c["wav"])
def test_golden_adx(cc, case):
    w = G.load(case["wav"])
    for a in case["adx"]:
        ref = G.load(a["file"])
        bd, bs, mode, hp, filt, ver = a["params"]
        assert diff(cc.AdxEncode(w, bd, bs, mode, hp, filt, ver, False), ref) is None, a["file"]
        assert G.sha(cc.AdxDecode(ref)) == a["decoded_sha"], a["file"]


# ------------------------------------------------------------------------------------------------ vs oracle, seeded
@pytest.mark.parametrize("seed,n,ch,sr", [(0, 4800, 2, 48000), (1, 9600, 1, 44100), (3, 32, 2, 48000), (4, 48000, 2, 48000),
                                            (5, 2048, 1, 8000), (6, 5000, 4, 48000), (7, 999, 2, 48000)])
@pytest.mark.parametrize("bd,bs,mode,ver", [(4, 18, 3, 4), (4, 18, 4, 4), (4, 18, 2, 3), (8, 18, 3, 5), (2, 18, 3, 4), (6, 26, 3, 4),
                                             (12, 26, 4, 4)])
def test_adx_vs_oracle(cc, seed, n, ch, sr, bd, bs, mode, ver):
    w = synth.wav(seed, n, ch, sr)
    ref = O.adx_encode(w, bd, bs, mode, 500, 0, ver)
    got = cc.AdxEncode(w, bd, bs, mode, 500, 0, ver, False)
    assert diff(got, ref) is None
    try:
        want = O.adx_decode(ref)
    except O.OracleError as e:             # e.g. zero-frame files: the 7-byte "(c)CRI" check hits the 80 01 trailer
        with pytest.raises(ValueError):
            cc.AdxDecode(ref)
        assert e.code == -9
        return
    assert diff(cc.AdxDecode(ref), want) is None


@pytest.mark.parametrize("mapping", ["chain", "file"])
def test_adx_both_mappings(cc, mapping, knobs):
    """Standard-layout files through the lane-per-chain kernels and through the wave-per-file kernels."""
    from pycricodecs_amd.batch import Job
    knobs(adx_mapping=mapping)
    z = np.zeros((1600, 2), dtype=np.int16)
    z[500:700, 0] = 20000
    wavs = [synth.wav(400 + i, 32 * (3 + 7 * i), 1 + (i % 2), 48000) for i in range(9)] + [synth.wav_bytes(z, 44100)]
    for mode, ver in ((3, 4), (4, 4), (2, 3)):
        enc = Job.adx_encode(wavs, mode=mode, version=ver)
        assert enc.dominant_kernel == ("k_adx_encode_wpf" if mapping == "file" else "k_adx_encode")
        adx, st = enc.run_host()
        assert not st.any()
        for a, w in zip(adx, wavs):
            assert diff(a, O.adx_encode(w, 4, 18, mode, 500, 0, ver)) is None
        adx[3] = adx[3][:len(adx[3]) // 2]                    # truncated input
        dec = Job.adx_decode(adx)
        assert dec.dominant_kernel == ("k_adx_decode_wpf" if mapping == "file" else "k_adx_decode")
        pcm, st = dec.run_host()
        assert not st.any()
        for p, a in zip(pcm, adx):
            assert diff(p, O.adx_decode(a)) is None


def test_adx_silence_clipping_truncation(cc):
    z = np.zeros((3200, 2), dtype=np.int16)
    z[1000:1100] = 32767
    z[1100:1200] = -32768
    z[2000:2032, 0] = np.arange(32) * 1000
    w = synth.wav_bytes(z, 48000)
    for mode in (2, 3, 4):
        ref = O.adx_encode(w, 4, 18, mode)
        assert diff(cc.AdxEncode(w, 4, 18, mode, 500, 0, 4, False), ref) is None
        assert diff(cc.AdxDecode(ref), O.adx_decode(ref)) is None
        cut = ref[:len(ref) // 2]                                  # truncated input: remaining rows decode to silence
        assert diff(cc.AdxDecode(cut), O.adx_decode(cut)) is None


def test_batch_adx_roundtrip(cc):
    from pycricodecs_amd.batch import Job
    wavs = [synth.wav(200 + i, 320 * (i + 1), 1 + (i % 2), 48000) for i in range(9)]
    enc = Job.adx_encode(wavs)
    adx, st = enc.run_host()
    assert not st.any()
    for a, w in zip(adx, wavs):
        assert diff(a, O.adx_encode(w)) is None
    dec = Job.adx_decode(adx)
    pcm, st = dec.run_host()
    assert not st.any()
    for p, a in zip(pcm, adx):
        assert diff(p, O.adx_decode(a)) is None


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
        outs, status, _ = run_job_floats(Job.adx_encode(wavs, bitdepth=bd, blocksize=bs, mode=2, filt=filt))
        assert not status.any() and [bytes(o) for o in outs] == refs
        # the decoder's per-block predictor select (adx.cpp:196-203) on files that pass the header check: blocks with filter
        # bits set anywhere but in the very first scale word
        if filt:
            spliced = []
            for w, r in zip(wavs, refs):
                r0 = O.adx_encode(w, bd, bs, 2, 500, 0, 4)
                hs = int.from_bytes(r0[2:4], "big") + 4
                spliced.append(r0[:hs + bs * r0[7]] + r[hs + bs * r0[7]:])     # first block row from the filter-0 file
            outs, status, _ = run_job_floats(Job.adx_decode(spliced))
            assert not status.any() and [bytes(o) for o in outs] == [O.adx_decode(r) for r in spliced]
        else:
            outs, status, _ = run_job_floats(Job.adx_decode(refs))
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
    outs, status, _ = run_job_floats(Job.adx_decode(items))
    assert not status.any()
    for i, (o, a) in enumerate(zip(outs, items)):
        assert bytes(o) == O.adx_decode(a), i


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
    outs, status, _ = run_job_floats(job)
    assert job.host_status[-1] == -304 and not job.host_status[:-1].any() and not status.any()
    for o, a in zip(outs[:-1], items[:-1]):
        assert bytes(o) == O.adx_decode(a)


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


@pytest.mark.parametrize("warm", ["100", "10", "1"])
def test_adx_segmented_decode_vs_oracle(cc, knobs, warm):
    """k_adx_seg_decode / _fix / _serial: files cut into segments decoded speculatively from a warm-up, verified and repaired.
    With the default warm-up nearly every speculation is right; at 10 % and 1 % of it most are wrong and the repair passes do
    the work -- the bytes are the oracle's either way (modes 2 and 3, 1 / 2 / 4 channels, several coefficient sets, a sample
    count that is not a whole row, full-scale noise)."""
    from pycricodecs_amd.batch import Job
    knobs(adx_mapping="seg")
    knobs(adx_warm_pct=int(warm))
    files = _adx_files()
    job = Job.adx_decode(files)
    assert job.dominant_kernel == "k_adx_seg_decode"
    refs = [O.adx_decode(f) for f in files]
    outs, st = run_job(job)
    assert not st.any() and not job.host_status.any()
    for i, (o, r) in enumerate(zip(outs, refs)):
        assert bytes(o) == r, i
    outs, st = job.run_host()                                   # and through the host path (scratch from the arena)
    for i, (o, r) in enumerate(zip(outs, refs)):
        assert bytes(o) == r, i


@pytest.mark.parametrize("warm", ["100", "2"])
def test_adx_segmented_decode_end_markers_and_truncation(cc, knobs, warm):
    """adx.cpp:405-406 inside a segmented file: an end-of-stream scale word on a row's first block (in the first segment, in a
    later one, right at a segment's first row), inputs cut in the middle of a row, a header that announces more blocks than the
    file holds, a sample count below a whole row -- everything after the end decodes to silence, in every later segment."""
    from pycricodecs_amd.batch import Job
    knobs(adx_mapping="seg")
    knobs(adx_warm_pct=int(warm))
    base = O.adx_encode(synth.wav(1300, 32 * 1200, 2, 48000))
    mono = O.adx_encode(synth.wav(1301, 32 * 800, 1, 48000))
    do = int.from_bytes(base[2:4], "big") + 4
    files = []
    for row in (0, 1, 37, 140, 599, 1199):
        b = bytearray(base)
        b[do + row * 36:do + row * 36 + 2] = b"\x80\x01"
        files.append(bytes(b))
    b = bytearray(base); b[do + 500 * 36 + 18:do + 500 * 36 + 20] = b"\x80\x01"      # on the SECOND channel's block: not an end marker
    try:
        O.adx_decode(bytes(b)); files.append(bytes(b))
    except O.OracleError:
        pass
    for cut in (do + 36 * 700 + 5, do + 36 * 3, do + 36 * 1199 + 35, do + 1):
        files.append(base[:cut])
    dm = int.from_bytes(mono[2:4], "big") + 4
    files.append(mono[:dm + 18 * 411 + 9])
    b = bytearray(mono); b[12:16] = (32 * 800 - 13).to_bytes(4, "big"); files.append(bytes(b))   # sample count inside the last row
    b = bytearray(mono); b[12:16] = (32 * 500 + 1).to_bytes(4, "big"); files.append(bytes(b))
    job = Job.adx_decode(files)
    assert job.dominant_kernel == "k_adx_seg_decode"
    outs, st = run_job(job)
    for i, (o, f) in enumerate(zip(outs, files)):
        try:
            want = O.adx_decode(f)
        except O.OracleError:
            want = None
        if want is None:
            assert job.host_status[i] != 0 or st[i] != 0, i
        else:
            assert bytes(o) == want, i


def test_adx_ten_second_file_takes_the_segmented_path(cc):
    """The drop-in single-file call on a 10 s stereo file (one or two chains of 480 000 dependent steps for the unsegmented
    kernels) runs as a few hundred segments by default; high-pass 0 (no decay: coefficients 8192, -4096) stays one segment."""
    from pycricodecs_amd.batch import Job
    w = synth.wav(1400, 480000, 2, 48000)
    a = O.adx_encode(w)
    assert Job.adx_decode([a]).dominant_kernel == "k_adx_seg_decode"
    assert cc.AdxDecode(a) == O.adx_decode(a)
    a0 = O.adx_encode(w, 4, 18, 3, 0, 0, 4)
    assert Job.adx_decode([a0]).dominant_kernel != "k_adx_seg_decode"
    assert cc.AdxDecode(a0) == O.adx_decode(a0)


@pytest.mark.parametrize("mode,hp", [(3, 500), (4, 500), (2, 500), (3, 2000)])
@pytest.mark.parametrize("warm", ["100", "5", "1"])
def test_adx_segmented_encode_vs_oracle(cc, knobs, warm, mode, hp):
    """k_adx_seg_encode: files cut into segments, each encoded by a wave of its own from a warm-up, verified against the previous
    segment's end state and repaired (passes 0 / 1 / 2).  At 5 % and 1 % of the default warm-up nearly every speculation is wrong
    and the repair passes write most of the bytes -- which are the oracle's either way."""
    from pycricodecs_amd.batch import Job
    knobs(adx_mapping="seg")
    knobs(adx_warm_pct=int(warm))
    wavs = _enc_wavs()
    job = Job.adx_encode(wavs, mode=mode, highpass=hp)
    assert job.dominant_kernel == "k_adx_seg_encode"
    outs, st = run_job(job)
    assert not st.any() and not job.host_status.any()
    for i, (o, w) in enumerate(zip(outs, wavs)):
        assert bytes(o) == O.adx_encode(w, 4, 18, mode, hp, 0, 4), i
    outs, st = job.run_host()
    for i, (o, w) in enumerate(zip(outs, wavs)):
        assert bytes(o) == O.adx_encode(w, 4, 18, mode, hp, 0, 4), i


def test_adx_ten_second_file_encodes_in_segments(cc):
    """The drop-in AdxEncode on a 10 s stereo file runs as a dozen segments by default (two chains of 480 000 dependent steps
    otherwise); typed (24-bit) input goes through the conversion scratch first; high-pass 0 stays one segment."""
    from pycricodecs_amd.batch import Job
    w = synth.wav(1600, 480000, 2, 48000)
    assert Job.adx_encode([w]).dominant_kernel == "k_adx_seg_encode"
    assert cc.AdxEncode(w, 4, 18, 3, 500, 0, 4, False) == O.adx_encode(w)
    w24 = synth.wav_typed(1601, 32 * 5000, 2, 48000, "s24")
    assert Job.adx_encode([w24]).dominant_kernel == "k_adx_seg_encode"
    assert cc.AdxEncode(w24, 4, 18, 3, 500, 0, 4, False) == O.adx_encode(w24)
    assert Job.adx_encode([w], highpass=0).dominant_kernel != "k_adx_seg_encode"
    assert cc.AdxEncode(w, 4, 18, 3, 0, 0, 4, False) == O.adx_encode(w, 4, 18, 3, 0, 0, 4)


@pytest.mark.parametrize("mode,hp", [(3, 500), (4, 500), (2, 500)])
@pytest.mark.parametrize("pct", ["100", "20", "3"])
def test_adx_lane_encode_vs_oracle(cc, knobs, pct, mode, hp):
    """k_adx_lane_encode / _serial: a lane per (file, channel, segment), every segment encoded from a guessed history and again from
    the previous segment's end until the two histories merge at a checkpoint.  With the default minimum segment length the files of
    this test are one to three segments; at 20 % and 3 % of it they are dozens of segments too short to merge in, so the files are
    flagged and the serial pass rewrites them -- the bytes are the oracle's either way."""
    from pycricodecs_amd.batch import Job
    knobs(adx_mapping="lane")
    knobs(adx_warm_pct=int(pct))
    wavs = _enc_wavs() + [synth.wav(1700, 32 * 2600, 2, 48000), synth.wav(1701, 32 * 2100 + 5, 1, 48000)]
    job = Job.adx_encode(wavs, mode=mode, highpass=hp)
    assert job.dominant_kernel == "k_adx_lane_encode"
    refs = [O.adx_encode(w, 4, 18, mode, hp, 0, 4) for w in wavs]
    outs, st = run_job(job)
    assert not st.any() and not job.host_status.any()
    for i, (o, r) in enumerate(zip(outs, refs)):
        assert bytes(o) == r, i
    outs, st = job.run_host()
    for i, (o, r) in enumerate(zip(outs, refs)):
        assert bytes(o) == r, i


def test_adx_lane_encode_many_files(cc, knobs):
    """A few hundred clips of shuffled lengths, mono and stereo, 24-bit input among them, in the lane mapping (the planner's own choice
    from about 8 M blocks on), all against the oracle."""
    from pycricodecs_amd.batch import Job
    knobs(adx_mapping="lane")
    rng = np.random.default_rng(99)
    uniq = [synth.wav(1800 + k, 32 * int(rng.integers(1, 1500)) + int(rng.integers(0, 32)), 1 + k % 2, 48000) for k in range(20)]
    uniq.append(synth.wav_typed(1830, 32 * 700, 2, 48000, "s24"))
    pick = rng.integers(0, len(uniq), 260)
    job = Job.adx_encode([uniq[k] for k in pick])
    assert job.dominant_kernel == "k_adx_lane_encode"
    outs, st = run_job(job)
    assert not st.any()
    refs = [O.adx_encode(u) for u in uniq]
    for i, k in enumerate(pick):
        assert bytes(outs[i]) == refs[k], i


def test_adx_segmented_decode_through_silence_and_pure_tones(cc):
    """Where histories do not merge: digital silence after a sound (the decoder's state sits at a fixed point of the recurrence, a
    different one for a different history) and noiseless periodic material (limit cycles).  Those chains are flagged and their files
    decoded again by the wave-per-file kernel; everything else in the job keeps its segments."""
    from pycricodecs_amd.batch import Job
    t = np.arange(32 * 4000)[:, None] / 48000.0
    tone = np.round(0.9 * 32767 * np.sin(2 * np.pi * 997.0 * t + np.array([[0.0, 0.7]]))).astype(np.int16)
    tone[:512] = (tone[:512] * (np.arange(512)[:, None] / 512.0) ** 2).astype(np.int16)
    gap = synth.pcm16(1900, 32 * 4000, 2, 48000)
    gap[32 * 700:32 * 2900] = 0                                 # 1.5 s of digital silence inside
    gap[32 * 3300:] = 0                                         # and at the end
    files = [O.adx_encode(synth.wav_bytes(tone, 48000)), O.adx_encode(synth.wav_bytes(gap, 48000)), O.adx_encode(synth.wav(1901, 32 * 4000, 2, 48000)),
             O.adx_encode(synth.wav_bytes(gap[:, :1].copy(), 48000))]
    job = Job.adx_decode(files)
    assert job.dominant_kernel == "k_adx_seg_decode"
    outs, st = run_job(job)
    assert not st.any()
    for i, (o, f) in enumerate(zip(outs, files)):
        assert bytes(o) == O.adx_decode(f), i
    enc = Job.adx_encode([synth.wav_bytes(tone, 48000), synth.wav_bytes(gap, 48000)])
    outs, st = run_job(enc)
    assert bytes(outs[0]) == files[0] and bytes(outs[1]) == files[1]


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


# ------------------------------------------------------------------------------------------------ a3: long silent chains
@pytest.mark.parametrize("warm", [100, 1])
def test_adx_segmented_decode_of_long_silent_chains(cc, knobs, warm):
    """A long file that is mostly digital silence: hundreds of silent segments in a row.  k_adx_seg_runs writes where each silent run
    began into the segments' records (a wave per chain, 64 segments a step), so a lane behind the run finds the last segment with
    sound in one load instead of walking the run back -- n^2 dependent loads per repair round before.  Segments forced
    short so that the chains have far more than 64 segments; silent heads, tails and a file of nothing but silence; bytes = oracle."""
    from pycricodecs_amd.batch import Job
    knobs(adx_mapping="seg", adx_warm_pct=warm, adx_seglen=1)
    files = []
    for k, (secs, ch, hp) in enumerate([(40, 2, 500), (30, 1, 500), (20, 2, 4000), (25, 2, 500)]):
        n = 48000 * secs
        x = np.zeros((n, ch), np.int32)
        if k != 3:
            burst = synth.pcm16(7000 + k, 9000, ch, 48000).astype(np.int32).reshape(-1, ch)
            for at in ((n // 7, n // 2, n - 20000) if k != 1 else (0, n // 3)):
                x[at:at + len(burst)] = burst[:max(0, min(len(burst), n - at))]
        files.append(O.adx_encode(synth.wav_bytes(x.astype(np.int16), 48000), 4, 18, 3, hp, 0, 4))
    job = Job.adx_decode(files)
    assert job.dominant_kernel == "k_adx_seg_decode"
    outs, st = run_job(job)
    assert not st.any() and not job.host_status.any()
    for i, (o, f) in enumerate(zip(outs, files)):
        assert bytes(o) == O.adx_decode(f), i
