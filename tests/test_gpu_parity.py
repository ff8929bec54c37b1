"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the committed golden vectors.
Bit-exact for everything (ADX bytes/PCM, HCA PCM16, crypt bytes)."""
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


def diff(a, b):
    if a == b:
        return None
    n = min(len(a), len(b))
    idx = [i for i in range(n) if a[i] != b[i]][:8]
    return "len %d vs %d, first diffs at %s" % (len(a), len(b), idx)


# ------------------------------------------------------------------------------------------------ golden
@pytest.mark.parametrize("case", MAN["cases"], ids=lambda c: c["wav"])
def test_golden_adx(cc, case):
    w = G.load(case["wav"])
    for a in case["adx"]:
        ref = G.load(a["file"])
        bd, bs, mode, hp, filt, ver = a["params"]
        assert diff(cc.AdxEncode(w, bd, bs, mode, hp, filt, ver, False), ref) is None, a["file"]
        assert G.sha(cc.AdxDecode(ref)) == a["decoded_sha"], a["file"]


@pytest.mark.parametrize("case", MAN["cases"], ids=lambda c: c["wav"])
def test_golden_hca_decode_and_crypt(cc, case):
    for h in case["hca"]:
        ref = G.load(h["file"])
        hs = int.from_bytes(ref[6:8], "big")
        assert G.sha(cc.HcaDecode(ref, hs, 0, 0)) == h["decoded_sha"], h["file"]
        enc = cc.HcaCrypt(ref, 1, hs, 56, KEY, 0)
        assert G.sha(enc) == h["enc56_sha"]
        assert G.sha(cc.HcaDecode(enc, hs, KEY, 0)) == h["enc56_decoded_sha"]
        assert G.sha(cc.HcaCrypt(ref, 1, hs, 56, 0x1234567, 0x4321)) == h["enc56_sub_sha"]
        assert G.sha(cc.HcaCrypt(ref, 1, hs, 1, 0, 0)) == h["enc1_sha"]
        assert G.sha(cc.HcaCrypt(enc, 0, hs, 0, KEY, 0)) == h["dec_of_enc56_sha"]


def test_golden_stored_decodes(cc):
    h = G.load("s0_3008_2_48000_q1.hca")
    assert diff(cc.HcaDecode(h, 96, 0, 0), G.load("s0_3008_2_48000_q1.decoded.wav")) is None
    a = G.load("s0_3008_2_48000_bd4_bs18_m3_v4.adx")
    assert diff(cc.AdxDecode(a), G.load("s0_3008_2_48000_bd4_bs18_m3_v4.decoded.wav")) is None


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


@pytest.mark.parametrize("seed,n,ch,sr", [(0, 4800, 2, 48000), (1, 9600, 1, 44100), (2, 3008, 2, 22050), (3, 100, 2, 48000),
                                            (4, 30000, 2, 32000), (6, 4096, 4, 48000), (7, 2500, 6, 48000)])
@pytest.mark.parametrize("q", [0, 1, 2, 3])
def test_hca_decode_vs_oracle(cc, seed, n, ch, sr, q):
    w = synth.wav(seed, n, ch, sr)
    hca = O.hca_encode(w, q)
    hs = int.from_bytes(hca[6:8], "big")
    assert diff(cc.HcaDecode(hca, hs, 0, 0), O.hca_decode(hca)) is None
    enc = O.hca_crypt(hca, 1, 56, KEY)
    assert diff(cc.HcaCrypt(hca, 1, hs, 56, KEY, 0), enc) is None
    assert diff(cc.HcaDecode(enc, hs, KEY, 0), O.hca_decode(enc, KEY)) is None
    assert diff(cc.HcaCrypt(enc, 0, hs, 0, KEY, 0), O.hca_crypt(enc, 0, 0, KEY)) is None


def test_hca_decode_errors(cc):
    hca = G.load("s0_3008_2_48000_q1.hca")
    bad = bytearray(hca)
    bad[300] ^= 0x55
    with pytest.raises(ValueError, match="Decoding error"):
        cc.HcaDecode(bytes(bad), 96, 0, 0)
    enc = O.hca_crypt(hca, 1, 56, KEY)
    with pytest.raises(ValueError, match="Decoding error"):
        cc.HcaDecode(enc, 96, KEY + 2, 0)
    with pytest.raises(ValueError, match="not a valid HCA header"):
        cc.HcaDecode(b"HCA\x00" + bytes(200), 96, 0, 0)
    with pytest.raises(ValueError, match="copyright"):
        cc.AdxDecode(bytes([0x80, 0, 0, 0x2C, 3, 18, 4, 2, 0, 0, 0xBB, 0x80, 0, 0, 0, 64, 1, 0xF4, 4, 0]) + bytes(200))
    with pytest.raises(ValueError, match="Bitdepth"):
        cc.AdxEncode(synth.wav(0, 320, 2), 1, 18, 3, 500, 0, 4, False)
    adx = bytearray(O.adx_encode(synth.wav(0, 320, 2)))
    for bs in (1, 2):                                          # no samples per block (found by the long header fuzz: the oracle crashed on it)
        adx[5] = bs
        with pytest.raises(ValueError):
            cc.AdxDecode(bytes(adx))
        with pytest.raises(O.OracleError):
            O.adx_decode(bytes(adx))


@pytest.mark.parametrize("f", MAN["forged"], ids=lambda f: f["file"])
def test_forged_golden(cc, f):
    data = G.load(f["file"])
    assert G.sha(cc.HcaDecode(data, int.from_bytes(data[6:8], "big"), 0, 0)) == f["decoded_sha"]


def test_hca_v2_random_frame_fuzz(cc):
    """Random-byte frames: accepted/rejected exactly like the oracle, identical PCM when accepted
    (exercises escape codes, out-of-range deltas, reads past the frame end, saturating conversions)."""
    for q, ch in ((1, 2), (2, 2), (3, 2), (1, 1)):
        base = O.hca_encode(synth.wav(0, 800, ch, 48000), q)
        hs = int.from_bytes(base[6:8], "big")
        for seed in range(24):
            f = hca_forge.random_frames(base, seed, density=1.0 if seed % 2 else 0.35)
            try:
                ref = O.hca_decode(f)
            except O.OracleError:
                with pytest.raises(ValueError):
                    cc.HcaDecode(f, hs, 0, 0)
                continue
            assert diff(cc.HcaDecode(f, hs, 0, 0), ref) is None, (q, ch, seed)


@pytest.mark.parametrize("q,ch,n", [(1, 2, 9000), (2, 2, 30000), (3, 2, 5000), (1, 1, 12000), (2, 4, 6000), (1, 6, 4000), (1, 4, 6000),
                                    (3, 1, 3000), (4, 1, 3000), (4, 2, 1100), (3, 2, 2100), (0, 4, 5000)])
def test_hca_v3_noise_fill(cc, q, ch, n):
    """v3.0 / min_resolution 0: noise reconstruction (hca.cpp:1602-1635); the generator state runs across frames, so
    multi-frame streams check k_hca_noise_scan + the jump-ahead.  Also random frames under the v3.0 rules."""
    base = hca_forge.forge_v3(O.hca_encode(synth.wav(30 + q, n, ch, 48000), q), 0)
    hs = int.from_bytes(base[6:8], "big")
    # (with HFR groups the v2.0 frames are not valid v3.0 frames: then both sides must reject, and only random frames decode)
    accepted = 0
    for seed in range(-1, 30):
        f = base if seed < 0 else hca_forge.random_frames(base, seed, density=1.0 if seed % 2 else 0.35)
        try:
            ref = O.hca_decode(f)
        except O.OracleError:
            with pytest.raises(ValueError):
                cc.HcaDecode(f, hs, 0, 0)
            continue
        accepted += 1
        assert diff(cc.HcaDecode(f, hs, 0, 0), ref) is None, seed
    assert accepted > 0


@pytest.mark.parametrize("ch", [1, 2, 3, 4, 5, 6, 7, 8])
def test_hca_multichannel_layouts(cc, ch):
    """Streams longer than one run of 8 frames (the run's halo steps) for every channel count, then forged comp chunks:
    joint-stereo bands, HFR groups, several tracks (stereo pairs starting on odd channels: 2 tracks x 3 channels) and
    both channel configs (hca.cpp:887-970), v2.0 and v3.0 (noise reconstruction), with random frames under each layout."""
    w = synth.wav(60 + ch, 10500, ch, 48000)
    for q in (1, 3, 4):
        h = O.hca_encode(w, q)
        assert diff(cc.HcaDecode(h, int.from_bytes(h[6:8], "big"), 0, 0), O.hca_decode(h)) is None, q
    base = O.hca_encode(w, 1)
    hs = int.from_bytes(base[6:8], "big")
    total, bb = base[0x22], base[0x23]
    accepted = tried = 0

    def takes(stream):
        try:
            O.hca_decode(stream)
            return True
        except O.OracleError:
            return False
    for tracks in (1, 2, 3):
        if ch % tracks:
            continue
        for config in (0, 1):
            for (stereo, hfr) in ((0, 0), (8, 0), (bb - 4, 0), (8, 4), (0, 3)):
                nb = bb - stereo if hfr == 0 else bb - stereo - 16
                f0 = hca_forge.forge_comp(base, track_count=tracks, channel_config=config, total=bb if hfr == 0 else total,
                                          base=nb, stereo=stereo, hfr=hfr)
                for seed in range(2):
                    fb = hca_forge.forge_v3(f0, 0) if (seed + tracks + config) % 2 else f0       # half of them as v3.0 with noise reconstruction
                    f = hca_forge.accepted_random_stream(fb, 100 * ch + seed, 0.3 if seed else 0.08, takes)
                    tried += 1
                    if f is None or not takes(f):
                        with pytest.raises(ValueError):
                            cc.HcaDecode(f if f is not None else hca_forge.random_frames(fb, 100 * ch + seed, 0.3), hs, 0, 0)
                        continue
                    accepted += 1
                    assert diff(cc.HcaDecode(f, hs, 0, 0), O.hca_decode(f)) is None, (tracks, config, stereo, hfr, seed)
    assert accepted >= tried // 2, (accepted, tried)


@pytest.mark.parametrize("ch", [1, 2, 4, 6])
def test_hca_batch_random_lengths(cc, ch):
    """One batch of streams of every length from a fraction of a frame to a few runs of 8 frames (the transform splits a run
    between its transform slots by frame count), plain and HFR qualities mixed, against the oracle item by item."""
    from pycricodecs_amd.batch import Job
    rng = np.random.default_rng(40 + ch)
    items = []
    for i in range(36):
        n = int(rng.integers(1, 26000)) if i % 3 else int(rng.integers(1, 1400))
        items.append(O.hca_encode(synth.wav(200 + i, n, ch, [48000, 44100, 32000][i % 3]), quality=1 if i % 4 else 3))
    outs, st = Job.hca_decode(items).run_host()
    for i, (o, h, code) in enumerate(zip(outs, items, st)):
        assert code == 0, i
        assert diff(bytes(o), O.hca_decode(h)) is None, (i, len(h))


def test_hca_header_with_wrapped_hfr_group_count(cc):
    """Found by tools/debug/header_fuzz_multi.py: total_band_count below base + stereo with HFR groups wraps the unsigned
    group count; the reference (and the oracle, before) segfault on it -- both sides reject the header."""
    base = O.hca_encode(synth.wav(74, 9500, 4, 48000), quality=3)
    assert base[0x25] > 0                                       # bands per HFR group
    bad = hca_forge.forge_comp(base, total=base[0x23] + base[0x24] - 1)
    with pytest.raises(O.OracleError):
        O.hca_decode(bad)
    with pytest.raises(ValueError, match="not a valid HCA header"):
        cc.HcaDecode(bad, int.from_bytes(bad[6:8], "big"), 0, 0)


@pytest.mark.parametrize("ch,q", [(1, 1), (2, 1), (4, 1), (2, 2), (2, 3), (1, 3), (4, 3), (2, 4)])
def test_hca_int8_and_int16_records_mixed(cc, ch, q):
    """Mono, stereo and four-channel formats (plain, and with intensity stereo / high-frequency reconstruction: q >= 2) keep a frame's quantised lines as int8 when no band of its 64-frame tile can exceed 8 bits, as
    int16 otherwise.  Streams whose later tiles (or single frames, in a batch that shares tiles) carry random high-resolution
    frames put both record forms next to each other: in one run of 8 frames, in one transform step, in one tile."""
    from pycricodecs_amd.batch import Job
    base = O.hca_encode(synth.wav(91, 90000, ch, 48000), q)             # 88 frames: tiles 0 and 1
    hs, fs = int.from_bytes(base[6:8], "big"), int.from_bytes(base[28:30], "big")
    nfr = int.from_bytes(base[16:20], "big")

    def takes(stream):
        try:
            O.hca_decode(stream)
            return True
        except O.OracleError:
            return False
    rnd = hca_forge.accepted_random_stream(base, 4242, 1.0, takes)     # every frame random (full density: wide values), all accepted
    assert rnd is not None
    variants = []
    for lo, hi in ((64, nfr), (66, 70), (0, 3), (60, 68), (7, 9)):      # random frames lo..hi-1, the rest as encoded
        b = bytearray(base)
        b[hs + lo * fs:hs + hi * fs] = rnd[hs + lo * fs:hs + hi * fs]
        variants.append(bytes(b))
    for v in variants:
        assert diff(cc.HcaDecode(v, hs, 0, 0), O.hca_decode(v)) is None
    # one batch: short plain streams around them, so that tiles mix streams of both kinds
    short = [O.hca_encode(synth.wav(300 + i, 2000 + 700 * i, ch, 48000), q) for i in range(6)]
    items = [short[0], variants[1], short[1], short[2], variants[2], short[3], variants[0], short[4], variants[4], short[5]]
    outs, st = Job.hca_decode(items).run_host()
    for i, (o, h, code) in enumerate(zip(outs, items, st)):
        assert code == 0 and diff(bytes(o), O.hca_decode(h)) is None, i


def test_hca_secondary_channel_without_coded_bands(cc):
    """Found by the long header fuzz (CRI_FUZZ_ITERS=3000): base_band_count 0 with joint-stereo bands leaves the secondary
    channel without any coded band, i.e. without spectra blocks in the parse -- the block walk must skip it."""
    for q in (2, 3):
        base = O.hca_encode(synth.wav(77, 9000, 2, 48000), q)
        hs = int.from_bytes(base[6:8], "big")
        total, bb, sb = base[0x22], base[0x23], base[0x24]
        for (nb, ns) in ((0, bb + sb), (0, 16), (16, bb + sb - 16)):
            f0 = hca_forge.forge_comp(base, base=nb, stereo=ns)
            checked = 0
            for seed in range(6):
                f = f0 if seed == 0 else hca_forge.random_frames(f0, 900 + seed, density=0.3)
                try:
                    ref = O.hca_decode(f)
                except O.OracleError:
                    with pytest.raises(ValueError):
                        cc.HcaDecode(f, hs, 0, 0)
                    continue
                checked += 1
                assert diff(cc.HcaDecode(f, hs, 0, 0), ref) is None, (q, nb, ns, seed)
            assert checked >= 1, (q, nb, ns)


def test_v3_delta_intensity_keeps_stale_entries(cc):
    """Found by tools/debug/frame_fuzz.py: a v3.0 delta-coded intensity list that runs out of range leaves the remaining
    entries at the previous frame's values (the reference returns early and ignores the error, hca.cpp:1185, 1405-1408)."""
    base = hca_forge.forge_v3(O.hca_encode(synth.wav(3, 4000, 2, 48000), 4), 0)
    hs = int.from_bytes(base[6:8], "big")
    hit = 0
    for seed in (59, 11, 23, 35, 47, 71, 83, 95, 107, 119):
        f = hca_forge.random_frames(base, seed, density=1.0 if seed % 2 else 0.35)
        try:
            ref = O.hca_decode(f)
        except O.OracleError:
            with pytest.raises(ValueError):
                cc.HcaDecode(f, hs, 0, 0)
            continue
        hit += 1
        assert diff(cc.HcaDecode(f, hs, 0, 0), ref) is None, seed
    assert hit >= 1


def test_hca_v3_noise_batch(cc):
    from pycricodecs_amd.batch import Job
    items = []
    for i in range(7):
        h = O.hca_encode(synth.wav(50 + i, 3000 + 2100 * i, 1 + i % 2, 44100), 1 + i % 3)
        items.append(hca_forge.forge_v3(h, 0) if i % 3 != 2 else h)
    outs, st = Job.hca_decode(items).run_host()
    good = 0
    for o, h, code in zip(outs, items, st):
        try:
            ref = O.hca_decode(h)
        except O.OracleError:
            assert code != 0
            continue
        good += 1
        assert code == 0 and diff(bytes(o), ref) is None
    assert good >= 5


# ------------------------------------------------------------------------------------------------ HCA encode
@pytest.mark.parametrize("case", MAN["cases"], ids=lambda c: c["wav"])
def test_golden_hca_encode(cc, case):
    w = G.load(case["wav"])
    for h in case["hca"]:
        assert diff(cc.HcaEncode(w, 0, h["quality"]), G.load(h["file"])) is None, h["file"]


@pytest.mark.parametrize("seed,n,ch,sr", [(0, 4800, 2, 48000), (1, 9600, 1, 44100), (2, 3008, 2, 22050), (3, 100, 2, 48000),
                                            (4, 30000, 2, 32000), (5, 2048, 1, 48000), (6, 4096, 4, 48000), (7, 2500, 6, 48000),
                                            (8, 1024, 2, 48000), (9, 7000, 8, 48000), (10, 6000, 3, 48000)])
@pytest.mark.parametrize("q", [0, 1, 2, 3, 4, 5])
def test_hca_encode_vs_oracle(cc, seed, n, ch, sr, q):
    w = synth.wav(seed, n, ch, sr)
    assert diff(cc.HcaEncode(w, 0, q), O.hca_encode(w, q)) is None


def test_hca_encode_special_signals(cc):
    n = 4096
    silence = synth.wav_bytes(np.zeros((n, 2), dtype=np.int16), 48000)
    full = np.zeros((n, 2), dtype=np.int16)
    full[::2] = 32767
    full[1::2] = -32768
    loud = synth.wav_bytes(full, 48000)
    rng = np.random.default_rng(5)
    noise = synth.wav_bytes(rng.integers(-32768, 32767, (n, 2), dtype=np.int16), 48000)
    left_only = np.zeros((n, 2), dtype=np.int16)
    left_only[:, 0] = synth.pcm16(3, n, 1)[:, 0]
    for w in (silence, loud, noise, synth.wav_bytes(left_only, 48000)):
        for q in (1, 2, 3):
            assert diff(cc.HcaEncode(w, 0, q), O.hca_encode(w, q)) is None, q


@pytest.mark.parametrize("t", MAN["typed"], ids=lambda t: "%s_%dch" % (t["kind"], t["args"][2]))
def test_typed_wav_encode(cc, t):
    """Non-16-bit WAV input: k_pcm_convert + both encoders against the reference's digests and the oracle."""
    w = synth.wav_typed(*t["args"], t["kind"])
    assert G.sha(w) == t["wav_sha"]
    adx = cc.AdxEncode(w, 4, 18, 3, 500, 0, 4, False)
    assert G.sha(adx) == t["adx_sha"]
    hca = cc.HcaEncode(w, False, 1)
    assert G.sha(hca) == t["hca_q1_sha"]
    w2 = synth.wav_typed(21, 5000, t["args"][2], 48000, t["kind"])
    for args in [(4, 18, 3, 500, 0, 4), (8, 34, 4, 500, 0, 5), (4, 18, 2, 500, 0, 3)]:
        assert diff(cc.AdxEncode(w2, *args, False), O.adx_encode(w2, *args)) is None
    assert diff(cc.HcaEncode(w2, False, 3), O.hca_encode(w2, quality=3)) is None


def test_typed_wav_batch(cc):
    """A batch mixing 16-bit and converted inputs (scratch regions for some items only)."""
    from pycricodecs_amd import batch
    ws = [synth.wav(1, 2000, 2, 44100), synth.wav_typed(2, 1500, 2, 44100, "f32"), synth.wav_typed(3, 900, 1, 22050, "u8"),
          synth.wav(4, 100, 1, 48000), synth.wav_typed(5, 2500, 2, 48000, "s24")]
    for mk, ora in ((lambda: batch.Job.adx_encode(ws), lambda w: O.adx_encode(w)), (lambda: batch.Job.hca_encode(ws, quality=2), lambda w: O.hca_encode(w, quality=2))):
        outs, status = mk().run_host()
        assert list(status) == [0] * len(ws)
        for w, o in zip(ws, outs):
            assert diff(bytes(o), ora(w)) is None


@pytest.mark.parametrize("t", MAN["loops"], ids=lambda t: "s%d_%d_%d" % (t["args"][0], t["loop"][0], t["loop"][1]))
def test_loop_golden(cc, t):
    """Looping WAV input: ADX loop header, the HCA encoder's loop feeding sequence + loop chunk, smpl chunk out of both decoders."""
    seed, n, ch, sr = t["args"]
    w = synth.wav_bytes(synth.pcm16(seed, n, ch, sr), sr, loop=tuple(t["loop"]))
    assert G.sha(w) == t["wav_sha"]
    for ver, e in t["adx"].items():
        a = cc.AdxEncode(w, 4, 18, 3, 500, 0, int(ver), False)
        assert G.sha(a) == e["sha"] and G.sha(cc.AdxDecode(a)) == e["decoded_sha"]
    assert G.sha(cc.AdxEncode(w, 4, 18, 3, 500, 0, 5, True)) == t["adx_v5_noloop_sha"]
    for q, e in t["hca"].items():
        h = cc.HcaEncode(w, False, int(q))
        assert G.sha(h) == e["sha"]
        assert G.sha(cc.HcaDecode(h, int.from_bytes(h[6:8], "big"), 0, 0)) == e["decoded_sha"]
    assert G.sha(cc.HcaEncode(w, True, 1)) == t["hca_q1_noloop_sha"]


@pytest.mark.parametrize("seed,n,ch,sr", [(0, 9000, 2, 48000), (1, 5000, 1, 44100), (2, 20000, 2, 22050), (3, 3000, 4, 48000), (4, 40000, 2, 48000)])
def test_hca_loop_encode_vs_oracle(cc, seed, n, ch, sr):
    pcm = synth.pcm16(seed, n, ch, sr)
    for loop in [(0, n), (100, n - 1), (1024, 2048), (1000, 2000), (2047, 2049), (n - 300, n), (n // 2, n // 2 + 1), (1, 2), (0, 1024), (3000, 2900),
                 (n - 1, n), (0, 0), (5, n + 700)]:
        w = synth.wav_bytes(pcm, sr, loop=loop)
        for q in (0, 3):
            assert diff(cc.HcaEncode(w, False, q), O.hca_encode(w, quality=q)) is None, (loop, q)


def test_front_end_encode_roundtrip(cc):
    from pycricodecs_amd import HCA, CriHcaQuality
    w = synth.wav(11, 5000, 2, 48000)
    h = HCA(w, key=KEY)
    enc = h.encode(encrypt=True, quality_level=CriHcaQuality.Middle)
    ref = O.hca_crypt(O.hca_encode(w, 2), 1, 56, KEY)
    assert diff(enc, ref) is None
    assert h.encrypted and h.hca["CipherType"] == 0 and h.filetype == "wav"   # header re-parsed before encrypting, like the reference
    assert diff(HCA(enc, key=KEY).decode(), O.hca_decode(ref, KEY)) is None
    h2 = HCA(enc, key=KEY)
    h2.decrypt(KEY)
    assert diff(h2.get_hca(), O.hca_encode(w, 2)) is None


# ------------------------------------------------------------------------------------------------ batch jobs
def test_batch_mixed_formats(cc):
    from pycricodecs_amd.batch import Job
    items, keys, refs = [], [], []
    for i, (n, ch, sr, q) in enumerate([(3000, 2, 48000, 1), (5000, 1, 44100, 1), (2048, 2, 48000, 2), (7000, 2, 48000, 3),
                                        (1024, 2, 48000, 1), (300, 2, 22050, 0), (4000, 2, 48000, 1)]):
        h = O.hca_encode(synth.wav(100 + i, n, ch, sr), q)
        if i % 2:
            h = O.hca_crypt(h, 1, 56, KEY + i)
            keys.append(KEY + i)
        else:
            keys.append(0)
        items.append(h)
        refs.append(O.hca_decode(h, keys[-1]))
    items.insert(3, b"garbage" * 30)
    keys.insert(3, 0)
    refs.insert(3, b"")
    job = Job.hca_decode(items, keys=keys)
    outs, status = job.run_host()
    assert job.host_status[3] == -201 and status[3] == -201
    for i, (o, r) in enumerate(zip(outs, refs)):
        assert diff(o, r) is None, i
    assert job.units == sum(int.from_bytes(h[16:20], "big") for h in items if h[:3] == b"HCA" or h[:1] == b"\xc8")


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


def test_awb_front_door(cc, tmp_path):
    """AFS2 bank -> one HCA decode job + one ADX decode job over the same blob; digests of the reference's per-item decode."""
    from pycricodecs_amd import awb
    a = MAN["awb"]
    bank = G.load(a["file"])
    b = awb.AWB(bank)
    assert (b.numfiles, b.align, b.subkey, b.headersize, b.ofs) == (a["numfiles"], a["align"], a["subkey"], a["headersize"], a["ofs"])
    assert [G.sha(x) for x in b.getfiles()] == [i["sha"] for i in a["items"]]
    wavs = b.decode_all(KEY)
    assert [G.sha(w) for w in wavs] == [i["decoded_sha"] for i in a["items"]]
    with pytest.raises(ValueError):
        b.decode_all(KEY + 1)                                  # wrong key: the HCA items fail their frame checks
    # extract() writes the reference's file names
    p = tmp_path / "sfx.awb"
    p.write_bytes(bank)
    awb.AWB(str(p)).extract(decode=True, key=KEY)
    names = sorted(x.name for x in tmp_path.iterdir())
    assert names == sorted(["sfx.awb"] + ["sfx_%d.%s" % (k, "wav" if i["kind"] == "hca" else "dat") for k, i in enumerate(a["items"])])
    assert G.sha((tmp_path / "sfx_0.wav").read_bytes()) == a["items"][0]["decoded_sha"]


def test_awb_large_mixed_bank(cc):
    """A few hundred short clips of both codecs (the game-SFX shape of BASELINE configs[4]) against the oracle."""
    import struct
    from pycricodecs_amd import awb
    rng = np.random.default_rng(5)
    clips, kinds = [], []
    for i in range(120):
        n = int(rng.integers(2, 40)) * 160
        ch = 1 + int(rng.integers(0, 2))
        w = synth.wav(300 + i % 17, n, ch, 48000)
        if i % 2:
            clips.append(O.adx_encode(w)); kinds.append("adx")
        else:
            clips.append(O.hca_crypt(O.hca_encode(w, quality=1 + i % 3), 1, 56, KEY, 0x77)); kinds.append("hca")
    align, n = 0x20, len(clips)
    hs0 = 16 + 2 * n + 4 * (n + 1)
    hs = hs0 + (-hs0 % align)
    offs, pos, body = [hs0], hs, b""
    for cb in clips:
        cb = cb + b"\0" * (-len(cb) % align)
        body += cb; pos += len(cb); offs.append(pos)
    head = struct.pack("<4sBBHIHH", b"AFS2", 2, 4, 2, n, align, 0x77) + b"".join(struct.pack("<H", i) for i in range(n)) + b"".join(struct.pack("<I", o) for o in offs)
    bank = head.ljust(hs, b"\0") + body
    wavs = awb.AWB(bank).decode_all(KEY)
    for cb, kind, wv in zip(clips, kinds, wavs):
        ref = O.hca_decode(cb, KEY, 0x77) if kind == "hca" else O.adx_decode(cb)
        assert diff(wv, ref) is None


def test_drop_in_extension_module(cc):
    """The CPython module `CriCodecs` built from csrc/pyext gives the same bytes as the ctypes binding."""
    import importlib.util
    import os
    import sysconfig
    from pycricodecs_amd import build
    path = os.path.join(build.LIBDIR, "CriCodecs" + sysconfig.get_config_var("EXT_SUFFIX"))
    spec = importlib.util.spec_from_file_location("CriCodecs", path)
    ext = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ext)
    w = synth.wav(21, 4000, 2, 48000)
    adx = ext.AdxEncode(w, 4, 18, 3, 500, 0, 4, False)
    assert adx == cc.AdxEncode(w, 4, 18, 3, 500, 0, 4, False) == O.adx_encode(w)
    assert ext.AdxDecode(adx) == O.adx_decode(adx)
    hca = ext.HcaEncode(w, 0, 1)
    assert hca == O.hca_encode(w, 1)
    enc = ext.HcaCrypt(hca, 1, 96, 56, KEY, 0)
    assert enc == O.hca_crypt(hca, 1, 56, KEY) and hca == O.hca_encode(w, 1)      # input not mutated
    assert ext.HcaDecode(enc, 96, KEY, 0) == O.hca_decode(enc, KEY)
    with pytest.raises(ValueError, match="Decoding error"):
        ext.HcaDecode(enc, 96, KEY + 2, 0)
    with pytest.raises(ValueError, match="Bitdepth"):
        ext.AdxEncode(w, 1, 18, 3, 500, 0, 4, False)


# ------------------------------------------------------------------------------------------------ malformed inputs
def _both(gpu_call, ora_call):
    """Run the device path and the oracle on the same input: same accept/reject decision, same bytes when accepted."""
    try:
        ref = ora_call()
    except O.OracleError:
        ref = None
    from pycricodecs_amd._capi import CriCodecsError
    try:
        got = gpu_call()
    except CriCodecsError as e:
        if e.code == -304:                                               # documented "valid but not on the device path" (e.g. > 64 ADX channels)
            return "unsupported", ref
        got = None
    except (ValueError, NotImplementedError, RuntimeError):
        got = None
    return got, ref


@pytest.mark.parametrize("kind", ["hca", "adx", "wav_adx", "wav_hca"])
def test_header_mutation_fuzz(cc, kind):
    """Random byte edits and truncations in the header region: the host planners must take the oracle's accept/reject
    decision and produce its bytes (and, above all, must not read or write out of bounds doing so)."""
    import os
    rng = np.random.default_rng({"hca": 1, "adx": 2, "wav_adx": 3, "wav_hca": 4}[kind] + 10 * int(os.environ.get("CRI_FUZZ_SEED", "0")))
    w = synth.wav(77, 3008, 2, 48000)
    base = {"hca": O.hca_encode(w, quality=2), "adx": O.adx_encode(w), "wav_adx": w, "wav_hca": w}[kind]
    region = {"hca": 96, "adx": 40, "wav_adx": 44, "wav_hca": 44}[kind]
    agree_ok = 0
    import os
    for it in range(int(os.environ.get("CRI_FUZZ_ITERS", "600"))):
        b = bytearray(base)
        if it % 8 == 7:
            b = b[:int(rng.integers(0, len(b)))]                        # truncation
        else:
            for _ in range(int(rng.integers(1, 4))):
                p = int(rng.integers(0, min(region, len(b))))
                b[p] = int(rng.integers(0, 256)) if rng.random() < 0.5 else b[p] ^ (1 << int(rng.integers(0, 8)))
            if kind == "hca" and it % 2 == 0:                            # half of the edits keep a valid header checksum
                hs0 = int.from_bytes(base[6:8], "big")
                b[6:8] = base[6:8]
                b[hs0 - 2:hs0] = hca_forge.crc16(bytes(b[:hs0 - 2])).to_bytes(2, "big")
        data = bytes(b)
        if kind == "hca":
            hs = int.from_bytes(data[6:8], "big") if len(data) >= 8 else 0
            got, ref = _both(lambda: cc.HcaDecode(data, hs, 0, 0), lambda: O.hca_decode(data))
        elif kind == "adx":
            got, ref = _both(lambda: cc.AdxDecode(data), lambda: O.adx_decode(data))
        elif kind == "wav_adx":
            got, ref = _both(lambda: cc.AdxEncode(data, 4, 18, 3, 500, 0, 4, False), lambda: O.adx_encode(data))
        else:
            got, ref = _both(lambda: cc.HcaEncode(data, False, 1), lambda: O.hca_encode(data, quality=1))
        if got == "unsupported":
            continue
        assert (got is None) == (ref is None), (kind, it, "device %s, oracle %s" % ("rejects" if got is None else "accepts", "rejects" if ref is None else "accepts"))
        if ref is not None:
            assert diff(got, ref) is None, (kind, it)
            agree_ok += 1
    assert agree_ok > 5


# ------------------------------------------------------------------------------------------------ USM audio (@SFA) layer
def test_usm_audio_demux_golden(cc):
    """Device demux (+ AudioMask for keyed ADX) against the reference USM.demux() digests; decode of the demuxed streams."""
    from pycricodecs_amd import usm
    for d in MAN["usm"]["demux"]:
        u = usm.USM(G.load(d["file"]), key=d["key"])
        out = u.demux()
        assert list(out) == ["@SFA_0"] and u.codecs["@SFA_0"] == d["codec"]
        assert len(out["@SFA_0"]) == d["sfa_0_len"] and G.sha(bytes(out["@SFA_0"])) == d["sfa_0_sha"], d["file"]
        stream = G.load(d["stream"])
        assert bytes(out["@SFA_0"]) == stream
        wav = u.decode_audio()["@SFA_0"]
        assert diff(wav, O.hca_decode(stream) if d["codec"] == 4 else O.adx_decode(stream)) is None
    # mutated chunk headers (padding, channel, type, signature, data offset): the device job against the numpy statement of
    # the rule, which tests/test_oracle_vs_reference.py pins against the reference's USM.demux()
    import usm_model
    rng = np.random.default_rng(12)
    checked = 0
    for d in MAN["usm"]["demux"]:
        base = G.load(d["file"])
        key = int(d["key"], 16) if isinstance(d["key"], str) else int(d["key"])
        heads = [c["payload_offset"] - 0x20 for c in usm.usm_index(base)][3:]
        for _ in range(25):
            data = usm_model.mutate(base, heads, rng)
            try:
                want = usm_model.demux(data, key)
            except NotImplementedError:
                with pytest.raises(NotImplementedError):
                    usm.USM(data, key=d["key"]).demux()
                continue
            got = usm.USM(data, key=d["key"]).demux()
            assert {int(k[5:]): bytes(v) for k, v in got.items()} == {k: bytes(v) for k, v in want.items() if len(v)}, d["file"]
            checked += 1
    assert checked > 50
    # a keyed ADX container read without the key stays masked (and differs)
    d = [x for x in MAN["usm"]["demux"] if x["codec"] == 2 and x["key"]][0]
    assert G.sha(bytes(usm.USM(G.load(d["file"])).demux()["@SFA_0"])) != d["sfa_0_sha"]
    with pytest.raises(NotImplementedError):
        usm.USM(G.load(MAN["usm"]["ref_built"]["file"])).demux()
    with pytest.raises(NotImplementedError):
        usm.USM(b"ABCD" + bytes(100))


def test_sfa_chunks_match_reference_builder(cc):
    """The @SFA data chunks the reference's USMBuilder wrote for an HCA stream (golden container) are the chunks
    sfa_chunks() produces, in order; the header, frame times and the trailing "#CONTENTS END" included."""
    from pycricodecs_amd import usm
    rb = MAN["usm"]["ref_built"]
    built, hca = G.load(rb["file"]), G.load(rb["audio"])
    (chunks,) = usm.sfa_chunks([hca], "hca")
    hs, fs = int.from_bytes(hca[6:8], "big"), int.from_bytes(hca[28:30], "big")
    assert len(chunks) == 1 + (len(hca) - hs) // fs
    pos = 0
    for k, c in enumerate(chunks):
        at = built.find(c, pos)
        assert at >= 0 and at % 0x10 == 0, k                   # every chunk, byte for byte, in stream order
        pos = at + len(c)
    assert chunks[-1].endswith(b"#CONTENTS END   ===============\x00")
    # frame payloads come back out
    pay = b"".join(c[0x20:0x20 + int.from_bytes(c[4:8], "big") - 0x18 - int.from_bytes(c[10:12], "big")] for c in chunks)
    assert pay == hca


@pytest.mark.parametrize("codec,key", [("adx", 0), ("adx", 0x0123456789ABCDEF), ("hca", 0), ("hca", 0x7F4551499DF55E68)])
def test_sfa_pack_demux_round_trip(cc, codec, key):
    """sfa_chunks -> a container -> USM.demux gives the streams back (several channels, masked ADX included);
    ADX chunk sizes follow usm.py:1164-1166."""
    from pycricodecs_amd import usm
    streams = []
    for i in range(3):
        w = synth.wav(90 + i, 4800 + 3200 * i, 1 + i % 2, [48000, 44100, 32000][i])
        streams.append(O.adx_encode(w) if codec == "adx" else O.hca_encode(w, 1 + i))
    lists = usm.sfa_chunks(streams, codec, key=key, encrypt_audio=bool(key) and codec == "adx")
    crid = G.load(MAN["usm"]["demux"][0]["file"])[:0x800]
    body = b""
    for k in range(max(len(l) for l in lists)):                # interleave the channels' chunks
        for l in lists:
            if k < len(l):
                body += l[k]
    u = usm.USM(crid + body, key=key if codec == "adx" else False)
    out = u.demux()
    assert list(out) == ["@SFA_0", "@SFA_1", "@SFA_2"]
    for i, st in enumerate(streams):
        assert bytes(out["@SFA_%d" % i]) == st, i
    if codec == "adx":
        for st, l in zip(streams, lists):
            rate, ch, bs = int.from_bytes(st[8:12], "big"), st[7], st[5]
            expect = int(rate // 29.97 // 32) * (bs * ch)
            sizes = [int.from_bytes(c[4:8], "big") - 0x18 - int.from_bytes(c[10:12], "big") for c in l]
            assert sizes[0] == int.from_bytes(st[2:4], "big") + 4 and all(x == expect for x in sizes[1:-2]) and sizes[-1] == bs
            if key:                                            # payload bytes from 0x140 on are masked (usm.py:1290-1300)
                m = usm.audio_mask(key)
                c = l[1]
                pl = c[0x20:0x20 + sizes[1]]
                plain = st[sizes[0]:sizes[0] + sizes[1]]
                assert pl[:0x140] == plain[:0x140] and pl[0x140:] == bytes(b ^ m[j % 32] for j, b in enumerate(plain[0x140:]))
