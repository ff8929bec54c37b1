"""GPU parity, through the C ABI, against the pinned CPU oracle and the committed golden vectors.
HCA decode (rows a10-a25, a38, f2): header forms, parse + transform kernels of every layout (plain / joint / wide / noise fill), pre-clamp floats at 0 ULP, crypt (a37).  PCM16 bit-exact."""
import numpy as np
import pytest

import golden_util as G
import hca_forge
import oracle_lib as O
from gpu_common import KEY, MAN, cc, diff, run_job, run_job_floats  # noqa: F401
from pycricodecs_amd import synth

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------------ a12 / a37: kernel instances without a test
def _many_key_streams(n):
    rng = np.random.default_rng(77)
    items, keys, subkeys, plain = [], [], [], []
    for i in range(n):
        w = synth.wav(800 + i, 1024 * int(rng.integers(2, 7)) + int(rng.integers(0, 900)), 1 + i % 2, 48000)
        h = O.hca_encode(w, 1 + i % 3)
        key = int(rng.integers(1, 2**63)) * 2 + 1
        sub = int(rng.integers(0, 65536)) if i % 3 == 0 else 0
        plain.append(h)
        items.append(O.hca_crypt(h, 1, 56, key, sub))
        keys.append(key)
        subkeys.append(sub)
    return items, keys, subkeys, plain


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
    outs, status, _ = run_job_floats(job)
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
    outs, status, (d_f, offs) = run_job_floats(job, floats=True)
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
    outs, status, (d_f, offs) = run_job_floats(job, floats=True)
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


@pytest.mark.parametrize("ch,q,v3", [(1, 1, False), (2, 1, False), (2, 3, False), (4, 2, False), (6, 1, False), (8, 3, False), (3, 1, False), (5, 2, False), (2, 1, True), (6, 1, True), (6, 2, True)])
def test_pcm_of_the_shipped_instances_is_the_clamp_of_the_validation_floats(cc, ch, q, v3):
    """The floats are stored by the `FLT = true` instances of the transform kernels (cri_job_run_floats), the shipped run uses the
    `FLT = false` ones: same arithmetic, different kernels.  Here both run on the same streams: the shipped run's WAVs equal the
    validation run's byte for byte, and their samples are (int)(f * 32768) of the validation run's floats, clamped to int16
    (hca.cpp:1987-1992 as the x86-64 build evaluates it), trimmed by the encoder delay."""
    from pycricodecs_amd.batch import Job
    items = []
    for seed, n in ((60, 7000), (61, 1024), (62, 2500)):
        h = O.hca_encode(synth.wav(seed + ch, n, ch, 48000), q)
        items.append(hca_forge.forge_v3(h, 0) if v3 else h)
    outs_v, status_v, (d_f, offs) = run_job_floats(Job.hca_decode(items), floats=True)
    outs_s, status_s = run_job(Job.hca_decode(items))
    fl = d_f.cpu().numpy()
    good = 0
    accepted = 0
    for i, h in enumerate(items):
        assert status_s[i] == status_v[i], i
        try:
            O.hca_decode(h)
            accepted += 1
        except O.OracleError:                                  # (a forged v3.0 header on frames of another layout -- all of 6 ch / quality 2: rejected by every side)
            assert status_s[i] != 0, i
            continue
        assert status_s[i] == 0, i
        good += 1
        assert bytes(outs_s[i]) == bytes(outs_v[i]), i
        wav = bytes(outs_s[i])
        at = wav.find(b"data") + 8
        pcm = np.frombuffer(wav, dtype="<i2", offset=at)
        delay, nch = int.from_bytes(h[20:22], "big"), h[12]
        v = (fl[int(offs[i]):int(offs[i + 1])] * np.float32(32768.0)).astype(np.float64)
        qv = np.where(np.isfinite(v) & (v > -2147483904.0) & (v < 2147483648.0), np.trunc(v), -2147483648.0)
        qv = np.clip(qv, -32768, 32767).astype(np.int16)
        assert np.array_equal(qv[delay * nch:delay * nch + pcm.size], pcm), i
    assert good == accepted and (good >= 2 or (ch, q, v3) == (6, 2, True))


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
    outs, status, (d_f, offs) = run_job_floats(job, floats=True)
    fl = d_f.cpu().numpy()
    for i, h in enumerate(items):
        ref = O.hca_decode_float(h)
        assert np.array_equal(fl[int(offs[i]):int(offs[i + 1])].view(np.uint32), ref.view(np.uint32)), i


def test_hca_decode_more_than_16_cipher_tables(cc):
    """A decode job whose streams carry 24 distinct keys (and subkeys): past 16 tables the parse kernel reads the cipher tables
    from global memory instead of an LDS copy -- k_hca_parse<false, false> (cri_hca_dec.hip, launch_hca_parse)."""
    from pycricodecs_amd.batch import Job
    items, keys, subkeys, _ = _many_key_streams(24)
    assert len(set(zip(keys, subkeys))) == 24
    job = Job.hca_decode(items, keys=keys, subkeys=subkeys)
    outs, status = run_job(job)
    assert not status.any() and not job.host_status.any()
    for i, (o, it) in enumerate(zip(outs, items)):
        assert bytes(o) == O.hca_decode(it, keys[i], subkeys[i]), i
    # a wrong key among them: that stream alone does what the oracle does with it (the frame checksum is taken over the
    # ciphered bytes, hca.cpp:1166-1169, so a wrong key is an unpack error or garbage PCM, not a checksum error)
    bad = list(keys)
    bad[5] ^= 0x10
    job = Job.hca_decode(items, keys=bad, subkeys=subkeys)
    outs, status = run_job(job)
    assert not np.delete(status, 5).any()
    try:
        want = O.hca_decode(items[5], bad[5], subkeys[5])
    except O.OracleError:
        want = None
    assert (status[5] != 0) == (want is None)
    if want is not None:
        assert bytes(outs[5]) == want


@pytest.mark.parametrize("encrypt", [1, 0])
def test_hca_crypt_more_than_16_cipher_tables(cc, encrypt):
    """HcaCrypt as one job over 24 streams with 24 keys, both directions, against the oracle."""
    from pycricodecs_amd.batch import Job
    enc, keys, subkeys, plain = _many_key_streams(24)
    src = plain if encrypt else enc
    job = Job.hca_crypt(src, encrypt, 56 if encrypt else 0, keys=keys, subkeys=subkeys)
    outs, status = run_job(job)
    assert not status.any() and not job.host_status.any()
    for i, o in enumerate(outs):
        assert bytes(o)[:len(src[i])] == O.hca_crypt(src[i], encrypt, 56 if encrypt else 0, keys[i], subkeys[i]), i


@pytest.mark.parametrize("fs", [8, 65400, 65535])
def test_hca_crypt_frames_that_do_not_fit_lds(cc, fs):
    """Frames near the 16-bit frame-size limit do not fit the wave-per-frame kernel's LDS image: launch_hca_crypt falls back to
    k_hca_crypt, lane per frame (8-byte frames are the wave-per-frame kernel's lower edge).  Oracle bytes (pinned to the reference for these
    sizes in tests/test_oracle_vs_reference.py::test_hca_crypt_extreme_frame_sizes), single call and as a job beside
    ordinary streams."""
    import hca_forge
    from pycricodecs_amd.batch import Job
    base = O.hca_encode(synth.wav(5, 3000, 2, 48000), 1)
    s = hca_forge.frame_size_stream(base, fs, 3, fs)
    hs = int.from_bytes(s[6:8], "big")
    e = cc.HcaCrypt(s, 1, hs, 56, KEY, 0)
    assert e == O.hca_crypt(s, 1, 56, KEY)
    assert cc.HcaCrypt(e, 0, hs, 0, KEY, 0) == s
    assert cc.HcaCrypt(s, 1, hs, 1, 0, 0) == O.hca_crypt(s, 1, 1, 0)
    job = Job.hca_crypt([base, s, base], 1, 56, keys=[KEY, KEY + 2, 0x1234567])
    outs, status = run_job(job)
    assert not status.any()
    for o, (src, k) in zip(outs, [(base, KEY), (s, KEY + 2), (base, 0x1234567)]):
        assert bytes(o)[:len(src)] == O.hca_crypt(src, 1, 56, k)


# ------------------------------------------------------------------------------------------------ wide layouts on the in-lane transform
@pytest.mark.parametrize("ch", [3, 5, 6, 7, 8])
def test_wide_plain_layouts_trims_and_alignments(cc, ch):
    """k_hca_transform_plain's wide form (a wave per four channels, whole sample frames stored from a shared staging piece):
    3, 5, 6, 7 and 8 channels against the oracle, with the delay / padding trims that decide how the PCM leaves -- an even
    delay (16-byte stores), an odd delay on an odd channel count (the sample-by-sample path for every frame), trims inside the
    first and the last frame, a trim longer than a frame, encrypted and plain, streams that end inside a run of eight frames."""
    import hca_forge
    from pycricodecs_amd.batch import Job
    items, keys = [], []
    for k, (n, delay, pad) in enumerate([(9000, 128, 0), (9000, 1, 0), (12000, 127, 77), (2048 * 5, 1029, 1500), (700, 0, 3), (1024 * 9, 2, 1)]):
        h = O.hca_encode(synth.wav(2100 + 10 * ch + k, n, ch, 48000), 1)
        h = hca_forge.forge_trim(h, delay, pad)
        if k % 2:
            h = O.hca_crypt(h, 1, 56, KEY)
        items.append(h); keys.append(KEY if k % 2 else 0)
    job = Job.hca_decode(items, keys=keys)
    outs, st = run_job(job)
    assert not st.any() and not job.host_status.any()
    for i, (o, h, key) in enumerate(zip(outs, items, keys)):
        assert bytes(o) == O.hca_decode(h, key), (ch, i)


# ------------------------------------------------------------------------------------------------ the general transform kernel stays covered
@pytest.mark.parametrize("ch,q,v3", [(1, 2, False), (2, 2, False), (2, 4, False), (4, 2, False), (4, 3, False), (6, 2, False), (8, 3, False), (3, 2, False), (5, 2, False), (7, 3, False),
                                      (2, 1, True), (2, 2, True), (1, 1, True), (4, 1, True), (3, 1, True), (5, 1, True)])
def test_general_transform_kernel_on_formats_the_inlane_kernel_takes(cc, knobs, ch, q, v3):
    """Joint-stereo / HFR / noise-fill formats of 1, 2, 4 (and, without noise fill, 6 and 8) channels run on k_hca_transform_plain's
    joint, wide and noise instances; k_hca_transform<false, C> and k_hca_transform_generic -- what noise fill on 3 and 5 to 8
    channels still uses -- are forced onto the same streams here (CRI_NO_INLANE, read when the job is created): floats and PCM equal to the oracle's,
    bit for bit."""
    import hca_forge
    import torch
    from pycricodecs_amd.batch import Job
    knobs(no_inlane=1)
    items = []
    for seed, n in ((50, 5000), (51, 9000), (52, 1024)):
        h = O.hca_encode(synth.wav(seed + ch, n, ch, 48000), q)
        items.append(hca_forge.forge_v3(h, 0) if v3 else h)
    job = Job.hca_decode(items)
    bufs = job.alloc("cuda:0")
    d_f, offs = job.run_floats(*bufs)
    torch.cuda.synchronize()
    outs = job.split(bytes(bufs[1].cpu().numpy()))
    status = bufs[3].cpu().numpy()[:job.n]
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
        assert np.array_equal(mine.view(np.uint32), ref.view(np.uint32)), i
        assert bytes(outs[i]) == O.hca_decode(h), i
    assert good >= 2


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
