"""GPU parity, round-3 additions (through the C ABI, against the pinned oracle and reference-generated vectors): the USM
builder's ADX @SFA branch pinned to the reference generator, the kernel instances round 2 left without a test (cipher
tables in global memory, the oversized-frame crypt kernel), the segmented ADX kernels and the pipelined host path."""
import json
import os

import numpy as np
import pytest

import golden_util as G
import oracle_lib as O
from pycricodecs_amd import synth

pytestmark = pytest.mark.gpu
KEY = G.KEY
MAN = G.manifest()
with open(os.path.join(G.GOLDEN, "sfa_adx.json")) as _f:
    SFA_ADX = json.load(_f)


@pytest.fixture(scope="module")
def cc():
    from pycricodecs_amd import CriCodecs, _capi
    assert _capi.lib().cri_device_available() == 1, "no HIP device: the GPU tests must run on the HIP path"
    return CriCodecs


def run_job(job):
    if os.environ.get("CRI_TEST_HOST_RUN") == "1":             # tools/asan_gpu.sh: no torch in the process -- through the library's own host path
        outs, st = job.run_host()
        return [bytes(o) for o in outs], st
    import torch
    bufs = job.alloc("cuda:0")
    job.run(*bufs)
    torch.cuda.synchronize()
    blob = bytes(bufs[1].cpu().numpy())
    status = bufs[3].cpu().numpy()[:job.n]
    return job.split(blob), status


# ------------------------------------------------------------------------------------------------ f3: USM builder, ADX branch
@pytest.mark.parametrize("c", SFA_ADX["cases"], ids=lambda c: "%s-key%x" % (c["file"], c["key"]))
def test_sfa_adx_chunks_match_reference_generator(cc, c):
    """usm.py:584-657 (the ADX branch of the @SFA generator) run UNMODIFIED in the build container over stand-in stream
    objects (tests/golden/make_golden_sfa_adx.py): every chunk -- header, size / padding / frame time fields, payload,
    AudioMask-ed bytes, the stream's tail block and the "#CONTENTS END" chunk glued to it -- byte for byte.  Includes
    streams shorter than one chunk (the floor modulo of usm.py:598 on a negative operand)."""
    from pycricodecs_amd import usm
    adx = G.load(c["file"])
    assert G.sha(adx) == c["adx_sha"]
    (chunks,) = usm.sfa_chunks([adx], "adx", key=c["key"], encrypt_audio=bool(c["key"]))
    assert len(chunks) == c["n_chunks"]
    for k, (got, want) in enumerate(zip(chunks, c["chunks"])):
        assert len(got) == want["len"] and int.from_bytes(got[4:8], "big") == want["size_field"], k
        assert int.from_bytes(got[10:12], "big") == want["padding"] and int.from_bytes(got[16:20], "big") == want["frame_time"], k
        assert G.sha(got) == want["sha"], k
    assert G.sha(b"".join(chunks)) == c["all_sha"]


def test_sfa_adx_two_streams_match_reference_generator(cc):
    """Two ADX streams in one builder: the channel number of a chunk is the stream's index (usm.py:606), chunk sizes are per
    stream (usm.py:1164-1166)."""
    from pycricodecs_amd import usm
    m = SFA_ADX["multi"]
    lists = usm.sfa_chunks([G.load(f) for f in m["files"]], "adx")
    assert [len(l) for l in lists] == m["n_chunks"]
    assert [G.sha(b"".join(l)) for l in lists] == m["all_sha"]


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


# ------------------------------------------------------------------------------------------------ host path (b)
def _ragged_hca_batch():
    """HCA streams of ragged lengths (one of them shorter than a frame's delay, i.e. no samples), one rejected header in the
    middle, repeats of the same bytes object."""
    rng = np.random.default_rng(5)
    uniq = [O.hca_crypt(O.hca_encode(synth.wav(300 + k, int(rng.integers(200, 30000)), 2, 48000), 1), 1, 56, KEY) for k in range(9)]
    items = [uniq[int(k)] for k in rng.integers(0, len(uniq), 70)]
    items[17] = b"HCA\0" + bytes(200)                           # rejected on the host
    items[40] = uniq[0][:96]                                    # a header without any frame
    return uniq, items


@pytest.mark.parametrize("order", ["default", "pipelined", "pipelined-small-pieces", "one-piece"])
def test_run_host_equals_device_resident_run(cc, knobs, order):
    """cri_job_run_host_items / _into (host buffers in and out, the arena's private streams) give the bytes and statuses of the
    device-resident cri_job_run, and those are the oracle's -- in one piece and pipelined (CRICODECS_HOST_SLICE_MIN=0: tile slices,
    uploads pulled by k_pull_host from page-locked or staged memory, downloads beside them), from separate items (staged; with
    CRICODECS_HOST_STAGE_PIECE=1000 every item straddles several staging pieces), from a pageable blob (locked for the call), from
    a page-locked blob, into pageable and into page-locked memory."""
    import ctypes as C
    from pycricodecs_amd import _capi
    from pycricodecs_amd.batch import Job, pinned_array
    if order != "default":
        knobs(host_slice_min=(1 << 62) if order == "one-piece" else 0)
    if order == "pipelined-small-pieces":
        knobs(host_stage_piece=1000)
    uniq, items = _ragged_hca_batch()
    job = Job.hca_decode(items, keys=[KEY] * len(items))
    assert job.host_status[17] != 0
    want, st_dev = run_job(job)
    st_want = np.where(job.host_status != 0, job.host_status, st_dev)
    for rep in range(2):                                        # (the second call runs on the cached arena)
        outs, st = job.run_host()
        assert (st == st_want).all()
        for i, (a, b) in enumerate(zip(outs, want)):
            assert bytes(a) == bytes(b), (rep, i)
    refs = {id(u): O.hca_decode(u, KEY) for u in uniq}
    for i, it in enumerate(items):
        if id(it) in refs:
            assert bytes(want[i]) == refs[id(it)], i
    # into page-locked memory
    buf = pinned_array(job.output_bytes)
    outs, st = job.run_host(out=buf)
    for i, (a, b) in enumerate(zip(outs, want)):
        assert bytes(a) == bytes(b), i
    # the blob form: a pageable blob, then the same bytes in page-locked memory at an odd address
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
    # bytes no kernel writes are zero, in every order (alignment gaps between the items)
    o = job.output_offsets
    for i in range(job.n - 1):
        end = int(o[i]) + len(want[i])
        assert not page_out[end:int(o[i + 1])].any(), i
    del outs
    del buf, pin_in


@pytest.mark.parametrize("order", ["pipelined", "pipelined-small-pieces", "one-piece"])
def test_run_host_items_at_caller_offsets(cc, knobs, order):
    """HCA streams placed at caller offsets with gaps between them (the gaps are zero on the device whatever the staging slots
    held before), decoded from the items' own buffers."""
    from pycricodecs_amd.batch import Job
    knobs(host_slice_min=(1 << 62) if order == "one-piece" else 0)
    if order == "pipelined-small-pieces":
        knobs(host_stage_piece=777)
    uniq, items = _ragged_hca_batch()
    items = [it for k, it in enumerate(items) if k != 17][:40]
    offs = np.zeros(len(items) + 1, dtype=np.uint64)
    for i, it in enumerate(items):
        offs[i + 1] = (int(offs[i]) + len(it) + 1000 + 37 * i) // 64 * 64
    big = Job.hca_decode([uniq[0]] * 3, keys=[KEY] * 3)         # leaves non-zero bytes in the arena's input buffer and staging slots
    big.run_host()
    job = Job.hca_decode(items, keys=[KEY] * len(items), offsets=offs)
    want, st_dev = run_job(job)
    outs, st = job.run_host()
    assert (st == np.where(job.host_status != 0, job.host_status, st_dev)).all()
    for i, (a, b) in enumerate(zip(outs, want)):
        assert bytes(a) == bytes(b), i
    refs = {id(u): O.hca_decode(u, KEY) for u in uniq}
    for i, it in enumerate(items):
        if id(it) in refs:
            assert bytes(outs[i]) == refs[id(it)], i


def test_run_host_blob_form_and_items_with_offsets(cc):
    """The blob form (cri_job_run_host_into) on a job made from one blob; a job made from items placed at caller offsets has no
    blob form (CRI_ERR_INVALID_ARG) and runs through cri_job_run_host_items."""
    import ctypes as C
    from pycricodecs_amd import _capi
    from pycricodecs_amd.batch import Job, pack
    adx = [O.adx_encode(synth.wav(610 + k, 3200 + 640 * k, 1 + k % 2, 48000)) for k in range(5)]
    refs = [O.adx_decode(a) for a in adx]
    blob, offs = pack(adx)
    h = C.c_void_p()
    rc = _capi.lib().cri_job_create_adx_decode(blob, offs.ctypes.data_as(C.POINTER(C.c_uint64)), len(adx), C.byref(h))
    assert rc == 0
    job = Job(h, blob, offs)
    outs, st = job.run_host()
    assert not st.any() and [bytes(o) for o in outs] == refs
    # items at 256-byte aligned device offsets
    aligned = np.zeros(len(adx) + 1, dtype=np.uint64)
    for i, a in enumerate(adx):
        aligned[i + 1] = (int(aligned[i]) + len(a) + 255) // 256 * 256
    job2 = Job.adx_decode(adx, offsets=aligned)
    outs, st = job2.run_host()
    assert not st.any() and [bytes(o) for o in outs] == refs
    out = np.empty(max(job2.output_bytes, 1), dtype=np.uint8)
    status = (C.c_int32 * len(adx))()
    assert _capi.lib().cri_job_run_host_into(job2._h, blob, out.ctypes.data, status) == -301


def test_single_file_calls_reuse_the_arena(cc):
    """Back-to-back single-file calls of different sizes and kinds (the arena grows, is reused, and cri_release_cache drops it)."""
    from pycricodecs_amd import _capi
    for rep in range(3):
        for n in (320, 48000, 4800, 96000):
            w = synth.wav(900 + n % 7, n, 2, 48000)
            a = cc.AdxEncode(w, 4, 18, 3, 500, 0, 4, False)
            assert a == O.adx_encode(w) and cc.AdxDecode(a) == O.adx_decode(a)
            h = cc.HcaEncode(w, False, 1)
            assert h == O.hca_encode(w, 1) and cc.HcaDecode(h, 96, 0, 0) == O.hca_decode(h)
        _capi.lib().cri_release_cache()


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


def test_run_host_from_two_threads_at_once(cc, knobs):
    """Two threads in the pipelined host path at the same time (one works on the device's arena, the other on buffers, streams
    and staging slots of its own for the call), both from the same items and into their own buffers, several times over."""
    import threading
    from pycricodecs_amd.batch import Job
    knobs(host_slice_min=0)
    knobs(host_stage_piece=4096)
    uniq, items = _ragged_hca_batch()
    items = [it for k, it in enumerate(items) if k != 17]
    refs = {id(u): O.hca_decode(u, KEY) for u in uniq}
    errors = []

    def work(seed):
        try:
            order = list(np.random.default_rng(seed).permutation(len(items)))
            mine = [items[i] for i in order]
            job = Job.hca_decode(mine, keys=[KEY] * len(mine))
            for rep in range(4):
                outs, st = job.run_host()
                for i, it in enumerate(mine):
                    if id(it) in refs and bytes(outs[i]) != refs[id(it)]:
                        errors.append((seed, rep, i)); return
        except Exception as e:                                  # noqa: BLE001
            errors.append((seed, repr(e)))

    ts = [threading.Thread(target=work, args=(s,)) for s in (1, 2, 3)]
    for t in ts: t.start()
    for t in ts: t.join()
    assert not errors, errors[:3]
