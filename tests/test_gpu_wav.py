"""GPU parity, through the C ABI, against the pinned CPU oracle and the committed golden vectors.
PCM / WAV glue (rows a8, a9, f4): non-16-bit WAV input, looping WAVs through both codecs, decoded WAV headers and placement."""
import numpy as np
import pytest

import golden_util as G
import oracle_lib as O
from gpu_common import KEY, MAN, cc, diff, run_job, run_job_floats  # noqa: F401
from pycricodecs_amd import synth

pytestmark = pytest.mark.gpu


def test_golden_stored_decodes(cc):
    h = G.load("s0_3008_2_48000_q1.hca")
    assert diff(cc.HcaDecode(h, 96, 0, 0), G.load("s0_3008_2_48000_q1.decoded.wav")) is None
    a = G.load("s0_3008_2_48000_bd4_bs18_m3_v4.adx")
    assert diff(cc.AdxDecode(a), G.load("s0_3008_2_48000_bd4_bs18_m3_v4.decoded.wav")) is None


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
    outs, status, _ = run_job_floats(Job.hca_decode([hca] + hq, keys=[KEY, 0, 0]))
    assert not status.any() and bytes(outs[0]) == ref and [bytes(o) for o in outs[1:]] == [O.hca_decode(h) for h in hq]
    for mapping_items in (3, 40):                              # wave-per-file and (padded with short clips) still correct
        items = [adx] + [O.adx_encode(synth.wav(92 + i, 640, 2, 48000)) for i in range(mapping_items - 1)]
        outs, status, _ = run_job_floats(Job.adx_decode(items))
        assert not status.any() and bytes(outs[0]) == O.adx_decode(adx)


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
