"""GPU parity, through the C ABI, against the pinned CPU oracle and the committed golden vectors.
HCA encode (rows a26-a36): k_hca_encode against the oracle and the reference's golden bytes, every quality and channel count, loop feeding.  Bit-exact."""
import ctypes as C

import numpy as np
import pytest

import golden_util as G
import oracle_lib as O
from gpu_common import KEY, MAN, cc, diff, run_job, run_job_floats  # noqa: F401
from pycricodecs_amd import synth

pytestmark = pytest.mark.gpu


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


# ------------------------------------------------------------------------------------------------ a35: a payload that runs into the checksum field
def test_hca_encode_when_the_payload_runs_into_the_checksum_field(cc):
    """The rate loop's bit count can come out a bit short of what the pack writes (hca.cpp:2771-2786 against 2920-2938), and the bit writer's
    buffer reaches to the frame's last byte (hca.cpp:2941): a full frame's last code then ends inside the two checksum bytes, which the
    reference OVERWRITES with the checksum (hca.cpp:2961-2962).  The library or-ed its checksum onto the stray bit (found by
    tools/parity_soak.py, one frame in 200 000 files: eight channels, 8 kHz, lowest quality; the fixture is four frames cut out around it).
    Bytes = oracle, through the single-file call and the batch job."""
    from pycricodecs_amd.batch import Job
    w = G.load("enc_payload_reaches_checksum_8ch_q4.wav")
    ref = O.hca_encode(w, 4)
    assert bytes(cc.HcaEncode(w, False, 4)) == ref
    outs, st = run_job(Job.hca_encode([w, w, synth.wav(1, 5000, 8, 8000), w], quality=4))
    assert not st.any()
    assert [bytes(o) for o in outs] == [ref, ref, O.hca_encode(synth.wav(1, 5000, 8, 8000), 4), ref]
