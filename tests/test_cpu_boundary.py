"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/cricodecs_hip.h
declares, and fails loudly (no CPU fallback) when there is no device.  No compute calls."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    from pycricodecs_amd import _capi, build
    build.build(verbose=False)
    return _capi


def test_exports_match_header(capi):
    hdr = open(os.path.join(ROOT, "include", "cricodecs_hip.h")).read()
    declared = set(re.findall(r"\b(cri_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"cri_job"}
    L = capi.lib()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert declared == set(capi.SYMBOLS)


def test_strerror_matches_reference_messages(capi):
    assert capi.strerror(-1) == "Invalid ADX file header."
    assert capi.strerror(-3) == "Encrypted ADX detected, unsupported."
    assert capi.strerror(-17) == "Provided Bitdepth does not fit correctly with the provided BlockSize"
    assert capi.strerror(-102) == "Invalid WAVE file header. Format info is not present."
    assert capi.strerror(-201) == "Header decoding error, the header is not a valid HCA header."
    assert capi.strerror(-202) == "Decoding error, either an incorrect key or an unknown exception."
    with pytest.raises(NotImplementedError):
        capi.raise_for(-3)
    with pytest.raises(ValueError):
        capi.raise_for(-9)
    with pytest.raises(ValueError):
        capi.raise_for(-213)


def test_no_cpu_fallback(capi):
    import torch
    if torch.cuda.is_available():
        pytest.skip("device present")
    from pycricodecs_amd import ADX, HCA
    import golden_util as G
    with pytest.raises(capi.CriCodecsError) as e:
        ADX.decode(G.load("s0_3008_2_48000_bd4_bs18_m3_v4.adx"))
    assert e.value.code == -303
    h = HCA(G.load("s0_3008_2_48000_q1.hca"))
    assert h.info()["FrameSize"] == 682 and h.info()["ChannelCount"] == 2
    with pytest.raises(capi.CriCodecsError):
        h.decode()


def test_front_end_header_walk():
    import golden_util as G
    from pycricodecs_amd import HCA
    h = HCA(G.load("s0_3008_2_48000_q1.hca"))
    i = h.info()
    assert (i["version"], i["HeaderSize"], i["SampleRate"], i["FrameCount"], i["EncoderDelay"]) == ("0x200", 96, 48000, 4, 128)
    assert (i["TotalBandCount"], i["BaseBandCount"], i["StereoBandCount"], i["CipherType"]) == (128, 128, 0, 0)
    assert h.filetype == "hca" and not h.encrypted
    frames = list(h.get_frames())
    assert len(frames) == 4 and all(len(f) == 682 and f[:2] == b"\xff\xff" for _, f in frames)
    w = HCA(G.load("s0_3008_2_48000.wav"))
    assert w.filetype == "wav" and w.info()["fmtChannelCount"] == 2 and w.info()["dataSize"] == 3008 * 4
    with pytest.raises(ValueError):
        w.decode()
    with pytest.raises(ValueError):
        HCA(b"nonsense-bytes-here")
    with pytest.raises(OverflowError):
        HCA(G.load("s0_3008_2_48000_q1.hca"), key=1 << 64)


def test_awb_index_matches_reference_reader():
    """cri_awb_index (host only) against what the reference's AWB class read from a bank its AWBBuilder wrote
    (tests/golden/make_golden.py): header fields, aligned offsets, item bytes; the oracle decodes every item to the
    reference's digest (HCA items with the bank's subkey mixed into the key, awb.py:72)."""
    import golden_util as G
    import oracle_lib as O
    from pycricodecs_amd import awb
    a = G.manifest()["awb"]
    bank = G.load(a["file"])
    assert G.sha(bank) == a["sha"]
    offs, kinds, subkey = awb.awb_index(bank)
    assert [int(x) for x in offs] == a["ofs"] and subkey == a["subkey"]
    assert [{1: "hca", 2: "adx"}[int(k)] for k in kinds] == [i["kind"] for i in a["items"]]
    for k, it in enumerate(a["items"]):
        item = bank[int(offs[k]):int(offs[k + 1])]
        assert len(item) == it["len"] and G.sha(item) == it["sha"]
        dec = O.hca_decode(item, G.KEY, subkey) if it["kind"] == "hca" else O.adx_decode(item)
        assert G.sha(dec) == it["decoded_sha"]
    with pytest.raises(ValueError):
        awb.awb_index(b"AFS3" + bank[4:])
    with pytest.raises(ValueError):
        awb.awb_index(bank[:5] + b"\x03" + bank[6:])          # offset int size 3: "Unknown int size."
    with pytest.raises(ValueError):
        awb.awb_index(bank[:40])



def test_info_dictionaries_match_the_reference_class():
    """HCA.info() of the mirror class against dictionaries captured from the reference's own PyCriCodecs.hca.HCA
    (tests/golden/make_golden.py): plain / encrypted / default-key / looped / v3.0 HCA headers and WAV inputs."""
    import golden_util as G
    from pycricodecs_amd.hca import HCA
    for ent in G.manifest()["py_info"]:
        got = HCA(G.load(ent["file"]), key=ent["key"]).info()
        got = {k: (v if isinstance(v, (int, str, float, bool, type(None))) else repr(v)) for k, v in got.items()}
        assert got == ent["info"], ent["label"]


def test_usm_audio_mask_and_index_match_reference(capi):
    """cri_usm_audio_mask against the masks the reference's USM.init_key derives, and cri_usm_index (host only) on the
    golden containers: the chunk walk, the payload ranges (their concatenation minus padding is the reference demuxer's
    stream), and the reference's failures (no CRID, unknown chunk, its own builder's malformed container)."""
    import golden_util as G
    from pycricodecs_amd import usm
    u = G.manifest()["usm"]
    for m in u["masks"]:
        assert usm.audio_mask(m["key"]).hex() == m["mask"], m["key"]
    with pytest.raises(ValueError):
        usm.audio_mask("0" * 17)
    for d in u["demux"]:
        data = G.load(d["file"])
        chunks = usm.usm_index(data)
        assert chunks[0]["fourcc"] == b"CRID" and chunks[-1]["fourcc"] == b"@SFA" and chunks[-1]["type"] == 2
        sfa = [c for c in chunks if c["fourcc"] == b"@SFA" and c["type"] == 0]
        if d["key"] is False or d["codec"] == 4:               # unmasked payloads: the gather alone reproduces the stream
            got = b"".join(data[c["payload_offset"]:c["payload_offset"] + c["payload_len"] - c["padding"]] for c in sfa)
            assert G.sha(got) == d["sfa_0_sha"] and len(got) == d["sfa_0_len"]
        assert sum(c["payload_len"] - c["padding"] for c in sfa) == d["sfa_0_len"]
    assert not u["ref_built"]["reference_demux_ok"]            # the reference cannot read what its builder writes ...
    with pytest.raises(NotImplementedError, match="Unsupported chunk type"):
        usm.usm_index(G.load(u["ref_built"]["file"]))          # ... and neither do we, the same way
    with pytest.raises(NotImplementedError, match="Unsupported file type"):
        usm.usm_index(b"RIFF" + bytes(60))
    good = G.load(u["demux"][0]["file"])
    with pytest.raises(NotImplementedError):
        usm.usm_index(good[:-0x30])                            # a chunk header cut short (the reference dies in struct.unpack)
    assert len(usm.usm_index(good[:-7])) == len(usm.usm_index(good))   # a short last payload is read as far as it goes
