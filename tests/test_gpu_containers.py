"""GPU parity, through the C ABI, against the pinned CPU oracle and the committed golden vectors.
Next rows f1 / f3: the AFS2 (AWB) front door and the USM @SFA audio layer."""
import json
import os
import struct

import numpy as np
import pytest

import golden_util as G
import oracle_lib as O
from gpu_common import KEY, MAN, cc, diff, run_job, run_job_floats  # noqa: F401
from pycricodecs_amd import synth

pytestmark = pytest.mark.gpu


with open(os.path.join(G.GOLDEN, "sfa_adx.json")) as _f:
    SFA_ADX = json.load(_f)


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
