#!/usr/bin/env python3
"""Generates tests/golden/* from the REAL reference (oracle/_ref/criref built from /root/reference).
Run in the build container only:  python tests/golden/make_golden.py
Writes small binary fixtures (inputs and the reference's outputs) plus manifest.json (sha256 of every
reference output, including the ones too large to commit)."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import hca_forge  # noqa: E402
import ref_tool as R  # noqa: E402
from pycricodecs_amd import synth  # noqa: E402

KEY = 0xCF222F1FE0748978


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def main():
    assert R.available(), "build oracle/_ref/criref first (make -C oracle ref)"
    man = {"key": hex(KEY), "cases": []}

    def put(name, data):
        with open(os.path.join(HERE, name), "wb") as f:
            f.write(data)

    wavs = [("s0_3008_2_48000", (0, 3008, 2, 48000)), ("s1_2048_1_44100", (1, 2048, 1, 44100)),
            ("s2_1600_2_22050", (2, 1600, 2, 22050))]
    for tag, (seed, n, ch, sr) in wavs:
        w = synth.wav(seed, n, ch, sr)
        put(tag + ".wav", w)
        case = {"wav": tag + ".wav", "wav_sha": sha(w), "adx": [], "hca": []}
        for (bd, bs, mode, filt, ver) in [(4, 18, 3, 0, 4), (4, 18, 4, 0, 4), (4, 18, 2, 0, 3), (8, 18, 3, 0, 5), (2, 10, 3, 0, 4)]:
            adx = R.adx_encode(w, bd, bs, mode, 500, filt, ver)
            dec = R.adx_decode(adx)
            name = "%s_bd%d_bs%d_m%d_v%d.adx" % (tag, bd, bs, mode, ver)
            put(name, adx)
            case["adx"].append({"file": name, "params": [bd, bs, mode, 500, filt, ver], "sha": sha(adx), "decoded_sha": sha(dec)})
        for q in (0, 1, 2, 3):
            hca = R.hca_encode(w, q)
            name = "%s_q%d.hca" % (tag, q)
            put(name, hca)
            enc = R.hca_crypt(hca, 1, 56, KEY)
            enc1 = R.hca_crypt(hca, 1, 1, 0)
            ent = {"file": name, "quality": q, "sha": sha(hca), "decoded_sha": sha(R.hca_decode(hca)),
                   "float_sha": sha(R.hca_decode_float(hca).tobytes()),
                   "enc56_sha": sha(enc), "enc56_decoded_sha": sha(R.hca_decode(enc, KEY)),
                   "enc56_sub_sha": sha(R.hca_crypt(hca, 1, 56, 0x1234567, 0x4321)),
                   "enc1_sha": sha(enc1), "dec_of_enc56_sha": sha(R.hca_crypt(enc, 0, 0, KEY))}
            case["hca"].append(ent)
        man["cases"].append(case)
    # one fully stored decode pair (WAV out) so a byte diff can be inspected without the reference
    w = synth.wav(0, 3008, 2, 48000)
    put("s0_3008_2_48000_q1.decoded.wav", R.hca_decode(R.hca_encode(w, 1)))
    put("s0_3008_2_48000_bd4_bs18_m3_v4.decoded.wav", R.adx_decode(R.adx_encode(w)))
    # forged streams: v3.0 noise fill, v1.x ATH, random frames
    forged = []
    base = {q: R.hca_encode(synth.wav(5, 2500, 2, 48000), q) for q in (1, 2)}
    for q in (1, 2):
        for tag, f in (("v3min0", hca_forge.forge_v3(base[q], 0)), ("v101", hca_forge.forge_v1(base[q], 0x0101))):
            name = "forged_q%d_%s.hca" % (q, tag)
            put(name, f)
            forged.append({"file": name, "float_sha": sha(R.hca_decode_float(f).tobytes()), "decoded_sha": sha(R.hca_decode(f))})
    one = R.hca_encode(synth.wav(0, 800, 2, 48000), 2)
    kept = 0
    for v3 in (False, True):
        b = hca_forge.forge_v3(one, 0) if v3 else one
        for seed in range(40):
            f = hca_forge.random_frames(b, seed, density=1.0 if seed % 2 else 0.35)
            try:
                fl = R.hca_decode_float(f)
            except R.RefError:
                continue
            name = "fuzz_%s_%02d.hca" % ("v3" if v3 else "v2", seed)
            put(name, f)
            forged.append({"file": name, "float_sha": sha(fl.tobytes()), "decoded_sha": sha(R.hca_decode(f))})
            kept += 1
            if kept % 4 == 0:
                break
    man["forged"] = forged
    # WAV sample formats other than 16-bit (PCM::Get_PCM16 conversions): inputs are regenerated from synth.wav_typed,
    # only digests are stored
    typed = []
    for kind in ("u8", "s24", "s32", "f32", "f64"):
        for ch in (1, 2):
            w = synth.wav_typed(7, 2600, ch, 44100, kind)
            typed.append({"kind": kind, "args": [7, 2600, ch, 44100], "wav_sha": sha(w), "adx_sha": sha(R.adx_encode(w)),
                          "hca_q1_sha": sha(R.hca_encode(w, 1))})
    man["typed"] = typed
    # (sample counts are multiples of 32: the reference's ADX decoder writes whole blocks past its buffer otherwise)
    # looping WAV input ('smpl' chunk): ADX loop header, HCA loop feeding path + 'loop' chunk, and the decoders' smpl output
    loops = []
    for (seed, n, ch, sr, ls, le) in [(0, 8992, 2, 48000, 1000, 8000), (1, 4992, 1, 44100, 0, 4992), (2, 7040, 2, 22050, 2047, 2049),
                                      (3, 12000, 2, 48000, 5000, 11000), (4, 4000, 1, 48000, 3700, 4000), (5, 6016, 2, 32000, 1, 2)]:
        w = synth.wav_bytes(synth.pcm16(seed, n, ch, sr), sr, loop=(ls, le))
        ent = {"args": [seed, n, ch, sr], "loop": [ls, le], "wav_sha": sha(w), "adx": {}, "hca": {}}
        for ver in (3, 4, 5):
            a = R.adx_encode(w, 4, 18, 3, 500, 0, ver, 0)
            ent["adx"][str(ver)] = {"sha": sha(a), "decoded_sha": sha(R.adx_decode(a))}
        ent["adx_v5_noloop_sha"] = sha(R.adx_encode(w, 4, 18, 3, 500, 0, 5, 1))
        for q in (1, 3):
            h = R.hca_encode(w, q)
            ent["hca"][str(q)] = {"sha": sha(h), "decoded_sha": sha(R.hca_decode(h))}
        ent["hca_q1_noloop_sha"] = sha(R.hca_encode(w, 1, 1))
        loops.append(ent)
    man["loops"] = loops
    # AFS2 / AWB bank written by the reference's own AWBBuilder and read back by its AWB class (pure Python; its compiled
    # extension is only stubbed for the import): header fields, item ranges, and the reference's decode of every item
    import tempfile
    import types
    sys.modules.setdefault("CriCodecs", types.ModuleType("CriCodecs"))
    sys.path.insert(0, "/root/reference")
    from PyCriCodecs.awb import AWB as RefAWB, AWBBuilder as RefAWBBuilder
    subkey = 0x1234
    clips = []
    for i, (n, ch, sr, kind) in enumerate([(2016, 2, 48000, "hca"), (1600, 2, 44100, "adx"), (3008, 1, 48000, "hca"), (992, 1, 22050, "adx"),
                                           (4000, 2, 48000, "hca"), (2400, 2, 32000, "adx")]):
        w = synth.wav(60 + i, n, ch, sr)
        cb = R.hca_crypt(R.hca_encode(w, 1 + i % 3), 1, 56, KEY, subkey) if kind == "hca" else R.adx_encode(w)
        clips.append(cb + b"\0" * (-len(cb) % 0x20))          # AWBBuilder.build_files only places files right when their sizes are aligned
    with tempfile.TemporaryDirectory() as td:
        paths = []
        for i, cbytes in enumerate(clips):
            pth = os.path.join(td, "%02d.bin" % i)
            with open(pth, "wb") as f:
                f.write(cbytes)
            paths.append(pth)
        outp = os.path.join(td, "bank.awb")
        RefAWBBuilder(paths, subkey=subkey, version=2, id_intsize=2, align=0x20).build(outp)
        bank = open(outp, "rb").read()
        ref = RefAWB(outp)
        items = list(ref.getfiles())
        ref.stream.close()
    put("bank_mixed.awb", bank)
    ent = {"file": "bank_mixed.awb", "sha": sha(bank), "numfiles": ref.numfiles, "align": ref.align, "subkey": ref.subkey,
           "headersize": ref.headersize, "ofs": [int(x) for x in ref.ofs], "items": []}
    for it in items:
        is_hca = it[:4] in (b"HCA\x00", b"\xc8\xc3\xc1\x00")
        ent["items"].append({"len": len(it), "sha": sha(it), "kind": "hca" if is_hca else "adx",
                             "decoded_sha": sha(R.hca_decode(it, KEY, subkey) if is_hca else R.adx_decode(it))})
    man["awb"] = ent
    # info() dictionaries of the reference's Python HCA class (pure Python header parse; the compiled extension stays stubbed)
    from PyCriCodecs.hca import HCA as RefHCA
    wl = synth.wav_bytes(synth.pcm16(0, 8992, 2, 48000), 48000, loop=(1000, 8000))
    py_info = []
    for label, data, key in [("hca plain", R.hca_encode(synth.wav(0, 3008, 2, 48000), 1), 0),
                             ("hca encrypted", R.hca_crypt(R.hca_encode(synth.wav(0, 3008, 2, 48000), 1), 1, 56, KEY), KEY),
                             ("hca encrypted, default key", R.hca_crypt(R.hca_encode(synth.wav(1, 2048, 1, 44100), 3), 1, 56, KEY), 0),
                             ("hca looped", R.hca_encode(wl, 2), 0),
                             ("hca v3 forged", hca_forge.forge_v3(R.hca_encode(synth.wav(5, 2500, 2, 48000), 1), 0), 0),
                             ("wav", synth.wav(2, 1600, 2, 22050), 0), ("wav looped", wl, 0)]:
        info = RefHCA(data, key=key).info()
        name = "pyinfo_%02d.bin" % len(py_info)
        put(name, data)
        py_info.append({"label": label, "file": name, "key": key, "info": {k: (v if isinstance(v, (int, str, float, bool, type(None))) else repr(v)) for k, v in info.items()}})
    man["py_info"] = py_info
    # USM audio layer (usm.py): the audio mask of a key, the reference builder's @SFA chunks for an HCA stream, and the
    # reference demuxer's output for well-formed containers.  The reference's USMBuilder writes a container its own USM
    # class cannot read (a metadata chunk's size field is 16 bytes long), and its ADX branch needs an ADX object the package
    # no longer has, so the demux fixtures are assembled here: CRID chunk from the reference builder, @SFA header chunk
    # from the reference's UTFBuilder, data chunks packed with USMChunkHeader, payloads masked with the reference's own
    # AudioMask.
    import struct
    from PyCriCodecs.usm import USM as RefUSM, USMBuilder as RefUSMBuilder
    from PyCriCodecs.utf import UTFBuilder as RefUTFBuilder
    from PyCriCodecs.chunk import UTFTypeValues, USMChunkHeader

    class _Keyed(RefUSM):
        def __init__(self, key):
            self.init_key(key)
    usm = {"masks": []}
    for key in (0x0123456789ABCDEF, 0xCF222F1FE0748978, 1, 0xFFFFFFFFFFFFFFFF, "7F4551499DF55E68", "1234"):
        usm["masks"].append({"key": key, "mask": bytes(_Keyed(key).audiomask).hex()})

    def tiny_ivf(nframes=3):
        body = b""
        for i in range(nframes):
            data = (b"\x82I\x83B" if i == 0 else b"\x86\x00") + bytes((i * 7 + k) & 0xFF for k in range(100 + 13 * i))
            body += struct.pack("<IQ", len(data), i) + data
        return struct.pack("<4sHH4sHHIIII", b"DKIF", 0, 32, b"VP90", 64, 64, 30, 1, nframes, 0) + body
    hca_stream = R.hca_encode(synth.wav(5, 3000, 2, 48000), 1)
    adx_stream = R.adx_encode(synth.wav(6, 9600, 2, 48000))
    with tempfile.TemporaryDirectory() as td:
        pi, ph = os.path.join(td, "t.ivf"), os.path.join(td, "t.hca")
        open(pi, "wb").write(tiny_ivf()); open(ph, "wb").write(hca_stream)
        bld = RefUSMBuilder(pi, audio=ph, audio_codec="hca")
        bld.build()
        built = bytes(bld.get_usm())
    put("usm_ref_built_hca.usm", built)
    put("usm_audio.hca", hca_stream)
    put("usm_audio.adx", adx_stream)
    try:
        RefUSM(built).demux()
        ref_reads_own = True
    except NotImplementedError:
        ref_reads_own = False
    usm["ref_built"] = {"file": "usm_ref_built_hca.usm", "sha": sha(built), "audio": "usm_audio.hca", "reference_demux_ok": ref_reads_own}

    def chunk(sig, payload, chno, typ, frame_time=0, frame_rate=2997):
        pad = -len(payload) % 0x20
        return USMChunkHeader.pack(sig, len(payload) + 0x18 + pad, 0, 0x18, pad, chno, 0, 0, typ, frame_time, frame_rate, 0, 0) + payload + b"\0" * pad

    def audio_header(codec):
        info = [{"audio_codec": (UTFTypeValues.uchar, codec), "ixsize": (UTFTypeValues.uint, 27860),
                 "metadata_count": (UTFTypeValues.uint, 0), "metadat_size": (UTFTypeValues.uint, 0),
                 "num_channels": (UTFTypeValues.uchar, 2), "sampling_rate": (UTFTypeValues.uint, 48000),
                 "total_samples": (UTFTypeValues.uint, 9600)}]
        b = RefUTFBuilder(info, table_name="AUDIO_HDRINFO")
        b.strings = b"<NULL>\x00" + b.strings
        return chunk(b"@SFA", b.parse(), 0, 1, 0, 30)
    crid = built[:0x800]                                   # (the reference builder's video header chunk follows it, 0x800-0xa00)
    assert built[0x800:0x804] == b"@SFV" and built[0xa00:0xa04] == b"@SFA"
    usm["demux"] = []
    for name, stream, codec, key, sizes in [("usm_hca_plain.usm", hca_stream, 4, False, None), ("usm_adx_plain.usm", adx_stream, 2, False, [292, 1800, 1800, 250, 1800]),
                                           ("usm_adx_keyed.usm", adx_stream, 2, 0x0123456789ABCDEF, [292, 1800, 1800, 250, 1800]),
                                           ("usm_hca_keyed.usm", hca_stream, 4, "7F4551499DF55E68", None)]:
        masker = _Keyed(key) if key else None
        parts = [crid, built[0x800:0xa00], audio_header(codec), chunk(b"@SFV", b"#HEADER END     ===============\x00", 0, 2, 0, 30),
                 chunk(b"@SFA", b"#HEADER END     ===============\x00", 0, 2, 0, 30)]
        pos, k = 0, 0
        if codec == 4:
            hs = int.from_bytes(stream[6:8], "big"); fs = int.from_bytes(stream[28:30], "big")
            cuts = [hs] + [fs] * ((len(stream) - hs) // fs)
        else:
            cuts = list(sizes)
            while sum(cuts) < len(stream):
                cuts.append(min(1800, len(stream) - sum(cuts)))
        for n in cuts:
            pay = stream[pos:pos + n]; pos += n
            pad = -len(pay) % 0x20
            body = bytearray(pay + b"\0" * pad)
            if masker is not None and codec == 2:
                body = masker.AudioMask(body)              # XOR: masking = unmasking (extractor variant, whole 8-byte words)
            parts.append(USMChunkHeader.pack(b"@SFA", len(body) + 0x18, 0, 0x18, pad, 0, 0, 0, 0, k * 100, 2997, 0, 0) + bytes(body))
            if k % 2 == 0:
                parts.append(chunk(b"@SFV", bytes((7 * k + j) & 0xFF for j in range(0x260 + 8 * k)), 0, 0, k * 100, 3000))
            k += 1
        parts.append(chunk(b"@SFV", b"#CONTENTS END   ===============\x00", 0, 2, 0, 30))
        parts.append(chunk(b"@SFA", b"#CONTENTS END   ===============\x00", 0, 2, 0, 30))
        data = b"".join(parts)
        put(name, data)
        ru = RefUSM(data, key=key)
        ru.demux()
        out = bytes(ru.output["@SFA_0"])
        assert out == stream, name
        usm["demux"].append({"file": name, "key": key, "codec": codec, "sfa_0_sha": sha(out), "sfa_0_len": len(out), "stream": "usm_audio.hca" if codec == 4 else "usm_audio.adx"})
    man["usm"] = usm
    # generator-independent known answers
    man["known"] = {"crc16_123456789": 0xFEE8,
                    "adx_coefs": {"500,48000": [7400, -3342], "500,44100": [7334, -3283], "500,22050": [6569, -2634], "0,48000": [8192, -4096]},
                    "cipher56_first16": "001440b36c5d81f1a893cc34e5d50b72"}
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(man, f, indent=1, sort_keys=True)
    print("golden: %d wav cases, %d forged" % (len(man["cases"]), len(forged)))


if __name__ == "__main__":
    main()
