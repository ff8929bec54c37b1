#!/usr/bin/env python3
"""Pins the ADX branch of the reference's @SFA chunk generator (PyCriCodecs/usm.py:584-657).

Run in the build container only:  python tests/golden/make_golden_sfa_adx.py

The branch cannot run as shipped: it wants `stream.sfaStream / .filetype / .Blocksize / .dataOffset` (and prepare_SFA,
usm.py:1152-1166, `.SamplingRate / .channelCount`), attributes of an ADX object this version of the package no longer has
(`ADX.encode` returns bytes, usm.py:454-463).  The generator itself is intact, so it is driven here UNMODIFIED: a
USMBuilder is made from a tiny IVF, `streams` is set to stand-in objects that carry exactly those six attributes over an
ADX file the real reference encoder (oracle/_ref/criref) wrote, `prepare_SFA()` / `prepare_SFV()` / `get_data()` are the
reference's own, and `build_usm` is overridden (in a subclass) by a method that captures the `SFA_chunks` argument -- the lists the lines
584-657 produced.  Written: the ADX inputs (small) and sfa_adx.json with every chunk's size, padding, frame time and
sha256 (plain and AudioMask-ed)."""
import hashlib
import json
import os
import struct
import sys
from io import BytesIO

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference")

import ref_tool as R  # noqa: E402
from pycricodecs_amd import synth  # noqa: E402


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def tiny_ivf(nframes=3):
    body = b""
    for i in range(nframes):
        data = (b"\x82I\x83B" if i == 0 else b"\x86\x00") + bytes((i * 7 + k) & 0xFF for k in range(100 + 13 * i))
        body += struct.pack("<IQ", len(data), i) + data
    return struct.pack("<4sHH4sHHIIII", b"DKIF", 0, 32, b"VP90", 64, 64, 30, 1, nframes, 0) + body


class StandIn:
    """What usm.py:584-657 and 1152-1166 read of an ADX stream object, nothing else."""

    def __init__(self, adx):
        self.sfaStream = BytesIO(adx)
        self.filetype = "adx"
        self.Blocksize = adx[5]
        self.dataOffset = int.from_bytes(adx[2:4], "big")
        self.SamplingRate = int.from_bytes(adx[8:12], "big")
        self.channelCount = adx[7]


def reference_chunks(adx_streams, key=False, encrypt_audio=False):
    from PyCriCodecs.usm import USMBuilder
    got = {}

    class Capture(USMBuilder):                     # (the class has __slots__: the one replaced method comes in by subclassing)
        def build_usm(self, SFV_list, SFA_chunks=False, SBT_chunks=None):
            got["sfa"] = SFA_chunks
    bld = Capture(tiny_ivf(), audio=False, key=key, audio_codec="adx", encryptAudio=encrypt_audio)
    bld.streams = [StandIn(a) for a in adx_streams]
    bld.audio = True
    bld.prepare_SFA()
    bld.prepare_SFV()
    bld.get_data()
    return [[bytes(c) for c in lst] for lst in got["sfa"]]


def main():
    assert R.available(), "build oracle/_ref/criref first (make -C oracle ref)"
    cases = []
    # (seed, samples, channels, rate, key): 9600 st 48k = several whole chunks + a floor-mod chunk; 1600 st = shorter than
    # header + one chunk (the operand of usm.py:598's % is negative); mono / other rates change the chunk size rule (1164-1166)
    specs = [(6, 9600, 2, 48000, 0), (6, 9600, 2, 48000, 0x0123456789ABCDEF), (21, 1600, 2, 48000, 0), (21, 1600, 2, 48000, 0xCF222F1FE0748978),
             (22, 640, 1, 48000, 0), (23, 7000, 1, 44100, 0), (24, 5000, 2, 22050, 0x7F4551499DF55E68), (25, 12000, 2, 32000, 0)]
    files = {}
    for seed, n, ch, sr, key in specs:
        name = "sfa_adx_%d_%d_%d_%d.adx" % (seed, n, ch, sr)
        if name not in files:
            files[name] = R.adx_encode(synth.wav(seed, n, ch, sr))
            with open(os.path.join(HERE, name), "wb") as f:
                f.write(files[name])
        adx = files[name]
        (chunks,) = reference_chunks([adx], key=key if key else False, encrypt_audio=bool(key))
        cases.append({"file": name, "key": key, "adx_sha": sha(adx), "n_chunks": len(chunks),
                      "chunks": [{"len": len(c), "size_field": int.from_bytes(c[4:8], "big"), "padding": int.from_bytes(c[10:12], "big"),
                                  "frame_time": int.from_bytes(c[16:20], "big"), "sha": sha(c)} for c in chunks],
                      "all_sha": sha(b"".join(chunks))})
    # two streams in one builder: channel numbers (usm.py:606 `self.streams.index(stream)`) and per-stream chunk sizes
    two = [files["sfa_adx_6_9600_2_48000.adx"], files["sfa_adx_23_7000_1_44100.adx"]]
    lists = reference_chunks(two)
    multi = {"files": ["sfa_adx_6_9600_2_48000.adx", "sfa_adx_23_7000_1_44100.adx"], "all_sha": [sha(b"".join(l)) for l in lists],
             "n_chunks": [len(l) for l in lists]}
    with open(os.path.join(HERE, "sfa_adx.json"), "w") as f:
        json.dump({"_about": "reference usm.py:584-657 driven unmodified through stand-in stream objects (make_golden_sfa_adx.py)",
                   "cases": cases, "multi": multi}, f, indent=1, sort_keys=True)
    print("sfa_adx: %d cases, chunk counts %s" % (len(cases), [c["n_chunks"] for c in cases]))


if __name__ == "__main__":
    main()
