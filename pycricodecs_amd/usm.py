"""The audio layer of USM containers on the device codecs (mirror of the @SFA parts of PyCriCodecs/usm.py).

`USM(data, key).demux()` returns the container's audio streams the way the reference's USM.demux() does ("@SFA_<chno>" ->
bytes), with the chunk payloads gathered -- and ADX payloads unmasked -- by one device job over the container as it sits
in HBM; `USM.decode_audio()` then decodes them with the batch ADX / HCA jobs.  `sfa_chunks()` is the other direction:
the @SFA chunk lists USMBuilder.get_data() generates for encoded audio streams (usm.py:578-716).  CRID / @UTF tables,
video (@SFV / VideoMask) and subtitles are the container's business and stay with the caller.
"""
import ctypes as C
from io import FileIO

from . import _capi
from .batch import Job

CODEC_ADX, CODEC_HCA = 2, 4


def audio_mask(key):
    """The 32-byte audio mask of a key (USM.init_key, usm.py:47-118); key: int, or hex string of up to 16 digits."""
    if isinstance(key, str):
        if len(key) > 16:
            raise ValueError("Inavild input key.")
        key = int(key.rjust(16, "0"), 16)
    elif not isinstance(key, int):
        raise ValueError("Invalid key format, must be either a string or an integer.")
    out = (C.c_uint8 * 32)()
    _capi.lib().cri_usm_audio_mask(key & 0xFFFFFFFFFFFFFFFF, out)
    return bytes(out)


def usm_index(data):
    """Chunk headers of a USM (usm.py:134-190) as a list of dicts; raises NotImplementedError like the reference."""
    buf = data if isinstance(data, bytes) else bytes(data)
    L = _capi.lib()
    n = C.c_uint32()
    rc = L.cri_usm_index(buf, len(buf), None, 0, C.byref(n))
    if rc:
        raise NotImplementedError(_capi.strerror(rc))
    arr = (_capi.UsmChunk * max(n.value, 1))()
    rc = L.cri_usm_index(buf, len(buf), arr, n.value, C.byref(n))
    if rc:
        raise NotImplementedError(_capi.strerror(rc))
    return [dict(fourcc=bytes(c.fourcc), chno=c.chno, type=c.type, padding=c.padding, payload_offset=c.payload_offset,
                 payload_len=c.payload_len, frame_time=c.frame_time, frame_rate=c.frame_rate) for c in arr[:n.value]]


def _key_int(key):
    if isinstance(key, str):
        return int(key.rjust(16, "0"), 16)
    return int(key)


class USM:
    """USM(filename or bytes, key=False): audio demux / decode.  `decrypt` is set when a key is given (usm.py:35-45)."""

    def __init__(self, filename, key=False):
        if isinstance(filename, str):
            with FileIO(filename) as f:
                self.data = f.readall()
        else:
            self.data = bytes(filename)
        if self.data[:4] != b"CRID":
            raise NotImplementedError("Unsupported file type: %r" % self.data[:4])
        self.decrypt = bool(key)
        self.key = _key_int(key) if key else 0
        self.demuxed = False
        self.output = {}
        self.codecs = {}

    def demux(self):
        """output["@SFA_<chno>"] = the channel's audio stream (usm.py:134-190, 263-277, 313-322)."""
        job = Job.usm_audio_demux(self.data, self.key, self.decrypt)
        outs, _status = job.run_host()
        self.output, self.codecs = {}, {}
        for tag, o in zip(job.item_tags if job.item_tags is not None else [], outs):
            name = "@SFA_%d" % (int(tag) & 0xFFFF)
            self.output[name] = bytearray(o)
            self.codecs[name] = int(tag) >> 16
        self.demuxed = True
        return self.output

    def decode_audio(self, hca_key=0):
        """WAV bytes per audio stream: one batch job per codec over the demuxed streams."""
        if not self.demuxed:
            self.demux()
        names = list(self.output)
        res = {}
        for codec, make in ((CODEC_ADX, lambda it: Job.adx_decode(it)), (CODEC_HCA, lambda it: Job.hca_decode(it, keys=[hca_key] * len(it)))):
            sel = [n for n in names if self.codecs[n] == codec]
            if not sel:
                continue
            job = make([bytes(self.output[n]) for n in sel])
            outs, status = job.run_host()
            for n, o, st, hst in zip(sel, outs, status, job.host_status):
                if st != 0 or hst != 0:
                    _capi.raise_for(int(hst) if hst else int(st))
                res[n] = bytes(o)
        return res


def sfa_chunks(streams, audio_codec="adx", key=0, encrypt_audio=False):
    """SFA_chunks of USMBuilder.get_data() (usm.py:578-716): per stream, the list of its @SFA chunks (the "#CONTENTS END"
    chunk is part of the last element, as in the reference)."""
    codec = {"adx": CODEC_ADX, "hca": CODEC_HCA}[audio_codec.lower()]
    if encrypt_audio and not key:
        raise ValueError("Cannot encrypt Audio without key.")
    job = Job.sfa_pack([bytes(s) for s in streams], codec, _key_int(key) if key else 0, encrypt_audio)
    outs, status = job.run_host()
    res = []
    for o, hst in zip(outs, job.host_status):
        if hst != 0:
            _capi.raise_for(int(hst))
        o = bytes(o)
        chunks, pos = [], 0
        while pos + 0x20 <= len(o):
            size = int.from_bytes(o[pos + 4:pos + 8], "big") + 8
            if o[pos + 15] == 2 and chunks:
                chunks[-1] += o[pos:pos + size]
            else:
                chunks.append(o[pos:pos + size])
            pos += size
        res.append(chunks)
    return res
