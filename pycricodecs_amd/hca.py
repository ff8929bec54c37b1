"""HCA front-end with the reference's call surface (/root/reference/PyCriCodecs/hca.py:20-308).

Own implementation of the Python-side header walk (`info()` keys, key/subkey validation, default key, WAV
constraints); every codec call goes to the device through CriCodecs.Hca*.
"""
import struct
from io import BytesIO, FileIO

from . import CriCodecs
from .chunk import CriHcaQuality, HCAType

DEFAULT_KEY = 0xCF222F1FE0748978       # hca.py:91-92


class HCA:
    def __init__(self, stream, key: int = 0, subkey: int = 0) -> None:
        if isinstance(stream, str):
            with FileIO(stream) as f:
                stream = f.read()
        self._data = bytes(bytearray(stream))
        self.key = int(key, 16) if isinstance(key, str) else key
        self.subkey = int(subkey, 16) if isinstance(subkey, str) else subkey
        self.hcabytes = b""
        self.wavbytes = b""
        self.looping = False
        self.encrypted = False
        self._hca_stream = self._data
        self._parse(self._data)

    # -- header walk (hca.py:78-236)
    def _parse(self, data: bytes) -> None:
        if len(data) < 8:
            raise ValueError("Invalid HCA or WAV file.")
        self.HcaSig, self.version, self.header_size = struct.unpack(">4sHH", data[:8])
        if self.HcaSig in (HCAType.HCA.value, HCAType.EHCA.value):
            if not self.hcabytes:
                self.filetype = "hca"
            self.encrypted = self.HcaSig == HCAType.EHCA.value
            if self.encrypted and not self.key:
                self.key = DEFAULT_KEY
            elif self.key < 0:
                raise ValueError("HCA key cannot be a negative.")
            elif self.key > 0xFFFFFFFFFFFFFFFF:
                raise OverflowError("HCA key cannot exceed the maximum size of 8 bytes.")
            elif self.subkey < 0:
                raise ValueError("HCA subkey cannot be a negative.")
            elif self.subkey > 0xFFFF:
                raise OverflowError("HCA subkey cannot exceed 65535.")
            pos = 8
            fmtsig, temp, framecount, delay, padding = struct.unpack(">4sIIHH", data[pos:pos + 16])
            pos += 16
            self.hca = dict(Encrypted=self.encrypted, Header=self.HcaSig, version=hex(self.version), HeaderSize=self.header_size,
                            FmtSig=fmtsig, ChannelCount=temp >> 24, SampleRate=temp & 0xFFFFFF, FrameCount=framecount,
                            EncoderDelay=delay, EncoderPadding=padding)
            while True:
                sig = bytes(b & 0x7F for b in data[pos:pos + 4])
                if sig == b"comp":
                    v = struct.unpack(">4sHBBBBBBBBBB", data[pos:pos + 16]); pos += 16
                    self.hca.update(CompSig=v[0], FrameSize=v[1], MinResolution=v[2], MaxResolution=v[3], TrackCount=v[4],
                                    ChannelConfig=v[5], TotalBandCount=v[6], BaseBandCount=v[7], StereoBandCount=v[8],
                                    BandsPerHfrGroup=v[9], ReservedByte1=v[10], ReservedByte2=v[11])
                elif sig == b"ciph":
                    v = struct.unpack(">4sH", data[pos:pos + 6]); pos += 6
                    if v[1] == 1:
                        self.encrypted = True
                    self.hca.update(CiphSig=v[0], CipherType=v[1])
                elif sig == b"loop":
                    self.looping = True
                    v = struct.unpack(">4sIIHH", data[pos:pos + 16]); pos += 16
                    self.hca.update(LoopSig=v[0], LoopStart=v[1], LoopEnd=v[2], LoopStartDelay=v[3], LoopEndPadding=v[4])
                elif sig == b"dec\x00":
                    v = struct.unpack(">4sHBBBBBB", data[pos:pos + 12]); pos += 12
                    self.hca.update(DecSig=v[0], FrameSize=v[1], MinResolution=v[3], MaxResolution=v[2], TotalBandCount=v[4],
                                    BaseBandCoung=v[5], TrackCount=v[6] >> 4, ChannelConfig=v[6] & 0xF, StereoType=v[7])
                elif sig == b"ath\x00":
                    v = struct.unpack(">4sH", data[pos:pos + 6]); pos += 6
                    self.hca.update(AthSig=v[0], TableType=v[1])
                elif sig == b"vbr\x00":
                    v = struct.unpack(">4sHH", data[pos:pos + 8]); pos += 8
                    self.hca.update(VbrSig=v[0], MaxFrameSize=v[1], NoiseLevel=v[2])
                elif sig == b"rva\x00":
                    v = struct.unpack(">4sf", data[pos:pos + 8]); pos += 8
                    self.hca.update(RvaSig=v[0], Volume=v[1])
                else:
                    break
            self.hca.update(Crc16=data[pos:pos + 2])
        elif self.HcaSig == b"RIFF":
            self.filetype = "wav"
            (self.riffSignature, self.riffSize, self.wave, self.fmt, self.fmtSize, self.fmtType, self.fmtChannelCount,
             self.fmtSamplingRate, self.fmtSamplesPerSec, self.fmtSamplingSize, self.fmtBitCount) = struct.unpack("<4sI4s4sIHHIIHH", data[:36])
            if self.riffSignature == b"RIFF" and self.wave == b"WAVE" and self.fmt == b"fmt ":
                if self.fmtBitCount != 16:
                    raise ValueError(f"WAV bitdepth of {self.fmtBitCount} is not supported, only 16 bit WAV files are supported.")
                elif self.fmtSize != 16:
                    raise ValueError(f"WAV file has an FMT chunk of an unsupported size: {self.fmtSize}, the only supported size is 16.")
                pos = 36
                if data[pos:pos + 4] == b"smpl":
                    self.looping = True
                    v = struct.unpack("<4sIIIIIIIIIIIIIIII", data[pos:pos + 68])
                    self.LoopCount, self.LoopStartSample, self.LoopEndSample = v[9], v[13], v[14]
                    if self.LoopCount != 1:
                        self.looping = False
                    pos += 8 + v[1]
                else:
                    self.looping = False
                if data[pos:pos + 4] == b"note":
                    pos += 8 + struct.unpack("<I", data[pos + 4:pos + 8])[0]
                if data[pos:pos + 4] == b"data":
                    self.dataSig, self.dataSize = struct.unpack("<4sI", data[pos:pos + 8])
                else:
                    raise ValueError("Invalid or an unsupported wav file.")
        else:
            raise ValueError("Invalid HCA or WAV file.")

    def info(self) -> dict:
        """Returns info related to the input file (hca.py:238-244)."""
        if self.filetype == "hca":
            return self.hca
        return dict(RiffSignature=self.riffSignature.decode(), riffSize=self.riffSize, WaveSignature=self.wave.decode(),
                    fmtSignature=self.fmt.decode(), fmtSize=self.fmtSize, fmtType=self.fmtType, fmtChannelCount=self.fmtChannelCount,
                    fmtSamplingRate=self.fmtSamplingRate, fmtSamplesPerSec=self.fmtSamplesPerSec, fmtSamplingSize=self.fmtSamplingSize,
                    fmtBitCount=self.fmtBitCount, dataSignature=self.dataSig.decode(), dataSize=self.dataSize)

    def decode(self) -> bytes:
        """hca.py:246-253."""
        if self.filetype == "wav":
            raise ValueError("Input type for decoding must be an HCA file.")
        self.wavbytes = CriCodecs.HcaDecode(self._hca_stream, self.header_size, self.key, self.subkey)
        return bytes(self.wavbytes)

    def encode(self, force_not_looping: bool = False, encrypt: bool = False, keyless: bool = False,
               quality_level: CriHcaQuality = CriHcaQuality.High) -> bytes:
        """hca.py:255-274."""
        if self.filetype == "hca":
            raise ValueError("Input type for encoding must be a WAV file.")
        if force_not_looping is False or force_not_looping == 0:
            force = 0
        elif force_not_looping is True or force_not_looping == 1:
            force = 1
        else:
            raise ValueError("Forcing the encoder to not loop is by either False or True.")
        if quality_level not in list(CriHcaQuality):
            raise ValueError("Chosen quality level is not valid or is not the appropiate enumeration value.")
        self.hcabytes = CriCodecs.HcaEncode(self._data, force, quality_level.value)
        self._hca_stream = self.hcabytes
        self._parse(self.hcabytes)
        if encrypt:
            if self.key == 0 and not keyless:
                self.key = DEFAULT_KEY
            self.encrypt(self.key, keyless)      # sic: `keyless` lands in the subkey parameter (hca.py:273)
        return self.get_hca()

    def encrypt(self, keycode: int, subkey: int = 0, keyless: bool = False) -> None:
        """hca.py:276-281."""
        if self.encrypted:
            raise ValueError("HCA is already encrypted.")
        self.encrypted = True
        self._hca_stream = CriCodecs.HcaCrypt(self.get_hca(), 1, self.header_size, (1 if keyless else 56), keycode, int(subkey))

    def decrypt(self, keycode: int, subkey: int = 0) -> None:
        """hca.py:283-288."""
        if not self.encrypted:
            raise ValueError("HCA is already decrypted.")
        self.encrypted = False
        self._hca_stream = CriCodecs.HcaCrypt(self.get_hca(), 0, self.header_size, 0, keycode, int(subkey))

    def get_hca(self) -> bytes:
        """The HCA file bytes after encoding / encrypting / decrypting (hca.py:290-295)."""
        return bytes(self._hca_stream)

    def get_frames(self):
        """Yields (frame number, frame bytes) (hca.py:297-301)."""
        data = self._hca_stream
        fs = self.hca["FrameSize"]
        for i in range(self.hca["FrameCount"]):
            o = self.header_size + i * fs
            yield (i, data[o:o + fs])

    def get_header(self) -> bytes:
        """hca.py:303-308."""
        return bytes(self._hca_stream[:self.header_size])
