"""ADX front-end with the reference's call surface (/root/reference/PyCriCodecs/adx.py:3-14)."""
from . import CriCodecs


class ADX:
    """Pass either an `adx file` or a `wav file` in bytes to `decode` or `encode` respectively."""

    def decode(data: bytes) -> bytes:
        """Decodes ADX to WAV (adx.py:7-9)."""
        return CriCodecs.AdxDecode(bytes(data))

    def encode(data: bytes, BitDepth=0x4, Blocksize=0x12, Encoding=3, AdxVersion=0x4, Highpass_Frequency=0x1F4, Filter=0,
               force_not_looping=False) -> bytes:
        """Encodes WAV to ADX (adx.py:12-14; note the keyword order differs from the extension's positional order)."""
        return CriCodecs.AdxEncode(bytes(data), BitDepth, Blocksize, Encoding, Highpass_Frequency, Filter, AdxVersion, force_not_looping)
