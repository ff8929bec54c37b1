"""Helpers that forge HCA streams the reference encoder cannot emit (v3.0 headers, random frames),
so the decoder's noise-fill / delta-intensity / escape-code branches get exercised."""
import struct
import numpy as np


def crc16(data: bytes) -> int:
    s = 0
    for b in data:
        s ^= b << 8
        for _ in range(8):
            s = ((s << 1) ^ 0x8005) & 0xFFFF if s & 0x8000 else (s << 1) & 0xFFFF
    return s


def fix_header_crc(hca: bytearray) -> None:
    hs = struct.unpack(">H", hca[6:8])[0]
    hca[hs - 2:hs] = struct.pack(">H", crc16(bytes(hca[:hs - 2])))


def forge_v3(hca: bytes, min_res: int = 0, version: int = 0x0300) -> bytes:
    """Rewrite version / min_resolution of a v2.0 stream (comp chunk at 0x18) and fix the header CRC."""
    b = bytearray(hca)
    b[4:6] = struct.pack(">H", version)
    b[0x1E] = min_res
    fix_header_crc(b)
    return bytes(b)


def forge_v1(hca: bytes, version: int = 0x0101) -> bytes:
    """v1.x header: ATH type 1 becomes the default (no ath chunk)."""
    b = bytearray(hca)
    b[4:6] = struct.pack(">H", version)
    fix_header_crc(b)
    return bytes(b)


def forge_comp(hca: bytes, track_count=None, channel_config=None, total=None, base=None, stereo=None, hfr=None) -> bytes:
    """Rewrite fields of the comp chunk (at 0x18: track count, channel config, band counts) and fix the header CRC."""
    b = bytearray(hca)
    assert bytes(x & 0x7F for x in b[0x18:0x1C]) == b"comp"
    for off, v in ((0x20, track_count), (0x21, channel_config), (0x22, total), (0x23, base), (0x24, stereo), (0x25, hfr)):
        if v is not None:
            b[off] = v
    fix_header_crc(b)
    return bytes(b)


def random_frames(hca: bytes, seed: int, density: float = 1.0) -> bytes:
    """Replace every frame payload with seeded random bytes (sync forced, CRC fixed)."""
    b = bytearray(hca)
    hs = struct.unpack(">H", b[6:8])[0]
    fs = struct.unpack(">H", b[0x1C:0x1E])[0]
    nfr = struct.unpack(">I", b[0x10:0x14])[0]
    rng = np.random.default_rng(seed)
    for f in range(nfr):
        o = hs + f * fs
        body = rng.integers(0, 256, fs - 4, dtype=np.uint8)
        if density < 1.0:
            body = np.where(rng.random(fs - 4) < density, body, 0).astype(np.uint8)
        fr = b"\xff\xff" + body.tobytes()
        b[o:o + fs] = fr + struct.pack(">H", crc16(fr))
    return bytes(b)


def accepted_random_stream(hca: bytes, seed: int, density: float, accept):
    """random_frames, with every frame that `accept` (a one-frame stream -> bool) turns down replaced by a copy of one it
    takes, so that long multi-channel streams survive the decoder's per-frame checks.  None if no frame is accepted."""
    b = bytearray(random_frames(hca, seed, density))
    hs = struct.unpack(">H", b[6:8])[0]
    fs = struct.unpack(">H", b[0x1C:0x1E])[0]
    nfr = struct.unpack(">I", b[0x10:0x14])[0]
    one = bytearray(b[:hs])
    one[0x10:0x14] = struct.pack(">I", 1)
    one[0x16:0x18] = b"\0\0"
    fix_header_crc(one)
    frames = [bytes(b[hs + f * fs:hs + (f + 1) * fs]) for f in range(nfr)]
    ok = [accept(bytes(one) + fr) for fr in frames]
    good = [fr for fr, k in zip(frames, ok) if k]
    if not good:
        return None
    for f in range(nfr):
        if not ok[f]:
            frames[f] = good[f % len(good)]
    return bytes(b[:hs]) + b"".join(frames)
