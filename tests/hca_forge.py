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


def forge_header(hca: bytes, version=0x0200, dec=None, vbr=None, ath=None, loop=None, ciph=0, rva=None, comm=None, pad=True,
                 frame_size=None) -> bytes:
    """Rebuild the header of a v2.0 stream in any of the chunk forms clHCA_DecodeHeader accepts (hca.cpp:628-845), frames
    untouched.  Chunk order is the reference's: HCA, fmt, comp | dec, vbr, ath, loop, ciph, rva, comm, pad.
      dec  = None keeps the comp chunk; dict(stereo_type=.., base=..) writes the v1.x "dec" chunk (hca.cpp:710-727) with
             the comp chunk's frame size / band count (base = base band count when stereo_type != 0)
      vbr  = (max_frame_size, noise_level); ath = type or None; loop = (start_frame, end_frame, start_delay, end_padding)
      ciph = type or None (no chunk); rva = float volume; comm = comment bytes; pad = pad the header to a multiple of 0x20
    """
    b = bytearray(hca)
    hs = struct.unpack(">H", b[6:8])[0]
    assert bytes(x & 0x7F for x in b[0x18:0x1C]) == b"comp"
    fmt = bytes(b[8:0x18])
    comp = bytes(b[0x18:0x28])
    fs = struct.unpack(">H", comp[4:6])[0] if frame_size is None else frame_size
    out = bytearray()
    out += b"fmt\0" + fmt[4:]
    if dec is None:
        out += b"comp" + struct.pack(">H", fs) + comp[6:]
    else:
        total = comp[10]
        base = dec.get("base", total)
        out += b"dec\0" + struct.pack(">HBBBBBB", fs, comp[6], comp[7], total - 1, base - 1, (comp[8] << 4) | (comp[9] & 0xF),
                                      dec.get("stereo_type", 0))
    if vbr is not None:
        out += b"vbr\0" + struct.pack(">HH", vbr[0], vbr[1])
    if ath is not None:
        out += b"ath\0" + struct.pack(">H", ath)
    if loop is not None:
        out += b"loop" + struct.pack(">IIHH", *loop)
    if ciph is not None:
        out += b"ciph" + struct.pack(">H", ciph)
    if rva is not None:
        out += b"rva\0" + struct.pack(">f", rva)
    if comm is not None:
        out += b"comm" + bytes([len(comm)]) + comm
    n = 8 + len(out) + 2
    if pad:
        tgt = (n + 4 + 0x1F) // 0x20 * 0x20
        out += b"pad\0" + bytes(tgt - n - 4)
        n = tgt
    head = bytearray(b"HCA\0" + struct.pack(">HH", version, n) + bytes(out) + b"\0\0")
    head[n - 2:n] = struct.pack(">H", crc16(bytes(head[:n - 2])))
    return bytes(head) + bytes(b[hs:])


def one_frame_stream(hca: bytes) -> bytes:
    """The same header with frame_count 1 and no delay/padding accounting beyond what one frame holds; first frame kept."""
    b = bytearray(hca)
    hs = struct.unpack(">H", b[6:8])[0]
    fs = 0
    for off in range(8, hs - 4):
        if bytes(x & 0x7F for x in b[off:off + 4]) in (b"comp", b"dec\0"):
            fs = struct.unpack(">H", b[off + 4:off + 6])[0]
            break
    b[0x10:0x14] = struct.pack(">I", 1)
    b[0x14:0x18] = b"\0\0\0\0"
    fix_header_crc(b)
    return bytes(b[:hs + fs])


def header_form_streams(hca_encode, wav):
    """Streams in every header form clHCA_DecodeHeader takes besides the encoder's own (hca.cpp:710-830): the v1.x `dec`
    chunk (stereo_type 0 and != 0), explicit `ath` 0 / 1, `vbr` (never acceptable: it needs frame_size 0), `rva`, `comm`.
    q2 (Middle) streams carry stereo bands and no HFR, so a `dec` chunk can describe them exactly.
    hca_encode(wav_bytes, quality) and wav(seed, n, ch, sr) are passed in (oracle or reference encoder, synth.wav)."""
    q1 = hca_encode(wav(21, 5000, 2, 48000), 1)
    q2 = hca_encode(wav(22, 7000, 2, 48000), 2)
    m1 = hca_encode(wav(23, 4000, 1, 44100), 1)
    base2 = q2[0x23]
    return {
        "v101_dec_mono_default_ath": forge_header(m1, version=0x0101, dec=dict(stereo_type=0), ath=None),
        "v101_dec_st0": forge_header(q1, version=0x0101, dec=dict(stereo_type=0), ath=None),
        "v102_dec_st1_ath0_rva_comm": forge_header(q2, version=0x0102, dec=dict(stereo_type=1, base=base2), ath=0, rva=0.75, comm=b"made by forge"),
        "v103_dec_st1_ath1_loop": forge_header(q2, version=0x0103, dec=dict(stereo_type=2, base=base2), ath=1, loop=(1, 5, 10, 20)),
        "v102_dec_st0_on_joint_frames": forge_header(q2, version=0x0102, dec=dict(stereo_type=0), ath=0),
        "v200_comp_ath1_rva": forge_header(q1, version=0x0200, ath=1, rva=2.0),
        "v300_comp_ath0_comm_nopad": forge_header(q1, version=0x0300, ath=0, comm=b"x" * 40, pad=False),
        "v200_vbr_with_frame_size": forge_header(q1, version=0x0200, vbr=(0x100, 2)),
        "v200_vbr_frame_size_0": forge_header(q1, version=0x0200, vbr=(0x100, 2), frame_size=0),
        "v101_dec_vbr_bad_max": forge_header(q1, version=0x0101, dec=dict(stereo_type=0), vbr=(8, 0), frame_size=0),
        "v200_ath2": forge_header(q1, version=0x0200, ath=2),
        "v101_dec_ath_only_44_bytes": forge_header(m1, version=0x0101, dec=dict(stereo_type=0), ath=1, ciph=None, pad=False),
    }


def frame_size_stream(hca: bytes, frame_size: int, nfr: int, seed: int) -> bytes:
    """The header of `hca` rewritten to `frame_size` / `nfr` frames of seeded random bytes (sync + CRC in place): material for
    HcaCrypt, which only walks frames -- frame sizes near 65535 take the crypt kernel that does not stage a frame in LDS."""
    h = bytearray(forge_header(hca, frame_size=frame_size))
    hs = struct.unpack(">H", h[6:8])[0]
    h[0x10:0x14] = struct.pack(">I", nfr)
    fix_header_crc(h)
    rng = np.random.default_rng(seed)
    body = b""
    for _ in range(nfr):
        fr = b"\xff\xff" + rng.integers(0, 256, frame_size - 4, dtype=np.uint8).tobytes()
        body += fr + struct.pack(">H", crc16(fr))
    return bytes(h[:hs]) + body


def forge_trim(hca: bytes, delay: int, padding: int) -> bytes:
    """Rewrite encoder_delay / encoder_padding of the fmt chunk (hca.cpp:662-687: bytes 0x14-0x17 of a stream whose fmt chunk
    follows the signature) and fix the header CRC: the decoder drops `delay` samples at the start and `padding` at the end."""
    b = bytearray(hca)
    assert bytes(x & 0x7F for x in b[8:12]) == b"fmt\0"
    b[0x14:0x18] = struct.pack(">HH", delay, padding)
    fix_header_crc(b)
    return bytes(b)
