"""numpy statement of the USM audio demux rule (usm.py:134-190, 263-277, 313-322) on top of the host chunk walk
(cri_usm_index): test infrastructure shared by the reference fuzz (tests/test_oracle_vs_reference.py, where it is pinned
against the reference's USM.demux()) and the device parity test."""
import numpy as np

from pycricodecs_amd import usm


def header_codec(data, chunks):
    """audio_codec of the last @SFA header chunk (type 1) -- read here from the table's single row by its known layout:
    the first per-row column of AUDIO_HDRINFO (usm.py:948-957) is the uchar audio_codec."""
    codec = 0
    for c in chunks:
        if c["fourcc"] == b"@SFA" and c["type"] == 1 and data[c["payload_offset"]:c["payload_offset"] + 4] == b"@UTF":
            t = data[c["payload_offset"]:c["payload_offset"] + c["payload_len"]]
            rows = int.from_bytes(t[8:12], "big") + 8
            codec = t[rows]
    return codec


def demux(data, key):
    """{chno: bytes} of the @SFA type-0 payloads; key: int (0 = no decryption)."""
    chunks = usm.usm_index(data)
    codec = header_codec(data, chunks)
    mask = np.frombuffer(usm.audio_mask(key), dtype=np.uint8) if key else None
    out = {}
    for c in chunks:
        if c["fourcc"] != b"@SFA" or c["type"] != 0:
            continue
        p = np.frombuffer(data, dtype=np.uint8, count=c["payload_len"], offset=c["payload_offset"]).copy()
        if key and codec == 2 and len(p) > 0x140:
            nw = (len(p) - 0x140) // 8 * 8
            p[0x140:0x140 + nw] ^= np.resize(mask, nw)
        out.setdefault(c["chno"], bytearray()).extend(p[:max(len(p) - c["padding"], 0)].tobytes())
    return out


def mutate(base, heads, rng):
    """one random edit of a chunk header field (padding, channel, type, signature, data offset)"""
    b = bytearray(base)
    h = int(rng.choice(heads))
    field = int(rng.integers(0, 5))
    if field == 0:
        b[h + 10:h + 12] = int(rng.integers(0, 0x60)).to_bytes(2, "big")
    elif field == 1:
        b[h + 12] = int(rng.integers(0, 2))
    elif field == 2:
        b[h + 15] = int(rng.integers(0, 4))
    elif field == 3:
        b[h:h + 4] = [b"@SFA", b"@SFV", b"@CUE", b"XXXX"][int(rng.integers(0, 4))]
    else:
        b[h + 9] = 0x18 + 8 * int(rng.integers(0, 3))
    return bytes(b)
