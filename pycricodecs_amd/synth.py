"""Seeded synthetic PCM/WAV generator used by tests, golden-vector generation and bench.py.

G(seed, n, ch, sr) is the generator SURVEY.md section 8(d) defines: per channel c
    0.4*sin(2*pi*f*t) + 0.2*sin(2*pi*3.1*f*t + c) + 0.05*N(0,1),  f = 220*(c+1) + 30*(seed mod 64) Hz
times a quadratic fade-in over the first 512 samples, rounded and clipped to int16.
"""
import struct
import numpy as np


def pcm16(seed: int, n: int, ch: int = 2, sr: int = 48000) -> np.ndarray:
    """Return int16 array of shape (n, ch) (interleaved when flattened)."""
    rng = np.random.default_rng(seed)
    t = np.arange(n, dtype=np.float64) / sr
    out = np.empty((n, ch), dtype=np.float64)
    for c in range(ch):
        f = 220.0 * (c + 1) + 30.0 * (seed % 64)
        out[:, c] = (0.4 * np.sin(2 * np.pi * f * t) + 0.2 * np.sin(2 * np.pi * 3.1 * f * t + c)
                     + 0.05 * rng.standard_normal(n))
    fade = np.ones(n)
    m = min(512, n)
    fade[:m] = (np.arange(m) / 512.0) ** 2
    out *= fade[:, None]
    return np.clip(np.round(out * 32767.0), -32768, 32767).astype(np.int16)


def wav_bytes(pcm: np.ndarray, sr: int = 48000, loop=None) -> bytes:
    """Canonical 44-byte-header RIFF/WAVE (fmt size 16) around an (n, ch) int16 array.
    loop=(start, end) inserts a one-loop 'smpl' chunk between fmt and data."""
    pcm = np.ascontiguousarray(pcm, dtype="<i2")
    n, ch = pcm.shape
    data = pcm.tobytes()
    fmt = struct.pack("<4sIHHIIHH", b"fmt ", 16, 1, ch, sr, sr * ch * 2, ch * 2, 16)
    smpl = b""
    if loop is not None:
        smpl = struct.pack("<4sI9I6I", b"smpl", 0x3C, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, loop[0], loop[1], 0, 0)
    body = b"WAVE" + fmt + smpl + struct.pack("<4sI", b"data", len(data)) + data
    return b"RIFF" + struct.pack("<I", len(body)) + body


def wav(seed: int, n: int, ch: int = 2, sr: int = 48000) -> bytes:
    return wav_bytes(pcm16(seed, n, ch, sr), sr)


def wav_typed(seed: int, n: int, ch: int, sr: int, kind: str) -> bytes:
    """Same signal as wav(), stored as 'u8' | 's24' | 's32' | 'f32' | 'f64' samples (plain fmt chunk, no extension).
    Floats deliberately overshoot +-1.0 a little so that the converter's clamp is exercised."""
    x = pcm16(seed, n, ch, sr).astype(np.float64)
    if kind == "u8":
        data, bits, ss, mode = ((x / 256.0).round().clip(-128, 127) + 128).astype(np.uint8).tobytes(), 8, 1, 1
    elif kind == "s24":
        v = (x * 256.0 + (np.arange(x.size).reshape(x.shape) % 251)).astype(np.int32)
        b = v.astype("<i4").view(np.uint8).reshape(-1, 4)[:, :3]
        data, bits, ss, mode = np.ascontiguousarray(b).tobytes(), 24, 3, 1
    elif kind == "s32":
        v = (x * 65536.0 + (np.arange(x.size).reshape(x.shape) % 65521)).astype("<i4")
        data, bits, ss, mode = v.tobytes(), 32, 4, 1
    elif kind == "f32":
        data, bits, ss, mode = (x / 32000.0).astype("<f4").tobytes(), 32, 4, 3
    elif kind == "f64":
        data, bits, ss, mode = (x / 32000.0).astype("<f8").tobytes(), 64, 8, 3
    else:
        raise ValueError(kind)
    fmt = struct.pack("<4sIHHIIHH", b"fmt ", 16, mode, ch, sr, sr * ch * ss, ch * ss, bits)
    body = b"WAVE" + fmt + struct.pack("<4sI", b"data", len(data)) + data
    return b"RIFF" + struct.pack("<I", len(body)) + body
