"""Drop-in for the reference's CPython extension module `CriCodecs` (/root/reference/CriCodecs/CriCodecs.cpp:8-17):
same five codec functions, same positional arguments, same exception types and messages -- every one of them
executes on the MI355X through lib/libcricodecs_hip.so.  (CriLaylaCompress/Decompress are container compression,
outside the hot path, and are not provided.)"""
import ctypes as C

from . import _capi


def AdxDecode(data) -> bytes:
    """adx.cpp:546-558."""
    data = bytes(data)
    out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    _capi.raise_for(_capi.lib().cri_adx_decode(data, len(data), C.byref(out), C.byref(n)))
    return _capi.take(out, n)


def AdxEncode(data, bitdepth, blocksize, encoding, highpass, filter, adx_version, force_no_looping) -> bytes:
    """adx.cpp:517-544 (argument order of the "y#IIIIIIp" parse at adx.cpp:527)."""
    data = bytes(data)
    out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    _capi.raise_for(_capi.lib().cri_adx_encode(data, len(data), bitdepth, blocksize, encoding, highpass, filter, adx_version,
                                               1 if force_no_looping else 0, C.byref(out), C.byref(n)))
    return _capi.take(out, n)


def HcaDecode(data, header_size, key, subkey) -> bytes:
    """hca.cpp:3340-3457."""
    data = bytes(data)
    out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    _capi.raise_for(_capi.lib().cri_hca_decode(data, len(data), header_size, key & 0xFFFFFFFFFFFFFFFF, subkey & 0xFFFF,
                                               C.byref(out), C.byref(n)))
    return _capi.take(out, n)


def HcaEncode(data, force_nolooping, quality) -> bytes:
    """hca.cpp:3459-3489."""
    data = bytes(data)
    out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    _capi.raise_for(_capi.lib().cri_hca_encode(data, len(data), int(force_nolooping), int(quality), C.byref(out), C.byref(n)))
    return _capi.take(out, n)


def HcaCrypt(buf, crypt, header_size, type, key, subkey) -> bytes:
    """hca.cpp:3271-3337.  Works on a private copy (the reference also mutates the caller's buffer in place,
    even an immutable bytes object -- SURVEY.md 9-22 -- which is not reproduced)."""
    b = C.create_string_buffer(bytes(buf), len(buf))
    _capi.raise_for(_capi.lib().cri_hca_crypt(b, len(buf), int(crypt), int(header_size), int(type), key & 0xFFFFFFFFFFFFFFFF, subkey & 0xFFFF))
    return b.raw
