"""ctypes binding of oracle/liboracle.so (test infrastructure only -- never imported by the product)."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        san = os.environ.get("CRI_TEST_SANITIZED") == "1"       # tests/test_sanitizers.py: the same restatement under ASan + UBSan
        path = os.path.join(ROOT, "oracle", "liboracle_san.so" if san else "liboracle.so")
        if not os.path.exists(path):
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), os.path.basename(path)], check=True)
        L = C.CDLL(path)
        u8p, szp = C.POINTER(C.c_uint8), C.POINTER(C.c_size_t)
        L.ora_adx_decode.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(u8p), szp]
        L.ora_adx_encode.argtypes = [C.c_char_p, C.c_size_t] + [C.c_uint32] * 6 + [C.c_int, C.POINTER(u8p), szp]
        L.ora_hca_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64, C.c_uint16, C.POINTER(u8p), szp]
        L.ora_hca_decode_float.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64, C.c_uint16, C.POINTER(C.POINTER(C.c_float)), szp]
        L.ora_hca_encode.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint32, C.POINTER(u8p), szp]
        L.ora_hca_crypt.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint16]
        L.ora_crc16.argtypes = [C.c_char_p, C.c_size_t]
        L.ora_crc16.restype = C.c_uint16
        L.ora_cipher_table.argtypes = [C.c_uint32, C.c_uint64, C.c_char_p]
        L.ora_adx_coefficients.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_int32)]
        L.ora_free.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


class OracleError(Exception):
    def __init__(self, code):
        super().__init__("oracle error %d" % code)
        self.code = code


def _take(rc, out, n):
    if rc:
        raise OracleError(rc)
    data = C.string_at(out, n.value)
    lib().ora_free(out)
    return data


def adx_decode(data):
    out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    return _take(lib().ora_adx_decode(bytes(data), len(data), C.byref(out), C.byref(n)), out, n)


def adx_encode(wav, bitdepth=4, blocksize=18, mode=3, highpass=500, filt=0, version=4, force_no_loop=False):
    out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    return _take(lib().ora_adx_encode(bytes(wav), len(wav), bitdepth, blocksize, mode, highpass, filt, version,
                                      int(force_no_loop), C.byref(out), C.byref(n)), out, n)


def hca_decode(data, key=0, subkey=0):
    out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    return _take(lib().ora_hca_decode(bytes(data), len(data), key, subkey, C.byref(out), C.byref(n)), out, n)


def hca_decode_float(data, key=0, subkey=0):
    import numpy as np
    out, n = C.POINTER(C.c_float)(), C.c_size_t()
    rc = lib().ora_hca_decode_float(bytes(data), len(data), key, subkey, C.byref(out), C.byref(n))
    if rc:
        raise OracleError(rc)
    arr = np.ctypeslib.as_array(out, shape=(n.value,)).copy()
    lib().ora_free(out)
    return arr


def hca_encode(wav, quality=1, force_no_loop=False):
    out, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    return _take(lib().ora_hca_encode(bytes(wav), len(wav), int(force_no_loop), quality, C.byref(out), C.byref(n)), out, n)


def hca_crypt(data, encrypt, ctype, key, subkey=0):
    buf = C.create_string_buffer(bytes(data), len(data))
    rc = lib().ora_hca_crypt(buf, len(data), int(encrypt), ctype, key, subkey)
    if rc:
        raise OracleError(rc)
    return buf.raw


def crc16(data):
    return lib().ora_crc16(bytes(data), len(data))


def cipher_table(ctype, key):
    buf = C.create_string_buffer(256)
    rc = lib().ora_cipher_table(ctype, key, buf)
    if rc:
        raise OracleError(rc)
    return buf.raw


def adx_coefficients(highpass, rate):
    c = (C.c_int32 * 2)()
    lib().ora_adx_coefficients(highpass, rate, c)
    return c[0], c[1]
