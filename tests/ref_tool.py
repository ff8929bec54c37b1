"""Runs the real reference (oracle/_ref/criref, built from /root/reference by oracle/Makefile).
Only usable where the tool has been built; tests that need it skip otherwise."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "oracle", "_ref", "criref")


def available():
    return os.path.exists(TOOL) and os.access(TOOL, os.X_OK)


class RefError(Exception):
    def __init__(self, code):
        super().__init__("criref exit %d" % code)
        self.code = code


def _run(cmd, data, *args):
    with tempfile.TemporaryDirectory() as td:
        a, b = os.path.join(td, "in"), os.path.join(td, "out")
        with open(a, "wb") as f:
            f.write(data)
        env = dict(os.environ, MALLOC_PERTURB_="255")   # zero-on-alloc heap for the reference's plain new[]
        p = subprocess.run([TOOL, cmd, a, b] + [str(x) for x in args], capture_output=True, env=env)
        if p.returncode:
            raise RefError(p.returncode)
        with open(b, "rb") as f:
            return f.read()


def adx_encode(wav, bitdepth=4, blocksize=18, mode=3, highpass=500, filt=0, version=4, force=0):
    return _run("adxenc", wav, bitdepth, blocksize, mode, highpass, filt, version, int(force))


def adx_decode(adx):
    return _run("adxdec", adx)


def hca_encode(wav, quality=1, force_noloop=0):
    return _run("hcaenc", wav, quality, int(force_noloop))


def hca_decode(hca, key=0, subkey=0):
    return _run("hcadec", hca, hex(key), subkey)


def hca_decode_float(hca, key=0, subkey=0):
    import numpy as np
    return np.frombuffer(_run("hcadecf", hca, hex(key), subkey), dtype=np.float32)


def hca_crypt(hca, encrypt, ctype, key, subkey=0):
    return _run("hcacrypt", hca, int(encrypt), ctype, hex(key), subkey)
