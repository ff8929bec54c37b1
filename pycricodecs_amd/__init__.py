"""pycricodecs_amd -- MI355X-native ADX / HCA encode + decode core behind PyCriCodecs' own API.

    from pycricodecs_amd import ADX, HCA, CriHcaQuality      # same call surface as PyCriCodecs' adx.py / hca.py
    from pycricodecs_amd import CriCodecs                     # same five functions as the reference extension
    from pycricodecs_amd.batch import Job                     # device-resident batches (what bench.py measures)

Container formats (CPK / USM / UTF / ACB / AWB) are out of scope.
"""
from .chunk import CriHcaQuality, HCAType  # noqa: F401


def __getattr__(name):          # lazy: importing the package must not need the GPU library
    import importlib
    if name == "ADX":
        return importlib.import_module(".adx", __name__).ADX
    if name == "HCA":
        return importlib.import_module(".hca", __name__).HCA
    if name in ("CriCodecs", "batch", "synth"):
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
