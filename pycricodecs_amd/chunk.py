"""Enums the HCA front-end shares with callers (/root/reference/PyCriCodecs/chunk.py:42-44, 68-73)."""
from enum import Enum


class HCAType(Enum):
    HCA = b"HCA\x00"
    EHCA = b"\xC8\xC3\xC1\x00"      # header magic with the encryption mask applied


class CriHcaQuality(Enum):
    Highest = 0
    High = 1
    Middle = 2
    Low = 3
    Lowest = 5                       # sic: the extension treats 5 as High
