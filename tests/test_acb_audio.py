"""pycricodecs_amd.acb.AcbAudio -- the audio side of the reference's ACB.extract (acb.py:141-154) with the bank decoded by one batch job.
CPU: names, order, payloads and the EncodeType -> extension table against the reference's own method run in this container (skipped
where /root/reference is absent); no CPU fallback for the decode.  GPU: the decoded bank against the oracle, file by file."""
import os
import struct
import sys

import numpy as np
import pytest

import oracle_lib as O
from pycricodecs_amd import synth
from pycricodecs_amd.acb import AcbAudio, get_extension

KEY = 0xCF222F1FE0748978
REF = "/root/reference"


def make_bank(subkey=0x1357, n=7):
    """(AFS2 bytes, EncodeTypes, items): HCA (encrypted with the bank's subkey) and ADX waveforms alternating, one of an unknown type."""
    items, types = [], []
    for i in range(n):
        w = synth.wav(300 + i, 900 + 700 * i, 1 + i % 2, 48000)
        if i % 3 == 2:
            items.append(O.adx_encode(w)); types.append(0)
        elif i == 3:
            items.append(b"\x01\x02\x03 not audio"); types.append(14)
        else:
            items.append(O.hca_crypt(O.hca_encode(w, 1 + i % 3), 1, 56, KEY, subkey)); types.append(2 if i % 2 else 6)
    align = 0x20
    hs0 = 16 + 2 * n + 4 * (n + 1)
    pos = hs0 + (-hs0 % align)
    offs, parts = [hs0], []
    for b in items:
        pad = b + b"\0" * (-len(b) % align)
        parts.append(pad); pos += len(pad); offs.append(pos)
    head = struct.pack("<4sBBHIHH", b"AFS2", 2, 4, 2, n, align, subkey) + (np.arange(n) & 0xFFFF).astype("<u2").tobytes() + np.array(offs, dtype="<u4").tobytes()
    return head.ljust(hs0 + (-hs0 % align), b"\0") + b"".join(parts), types, items


def test_extension_table_and_names():
    bank, types, items = make_bank()
    a = AcbAudio(bank, types)
    assert a.names(False) == ["0.hca", "1.hca", "2.adx", "3", "4.hca", "5.adx", "6.hca"]
    assert a.names(True) == ["0.wav", "1.wav", "2.adx", "3", "4.wav", "5.adx", "6.wav"]
    got = a.files(decode=False)
    assert [n for n, _ in got] == a.names(False)
    for (_, payload), item in zip(got, items):
        assert bytes(payload[:len(item)]) == item and not any(payload[len(item):])      # (the bank's alignment padding travels with the item, as in the reference)
    with pytest.raises(IndexError):
        AcbAudio(bank, types[:-1])


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference is only present in the build container")
def test_against_the_reference_method(tmp_path):
    sys.path.insert(0, REF)
    from PyCriCodecs.acb import ACB
    from PyCriCodecs.awb import AWB as RefAWB
    for t in range(0, 40):
        assert get_extension(t) == ACB.get_extension(None, t), t
    bank, types, _ = make_bank()
    ref = ACB.__new__(ACB)                                     # (no @UTF parse: the two attributes extract() reads)
    ref.payload = [{"WaveformTable": [{"EncodeType": (0, t)} for t in types]}]
    ref.awb = RefAWB(bank)
    d_ref, d_mine = tmp_path / "ref", tmp_path / "mine"
    ACB.extract(ref, decode=False, dirname=str(d_ref))
    AcbAudio(bank, types).extract(decode=False, dirname=str(d_mine))
    assert sorted(os.listdir(d_ref)) == sorted(os.listdir(d_mine)) and len(os.listdir(d_ref)) == len(types)
    for name in os.listdir(d_ref):
        assert (d_ref / name).read_bytes() == (d_mine / name).read_bytes(), name


def test_no_cpu_fallback_for_the_decode():
    from pycricodecs_amd import _capi
    if _capi.lib().cri_device_available():
        pytest.skip("a device is present")
    bank, types, _ = make_bank()
    with pytest.raises(Exception) as e:
        AcbAudio(bank, types).files(decode=True, key=KEY)
    assert "oracle" not in str(e.value).lower()


@pytest.mark.gpu
def test_decode_of_the_whole_bank_in_one_job(tmp_path):
    subkey = 0x2468
    bank, types, items = make_bank(subkey)
    a = AcbAudio(bank, types)
    got = a.files(decode=True, key=KEY)
    assert [n for n, _ in got] == a.names(True)
    for (name, payload), item, t in zip(got, items, types):
        if get_extension(t) == ".hca":
            assert bytes(payload) == O.hca_decode(item, KEY, subkey), name       # what HCA(item, key, subkey).decode() returns (acb.py:149)
        else:
            assert bytes(payload[:len(item)]) == item
    a.extract(decode=True, key=KEY, dirname=str(tmp_path / "out"))
    assert sorted(os.listdir(tmp_path / "out")) == sorted(a.names(True))
    # a wrong key: the checksum covers the enciphered bytes, so the frames pass it and are parsed as garbage -- which the reference either
    # rejects (a frame that does not parse: "incorrect key or unknown exception") or decodes to wrong samples.  The same here, key by key.
    right = {n: O.hca_decode(items[n], KEY, subkey) for n in (0, 1, 4, 6)}
    wrong_samples = refused = 0
    for k in (KEY + 1, KEY + 2, KEY ^ 0xFFFF):
        for n in (0, 1, 4, 6):
            one = AcbAudio(bank, [t if i == n else 14 for i, t in enumerate(types)])      # only waveform n is decoded
            try:
                want = O.hca_decode(items[n], k, subkey)
            except O.OracleError:
                with pytest.raises(ValueError):
                    one.files(decode=True, key=k)
                refused += 1
                continue
            got_n = one.files(decode=True, key=k)[n][1]
            assert bytes(got_n) == want != right[n], (n, hex(k))
            wrong_samples += 1
    assert wrong_samples >= 2 and refused >= 2
