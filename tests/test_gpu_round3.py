"""GPU parity, round-3 additions (through the C ABI, against the pinned oracle and reference-generated vectors): the USM
builder's ADX @SFA branch pinned to the reference generator, the kernel instances round 2 left without a test (cipher
tables in global memory, the oversized-frame crypt kernel), the segmented ADX kernels and the pipelined host path."""
import json
import os

import numpy as np
import pytest

import golden_util as G
import oracle_lib as O
from pycricodecs_amd import synth

pytestmark = pytest.mark.gpu
KEY = G.KEY
MAN = G.manifest()
with open(os.path.join(G.GOLDEN, "sfa_adx.json")) as _f:
    SFA_ADX = json.load(_f)


@pytest.fixture(scope="module")
def cc():
    from pycricodecs_amd import CriCodecs, _capi
    assert _capi.lib().cri_device_available() == 1, "no HIP device: the GPU tests must run on the HIP path"
    return CriCodecs


def run_job(job):
    import torch
    bufs = job.alloc("cuda:0")
    job.run(*bufs)
    torch.cuda.synchronize()
    blob = bytes(bufs[1].cpu().numpy())
    status = bufs[3].cpu().numpy()[:job.n]
    return job.split(blob), status


# ------------------------------------------------------------------------------------------------ f3: USM builder, ADX branch
@pytest.mark.parametrize("c", SFA_ADX["cases"], ids=lambda c: "%s-key%x" % (c["file"], c["key"]))
def test_sfa_adx_chunks_match_reference_generator(cc, c):
    """usm.py:584-657 (the ADX branch of the @SFA generator) run UNMODIFIED in the build container over stand-in stream
    objects (tests/golden/make_golden_sfa_adx.py): every chunk -- header, size / padding / frame time fields, payload,
    AudioMask-ed bytes, the stream's tail block and the "#CONTENTS END" chunk glued to it -- byte for byte.  Includes
    streams shorter than one chunk (the floor modulo of usm.py:598 on a negative operand)."""
    from pycricodecs_amd import usm
    adx = G.load(c["file"])
    assert G.sha(adx) == c["adx_sha"]
    (chunks,) = usm.sfa_chunks([adx], "adx", key=c["key"], encrypt_audio=bool(c["key"]))
    assert len(chunks) == c["n_chunks"]
    for k, (got, want) in enumerate(zip(chunks, c["chunks"])):
        assert len(got) == want["len"] and int.from_bytes(got[4:8], "big") == want["size_field"], k
        assert int.from_bytes(got[10:12], "big") == want["padding"] and int.from_bytes(got[16:20], "big") == want["frame_time"], k
        assert G.sha(got) == want["sha"], k
    assert G.sha(b"".join(chunks)) == c["all_sha"]


def test_sfa_adx_two_streams_match_reference_generator(cc):
    """Two ADX streams in one builder: the channel number of a chunk is the stream's index (usm.py:606), chunk sizes are per
    stream (usm.py:1164-1166)."""
    from pycricodecs_amd import usm
    m = SFA_ADX["multi"]
    lists = usm.sfa_chunks([G.load(f) for f in m["files"]], "adx")
    assert [len(l) for l in lists] == m["n_chunks"]
    assert [G.sha(b"".join(l)) for l in lists] == m["all_sha"]
