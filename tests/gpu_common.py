"""Shared by the GPU parity tests (tests/test_gpu_*.py, one file per component of SURVEY.md section 8): the module under test through the
C ABI, a byte diff that says where, and the two ways a batch job is run on device buffers."""
import os

import pytest

import golden_util as G

KEY = G.KEY
MAN = G.manifest()


@pytest.fixture(scope="module")
def cc():
    from pycricodecs_amd import CriCodecs, _capi
    assert _capi.lib().cri_device_available() == 1, "no HIP device: the GPU tests must run on the HIP path"
    return CriCodecs


def diff(a, b):
    if a == b:
        return None
    n = min(len(a), len(b))
    idx = [i for i in range(n) if a[i] != b[i]][:8]
    return "len %d vs %d, first diffs at %s" % (len(a), len(b), idx)


def run_job(job, stream=None):
    """(outputs per item, status) of a job run on device buffers (CRI_TEST_HOST_RUN=1, tools/asan_gpu.sh: no torch in the process --
    through the library's own host path)."""
    if os.environ.get("CRI_TEST_HOST_RUN") == "1":
        outs, st = job.run_host()
        return [bytes(o) for o in outs], st
    import torch
    bufs = job.alloc("cuda:0")
    job.run(*bufs, stream=stream)
    torch.cuda.synchronize()
    blob = bytes(bufs[1].cpu().numpy())
    status = bufs[3].cpu().numpy()[:job.n]
    return job.split(blob), status


def run_job_floats(job, floats=False):
    """(outputs per item, status, pre-clamp floats or None): cri_job_run_floats, the validation instances of the transform kernels."""
    import torch
    bufs = job.alloc("cuda:0")
    fl = None
    if floats:
        fl = job.run_floats(*bufs)
    else:
        job.run(*bufs)
    torch.cuda.synchronize()
    blob = bytes(bufs[1].cpu().numpy())
    status = bufs[3].cpu().numpy()[:job.n]
    return job.split(blob), status, fl
