"""Builds pycricodecs_amd/lib/libcricodecs_hip.so (the C-ABI library, include/cricodecs_hip.h) with hipcc for gfx950.

    python -m pycricodecs_amd.build [--force]

hipcc cross-compiles without a GPU.  -ffp-contract=off is load-bearing: the reference's float work is single
rounded multiplies/adds (x86-64 SSE2 build, no FMA), and bit-exact HCA output depends on not fusing them.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libcricodecs_hip.so")
TESTING_LIB = os.path.join(LIBDIR, "libcricodecs_hip_testing.so")   # parity-test build: the same objects, cri_capi.cpp and the test kernels with -DCRI_TESTING
TESTING_SOURCES = ["cri_capi.cpp", "cri_testing.hip"]
SOURCES = ["cri_host.cpp", "cri_hca_dec.hip", "cri_hca_enc.hip", "cri_adx.hip", "cri_misc.hip", "cri_capi.cpp"]
HEADERS = ["cri_host.h", "cri_kernels.h", "cri_types.h", "cri_tables.h", "cri_imdct_tables.h", "cri_device.h", "cri_dct_lane.h", "../../include/cricodecs_hip.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wall", "-Wno-unused-function", "-Wno-unused-result"]


def _stale():
    if not os.path.exists(LIB) or not os.path.exists(TESTING_LIB):
        return True
    t = min(os.path.getmtime(LIB), os.path.getmtime(TESTING_LIB))
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS + ["cri_testing.hip"])


def build(force=False, verbose=True):
    if not force and not _stale():
        import sysconfig
        ext = os.path.join(LIBDIR, "CriCodecs" + sysconfig.get_config_var("EXT_SUFFIX"))
        if not os.path.exists(ext) or os.path.getmtime(os.path.join(CSRC, "pyext", "CriCodecs_ext.cpp")) > os.path.getmtime(ext):
            build_extension(verbose)
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    for src in SOURCES:
        obj = os.path.join(LIBDIR, src.rsplit(".", 1)[0] + ".o")
        cmd = [hipcc] + FLAGS + os.environ.get("CRI_HIPCC_EXTRA", "").split() + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB, "-Wl,-rpath,/opt/rocm/lib", "-Wl,--no-undefined"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    # the testing build: same objects, except the planner (knobs settable) and the test-only kernels
    tobjs = [o for o in objs if not o.endswith("cri_capi.o")]
    for src in TESTING_SOURCES:
        obj = os.path.join(LIBDIR, src.rsplit(".", 1)[0] + "_testing.o")
        cmd = [hipcc] + FLAGS + ["-DCRI_TESTING"] + os.environ.get("CRI_HIPCC_EXTRA", "").split() + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        tobjs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + tobjs + ["-o", TESTING_LIB, "-Wl,-rpath,/opt/rocm/lib", "-Wl,--no-undefined"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    build_extension(verbose)
    return LIB


def build_extension(verbose=True):
    """The drop-in CPython module `CriCodecs` (csrc/pyext): plain g++, it only dlopen()s the HIP library."""
    import sysconfig
    src = os.path.join(CSRC, "pyext", "CriCodecs_ext.cpp")
    out = os.path.join(LIBDIR, "CriCodecs" + sysconfig.get_config_var("EXT_SUFFIX"))
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-I" + sysconfig.get_paths()["include"], src, "-o", out, "-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
