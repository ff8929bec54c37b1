"""Builds pycricodecs_amd/lib/libcricodecs_hip.so (the C-ABI library, include/cricodecs_hip.h) with hipcc for gfx950.

    python -m pycricodecs_amd.build [--force]

hipcc cross-compiles without a GPU.  -ffp-contract=off is load-bearing: the reference's float work is single
rounded multiplies/adds (x86-64 SSE2 build, no FMA), and bit-exact HCA output depends on not fusing them.

Build identity.  Prebuilt libraries travel to the GPU box with the tree, so "is this .so the tree's source?" must be answerable
without trusting file times: `source_id()` is a sha256 over every file of csrc/ (+ the public header and the
compiler flags), it is compiled into both libraries (`cri_build_id()`, csrc/cri_capi.cpp), and a library is stale exactly when the
id embedded in its file differs from the tree's.  tests/test_build_id.py holds both libraries to it on the CPU and on the GPU box.
"""
import hashlib
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.environ.get("CRICODECS_LIB_DIR") or os.path.join(HERE, "lib")      # (another directory for a build of another kind: tools/asan_gpu.sh)
LIB = os.path.join(LIBDIR, "libcricodecs_hip.so")
TESTING_LIB = os.path.join(LIBDIR, "libcricodecs_hip_testing.so")   # parity-test build: the same objects, cri_capi.cpp and the test kernels with -DCRI_TESTING
TESTING_SOURCES = ["cri_capi.cpp", "cri_testing.hip"]
SOURCES = ["cri_host.cpp", "cri_hca_dec.hip", "cri_hca_enc.hip", "cri_adx.hip", "cri_misc.hip", "cri_capi.cpp"]
PUBLIC_HEADER = os.path.join(HERE, "..", "include", "cricodecs_hip.h")
ARCH = os.environ.get("CRI_OFFLOAD_ARCH", "gfx950")          # (gfx950:xnack+ for the AddressSanitizer build, tools/asan_gpu.sh)
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wall", "-Wno-unused-function", "-Wno-unused-result"]
ID_MARK = b"CRI_BUILD_ID="


def _hipcc():
    return os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _extra():
    """Extra compiler flags (sanitizer builds, -DCRI_ENC_PROFILE).  The product sources hold no switch that changes results; a flag
    that looks like one of the old timing experiments (-DEXP_*: wrong samples, tools/debug/experiments/) is refused outright."""
    flags = os.environ.get("CRI_HIPCC_EXTRA", "").split()
    bad = [f for f in flags if f.startswith("-DEXP_")]
    if bad:
        raise OSError("CRI_HIPCC_EXTRA holds %s: experiment switches are not part of the product tree (tools/debug/experiments/variant.sh builds them in a scratch copy)" % " ".join(bad))
    return flags


def _link_extra():
    """Sanitizer flags of CRI_HIPCC_EXTRA go to the link as well (the device runtime of AddressSanitizer is linked there)."""
    return [f for f in _extra() if f.startswith("-fsanitize") or f == "-shared-libsan"]


def _no_undefined():
    return [] if _link_extra() else ["-Wl,--no-undefined"]           # (a sanitized library resolves its runtime at load time)


def _headers():
    """Every header a translation unit of csrc/ can see (all of csrc/*.h + the public header)."""
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [PUBLIC_HEADER]


def _digest(paths, salt):
    h = hashlib.sha256()
    h.update(salt.encode())
    for p in paths:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()


def source_id():
    """Identity of what the libraries are built from: sources, headers, flags (incl. CRI_HIPCC_EXTRA).  24 hex digits."""
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))
    files.append(os.path.join(CSRC, "pyext", "CriCodecs_ext.cpp"))
    return _digest(files + _headers(), " ".join(FLAGS + _extra()))[:24]


def embedded_id(path):
    """The id compiled into a built library, read from the file (nothing is loaded); None if there is none."""
    try:
        with open(path, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    m = re.search(re.escape(ID_MARK) + rb"([0-9a-f]{24})", blob)
    return m.group(1).decode() if m else None


def _stale():
    want = source_id()
    return embedded_id(LIB) != want or embedded_id(TESTING_LIB) != want


def _compile(src, obj, defines, verbose):
    """Compiles src -> obj unless obj was made from exactly these inputs (a hash of source + headers + command beside the object)."""
    hipcc = _hipcc()
    cmd = [hipcc] + FLAGS + defines + _extra() + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
    key = _digest([os.path.join(CSRC, src)] + _headers(), " ".join(cmd))      # (the stamp is written after the object: a build that died leaves no stamp)
    stamp = obj + ".key"
    try:
        with open(stamp) as f:
            if f.read().strip() == key and os.path.exists(obj):
                return obj
    except OSError:
        pass
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(stamp, "w") as f:
        f.write(key)
    return obj


class _Lock:
    """One builder at a time per library directory (ranks of a torchrun job and pytest-xdist workers import at the same moment): an
    flock on LIBDIR/.build.lock; whoever comes second finds the libraries fresh and builds nothing."""

    def __enter__(self):
        import fcntl
        os.makedirs(LIBDIR, exist_ok=True)
        self.fh = open(os.path.join(LIBDIR, ".build.lock"), "w")
        fcntl.flock(self.fh, fcntl.LOCK_EX)
        return self

    def __exit__(self, *exc):
        import fcntl
        fcntl.flock(self.fh, fcntl.LOCK_UN)
        self.fh.close()
        return False


def _link(cmd_head, objs, out, verbose):
    """Links into a temporary name and renames: a process that dlopen()s `out` meanwhile sees the old file or the new one, never half of one."""
    tmp = out + ".tmp.%d" % os.getpid()
    cmd = cmd_head + objs + ["-o", tmp, "-Wl,-rpath,/opt/rocm/lib"] + _no_undefined()
    if verbose:
        print(" ".join(cmd), flush=True)
    try:
        subprocess.run(cmd, check=True)
        os.replace(tmp, out)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)


def build(force=False, verbose=True):
    with _Lock():
        return _build_locked(force, verbose)


def _build_locked(force, verbose):
    if not force and not _stale():
        import sysconfig
        ext = os.path.join(LIBDIR, "CriCodecs" + sysconfig.get_config_var("EXT_SUFFIX"))
        if not os.path.exists(ext) or os.path.getmtime(os.path.join(CSRC, "pyext", "CriCodecs_ext.cpp")) > os.path.getmtime(ext):
            build_extension(verbose)
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    if force:
        for f in os.listdir(LIBDIR):
            if f.endswith(".key"):
                os.remove(os.path.join(LIBDIR, f))
    hipcc = _hipcc()
    bid = source_id()
    iddef = ['-DCRI_BUILD_ID_STRING="%s%s"' % (ID_MARK.decode(), bid)]
    objs = []
    for src in SOURCES:
        obj = os.path.join(LIBDIR, src.rsplit(".", 1)[0] + ".o")
        objs.append(_compile(src, obj, iddef if src == "cri_capi.cpp" else [], verbose))
    _link([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC"] + _link_extra(), objs, LIB, verbose)
    # the testing build: same objects, except the planner (knobs settable) and the test-only kernels
    tobjs = [o for o in objs if not o.endswith("cri_capi.o")]
    for src in TESTING_SOURCES:
        obj = os.path.join(LIBDIR, src.rsplit(".", 1)[0] + "_testing.o")
        tobjs.append(_compile(src, obj, ["-DCRI_TESTING"] + (iddef if src == "cri_capi.cpp" else []), verbose))
    _link([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC"] + _link_extra(), tobjs, TESTING_LIB, verbose)
    build_extension(verbose)
    assert embedded_id(LIB) == bid and embedded_id(TESTING_LIB) == bid, "the build id did not make it into the libraries"
    return LIB


def build_extension(verbose=True):
    """The drop-in CPython module `CriCodecs` (csrc/pyext): plain g++, it only dlopen()s the HIP library."""
    import sysconfig
    src = os.path.join(CSRC, "pyext", "CriCodecs_ext.cpp")
    out = os.path.join(LIBDIR, "CriCodecs" + sysconfig.get_config_var("EXT_SUFFIX"))
    tmp = out + ".tmp.%d" % os.getpid()
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-I" + sysconfig.get_paths()["include"], src, "-o", tmp, "-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    try:
        subprocess.run(cmd, check=True)
        os.replace(tmp, out)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return out


if __name__ == "__main__":
    if "--id" in sys.argv:
        print("tree", source_id(), "lib", embedded_id(LIB), "testing", embedded_id(TESTING_LIB))
    else:
        build(force="--force" in sys.argv)
