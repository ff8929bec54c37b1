"""hostwave (TEST INFRASTRUCTURE): builds the product's C-ABI library from the product's own sources for x86-64 under the workgroup
emulator -- tests/hostwave/lib/libcricodecs_hip.so and libcricodecs_hip_testing.so, the same pair of names pycricodecs_amd/lib/ holds,
so that `CRICODECS_LIB_DIR=tests/hostwave/lib` points the unchanged Python package (and through it the unchanged GPU parity tests) at
the emulated build.  The build id embedded is the tree's (pycricodecs_amd.build.source_id()): an emulated library of other sources is
refused by the binding like any stale library.

    python tests/hostwave/build.py [--force]

Nothing in the product path builds, loads or links this; `pycricodecs_amd/lib/` is never written here.
"""
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import translate as T                                            # noqa: E402
from pycricodecs_amd import build as PB                          # noqa: E402

CSRC = PB.CSRC
OUT = os.path.join(HERE, "lib")
WORK = os.path.join(HERE, "build")
CXX = os.environ.get("HOSTWAVE_CXX", "/opt/rocm/lib/llvm/bin/clang++")
FLAGS = ["-x", "c++", "-std=c++17", "-O1", "-g1", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-fPIC", "-pthread", "-fno-strict-aliasing",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result", "-Wno-unused-variable", "-Wno-unknown-attributes", "-Wno-unknown-pragmas", "-Wno-pass-failed",
         "-I" + os.path.join(HERE, "include"), "-I/opt/rocm/include"]


def _key(paths, salt):
    h = hashlib.sha256(salt.encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build(force=False, verbose=True, traffic=False):
    """traffic=True: the census build (tools/traffic_census.py) into lib_traffic/ -- the kernel files compiled with -fsanitize=thread and
    linked against hostwave.cpp's own __tsan_* hooks (no sanitizer runtime); analysis only, not used by any test of parity."""
    global OUT, WORK
    if traffic:
        OUT, WORK = os.path.join(HERE, "lib_traffic"), os.path.join(HERE, "build_traffic")
    os.makedirs(OUT, exist_ok=True)
    src_dir = os.path.join(WORK, "pkg", "csrc")
    os.makedirs(src_dir, exist_ok=True)
    os.makedirs(os.path.join(WORK, "include"), exist_ok=True)
    shutil.copy(PB.PUBLIC_HEADER, os.path.join(WORK, "include", "cricodecs_hip.h"))
    names = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".cpp")))
    for f in names:
        with open(os.path.join(CSRC, f)) as fh:
            text = T.translate(fh.read(), f)
        dst = os.path.join(src_dir, f)
        if not os.path.exists(dst) or open(dst).read() != text:
            with open(dst, "w") as fh:
                fh.write(text)
    bid = PB.source_id()
    iddef = ['-DCRI_BUILD_ID_STRING="%s%s"' % (PB.ID_MARK.decode(), bid)]
    deps = [os.path.join(src_dir, f) for f in names if f.endswith(".h")] + [os.path.join(HERE, "include", "hip", "hip_runtime.h")]

    def compile_one(src, obj, defines):
        extra = []
        if traffic:
            extra = ["-DHOSTWAVE_TRAFFIC"] + (["-fsanitize=thread"] if src.endswith(".hip") else [])
        cmd = [CXX] + FLAGS + extra + defines + ["-c", src, "-o", obj]
        key = _key([src] + deps, " ".join(cmd))
        try:
            if not force and open(obj + ".key").read() == key and os.path.exists(obj):
                return obj
        except OSError:
            pass
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        with open(obj + ".key", "w") as f:
            f.write(key)
        return obj

    jobs = [(os.path.join(HERE, "hostwave.cpp"), os.path.join(WORK, "hostwave.o"), [])]
    for s in PB.SOURCES:
        jobs.append((os.path.join(src_dir, s), os.path.join(WORK, s.rsplit(".", 1)[0] + ".o"), iddef if s == "cri_capi.cpp" else []))
    for s in PB.TESTING_SOURCES:
        jobs.append((os.path.join(src_dir, s), os.path.join(WORK, s.rsplit(".", 1)[0] + "_testing.o"), ["-DCRI_TESTING"] + (iddef if s == "cri_capi.cpp" else [])))
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(lambda j: compile_one(*j), jobs))
    rt = objs[0]
    prod = objs[1:1 + len(PB.SOURCES)]
    test = [o for o in prod if not o.endswith("cri_capi.o")] + objs[1 + len(PB.SOURCES):]
    for out, group in ((os.path.join(OUT, "libcricodecs_hip.so"), prod), (os.path.join(OUT, "libcricodecs_hip_testing.so"), test)):
        cmd = [CXX, "-shared", "-fPIC", "-pthread", rt] + group + ["-o", out + ".tmp", "-Wl,--no-undefined", "-Wl,-Bsymbolic", "-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        os.replace(out + ".tmp", out)
        assert PB.embedded_id(out) == bid
    # the drop-in CPython module (plain host code): it dlopen()s the libcricodecs_hip.so next to itself, here the emulated one
    import sysconfig
    ext = os.path.join(OUT, "CriCodecs" + sysconfig.get_config_var("EXT_SUFFIX"))
    src = os.path.join(CSRC, "pyext", "CriCodecs_ext.cpp")
    if force or not os.path.exists(ext) or os.path.getmtime(src) > os.path.getmtime(ext):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-I" + sysconfig.get_paths()["include"], src, "-o", ext + ".tmp", "-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        os.replace(ext + ".tmp", ext)
    return OUT


def build_selftest():
    """tests/hostwave/selftest.cpp (the emulator's own instructions against their closed forms) -> build/selftest"""
    os.makedirs(WORK, exist_ok=True)
    with open(os.path.join(HERE, "selftest.cpp")) as f:
        text = T.translate(f.read(), "selftest.cpp")
    src = os.path.join(WORK, "selftest_translated.cpp")
    with open(src, "w") as f:
        f.write(text)
    out = os.path.join(WORK, "selftest")
    subprocess.run([CXX] + [f for f in FLAGS if f != "-fPIC"] + [src, os.path.join(HERE, "hostwave.cpp"), "-o", out], check=True)
    return out


if __name__ == "__main__":
    if "--selftest" in sys.argv:
        sys.exit(subprocess.run([build_selftest()]).returncode)
    build(force="--force" in sys.argv, traffic="--traffic" in sys.argv)
