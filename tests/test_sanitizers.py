"""SURVEY section 5's sanitizer job for the code that reads untrusted bytes on the HOST: the product's header logic
(pycricodecs_amd/csrc/cri_host.cpp through tests/shim/host_shim.cpp) and the oracle's restatement, both rebuilt with
-fsanitize=address,undefined and driven by the existing tests -- the RIFF / ADX / HCA header walks, the 3000-mutation header fuzz, the
golden vectors, the encoder table rules -- inside a python started with libasan preloaded and PYTHONMALLOC=malloc, so that every
bytes object the tests hand to the C side is an exact-size heap block with red zones (ctypes buffers out of pymalloc arenas would hide a
short over-read).  Any report fails the run (abort_on_error; UBSan without recovery)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.timeout(900)
def test_host_logic_and_oracle_under_asan_ubsan():
    asan = _lib("libasan.so")
    if not asan:
        pytest.skip("no libasan for this gcc")
    # the two sanitized libraries are built here, by tools that are NOT running under the preloaded runtime
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle_san.so"], check=True, capture_output=True)
    csrc = os.path.join(ROOT, "pycricodecs_amd", "csrc")
    shim = os.path.join(ROOT, "tests", "shim", "libhost_shim_san.so")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-Wall", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                    os.path.join(ROOT, "tests", "shim", "host_shim.cpp"), os.path.join(csrc, "cri_host.cpp"), "-o", shim], check=True, capture_output=True)
    env = dict(os.environ, CRI_TEST_SANITIZED="1", LD_PRELOAD=asan, PYTHONMALLOC="malloc",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    p = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", "-m", "not gpu",
                        os.path.join(ROOT, "tests", "test_host_logic.py"), os.path.join(ROOT, "tests", "test_oracle_golden.py")],
                       cwd=ROOT, env=env, capture_output=True, text=True)
    out = p.stdout + p.stderr
    assert "AddressSanitizer" not in out and "runtime error" not in out, out[-4000:]
    assert p.returncode == 0, out[-4000:]
    assert " passed" in out
    # the run above really was the sanitized pair
    for f in ("tests/shim/libhost_shim_san.so", "oracle/liboracle_san.so"):
        syms = subprocess.run(["nm", "-D", os.path.join(ROOT, f)], capture_output=True, text=True).stdout
        assert "__asan_init" in syms and "__ubsan_handle" in syms, f
