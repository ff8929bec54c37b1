"""The libraries that get loaded are the tree's sources.  Prebuilt .so files travel to the GPU box with the tree;
`pycricodecs_amd/build.py: source_id()` (sha256 over csrc/, the public header and the compiler flags) is compiled into both of them
(`cri_build_id()`), and `_capi._bind` refuses -- rebuilds -- a library whose embedded id is not the tree's.  These tests hold both
libraries to that, on the CPU (the file and the loaded symbol) and on the GPU box (what the GPU tests actually ran on)."""
import ctypes as C
import os
import shutil
import subprocess
import sys

import pytest

from pycricodecs_amd import build as B


def _ids():
    from pycricodecs_amd import _capi
    prod = _capi.lib().cri_build_id().decode()
    with _capi.testing_knobs() as L:
        test = L.cri_build_id().decode()
        assert L.cri_is_testing_build() == 1
    return prod, test


def test_both_libraries_carry_the_tree_id():
    want = B.source_id()
    assert len(want) == 24 and int(want, 16) >= 0
    assert B.embedded_id(B.LIB) == want, "libcricodecs_hip.so was not built from this tree: python -m pycricodecs_amd.build"
    assert B.embedded_id(B.TESTING_LIB) == want, "libcricodecs_hip_testing.so was not built from this tree"
    assert _ids() == (want, want)
    from pycricodecs_amd import _capi
    assert not hasattr(_capi.lib(), "cri_is_testing_build") and not hasattr(_capi.lib(), "cri_test_set")


def test_every_header_of_csrc_is_part_of_the_id(tmp_path):
    """An edit to ANY file under csrc/ (the two headers with the exact-by-enumeration device functions were missing from the old
    list) changes the id -- checked on a copy of the tree's sources."""
    src = os.path.join(tmp_path, "pkg")
    shutil.copytree(os.path.dirname(B.CSRC), src, ignore=shutil.ignore_patterns("lib", "__pycache__"))
    os.makedirs(os.path.join(tmp_path, "include"))
    shutil.copy(B.PUBLIC_HEADER, os.path.join(tmp_path, "include", "cricodecs_hip.h"))
    code = "import sys; sys.path.insert(0, %r); import pkg.build as b; print(b.source_id())" % str(tmp_path)
    base = subprocess.run([sys.executable, "-c", code], check=True, capture_output=True, text=True).stdout.strip()
    assert base == B.source_id()
    names = sorted(f for f in os.listdir(os.path.join(src, "csrc")) if f.endswith((".h", ".hip", ".cpp")))
    assert "cri_adx_quant.h" in names and "cri_hca_enc_cost.h" in names
    for name in names + [os.path.join("pyext", "CriCodecs_ext.cpp")]:
        path = os.path.join(src, "csrc", name)
        with open(path, "a") as f:
            f.write("\n// touched\n")
        got = subprocess.run([sys.executable, "-c", code], check=True, capture_output=True, text=True).stdout.strip()
        assert got != base, name
        base = got


def test_a_stale_library_is_refused(tmp_path, monkeypatch):
    """A library whose embedded id is not the tree's is not loaded: with CRICODECS_NO_REBUILD=1 binding it is an error."""
    from pycricodecs_amd import _capi
    stale = os.path.join(tmp_path, "libcricodecs_hip.so")
    blob = open(B.LIB, "rb").read()
    mark = B.ID_MARK + B.source_id().encode()
    assert blob.count(mark) == 1
    open(stale, "wb").write(blob.replace(mark, B.ID_MARK + b"0" * 24))
    monkeypatch.setenv("CRICODECS_NO_REBUILD", "1")
    with pytest.raises(OSError, match="not built from this tree"):
        _capi._bind(stale)
    _capi._bind(B.LIB)                                             # the real one binds


@pytest.mark.gpu
def test_the_gpu_box_runs_the_tree_sources():
    """On the GPU box: the two libraries this process loaded carry the id of the sources that travelled with them."""
    from pycricodecs_amd import _capi
    assert _capi.lib().cri_device_available() == 1
    want = B.source_id()
    assert _ids() == (want, want)
    loaded = [l.split()[-1] for l in open("/proc/self/maps") if "libcricodecs_hip" in l]
    assert loaded and all(B.embedded_id(p) == want for p in set(loaded)), set(loaded)


def test_no_experiment_switches_in_product_sources(monkeypatch):
    """No preprocessor flag can make the product library produce wrong samples: the timing experiments of rounds 4 / 5 (-DEXP_*) are
    patches under tools/debug/experiments/ applied to scratch copies, csrc/ holds no `EXP_` text, and the build refuses such a define."""
    import glob
    hits = []
    for p in glob.glob(os.path.join(B.CSRC, "**", "*"), recursive=True):
        if os.path.isfile(p) and p.endswith((".h", ".hip", ".cpp")):
            with open(p, errors="replace") as f:
                hits += ["%s:%d" % (os.path.basename(p), i + 1) for i, line in enumerate(f) if "EXP_" in line or "HCA_ABL_" in line]
    assert hits == []
    monkeypatch.setenv("CRI_HIPCC_EXTRA", "-DEXP_FOLD_INPUT")
    with pytest.raises(OSError, match="experiment switches"):
        B.source_id()
    monkeypatch.setenv("CRI_HIPCC_EXTRA", "-DCRI_ENC_PROFILE")
    assert B.source_id() != ""


def test_experiment_patches_apply_to_the_product_sources(tmp_path):
    """tools/debug/experiments/*.patch stay in step with csrc/ (they are applied to a scratch copy on the GPU box, variant.sh)."""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    patches = sorted(glob.glob(os.path.join(root, "tools", "debug", "experiments", "*.patch")))
    assert len(patches) >= 4
    for p in patches:
        dst = os.path.join(tmp_path, os.path.basename(p)[:-6])
        shutil.copytree(B.CSRC, os.path.join(dst, "pycricodecs_amd", "csrc"))
        r = subprocess.run(["patch", "-p1", "--dry-run", "-i", p], cwd=dst, capture_output=True, text=True)
        assert r.returncode == 0, (p, r.stdout, r.stderr)
