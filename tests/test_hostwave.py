"""The `-m gpu` parity tests, run WITHOUT a GPU against the product's own kernel sources under a lockstep wave64 emulation (tests/hostwave/).

tests/hostwave/build.py compiles pycricodecs_amd/csrc/*.hip|cpp for x86-64 (clang, -ffp-contract=off) against a stand-in for
<hip/hip_runtime.h>: one fiber per lane, cross-lane instructions (DPP, ds_swizzle, ds_bpermute, readlane, ballots) as rendezvous of a
wave's fibers, LDS as a per-workgroup arena with a guard page behind the launch's dynamic size, device memory = host memory.  The only
rewriting of the sources is tests/hostwave/translate.py's: inline gfx950 assembly to the same single operation in C++, `__shared__`
declarations to references into the arena.  The result is the same C ABI under the same file name; CRICODECS_LIB_DIR points the unchanged
Python package at it and the unchanged tests/test_gpu_*.py compare it with the pinned oracle and the golden vectors.

What this holds the kernels to: indices, bit work, cross-lane patterns, LDS layouts and sizes, float operation order -- on every case
the GPU suite has.  What only the GPU run can show: the compiler's gfx950 code, timing, the runtime (graphs, streams).  It is test
infrastructure: nothing under pycricodecs_amd/ includes, links or loads it, and on a GPU box the same tests run on the HIP library.
"""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HW = os.path.join(ROOT, "tests", "hostwave")
CXX = os.environ.get("HOSTWAVE_CXX", "/opt/rocm/lib/llvm/bin/clang++")
sys.path.insert(0, HW)

# what needs a real device or the real runtime (stream capture, a second process on the GPU, seconds of sustained random banks)
NEEDS_DEVICE = [
    "tests/test_gpu_boundary.py::test_job_run_captured_in_a_hip_graph",
    "tests/test_gpu_boundary.py::test_a_failed_launch_is_reported_also_while_capturing",
    "tests/test_gpu_boundary.py::test_job_destroyed_with_work_in_flight",
    "tests/test_gpu_boundary.py::test_randomised_parity_soak_for_a_few_seconds",     # (run below with its own budget)
    "tests/test_gpu_multirank.py",
]


@pytest.fixture(scope="module")
def emulated_lib():
    if not os.path.exists(CXX):
        pytest.skip("no clang++ at %s" % CXX)
    r = subprocess.run([sys.executable, os.path.join(HW, "build.py")], cwd=ROOT, capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lib = os.path.join(HW, "lib")
    assert os.path.exists(os.path.join(lib, "libcricodecs_hip.so")) and os.path.exists(os.path.join(lib, "libcricodecs_hip_testing.so"))
    return lib


def _env(lib):
    env = dict(os.environ)
    # HOSTWAVE_GUARD: every device buffer a test hands to a job lies between two inaccessible pages, scratch is poisoned (tests/hostwave/mode.py)
    env.update(CRI_TEST_HOSTWAVE="1", CRICODECS_LIB_DIR=lib, CRICODECS_NO_REBUILD="1", HOSTWAVE_THREADS="4", HOSTWAVE_GUARD="1")
    return env


def test_emulator_instructions_against_their_closed_forms():
    """tests/hostwave/selftest.cpp: every DPP control the emulator models (quad permutations, row shifts / rotations / mirrors /
    broadcasts with row and bank masks and bound_ctrl), ds_swizzle's bit and quad modes, ds_bpermute, readlane, ballots, mbcnt, HIP's
    shuffles with widths, v_perm / alignbit / bfe / sad / mul24 / med3 / cvt_pk, exchanges inside a divergent branch, barriers between
    four waves with two of them leaving early, LDS carving -- and a read one byte past a launch's dynamic LDS dies on the guard page
    with a message that says so."""
    if not os.path.exists(CXX):
        pytest.skip("no clang++ at %s" % CXX)
    import build as HB
    exe = HB.build_selftest()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr
    r = subprocess.run([exe, "overrun"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "LDS byte 4096 of a launch with 4096 bytes of dynamic LDS" in r.stderr, r.stdout + r.stderr


def test_emulated_library_is_the_trees_sources(emulated_lib):
    """Same build id as the tree (the binding refuses anything else), same exported C ABI as the HIP library's header."""
    from pycricodecs_amd import build as B
    for name in ("libcricodecs_hip.so", "libcricodecs_hip_testing.so"):
        assert B.embedded_id(os.path.join(emulated_lib, name)) == B.source_id()
    with open(os.path.join(ROOT, "include", "cricodecs_hip.h")) as f:
        declared = set(re.findall(r"\b(cri_[a-z0-9_]+)\s*\(", f.read()))
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(emulated_lib, "libcricodecs_hip.so")], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\bT (cri_[a-z0-9_]+)", out))
    assert declared and declared <= exported, sorted(declared - exported)


def test_the_package_refuses_the_emulated_library_outside_the_test_suite(emulated_lib):
    """CRICODECS_LIB_DIR pointing at tests/hostwave/lib without CRI_TEST_HOSTWAVE=1 is an error, not a CPU path."""
    env = _env(emulated_lib)
    env.pop("CRI_TEST_HOSTWAVE")
    r = subprocess.run([sys.executable, "-c", "from pycricodecs_amd import _capi; _capi.lib()"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "emulated TEST build" in r.stderr, r.stderr[-1500:]


def test_gpu_parity_suite_on_the_emulated_kernels(emulated_lib):
    """Every `-m gpu` test of tests/ (test_gpu_{adx,hca_decode,hca_encode,wav,boundary,containers}.py, test_acb_audio.py, test_build_id.py)
    except the handful that need the real runtime: all pass on the emulated kernels -- the same assertions, oracle and golden vectors the
    GPU box's run uses -- with every device buffer fenced by inaccessible pages, scratch and device allocations poisoned, and LDS ending at
    a guard page: no kernel touches memory outside its buffers or depends on what memory held."""
    files = ["tests"]
    cmd = [sys.executable, "-m", "pytest"] + files + ["-m", "gpu", "-q", "-p", "no:cacheprovider", "--timeout", "900", "-n", "6"]
    for d in NEEDS_DEVICE:
        cmd += ["--deselect", d]
    r = subprocess.run(cmd, cwd=ROOT, env=_env(emulated_lib), capture_output=True, text=True, timeout=3000)
    tail = r.stdout[-4000:] + r.stderr[-2000:]
    assert r.returncode == 0, tail
    m = re.search(r"(\d+) passed", r.stdout)
    assert m and int(m.group(1)) >= 430, tail
    assert " failed" not in r.stdout and " error" not in r.stdout, tail


@pytest.mark.parametrize("order", ["reverse"])      # ("rotate" passes as well: tests/hostwave/README.md)
def test_kernels_with_several_waves_do_not_depend_on_the_order_the_waves_run_in(emulated_lib, order):
    """HOSTWAVE_ORDER: the emulator starts a workgroup with its last lane and runs the ring backwards (reverse), or starts with a wave
    drawn from the block number (rotate).  The kernels whose workgroups hold several waves -- k_hca_encode (frames x channels), the wide
    transforms (a wave per four channels), k_adx_lane_encode -- and everything else of the encoder / ADX / WAV tests give the same bytes:
    what one wave reads of another's LDS or global data is behind a barrier."""
    env = _env(emulated_lib)
    env["HOSTWAVE_ORDER"] = order
    # (the whole of test_gpu_{hca_encode,adx,wav,hca_decode}.py passes this way too -- 391 tests, a minute; here: the encoder, the lane
    #  encoder, the wide layouts and the WAV paths)
    cmd = [sys.executable, "-m", "pytest", "tests/test_gpu_hca_encode.py", "tests/test_gpu_adx.py", "tests/test_gpu_wav.py", "tests/test_gpu_hca_decode.py", "-k",
           "(hca_encode or lane or multichannel or wide or nine_to_sixteen or typed or ten_second) and not fuzz and not exhaustive and not band_cost",
           "-m", "gpu", "-q", "-p", "no:cacheprovider", "--timeout", "900", "-n", "6"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=3000)
    assert r.returncode == 0 and " failed" not in r.stdout, r.stdout[-3000:] + r.stderr[-1000:]
    m = re.search(r"(\d+) passed", r.stdout)
    assert m and int(m.group(1)) >= 40, r.stdout[-1500:]


def test_randomised_parity_soak_on_the_emulated_kernels(emulated_lib):
    """tools/parity_soak.py (random banks of WAVs through every batch job and the single-file calls, every output against the oracle,
    corrupted and forged streams among them) for twenty seconds on the emulated kernels: no mismatch."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "parity_soak.py"), "20", "606", "0"], cwd=ROOT, env=_env(emulated_lib), capture_output=True, text=True, timeout=1200)
    tail = "\n".join(r.stdout.splitlines()[-12:]) + r.stderr[-2000:]
    assert r.returncode == 0, tail
    m = re.search(r"(\d+) comparisons.*?(\d+) mismatch", r.stdout.splitlines()[-1] if r.stdout else "")
    assert "mismatches 0" in r.stdout or (m and int(m.group(2)) == 0), tail


def _bench(lib, tmp_path, argv, timeout=1500):
    import json
    env = _env(lib)
    env["BENCH_DETAIL_DIR"] = str(tmp_path)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bench_dry_run.py")] + argv, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [x for x in r.stdout.split("\n") if x.strip().startswith("{")]
    assert len(lines) == 1 and len(lines[0]) <= 4096, r.stdout[-1500:]
    with open(os.path.join(str(tmp_path), "bench_detail.json")) as f:
        return json.loads(lines[0]), json.load(f)


def test_bench_default_run_on_the_emulated_kernels(emulated_lib, tmp_path):
    """bench.py's default invocation at toy sizes with the REAL batch.Job over the emulated library (tests/bench_dry_run.py under
    CRI_TEST_HOSTWAVE=1; tests/test_bench_dry_run.py runs the same flow on an oracle-backed double of the Job): the headline with its
    event read-out and record census, the sustained loop, every secondary at both sizes -- the library's own host paths, the USM audio
    layer and the five single-file calls among them --, the other BASELINE configurations, the reference on the host's cores; every
    output of every job verified against the oracle; one line of <= 4 KB."""
    line, detail = _bench(emulated_lib, tmp_path, ["--streams", "6", "--unique", "2", "--seconds", "0.3", "--steps", "2", "--warmup", "1", "--secondary-streams", "4",
                                                    "--awb-clips", "24", "--awb-durations", "6", "--config-awb-clips", "24", "--config-items-scale", "0.0005",
                                                    "--config-seconds-scale", "0.02", "--cpu-seconds", "0.4", "--config-cpu-seconds", "0.4", "--sustain", "0.2"])
    assert line["value"] > 0 and line["config"]["verified"]["items"] == 6 and line["cpu_baseline"]["value"] > 0
    r = line["roofline"]
    assert set(r["kernel_ms_per_step"]) == {"k_hca_parse", "k_hca_transform"} and r["dominant_kernel"]["name"] in r["kernel_ms_per_step"]
    assert r["frac"] == r["frac_end_to_end"] and r["algorithmic_bytes_per_launch"] == 6 * 15 * (682 + 4096)
    sec = detail["secondary"]
    forms = {"hca_decode_middle": [3], "hca_decode_6ch": [10], "hca_decode_v3_noise_fill": [4], "hca_decode_6ch_v3_noise_fill": [12], "hca_decode_6ch_middle": [11], "hca_decode_sparse_spectra": [2]}
    for k, f in forms.items():                                 # the transform instance each secondary takes (cri_job_hca_groups), verified items at both sizes
        assert sec[k]["transform_forms"] == f and sec[k]["verified_items"] > 0 and sec[k + "_full"]["verified_items"] > 0, k
    assert sec["hca_decode_sparse_spectra"]["record_forms"].startswith("0 of 60 frames crossed scratch as int8")      # sparse material: int16 lines
    assert sec["hca_decode_host"]["verified_items"] > 0 and sec["adx_decode_host"]["verified_items"] > 0 and sec["usm_demux"]["verified_items"] > 0
    assert sec["hca_encode"]["verified_items"] > 0 and sec["adx_roundtrip"]["verified_items"] > 0 and sec["awb_mixed_decode"]["verified"]["items"] == 24
    assert all(sec["single_call_ms"][k][0] > 0 for k in ("AdxDecode", "AdxEncode", "HcaDecode", "HcaEncode", "HcaCrypt"))
    cfgs = sec["baseline_configs"]
    assert set(k.split(" ")[0] for k in cfgs) == {"configs[1]", "configs[3]", "configs[4]"} and all(v["verified"]["items"] > 0 for v in cfgs.values())


@pytest.mark.parametrize("workload", ["hca_encode", "adx_roundtrip"])
def test_bench_workloads_on_the_emulated_kernels(emulated_lib, tmp_path, workload):
    line, _ = _bench(emulated_lib, tmp_path, ["--workload", workload, "--no-cpu", "--no-secondary", "--streams", "6", "--unique", "2", "--seconds", "0.3", "--steps", "1", "--warmup", "0"])
    assert line["config"]["verified"]["items"] == (12 if workload == "adx_roundtrip" else 6) and line["value"] > 0


def test_bench_two_rank_launcher_on_the_emulated_kernels(emulated_lib, tmp_path):
    """bench.py --gpus 2 --workload awb_mixed --scaling strong: two ranks (gloo), each decoding its share of the bank on the emulated
    kernels, gather_bytes_to_root, and the root's check of EVERY gathered item of both ranks against the oracle."""
    line, _ = _bench(emulated_lib, tmp_path, ["--gpus", "2", "--workload", "awb_mixed", "--scaling", "strong", "--awb-clips", "40", "--awb-durations", "12", "--steps", "1", "--warmup", "1", "--no-cpu"])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["config"]["gathered_items_verified_on_root"] == 40 and line["config"]["gathered_bytes_on_root"] > 0


def test_translator_knows_every_instruction_it_meets_and_refuses_the_rest():
    """tests/hostwave/translate.py: the inline assembly of the sources maps to single C++ operations with the modifiers decoded
    (op_sel / neg_lo / neg_hi of the packed instructions, the DPP controls); an instruction it does not know stops the build."""
    import translate as T
    t = lambda s: T.translate_line(s, "x:1")
    assert t('asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));') == "r = hw::v_pk<'+'>(a, b, 0, 3, 0, 2);"
    assert t('asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(q) : "v"(u), "v"(tw));') == "q = hw::v_pk<'*'>(u, tw, 3, 1, 0, 0);"
    assert t('asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(p));') == "r = hw::v_pk<'+'>(p, p, 2, 2, 0, 2);"
    assert "0x128" in t('asm("v_mul_f32_dpp %0, %1, -%2 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(r) : "v"(v), "v"(c));') and "(-(c))" in \
        t('asm("v_mul_f32_dpp %0, %1, -%2 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(r) : "v"(v), "v"(c));')
    assert t('x = 1; asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(s) : "v"(a), "v"(f(b, c)), "v"(p >> 12)); y = 2;') == "x = 1; s = hw::v_mad_i32_i24(a, f(b, c), p >> 12); y = 2;"
    assert t('if (q) asm volatile("ds_add_u32 %0, %1\\n\\ts_waitcnt lgkmcnt(0)" :: "v"(lds_address(&s[i])), "v"(raw) : "memory");') == "if (q) hw::ds_add_u32(lds_address(&s[i]), raw);"
    assert t('asm volatile("" : "+v"(lane));') == "((void)0);"
    assert t('// asm("v_nop") in a comment stays') == '// asm("v_nop") in a comment stays'
    assert t("    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];") == "    uint8_t* const smem = (uint8_t*)hw::dyn_lds();"
    assert "hw::static_lds(sizeof(xl_lds_t), 16" in t("    __shared__ __attribute__((aligned(16))) int32_t xl[64];")
    with pytest.raises(SystemExit):
        t('asm("v_dot2_f32_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));')
    # and every source of the product translates line for line
    csrc = os.path.join(ROOT, "pycricodecs_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h", ".cpp")):
            with open(os.path.join(csrc, f)) as fh:
                src = fh.read()
            out = T.translate(src, f)
            assert out.count("\n") == src.count("\n"), f
            code = "\n".join(l[:T._comment_start(l)] for l in out.split("\n"))
            assert not re.search(r"\basm\s*(volatile\s*)?\(", code) and "__shared__" not in code, f
