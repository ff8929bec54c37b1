"""bench.py end to end on the CPU with a test double for the device (tests/fake_device.py: the oracle behind batch.Job's interface):
every workload, the default run with its secondaries at both sizes and the other BASELINE configurations, and the two-rank launcher
with the gather onto rank 0.  What is checked is the script -- control flow, verification, the one-line contract, the detail file --
not a number: the pool's GPUs were closed to the build for most of round 6, and a bench that only ever ran in the builder's head is how
round 5 lost its record."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import fake_device  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "detail")


def run_main(monkeypatch, capsys, tmp_path, argv):
    fake_device.install(bench, monkeypatch)
    monkeypatch.setattr(bench, "DETAIL_ROOT", str(tmp_path))
    monkeypatch.setenv("BENCH_DETAIL_DIR", str(tmp_path))
    monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    bench.main()
    out = capsys.readouterr().out
    lines = [x for x in out.split("\n") if x.strip()]
    assert len(lines) == 1, out[-2000:]
    assert len(lines[0]) <= bench.LINE_LIMIT
    line = json.loads(lines[0])
    for k in CONTRACT:
        assert k in line, k
    with open(tmp_path / bench.DETAIL_NAME) as fh:
        detail = json.load(fh)
    assert detail["value"] == line["value"]
    return line, detail


SMALL = ["--streams", "6", "--unique", "2", "--seconds", "0.3", "--steps", "1", "--warmup", "0"]


@pytest.mark.parametrize("workload", ["hca_decode", "hca_encode", "adx_roundtrip"])
def test_each_workload_prints_the_contract(monkeypatch, capsys, tmp_path, workload):
    line, detail = run_main(monkeypatch, capsys, tmp_path, ["--workload", workload, "--no-cpu", "--no-secondary"] + SMALL)
    assert line["config"]["verified"]["items"] == (12 if workload == "adx_roundtrip" else 6)
    r = line["roofline"]
    assert r["frac"] == r["frac_end_to_end"] and r["achieved"] >= 0
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (line["ms_per_step"] * 1e-3) / 1e9) <= 0.02 * r["achieved"] + 0.01
    d = r["dominant_kernel"]
    assert d["name"] in r["kernel_ms_per_step"] and d["own_algorithmic_bytes"] <= r["algorithmic_bytes_per_launch"]
    if workload == "hca_decode":                                # the transform owns the PCM bytes only, not the frame bytes the parse reads
        assert d["name"] == "k_hca_transform" and d["own_algorithmic_bytes"] == 4096 * line["config"]["frames_per_stream"] * 6


def test_awb_workload(monkeypatch, capsys, tmp_path):
    line, detail = run_main(monkeypatch, capsys, tmp_path, ["--workload", "awb_mixed", "--awb-clips", "30", "--awb-durations", "8", "--no-cpu", "--steps", "1", "--warmup", "1"])
    assert line["unit"] == "frames/s" and line["config"]["hca_frames"] > 0 and line["config"]["adx_frames"] > 0
    assert detail["config"]["verified"]["items"] == 30


def test_default_run_with_secondaries_and_cpu_baseline(monkeypatch, capsys, tmp_path):
    """The default invocation's flow at toy sizes: headline + sustained loop + every secondary at two sizes + the other BASELINE
    configurations + the reference on the host's cores (cpu_baseline: the real reference binary when oracle/_ref/criref is there)."""
    line, detail = run_main(monkeypatch, capsys, tmp_path, SMALL[:6] + ["--steps", "2", "--warmup", "1", "--secondary-streams", "4", "--awb-clips", "24", "--awb-durations", "6",
                                                                           "--config-awb-clips", "24", "--config-items-scale", "0.0005", "--config-seconds-scale", "0.02",
                                                                           "--cpu-seconds", "0.4", "--config-cpu-seconds", "0.4", "--sustain", "0.2"])
    b = line["cpu_baseline"]
    assert b["value"] > 0 and b["cores"] == 1 and b["kind"] in ("reference", "port")
    assert line["config"]["sustained"]["frames_per_s"] > 0
    sec = detail["secondary"]
    for k in ("hca_decode_sparse_spectra", "hca_decode_middle", "hca_decode_lowest", "hca_decode_6ch", "hca_decode_8ch", "hca_decode_v3_noise_fill", "hca_decode_6ch_v3_noise_fill"):
        assert sec[k]["verified_items"] > 0 and sec[k]["frac_end_to_end"] >= 0, k
        assert sec[k + "_full"]["verified_items"] > 0 and sec[k + "_full"]["channel_frames_per_s"] > 0, k      # the same row at the headline's size
    assert sec["hca_decode_6ch_full"]["frames"] * 6 == sec["hca_decode_sparse_spectra_full"]["frames"] * 2      # equal channel-frames (6 streams x 2 ch = 2 streams x 6 ch)
    cfgs = sec["baseline_configs"]
    assert set(k.split(" ")[0] for k in cfgs) == {"configs[1]", "configs[3]", "configs[4]"}
    assert all("NOT the written size" in v["workload"] for v in cfgs.values())
    assert all(v["roofline"]["frac"] == v["roofline"]["frac_end_to_end"] for v in cfgs.values())
    assert "configs[3]" in line["other_configs_M_per_s"]


def test_full_size_secondaries_respect_the_wall_time_budget(monkeypatch, capsys, tmp_path):
    """--full-secondary-budget: past it the full-size forms are skipped and say so; the 1000-stream forms, the configurations and the line stay."""
    line, detail = run_main(monkeypatch, capsys, tmp_path, SMALL[:6] + ["--steps", "1", "--warmup", "0", "--secondary-streams", "4", "--awb-clips", "24", "--awb-durations", "6",
                                                                           "--config-awb-clips", "24", "--config-items-scale", "0.0005", "--config-seconds-scale", "0.02",
                                                                           "--no-cpu", "--sustain", "0", "--full-secondary-budget", "0"])
    sec = detail["secondary"]
    assert sec["hca_decode_middle"]["verified_items"] > 0 and "skipped" in sec["hca_decode_middle_full"] and "skipped" in sec["hca_decode_6ch_full"]
    assert "hca_decode_middle_full" not in sec["summary_M_per_s"] and len(sec["baseline_configs"]) >= 3


@pytest.mark.timeout(600)
def test_two_rank_launcher_gathers_and_verifies_on_the_root():
    """bench.py --gpus 2 --workload awb_mixed --scaling strong over gloo: the LPT deal, both ranks' jobs, gather_bytes_to_root, and the
    root's check of every gathered item of both ranks against the oracle (what tests/test_gpu_multirank.py runs on the GPU box)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("BENCH_DETAIL_DIR", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bench_dry_run.py"), "--gpus", "2", "--workload", "awb_mixed", "--scaling", "strong", "--awb-clips", "40",
                        "--awb-durations", "12", "--steps", "1", "--warmup", "1", "--no-cpu"], capture_output=True, text=True, timeout=500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [x for x in r.stdout.split("\n") if x.strip().startswith("{")]
    assert len(lines) == 1 and len(lines[0]) <= bench.LINE_LIMIT, r.stdout[-1500:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["config"]["gathered_items_verified_on_root"] == 40 and line["config"]["gathered_bytes_on_root"] > 0
