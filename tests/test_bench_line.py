"""The bench line contract (CPU): bench.py prints ONE JSON line of at most 4 KB that round-trips through json and carries the keys the
driver and SURVEY 8(d) ask for; everything else goes to bench_detail.json.  (Round 5's line had grown to 22 KB and the driver's record
of it was `parsed: null`.)"""
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def full_lines():
    """Full result objects of earlier runs (profiles/*bench*.json: what emit() now writes to bench_detail.json)."""
    out = []
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[56]_*bench*.json"))):
        try:
            with open(p) as fh:
                d = json.load(fh)
        except ValueError:
            continue
        if isinstance(d, dict) and "metric" in d and "config" in d:
            out.append((os.path.basename(p), d))
    return out


def test_there_are_recorded_lines_to_check():
    assert len(full_lines()) >= 4


@pytest.mark.parametrize("name,full", full_lines())
def test_compact_line_is_small_and_complete(name, full):
    line = json.dumps(bench.compact_line(full))
    assert len(line) <= bench.LINE_LIMIT, (name, len(line))
    back = json.loads(line)
    for k in CONTRACT:
        assert k in back, (name, k)
    assert isinstance(back["config"].get("workload"), str) and back["config"]["workload"]
    assert "model" not in back["config"]
    if "roofline" in full:
        r = back["roofline"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in r, (name, k)
    if "cpu_baseline" in full:
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in back["cpu_baseline"], (name, k)
    assert back["detail"] == bench.DETAIL_NAME


def test_emit_prints_one_short_line_and_writes_the_detail_file(tmp_path, monkeypatch, capsys):
    name, full = max(full_lines(), key=lambda t: len(json.dumps(t[1])))
    assert len(json.dumps(full)) > 8000                         # (a default run's result with its secondaries)
    monkeypatch.setattr(bench, "DETAIL_ROOT", str(tmp_path))
    monkeypatch.setenv("BENCH_DETAIL_DIR", str(tmp_path / "out"))
    bench.emit(full)
    cap = capsys.readouterr()
    lines = [x for x in cap.out.split("\n") if x.strip()]
    assert len(lines) == 1 and len(lines[0]) <= bench.LINE_LIMIT
    assert json.loads(lines[0])["value"] == full["value"]
    for d in (tmp_path, tmp_path / "out"):
        with open(d / bench.DETAIL_NAME) as fh:
            assert json.load(fh)["secondary"].keys() == full["secondary"].keys()


def test_roofline_frac_is_the_end_to_end_figure():
    kms = {"k_hca_parse": 8.5, "k_hca_transform": 10.5}
    units = 4690000
    r = bench.roofline_of(4096 * units, 4778 * units, kms, 19.2e-3, {"traffic": 18750 * units})
    e2e = 4778 * units / 19.2e-3 / 1e9
    assert abs(r["achieved"] - e2e) < 0.01 and abs(r["frac"] - e2e / 8000.0) < 1e-5 and r["frac"] == r["frac_end_to_end"]
    d = r["dominant_kernel"]
    assert d["name"] == "k_hca_transform" and abs(d["achieved"] - 4096 * units / 10.5e-3 / 1e9) < 0.01
    assert d["frac"] < 0.25 and r["frac"] < 0.15                 # (round 5 reported 0.27 for this: whole-path bytes over one kernel's time)
    assert abs(r["traffic_over_algorithmic"] - 18750 / 4778) < 1e-3
