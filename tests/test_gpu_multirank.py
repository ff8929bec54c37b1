"""The multi-rank launcher on the GPU box: `bench.py --gpus 2` starts two ranks under torch.distributed.run; with
CRICODECS_BENCH_SHARE_GPU=1 they share the one GPU of the box and rendezvous over gloo (RCCL refuses two ranks on one device), so
what runs here is everything of the N > 1 path except RCCL itself: the launcher, the LPT deal of a fixed AWB bank (--scaling strong),
each rank's HCA + ADX jobs on the device, the variable-length gather of the decoded PCM onto rank 0, and rank 0's check of EVERY
gathered item of every rank against the CPU oracle.  The RCCL point-to-point batch (pycricodecs_amd/shard.py) itself has never run on
device tensors: no multi-GPU box is available to the builder (README)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, timeout=600):
    env = dict(os.environ, CRICODECS_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.pop("BENCH_DETAIL_DIR", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [x for x in r.stdout.split("\n") if x.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    assert len(lines[0]) <= 4096
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_two_ranks_decode_a_fixed_awb_bank_and_gather_it_on_the_root():
    clips = 600
    line = _bench(["--gpus", "2", "--workload", "awb_mixed", "--scaling", "strong", "--awb-clips", str(clips), "--awb-durations", "150", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-secondary"])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and "launcher_smoke_test" in line
    cfg = line["config"]
    assert cfg["gathered_items_verified_on_root"] == clips         # every clip of both ranks, byte for byte against the oracle, in the root's gathered buffer
    assert cfg["gathered_bytes_on_root"] > 0 and cfg["hca_frames"] > 0 and cfg["adx_frames"] > 0
    assert line["value"] > 0 and line["roofline"]["frac"] > 0


@pytest.mark.timeout(900)
def test_two_ranks_of_the_headline_workload_weak_scaling():
    line = _bench(["--gpus", "2", "--streams", "256", "--unique", "8", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-secondary"])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    assert line["config"]["verified"]["items_all_ranks"] == 512 and line["config"]["streams_per_gpu"] == 256
