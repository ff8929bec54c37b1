#!/bin/bash
# ON THE GPU BOX: the shader / memory clocks and the power draw while a kernel family runs in a loop (rocm-smi sampled once a second).
#   bash tools/debug/clocks_under_load.sh
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
cat > /tmp/loop.py <<'PY'
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch, bench as B
from pycricodecs_amd import synth
from pycricodecs_amd.batch import Job
what = sys.argv[1]
if what == "dec":
    job = Job.hca_decode(B.tile(B.make_hca_streams(8, 10.0, 0, 1, "tonal"), 4000), keys=[B.KEY] * 4000)
else:
    ws = [synth.wav(i, 480000, 2, 48000) for i in range(8)]; job = Job.hca_encode((ws * 250)[:2000], quality=1)
bufs = job.alloc("cuda:0")
job.run(*bufs); torch.cuda.synchronize(); print("ready", flush=True)
t0 = time.time(); n = 0
while time.time() - t0 < 9:
    for _ in range(20): job.run(*bufs)
    torch.cuda.synchronize(); n += 20
print("%s: %.3f ms per run over %.1f s" % (what, (time.time() - t0) / n * 1e3, time.time() - t0), flush=True)
PY
echo "== idle"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | head -6
for w in dec enc; do
  python /tmp/loop.py $w 2>/dev/null > /tmp/loop_$w.log &
  pid=$!
  while ! grep -q ready /tmp/loop_$w.log 2>/dev/null; do sleep 0.5; kill -0 $pid 2>/dev/null || break; done
  for i in 1 2 3 4 5 6; do sleep 1.2; echo "== $w sample $i"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | head -4; done
  wait $pid; tail -1 /tmp/loop_$w.log
done
