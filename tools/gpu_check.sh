#!/bin/bash
# Run ON THE GPU BOX via gpurun: the GPU test suite, the default bench line, the other workloads, and the launcher smoke test.
#   OUT=gpurun_out/<tag> bash tools/gpu_check.sh [tests] [bench] [workloads] [launcher]
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=${OUT:-gpurun_out/check}
mkdir -p $OUT
WHAT="${@:-tests bench workloads launcher}"
for w in $WHAT; do
case $w in
tests)
  timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -15 $OUT/pytest.log ;;
bench)
  timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err; head -c 3000 $OUT/bench.json; echo ;;
workloads)
  timeout 900 python bench.py --workload hca_encode --steps 3 --warmup 1 > $OUT/bench_hca_encode.json 2> $OUT/bench_hca_encode.err; echo "hca_encode rc=$?"; tail -2 $OUT/bench_hca_encode.err; head -c 1500 $OUT/bench_hca_encode.json; echo
  timeout 600 python bench.py --workload adx_roundtrip > $OUT/bench_adx_roundtrip.json 2> $OUT/bench_adx_roundtrip.err; echo "adx_roundtrip rc=$?"; tail -2 $OUT/bench_adx_roundtrip.err; head -c 1500 $OUT/bench_adx_roundtrip.json; echo
  timeout 600 python bench.py --workload awb_mixed > $OUT/bench_awb.json 2> $OUT/bench_awb.err; echo "awb rc=$?"; tail -2 $OUT/bench_awb.err; head -c 1200 $OUT/bench_awb.json; echo ;;
launcher)
  CRICODECS_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --streams 1000 --no-cpu --no-secondary > $OUT/bench_launcher2.json 2> $OUT/bench_launcher2.err; echo "launcher rc=$?"; tail -3 $OUT/bench_launcher2.err; head -c 1200 $OUT/bench_launcher2.json; echo ;;
esac
done
