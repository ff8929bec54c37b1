#!/bin/bash
# Run ON THE GPU BOX via gpurun: round 6's checks.   OUT=gpurun_out/<tag> bash tools/gpu_r06.sh [tests] [bench] [workloads] [power] ...
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=${OUT:-gpurun_out/r06}
mkdir -p $OUT
export BENCH_DETAIL_DIR=$GRAFT_REPO_ROOT/$OUT
WHAT="${@:-tests bench}"
for w in $WHAT; do
case $w in
tests)
  timeout 1800 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -8 $OUT/pytest.log ;;
bench)
  /usr/bin/time -v -o $OUT/bench.time timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; grep Elapsed $OUT/bench.time; wc -c $OUT/bench.json; cat $OUT/bench.json; cp bench_detail.json $OUT/bench_detail.json 2>/dev/null ;;
workloads)
  for wl in hca_encode adx_roundtrip awb_mixed; do
    BENCH_DETAIL_DIR=$GRAFT_REPO_ROOT/$OUT/$wl timeout 900 python bench.py --workload $wl $([ $wl = hca_encode ] && echo "--steps 3 --warmup 1") > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err; echo "$wl rc=$?"; cat $OUT/bench_$wl.json
  done ;;
abdecode)
  # round 6's two decoder changes (code descriptions in the frame records; shared line for lanes that do not ask) against round 5's decoder
  # (tools/debug/experiments/baseline_round5_decoder.patch: correct output), alternating on this box: kernels by HIP events, then clock / power / ms
  for rep in 1 2 3; do
    echo -n "[product] " | tee -a $OUT/ab_decode.txt; python tools/debug/dec_kernels.py 10000 2>&1 | tail -1 | tee -a $OUT/ab_decode.txt
    bash tools/debug/experiments/variant.sh "baseline_round5_decoder" python tools/debug/dec_kernels.py 10000 2>&1 | tail -1 | tee -a $OUT/ab_decode.txt
  done
  for q in 3; do
    echo -n "[product q$q] " | tee -a $OUT/ab_decode.txt; python tools/debug/dec_kernels.py 10000 tonal $q 2>&1 | tail -1 | tee -a $OUT/ab_decode.txt
    bash tools/debug/experiments/variant.sh "baseline_round5_decoder" python tools/debug/dec_kernels.py 10000 tonal $q 2>&1 | tail -1 | tee -a $OUT/ab_decode.txt
  done
  echo -n "[product] " | tee -a $OUT/ab_decode.txt; python tools/debug/dec_power.py 10000 5 2>&1 | tail -1 | tee -a $OUT/ab_decode.txt
  bash tools/debug/experiments/variant.sh "baseline_round5_decoder" python tools/debug/dec_power.py 10000 5 2>&1 | tail -1 | tee -a $OUT/ab_decode.txt ;;
power)
  # the decode with its traffic cut in ONE build (wrong output; tools/debug/experiments): clock, power, ms beside the product tree's, alternating on this box
  for rep in 1 2; do
    echo -n "[product] " | tee -a $OUT/dec_power.txt; python tools/debug/dec_power.py 10000 5 2>&1 | tail -1 | tee -a $OUT/dec_power.txt
    for v in "fold_input" "fold_input fold_descriptions" "fold_input fold_descriptions fold_lines"; do
      bash tools/debug/experiments/variant.sh "$v" python tools/debug/dec_power.py 10000 5 2>&1 | tail -1 | tee -a $OUT/dec_power.txt
    done
  done ;;
esac
done
