export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r05_d; mkdir -p $OUT
OUT=$OUT bash tools/gpu_check.sh tests bench > $OUT/check.log 2>&1
timeout 600 python bench.py --workload hca_encode --steps 3 --warmup 1 > $OUT/bench_hca_encode.json 2> $OUT/bench_hca_encode.err
for w in hca_decode hca_encode; do
 if [ $w = hca_decode ]; then C="python bench.py --no-cpu --no-secondary --no-verify --steps 5 --warmup 2"; else C="python bench.py --workload hca_encode --no-cpu --no-verify --steps 3 --warmup 1"; fi
 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/t_$w -o t -- $C > $OUT/trace_$w.log 2>&1
 find /tmp/t_$w -name "*kernel_stats.csv" -exec cp {} $OUT/${w}_kernel_stats.csv \;
done
tail -20 $OUT/check.log | cut -c1-600; head -5 $OUT/hca_decode_kernel_stats.csv; head -4 $OUT/hca_encode_kernel_stats.csv; cut -c1-400 $OUT/bench_hca_encode.json
