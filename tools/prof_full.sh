#!/bin/bash
# Run ON THE GPU BOX via gpurun: full-size bench line + rocprofv3 kernel-trace stats of the same command.
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/full
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
tail -2 $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python bench.py --no-cpu --no-secondary > $OUT/trace.log 2>&1   # headline workload only, so per-kernel averages are those of the bench line
find $OUT/trace -name "*kernel_stats.csv" -exec head -8 {} \;
