#!/bin/bash
# Profiling helper run ON THE GPU BOX via gpurun: kernel trace + two PMC passes of a small bench run.
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
ARGS="${BENCH_ARGS:---streams 1000 --steps 3 --warmup 1 --no-cpu}"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc1 -o pmc1 -- python bench.py $ARGS > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM -d $OUT/pmc2 -o pmc2 -- python bench.py $ARGS > $OUT/pmc2.log 2>&1
find $OUT -name "*.csv" | head -20
python - <<'PY'
import csv, glob, collections, os
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/prof'
for f in glob.glob(out+'/trace/**/*kernel_stats.csv', recursive=True):
    print(open(f).read()[:3000])
for tag in ('pmc1','pmc2'):
    for f in glob.glob(out+'/%s/**/*counter_collection.csv'%tag, recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'][:40]; agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
        for k,v in agg.items():
            print(tag,k,{a:round(b) for a,b in v.items()})
PY
