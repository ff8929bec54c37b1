#!/bin/bash
# Profiling helper run ON THE GPU BOX via gpurun: kernel-trace stats + two PMC passes of a (small) bench run.
# Only compact summaries are left under gpurun_out/prof (the raw rocprofv3 output is deleted).
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof
RAW=/tmp/prof_raw
rm -rf $OUT $RAW; mkdir -p $OUT $RAW
cd $GRAFT_REPO_ROOT
ARGS="${BENCH_ARGS:---streams 1000 --steps 3 --warmup 1 --no-cpu --no-secondary}"
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- python bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $RAW/pmc1 -o pmc1 -- python bench.py $ARGS > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $RAW/pmc2 -o pmc2 -- python bench.py $ARGS > $OUT/pmc2.log 2>&1
if [ -n "$PMC3" ]; then
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $RAW/pmc3 -o pmc3 -- python bench.py $ARGS > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $RAW/pmc4 -o pmc4 -- python bench.py $ARGS > $OUT/pmc4.log 2>&1
fi
find $RAW -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
python - <<'PY'
import csv, glob, collections, os, json
raw='/tmp/prof_raw'; out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/prof'
res={}
for f in glob.glob(raw+'/pmc*/**/*counter_collection.csv', recursive=True):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.Counter())
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if 'cri::' not in k: continue
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k][r['Counter_Name']]+=1
    for k,v in agg.items():
        for c,val in v.items(): res.setdefault(k,{})[c]=round(val/cnt[k][c])
json.dump(res,open(out+'/pmc_summary.json','w'),indent=1,sort_keys=True)
print(json.dumps(res,indent=1,sort_keys=True))
print(open(out+'/kernel_stats.csv').read()[:1500] if os.path.exists(out+'/kernel_stats.csv') else 'no stats')
PY
rm -rf $RAW
