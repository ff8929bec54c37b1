#!/bin/bash
# Run ON THE GPU BOX via gpurun: HBM-side traffic of the decode kernels from the L2 fabric counters, one counter per
# pass (MI355X_MICROARCH.md, HBM section: FETCH_SIZE and WRITE_SIZE do not fit one pass; no tracing domains with --pmc).
# Writes gpurun_out/traffic/traffic_raw.json = average counter value per dispatch and kernel (unit: KB as rocprofv3 reports).
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/traffic
RAW=/tmp/traffic_raw
rm -rf $OUT $RAW; mkdir -p $OUT $RAW
cd $GRAFT_REPO_ROOT
ARGS="${BENCH_ARGS:---steps 3 --warmup 1 --no-cpu --no-secondary}"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $RAW/f -o f -- python bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $RAW/w -o w -- python bench.py $ARGS > $OUT/write.log 2>&1
python - <<'PY'
import csv, glob, collections, os, json
raw='/tmp/traffic_raw'; out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/traffic'
res={}
for f in glob.glob(raw+'/**/*counter_collection.csv', recursive=True):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.Counter())
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if 'cri::' not in k: continue
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k][r['Counter_Name']]+=1
    for k,v in agg.items():
        for c,val in v.items(): res.setdefault(k,{})[c]=val/cnt[k][c]; res[k]['dispatches_'+c]=cnt[k][c]
json.dump(res,open(out+'/traffic_raw.json','w'),indent=1,sort_keys=True)
print(json.dumps(res,indent=1,sort_keys=True))
PY
tail -1 $OUT/fetch.log
rm -rf $RAW
