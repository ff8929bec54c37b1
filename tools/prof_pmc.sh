#!/bin/bash
# Run ON THE GPU BOX via gpurun: rocprofv3 --pmc passes (counters only, no tracing domains) over a command.
#   PMC_SETS="A B C;D E" CMD="python tools/debug/parse_time.py" bash tools/prof_pmc.sh
# Writes gpurun_out/pmc/pmc.json = average counter value per dispatch, per kernel.
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
RAW=/tmp/pmc_raw
rm -rf $OUT $RAW; mkdir -p $OUT $RAW
cd $GRAFT_REPO_ROOT
CMD="${CMD:-python tools/debug/parse_time.py}"
i=0
IFS=';' read -ra SETS <<< "$PMC_SETS"
for set in "${SETS[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $RAW/p$i -o p -- $CMD > $OUT/pass$i.log 2>&1
  tail -1 $OUT/pass$i.log
done
python - <<'PY'
import csv, glob, collections, os, json
raw='/tmp/pmc_raw'; out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/pmc'
res={}
for f in glob.glob(raw+'/**/*counter_collection.csv', recursive=True):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.Counter())
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if 'cri::' not in k: continue
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k][r['Counter_Name']]+=1
    for k,v in agg.items():
        for c,val in v.items(): res.setdefault(k,{})[c]=val/cnt[k][c]
json.dump(res,open(out+'/pmc.json','w'),indent=1,sort_keys=True)
print(json.dumps(res,indent=1,sort_keys=True))
PY
rm -rf $RAW
