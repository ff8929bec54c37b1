#!/bin/bash
# Run ON THE GPU BOX via gpurun: the round's measurement set.
#   1. the default bench line (headline + secondaries + CPU baseline) and the other workloads' lines
#   2. rocprofv3 --kernel-trace --stats of the headline-only run and of the hca_encode / adx_roundtrip / awb_mixed / crypt runs
#   3. rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, no tracing domains) of the same commands
# Compact summaries land in gpurun_out/$TAG; raw rocprofv3 output is deleted.  Every profiler pass runs under `timeout`: one
# counter pass of the full-size batch once sat for 39 minutes without finishing (SKIP_BENCH=1 skips step 1).
export TMPDIR=/tmp
TAG=${TAG:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
RAW=/tmp/prof_raw
rm -rf $RAW; mkdir -p $OUT $RAW
cd $GRAFT_REPO_ROOT
if [ -z "$SKIP_BENCH" ]; then
python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -1 $OUT/bench.err
python bench.py --workload hca_encode --steps 3 --warmup 1 > $OUT/bench_hca_encode.json 2> $OUT/bench_hca_encode.err
python bench.py --workload adx_roundtrip > $OUT/bench_adx_roundtrip.json 2> $OUT/bench_adx_roundtrip.err
python bench.py --workload awb_mixed > $OUT/bench_awb_mixed.json 2> $OUT/bench_awb_mixed.err
fi
declare -A CMDS
CMDS[hca_decode]="python bench.py --no-cpu --no-secondary --no-verify --steps 5 --warmup 2"
CMDS[hca_encode]="python bench.py --workload hca_encode --no-cpu --no-verify --steps 3 --warmup 1"
CMDS[adx_roundtrip]="python bench.py --workload adx_roundtrip --no-cpu --no-verify"
CMDS[awb_mixed]="python bench.py --workload awb_mixed --no-verify"
CMDS[hca_crypt]="python tools/debug/crypt_time.py"
# (the counter passes of the full-size decode use three dispatches: with seven the FETCH_SIZE pass did not finish, twice)
PMC_hca_decode="python bench.py --no-cpu --no-secondary --no-verify --steps 2 --warmup 1"
for w in hca_decode hca_encode adx_roundtrip awb_mixed hca_crypt; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/t_$w -o t -- ${CMDS[$w]} > $OUT/trace_$w.log 2>&1
  find $RAW/t_$w -name "*kernel_stats.csv" -exec cp {} $OUT/${w}_kernel_stats.csv \;
  PC=${CMDS[$w]}; if [ $w = hca_decode ]; then PC=$PMC_hca_decode; fi
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $RAW/f_$w -o f -- $PC > $OUT/fetch_$w.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $RAW/w_$w -o w -- $PC > $OUT/write_$w.log 2>&1
done
python - <<'PY'
import csv, glob, collections, os, json
raw='/tmp/prof_raw'; out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/'+os.environ.get('TAG','r02')
res={}
for d in sorted(glob.glob(raw+'/[fw]_*')):
    w=os.path.basename(d)[2:]
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.Counter())
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('(')[0]
            if 'cri::' not in k: continue
            agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k][r['Counter_Name']]+=1
        for k,v in agg.items():
            for c,val in v.items():
                res.setdefault(w,{}).setdefault(k,{})[c]=val/cnt[k][c]; res[w][k]['dispatches_'+c]=cnt[k][c]
json.dump(res,open(out+'/traffic_raw.json','w'),indent=1,sort_keys=True)
print(json.dumps(res,indent=1,sort_keys=True)[:6000])
PY
for w in hca_decode hca_encode adx_roundtrip awb_mixed hca_crypt; do echo "== $w"; head -6 $OUT/${w}_kernel_stats.csv; done
rm -rf $RAW
