#!/bin/bash
# Run ON THE GPU BOX via gpurun: round 5's measurement set at the checked-out commit.
#   TAG=r06_a COMMIT=$(git rev-parse --short HEAD) bash tools/prof_r06.sh        (SKIP_BENCH=1: profiles only)
#   1. bench lines: default (headline + every BASELINE configuration + secondaries + CPU baselines) and the three --workload forms
#   2. rocprofv3 --kernel-trace --stats of the headline-only run, the other workloads, the secondaries at 1000 streams, the wide layouts
#   3. counters, every pass its own rocprofv3 run with --pmc only (no trace domains), AT THE SIZE THE NUMBERS ARE QUOTED ON for the HCA
#      decode (10 000 streams) and encode (10 000 x 30 s) -- with --kernel-include-regex the full-size passes take seconds; the ADX round
#      trip is its written size anyway (1000 x 10 s); 1000-stream passes are kept beside them (occupancy of a chip filled 1.8 times):
#      FETCH_SIZE / WRITE_SIZE (HBM-side traffic) and two SQ sets for HCA decode (tonal, sparse), HCA encode, the ADX round trip;
#      plus the calibration streams (tools/debug/hbm_calibrate.py) that give FETCH_SIZE / WRITE_SIZE their per-width scale.
#      Every dispatch is kept (counters_raw.json holds sums and dispatch counts): kernels that run several times per step are SUMMED.
#   4. the encoder's phase split (a library rebuilt with -DCRI_ENC_PROFILE, then rebuilt plain)
# Compact summaries land in gpurun_out/$TAG; raw rocprofv3 output is deleted.  Every profiler pass runs under `timeout`.
export TMPDIR=/tmp
TAG=${TAG:-r06}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
RAW=/tmp/prof_raw
rm -rf $RAW; mkdir -p $OUT $RAW
cd $GRAFT_REPO_ROOT
echo "${COMMIT:-unknown}" > $OUT/commit.txt
if [ -z "$SKIP_BENCH" ]; then
# (round 6: stdout is ONE line <= 4 KB; the full result goes to $BENCH_DETAIL_DIR/bench_detail.json)
BENCH_DETAIL_DIR=$OUT/detail_default timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; wc -c $OUT/bench.json
BENCH_DETAIL_DIR=$OUT/detail_hca_encode timeout 900 python bench.py --workload hca_encode --steps 3 --warmup 1 > $OUT/bench_hca_encode.json 2> $OUT/bench_hca_encode.err
BENCH_DETAIL_DIR=$OUT/detail_adx_roundtrip timeout 600 python bench.py --workload adx_roundtrip > $OUT/bench_adx_roundtrip.json 2> $OUT/bench_adx_roundtrip.err
BENCH_DETAIL_DIR=$OUT/detail_awb_mixed timeout 600 python bench.py --workload awb_mixed > $OUT/bench_awb_mixed.json 2> $OUT/bench_awb_mixed.err
for w in default hca_encode adx_roundtrip awb_mixed; do cp $OUT/detail_$w/bench_detail.json $OUT/bench_detail_$w.json 2>/dev/null; done
fi
if [ -z "$SKIP_PROFILES" ]; then
declare -A CMDS
CMDS[hca_decode]="python bench.py --no-cpu --no-secondary --no-verify --steps 5 --warmup 2"
CMDS[hca_encode]="python bench.py --workload hca_encode --no-cpu --no-verify --steps 3 --warmup 1"
CMDS[adx_roundtrip]="python bench.py --workload adx_roundtrip --no-cpu --no-verify"
CMDS[awb_mixed]="python bench.py --workload awb_mixed --no-verify --no-cpu --awb-clips 100000"
CMDS[hca_crypt]="python tools/debug/crypt_time.py"
CMDS[secondaries_1000]="python bench.py --streams 1000 --no-cpu --no-verify --steps 2 --warmup 1 --host-streams 500 --config-awb-clips 12500"
CMDS[wide_layouts]="python tools/debug/wide_layouts.py 1000"
CMDS[enc_layouts]="python tools/debug/enc_layouts.py"
for w in hca_decode hca_encode adx_roundtrip awb_mixed hca_crypt secondaries_1000 wide_layouts enc_layouts; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/t_$w -o t -- ${CMDS[$w]} > $OUT/trace_$w.log 2>&1
  find $RAW/t_$w -name "*kernel_stats.csv" -exec cp {} $OUT/${w}_kernel_stats.csv \;
done
# counter passes (STEPS = warm-up + timed steps of each command: dispatches are summed and divided by it, tools/make_profiles_r06.py)
declare -A PCMD
PCMD[hca_decode_full]="python bench.py --no-cpu --no-secondary --no-verify --steps 1 --warmup 0"
PCMD[hca_decode_sparse_full]="python bench.py --data sparse --no-cpu --no-secondary --no-verify --steps 1 --warmup 0"
PCMD[hca_encode_full]="python bench.py --workload hca_encode --no-cpu --no-verify --steps 1 --warmup 0"
PCMD[hca_decode]="python bench.py --streams 1000 --no-cpu --no-secondary --no-verify --steps 3 --warmup 1"
PCMD[hca_decode_sparse]="python bench.py --streams 1000 --data sparse --no-cpu --no-secondary --no-verify --steps 3 --warmup 1"
PCMD[hca_encode]="python bench.py --workload hca_encode --streams 1000 --seconds 10 --no-cpu --no-verify --steps 3 --warmup 1"
PCMD[adx_roundtrip]="python bench.py --workload adx_roundtrip --no-cpu --no-verify --steps 3 --warmup 1"
PCMD[adx_roundtrip_sfx]="python bench.py --workload adx_roundtrip --data sfx --no-cpu --no-verify --steps 3 --warmup 1"
PCMD[calibration]="python tools/debug/hbm_calibrate.py"
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
SQ2="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE"
KRE="k_hca_|k_adx_|k_test_stream"
for w in calibration hca_decode_full hca_decode_sparse_full hca_encode_full hca_decode hca_decode_sparse hca_encode adx_roundtrip adx_roundtrip_sfx; do
  timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$KRE" --output-format csv -d $RAW/f_$w -o f -- ${PCMD[$w]} > $OUT/fetch_$w.log 2>&1
  timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$KRE" --output-format csv -d $RAW/w_$w -o w -- ${PCMD[$w]} > $OUT/write_$w.log 2>&1
  if [ $w != calibration ]; then
  timeout 400 rocprofv3 --pmc $SQ1 --kernel-include-regex "$KRE" --output-format csv -d $RAW/s_$w -o s -- ${PCMD[$w]} > $OUT/sq1_$w.log 2>&1
  timeout 400 rocprofv3 --pmc $SQ2 --kernel-include-regex "$KRE" --output-format csv -d $RAW/q_$w -o q -- ${PCMD[$w]} > $OUT/sq2_$w.log 2>&1
  fi
done
python - <<'PY'
import csv, glob, collections, os, json
raw='/tmp/prof_raw'; out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/'+os.environ.get('TAG','r06')
res={}
for d in sorted(glob.glob(raw+'/[fwsq]_*')):
    w=os.path.basename(d)[2:]
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.Counter())
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('(')[0]
            if 'cri::' not in k: continue
            agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k][r['Counter_Name']]+=1
        for k,v in agg.items():
            for c,val in v.items():
                e=res.setdefault(w,{}).setdefault(k,{})
                e[c]=val; e['dispatches']=cnt[k][c]                      # SUM over every dispatch of the run, and how many there were
json.dump(res,open(out+'/counters_raw.json','w'),indent=1,sort_keys=True)
print(json.dumps({w:{k:v.get('dispatches') for k,v in ks.items()} for w,ks in res.items()},indent=1)[:3000])
PY
rm -rf $RAW
fi   # SKIP_PROFILES
# the encoder's phase split
if [ -z "$SKIP_PHASES" ]; then
export CRI_HIPCC_EXTRA=-DCRI_ENC_PROFILE            # (for the runs too: the binding only loads a library built with the flags it is told)
python -m pycricodecs_amd.build > /dev/null 2>&1
python tools/debug/enc_phases.py 2 > $OUT/hca_encode_phases.txt 2>&1
python tools/debug/enc_phases.py 2 3 > $OUT/hca_encode_phases_low.txt 2>&1
python tools/debug/enc_phases.py 8 > $OUT/hca_encode_phases_8ch.txt 2>&1
unset CRI_HIPCC_EXTRA
python -m pycricodecs_amd.build > /dev/null 2>&1
tail -15 $OUT/hca_encode_phases.txt
fi
for w in hca_decode hca_encode adx_roundtrip awb_mixed; do echo "== $w"; head -8 $OUT/${w}_kernel_stats.csv; done
