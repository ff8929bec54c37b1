#!/bin/bash
# Developer loop for k_hca_encode (GPU box): parity tests that touch the encoder, a 1000-stream bench line, instruction counters.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${TAG:-enc}
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "encode or loop or typed or front_end or sfa" > $OUT/pytest_enc.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_enc.log
timeout 600 python bench.py --workload hca_encode --streams 1000 --seconds 10 --steps 5 --warmup 2 --no-cpu > $OUT/enc1000.json 2> $OUT/enc1000.err; echo "bench rc=$?"; python -c "
import json;d=json.load(open('$OUT/enc1000.json'));print(d['value']/1e6,'M frames/s',d['ms_per_step'],'ms')"
if [ -z "$NO_PMC" ]; then
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
PMC_SETS="$SQ1;GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" CMD="python bench.py --workload hca_encode --streams 1000 --seconds 10 --no-cpu --no-verify --steps 3 --warmup 1" bash tools/prof_pmc.sh > $OUT/pmc.log 2>&1
python - <<PY
import json
d=json.load(open('gpurun_out/pmc/pmc.json'))
for k,v in d.items():
    if 'encode' in k:
        fr=469000.0
        print(k, 'VALU/frame %.0f SALU/frame %.0f LDS/frame %.0f conflicts/frame %.0f busy %.3f'%(v['SQ_INSTS_VALU']/fr, v['SQ_INSTS_SALU']/fr, v['SQ_INSTS_LDS']/fr, v['SQ_LDS_BANK_CONFLICT']/fr, 4*v['SQ_INSTS_VALU']/1024/(v['GRBM_GUI_ACTIVE']/8)))
json.dump(d,open('$OUT/pmc.json','w'),indent=1)
PY
fi
