#!/bin/bash
# ON THE GPU BOX: k_hca_parse feed policy (eager checkpoint period / landing delay) on the full-size headline batch
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-secondary --no-cpu --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$1', d['value'], d['roofline']['kernel_ms_per_step'], d['config']['verified']['items'])"; }
for V in ${VARIANTS:-"3 2" "4 2" "4 3" "2 1" "5 3"}; do
  set -- $V
  CRI_HIPCC_EXTRA="-DHCA_FEED_SYNC=$1 -DHCA_FEED_LAND=$2" python -m pycricodecs_amd.build --force > /dev/null 2>&1
  run "SYNC=$1 LAND=$2"
done
python -m pycricodecs_amd.build --force > /dev/null 2>&1
