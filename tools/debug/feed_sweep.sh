#!/bin/bash
# ON THE GPU BOX: k_hca_parse feed policy (every how many blocks the lanes top up together) on the full-size headline batch
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-secondary --no-cpu --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$1', d['value'], d['roofline']['kernel_ms_per_step'], d['config']['verified']['items'])"; }
for V in ${VARIANTS:-1 2 3 4 5}; do
  CRI_HIPCC_EXTRA="-DHCA_FEED_SYNC=$V" python -m pycricodecs_amd.build --force > /dev/null 2>&1
  run "SYNC=$V"
done
python -m pycricodecs_amd.build --force > /dev/null 2>&1
