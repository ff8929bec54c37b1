#!/bin/bash
# ON THE GPU BOX: what k_hca_transform_plain's memory traffic costs -- timing without its line loads / without its PCM stores (results are wrong on purpose)
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-secondary --no-cpu --no-verify --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$1', d['roofline']['kernel_ms_per_step'])"; }
for V in "" "-DHCA_ABL_NOLINES" "-DHCA_ABL_NOPCM" "-DHCA_ABL_NOPCM -DHCA_ABL_NOLINES"; do
  CRI_HIPCC_EXTRA="$V" python -m pycricodecs_amd.build --force > /dev/null 2>&1
  run "[$V]"
done
python -m pycricodecs_amd.build --force > /dev/null 2>&1
