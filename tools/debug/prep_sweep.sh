#!/bin/bash
# ON THE GPU BOX: k_hca_prepare variants (chunks per visit, waves per CU) on the full-size headline batch
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-secondary --no-cpu --no-verify --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$1', d['roofline']['kernel_ms_per_step'])"; }
for G in 1 2 4 8; do
  CRI_HIPCC_EXTRA="-DHCA_PREP_GROUP=$G -DHCA_PREP_WAVES=$([ $G -ge 8 ] && echo 4 || echo 8)" python -m pycricodecs_amd.build --force > /dev/null 2>&1
  for L in 0 5 10 20 40; do CRI_PREP_LDS_KB=$L run "G=$G lds=$L"; done
done
