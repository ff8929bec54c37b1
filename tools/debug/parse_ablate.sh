#!/bin/bash
# ON THE GPU BOX: where k_hca_parse waits -- timing with one kind of memory traffic removed at a time (results are wrong on purpose)
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-secondary --no-cpu --no-verify --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$1', d['roofline']['kernel_ms_per_step'])"; }
for V in "" "-DHCA_ABL_NOSTORE" "-DHCA_ABL_NOLOAD" "-DHCA_ABL_NOMETA" "-DHCA_ABL_NOSTORE -DHCA_ABL_NOLOAD -DHCA_ABL_NOMETA"; do
  CRI_HIPCC_EXTRA="$V" python -m pycricodecs_amd.build --force > /dev/null 2>&1
  run "[$V]"
done
python -m pycricodecs_amd.build --force > /dev/null 2>&1
