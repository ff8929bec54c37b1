#!/bin/bash
# ON THE GPU BOX: A/B of two source trees on the same box: _ab_old/pycricodecs_amd/csrc (A) against pycricodecs_amd/csrc (B), twice each
cd $GRAFT_REPO_ROOT
cp -r pycricodecs_amd/csrc /tmp/csrc_new
run() { python bench.py --no-secondary --no-cpu --steps ${STEPS:-8} --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$1', d['value'], d['roofline']['kernel_ms_per_step'], d['config']['verified']['items'])"; }
for rep in 1 2; do
  cp _ab_old/pycricodecs_amd/csrc/* pycricodecs_amd/csrc/; python -m pycricodecs_amd.build > /dev/null 2>&1; run "A(old)"
  cp /tmp/csrc_new/* pycricodecs_amd/csrc/; python -m pycricodecs_amd.build > /dev/null 2>&1; run "B(new)"
done
