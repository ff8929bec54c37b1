#!/bin/bash
# ON THE GPU BOX: A/B of two source trees for the HCA encoder: _ab_old/pycricodecs_amd/csrc (A) against pycricodecs_amd/csrc (B), twice each
cd $GRAFT_REPO_ROOT
cp -r pycricodecs_amd/csrc /tmp/csrc_new
for rep in 1 2; do
  cp _ab_old/pycricodecs_amd/csrc/* pycricodecs_amd/csrc/; python -m pycricodecs_amd.build --force > /dev/null 2>&1; echo "A(old)"; python tools/debug/enc_quality_times.py 2 2>&1 | grep "quality 1\|quality 3"
  cp /tmp/csrc_new/* pycricodecs_amd/csrc/; python -m pycricodecs_amd.build --force > /dev/null 2>&1; echo "B(new)"; python tools/debug/enc_quality_times.py 2 2>&1 | grep "quality 1\|quality 3"
done
