#!/bin/bash
# ON THE GPU BOX: A/B of libraries built IN THE BUILD CONTAINER with different compile flags, each in its own directory
# (CRICODECS_LIB_DIR=pycricodecs_amd/lib_<name> CRI_HIPCC_EXTRA="<flags>" python -m pycricodecs_amd.build), alternating on one box:
#   VARIANTS="base: ilp:-mllvm -amdgpu-sched-strategy=max-ilp" (separated by |) bash tools/debug/ab_libs.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
IFS='|' read -ra VS <<< "$VARIANTS"
for rep in 1 2; do
  for v in "${VS[@]}"; do
    n=${v%%:*}; f=${v#*:}
    if [ "$n" = base ]; then unset CRICODECS_LIB_DIR CRI_HIPCC_EXTRA; else export CRICODECS_LIB_DIR=$GRAFT_REPO_ROOT/pycricodecs_amd/lib_$n CRI_HIPCC_EXTRA="$f"; fi
    export CRICODECS_NO_REBUILD=1
    echo -n "$n dec: "; python tools/debug/dec_kernels.py 10000 2>&1 | tail -1
    echo -n "$n enc: "; python tools/debug/enc_time.py 2 1 2>&1 | tail -1
    echo -n "$n adx: "; python bench.py --workload adx_roundtrip --no-cpu --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline'].get('kernel_ms_per_step'))"
  done
done
