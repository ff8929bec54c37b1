#!/bin/bash
# ON THE GPU BOX: A/B of two sets of compile flags (CRI_HIPCC_EXTRA) for the HCA encoder on ONE box, alternating, twice each:
#   FLAGS_A="" FLAGS_B="-DENC_MIN_WAVES_PER_SIMD=6" [FLAGS_C=...] [QUALS="1 3"] [CMD="python tools/debug/dec_kernels.py 10000 sparse"] bash tools/debug/ab_flags.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
QUALS=${QUALS:-1}
for rep in 1 2; do
  for v in A B C D; do
    var=FLAGS_$v
    [ -z "${!var+x}" ] && continue
    export CRI_HIPCC_EXTRA="${!var}"
    python -m pycricodecs_amd.build > /dev/null 2>&1 || echo "build failed: $v"
    if [ -n "$CMD" ]; then echo -n "$v [${!var}]: "; $CMD 2>&1 | tail -1
    else for q in $QUALS; do echo -n "$v [${!var}] q$q: "; python tools/debug/enc_time.py ${CH:-2} $q 2>&1 | tail -1; done; fi
  done
done
