#!/bin/bash
# ON THE GPU BOX: the parse's synchronised top-up interval (-DHCA_FEED_SYNC=n) A/B on one box, alternating, twice.  The variants are built
# IN THE BUILD CONTAINER, each into a directory of its own:
#   for n in 2 4 5; do CRICODECS_LIB_DIR=$PWD/pycricodecs_amd/lib_fs$n CRI_HIPCC_EXTRA="-DHCA_FEED_SYNC=$n" python -m pycricodecs_amd.build; done
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp CRICODECS_NO_REBUILD=1
for rep in 1 2; do
 for n in base 2 4 5; do
  if [ $n = base ]; then unset CRICODECS_LIB_DIR CRI_HIPCC_EXTRA; else export CRICODECS_LIB_DIR=$GRAFT_REPO_ROOT/pycricodecs_amd/lib_fs$n CRI_HIPCC_EXTRA="-DHCA_FEED_SYNC=$n"; fi
  echo -n "feed_sync $n tonal: "; python tools/debug/dec_kernels.py 10000 2>&1 | tail -1
  [ $rep = 1 ] && { echo -n "feed_sync $n sparse: "; python tools/debug/dec_kernels.py 10000 sparse 2>&1 | tail -1; }
 done
done
