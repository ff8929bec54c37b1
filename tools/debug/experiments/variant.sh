#!/bin/bash
# ON THE GPU BOX (or here, to check that the patches apply and compile): a throw-away copy of the repository with the named
# experiment patches applied, built, and a command run INSIDE the copy.  The product tree is never touched.
#   bash tools/debug/experiments/variant.sh "fold_input fold_descriptions" python tools/debug/dec_power.py 10000 5
#   bash tools/debug/experiments/variant.sh "" python tools/debug/dec_power.py 10000 5          (the unpatched copy: the baseline of an A/B)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
NAMES="$1"; shift
TAG=$(echo "${NAMES:-base}" | tr ' ' '+')
V=/tmp/cri_variant_$TAG
if [ ! -d $V ]; then
  mkdir -p $V
  (cd $ROOT && tar cf - --exclude=.git --exclude=gpurun_out --exclude=profiles --exclude='pycricodecs_amd/lib*' --exclude=__pycache__ .) | (cd $V && tar xf -)
  mkdir -p $V/profiles
  for n in $NAMES; do (cd $V && patch -p1 --no-backup-if-mismatch < $ROOT/tools/debug/experiments/$n.patch > /dev/null) || { echo "patch $n does not apply"; exit 1; }; done
  (cd $V && python -m pycricodecs_amd.build > build.log 2>&1) || { tail -5 $V/build.log; echo "variant $TAG: build failed"; exit 1; }
fi
cd $V
export GRAFT_REPO_ROOT=$V
echo -n "[$TAG] "
"$@"
