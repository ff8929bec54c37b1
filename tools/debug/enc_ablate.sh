#!/bin/bash
# Developer tool (GPU box): end-to-end time of k_hca_encode with one piece changed at a time (results are wrong for most variants:
# timing only).  Rebuilds only cri_hca_enc.hip and relinks.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
L=pycricodecs_amd/lib
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function -Wno-unused-result"
cp $L/libcricodecs_hip.so /tmp/lib_keep.so; cp $L/cri_hca_enc.o /tmp/enc_keep.o
for v in ${VARIANTS:-base -DENC_MAX_WAVES=2 -DENC_ABL_NOBARRIER -DENC_ABL_STEPS=1 -DENC_ABL_NOPUT}; do
  f="$v"; [ "$v" = "base" ] && f=""
  /opt/rocm/bin/hipcc $FLAGS $f -x hip -c pycricodecs_amd/csrc/cri_hca_enc.hip -o $L/cri_hca_enc.o 2>/dev/null || { echo "$v: compile failed"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $L/cri_host.o $L/cri_hca_dec.o $L/cri_hca_enc.o $L/cri_adx.o $L/cri_misc.o $L/cri_capi.o -o $L/libcricodecs_hip.so -Wl,-rpath,/opt/rocm/lib
  for ch in ${CHS:-2}; do
  r=$(timeout 300 python tools/debug/enc_time.py $ch ${QUAL:-1} 2>&1 | tail -1)
  echo "$v ch=$ch: $r"
  done
done
cp /tmp/lib_keep.so $L/libcricodecs_hip.so; cp /tmp/enc_keep.o $L/cri_hca_enc.o
