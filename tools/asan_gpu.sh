#!/bin/bash
# Run ON THE GPU BOX via gpurun: the malformed-input GPU tests on an AddressSanitizer build of the library (device code instrumented:
# every global / LDS access of the kernels is checked against the shadow of the allocations, hipMalloc'ed buffers get red zones).
#   The build (8 minutes of hipcc) is made in the build container and travels with the tree:
#     CRICODECS_LIB_DIR=$PWD/pycricodecs_amd/lib_asan CRI_OFFLOAD_ARCH=gfx950:xnack+ CRI_HIPCC_EXTRA="-fsanitize=address -shared-libsan -g" python -m pycricodecs_amd.build
#     hipcc -fsanitize=address -shared-libsan -g -O1 -x c++ tools/asan_fuzz.cpp -o pycricodecs_amd/lib_asan/asan_fuzz \
#           -Lpycricodecs_amd/lib_asan -lcricodecs_hip -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib
#   This ROCm has no sanitizer build of the HIP runtime (/opt/rocm/lib/asan): a device-side report reaches the host as hostcall service
#   4, for which the plain runtime has no handler -- the process dies with "Hostcall: no handler found for service ID 4".  That IS the
#   detection (the positive control below shows it for a 12-byte over-read and over-write of a hipMalloc'ed buffer); a clean run is a run
#   without it.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${TAG:-asan}
mkdir -p $OUT
ASANDIR=$(dirname $(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1))
export LD_LIBRARY_PATH=$ASANDIR:/opt/rocm/lib:$LD_LIBRARY_PATH
export HSA_XNACK=1
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1:protect_shadow_gap=0
# positive control
( cd tools/probes && /opt/rocm/bin/hipcc --offload-arch=gfx950:xnack+ -fsanitize=address -shared-libsan -g asan_probe.hip -o /tmp/asan_probe > /dev/null 2>&1
  for m in ok read write; do echo "== probe $m"; timeout 60 /tmp/asan_probe $m 2>&1 | grep -v "^$" | head -3; done ) > $OUT/probe.log 2>&1
cat $OUT/probe.log
# the library, driven by tools/asan_fuzz.cpp (no Python in the process: the sanitizer's allocation hooks fail at runtime start-up inside one);
# the harness was linked in the build container next to the sanitized library (pycricodecs_amd/lib_asan/asan_fuzz)
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1
timeout ${LIMIT:-1500} pycricodecs_amd/lib_asan/asan_fuzz ${ROUNDS:-40} ${SEED:-1} tests/golden/*.hca tests/golden/*.adx tests/golden/*.wav tests/golden/*.bin > $OUT/asan_fuzz.log 2>&1
echo "asan_fuzz rc=$?"
head -1 $OUT/asan_fuzz.log; tail -3 $OUT/asan_fuzz.log | cut -c1-300
echo "reports: $(grep -c "service ID 4\|AddressSanitizer" $OUT/asan_fuzz.log)"
