// cri_device.h -- small device-side helpers shared by the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "cri_types.h"

namespace cri {
__device__ __forceinline__ uint32_t ld_u32_unaligned(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint32_t bswap32(uint32_t v) { return __builtin_bswap32(v); }
__device__ __forceinline__ uint64_t readlane64(uint64_t v, int l) {
    uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, l), hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), l);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ int32_t clamp_sym(int32_t v, int32_t limit) { return v > limit ? limit : (v < ~limit ? ~limit : v); }

// Ordering point for LDS traffic inside ONE wave (every kernel here runs one wave per workgroup).  A wave's LDS
// instructions execute in issue order, so lanes exchanging data through LDS need no s_barrier and no wait on the vector
// memory counters -- only the compiler must keep the accesses in program order.  Unlike __syncthreads() this does not
// drain outstanding global loads/stores (vmcnt), so prefetches and PCM stores stay in flight across it.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// stream lookup: largest s in [lo, hi) with streams[s].first_frame <= g
__device__ __forceinline__ uint32_t find_stream(const HcaStream* streams, uint32_t lo, uint32_t hi, uint32_t g) {
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (streams[mid].first_frame <= g) lo = mid; else hi = mid;
    }
    return lo;
}
}  // namespace cri
