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

typedef float f2 __attribute__((ext_vector_type(2)));

// (float)(int8_t)(w >> 8*B) in one instruction (the compiler finds the SDWA form for some of a dword's bytes only)
template <int B> __device__ __forceinline__ float cvt_f32_i8(uint32_t w) {
    float r;
    if (B == 0) asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0" : "=v"(r) : "v"(w));
    else if (B == 1) asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(r) : "v"(w));
    else if (B == 2) asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(r) : "v"(w));
    else asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3" : "=v"(r) : "v"(w));
    return r;
}

// (float)(int16_t)(w >> 16*H) in one instruction
template <int H> __device__ __forceinline__ float cvt_f32_i16(uint32_t w) {
    float r;
    if (H == 0) asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(r) : "v"(w));
    else asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(r) : "v"(w));
    return r;
}

// LDS address of a pointer into shared memory, and 16-bit stores through such an address (no base to add per access)
typedef __attribute__((address_space(3))) uint16_t lds_u16;
__device__ __forceinline__ uint32_t lds_address(const void* p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint8_t*)p; }

// packed adds whose second operand is negated in one half only: the modifier does it inside the instruction (the compiler builds
// {b.x, -b.y} with a packed negate and a move first -- two more instructions per use, 24 per pass of four transforms)
__device__ __forceinline__ f2 pk_add_neg_hi(f2 a, f2 b) { f2 r; asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }   // {a.x + b.x, a.y - b.y}
__device__ __forceinline__ f2 pk_add_neg_lo(f2 a, f2 b) { f2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }   // {a.x - b.x, a.y + b.y}
__device__ __forceinline__ f2 pk_sum_diff(f2 p) { f2 r; asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(p)); return r; }   // {p.x + p.y, p.x - p.y}

// packed fp32 pair (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 operate on these)
// value of the lane whose lane16 differs in bit X (X = 1, 2, 4, 8), via DPP
template <int X> __device__ __forceinline__ float lane16_xor(float v) {
    const int i = __float_as_int(v);
    if (X == 1) return __int_as_float(__builtin_amdgcn_update_dpp(0, i, 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    if (X == 2) return __int_as_float(__builtin_amdgcn_update_dpp(0, i, 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
#ifdef CRI_XOR8_SWIZZLE
    if (X == 8) return __int_as_float(__builtin_amdgcn_ds_swizzle(i, 0x201F));
#endif
    if (X == 8) return __int_as_float(__builtin_amdgcn_update_dpp(0, i, 0x128, 0xF, 0xF, true));   // row_ror:8
#ifdef CRI_XOR4_DPP
    const int t = __builtin_amdgcn_update_dpp(0, i, 0x104, 0xF, 0xF, true);                        // row_shl:4 (right for banks 0,2)
    return __int_as_float(__builtin_amdgcn_update_dpp(t, i, 0x114, 0xF, 0xA, false));              // row_shr:4 into banks 1,3
#else
    // lane ^ 4 has no single DPP pattern (it would take two VALU moves); ds_swizzle's bit mode (and 0x1F, or 0, xor 4)
    // does it in one instruction on the LDS crossbar, off the VALU the kernels are bound by
    return __int_as_float(__builtin_amdgcn_ds_swizzle(i, 0x101F));
#endif
}
template <int X> __device__ __forceinline__ f2 lane16_xor2(f2 v) { f2 r; r.x = lane16_xor<X>(v.x); r.y = lane16_xor<X>(v.y); return r; }

__device__ __forceinline__ float fneg_if(float v, bool n) { return n ? -v : v; }

// stream lookup: largest s in [lo, hi) with streams[s].first_frame <= g
__device__ __forceinline__ uint32_t find_stream(const HcaStream* streams, uint32_t lo, uint32_t hi, uint32_t g) {
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (streams[mid].first_frame <= g) lo = mid; else hi = mid;
    }
    return lo;
}
}  // namespace cri
