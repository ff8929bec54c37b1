// cri_hca_enc_cost.h -- how k_hca_encode's rate loop costs a band without quantising it (tables: hca_enc_build_tables, cri_host.cpp;
// layout HCA_ET_CLS / HCA_ET_CP, cri_types.h).  Shared with the test build's exhaustive check against the reference's rule
// (cri_testing.hip; tests/test_gpu_round4.py::test_hca_encoder_band_cost_rule_on_the_device).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "cri_types.h"

namespace cri {

// class of a scaled spectrum (|v| < 1): how many of the fifteen resolutions' length thresholds of its sign it reaches.  cls =
// the HCA_ET_CLS rows, one per half-binade and sign: {A, classes below}.  The row's byte offset is the float's top ten bits times 8
// (bit 30 is clear: |v| < 1) -- a shift and a mask; then one compare and an add with carry
__device__ __forceinline__ uint32_t enc_class(const uint8_t* cls, float v) {
    const uint2 row = *(const uint2*)(cls + ((__float_as_uint(v) >> 19) & 0x1FF8u));
    return row.y + (fabsf(v) >= __uint_as_float(row.x) ? 1u : 0u);
}
__device__ __forceinline__ uint32_t enc_on_clamp(float v) { return __float_as_uint(v) == HCA_ENC_CLAMP_BITS ? 1u : 0u; }

// bits of a band's 8 spectra at the resolution of table row `row` (HCA_ET_CP: {(16 - rank) in every byte, 8 * shortest |
// resolution << 20 | anomaly << 28}) -- the inner part of CalculateUsedBits, hca.cpp:2771-2786.  cl = the spectra's classes, a byte
// each; ntop = how many of them sit on ScaleSpectra's clamp (only looked at when `tops`: the rare frame that has any)
__device__ __forceinline__ int enc_band_cost(uint2 row, uint32_t cl0, uint32_t cl1, uint32_t ntop, bool tops) {
    int n = (int)(row.y & 0xFF);
    n += __builtin_popcount((cl0 + row.x) & 0x10101010u) + __builtin_popcount((cl1 + row.x) & 0x10101010u);
    if (tops) n -= (row.y >> 28) & 1 ? (int)(ntop * (((row.y & 0xFF) >> 3) + 1)) : 0;
    return n;
}
// The same for the rate loop's search steps, which only want the SUM over many bands: `acc` plus the row's second word as it is plus
// the two counts -- the low 20 bits of a sum of these are the bits (the resolution / anomaly fields above them add up to junk that
// never carries downwards).  Six instructions a band: two adds, two ands, two counts that accumulate.
#ifdef __HIPCC__                 // (gfx950 instructions by name: not part of the host-side enumeration of this header, tests/shim/device_fn_host.cpp)
__device__ __forceinline__ uint32_t enc_band_cost_raw(uint2 row, uint32_t cl0, uint32_t cl1, uint32_t acc) {
    // (v_bcnt_u32_b32 adds its second operand: written out, because the compiler counts into a fresh register and adds afterwards)
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"((cl0 + row.x) & 0x10101010u), "v"(acc));
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(acc) : "v"((cl1 + row.x) & 0x10101010u), "v"(r));
    return acc;
}
#endif
#define ENC_BITS_MASK 0xFFFFFu

}  // namespace cri
