// cri_hca_enc_cost.h -- how k_hca_encode's rate loop costs a band without quantising it (tables: hca_enc_build_tables, cri_host.cpp;
// layout HCA_ET_CLS / HCA_ET_CP, cri_types.h).  Shared with the test build's exhaustive check against the reference's rule
// (cri_testing.hip; tests/test_gpu_round4.py::test_hca_encoder_band_cost_rule_on_the_device).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "cri_types.h"

namespace cri {

// class of a scaled spectrum (|v| < 1): how many of the fifteen resolutions' length thresholds of its sign it reaches.  cls =
// the HCA_ET_CLS rows: per binade of |v| (exponent field 114 .. 126; anything smaller: row 0) and sign {A, B, classes below}
__device__ __forceinline__ uint32_t enc_class(const uint4* cls, float v) {
    const uint32_t u = __float_as_uint(v);
    int e = (int)((u >> 23) & 0xFF) - 114;
    e = e < 0 ? 0 : e;                                     // (|v| < 1: at most 12)
    const uint4 row = cls[2 * e + (int)(u >> 31)];
    return row.z + (fabsf(v) >= __uint_as_float(row.x) ? 1u : 0u) + (fabsf(v) >= __uint_as_float(row.y) ? 1u : 0u);
}
__device__ __forceinline__ uint32_t enc_on_clamp(float v) { return __float_as_uint(v) == HCA_ENC_CLAMP_BITS ? 1u : 0u; }

// bits of a band's 8 spectra at the resolution of table row `row` (HCA_ET_CP: {(16 - rank) in every byte, 8 * shortest |
// anomaly << 8 | resolution << 16}) -- the inner part of CalculateUsedBits, hca.cpp:2771-2786.  cl = the spectra's classes, a byte
// each; ntop = how many of them sit on ScaleSpectra's clamp (only looked at when `tops`: the rare frame that has any)
__device__ __forceinline__ int enc_band_cost(uint2 row, uint32_t cl0, uint32_t cl1, uint32_t ntop, bool tops) {
    int n = (int)(row.y & 0xFF);
    n += __builtin_popcount((cl0 + row.x) & 0x10101010u) + __builtin_popcount((cl1 + row.x) & 0x10101010u);
    if (tops) n -= (row.y >> 8) & 1 ? (int)(ntop * (((row.y & 0xFF) >> 3) + 1)) : 0;
    return n;
}

}  // namespace cri
