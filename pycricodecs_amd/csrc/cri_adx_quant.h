// cri_adx_quant.h -- the two float forms of the ADX encoder's quantiser (adx.cpp:256-261: delta +- scale / 2, C division by
// scale, clamp to [~limit, limit]), shared by the kernels that use them (cri_adx.hip) and by the exhaustive check against the
// integer rule (cri_testing.hip, test build only: tests/test_gpu_round4.py::test_adx_float_quantisers_exhaustive).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cri {

struct AdxQuantSmall {          // per block: scale >= 1, limit = 2^(bitdepth - 1) - 1 <= 127
    float rcp, half_rcp; int32_t hs, cap, limit;
    __device__ __forceinline__ AdxQuantSmall(uint32_t scale, int32_t lim)
        : rcp(1.0f / (float)scale), half_rcp(0.5f * (1.0f / (float)scale)), hs((int32_t)(scale >> 1)), cap((lim + 2) * (int32_t)scale), limit(lim) {}
    // delta / scale is clamped to [~limit, limit] right after, so only quotients up to limit + 1 matter: with |delta| capped at
    // (limit + 2) * scale (exact in fp32 for bitdepths <= 8) floor((n + 0.5) / scale) comes out of one fma and a truncation -- the
    // 0.5 keeps exact quotients >= 1.2e-4 away from an integer, against an error below 129 * 2^-22
    __device__ __forceinline__ int32_t operator()(int32_t delta) const {
        const bool neg = delta < 0;
        int32_t an = (neg ? -delta : delta) + hs;
        an = an < cap ? an : cap;
        int32_t q = (int32_t)__builtin_fmaf((float)an, rcp, half_rcp);
        const int32_t qmax = neg ? limit + 1 : limit;
        q = q < qmax ? q : qmax;
        return neg ? -q : q;
    }
};

struct AdxQuantLane {           // bitdepth 4: sign(d) * floor((|d| + hs) / scale) clamped to [-8, 7], symmetric in the sign so that the
    float rcp, adj;             // chain from one sample to the next has no compare / select pair on it:
    __device__ __forceinline__ AdxQuantLane(uint32_t scale) : rcp(1.0f / (float)scale), adj(((float)(scale >> 1) + 0.5f) * (1.0f / (float)scale)) {}
    //   code = clamp(trunc(fma(d, 1 / scale, copysign((hs + 0.5) / scale, d))))
    // exact wherever it matters: for |d| + hs <= 9 * scale the quotient (|d| + hs + 0.5) / scale is at least 0.5 / 4096 = 1.2e-4 away
    // from an integer and the float error is below 3e-6; beyond, both sides are past +-8 and clamp alike
    __device__ __forceinline__ int32_t operator()(int32_t d) const {
        const float df = (float)d;
        const float sadj = __uint_as_float((__float_as_uint(adj) & 0x7FFFFFFFu) | (__float_as_uint(df) & 0x80000000u));
        int32_t code = (int32_t)__builtin_fmaf(df, rcp, sadj);
        return code > 7 ? 7 : (code < -8 ? -8 : code);
    }
};

// the reference's rule (adx.cpp:256-261), integer
__device__ __forceinline__ int32_t adx_quant_reference(int32_t delta, int32_t scale, int32_t limit) {
    delta = delta > 0 ? delta + (scale >> 1) : delta - (scale >> 1);
    delta /= scale;
    return delta > limit ? limit : (delta < ~limit ? ~limit : delta);
}

}  // namespace cri
