// cri_dct_lane.h -- the 128-point DCT-IV of the HCA IMDCT (hca.cpp:1898-1980) on 16 lanes x 8 registers.
//
// A transform's 128 values live at physical position p = lane16 * 8 + reg.  In its in-place form (tools/gen_tables.py,
// imdct_inplace_maps) the DCT-IV is 14 butterfly stages between positions that differ in ONE bit of p: bits 0..2 are
// register pairs, bits 3..6 are lane16 ^ 1, 2, 4, 8.  Every multiply and add of the reference is kept as its own IEEE
// operation in the reference's order (no FMA contraction); the only fused form used is fma(x, +-1, y), whose product is
// exact, so it rounds exactly like the add it replaces.
//
// Arithmetic is issued as packed fp32 (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32: two lanes-worth of work per VALU
// issue slot); registers are paired along bit 0 (pair k holds reg 2k, 2k+1).
//
// Include inside `namespace cri`, after cri_device.h (f2, lane16_xor) and cri_imdct_tables.h (HCA_DCT_REGSIGN).
#pragma once

struct DctLane {
    float s[11], c[11];        // per-lane twiddles (rotation stages 0-4: one each, stage 5: two, stage 6: four); c of stages 0-3 is role-folded
    float sg[4];               // cross-lane sum/difference stage k: +1 in the `a` lane, -1 in the `b` lane (bit k of lane16)
};

__device__ __forceinline__ void dct_lane_init(DctLane& L, uint32_t l16, const float (*lane_sin)[11], const float (*lane_cos)[11]) {
#pragma unroll
    for (int i = 0; i < 11; i++) { L.s[i] = lane_sin[l16][i]; L.c[i] = lane_cos[l16][i]; }
    // rotation stage k exchanges over lane16 ^ (8 >> k): the `a` lane computes a*sin - b*cos, the `b` lane a*cos + b*sin
    L.c[0] = fneg_if(L.c[0], !(l16 & 8)); L.c[1] = fneg_if(L.c[1], !(l16 & 4)); L.c[2] = fneg_if(L.c[2], !(l16 & 2)); L.c[3] = fneg_if(L.c[3], !(l16 & 1));
#pragma unroll
    for (int k = 0; k < 4; k++) L.sg[k] = (l16 >> k) & 1 ? -1.0f : 1.0f;
}

#define DCT_SIGNED2(c, ST, k) f2{fneg_if((c), HCA_DCT_REGSIGN(ST, 2 * (k))), fneg_if((c), HCA_DCT_REGSIGN(ST, 2 * (k) + 1))}

template <int X, int K> __device__ __forceinline__ void sumdiff_cross(f2 p[4], const DctLane& L) {
    const f2 sg = {L.sg[K], L.sg[K]};
#pragma unroll
    for (int k = 0; k < 4; k++) p[k] = __builtin_elementwise_fma(p[k], sg, lane16_xor2<X>(p[k]));   // a+b in the `a` lane, a-b in the `b` lane
}
// partner(v) * (NEG ? -c : c) in ONE instruction where the exchange is a DPP pattern (lane16 ^ 1, 2, 8): v_mul_f32 takes the
// permuted operand itself, so the v_mov_b32_dpp that would feed a packed multiply disappears (a packed multiply cannot carry
// DPP).  The product is the same single IEEE multiply.
template <int X, bool NEG> __device__ __forceinline__ float mul_partner(float v, float c) {
    float r;
    if (X == 1) {
        if (NEG) asm("v_mul_f32_dpp %0, %1, -%2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(r) : "v"(v), "v"(c));
        else asm("v_mul_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(r) : "v"(v), "v"(c));
    } else if (X == 2) {
        if (NEG) asm("v_mul_f32_dpp %0, %1, -%2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(r) : "v"(v), "v"(c));
        else asm("v_mul_f32_dpp %0, %1, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(r) : "v"(v), "v"(c));
    } else {
        if (NEG) asm("v_mul_f32_dpp %0, %1, -%2 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(r) : "v"(v), "v"(c));
        else asm("v_mul_f32_dpp %0, %1, %2 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(r) : "v"(v), "v"(c));
    }
    return r;
}
template <int X, int ST> __device__ __forceinline__ void rotate_cross(f2 p[4], const DctLane& L) {
    const f2 sn = {L.s[ST], L.s[ST]};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const f2 p1 = p[k] * sn;
        f2 p2;                                                                     // -b*cos in the `a` lane, +a*cos in the `b` lane
#if defined(CRI_XOR8_SWIZZLE) || defined(CRI_DCT_NO_DPP_MUL)
        p2 = lane16_xor2<X>(p[k]) * DCT_SIGNED2(L.c[ST], ST, k);
#else
        if (X == 4) p2 = lane16_xor2<X>(p[k]) * DCT_SIGNED2(L.c[ST], ST, k);       // lane16 ^ 4 goes through the swizzle crossbar
        else {
            // (the register's sign is a constant once the loop is unrolled)
            p2.x = HCA_DCT_REGSIGN(ST, 2 * k) ? mul_partner<X, true>(p[k].x, L.c[ST]) : mul_partner<X, false>(p[k].x, L.c[ST]);
            p2.y = HCA_DCT_REGSIGN(ST, 2 * k + 1) ? mul_partner<X, true>(p[k].y, L.c[ST]) : mul_partner<X, false>(p[k].y, L.c[ST]);
        }
#endif
        p[k] = p1 + p2;
    }
}
// rotation between registers r and r | BIT, BIT = 4 or 2: both members of a pair rotate against the same other pair
template <int BIT, int ST> __device__ __forceinline__ void rotate_pairs(f2 p[4], const DctLane& L) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if ((2 * k) & BIT) continue;
        const int kb = k + BIT / 2, ti = ST == 4 ? 4 : 5 + (k >> 1);
        const f2 sn = {L.s[ti], L.s[ti]}, cs = DCT_SIGNED2(L.c[ti], ST, k);
        const f2 a = p[k], b = p[kb];
        const f2 as = a * sn, bc = b * cs, ac = a * cs, bs = b * sn;
        p[k] = as - bc;
        p[kb] = ac + bs;
    }
}
// rotation between the two registers of a pair
template <int ST> __device__ __forceinline__ void rotate_within(f2 p[4], const DctLane& L) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float sn = L.s[7 + k], cs = fneg_if(L.c[7 + k], HCA_DCT_REGSIGN(ST, 2 * k));
        const f2 u = p[k] * f2{sn, sn};                    // (a*sin, b*sin)
        const f2 w = p[k].yx * f2{cs, cs};                 // (b*cos, a*cos)
        p[k] = pk_add_neg_lo(u, w);                        // (a*sin - b*cos, b*sin + a*cos)
    }
}

__device__ __forceinline__ void dct4_inplace(f2 p[4], const DctLane& L) {
    // sum/difference stages 0..2: register bits 0, 1, 2
#pragma unroll
    for (int k = 0; k < 4; k++) p[k] = pk_sum_diff(p[k]);
    { const f2 a0 = p[0], a2 = p[2]; p[0] = a0 + p[1]; p[1] = a0 - p[1]; p[2] = a2 + p[3]; p[3] = a2 - p[3]; }
    { const f2 a0 = p[0], a1 = p[1]; p[0] = a0 + p[2]; p[2] = a0 - p[2]; p[1] = a1 + p[3]; p[3] = a1 - p[3]; }
    sumdiff_cross<1, 0>(p, L); sumdiff_cross<2, 1>(p, L); sumdiff_cross<4, 2>(p, L); sumdiff_cross<8, 3>(p, L);   // stages 3..6
    rotate_cross<8, 0>(p, L); rotate_cross<4, 1>(p, L); rotate_cross<2, 2>(p, L); rotate_cross<1, 3>(p, L);       // rotation stages 0..3
    rotate_pairs<4, 4>(p, L); rotate_pairs<2, 5>(p, L); rotate_within<6>(p, L);                                   // rotation stages 4..6
}
