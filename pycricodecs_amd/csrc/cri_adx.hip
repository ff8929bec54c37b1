// cri_adx.hip -- ADX kernels for gfx950, one lane per (file, channel) chain.
//   k_adx_decode     ChannelFrame::Decode /root/reference/CriCodecs/adx.cpp:189-214 + block loop 404-413
//   k_adx_encode     ChannelFrame::Encode adx.cpp:215-273 + frame loop 492-497
#include <hip/hip_runtime.h>
#include "cri_kernels.h"
#include "cri_device.h"
#include "../../include/cricodecs_hip.h"

#define CRI_TABLE_QUAL static __device__ const
#include "cri_tables.h"

namespace cri {
// ------------------------------------------------------------------------------------------------------------
// ADX: one lane per (file, channel) chain
// ------------------------------------------------------------------------------------------------------------
__global__ void k_adx_decode(AdxArgs a) {
    const uint32_t chain = blockIdx.x * blockDim.x + threadIdx.x;
    if (chain >= a.chains) return;
    const AdxStream S = a.streams[a.chain_stream[chain]];
    const uint32_t ch = chain - S.first_chain, C = S.channels, bs = S.blocksize, bd = S.bitdepth, spb = S.samples_per_block;
    int32_t h1 = a.history[2 * chain], h2 = a.history[2 * chain + 1];
    int32_t c0 = S.coef0, c1 = S.coef1;
    const uint8_t* src = a.in + S.src_offset;
    const uint8_t* end = a.in + S.src_end;
    int16_t* out = (int16_t*)(a.out + S.dst_offset);
    uint32_t done = 0;
    for (uint32_t fr = 0; fr < S.frames; fr++) {
        const uint8_t* fb = src + (uint64_t)fr * bs * C;
        if (fb + 2 > end) break;
        if (fb[0] == 0x80 && fb[1] == 0x01) break;              // EOF scale (adx.cpp:405-406)
        if (fb + (uint64_t)bs * C > end) break;
        const uint8_t* blk = fb + ch * bs;
        int32_t scale = ((int32_t)blk[0] << 8) | blk[1];
        if (S.mode == 4) scale = (int32_t)(1u << ((12 - scale) & 31));
        else if (S.mode == 2) {
            uint32_t pred = ((uint32_t)scale >> 13) & 7;
            scale = (scale & 0x1FFF) + 1;
            c0 = pred < 4 ? ADX_STATIC_COEFS[pred * 2] : 0;
            c1 = pred < 4 ? ADX_STATIC_COEFS[pred * 2 + 1] : 0;
        } else scale += 1;
        uint32_t acc = 0, have = 0, bytepos = 2;
        for (uint32_t s = 0; s < spb; s++) {
            while (have < bd) { acc = (acc << 8) | blk[bytepos++]; have += 8; }
            uint32_t raw = (acc >> (have - bd)) & ((1u << bd) - 1);
            have -= bd;
            int32_t v = (int32_t)(raw << (32 - bd)) >> (32 - bd);
            v = v * scale + ((c0 * h1) >> 12) + ((c1 * h2) >> 12);
            v = clamp_sym(v, 0x7FFF);
            const uint64_t idx = (uint64_t)fr * spb + s;
            if (idx < S.samples) out[idx * C + ch] = (int16_t)v;
            h2 = h1; h1 = (int32_t)(int16_t)v;
        }
        done = fr + 1;
    }
    // rows never reached (EOF marker / truncated input) decode to silence (the reference leaves them uninitialised)
    for (uint64_t idx = (uint64_t)done * spb; idx < S.samples; idx++) out[idx * C + ch] = 0;
}
void launch_adx_decode(const AdxArgs& a, hipStream_t s) {
    if (a.chains) hipLaunchKernelGGL(k_adx_decode, dim3((a.chains + 63) / 64), dim3(64), 0, s, a);
}

__device__ __forceinline__ int32_t adx_sample(const uint8_t* pcm, uint64_t idx, uint32_t C, uint32_t ch, uint32_t valid) {
    if (idx >= valid) return 0;                                  // zero padding of the last rows (adx.cpp:453-456)
    const uint8_t* p = pcm + (idx * C + ch) * 2;
    return (int32_t)(int16_t)(p[0] | (p[1] << 8));
}

__global__ void k_adx_encode(AdxArgs a) {
    const uint32_t chain = blockIdx.x * blockDim.x + threadIdx.x;
    if (chain >= a.chains) return;
    const AdxStream S = a.streams[a.chain_stream[chain]];
    const uint32_t ch = chain - S.first_chain, C = S.channels, bs = S.blocksize, bd = S.bitdepth, spb = S.samples_per_block;
    int32_t h1 = a.history[2 * chain], h2 = a.history[2 * chain + 1];
    const int32_t c0 = S.coef0, c1 = S.coef1, limit = (1 << (bd - 1)) - 1;
    const uint8_t* pcm = a.in + S.src_offset;
    uint8_t* dst = a.out + S.dst_offset;
    for (uint32_t fr = 0; fr < S.frames; fr++) {
        uint8_t* blk = dst + ((uint64_t)fr * C + ch) * bs;
        const uint64_t s0 = (uint64_t)fr * spb;
        // pass A: residual range with raw-sample history (adx.cpp:221-230)
        int32_t mn = 0, mx = 0, o1 = h1, o2 = h2;
        for (uint32_t i = 0; i < spb; i++) {
            int32_t x = adx_sample(pcm, s0 + i, C, ch, S.samples);
            int32_t r = ((int32_t)((uint32_t)x << 12) - c0 * h1 - c1 * h2) >> 12;
            mn = r < mn ? r : mn; mx = r > mx ? r : mx;
            h2 = h1; h1 = x;
        }
        if (!mn && !mx) { for (uint32_t i = 0; i < bs; i++) blk[i] = 0; continue; }   // adx.cpp:231-234
        int32_t qa = mx / limit, qb = mn / ~limit;
        uint32_t scale = (uint32_t)(qa > qb ? qa : qb) & 0xFFFF;
        if (scale > 0x1000) scale = 0x1000;
        uint32_t word;
        if (S.mode == 4) {
            uint32_t power = scale ? (32 - __clz((int)scale)) : 0;       // log2 + 1 (adx.cpp:241-244)
            scale = (1u << power) & 0xFFFF;
            word = (uint32_t)(12 - (int32_t)power) & 0xFFFF;
        } else if (S.mode == 2) word = (S.filter_bits | (scale & 0x1FFF)) & 0xFFFF;
        else word = scale;
        // first byte of a block is OR-ed into whatever the header writer left there (IO.cpp:139)
        uint8_t stale = 0;
        { uint64_t rel = (uint64_t)(blk - dst); if (rel < S.stale_len) stale = a.stale[S.stale_offset + rel]; }
        blk[0] = (uint8_t)(word >> 8) | stale; blk[1] = (uint8_t)word;
        h1 = o1; h2 = o2;
        uint32_t acc = 0, have = 0, bytepos = 2;
        for (uint32_t i = 0; i < spb; i++) {                       // pass B (adx.cpp:254-271)
            int32_t x = adx_sample(pcm, s0 + i, C, ch, S.samples);
            int32_t delta = ((int32_t)((uint32_t)x << 12) - c0 * h1 - c1 * h2) >> 12;
            if (!scale) scale = 1;
            delta = delta > 0 ? delta + (int32_t)(scale >> 1) : delta - (int32_t)(scale >> 1);
            delta /= (int32_t)scale;
            delta = clamp_sym(delta, limit);
            int32_t sim = (int32_t)(((uint32_t)delta << 12) * scale + (uint32_t)(c0 * h1) + (uint32_t)(c1 * h2)) >> 12;
            sim = clamp_sym(sim, 0x7FFF);
            h2 = h1; h1 = (int32_t)(int16_t)sim;
            acc = (acc << bd) | ((uint32_t)delta & ((1u << bd) - 1)); have += bd;
            while (have >= 8) { blk[bytepos++] = (uint8_t)(acc >> (have - 8)); have -= 8; }
        }
    }
}
void launch_adx_encode(const AdxArgs& a, hipStream_t s) {
    if (a.chains) hipLaunchKernelGGL(k_adx_encode, dim3((a.chains + 63) / 64), dim3(64), 0, s, a);
}

}  // namespace cri
