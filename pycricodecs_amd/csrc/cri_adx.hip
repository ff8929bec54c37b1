// cri_adx.hip -- ADX kernels for gfx950, one lane per (file, channel) chain, LDS-staged.
//   k_adx_decode     ChannelFrame::Decode /root/reference/CriCodecs/adx.cpp:189-214 + block loop 404-413
//   k_adx_encode     ChannelFrame::Encode adx.cpp:215-273 + frame loop 492-497
//
// The ADPCM recurrence  s = clamp(d*scale + (c0*h1 >> 12) + (c1*h2 >> 12))  has two floor shifts and a clamp inside the
// loop, so it is not an associative scan: the only exact parallelism is across chains.  A wave therefore owns 64 chains
// (32 stereo files) and advances them in lock step, T block rows per round:
//   1. the wave copies, file by file, the next T rows of compressed blocks (decode) / PCM (encode) from HBM into LDS with
//      fully coalesced 256-byte loads (the blocks of one file are contiguous: ch0,ch1,ch0,ch1,...),
//   2. every lane runs its chain over its T blocks out of LDS and leaves the result in LDS in the file's output layout
//      (interleaved PCM16 / interleaved blocks),
//   3. the wave copies the output regions to HBM, again 256 contiguous bytes per instruction.
// File regions in LDS are padded by 4 bytes each so that the lanes of a wave fall on different banks.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "cri_kernels.h"
#include "cri_device.h"
#include "cri_adx_quant.h"
#include "../../include/cricodecs_hip.h"

#define CRI_TABLE_QUAL static __device__ const
#include "cri_tables.h"

namespace cri {

// wave-cooperative copy HBM -> LDS of nbytes (LDS offset 4-byte aligned, global address arbitrary); bytes at or past
// `limit` read as zero
__device__ __forceinline__ void copy_in(uint8_t* lds, const uint8_t* src, const uint8_t* limit, uint32_t nbytes, uint32_t lane) {
    const uint32_t nd = (nbytes + 3) >> 2;
    for (uint32_t d = lane; d < nd; d += 64) {
        const uint8_t* p = src + 4 * (uint64_t)d;
        uint32_t v = 0;
        if (p + 4 <= limit) v = ld_u32_unaligned(p);
        else for (int k = 0; k < 4; k++) if (p + k < limit) v |= (uint32_t)p[k] << (8 * k);
        ((uint32_t*)lds)[d] = v;
    }
}
// wave-cooperative copy LDS -> HBM of nbytes (multiple of 2; dst 2-byte aligned at least)
__device__ __forceinline__ void copy_out(uint8_t* dst, const uint8_t* lds, uint32_t nbytes, uint32_t lane) {
    const uint32_t nd = nbytes >> 2;
    const bool al4 = (((uint64_t)dst) & 3) == 0;
    for (uint32_t d = lane; d < nd; d += 64) {
        const uint32_t v = ((const uint32_t*)lds)[d];
        if (al4) ((uint32_t*)dst)[d] = v;
        else { uint8_t* q = dst + 4 * (uint64_t)d; q[0] = (uint8_t)v; q[1] = (uint8_t)(v >> 8); q[2] = (uint8_t)(v >> 16); q[3] = (uint8_t)(v >> 24); }
    }
    for (uint32_t b = (nd << 2) + lane; b < nbytes; b += 64) dst[b] = lds[b];
}


// Stage-in of one round for the whole wave: the files' regions are fetched B files at a time with every load of a batch
// issued before the first LDS store, so a round pays a handful of memory latencies instead of one per file.
// lim_* / src_* / off_* / nb_* are per-lane values of the file LEADER lanes (read with readlane).
template <int B, int Q>
__device__ __forceinline__ void stage_in(uint8_t* lds_base, uint64_t leaders, uint32_t lane, uint64_t src_l, uint64_t lim_l, uint32_t off_l, uint32_t nbytes_l) {
    while (leaders) {
        int ls[B]; int n = 0;
#pragma unroll
        for (int j = 0; j < B; j++) { ls[j] = 0; if (leaders) { ls[j] = __builtin_ctzll(leaders); leaders &= leaders - 1; n = j + 1; } }
        uint32_t v[B][Q];
        bool big = false;
#pragma unroll
        for (int j = 0; j < B; j++) {
            if (j >= n) continue;
            const uint8_t* p = (const uint8_t*)readlane64(src_l, ls[j]);
            const uint8_t* lim = (const uint8_t*)readlane64(lim_l, ls[j]);
            const uint32_t nd = (__builtin_amdgcn_readlane(nbytes_l, ls[j]) + 3) >> 2;
            if (nd > 64 * Q) { big = true; continue; }
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const uint32_t d = lane + 64 * q;
                v[j][q] = 0;
                if (d < nd) {
                    const uint8_t* a = p + 4 * (uint64_t)d;
                    if (a + 4 <= lim) v[j][q] = ld_u32_unaligned(a);
                    else for (int k = 0; k < 4; k++) if (a + k < lim) v[j][q] |= (uint32_t)a[k] << (8 * k);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < B; j++) {
            if (j >= n) continue;
            const uint32_t nb = __builtin_amdgcn_readlane(nbytes_l, ls[j]), nd = (nb + 3) >> 2;
            uint8_t* l = lds_base + __builtin_amdgcn_readlane(off_l, ls[j]);
            if (nd > 64 * Q) { copy_in(l, (const uint8_t*)readlane64(src_l, ls[j]), (const uint8_t*)readlane64(lim_l, ls[j]), nb, lane); continue; }
#pragma unroll
            for (int q = 0; q < Q; q++) { const uint32_t d = lane + 64 * q; if (d < nd) ((uint32_t*)l)[d] = v[j][q]; }
        }
        (void)big;
    }
}

struct ChainCtx {
    AdxStream S; uint32_t ch; bool valid, leader;
    uint32_t in_off, out_off;       // LDS offsets of this chain's FILE regions
};

// LDS region offsets: prefix sums over the wave's files (leaders), +4 bytes of padding per region
__device__ __forceinline__ void plan_regions(ChainCtx& X, uint32_t in_bytes, uint32_t out_bytes, uint32_t lane, uint32_t& in_total, uint32_t& out_total) {
    const uint32_t ib = X.leader ? ((in_bytes + 3) & ~3u) + 4 : 0, ob = X.leader ? ((out_bytes + 3) & ~3u) + 4 : 0;
    uint32_t ipre = ib, opre = ob;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t ti = __shfl_up(ipre, o), to = __shfl_up(opre, o);
        if ((int)lane >= o) { ipre += ti; opre += to; }
    }
    in_total = __shfl(ipre, 63); out_total = __shfl(opre, 63);
    // a non-leader lane uses its file's leader offsets (leader = lane - ch)
    const uint32_t iex = ipre - ib, oex = opre - ob;
    X.in_off = __shfl(iex, (int)(lane - X.ch)); X.out_off = __shfl(oex, (int)(lane - X.ch));
}

__global__ __launch_bounds__(64) void k_adx_decode(AdxArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t lane = threadIdx.x, chain = blockIdx.x * 64 + lane, T = a.rows_per_round;
    ChainCtx X;
    const uint32_t sidx = chain < a.chains ? a.chain_stream[chain] : 0xFFFFFFFFu;   // the planner pads so that a file never straddles waves
    X.valid = sidx != 0xFFFFFFFFu;
    X.S = a.streams[X.valid ? sidx : 0];
    X.ch = X.valid ? chain - X.S.first_chain : 0;
    X.leader = X.valid && X.ch == 0;
    const AdxStream& S = X.S;
    const uint32_t C = S.channels, bs = S.blocksize, bd = S.bitdepth, spb = S.samples_per_block;
    uint32_t in_total, out_total;
    plan_regions(X, T * C * bs, T * spb * C * 2, lane, in_total, out_total);
    uint8_t* lin = smem;
    uint8_t* lout = smem + a.lds_in_bytes;
    int32_t h1 = X.valid ? a.history[2 * chain] : 0, h2 = X.valid ? a.history[2 * chain + 1] : 0;
    int32_t c0 = S.coef0, c1 = S.coef1;
    const uint8_t* src = a.in + S.src_offset;
    const uint8_t* end = a.in + S.src_end;
    uint8_t* dst = a.out + S.dst_offset;
    bool stopped = !X.valid;
    const bool std4 = __all(!X.valid || (bs == 18 && bd == 4 && S.mode != 4));   // mode 4 scales can exceed 24 bits
    uint32_t max_frames = X.valid ? S.frames : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t t = __shfl_xor(max_frames, o); max_frames = t > max_frames ? t : max_frames; }

    for (uint32_t f0 = 0; f0 < max_frames; f0 += T) {
        // 1. stage the next T rows of every file of this wave
        {
            const bool act = X.leader && f0 < S.frames;
            const uint32_t nrows = act ? (S.frames - f0 < T ? S.frames - f0 : T) : 0;
            stage_in<16, 2>(lin, __ballot(act), lane, (uint64_t)(src + (uint64_t)f0 * C * bs), (uint64_t)end, X.in_off, nrows * C * bs);
        }
        wave_lds_sync();
        // 2. every lane decodes its chain's T blocks out of LDS
        if (X.valid) {
            const uint8_t* fin = lin + X.in_off;
            int16_t* fo = (int16_t*)(lout + X.out_off);
            for (uint32_t t = 0; t < T && f0 + t < S.frames; t++) {
                const uint32_t fr = f0 + t;
                const uint8_t* row = fin + t * C * bs;
                const uint64_t row_global = (uint64_t)fr * C * bs;
                if (!stopped) {
                    // EOF scale on the row's first block, or a row the input does not fully contain (adx.cpp:405-406)
                    if (src + row_global + (uint64_t)bs * C > end || (row[0] == 0x80 && row[1] == 0x01)) stopped = true;
                }
                const uint8_t* blk = row + X.ch * bs;
                int32_t scale = ((int32_t)blk[0] << 8) | blk[1];
                if (S.mode == 4) scale = (int32_t)(1u << ((12 - scale) & 31));
                else if (S.mode == 2) {
                    const uint32_t pred = ((uint32_t)scale >> 13) & 7;
                    scale = (scale & 0x1FFF) + 1;
                    c0 = pred < 4 ? ADX_STATIC_COEFS[pred * 2] : 0;
                    c1 = pred < 4 ? ADX_STATIC_COEFS[pred * 2 + 1] : 0;
                } else scale += 1;
                int16_t* o = fo + (uint64_t)t * spb * C + X.ch;
                if (std4 && !__any(stopped)) {
                    // (the usual row: no chain of the wave has ended -- no per-sample selects)
#pragma unroll
                    for (uint32_t w = 0; w < 8; w++) {
                        const uint32_t wd = ((uint32_t)blk[2 + 2 * w] << 8) | blk[3 + 2 * w];
#pragma unroll
                        for (uint32_t k = 0; k < 4; k++) {
                            const int32_t code = (int32_t)(wd << (16 + 4 * k)) >> 28;
                            const int32_t v = clamp_sym(__mul24(code, scale) + (__mul24(c0, h1) >> 12) + (__mul24(c1, h2) >> 12), 0x7FFF);
                            h2 = h1; h1 = v;
                            o[(uint64_t)(4 * w + k) * C] = (int16_t)v;
                        }
                    }
                    continue;
                }
                if (std4) {
                    // blocksize 18 / bitdepth 4: 8 big-endian 16-bit words of four codes each; 24-bit multiplies are exact here
                    // (|code| < 2^3, scale < 2^14, |coef| < 2^13, |history| < 2^15)
#pragma unroll
                    for (uint32_t w = 0; w < 8; w++) {
                        const uint32_t wd = ((uint32_t)blk[2 + 2 * w] << 8) | blk[3 + 2 * w];
#pragma unroll
                        for (uint32_t k = 0; k < 4; k++) {
                            const int32_t code = (int32_t)(wd << (16 + 4 * k)) >> 28;
                            int32_t v = __mul24(code, scale) + (__mul24(c0, h1) >> 12) + (__mul24(c1, h2) >> 12);
                            v = clamp_sym(v, 0x7FFF);
                            if (stopped) v = 0; else { h2 = h1; h1 = v; }
                            o[(uint64_t)(4 * w + k) * C] = (int16_t)v;
                        }
                    }
                    continue;
                }
                uint32_t acc = 0, have = 0, bytepos = 2;
                for (uint32_t s = 0; s < spb; s++) {
                    while (have < bd) { acc = (acc << 8) | blk[bytepos++]; have += 8; }
                    const uint32_t raw = (acc >> (have - bd)) & ((1u << bd) - 1);
                    have -= bd;
                    int32_t v = (int32_t)(raw << (32 - bd)) >> (32 - bd);
                    v = v * scale + ((c0 * h1) >> 12) + ((c1 * h2) >> 12);
                    v = clamp_sym(v, 0x7FFF);
                    if (stopped) v = 0;                                   // rows never reached decode to silence
                    else { h2 = h1; h1 = v; }
                    o[(uint64_t)s * C] = (int16_t)v;
                }
            }
        }
        wave_lds_sync();
        // 3. copy the decoded rows out, clipped to the stream's sample count
        for (uint32_t l = 0; l < 64; l++) {
            if (!__builtin_amdgcn_readlane((int)X.leader, l)) continue;
            const uint32_t frames_l = __builtin_amdgcn_readlane(S.frames, l);
            if (f0 >= frames_l) continue;
            const uint32_t spb_l = __builtin_amdgcn_readlane(spb, l), C_l = __builtin_amdgcn_readlane(C, l), samples_l = __builtin_amdgcn_readlane(S.samples, l);
            const uint32_t nrows = frames_l - f0 < T ? frames_l - f0 : T;
            uint64_t s0 = (uint64_t)f0 * spb_l, s1 = s0 + (uint64_t)nrows * spb_l;
            if (s1 > samples_l) s1 = samples_l;
            if (s1 <= s0) continue;
            uint8_t* q = (uint8_t*)readlane64((uint64_t)dst, l) + s0 * C_l * 2;
            copy_out(q, lout + __builtin_amdgcn_readlane(X.out_off, l), (uint32_t)((s1 - s0) * C_l * 2), lane);
        }
        wave_lds_sync();
    }
}

void launch_adx_decode(const AdxArgs& a, hipStream_t s) {
    if (!a.chains) return;
    const uint32_t lds = a.lds_in_bytes + a.lds_out_bytes;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k_adx_decode, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_adx_decode, dim3((a.chains + 63) / 64), dim3(64), lds, s, a);
}

__global__ __launch_bounds__(64) void k_adx_encode(AdxArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t lane = threadIdx.x, chain = blockIdx.x * 64 + lane, T = a.rows_per_round;
    ChainCtx X;
    const uint32_t sidx = chain < a.chains ? a.chain_stream[chain] : 0xFFFFFFFFu;
    X.valid = sidx != 0xFFFFFFFFu;
    X.S = a.streams[X.valid ? sidx : 0];
    X.ch = X.valid ? chain - X.S.first_chain : 0;
    X.leader = X.valid && X.ch == 0;
    const AdxStream& S = X.S;
    const uint32_t C = S.channels, bs = S.blocksize, bd = S.bitdepth, spb = S.samples_per_block;
    uint32_t in_total, out_total;
    plan_regions(X, T * spb * C * 2, T * C * bs, lane, in_total, out_total);
    uint8_t* lin = smem;
    uint8_t* lout = smem + a.lds_in_bytes;
    int32_t h1 = X.valid ? a.history[2 * chain] : 0, h2 = X.valid ? a.history[2 * chain + 1] : 0;
    const int32_t c0 = S.coef0, c1 = S.coef1, limit = (1 << (bd - 1)) - 1;
    const uint8_t* pcm = (S.src_in_scratch ? a.scratch : a.in) + S.src_offset;
    const uint8_t* pcm_end = pcm + (uint64_t)S.samples * C * 2;       // samples past the input are zero padding (adx.cpp:453-456)
    uint8_t* dst = a.out + S.dst_offset;
    uint32_t max_frames = X.valid ? S.frames : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t t = __shfl_xor(max_frames, o); max_frames = t > max_frames ? t : max_frames; }

    for (uint32_t f0 = 0; f0 < max_frames; f0 += T) {
        {
            const bool act = X.leader && f0 < S.frames;
            const uint32_t nrows = act ? (S.frames - f0 < T ? S.frames - f0 : T) : 0;
            stage_in<4, 6>(lin, __ballot(act), lane, (uint64_t)(pcm + (uint64_t)f0 * spb * C * 2), (uint64_t)pcm_end, X.in_off, nrows * spb * C * 2);
        }
        wave_lds_sync();
        if (X.valid) {
            const int16_t* fin = (const int16_t*)(lin + X.in_off);
            uint8_t* fo = lout + X.out_off;
            for (uint32_t t = 0; t < T && f0 + t < S.frames; t++) {
                const int16_t* x = fin + (uint64_t)t * spb * C + X.ch;
                uint8_t* blk = fo + (t * C + X.ch) * bs;
                // pass A: residual range with raw-sample history (adx.cpp:221-230)
                int32_t mn = 0, mx = 0;
                const int32_t o1 = h1, o2 = h2;
                for (uint32_t i = 0; i < spb; i++) {
                    const int32_t v = x[(uint64_t)i * C];
                    const int32_t r = ((int32_t)((uint32_t)v << 12) - c0 * h1 - c1 * h2) >> 12;
                    mn = r < mn ? r : mn; mx = r > mx ? r : mx;
                    h2 = h1; h1 = v;
                }
                if (!mn && !mx) { for (uint32_t i = 0; i < bs; i++) blk[i] = 0; continue; }   // adx.cpp:231-234
                const int32_t qa = mx / limit, qb = mn / ~limit;
                uint32_t scale = (uint32_t)(qa > qb ? qa : qb) & 0xFFFF;
                if (scale > 0x1000) scale = 0x1000;
                uint32_t word;
                if (S.mode == 4) {
                    const uint32_t power = scale ? (32 - __clz((int)scale)) : 0;      // log2 + 1 (adx.cpp:241-244)
                    scale = (1u << power) & 0xFFFF;
                    word = (uint32_t)(12 - (int32_t)power) & 0xFFFF;
                } else if (S.mode == 2) word = (S.filter_bits | (scale & 0x1FFF)) & 0xFFFF;
                else word = scale;
                // first byte of a block is OR-ed into whatever the header writer left there (IO.cpp:139)
                uint8_t stale = 0;
                { const uint64_t rel = ((uint64_t)(f0 + t) * C + X.ch) * bs; if (rel < S.stale_len) stale = a.stale[S.stale_offset + rel]; }
                blk[0] = (uint8_t)(word >> 8) | stale; blk[1] = (uint8_t)word;
                h1 = o1; h2 = o2;
                if (!scale) scale = 1;
                // (bitdepths <= 8: the float form of the quantiser, cri_adx_quant.h -- the same shortening of the serial chain as in
                //  k_adx_encode_wpf)
                const bool small = limit <= 127;
                const AdxQuantSmall quant(scale, limit);
                const float rcp = quant.rcp;
                const int32_t hs = (int32_t)(scale >> 1), iscale = (int32_t)scale;
                uint32_t acc = 0, have = 0, bytepos = 2;
                for (uint32_t i = 0; i < spb; i++) {                       // pass B (adx.cpp:254-271)
                    const int32_t v = x[(uint64_t)i * C];
                    const int32_t pred = __mul24(c0, h1) + __mul24(c1, h2);
                    int32_t delta = ((int32_t)((uint32_t)v << 12) - pred) >> 12;
                    if (small) delta = quant(delta);
                    else {
                        delta = delta > 0 ? delta + hs : delta - hs;
                        // delta /= scale (truncating), exact: float estimate + correction; |delta| < 2^22 here
                        const uint32_t an = (uint32_t)(delta < 0 ? -delta : delta);
                        uint32_t q;
                        if (an < (1u << 22)) {
                            q = (uint32_t)((float)an * rcp);
                            int32_t r = (int32_t)an - (int32_t)(q * scale);
                            if (r < 0) { q--; r += (int32_t)scale; }
                            if (r >= (int32_t)scale) q++;
                        } else q = an / scale;
                        delta = delta < 0 ? -(int32_t)q : (int32_t)q;
                        delta = clamp_sym(delta, limit);
                    }
                    int32_t sim = (int32_t)(((uint32_t)__mul24(delta, iscale) << 12) + (uint32_t)pred) >> 12;
                    sim = clamp_sym(sim, 0x7FFF);
                    h2 = h1; h1 = sim;
                    acc = (acc << bd) | ((uint32_t)delta & ((1u << bd) - 1)); have += bd;
                    while (have >= 8) { blk[bytepos++] = (uint8_t)(acc >> (have - 8)); have -= 8; }
                }
            }
        }
        wave_lds_sync();
        for (uint32_t l = 0; l < 64; l++) {
            if (!__builtin_amdgcn_readlane((int)X.leader, l)) continue;
            const uint32_t frames_l = __builtin_amdgcn_readlane(S.frames, l);
            if (f0 >= frames_l) continue;
            const uint32_t rowb = __builtin_amdgcn_readlane(C * bs, l);
            const uint32_t nrows = frames_l - f0 < T ? frames_l - f0 : T;
            uint8_t* q = (uint8_t*)readlane64((uint64_t)dst, l) + (uint64_t)f0 * rowb;
            copy_out(q, lout + __builtin_amdgcn_readlane(X.out_off, l), nrows * rowb, lane);
        }
        wave_lds_sync();
    }
}

void launch_adx_encode(const AdxArgs& a, hipStream_t s) {
    if (!a.chains) return;
    const uint32_t lds = a.lds_in_bytes + a.lds_out_bytes;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k_adx_encode, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_adx_encode, dim3((a.chains + 63) / 64), dim3(64), lds, s, a);
}


// ------------------------------------------------------------------------------------------------------------
// Wave-per-file variants for batches with few chains (BASELINE configs[1]: 1 000 files = 2 000 chains would keep 32 of
// 1 024 SIMDs busy in the lane-per-chain mapping).  Standard layout only: blocksize 18, bitdepth 4, 1 or 2 channels, so
// that one ADX frame is one wavefront: lanes 0-31 = the 32 samples of channel 0's block, lanes 32-63 = channel 1's.
// The per-sample parts (nibble unpack, code*scale; encoder pass A residuals + min/max) run across the lanes; the encoder
// walks its recurrence sample by sample with v_readlane broadcasts, identically in every lane of a half, the decoder hands
// a frame's products through LDS to registers (see k_adx_decode_wpf).
// ------------------------------------------------------------------------------------------------------------
// min / max over the 32 lanes of a half, the result in every lane: xor 1, 2 as quad permutes, xor 4 and 8 as the mirrors of 8 and
// 16 lanes (lanes of a reduced group already agree), xor 16 through the swizzle crossbar -- DPP modifiers on the min / max
// themselves, so the five steps cost five dependent VALU instructions and one crossbar hop instead of five LDS round trips
template <bool MAX>
__device__ __forceinline__ int32_t half_reduce(int32_t v) {
    auto op = [](int32_t a, int32_t b) { return MAX ? (a > b ? a : b) : (a < b ? a : b); };
    v = op(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true));       // quad_perm [1,0,3,2]
    v = op(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true));       // quad_perm [2,3,0,1]
    v = op(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true));      // row_half_mirror
    v = op(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true));      // row_mirror
    v = op(v, __builtin_amdgcn_ds_swizzle(v, 0x401F));                        // lane ^ 16 (bit mode: and 0x1F, xor 0x10)
    return v;
}
__device__ __forceinline__ int32_t half_min(int32_t v) { return half_reduce<false>(v); }
__device__ __forceinline__ int32_t half_max(int32_t v) { return half_reduce<true>(v); }

// Decode.  A wave per file runs alone on its SIMD, so every dependent instruction and every memory round trip is paid in
// full; the kernel is organised around that:
//  * input: coalesced dwords of a chunk of K frames, requested two chunks ahead and landed in LDS one chunk ahead.  vmcnt
//    counts loads and stores together, in order, so the wait for a chunk also waits for the stores issued just before it
//    (microseconds for a lone wave): large chunks make that rare;
//  * a round of R frames: code * scale across the lanes into LDS, then the first lane of each half walks its channel's
//    recurrence on registers -- d + (c0*h1 >> 12) + (c1*h2 >> 12) = (c0*h1 + ((d + (c1*h2 >> 12)) << 12)) >> 12, whose
//    second term does not depend on the newest sample: 6 VALU instructions per sample, the dependent chain is
//    multiply-add, shift, clamp -- and the samples go back through LDS for one coalesced store per frame.
__global__ __launch_bounds__(64) void k_adx_decode_wpf(AdxArgs a) {
    constexpr int R = 4;                                                // frames per round
    constexpr uint32_t K = 64, KW = K * 36 / 4 / 64;                    // frames per input chunk; dwords per lane of a chunk (9)
    __shared__ __attribute__((aligned(16))) int32_t dl[R][2][32];      // [frame][half][sample] code * scale
    __shared__ __attribute__((aligned(16))) int32_t ol[R][2][32];      // [frame][half][sample] decoded samples
    __shared__ __attribute__((aligned(16))) uint8_t stage[2][K * 36];
    // As the segmented decoder's fallback (a.seg_flags set) the launch is a fixed number of workgroups that walk k_adx_seg_list's list of
    // the files with a flagged chain -- usually empty: a workgroup per file that looked at its flags and left cost the 100 000-clip bank
    // 1.4 ms of every step.
    const uint32_t n_mine = a.seg_flags ? a.seg_flags[a.chains] : gridDim.x;
    for (uint32_t li = blockIdx.x; li < n_mine; li += gridDim.x) {
    const AdxStream S = a.streams[a.seg_flags ? a.seg_flags[a.chains + 1 + li] : (a.wpf_order ? a.wpf_order[li] : li)];
    const uint32_t lane = threadIdx.x, half = lane >> 5, s = lane & 31, C = S.channels;
    const bool act = half < C;
    const uint32_t chain = S.first_chain + (act ? half : 0);
    int32_t h1 = a.history[2 * chain], h2 = a.history[2 * chain + 1];
    const uint8_t* src = a.in + S.src_offset;
    const uint8_t* end = a.in + S.src_end;
    uint8_t* out = a.out + S.dst_offset;
    const uint32_t rowb = 18 * C;
    const uint32_t chunk_bytes = K * rowb, chunk_dwords = (chunk_bytes + 3) / 4;
    uint32_t w[KW], wsh[KW];                               // dwords in flight; right shift that drops the bytes before a dword re-based at the stream end (32: nothing to keep)
    auto request = [&](uint32_t chunk) {                   // branch-free: the loads only leave, nothing here waits for them
        const uint8_t* base = src + (uint64_t)chunk * chunk_bytes;
#pragma unroll
        for (uint32_t i = 0; i < KW; i++) {
            const uint32_t d = i * 64 + lane;
            const uint8_t* p = base + 4 * d;
            const bool some = d < chunk_dwords && p < end, whole = some && p + 4 <= end;
            wsh[i] = whole ? 0u : (some ? 8u * (4u - (uint32_t)(end - p)) : 32u);
            w[i] = ld_u32_unaligned(whole ? p : end - 4);  // (the stream's last dword when the wanted one crosses its end; block data follows a header, so end - 4 is inside the blob)
        }
    };
    auto land = [&](uint32_t chunk) {
#pragma unroll
        for (uint32_t i = 0; i < KW; i++) { const uint32_t d = i * 64 + lane; if (d < chunk_dwords) ((uint32_t*)stage[chunk & 1])[d] = wsh[i] < 32 ? w[i] >> wsh[i] : 0u; }
    };
    auto fetch = [&](uint32_t fr, uint32_t& b, uint32_t& sc) {           // this lane's code byte and its block's scale word of frame fr
        // (unconditional LDS reads, selected afterwards: the four frames of a round then wait for them once)
        const uint8_t* row = src + (uint64_t)fr * rowb;
        const bool ok = fr < S.frames && row + rowb <= end && act;
        const uint8_t* blk = stage[(fr / K) & 1] + (fr % K) * rowb + (act ? half : 0) * 18;
        const uint32_t bb = blk[2 + (s >> 1)], s0 = blk[0], s1 = blk[1];
        b = ok ? bb : 0u;
        sc = ok ? ((s0 << 8) | s1) : 0u;
    };
    request(0); land(0); request(1);
    wave_lds_sync();
    uint32_t done = 0;
    bool stopped = false;
    for (uint32_t f0 = 0; f0 < S.frames && !stopped; f0 += R) {
        if (f0 % K == 0) {                                 // entering chunk f0 / K: the next one lands, the one after is requested
            land(f0 / K + 1); request(f0 / K + 2);
            wave_lds_sync();
        }
        uint32_t b[R], sc[R];
#pragma unroll
        for (int t = 0; t < R; t++) fetch(f0 + t, b[t], sc[t]);
        // The usual round -- four whole frames inside the stream and the sample count, constant coefficients, scale = word + 1
        // small, no EOF marker -- runs as straight-line code: a lone wave pays for every branch and every separate LDS wait.
        {
            bool full = S.mode == 3 && f0 + R <= S.frames && src + (uint64_t)(f0 + R) * rowb <= end && (uint64_t)(f0 + R) * 32 <= S.samples;
#pragma unroll
            for (int t = 0; t < R; t++) full = full && __builtin_amdgcn_readlane(sc[t], 0) != 0x8001 && __all(sc[t] < 0x9000);
            if (full) {
#pragma unroll
                for (int t = 0; t < R; t++) {
                    const int32_t code = (int32_t)((s & 1 ? b[t] << 28 : b[t] << 24) & 0xF0000000u) >> 28;
                    dl[t][half][s] = (code * ((int32_t)sc[t] + 1)) << 12;      // (|code * scale| < 2^19: shifted here, by all lanes at once, not on the chain)
                }
                wave_lds_sync();
                const int32_t c0 = S.coef0, c1 = S.coef1;
#pragma unroll
                for (int t = 0; t < R; t++) {
                    int32_t d[32];
#pragma unroll
                    for (int k = 0; k < 32; k += 4) { const int4 q = *(const int4*)&dl[t][half][k]; d[k] = q.x; d[k + 1] = q.y; d[k + 2] = q.z; d[k + 3] = q.w; }
                    int32_t v1 = h1, v2 = h2;
                    // (d + (c1*h2 >> 12)) << 12 = (d << 12) + (c1*h2 with its low 12 bits cleared): five instructions per sample
                    // (the sum's low 12 bits are the product's: clearing them after the add is the same, and one multiply-add)
                    int32_t pre = (int32_t)((uint32_t)(__mul24(c1, v2) + d[0]) & 0xFFFFF000u);
#pragma unroll
                    for (int k = 0; k < 32; k++) {
                        const int32_t v = clamp_sym((__mul24(c0, v1) + pre) >> 12, 0x7FFF);
                        if (k < 31) pre = (int32_t)((uint32_t)(__mul24(c1, v1) + d[k + 1]) & 0xFFFFF000u);
                        d[k] = v;
                        v2 = v1; v1 = v;
                    }
                    h1 = d[31]; h2 = d[30];
                    if (s == 0) {
#pragma unroll
                        for (int k = 0; k < 32; k += 4) *(int4*)&ol[t][half][k] = make_int4(d[k], d[k + 1], d[k + 2], d[k + 3]);
                    }
                }
                wave_lds_sync();
                int32_t m0[R], m1[R];
#pragma unroll
                for (int t = 0; t < R; t++) { m0[t] = ol[t][0][s]; m1[t] = ol[t][1][s]; }
                if (half == 0) {
                    if (C == 2) {
#pragma unroll
                        for (int t = 0; t < R; t++) ((uint32_t*)out)[(uint64_t)(f0 + t) * 32 + s] = ((uint32_t)m0[t] & 0xFFFF) | ((uint32_t)m1[t] << 16);
                    } else {
#pragma unroll
                        for (int t = 0; t < R; t++) ((int16_t*)out)[(uint64_t)(f0 + t) * 32 + s] = (int16_t)m0[t];
                    }
                }
                done = f0 + R;
                wave_lds_sync();
                continue;
            }
        }
        // per-sample part of the round's frames: code * scale into LDS; the round ends at an EOF marker / truncated row
        int32_t c0t[R], c1t[R];
        bool fastt[R];
        uint32_t nv = 0;
#pragma unroll
        for (int t = 0; t < R; t++) {
            c0t[t] = S.coef0; c1t[t] = S.coef1; fastt[t] = false;
            const uint32_t fr = f0 + t;
            if (fr >= S.frames || stopped) continue;
            const uint8_t* row = src + (uint64_t)fr * rowb;
            const uint32_t sc0 = __builtin_amdgcn_readlane(sc[t], 0);            // channel 0's scale word: EOF marker check (adx.cpp:405-406)
            if (row + rowb > end || sc0 == 0x8001) { stopped = true; continue; }
            int32_t scale = (int32_t)sc[t];
            if (S.mode == 4) scale = (int32_t)(1u << ((12 - scale) & 31));
            else if (S.mode == 2) {
                const uint32_t pred = ((uint32_t)scale >> 13) & 7;
                scale = (scale & 0x1FFF) + 1;
                c0t[t] = pred < 4 ? ADX_STATIC_COEFS[pred * 2] : 0;
                c1t[t] = pred < 4 ? ADX_STATIC_COEFS[pred * 2 + 1] : 0;
            } else scale += 1;
            const int32_t code = (int32_t)((s & 1 ? b[t] << 28 : b[t] << 24) & 0xF0000000u) >> 28;
            dl[t][half][s] = code * scale;
            fastt[t] = __all(scale >= 0 && scale <= 0x9000);
            nv = t + 1;
        }
        wave_lds_sync();
        // the recurrence (adx.cpp:198-214) on registers: every lane of a half walks the same values, lanes 0 and 32 keep them
#pragma unroll
        for (int t = 0; t < R; t++) {
            if ((uint32_t)t >= nv) break;
            const int32_t c0 = c0t[t], c1 = c1t[t];
            int32_t d[32];
#pragma unroll
            for (int k = 0; k < 32; k += 4) { const int4 q = *(const int4*)&dl[t][half][k]; d[k] = q.x; d[k + 1] = q.y; d[k + 2] = q.z; d[k + 3] = q.w; }
            if (fastt[t]) {                                // |d| <= 8 * 0x9000 keeps the multiply-add form inside 32 bits
                int32_t v1 = h1, v2 = h2;
                int32_t pre = (d[0] + (__mul24(c1, v2) >> 12)) << 12;
#pragma unroll
                for (int k = 0; k < 32; k++) {
                    const int32_t v = clamp_sym((__mul24(c0, v1) + pre) >> 12, 0x7FFF);
                    if (k < 31) pre = (d[k + 1] + (__mul24(c1, v1) >> 12)) << 12;
                    d[k] = v;
                    v2 = v1; v1 = v;
                }
            } else {
                // prediction terms: p1 = c0*h1 >> 12 and p2 = c1*h2 >> 12 of the next sample; r1 = c1*h1 >> 12 becomes p2 one step later
                int32_t p1 = __mul24(c0, h1) >> 12, p2 = __mul24(c1, h2) >> 12, r1 = __mul24(c1, h1) >> 12;
#pragma unroll
                for (int k = 0; k < 32; k++) {
                    const int32_t v = clamp_sym(d[k] + p1 + p2, 0x7FFF);
                    p2 = r1;
                    p1 = __mul24(c0, v) >> 12;
                    r1 = __mul24(c1, v) >> 12;
                    d[k] = v;
                }
            }
            h1 = d[31]; h2 = d[30];
            if (s == 0) {
#pragma unroll
                for (int k = 0; k < 32; k += 4) *(int4*)&ol[t][half][k] = make_int4(d[k], d[k + 1], d[k + 2], d[k + 3]);
            }
        }
        wave_lds_sync();
        // (all of the round's samples are read back before the first store: one LDS latency per round, not two per frame)
        int32_t m0[R], m1[R];
#pragma unroll
        for (int t = 0; t < R; t++) { m0[t] = ol[t][0][s]; m1[t] = ol[t][1][s]; }
#pragma unroll
        for (int t = 0; t < R; t++) {
            if ((uint32_t)t >= nv) break;
            const uint64_t idx = (uint64_t)(f0 + t) * 32 + s;
            if (half == 0 && idx < S.samples) {
                if (C == 2) ((uint32_t*)out)[idx] = ((uint32_t)m0[t] & 0xFFFF) | ((uint32_t)m1[t] << 16);
                else ((int16_t*)out)[idx] = (int16_t)m0[t];
            }
        }
        done = f0 + nv;
        wave_lds_sync();
    }
    // rows never reached (EOF marker / truncated input) decode to silence
    for (uint64_t i = (uint64_t)done * 32 * C + lane; i < (uint64_t)S.samples * C; i += 64) ((int16_t*)out)[i] = 0;
    wave_lds_sync();
    }
}
// the files the fallback decodes again: mono / stereo files with a flagged chain (seg_flags[chains] = their number, the list behind it)
__global__ void k_adx_seg_list(AdxArgs a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_streams) return;
    const uint32_t C = a.streams[i].channels, fc = a.streams[i].first_chain;
    if (C > 2 || !(a.seg_flags[fc] | (C == 2 ? a.seg_flags[fc + 1] : 0u))) return;
    a.seg_flags[a.chains + 1 + atomicAdd(&a.seg_flags[a.chains], 1u)] = i;
}

// a chain's state between two rows: its two history samples
__device__ __forceinline__ uint32_t seg_pack(int32_t h1, int32_t h2) { return ((uint32_t)h1 & 0xFFFFu) | ((uint32_t)h2 << 16); }
__device__ __forceinline__ void seg_unpack(uint32_t s, int32_t& h1, int32_t& h2) { h1 = (int32_t)(int16_t)(s & 0xFFFF); h2 = (int32_t)(int16_t)(s >> 16); }

// Rows [row_begin, row_end) of stream S by one wave: lanes 0-31 = channel 0's block, 32-63 = channel 1's.  Rows below
// `store_from` are encoded but not stored (the warm-up of a segment, see the segmented kernels below).  h1 / h2: the half's
// history, the same in every lane of a half.  after_round(next_row) runs after every round of R rows -- rounds start at
// row_begin, R apart -- and ends the walk when it returns true (wave-uniform).
#define ADX_WPF_R 4
template <int C, class AfterRound>
__device__ __forceinline__ void adx_wpf_rows(const AdxArgs& a, const AdxStream& S, uint32_t row_begin, uint32_t row_end, uint32_t store_from,
                                             int32_t& h1, int32_t& h2, uint8_t* blk_img, int32_t* xl, AfterRound&& after_round) {
    const uint32_t lane = threadIdx.x, half = lane >> 5, s = lane & 31;
    const bool act = half < C;
    const int32_t c0 = S.coef0, c1 = S.coef1;
    const uint8_t* pcm = (S.src_in_scratch ? a.scratch : a.in) + S.src_offset;
    uint8_t* dst = a.out + S.dst_offset;
    constexpr int R = ADX_WPF_R;
    int32_t nx[R];
    auto fetch = [&](uint32_t fr, int32_t& x) {
        x = 0;
        const uint64_t idx = (uint64_t)fr * 32 + s;
        if (fr < row_end && idx < S.samples && act) { const uint8_t* p = pcm + (idx * C + half) * 2; x = (int32_t)(int16_t)(p[0] | (p[1] << 8)); }
    };
#pragma unroll
    for (int t = 0; t < R; t++) fetch(row_begin + (uint32_t)t, nx[t]);
    for (uint32_t f0 = row_begin; f0 < row_end; f0 += R) {
        int32_t xr[R];
#pragma unroll
        for (int t = 0; t < R; t++) xr[t] = nx[t];
#pragma unroll
        for (int t = 0; t < R; t++) fetch(f0 + R + t, nx[t]);
#pragma unroll
        for (int t = 0; t < R; t++) {
            const uint32_t fr = f0 + t;
            if (fr >= row_end) break;
            const int32_t x = xr[t];
            // pass A (adx.cpp:221-230): residual against the two previous RAW samples; lanes 0/1 of a block see the carried history
            // (the previous lanes' samples: wave_shr:1 twice; what the first two lanes of a half receive is replaced just below)
            int32_t p1 = __builtin_amdgcn_update_dpp(0, x, 0x138, 0xF, 0xF, false), p2 = __builtin_amdgcn_update_dpp(0, p1, 0x138, 0xF, 0xF, false);
            if (s == 0) { p1 = h1; p2 = h2; } else if (s == 1) p2 = h1;
            const int32_t r = ((int32_t)((uint32_t)x << 12) - c0 * p1 - c1 * p2) >> 12;
            int32_t mn = half_min(r < 0 ? r : 0), mx = half_max(r > 0 ? r : 0);
            const bool silent = !mn && !mx;                                            // adx.cpp:231-234
            const int32_t qa = mx / 7, qb = (int32_t)((uint32_t)(-mn) >> 3);           // Maximum/Limit, Minimum/~Limit with Limit = 7
            uint32_t scale = (uint32_t)(qa > qb ? qa : qb) & 0xFFFF;
            if (scale > 0x1000) scale = 0x1000;
            uint32_t word;
            if (S.mode == 4) {
                const uint32_t power = scale ? (32 - __clz((int)scale)) : 0;
                scale = (1u << power) & 0xFFFF;
                word = (uint32_t)(12 - (int32_t)power) & 0xFFFF;
            } else if (S.mode == 2) word = (S.filter_bits | (scale & 0x1FFF)) & 0xFFFF;
            else word = scale;
            if (!scale) scale = 1;
            // pass B (adx.cpp:254-271) is one serial chain of 32 steps per block, so its length in dependent instructions is
            // what the kernel's time is made of.  The quantiser -- delta +- scale/2, C division by scale, clamp to [-8, 7] -- is
            // a monotone step function of delta with 15 steps, so the code is a COUNT: lanes 0..14 of a half hold the
            // thresholds  delta >= -(j*scale - hs) + 1 (j = 8..1: code > -j)  and  delta >= j*scale - hs (j = 1..7: code >= j),
            // scaled by 4096 so that they apply to (x << 12) - prediction before its shift, and
            //     code = popcount(compare mask of the half) - 8.
            // The simulated sample is code * scale + (prediction >> 12) exactly ((code * scale) << 12 has no low bits).  Per
            // step the next one waits for: multiply-add, subtract, compare, popcount (scalar), select, multiply-add, clamp.
            const int32_t hs = (int32_t)(scale >> 1), iscale = (int32_t)scale;
            int32_t thr;
            {
                const int32_t j = (int32_t)s;
                const int32_t tt = j < 8 ? 1 - ((8 - j) * iscale - hs) : (j - 7) * iscale - hs;
                thr = j < 15 ? tt * 4096 : 0x7FFFFFFF;
            }
            // (a lone wave issues an instruction every four cycles whatever it is, so the instruction count per step matters as
            //  much as the chain: the sample of step k comes to the half's lanes through one crossbar read, the two halves'
            //  counts through one vector popcount each)
            // (x << 12) of the half's 32 samples in every lane, gathered before the chain starts: one LDS store per lane, eight
            // 16-byte broadcast reads
            int32_t xs[32];
            wave_lds_sync();
            xl[lane] = (int32_t)((uint32_t)x << 12);
            wave_lds_sync();
#pragma unroll
            for (int k = 0; k < 32; k += 4) { const int4 q = *(const int4*)&xl[half * 32 + k]; xs[k] = q.x; xs[k + 1] = q.y; xs[k + 2] = q.z; xs[k + 3] = q.w; }
            const int32_t raw1 = xs[31] >> 12, raw2 = xs[30] >> 12;                   // the half's last two raw samples
            __builtin_amdgcn_sched_barrier(0);                                         // (keep the gather out of the chain: there every read would be waited for)
            int32_t g1 = h1, g2 = h2, mine = 0;
            int32_t c1g2 = __mul24(c1, g2);
#pragma unroll
            for (int k = 0; k < 32; k++) {
                const int32_t pred = __mul24(c0, g1) + c1g2;
                const uint64_t m = __ballot((xs[k] - pred) >= thr);
                // code = popcount(the half's compare bits) - 8 (v_bcnt_u32_b32 adds its second operand)
                const uint32_t mh = (C == 1 || !half) ? (uint32_t)m : (uint32_t)(m >> 32);
                const int32_t code = (int32_t)__builtin_popcount(mh) + (-8);
                int32_t sim;                                                           // code * scale + (pred >> 12): 24-bit operands
                asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(sim) : "v"(code), "v"(iscale), "v"(pred >> 12));
                sim = clamp_sym(sim, 0x7FFF);
                c1g2 = __mul24(c1, g1);
                g2 = g1; g1 = sim;
                mine = (int)s == k ? code : mine;
            }
            h1 = silent ? raw1 : g1; h2 = silent ? raw2 : g2;
            if (fr < store_from) continue;                                             // (wave-uniform) a warm-up row: only the history counts
            // two 4-bit codes per byte, first sample in the high nibble; even lanes hold the byte
            const uint32_t nib = silent ? 0u : ((uint32_t)mine & 15);
            const uint32_t nxt = (uint32_t)__shfl_xor((int)nib, 1);
            wave_lds_sync();
            if (act) {
                if (!(s & 1)) blk_img[half * 18 + 2 + (s >> 1)] = (uint8_t)((nib << 4) | nxt);
                if (s == 0) { blk_img[half * 18] = silent ? 0 : (uint8_t)(word >> 8); blk_img[half * 18 + 1] = silent ? 0 : (uint8_t)word; }
            }
            wave_lds_sync();
            uint8_t* row = dst + (uint64_t)fr * 18 * C;
            if (lane < 18 * C) row[lane] = blk_img[lane];
        }
        if (after_round(f0 + R < row_end ? f0 + R : row_end)) break;
    }
}

// (an instance per channel count, both launched over all streams: a block whose stream has the other count leaves at once)
template <int C>
__global__ __launch_bounds__(64) void k_adx_encode_wpf(AdxArgs a) {
    __shared__ uint8_t blk_img[40];
    __shared__ __attribute__((aligned(16))) int32_t xl[64];   // a frame's (sample << 12), by lane
    const AdxStream S = a.streams[a.wpf_order ? a.wpf_order[blockIdx.x] : blockIdx.x];
    if (S.channels != (uint32_t)C) return;
    const uint32_t half = threadIdx.x >> 5;
    const uint32_t chain = S.first_chain + (half < C ? half : 0);
    int32_t h1 = a.history[2 * chain], h2 = a.history[2 * chain + 1];
    adx_wpf_rows<C>(a, S, 0, S.frames, 0, h1, h2, blk_img, xl, [](uint32_t) { return false; });
}

// ------------------------------------------------------------------------------------------------------------
// Segmented chains (encode)
// ------------------------------------------------------------------------------------------------------------
// The encoder forgets its history too: the reconstruction it carries from block to block tracks the input within half a
// quantiser step whatever it started from, and two encodes of the same samples from different histories end up with identical
// histories -- later than the decoder's merge (the errors are requantised every sample): 120 rows on average for the bench's
// tonal material, 500 for sparse narrow-band material, 600 / 2200 at worst (coefficients 7400, -3342; oracle-side measurement).
// The same three passes as for the decoder, a WAVE per (file, segment) because the per-row work is the wave-per-file encoder's:
//   pass 0   every segment: `warm_rows` rows before it are encoded from the raw samples as history (nothing stored), the state
//            at the segment's first row is recorded, the segment is encoded and stored, the history after every round of four
//            rows goes to a checkpoint array (the output holds codes, not histories), the state at the end is recorded;
//   pass 1   a segment whose start state is not the previous segment's end state is encoded again from the right state until,
//            at a round's end, BOTH channels' histories equal the checkpoints (the stored blocks from there on are right);
//            reaching the segment's end with another state than recorded flags the file;
//   pass 2   flagged files only: the segments in order from the header history, repairing what is still inconsistent.
// rec[(segment * 2 + channel) * 4] = {start state, end state, end state after repair, -};  AdxStream::rows_avail = the file's
// first checkpoint (a checkpoint per round and channel).
__device__ __forceinline__ bool seg_enc_locate(const AdxArgs& a, uint32_t b, AdxStream& S, uint32_t& k) {
    if (b >= a.seg_lanes) return false;
    uint32_t lo = 0, hi = a.n_streams;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (a.seg_first[mid] <= b) lo = mid; else hi = mid; }
    S = a.streams[lo];
    k = b - S.first_seg;
    return k < S.seg_count;
}
// re-encode of segment k from (h1, h2) against the checkpoints; true: merged with what is stored before the segment's end
template <int C>
__device__ __forceinline__ bool seg_enc_repair(const AdxArgs& a, const AdxStream& S, uint32_t k, int32_t& h1, int32_t& h2, uint8_t* blk_img, int32_t* xl) {
    const uint32_t lane = threadIdx.x, half = lane >> 5, s = lane & 31;
    const bool act = half < C;
    const uint32_t r0 = k * S.seg_rows, r1 = r0 + S.seg_rows < S.frames ? r0 + S.seg_rows : S.frames;
    bool merged = false;
    adx_wpf_rows<C>(a, S, r0, r1, r0, h1, h2, blk_img, xl, [&](uint32_t next) {
        uint32_t* ck = a.seg_ckpt + ((uint64_t)S.rows_avail + (next + ADX_WPF_R - 1) / ADX_WPF_R - 1) * 2 + (act ? half : 0);
        const uint32_t now = seg_pack(h1, h2), was = *ck;
        const bool same = !act || was == now;
        __builtin_amdgcn_wave_barrier();
        if (act && s == 0) *ck = now;
        merged = __all(same);
        return merged;
    });
    return merged;
}

template <int C>
__global__ __launch_bounds__(64) void k_adx_seg_encode(AdxArgs a, uint32_t pass) {
    __shared__ uint8_t blk_img[40];
    __shared__ __attribute__((aligned(16))) int32_t xl[64];
    const uint32_t lane = threadIdx.x, half = lane >> 5, s = lane & 31;
    const bool act = half < C;
    AdxStream S; uint32_t k;
    if (pass < 2) {
        if (!seg_enc_locate(a, blockIdx.x, S, k) || S.channels != (uint32_t)C) return;
    } else {
        if (blockIdx.x >= a.n_streams) return;
        S = a.streams[blockIdx.x]; k = 0;
        if (S.channels != (uint32_t)C || !a.seg_flags[blockIdx.x]) return;
    }
    const uint32_t chain = S.first_chain + (act ? half : 0);
    auto rec_of = [&](uint32_t kk) { return a.seg_state + (((uint64_t)S.first_seg + kk) * 2 + (act ? half : 0)) * 4; };
    if (pass == 0) {
        const uint32_t r0 = k * S.seg_rows, r1 = r0 + S.seg_rows < S.frames ? r0 + S.seg_rows : S.frames;
        const uint32_t w0 = k == 0 ? 0 : (r0 > S.warm_rows ? r0 - S.warm_rows : 0);
        int32_t h1, h2;
        if (k == 0) { h1 = a.history[2 * chain]; h2 = a.history[2 * chain + 1]; }
        else {                                                       // a guess: the raw samples before the warm-up (the reconstruction is near them)
            const uint8_t* pcm = (S.src_in_scratch ? a.scratch : a.in) + S.src_offset;
            auto raw = [&](uint64_t idx) { const uint8_t* p = pcm + (idx * C + (act ? half : 0)) * 2; return idx < S.samples ? (int32_t)(int16_t)(p[0] | (p[1] << 8)) : 0; };
            h1 = w0 ? raw((uint64_t)w0 * 32 - 1) : 0; h2 = w0 ? raw((uint64_t)w0 * 32 - 2) : 0;
        }
        uint32_t spec = seg_pack(h1, h2);
        adx_wpf_rows<C>(a, S, w0, r1, r0, h1, h2, blk_img, xl, [&](uint32_t next) {
            if (next == r0) spec = seg_pack(h1, h2);
            if (next > r0 && act && s == 0) a.seg_ckpt[((uint64_t)S.rows_avail + (next + ADX_WPF_R - 1) / ADX_WPF_R - 1) * 2 + half] = seg_pack(h1, h2);
            return false;
        });
        if (act && s == 0) { uint32_t* rec = rec_of(k); const uint32_t e = seg_pack(h1, h2); rec[0] = spec; rec[1] = e; rec[2] = e; rec[3] = 0; }
    } else if (pass == 1) {
        if (k == 0) return;
        const uint32_t prev_end = rec_of(k - 1)[1], mine = rec_of(k)[0];
        if (!__any(act && prev_end != mine)) return;                 // the speculation was right for every channel
        int32_t h1, h2;
        seg_unpack(prev_end, h1, h2);
        const bool merged = seg_enc_repair<C>(a, S, k, h1, h2, blk_img, xl);
        const uint32_t e = seg_pack(h1, h2);
        const bool changed = !merged && act && e != rec_of(k)[1];
        if (changed && s == 0) rec_of(k)[2] = e;
        if (__any(changed) && lane == 0) {                           // the next segment started from a stale state
            uint32_t lo = 0, hi = a.n_streams;
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (a.seg_first[mid] <= blockIdx.x) lo = mid; else hi = mid; }
            a.seg_flags[lo] = 1u;
        }
    } else {
        int32_t h1 = a.history[2 * chain], h2 = a.history[2 * chain + 1];
        uint32_t cur = seg_pack(h1, h2), used = cur;
        for (k = 0; k < S.seg_count; k++) {
            uint32_t end = rec_of(k)[2];
            if (__any(act && used != cur)) {                         // the segment's blocks were encoded from `used`
                seg_unpack(cur, h1, h2);
                if (!seg_enc_repair<C>(a, S, k, h1, h2, blk_img, xl)) end = seg_pack(h1, h2);
            }
            used = rec_of(k)[1];
            cur = end;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Segmented chains (decode): speculate, verify, repair
// ------------------------------------------------------------------------------------------------------------
// The recurrence is serial, but the decoder FORGETS: two decodes of the same blocks from different histories differ by an
// error that the prediction filter (poles near 0.9 for the standard coefficients) shrinks to a few units within tens of
// samples, and the floor shifts then merge the two trajectories exactly -- after 400 samples on average for 500 Hz / 48 kHz
// (coefficients 7400, -3342), practically always within 3000; the time scales with 1 / (4096 - c0 - c1).  Once two decodes
// hold the same two history samples they are identical for good.  So a file is cut into segments of `seg_rows` block rows and
//   1. k_adx_seg_decode   every (segment, channel) is a lane: it decodes `warm_rows` rows before its segment from a zero
//                         history (nothing stored), records the state it arrives with, decodes and stores its segment and
//                         records the state it ends with.  Segment 0 starts from the header's history: it is exact;
//   2. k_adx_seg_fix      (three rounds) lane per segment again: if the state the previous segment ended with is not the state
//                         this one was decoded from, it is decoded again from the right state, row by row, until its state equals
//                         what is stored there (from that row on the stored samples are right already).  A chain whose ends
//                         still moved in the last round is flagged;
//   3. flagged chains     mono / stereo files: k_adx_decode_wpf decodes them again from their first block; other layouts (and
//                         chains with an end-of-stream marker) walk their segments in k_adx_seg_serial, lane per chain.
// Every unflagged chain is exact by induction: segment k's samples are the decode from the recorded end state of segment
// k - 1, and no recorded end state changed.  Flagged chains are exact by construction.  The result never depends on the
// warm-up length -- only the time does.  One 10 s file becomes some hundred lanes of 150 rows instead of two of 15 000.
// Standard layout only (blocksize 18, bitdepth 4, modes 2 / 3): a block is 16 code bytes, a row of samples fits registers.
struct SegLane {
    AdxStream S; uint32_t ch, k, r0, r1, w0; bool valid;
    const uint8_t* src; uint8_t* dst; uint32_t rowb;
};
__device__ __forceinline__ bool seg_locate(const AdxArgs& a, uint32_t g, SegLane& X) {
    X.valid = g < a.seg_lanes;
    uint32_t lo = 0, hi = a.n_streams;
    const uint32_t gg = X.valid ? g : 0;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (a.seg_first[mid] <= gg) lo = mid; else hi = mid; }
    X.S = a.streams[lo];
    const uint32_t local = gg - X.S.first_seg, C = X.S.channels;
    X.k = local / C; X.ch = local - X.k * C;
    if (X.k >= X.S.seg_count) { X.valid = false; X.k = 0; X.ch = 0; }           // (a padding lane: stereo pairs start on even lanes)
    X.r0 = X.k * X.S.seg_rows;
    X.r1 = X.r0 + X.S.seg_rows < X.S.frames ? X.r0 + X.S.seg_rows : X.S.frames;
    X.w0 = X.k == 0 ? 0 : (X.r0 > X.S.warm_rows ? X.r0 - X.S.warm_rows : 0);
    X.src = a.in + X.S.src_offset; X.dst = a.out + X.S.dst_offset; X.rowb = 18 * C;
    return X.valid;
}
// a block's scale word -> scale and the block's coefficients (adx.cpp:196-203)
__device__ __forceinline__ void seg_scale(const AdxStream& S, uint32_t word, int32_t& scale, int32_t& c0, int32_t& c1) {
    if (S.mode == 2) {
        const uint32_t pred = (word >> 13) & 7;
        scale = (int32_t)(word & 0x1FFF) + 1;
        c0 = pred < 4 ? ADX_STATIC_COEFS[pred * 2] : 0;
        c1 = pred < 4 ? ADX_STATIC_COEFS[pred * 2 + 1] : 0;
    } else { scale = (int32_t)word + 1; c0 = S.coef0; c1 = S.coef1; }
}
__device__ __forceinline__ uint32_t ld_be16(const uint8_t* p) { uint16_t v; __builtin_memcpy(&v, p, 2); return ((uint32_t)v >> 8) | (((uint32_t)v & 0xFF) << 8); }
// one block of 32 samples on registers (adx.cpp:204-213); 24-bit multiplies are exact: |code| <= 8, scale <= 2^16,
// |coefficient| <= 2^13, |history| <= 2^15
__device__ __forceinline__ void seg_block(const uint4& cw, int32_t scale, int32_t c0, int32_t c1, int32_t& h1, int32_t& h2, int32_t (&s)[32]) {
    const uint32_t w[4] = {cw.x, cw.y, cw.z, cw.w};
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint32_t d = w[i >> 2]; const uint32_t sh = 8 * (i & 3);
        int32_t code = __builtin_amdgcn_sbfe(d, sh + 4, 4);                      // first sample of a byte: the high nibble
        int32_t v = clamp_sym(__mul24(code, scale) + (__mul24(c0, h1) >> 12) + (__mul24(c1, h2) >> 12), 0x7FFF);
        h2 = h1; h1 = v; s[2 * i] = v;
        code = __builtin_amdgcn_sbfe(d, sh, 4);
        v = clamp_sym(__mul24(code, scale) + (__mul24(c0, h1) >> 12) + (__mul24(c1, h2) >> 12), 0x7FFF);
        h2 = h1; h1 = v; s[2 * i + 1] = v;
    }
}
// a row's samples to the WAV: sample i of channel ch at ((row * 32 + i) * C + ch) * 2.  Whole rows of mono / stereo files leave
// as 64 contiguous bytes per lane (stereo: the two lanes of a pair trade halves, so that lane ch stores half ch of the row's
// 128 interleaved bytes); anything else sample by sample.  `fast` must be the same in both lanes of a stereo pair.
// a whole row of a mono / stereo file as the 64 contiguous bytes this lane owns (stereo: the two lanes of a pair trade halves, so that
// lane ch holds half ch of the row's 128 interleaved bytes)
__device__ __forceinline__ void seg_row_pack(const SegLane& X, const int32_t (&s)[32], uint32_t (&o)[16]) {
    uint32_t P[16];
#pragma unroll
    for (int j = 0; j < 16; j++) P[j] = __builtin_amdgcn_perm((uint32_t)s[2 * j + 1], (uint32_t)s[2 * j], 0x05040100u);
    if (X.S.channels == 2) {
        const bool hi = X.ch != 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t keep = hi ? P[8 + j] : P[j], send = hi ? P[j] : P[8 + j];
            const uint32_t recv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)send, 0xB1, 0xF, 0xF, true);   // the pair's other lane (lane ^ 1)
            const uint32_t L = hi ? recv : keep, R = hi ? keep : recv;
            o[2 * j] = __builtin_amdgcn_perm(R, L, 0x05040100u);
            o[2 * j + 1] = __builtin_amdgcn_perm(R, L, 0x07060302u);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 16; j++) o[j] = P[j];
    }
}
__device__ __forceinline__ uint8_t* seg_row_address(const SegLane& X, uint32_t row) {
    const uint32_t C = X.S.channels;
    return X.dst + ((uint64_t)row * 32 * C + (C == 2 ? X.ch * 32 : 0)) * 2;
}
// a row's samples to the WAV: sample i of channel ch at ((row * 32 + i) * C + ch) * 2.  Whole rows of mono / stereo files leave
// as 64 contiguous bytes per lane; anything else sample by sample.  `fast` must be the same in both lanes of a stereo pair.
__device__ __forceinline__ void seg_store_row(const SegLane& X, uint32_t row, const int32_t (&s)[32], bool fast) {
    const uint32_t C = X.S.channels;
    if (fast && C <= 2) {
        uint32_t o[16];
        seg_row_pack(X, s, o);
        uint4* q = (uint4*)seg_row_address(X, row);
#pragma unroll
        for (int j = 0; j < 4; j++) q[j] = make_uint4(o[4 * j], o[4 * j + 1], o[4 * j + 2], o[4 * j + 3]);
        return;
    }
    int16_t* q = (int16_t*)X.dst;
#pragma unroll
    for (int i = 0; i < 32; i++) {
        const uint64_t idx = (uint64_t)row * 32 + i;
        if (idx < X.S.samples) q[idx * C + X.ch] = (int16_t)s[i];
    }
}
// end-of-stream test of a row (adx.cpp:405-406 and the reads past the input): the scale word of the row's FIRST block
__device__ __forceinline__ bool seg_row_ends(const SegLane& X, uint32_t row) {
    return row >= X.S.rows_avail || ld_be16(X.src + (uint64_t)row * X.rowb) == 0x8001u;
}

// ---- digital silence.  A block of zeros (scale word 0, codes 0: what the encoder writes for silent input, adx.cpp:231-234) makes the
// decoder run its predictor on its own output: the state decays to a fixed point or a small limit cycle of the floor arithmetic that
// depends on where it came from (-100, -100 maps to -100 for the standard coefficients) -- two decodes from different histories never
// merge in there, so a speculative segment inside a silent stretch is wrong until the sound comes back.  But nothing has to be
// decoded to know a silent stretch's end state: it is n steps of  v = (c0 * h1 >> 12) + (c1 * h2 >> 12)  from the state the stretch
// began with, and those are walked with cycle detection (Brent: at most the transient plus two periods -- a few hundred steps).
// k_adx_seg_decode marks the segments that are silent from their first row to their last (record word 3); k_adx_seg_fix then gives
// every segment behind a RUN of silent ones its exact start state in one round, from the last segment with sound before the run.
#define SEG_NO_STOP 0xFFFFFFFFu
#define SEG_SILENT 0xFFFFFFFEu          // record word 3: no end-of-stream row, every block of the segment is zeros
#define SEG_SILENT_AT 0xC0000000u       // ... the same, and the low 30 bits say where the silent run began: 1 + the chain's last segment with sound
                                        // before this one, 0 if there is none (k_adx_seg_runs; stop rows are below 2^27)
__device__ __forceinline__ bool seg_is_silent(uint32_t w3) { return w3 >= SEG_SILENT_AT && w3 != SEG_NO_STOP; }
__device__ __forceinline__ uint32_t seg_stop_of(uint32_t w3) { return w3 >= SEG_SILENT_AT ? SEG_NO_STOP : w3; }
__device__ __forceinline__ int32_t seg_idle_step(int32_t c0, int32_t c1, int32_t h1, int32_t h2) {
    return clamp_sym((__mul24(c0, h1) >> 12) + (__mul24(c1, h2) >> 12), 0x7FFF);      // seg_block with code 0
}
__device__ __forceinline__ uint32_t seg_idle_advance(int32_t c0, int32_t c1, uint32_t state, uint64_t n) {
    int32_t h1, h2;
    seg_unpack(state, h1, h2);
    uint32_t tortoise = state, power = 1, lam = 0;
    bool looking = true;
    for (uint64_t step = 0; step < n;) {
        const int32_t v = seg_idle_step(c0, c1, h1, h2);
        h2 = h1; h1 = v; step++;
        if (!looking) continue;
        lam++;
        const uint32_t cur = seg_pack(h1, h2);
        if (cur == tortoise) { n = step + (n - step) % lam; looking = false; }     // on the cycle: only the remainder is left to walk
        else if (lam == power) { tortoise = cur; power <<= 1; lam = 0; }
    }
    return seg_pack(h1, h2);
}

__global__ __launch_bounds__(64) void k_adx_seg_decode(AdxArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t seg_stage[64 * 80];      // a row of 64 bytes per lane, 80 apart (off each other's banks)
    __shared__ uint8_t* seg_to[64];
    const uint32_t lane = threadIdx.x, g = blockIdx.x * 64 + lane;
    SegLane X;
    seg_locate(a, g, X);
    const AdxStream& S = X.S;
    const uint32_t n = X.valid ? X.r1 - X.w0 : 0, chain = S.first_chain + X.ch;
    int32_t h1 = 0, h2 = 0;
    if (X.valid && X.k == 0) { h1 = a.history[2 * chain]; h2 = a.history[2 * chain + 1]; }
    uint32_t spec = seg_pack(h1, h2), stop_row = SEG_NO_STOP;
    bool stopped = false, all_zero = true;
    uint32_t nmax = n;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)nmax, o); nmax = t > nmax ? t : nmax; }
    // the next row's bytes are asked for while the current row is decoded
    uint4 cw = make_uint4(0, 0, 0, 0); uint32_t word = 0; bool ends = true;
    auto fetch = [&](uint32_t row, bool on) {
        if (on && row < S.rows_avail) {
            const uint8_t* p = X.src + (uint64_t)row * X.rowb;
            ends = ld_be16(p) == 0x8001u;
            p += X.ch * 18;
            word = ld_be16(p);
            __builtin_memcpy(&cw, p + 2, 16);
        } else ends = true;
    };
    fetch(X.w0, n > 0);
    for (uint32_t t = 0; t < nmax; t++) {
        const bool act = t < n;
        const uint32_t row = X.w0 + t;
        uint8_t* row_to = nullptr;                                   // where this lane's staged row goes (null: nothing staged)
        const uint4 ccw = cw; const uint32_t cword = word; const bool cends = ends;
        fetch(row + 1, act && t + 1 < n && !stopped && !cends);
        if (act) {
            if (row == X.r0) spec = seg_pack(h1, h2);
            if (!stopped && cends) { stopped = true; stop_row = row > X.r0 ? row : X.r0; }
            if (row >= X.r0) all_zero = all_zero && !stopped && cword == 0 && (ccw.x | ccw.y | ccw.z | ccw.w) == 0;
            int32_t s[32];
            if (!stopped) {
                int32_t scale, c0, c1;
                seg_scale(S, cword, scale, c0, c1);
                seg_block(ccw, scale, c0, c1, h1, h2, s);
            } else {
#pragma unroll
                for (int i = 0; i < 32; i++) s[i] = 0;                 // rows never reached decode to silence
            }
            if (row >= X.r0) {
                if (S.channels <= 2 && (uint64_t)(row + 1) * 32 <= S.samples) {      // a whole row: through the wave's staging piece (below)
                    uint32_t o[16];
                    seg_row_pack(X, s, o);
                    uint4* sl = (uint4*)(seg_stage + lane * 80);
#pragma unroll
                    for (int j = 0; j < 4; j++) sl[j] = make_uint4(o[4 * j], o[4 * j + 1], o[4 * j + 2], o[4 * j + 3]);
                    row_to = seg_row_address(X, row);
                } else seg_store_row(X, row, s, false);
            }
        }
        // The lanes' rows lie megabytes apart; stored lane by lane each instruction wrote 64 sixteen-byte pieces (1.46 x the PCM in
        // WRITE_SIZE).  Through LDS, four consecutive lanes store one lane's 64 bytes -- eight a stereo pair's whole 128-byte line.
        seg_to[lane] = row_to;
        wave_lds_sync();
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t L = 16 * j + (lane >> 2), qq = lane & 3;
            uint8_t* to = seg_to[L];
            if (to) *(uint4*)(to + 16 * qq) = *(const uint4*)(seg_stage + L * 80 + 16 * qq);
        }
        wave_lds_sync();
    }
    if (X.valid) {
        uint32_t* rec = a.seg_state + 4 * (uint64_t)g;
        const uint32_t e = seg_pack(h1, h2);
        rec[0] = spec; rec[1] = e; rec[2] = e; rec[3] = stop_row != SEG_NO_STOP ? stop_row : (all_zero ? SEG_SILENT : SEG_NO_STOP);
        if (stop_row != SEG_NO_STOP) atomicOr(&a.seg_flags[chain], 1u);
    }
}

// Decodes rows [r0, r_end) of a segment again from (h1, h2), storing them, until the state after a row equals what is stored
// there (`true`: from that row on the stored samples are this trajectory's already).  Whole rows only can be compared.
__device__ __forceinline__ bool seg_repair(const SegLane& X, uint32_t r_end, int32_t& h1, int32_t& h2, bool silent = false) {
    const AdxStream& S = X.S;
    const int16_t* q = (const int16_t*)X.dst;
    for (uint32_t row = X.r0; row < r_end; row++) {
        if (silent && h1 == h2) {                                    // a silent segment at a fixed point of the recurrence: the rest is that constant
            int32_t sc, c0, c1;
            seg_scale(S, 0, sc, c0, c1);
            if (seg_idle_step(c0, c1, h1, h2) == h1) {
                int32_t s[32];
#pragma unroll
                for (int i = 0; i < 32; i++) s[i] = h1;
                for (; row < r_end; row++) seg_store_row(X, row, s, false);
                return false;
            }
        }
        const uint8_t* p = X.src + (uint64_t)row * X.rowb + X.ch * 18;
        uint4 cw; __builtin_memcpy(&cw, p + 2, 16);
        int32_t scale, c0, c1;
        seg_scale(S, ld_be16(p), scale, c0, c1);
        int32_t s[32];
        seg_block(cw, scale, c0, c1, h1, h2, s);
        const bool whole = (uint64_t)(row + 1) * 32 <= S.samples;
        bool same = false;
        if (whole) {
            const uint64_t i31 = ((uint64_t)row * 32 + 31) * S.channels + X.ch, i30 = i31 - S.channels;
            same = (int32_t)q[i31] == s[31] && (int32_t)q[i30] == s[30];
        }
        seg_store_row(X, row, s, false);
        if (same) return true;
    }
    return false;
}

// One round of repairs.  rec[0] = the state the segment's stored samples were decoded from, rec[1] / rec[2] = the state they end
// with before / after the round (the two alternate: a round reads the previous segment's end as the previous round left it, so
// lanes never read what a neighbour is writing).  A merge takes longer than a segment now and then (clean narrow-band material:
// 6500 samples at worst against 400 on average); a second and third round settle those chains -- only if the LAST round still moved
// an end is the chain flagged for the serial pass.
__global__ __launch_bounds__(64) void k_adx_seg_fix(AdxArgs a, uint32_t round, uint32_t last) {
    const uint32_t g = blockIdx.x * 64 + threadIdx.x;
    SegLane X;
    if (!seg_locate(a, g, X)) return;
    uint32_t* rec = a.seg_state + 4 * (uint64_t)g;
    const uint32_t src = 1 + (round & 1), dst = 1 + ((round + 1) & 1);
    const uint32_t my_end = rec[src];
    if (X.k == 0) { rec[dst] = my_end; return; }
    const uint32_t C = X.S.channels;
    uint32_t prev_end = a.seg_state[4 * (uint64_t)(g - C) + src];
    const uint32_t prev_w3 = a.seg_state[4 * (uint64_t)(g - C) + 3];
    if (seg_is_silent(prev_w3)) {
        // behind a run of silent segments: the start state follows from the last segment with sound before the run (or the header's
        // history), however long the run -- no decode, and no waiting for the run's segments to be repaired one round after the other.
        // Long chains carry the run's beginning in the record (k_adx_seg_runs: one load); short ones are walked back here -- a walk per lane
        // and round over a chain of n silent segments would be n^2 dependent loads (an hour of silence is 250 000 segments).
        uint32_t j; bool from_start;
        if (prev_w3 != SEG_SILENT) { const uint32_t at = prev_w3 & 0x3FFFFFFFu; from_start = at == 0; j = from_start ? 0 : at - 1; }
        else {
            j = X.k - 1;
            while (j > 0 && seg_is_silent(a.seg_state[4 * ((uint64_t)g - (uint64_t)(X.k - j) * C) + 3])) j--;
            from_start = seg_is_silent(a.seg_state[4 * ((uint64_t)g - (uint64_t)(X.k - j) * C) + 3]);      // (j == 0 and silent too)
        }
        uint32_t state; uint32_t from_row;
        if (from_start) { const uint32_t chain = X.S.first_chain + X.ch; state = seg_pack(a.history[2 * chain], a.history[2 * chain + 1]); from_row = 0; }
        else { state = a.seg_state[4 * ((uint64_t)g - (uint64_t)(X.k - j) * C) + src]; from_row = (j + 1) * X.S.seg_rows; }
        int32_t sc, c0, c1;
        seg_scale(X.S, 0, sc, c0, c1);
        prev_end = seg_idle_advance(c0, c1, state, (uint64_t)(X.r0 - from_row) * 32);
    }
    if (prev_end == rec[0]) { rec[dst] = my_end; return; }          // decoded from the right state already
    int32_t h1, h2;
    seg_unpack(prev_end, h1, h2);
    const uint32_t stop_row = seg_stop_of(rec[3]), r_end = stop_row < X.r1 ? stop_row : X.r1;
    const bool merged = seg_repair(X, r_end, h1, h2, seg_is_silent(rec[3]));
    const uint32_t e = merged ? my_end : seg_pack(h1, h2);
    rec[0] = prev_end; rec[dst] = e;
    if (last && e != my_end) atomicOr(&a.seg_flags[X.S.first_chain + X.ch], 1u);     // the next segment started from a stale state
}

// Where the silent runs of long chains begin: a wave per chain takes 64 of its segments at a time and writes, into the record of every
// silent one, 1 + the last segment with sound before it (a ballot and a count of leading zeros per step, the carry in a scalar).
#define SEG_RUNS_MIN 48u                // chains of fewer segments are walked back by k_adx_seg_fix itself
__global__ __launch_bounds__(64) void k_adx_seg_runs(AdxArgs a) {
    const uint32_t lane = threadIdx.x;
    for (uint32_t chain = blockIdx.x; chain < a.chains; chain += gridDim.x) {
        uint32_t lo = 0, hi = a.n_streams;
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (a.streams[mid].first_chain <= chain) lo = mid; else hi = mid; }
        const AdxStream S = a.streams[lo];
        if (S.seg_count < SEG_RUNS_MIN) continue;
        const uint32_t ch = chain - S.first_chain, C = S.channels;
        uint32_t carry = 0;
        for (uint32_t k0 = 0; k0 < S.seg_count; k0 += 64) {
            const uint32_t k = k0 + lane;
            const bool on = k < S.seg_count;
            uint32_t* rec = a.seg_state + 4 * ((uint64_t)S.first_seg + (uint64_t)(on ? k : 0) * C + ch);
            const bool silent = on && seg_is_silent(rec[3]);
            const uint64_t sound = __builtin_amdgcn_ballot_w64(on && !silent);
            const uint64_t below = sound & (~0ull >> (63 - lane));                  // segments with sound at or before this lane's
            const uint32_t last = below ? k0 + 64 - (uint32_t)__builtin_clzll(below) : carry;
            if (silent) rec[3] = SEG_SILENT_AT | last;
            if (sound) carry = k0 + 64 - (uint32_t)__builtin_clzll(sound);
        }
    }
}

__global__ __launch_bounds__(64) void k_adx_seg_serial(AdxArgs a, uint32_t fin) {
    const uint32_t chain = blockIdx.x * 64 + threadIdx.x;
    if (chain >= a.chains || !a.seg_flags[chain]) return;
    SegLane X;
    seg_locate(a, 0, X);                                             // (fills the fields that do not depend on the lane)
    {   // chain -> stream
        uint32_t lo = 0, hi = a.n_streams;
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (a.streams[mid].first_chain <= chain) lo = mid; else hi = mid; }
        X.S = a.streams[lo];
    }
    const AdxStream& S = X.S;
    if (S.channels <= 2) return;                                     // (mono / stereo files: the wave-per-file decoder takes flagged ones, see launch_adx_decode_seg)
    X.ch = chain - S.first_chain; X.valid = true;
    X.src = a.in + S.src_offset; X.dst = a.out + S.dst_offset; X.rowb = 18 * S.channels;
    int32_t h1 = a.history[2 * chain], h2 = a.history[2 * chain + 1];
    uint32_t cur = seg_pack(h1, h2);
    bool stopped = false;
    for (uint32_t k = 0; k < S.seg_count; k++) {
        const uint64_t g = (uint64_t)S.first_seg + (uint64_t)k * S.channels + X.ch;
        uint32_t* rec = a.seg_state + 4 * g;
        X.k = k; X.r0 = k * S.seg_rows; X.r1 = X.r0 + S.seg_rows < S.frames ? X.r0 + S.seg_rows : S.frames;
        if (stopped) {                                               // everything after the end of the stream is silence
            int16_t* q = (int16_t*)X.dst;
            const uint64_t i0 = (uint64_t)X.r0 * 32, i1 = (uint64_t)X.r1 * 32 < S.samples ? (uint64_t)X.r1 * 32 : S.samples;
            for (uint64_t i = i0; i < i1; i++) q[i * S.channels + X.ch] = 0;
            continue;
        }
        const uint32_t stop_row = seg_stop_of(rec[3]), r_end = stop_row < X.r1 ? stop_row : X.r1;
        uint32_t end = rec[fin];                                     // (fin: where the last repair round left the ends)
        if (rec[0] != cur) {                                         // the segment's samples were decoded from rec[0]
            seg_unpack(cur, h1, h2);
            if (!seg_repair(X, r_end, h1, h2, seg_is_silent(rec[3]))) end = seg_pack(h1, h2);
        }
        if (stop_row != SEG_NO_STOP) stopped = true;
        cur = end;
    }
}

// ------------------------------------------------------------------------------------------------------------
// Segmented chains (encode), a LANE per (file, channel, segment): batches of many files
// ------------------------------------------------------------------------------------------------------------
// The wave-per-(file, segment) encoder above spends a wave's 64 lanes on two chains; with hundreds of files that is what its
// time is made of (its VALU pipes are busy 93 % of the time, profiles/r03_a_pmc_1000streams.json).  Here every (file, channel,
// segment) is one lane and a block row is ~1000 instructions for 64 chains at once.  There is no warm-up: pass 0 encodes every
// segment from a guessed history (the raw samples before it), pass 1 encodes every segment AGAIN from the previous segment's
// recorded end state until the history at a checkpoint (every four rows) equals what pass 0 left there -- for most segments some
// hundred rows -- and pass 2 walks flagged files.  The two lanes of a stereo pair move in lock step (they share the loads of
// a row's samples and the stores of its two blocks), so a pair repairs together and merges when both channels have.
// Records and checkpoints as for the wave form: rec[(lane) * 4] = {start state, end state, end state after repair, -},
// checkpoint ((file's first + round) * 2 + channel).  Standard layout, one or two channels.
struct EncLane {
    AdxStream S; uint32_t ch, k, r0, r1, stream; bool valid;
    const uint8_t* pcm; uint8_t* dst;
};
__device__ __forceinline__ bool enc_lane_locate(const AdxArgs& a, uint32_t g, EncLane& X) {
    X.valid = g < a.seg_lanes;
    uint32_t lo = 0, hi = a.n_streams;
    const uint32_t gg = X.valid ? g : 0;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (a.seg_first[mid] <= gg) lo = mid; else hi = mid; }
    X.S = a.streams[lo]; X.stream = lo;
    const uint32_t local = gg - X.S.first_seg, C = X.S.channels;
    X.k = local / C; X.ch = local - X.k * C;
    if (X.k >= X.S.seg_count) { X.valid = false; X.k = 0; X.ch = 0; }           // (a padding lane: stereo pairs start on even lanes)
    X.r0 = X.k * X.S.seg_rows;
    X.r1 = X.r0 + X.S.seg_rows < X.S.frames ? X.r0 + X.S.seg_rows : X.S.frames;
    X.pcm = (X.S.src_in_scratch ? a.scratch : a.in) + X.S.src_offset; X.dst = a.out + X.S.dst_offset;
    return X.valid;
}
// this lane's channel of row `row`: 32 samples.  A stereo pair loads the row's 128 bytes once (64 per lane) and trades halves.
// A row's samples in two steps, so that the kernel can ask for the next row's before it encodes this one (a lane has one row in
// flight and less than one wave shares its SIMD: without that, every row waited a full memory latency for its samples).
// enc_lane_fetch: the raw words of a whole row of a mono / stereo file (false: a row cut by the end of the input, or another
// layout -- enc_lane_load then reads it sample by sample); enc_lane_unpack: this lane's 32 samples out of them.
__device__ __forceinline__ bool enc_lane_fetch(const EncLane& X, uint32_t row, uint32_t (&own)[16]) {
    const uint32_t C = X.S.channels;
    const bool whole = (uint64_t)(row + 1) * 32 <= X.S.samples;      // (samples past the input are zero padding, adx.cpp:453-456)
    if (!whole || C > 2) return false;
    const uint8_t* p = C == 2 ? X.pcm + ((uint64_t)row * 32 + 16 * X.ch) * 4 : X.pcm + (uint64_t)row * 64;
#pragma unroll
    for (int j = 0; j < 4; j++) { uint4 v; __builtin_memcpy(&v, p + 16 * j, 16); own[4 * j] = v.x; own[4 * j + 1] = v.y; own[4 * j + 2] = v.z; own[4 * j + 3] = v.w; }
    return true;
}
__device__ __forceinline__ void enc_lane_unpack(const EncLane& X, const uint32_t (&own)[16], int32_t (&x)[32]) {
    if (X.S.channels == 2) {
        // lane ch 0 holds samples 0..15 of both channels, lane ch 1 samples 16..31: one permute per dword makes {x[i], x[16 + i]} of MY channel
        const uint32_t sel = X.ch ? 0x03020706u : 0x05040100u;       // ch 0: own.lo | partner.lo << 16;  ch 1: partner.hi | own.hi << 16
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const uint32_t partner = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)own[i], 0xB1, 0xF, 0xF, true);
            const uint32_t pr = __builtin_amdgcn_perm(partner, own[i], sel);
            x[i] = (int32_t)(int16_t)(pr & 0xFFFF); x[16 + i] = (int32_t)pr >> 16;
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 16; i++) { x[2 * i] = (int32_t)(int16_t)(own[i] & 0xFFFF); x[2 * i + 1] = (int32_t)own[i] >> 16; }
}
__device__ __forceinline__ void enc_lane_load(const EncLane& X, uint32_t row, int32_t (&x)[32]) {
    const uint32_t C = X.S.channels;
    uint32_t own[16];
    if (enc_lane_fetch(X, row, own)) { enc_lane_unpack(X, own, x); return; }
#pragma unroll
    for (int i = 0; i < 32; i++) {
        const uint64_t idx = (uint64_t)row * 32 + i;
        int16_t v = 0;
        if (idx < X.S.samples) __builtin_memcpy(&v, X.pcm + (idx * C + X.ch) * 2, 2);
        x[i] = v;
    }
}
// ChannelFrame::Encode (adx.cpp:215-273) of one block on registers: the scale word, the 16 code bytes as four little-endian
// words, the new history.  (The quantiser is k_adx_encode's float form: exact for |delta| capped at (limit + 2) * scale.)
// Pass A (adx.cpp:221-230) takes the range of the residuals against the RAW samples before each one -- only the first two of a block see the
// encoder's own state (its reconstruction of the block before).  enc_lane_range_tail is the part that depends on nothing but the block's own
// samples (residuals 2 .. 31): k_adx_lane_encode computes it for the NEXT row inside the current row's iteration, where it fills the issue slots
// that the serial chain of pass B leaves empty (a lane alone on its SIMD issues a dependent instruction every other slot at best).
__device__ __forceinline__ void enc_lane_range_tail(const AdxStream& S, const int32_t (&x)[32], int32_t& mn, int32_t& mx) {
    const int32_t c0 = S.coef0, c1 = S.coef1;
    mn = 0; mx = 0;
#pragma unroll
    for (int i = 2; i < 32; i++) {
        const int32_t r = ((int32_t)((uint32_t)x[i] << 12) - __mul24(c0, x[i - 1]) - __mul24(c1, x[i - 2])) >> 12;
        mn = r < mn ? r : mn; mx = r > mx ? r : mx;
    }
}
__device__ __forceinline__ void enc_lane_block_pre(const AdxStream& S, const int32_t (&x)[32], int32_t mn, int32_t mx, int32_t& h1, int32_t& h2, uint32_t& word, uint32_t (&cw)[4]) {
    const int32_t c0 = S.coef0, c1 = S.coef1;
    {   // the first two residuals: against the state the block starts from
        const int32_t r0 = ((int32_t)((uint32_t)x[0] << 12) - __mul24(c0, h1) - __mul24(c1, h2)) >> 12;
        const int32_t r1 = ((int32_t)((uint32_t)x[1] << 12) - __mul24(c0, x[0]) - __mul24(c1, h1)) >> 12;
        mn = r0 < mn ? r0 : mn; mx = r0 > mx ? r0 : mx;
        mn = r1 < mn ? r1 : mn; mx = r1 > mx ? r1 : mx;
    }
    const bool silent = !mn && !mx;                               // adx.cpp:231-234: a block of zeros, and the history stays raw (selected at the end:
                                                                  // no branch, so that the caller's independent work can share this block's schedule)
    const int32_t qa = mx / 7, qb = (int32_t)((uint32_t)(-mn) >> 3);
    uint32_t scale = (uint32_t)(qa > qb ? qa : qb) & 0xFFFF;
    if (scale > 0x1000) scale = 0x1000;
    if (S.mode == 4) {
        const uint32_t power = scale ? (32 - __clz((int)scale)) : 0;
        scale = (1u << power) & 0xFFFF;
        word = (uint32_t)(12 - (int32_t)power) & 0xFFFF;
    } else if (S.mode == 2) word = (S.filter_bits | (scale & 0x1FFF)) & 0xFFFF;
    else word = scale;
    if (!scale) scale = 1;
    const AdxQuantLane quant(scale);                              // the quantiser (adx.cpp:256-261) as one sign-symmetric float step: cri_adx_quant.h
    const int32_t iscale = (int32_t)scale;
    int32_t g1 = h1, g2 = h2;
    int32_t c1g2 = __mul24(c1, g2);
#pragma unroll
    for (int w = 0; w < 4; w++) {
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {                                // pass B (adx.cpp:254-271)
            const int32_t pred = __mul24(c0, g1) + c1g2;
            const int32_t d = ((int32_t)((uint32_t)x[8 * w + i] << 12) - pred) >> 12;
            const int32_t code = quant(d);
            // = ((code * scale << 12) + pred) >> 12: the product has no low bits.  (As v_mad_i32_i24 by hand: knowing code's range the compiler
            //  drops the 24-bit form and takes v_mad_u64_u32, a quarter-rate instruction, onto the chain from one sample to the next.)
            int32_t sim;
            asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(sim) : "v"(code), "v"(iscale), "v"(pred >> 12));
            sim = clamp_sym(sim, 0x7FFF);
            c1g2 = __mul24(c1, g1);
            g2 = g1; g1 = sim;
            acc = (acc << 4) | ((uint32_t)code & 15u);               // first sample in the high nibble of the first byte
        }
        cw[w] = __builtin_bswap32(acc);
    }
    h1 = silent ? x[31] : g1; h2 = silent ? x[30] : g2;
    word = silent ? 0u : word;
#pragma unroll
    for (int w = 0; w < 4; w++) cw[w] = silent ? 0u : cw[w];
}
__device__ __forceinline__ void enc_lane_block(const AdxStream& S, const int32_t (&x)[32], int32_t& h1, int32_t& h2, uint32_t& word, uint32_t (&cw)[4]) {
    int32_t mn, mx;
    enc_lane_range_tail(S, x, mn, mx);
    enc_lane_block_pre(S, x, mn, mx, h1, h2, word, cw);
}
// the row's blocks to the file.  Stereo: the pair's 36 bytes are nine aligned words; lane ch 0 stores words 0..4 (the last one
// carries its own last two code bytes and the partner's scale word), lane ch 1 words 5..8.
__device__ __forceinline__ void enc_lane_store(const EncLane& X, uint32_t row, uint32_t word, const uint32_t (&cw)[4], bool pair) {
    const uint32_t C = X.S.channels;
    const uint32_t sw = ((word >> 8) & 0xFF) | ((word & 0xFF) << 8);                 // big-endian scale word
    if (pair) {
        const uint32_t psw = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sw, 0xB1, 0xF, 0xF, true);
        uint32_t* q = (uint32_t*)(X.dst + (uint64_t)row * 36);
        if (X.ch == 0) {
            typedef uint32_t u4u __attribute__((ext_vector_type(4), aligned(4)));
            *(u4u*)q = u4u{sw | (cw[0] << 16), __builtin_amdgcn_alignbit(cw[1], cw[0], 16), __builtin_amdgcn_alignbit(cw[2], cw[1], 16), __builtin_amdgcn_alignbit(cw[3], cw[2], 16)};
            q[4] = (cw[3] >> 16) | (psw << 16);
        } else {
            typedef uint32_t u4u __attribute__((ext_vector_type(4), aligned(4)));
            *(u4u*)(q + 5) = u4u{cw[0], cw[1], cw[2], cw[3]};
        }
        return;
    }
    uint8_t* q = X.dst + ((uint64_t)row * C + X.ch) * 18;
    const uint16_t s16 = (uint16_t)sw;
    __builtin_memcpy(q, &s16, 2);
#pragma unroll
    for (int w = 0; w < 4; w++) { const uint16_t lo = (uint16_t)cw[w], hi = (uint16_t)(cw[w] >> 16); __builtin_memcpy(q + 2 + 4 * w, &lo, 2); __builtin_memcpy(q + 4 + 4 * w, &hi, 2); }
}

// pass 0: every segment from a guessed history (the raw samples before it).  pass r >= 1, a repair round: a segment whose bytes
// were not encoded from the end state its predecessor has NOW is encoded again from that state until its histories meet the
// recorded checkpoints (from there on the bytes are right already) or it ends; a segment that ends differently than before makes
// its successor stale, which the next round repairs -- the end states live in two slots written alternately (rec[1 + (r & 1)]),
// so a round reads what the round before left while it writes its own.  A file in which an end state still moved in the last
// round is flagged (seg_flags = rounds) and walked in order by k_adx_lane_encode_serial.
//   rec[0] = the state the segment's bytes were encoded from, rec[1] / rec[2] = its end state after the even / odd rounds
__global__ __launch_bounds__(256) void k_adx_lane_encode(AdxArgs a, uint32_t pass) {
    // (four waves to the workgroup, nothing shared between them: see launch_adx_encode_lane)
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    EncLane X;
    enc_lane_locate(a, g, X);
    const AdxStream& S = X.S;
    const uint32_t C = S.channels, chain = S.first_chain + X.ch;
    const bool pair = C == 2;                                        // (a pair's lanes are g, g ^ 1: the planner starts stereo files on even lanes)
    uint32_t* rec = a.seg_state + 4 * (uint64_t)(X.valid ? g : 0);
    uint32_t* ck = a.seg_ckpt + ((uint64_t)S.rows_avail * 2 + X.ch);
    int32_t h1 = 0, h2 = 0;
    bool run = X.valid && X.r1 > X.r0;
    uint32_t want_end = 0, new_start = 0, warm_from = 0;
    const uint32_t slot_r = 1 + ((pass - 1) & 1), slot_w = 1 + (pass & 1);      // (rounds: the slot read, the slot written)
    if (pass == 0) {
        if (X.valid) {
            // the warm-up: `warm_rows` rows before the segment are encoded from the raw samples before THEM as history, nothing
            // stored -- the encoder's reconstruction has usually found the true trajectory by the segment's first row (a warm-up
            // that reaches the file's start begins with the header's history and is exact)
            const uint32_t w0 = X.k == 0 ? 0 : (X.r0 > S.warm_rows ? X.r0 - S.warm_rows : 0);
            if (w0 == 0) { h1 = a.history[2 * chain]; h2 = a.history[2 * chain + 1]; }
            else {
                auto raw = [&](uint64_t idx) { int16_t v = 0; if (idx < S.samples) __builtin_memcpy(&v, X.pcm + (idx * C + X.ch) * 2, 2); return (int32_t)v; };
                h1 = raw((uint64_t)w0 * 32 - 1); h2 = raw((uint64_t)w0 * 32 - 2);
            }
            warm_from = w0;
        }
        // (the pair's lanes warm up over the same rows: enc_lane_load's exchange needs both)
        uint32_t wn = X.valid && X.r1 > X.r0 ? X.r0 - warm_from : 0, wmax = wn;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)wmax, o); wmax = t > wmax ? t : wmax; }
        // (the same pipeline as the segment's loop below: two thirds of a lane's rows are warm-up rows)
        int32_t wa[32]; int32_t wmn = 0, wmx = 0;
#pragma unroll
        for (int i = 0; i < 32; i++) wa[i] = 0;
        uint32_t wcur[16] = {};
        bool wcur_ok = false;
        if (wn > 0) {
            enc_lane_load(X, warm_from, wa);
            enc_lane_range_tail(S, wa, wmn, wmx);
            wcur_ok = wn > 1 && enc_lane_fetch(X, warm_from + 1, wcur);
        }
        for (uint32_t t = 0; t < wmax; t++) {
            if (t < wn) {
                const bool has_next = t + 1 < wn, next_whole = wcur_ok;
                int32_t xb[32];
                enc_lane_unpack(X, wcur, xb);
                wcur_ok = t + 2 < wn && enc_lane_fetch(X, warm_from + t + 2, wcur);
                int32_t qmn, qmx;
                uint32_t word, cw[4];
                enc_lane_range_tail(S, xb, qmn, qmx);
                enc_lane_block_pre(S, wa, wmn, wmx, h1, h2, word, cw);
                if (has_next && !next_whole) { enc_lane_load(X, warm_from + t + 1, xb); enc_lane_range_tail(S, xb, qmn, qmx); }
#pragma unroll
                for (int i = 0; i < 32; i++) wa[i] = xb[i];
                wmn = qmn; wmx = qmx;
            }
        }
        if (X.valid) rec[0] = seg_pack(h1, h2);
    } else {
        bool mis = false;
        if (X.valid) want_end = rec[slot_r];
        if (X.valid && X.k > 0) {
            const uint32_t prev_end = a.seg_state[4 * (uint64_t)(g - C) + slot_r];
            mis = prev_end != rec[0];
            seg_unpack(prev_end, h1, h2);
            new_start = prev_end;
        }
        // (the exchange runs in every lane, outside any condition on `mis`: a lane that is masked off reads as 0)
        const bool partner_mis = __builtin_amdgcn_update_dpp(0, (int)mis, 0xB1, 0xF, 0xF, true) != 0;
        const bool pmis = mis || (pair && partner_mis);
        run = run && X.k > 0 && pmis;                                // the pair repairs together: the channel that was right re-encodes the same bytes
    }
    uint32_t nrows = run ? X.r1 - X.r0 : 0, nmax = nrows;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)nmax, o); nmax = t > nmax ? t : nmax; }
    bool merged = false;
    // The current row's samples (xa) and the tail of its residual range come from the iteration before; the next row's raw words (cur) were
    // asked for two rows ahead.  One iteration = unpack the next row, ask for the one after it, then ONE block of code in which the next row's
    // range (independent work) and this row's serial pass B share the schedule.
    int32_t xa[32]; int32_t pmn = 0, pmx = 0;
#pragma unroll
    for (int i = 0; i < 32; i++) xa[i] = 0;
    uint32_t cur[16] = {};
    bool cur_ok = false;
    if (nrows > 0) {
        enc_lane_load(X, X.r0, xa);
        enc_lane_range_tail(S, xa, pmn, pmx);
        cur_ok = nrows > 1 && enc_lane_fetch(X, X.r0 + 1, cur);
    }
    for (uint32_t t = 0; t < nmax; t++) {
        const bool act = run && t < nrows && !merged;
        if (__builtin_expect(!__any(act), 0)) break;
        if (act) {
            const uint32_t row = X.r0 + t;
            const bool has_next = t + 1 < nrows, next_whole = cur_ok;
            int32_t xb[32];
            enc_lane_unpack(X, cur, xb);                             // (meaningless without a next row: nothing reads it then)
            cur_ok = t + 2 < nrows && enc_lane_fetch(X, row + 2, cur);
            int32_t qmn, qmx;
            uint32_t word, cw[4];
            enc_lane_range_tail(S, xb, qmn, qmx);
            enc_lane_block_pre(S, xa, pmn, pmx, h1, h2, word, cw);
            enc_lane_store(X, row, word, cw, pair);
            if (has_next && !next_whole) { enc_lane_load(X, row + 1, xb); enc_lane_range_tail(S, xb, qmn, qmx); }      // (a row cut by the end of the input)
#pragma unroll
            for (int i = 0; i < 32; i++) xa[i] = xb[i];
            pmn = qmn; pmx = qmx;
            if (((row + 1) & 3) == 0 || row + 1 == X.r1) {
                uint32_t* c = ck + (uint64_t)((row + 4) / 4 - 1) * 2;
                const uint32_t now = seg_pack(h1, h2);
                if (pass == 0) *c = now;
                else {
                    const bool mine = *c == now;
                    *c = now;
                    const bool partner_same = __builtin_amdgcn_update_dpp(0, (int)mine, 0xB1, 0xF, 0xF, true) != 0;
                    merged = mine && (!pair || partner_same);
                }
            }
        }
    }
    if (!X.valid) return;
    const uint32_t e = seg_pack(h1, h2);
    if (pass == 0) { rec[1] = e; rec[2] = e; rec[3] = 0; }
    else {
        const uint32_t now_end = run && !merged ? e : want_end;
        rec[slot_w] = now_end;
        if (run) rec[0] = new_start;
        if (now_end != want_end) a.seg_flags[X.stream] = pass;       // the next segment started from a stale state: the next round's work
    }
}

// pass 2: flagged files, a lane (pair) per file, the segments in order
__global__ __launch_bounds__(64) void k_adx_lane_encode_serial(AdxArgs a, uint32_t rounds) {
    const uint32_t chain = blockIdx.x * 64 + threadIdx.x;
    EncLane X;
    X.valid = chain < a.chains;
    uint32_t lo = 0, hi = a.n_streams;
    const uint32_t cc = X.valid ? chain : 0;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (a.streams[mid].first_chain <= cc) lo = mid; else hi = mid; }
    X.S = a.streams[lo]; X.stream = lo;
    const AdxStream& S = X.S;
    X.ch = cc - S.first_chain;
    if (X.ch >= S.channels) { X.valid = false; X.ch = 0; }
    X.pcm = (S.src_in_scratch ? a.scratch : a.in) + S.src_offset; X.dst = a.out + S.dst_offset;
    const uint32_t C = S.channels;
    const bool pair = C == 2;
    const bool go = X.valid && a.seg_flags[lo] == rounds;            // an end state moved in the last round
    const uint32_t slot = 1 + (rounds & 1);                          // (the slot the last round wrote)
    if (!__any(go)) return;
    int32_t h1 = 0, h2 = 0;
    if (go) { h1 = a.history[2 * cc]; h2 = a.history[2 * cc + 1]; }
    uint32_t cur = seg_pack(h1, h2);
    uint32_t* ck = a.seg_ckpt + ((uint64_t)S.rows_avail * 2 + X.ch);
    uint32_t kmax = go ? S.seg_count : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)kmax, o); kmax = t > kmax ? t : kmax; }
    for (uint32_t k = 0; k < kmax; k++) {
        const bool on = go && k < S.seg_count;
        uint32_t* rec = a.seg_state + 4 * ((uint64_t)S.first_seg + (uint64_t)(on ? k : 0) * C + X.ch);
        uint32_t end = on ? rec[slot] : 0;
        bool mis = on && rec[0] != cur;                              // the segment's bytes were encoded from rec[0]
        const bool partner_mis = __builtin_amdgcn_update_dpp(0, (int)mis, 0xB1, 0xF, 0xF, true) != 0;
        const bool pmis = mis || (pair && partner_mis);
        const bool run = on && pmis;
        X.k = k; X.r0 = k * S.seg_rows; X.r1 = X.r0 + S.seg_rows < S.frames ? X.r0 + S.seg_rows : S.frames;
        uint32_t nrows = run ? X.r1 - X.r0 : 0, nmax = nrows;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)nmax, o); nmax = t > nmax ? t : nmax; }
        if (run) seg_unpack(cur, h1, h2);
        bool merged = false;
        for (uint32_t t = 0; t < nmax; t++) {
            const bool act = run && t < nrows && !merged;
            if (!__any(act)) break;
            if (act) {
                const uint32_t row = X.r0 + t;
                int32_t x[32];
                enc_lane_load(X, row, x);
                uint32_t word, cw[4];
                enc_lane_block(S, x, h1, h2, word, cw);
                enc_lane_store(X, row, word, cw, pair);
                if (((row + 1) & 3) == 0 || row + 1 == X.r1) {
                    uint32_t* c = ck + (uint64_t)((row + 4) / 4 - 1) * 2;
                    const uint32_t now = seg_pack(h1, h2);
                    const bool mine = *c == now;
                    *c = now;
                    const bool partner_same = __builtin_amdgcn_update_dpp(0, (int)mine, 0xB1, 0xF, 0xF, true) != 0;
                    merged = mine && (!pair || partner_same);
                }
            }
        }
        if (run && !merged) end = seg_pack(h1, h2);
        if (on) cur = end;
    }
}

void launch_adx_encode_lane(const AdxArgs& a, hipStream_t s) {
    if (!a.seg_lanes) return;
    constexpr uint32_t ROUNDS = 4;
    // (A lane's row is one long dependent chain at the SIMD's full issue rate: the launch takes (the most waves any SIMD was given) x (a
    //  lane's rows).  Workgroups of four waves land one wave to a SIMD; single-wave workgroups were found two deep on some SIMDs at 625
    //  waves -- 20 000 files of 1 s: 4.6 ms against 2.8.)
    for (uint32_t pass = 0; pass <= ROUNDS; pass++) hipLaunchKernelGGL(k_adx_lane_encode, dim3((a.seg_lanes + 255) / 256), dim3(256), 0, s, a, pass);
    hipLaunchKernelGGL(k_adx_lane_encode_serial, dim3((a.chains + 63) / 64), dim3(64), 0, s, a, ROUNDS);
}

void launch_adx_encode_seg(const AdxArgs& a, hipStream_t s) {
    if (!a.seg_lanes) return;
    for (uint32_t pass = 0; pass < 3; pass++) {
        const uint32_t blocks = pass < 2 ? a.seg_lanes : a.n_streams;
        hipLaunchKernelGGL(k_adx_seg_encode<2>, dim3(blocks), dim3(64), 0, s, a, pass);
        hipLaunchKernelGGL(k_adx_seg_encode<1>, dim3(blocks), dim3(64), 0, s, a, pass);
    }
}

void launch_adx_decode_seg(const AdxArgs& a, hipStream_t s) {
    if (!a.seg_lanes) return;
    constexpr uint32_t ROUNDS = 6;      // (a round with nothing to repair costs a launch; a chain that is still moving after the last one costs its whole file on the wave-per-file kernel -- 1.6 ms for a 2 s clip, however large the job)
    hipLaunchKernelGGL(k_adx_seg_decode, dim3((a.seg_lanes + 63) / 64), dim3(64), 0, s, a);
    if (a.seg_max_count >= SEG_RUNS_MIN) hipLaunchKernelGGL(k_adx_seg_runs, dim3(a.chains < 4096 ? a.chains : 4096), dim3(64), 0, s, a);
    for (uint32_t r = 0; r < ROUNDS; r++) hipLaunchKernelGGL(k_adx_seg_fix, dim3((a.seg_lanes + 63) / 64), dim3(64), 0, s, a, r, r + 1 == ROUNDS ? 1u : 0u);
    // Flagged chains.  Histories do not always merge: through digital silence the decoder's state just sits where the last sound left
    // it (the recurrence has fixed points away from zero: -100, -100 maps to -100), so a file with silent stretches keeps every
    // segment inside them inconsistent.  Such files are decoded again by the wave-per-file kernel from their first block (8.5 ms for
    // 10 s: what every file cost before the segments); only layouts it does not take (more than two channels) walk their segments
    // lane by lane.
    hipLaunchKernelGGL(k_adx_seg_list, dim3((a.n_streams + 255) / 256), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_adx_decode_wpf, dim3(a.n_streams < 1024 ? a.n_streams : 1024), dim3(64), 0, s, a);
    hipLaunchKernelGGL(k_adx_seg_serial, dim3((a.chains + 63) / 64), dim3(64), 0, s, a, 1 + (ROUNDS & 1));
}

void launch_adx_decode_wpf(const AdxArgs& a, uint32_t n_streams, hipStream_t s) {
    if (n_streams) hipLaunchKernelGGL(k_adx_decode_wpf, dim3(n_streams), dim3(64), 0, s, a);
}
void launch_adx_encode_wpf(const AdxArgs& a, uint32_t n_streams, hipStream_t s) {
    if (!n_streams) return;
    hipLaunchKernelGGL(k_adx_encode_wpf<2>, dim3(n_streams), dim3(64), 0, s, a);
    hipLaunchKernelGGL(k_adx_encode_wpf<1>, dim3(n_streams), dim3(64), 0, s, a);
}

}  // namespace cri
