// cri_hca_enc.hip -- HCA encoder kernel for gfx950 (MI355X, wave64): one wave per (frame, channel), one workgroup per frame
// (or a few frames: they share the LDS copy of the tables).
//
// Replaces EncodeFrame and everything under it (/root/reference/CriCodecs/hca.cpp:2965-2988):
//   PcmToFloat 2470-2479, mdct_transform 2529-2553 + DCT4 2481-2527, EncodeIntensityStereo 2561-2609,
//   CalculateScaleFactors 2611-2637, ScaleSpectra 2639-2654, CalculateHfrGroupAverages 2656-2674, CalculateHfrScale
//   2676-2706, CalculateFrameHeaderLength 2708-2750, CalculateNoiseLevel 2809-2832 / CalculateEvaluationBoundary
//   2852-2866 (both searches over CalculateUsedBits 2763-2790), CalculateFrameResolutions 2868-2876, QuantizeSpectra
//   2878-2892 and PackFrame 2894-2963 (incl. the frame CRC16).
// The frame feeding of Encode/HcaEncode (hca.cpp:2990-3107) reduces to "frame f = samples [1024f, 1024f+1024) of the
// input sequence, with the 128 samples before it as the MDCT history"; for looping input the sequence is zeros / the first
// sample / the main audio / the post-loop audio / zeros (HcaStream::enc_*), and it is all done by indexing.
//
// Shape.  A channel's 8 x 128 spectra are 16 per lane -- band pair (2 * lane, 2 * lane + 1), the order of the bit stream -- and
// stay in registers from ScaleSpectra to the last bit written.  Channels meet in three places only: intensity stereo (a pair's
// spectra, through LDS), the rate loop's bit totals and the bit positions of the pack (a few words through LDS and a workgroup
// barrier each).  The rate loop never quantises: the bits of a spectrum at resolution r are shortest[r] + (its class reaches
// rank[r]) (cri_host.cpp, hca_enc_build_tables) -- every spectrum is classed once, a band's eight classes are eight bytes, and a
// search step costs a band two adds, two ands and two popcounts.
//
// Everything the reference evaluates in floating point is evaluated here with the same single IEEE operations in the
// same order (sequential sums stay sequential, on one lane); bit allocation is integer work; the bit stream is assembled
// with wave prefix sums over code lengths and LDS atomic ORs.
#include <hip/hip_runtime.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <type_traits>
#include "cri_kernels.h"
#include "cri_device.h"
#include "cri_hca_enc_cost.h"
#include "../../include/cricodecs_hip.h"

namespace cri {

// Developer instrumentation (-DCRI_ENC_PROFILE through CRI_HIPCC_EXTRA; never in the shipped build): wave cycles per phase of
// k_hca_encode, summed over all waves
#ifdef CRI_ENC_PROFILE
__device__ unsigned long long g_enc_prof[1024][16];       // spread over 1024 slots so the atomics do not serialise on one address
#define ENC_MARK(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); prof_acc[k] += t_ - prof_t; prof_t = t_; } while (0)
#define ENC_PROF_FLUSH() do { if (lane == 0) for (int k_ = 0; k_ < 16; k_++) atomicAdd(&g_enc_prof[(g * C + c) & 1023][k_], prof_acc[k_]); } while (0)
#elif defined(CRI_ENC_ASM_MARKS)                           // (phase boundaries as comments in the ISA: tools/debug/enc_isa_phases.py counts between them)
#define ENC_MARK(k) asm volatile("; ENC_PHASE_END " #k)
#define ENC_PROF_FLUSH() do {} while (0)
#else
#define ENC_MARK(k) do {} while (0)
#define ENC_PROF_FLUSH() do {} while (0)
#endif

// inclusive prefix sum over the 64 lanes with DPP adds: row_shr 1, 2, 4, 8 inside each row of 16, then row_bcast 15 / 31
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);   // lane 15 of the previous row into rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);   // lane 31 into rows 2, 3
    return v;
}
__device__ __forceinline__ int wave_sum(int v) {          // total of the 64 lanes, wave-uniform (SGPR)
    return __builtin_amdgcn_readlane((int)wave_incl_scan_dpp((uint32_t)v), 63);
}

struct EncTab {            // views into the LDS copy of the table blob (HCA_ET_*, cri_types.h)
    const float *deq, *escale, *inv, *ibounds;
    const float4* win4;
    const f2* tw;
    const uint2* cp;
    const uint8_t *cls, *sfbase, *clen, *code, *ishuf;
};

// hca.cpp:2611-2623: the binary search over the ascending table returns the number of entries 0..62 that are <= v.  The
// table has 128/53 = 2.4 entries per octave, so that count is sfbase[exponent of v] (entries <= 2^exponent, built exactly
// at table-build time) plus at most three more compares -- two dependent LDS reads instead of six.
__device__ __forceinline__ int enc_find_scalefactor(const EncTab& T, float v) {
    int eb = (int)(__float_as_uint(v) >> 23) & 0xFF;
    eb = eb < 102 ? 102 : (eb > 133 ? 133 : eb);            // the table spans 2^-23 .. 2^3.5
    const int base = T.sfbase[eb - 102];
    const float e0 = T.deq[base], e1 = T.deq[base + 1], e2 = T.deq[base + 2];   // deq[] is padded past 63 with NaN
    return base + (e0 <= v ? 1 : 0) + (e1 <= v ? 1 : 0) + (e2 <= v ? 1 : 0);
}
__device__ __forceinline__ int enc_maxbits(int res) { return res > 7 ? res - 3 : (int)((0x44443320u >> (res * 4)) & 15); }

// (u.x*c + u.y*s, u.x*s - u.y*c): the rotation of hca.cpp:2515-2520 / 2493-2496 as three packed operations.  The twiddle is one
// register pair tw = {c, s}, and the operand selects of the packed multiplies pick u.x / u.y and c / s -- written out as
// instructions because the compiler builds the broadcast operands with moves first
__device__ __forceinline__ f2 enc_rot(f2 u, f2 tw) {
    f2 p, q;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(p) : "v"(u), "v"(tw));      // {u.x*c, u.x*s}
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(q) : "v"(u), "v"(tw));      // {u.y*s, u.y*c}
    return pk_add_neg_hi(p, q);
}

// MSB-first write of the `len` bits v (v < 2^len, len <= 26) at absolute bit position p of the frame (words are big-endian
// 32-bit): the bits land in one word or straddle two -- one 64-bit shift serves both cases
__device__ __forceinline__ void put_bits(uint32_t* words, uint32_t p, uint32_t v, uint32_t len) {
    const uint64_t t = (uint64_t)v << ((64u - (p & 31) - len) & 63);
    const uint32_t hi = (uint32_t)(t >> 32), lo = (uint32_t)t;
    if (hi) atomicOr(&words[p >> 5], hi);
    if (lo) atomicOr(&words[(p >> 5) + 1], lo);
}
// ---- LDS of one frame.  Exchange words first, then the frame image, then one region per channel.
#define ENC_X_HBITS 0        // int[8]      header bits per channel
#define ENC_X_STEP 32        // int[2][8]   a search step's bits per channel (two slots, used alternately)
#define ENC_X_TOTA 96        // int[8]      the channels' bits at the final noise level
#define ENC_X_INTEN 128      // uint8[8][8] intensity indices (written by the pair's primary, packed by the secondary)
#define ENC_X_ROWTOT 192     // uint[8 * C] bits of the 8 x C rows of spectra, in stream order (subframe, channel)
#define ENC_X_BYTES(C) (192 + 32 * (C))
#define ENC_CH_SPEC 0        // one piece, three lives: float[1216] the frame's samples of this channel (128 of history first; padded, see ENC_STG_PAD) until the MDCT has read them;
#define ENC_CH_STG 0         // float2[8][72] the MDCT's change of places; float[8][132] the channel's spectra (MDCT output, ENC_SP_ROW); afterwards int[130]: the
                             // boundary search's prefix
// (the region's last 256 bytes hold samples only -- the transposes and the spectra end at 4608 -- so what is written after the MDCT lives there)
#define ENC_CH_HAVG 4608     // float[8]
#define ENC_CH_HFRS 4640     // int[8]
#define ENC_CH_RATIO 4672    // float[8]       (with the two pieces before it: the 24 sums of EncodeIntensityStereo)
#define ENC_CH_SFAC 4704     // uint8[128]
#define ENC_CH_BYTES 4864    // (a multiple of 256: a thread's two stores to neighbouring channels' rows fuse into one ds_write2st64)
// The staged samples, 128 to a block: the MDCT's lanes read one sample each at (subframe) * 128 + (a constant) + 2 * (lane & 7) -- four
// subframes to a group of 32 lanes, all four on the same banks if the blocks lay 128 words apart.  Block k is shifted by (k >> 1) * 16 +
// (k & 1): any four consecutive blocks then start on banks {0, 1, 16, 17} (mod 32), and a group's 32 words fall on 32 banks.
#define ENC_STG_PAD(k) ((((k) >> 1) << 4) | ((k) & 1))
// The spectra, a row of 128 per subframe: rows lie 132 words apart -- a band of all eight subframes (what the MDCT's last stage stores, what
// the sequential sums of intensity stereo and HFR read) is then eight banks, not one
#define ENC_SP_ROW 132

struct EncFmt {
    uint32_t frame_size, total, base, stereo, groups, bpg, hfr_band_count, types;
    __device__ __forceinline__ uint32_t type(uint32_t c) const { return (types >> (2 * c)) & 3u; }
    __device__ __forceinline__ uint32_t coded(uint32_t c) const { return type(c) == CRI_CH_SECONDARY ? base : base + stereo; }
};

// the value of lane - 1 (lane 0: its own), through the LDS crossbar (gfx950 takes the DPP wave shifts in the assembler, but they do not
// shift across rows there -- measured)
__device__ __forceinline__ int wave_shr1(int v) {
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    return __builtin_amdgcn_ds_bpermute((lane > 0 ? lane - 1 : 0) * 4, v);
}

// CalculateFrameHeaderLength, hca.cpp:2708-2750, for one channel.  sf0 / sf1 = the scalefactors of the lane's bands 2 * lane and
// 2 * lane + 1 (0 past the coded range); d0 / d1 = their deltas against the band before (what WriteScalesFactors codes, hca.cpp:2894-2918)
__device__ __forceinline__ void enc_header_length(const EncFmt& F, int sf0, int sf1, uint32_t c, uint32_t lane, int& hbits, int& dbits, int& d0, int& d1) {
    const int coded = (int)F.coded(c);
    // a delta width db codes |delta| <= 2^(db-1) - 1 in db bits and the rest in db + 6: the length is
    // db * (coded - 1) + 6 * (deltas above the limit) -- the deltas above 0, 1, 3, 7, 15 are counted on the scalar unit (a compare per
    // count and band; nothing goes through the vector adders or LDS)
    d0 = sf0 - wave_shr1(sf1); d1 = sf1 - sf0;
    const uint32_t b0 = 2 * lane;
    const uint32_t a0 = b0 >= 1 && (int)b0 < coded ? (uint32_t)(d0 < 0 ? -d0 : d0) : 0u;
    const uint32_t a1 = (int)b0 + 1 < coded ? (uint32_t)(d1 < 0 ? -d1 : d1) : 0u;
    const bool any = __builtin_amdgcn_ballot_w64((sf0 | sf1) != 0) != 0;
    int min_len = 3, min_db = 0;
    if (any) {
        min_db = 6; min_len = 3 + 6 * coded;
#pragma unroll
        for (int db = 1; db < 6; db++) {
            const uint32_t lim = (1u << (db - 1)) - 1;
            const int above = __builtin_popcountll(__builtin_amdgcn_ballot_w64(a0 > lim)) + __builtin_popcountll(__builtin_amdgcn_ballot_w64(a1 > lim));
            const int length = 3 + 6 + db * (coded > 0 ? coded - 1 : 0) + 6 * above;
            if (length < min_len) { min_len = length; min_db = db; }
        }
    }
    if (F.type(c) == CRI_CH_SECONDARY) min_len += 32;
    else if (F.groups > 0) min_len += 6 * (int)F.groups;
    hbits = min_len; dbits = min_db;
}

// register budget = waves per SIMD the kernel is compiled for, by channel count (measured, tools/debug/enc_ablate.sh)
#ifdef ENC_MIN_WAVES_PER_SIMD
#define ENC_WAVES_PER_SIMD(CT) ENC_MIN_WAVES_PER_SIMD
#else
#define ENC_WAVES_PER_SIMD(CT) ((CT) == 1 ? 5 : 6)
#endif
#define ENC_WAVES_PER_SIMD_OF(c) ((uint32_t)ENC_WAVES_PER_SIMD((int)(c)))
#ifndef ENC_MAX_WAVES
#define ENC_MAX_WAVES 4      // waves of a workgroup when a frame has fewer channels than that (mono: 4 frames, stereo: 2)
#endif

// CT = channels of the format (1 .. 8); workgroup = FPG frames x CT waves
template <int CT>
// (waves per SIMD: 80 registers for six -- three dwords spilled in the stereo instance -- wherever the LDS lets six workgroups' worth of
//  waves onto a compute unit (two channels and more: 26 KB per four waves; stereo 155 -> 165 M frames/s against five waves at 90 registers);
//  mono's four frames per workgroup are 28 KB: five)
__global__ __launch_bounds__(64 * (CT > ENC_MAX_WAVES ? CT : (ENC_MAX_WAVES / CT) * CT), ENC_WAVES_PER_SIMD(CT)) void k_hca_encode(HcaEncArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_all[];
    constexpr uint32_t C = CT;
    constexpr bool XCH = C > 1;                            // the frame's waves exchange through LDS, a workgroup barrier each time
    const uint32_t tid = threadIdx.x, lane0 = tid & 63, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t fw = wave / C, c = wave - fw * C;       // frame of the workgroup, channel
    for (uint32_t i = tid; i < HCA_ET_LDS_BYTES / 16; i += blockDim.x) ((uint4*)smem_all)[i] = ((const uint4*)a.tables)[i];
    EncTab T;
    T.win4 = (const float4*)(smem_all + HCA_ET_WIN4); T.tw = (const f2*)(smem_all + HCA_ET_TW); T.deq = (const float*)(smem_all + HCA_ET_DEQ);
    T.escale = (const float*)(smem_all + HCA_ET_ESCALE); T.cp = (const uint2*)(smem_all + HCA_ET_CP); T.cls = a.tables + HCA_ET_CLS; T.inv = (const float*)(smem_all + HCA_ET_INV);
    T.ibounds = (const float*)(smem_all + HCA_ET_IBOUNDS); T.sfbase = smem_all + HCA_ET_SFBASE; T.clen = smem_all + HCA_ET_CLEN;
    T.code = smem_all + HCA_ET_CODE; T.ishuf = smem_all + HCA_ET_ISHUF;

    const HcaFormat* Fp = a.formats + a.format;
    EncFmt F;
    F.frame_size = Fp->frame_size; F.total = Fp->total_bands; F.base = Fp->base_bands; F.stereo = Fp->stereo_bands;
    F.groups = Fp->hfr_group_count; F.bpg = Fp->bands_per_hfr_group; F.hfr_band_count = Fp->hfr_band_count;
    { uint32_t t = 0; for (uint32_t k = 0; k < 16; k++) t |= (uint32_t)(Fp->type[k] & 3) << (2 * k); F.types = t; }
    const uint32_t nwords = (F.frame_size + 3) / 4 + 2;    // (spare words behind the frame: a write's second word)
    uint32_t* wg_vote = (uint32_t*)(smem_all + HCA_ET_LDS_BYTES);   // [2] (+ padding to 16 bytes)
    uint8_t* fr = smem_all + HCA_ET_LDS_BYTES + 16 + fw * a.lds_per_frame;
    int* X_hbits = (int*)(fr + ENC_X_HBITS); int* X_step = (int*)(fr + ENC_X_STEP); int* X_totA = (int*)(fr + ENC_X_TOTA);
    uint32_t* X_rowtot = (uint32_t*)(fr + ENC_X_ROWTOT); uint8_t* X_inten = fr + ENC_X_INTEN;
    uint32_t* words = (uint32_t*)(fr + ENC_X_BYTES(C));
    uint8_t* ch0 = fr + ENC_X_BYTES(C) + ((nwords * 4 + 15) & ~15u);
    uint8_t* chb = ch0 + c * ENC_CH_BYTES;
    float* sp = (float*)(chb + ENC_CH_SPEC);
    float* stg = (float*)(chb + ENC_CH_STG);
    uint8_t* sfac = chb + ENC_CH_SFAC;
    float* havg = (float*)(chb + ENC_CH_HAVG); int* hfrs = (int*)(chb + ENC_CH_HFRS);

    // Persistent workgroups: the launch has as many workgroups as the chip holds at once (or fewer), and each walks its share of the
    // frame groups -- the tables are copied, the format read and the pointers made once per workgroup, not once per two frames.
    for (uint32_t wgrp = blockIdx.x; wgrp < a.groups; wgrp += gridDim.x) {
        // (the lane number is made opaque per round: everything derived from it -- addresses, masks -- is then computed where it is used, as
        //  in a kernel without the loop, instead of being hoisted out and held in registers across the whole body: 140 spills otherwise)
        uint32_t lane = lane0;
        asm volatile("" : "+v"(lane));
        const uint32_t tidf = c * 64 + lane;               // thread within the frame
        uint32_t g = wgrp * a.frames_per_group + fw;
        if (g >= a.frames) g = a.frames - 1;                   // a spare frame slot of the last workgroup repeats the last frame (same bytes, same place)
        g = (uint32_t)__builtin_amdgcn_readfirstlane((int)g);  // (wave-uniform: the stream's fields below belong in scalar registers)
        if (tid < 2) wg_vote[tid] = 0;
#ifdef CRI_ENC_PROFILE
        unsigned long long prof_acc[16] = {0}; unsigned long long prof_t = __builtin_readcyclecounter();
#endif
        for (uint32_t i = tidf; i < nwords; i += 64 * C) words[i] = 0;
        if (XCH && tidf < 16) X_step[tidf] = 0;                // the search steps' sums (below)

        // frame -> stream
        uint32_t lo = a.stream_hint[g >> 4];                   // the stream of frame g & ~15; g's own is that one or one of the next few
        while (lo + 1 < a.stream_end && a.streams[lo + 1].first_frame <= g) lo++;
        lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)lo);
        const HcaStream st = a.streams[lo];
        const uint32_t f = g - st.first_frame;
        const uint8_t* pcm = (st.src_in_scratch ? a.scratch : a.in) + st.src_offset;

        // ---- the frame's samples, all channels, into the channels' staging rows: sample rel (-128 .. 1023, relative to the frame's
        // first sample) of channel k at staging row k, index rel + 128; zero outside the stream.  The frame's C waves load together.
        {
            const uint64_t F0 = (uint64_t)f * 1024;
            const int64_t nsamp = (int64_t)st.samples;
            // plain streams: readable rel range [rlo, rhi) of this frame and the address of rel = rlo
            const int rlo = F0 >= 128 ? -128 : -(int)F0;
            const int64_t hi64 = nsamp - (int64_t)F0;
            const int rhi = hi64 > 1024 ? 1024 : (hi64 < -128 ? -128 : (int)hi64);
            const uint8_t* fbase = pcm + (F0 + (int64_t)rlo) * C * 2;
            // (staged as floats: a sample is read by two subframes' folds -- converting here is 18 conversions per lane instead of 32 --
            //  and a row's stores are whole words of consecutive lanes)
            auto stage = [&](uint32_t k, uint32_t s, uint32_t blk, float v) { ((float*)(ch0 + k * ENC_CH_BYTES + ENC_CH_STG))[s + ENC_STG_PAD(blk)] = v; };   // blk = s >> 7
            if (!st.enc_loop && rlo == -128 && rhi == 1024) {      // the whole window lies inside the stream: 576 * C dwords, 9 per thread
                uint32_t dw[9];
#pragma unroll
                for (int j = 0; j < 9; j++) dw[j] = ld_u32_unaligned(fbase + 4 * (tidf + 64 * C * j));
#pragma unroll
                for (int j = 0; j < 9; j++) {
                    const uint32_t e0 = 2 * (tidf + 64 * C * j);
                    // (a thread's samples of round j lie in block j: its dwords start at sample frame (tidf + 64 C j) * 2 / C, and tidf * 2 / C < 128)
                    const float v0 = cvt_f32_i16<0>(dw[j]), v1 = cvt_f32_i16<1>(dw[j]);
                    if constexpr (C == 1) { stage(0, e0, j, v0); stage(0, e0 + 1, j, v1); }
                    else {
                        const uint32_t s0 = e0 / C, k0 = e0 - s0 * C;
                        stage(k0, s0, j, v0);
                        if constexpr (C % 2 == 0) stage(k0 + 1, s0, j, v1);
                        else { const uint32_t s1 = (e0 + 1) / C, k1 = e0 + 1 - s1 * C; stage(k1, s1, j, v1); }
                    }
                }
            } else {                                               // stream edges, loop streams
                const bool any_plain = rhi > rlo, have_any = st.enc_have > 0;
                for (uint32_t i = tidf; i < 1152 * C; i += 64 * C) {
                    const uint32_t s = i / C, k = i - s * C;
                    const int rel = (int)s - 128;
                    int16_t v = 0;
                    if (!st.enc_loop) {
                        if (any_plain && rel >= rlo && rel < rhi) __builtin_memcpy(&v, fbase + ((uint32_t)(rel - rlo) * C + k) * 2, 2);
                    } else if (have_any) {                         // the feeding sequence of hca.cpp:2990-3107 (see cri_types.h)
                        const int64_t n = (int64_t)F0 + rel, m = n - (int64_t)st.enc_pre, e = m - nsamp;
                        const bool in_pre = m < 0, in_main = !in_pre && m < nsamp, in_post = !in_pre && !in_main && e < (int64_t)st.enc_post;
                        const int64_t src = in_pre ? 0 : (in_main ? m : (int64_t)st.enc_loop_src + e);
                        const bool ok = n >= (int64_t)st.enc_pre_zero && (in_pre || in_main || (in_post && src < (int64_t)st.enc_loop_src_end)) && src < (int64_t)st.enc_have;
                        if (ok) __builtin_memcpy(&v, pcm + ((uint64_t)src * C + k) * 2, 2);
                    }
                    stage(k, s, s >> 7, (float)(int)v);
                }
            }
        }
        ENC_MARK(0);
        __syncthreads();                                       // tables, zeroed frame image, staged samples
        ENC_MARK(1);

        // ---- MDCT of the channel's 8 subframes: hca.cpp:2529-2553 (window + fold), 2481-2527 (DCT-IV), all eight at once in registers.
        // Subframe = lane >> 3; its 8 lanes hold the 64 complex points of the reference's in-place radix-2 network, 8 per lane.
        // First as point j = l8 + 8 * r in z[r]: the stages on bits 5, 4, 3 of j pair registers; then the points change places
        // through LDS (j = 8 * l8 + r) and the stages on bits 2, 1, 0 pair registers again -- no stage exchanges between lanes.
        // A stage keeps the sum in the lower point and rotates the difference into the upper one (twiddle row = bit, index = the
        // bits of j below it).  The window is held times 2^-15 (PcmToFloat's scale, hca.cpp:2470-2479: a power of two commutes with
        // the rounding of the product).
        {
            const uint32_t l8 = lane & 7, sfm = lane >> 3;
            // the subframe's 256 samples [n0 - 128, n0 + 128): the first 128 in block sfm, the others in block sfm + 1 (ENC_STG_PAD)
            const float* swa = stg + sfm * 128 + ENC_STG_PAD(sfm);
            const float* swb = stg + sfm * 128 + ENC_STG_PAD(sfm + 1);
            f2 z[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                // folded inputs k = 2j (even) and 127 - 2j (odd) of point j: hca.cpp:2532-2547 -- for k < 64 (even inputs of r < 4, odd ones
                // of r >= 4)  -w[63 - k] x[192 + k] + w[64 + k] x[191 - k],  else  w[k - 64] x[k - 64] + w[191 - k] x[191 - k];  the four
                // factors, signs included, are one row of the window table: two packed multiplies and a packed add
                const int ke = 2 * (int)l8 + 16 * r, ko = 127 - ke;
                const float4 w4 = T.win4[8 * r + l8];
                // (which half an index falls in depends on r alone: 192 + ke >= 128 always; ko - 64 = 63 - ke, ke - 64, 191 - ke, 191 - ko
                //  = 64 + ke and 192 + ko = 319 - ke cross at r = 4)
                f2 xa, xb;
                if (r < 4) { xa = f2{swb[192 + ke], swa[ko - 64]}; xb = f2{swb[191 - ke], swa[191 - ko]}; }
                else { xa = f2{swa[ke - 64], swb[192 + ko]}; xb = f2{swa[191 - ke], swb[191 - ko]}; }
                const f2 in = f2{w4.x, w4.y} * xa + f2{w4.z, w4.w} * xb;
                z[r] = enc_rot(in, T.tw[l8 + 8 * r]);
            }
    #define ENC_BFLY(LO, HI, TW) { const f2 d_ = z[LO] - z[HI]; z[LO] = z[LO] + z[HI]; z[HI] = enc_rot(d_, TW); }
            {   // bit 5: (z[r], z[r + 4]), twiddle [5][l8 + 8 r]
#pragma unroll
                for (int r = 0; r < 4; r++) ENC_BFLY(r, r + 4, T.tw[64 + l8 + 8 * r])
            }
            {   // bit 4: (z[r], z[r + 2]), twiddle [4][l8 + 8 (r & 1)]
                const f2 t0 = T.tw[96 + l8], t1 = T.tw[96 + l8 + 8];
                ENC_BFLY(0, 2, t0) ENC_BFLY(1, 3, t1) ENC_BFLY(4, 6, t0) ENC_BFLY(5, 7, t1)
            }
            {   // bit 3: (z[r], z[r + 1]), twiddle [3][l8]
                const f2 t0 = T.tw[112 + l8];
                ENC_BFLY(0, 1, t0) ENC_BFLY(2, 3, t0) ENC_BFLY(4, 5, t0) ENC_BFLY(6, 7, t0)
            }
            // change of places: point j sits at slot j + (j >> 3) of the subframe's 72 (the padding keeps both the stores --
            // l8 + 9 r -- and the loads -- 9 l8 + r -- off each other's banks).  The buffer lies over the channel's spectra and
            // staging rows: every sample has been read by now, and the spectra are stored after the last load below.
            f2* tb = (f2*)chb + sfm * 72;
            wave_lds_sync();
#pragma unroll
            for (int r = 0; r < 8; r++) tb[l8 + 9 * r] = z[r];
            wave_lds_sync();
#pragma unroll
            for (int r = 0; r < 8; r++) z[r] = tb[9 * l8 + r];
            wave_lds_sync();
            {   // bit 2: (z[r], z[r + 4]), twiddle [2][r]
#pragma unroll
                for (int r = 0; r < 4; r++) ENC_BFLY(r, r + 4, T.tw[120 + r])
            }
            {   // bit 1: (z[r], z[r + 2]), twiddle [1][r & 1]
                const f2 t0 = T.tw[124], t1 = T.tw[125];
                ENC_BFLY(0, 2, t0) ENC_BFLY(1, 3, t1) ENC_BFLY(4, 6, t0) ENC_BFLY(5, 7, t1)
            }
            {   // bit 0: (z[r], z[r + 1]), twiddle [0][0]
                const f2 t0 = T.tw[126];
                ENC_BFLY(0, 1, t0) ENC_BFLY(2, 3, t0) ENC_BFLY(4, 5, t0) ENC_BFLY(6, 7, t0)
            }
    #undef ENC_BFLY
            // point j = 8 l8 + r holds spectrum lines ishuf[2j], ishuf[2j + 1] (the inverse of the final shuffle), scaled by 1/8
            const uint4 op = *(const uint4*)(T.ishuf + 16 * l8);
            const uint32_t opw[4] = {op.x, op.y, op.z, op.w};
            float* out = sp + sfm * ENC_SP_ROW;
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const f2 o = z[r] * f2{0.125f, 0.125f};
                const uint32_t w = opw[r >> 1] >> (16 * (r & 1));
                out[w & 0xFF] = o.x; out[(w >> 8) & 0xFF] = o.y;
            }
        }
        const uint32_t mytype = F.type(c);
        ENC_MARK(2);

        // ---- EncodeIntensityStereo, hca.cpp:2561-2609 (sequential sums: one lane per subframe), by the pair's primary wave
        if (C > 1 && F.stereo > 0) {
            __syncthreads();                                   // both channels' spectra are in LDS
            if (mytype == CRI_CH_PRIMARY && c + 1 < C) {
                float* lsp = sp; float* rsp = (float*)(chb + ENC_CH_BYTES + ENC_CH_SPEC);
                // the three sums of a subframe are three chains of sequential adds (hca.cpp:2571-2577): a lane each -- lanes 0-7 sum |l|,
                // 8-15 |r|, 16-23 |l + r| -- eight bands fetched ahead of the adds
                float* sums = havg;                                // [24] (the HFR averages use this piece later)
                if (lane < 24) {
                    const uint32_t sfl = lane & 7, kind = lane >> 3;
                    const float* l = lsp + sfl * ENC_SP_ROW; const float* r = rsp + sfl * ENC_SP_ROW;
                    // |l|, |r| or |l + r| as |l * ml + r * mr| with factors 1 / 0: the products are exact and x + 0 is x, so the three kinds
                    // are one instruction sequence (written with selects, the compiler made a branch and a wait per band and kind out of it:
                    // this phase took a third of a Low-quality frame's time)
                    const float ml = kind == 1 ? 0.0f : 1.0f, mr = kind == 0 ? 0.0f : 1.0f;
                    float acc = 0;
                    uint32_t b = F.base;
                    for (; b + 8 <= F.total; b += 8) {
                        float t[8];
#pragma unroll
                        for (int k = 0; k < 8; k++) t[k] = fabsf(l[b + k] * ml + r[b + k] * mr);
#pragma unroll
                        for (int k = 0; k < 8; k++) acc += t[k];
                    }
                    for (; b < F.total; b++) acc += fabsf(l[b] * ml + r[b] * mr);
                    sums[lane] = acc;
                }
                wave_lds_sync();
                float myratio = 1.0f;                              // lanes 0-7: the subframe's ratio
                if (lane < 8) {
                    const float el = sums[lane], er = sums[8 + lane];
                    float et = sums[16 + lane];
                    et *= 2;
                    const float elr = er + el;
                    const float stored = 2 * el / elr;
                    float ratio = elr / et;
                    if (ratio < 0.5) ratio = 0.5f;
                    else if ((double)ratio > sqrt(2.0) / 2) ratio = (float)(sqrt(2.0) / 2);
                    // the first entry of the descending table below the stored value (hca.cpp:2591-2593) = one more than the entries 1 .. 12 at or above it
                    int q = 1;
#pragma unroll
                    for (int k = 1; k < 13; k++) q += T.ibounds[k] >= stored ? 1 : 0;
                    if (!(er > 0 || el > 0)) { q = 0; ratio = 1; }
                    X_inten[(c + 1) * 8 + lane] = (uint8_t)q;
                    myratio = ratio;
                }
                // (l + r) * ratio into the primary, zeros into the secondary: a lane takes a band of all eight subframes, everything
                // fetched before anything is stored
                float rt[8];
#pragma unroll
                for (int sf = 0; sf < 8; sf++) rt[sf] = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(myratio), sf));
                for (uint32_t b = F.base + lane; b < F.total; b += 64) {
                    float sv[8];
#pragma unroll
                    for (int sf = 0; sf < 8; sf++) sv[sf] = lsp[sf * ENC_SP_ROW + b] + rsp[sf * ENC_SP_ROW + b];
#pragma unroll
                    for (int sf = 0; sf < 8; sf++) { lsp[sf * ENC_SP_ROW + b] = sv[sf] * rt[sf]; rsp[sf * ENC_SP_ROW + b] = 0; }
                }
            }
            __syncthreads();
        } else wave_lds_sync();

        ENC_MARK(3);
        // ---- CalculateHfrGroupAverages, hca.cpp:2656-2674 (sequential sums: one lane per group).  It reads the unscaled
        //      spectra of the bands above the coded range.
        const int hfr_start = (int)(F.stereo + F.base);
        const bool hfr = F.groups > 0 && mytype != CRI_CH_SECONDARY;
        if (hfr) {
            const int bpg = (int)F.bpg;
            if (lane < F.groups) {
                const int grp = (int)lane;
                float sum = 0.0f; int count = 0;
                for (int i = 0; i < bpg; i++) {
                    const int band = hfr_start + grp * bpg + i;
                    if (band >= 128) break;
                    for (int sf = 0; sf < 8; sf++) sum += fabsf(sp[sf * ENC_SP_ROW + band]);
                    count += 8;
                }
                havg[grp] = sum / (float)count;
            }
        }

        // ---- CalculateScaleFactors + ScaleSpectra, hca.cpp:2625-2654: bands 2 * lane, 2 * lane + 1 -> registers
        const uint32_t coded = F.coded(c);
        const uint32_t b0 = 2 * lane, b1 = b0 + 1;
        f2 xr[8];                                              // xr[subframe] = {band b0, band b1}, scaled
        uint32_t cl[2][2];                                     // cl[band]: the classes (cri_host.cpp) of its 8 spectra, a byte each
        int sfr[2]; uint32_t ntop[2];
        {
            float m0 = 0, m1 = 0;
#pragma unroll
            for (int sf = 0; sf < 8; sf++) {
                xr[sf] = *(const f2*)(sp + sf * ENC_SP_ROW + b0);
                m0 = fmaxf(m0, fabsf(xr[sf].x)); m1 = fmaxf(m1, fabsf(xr[sf].y));      // (the largest magnitude, hca.cpp:2627-2631: no NaNs here)
            }
            uint32_t s0 = (uint32_t)enc_find_scalefactor(T, m0), s1 = (uint32_t)enc_find_scalefactor(T, m1);
            s0 = b0 < coded ? s0 : 0u; s1 = b1 < coded ? s1 : 0u;
            sfr[0] = (int)s0; sfr[1] = (int)s1;
            *(uint16_t*)(sfac + b0) = (uint16_t)(s0 | s1 << 8);
            // ScaleSpectra, hca.cpp:2639-2654: a band without a scalefactor (and every band past the coded range) is zeros -- the table's
            // entry 0 is 0.0 for that (cri_host.cpp; a zero of either sign: nothing below tells them apart)
            const f2 e01 = f2{T.escale[s0], T.escale[s1]};
#pragma unroll
            for (int sf = 0; sf < 8; sf++) {
                const f2 v = xr[sf] * e01;
                xr[sf] = f2{__builtin_amdgcn_fmed3f(v.x, -0.9999999f, 0.9999999f), __builtin_amdgcn_fmed3f(v.y, -0.9999999f, 0.9999999f)};   // hca.cpp:2646-2649 (the products are never NaN)
            }
            const float e0 = e01.x, e1 = e01.y;
            // values that sit on the clamp (the quantiser's one irregular input, cri_host.cpp): counted per band, in the rare frame that has any
            ntop[0] = ntop[1] = 0;
            if (__builtin_amdgcn_ballot_w64(m0 * e0 >= 0.9999999f || m1 * e1 >= 0.9999999f) != 0) {
#pragma unroll
                for (int sf = 0; sf < 8; sf++) {
                    ntop[0] += enc_on_clamp(xr[sf].x); ntop[1] += enc_on_clamp(xr[sf].y);
                }
            }
            // class of every spectrum: how many of the fifteen resolutions' thresholds (of its sign) it reaches
            cl[0][0] = cl[0][1] = cl[1][0] = cl[1][1] = 0;
#pragma unroll
            for (int sf = 0; sf < 8; sf++) {
#pragma unroll
                for (int b = 0; b < 2; b++) {
                    const uint32_t k = enc_class(T.cls, b ? xr[sf].y : xr[sf].x);
                    cl[b][sf >> 2] |= k << (8 * (sf & 3));
                }
                __builtin_amdgcn_sched_barrier(0);             // (a few rows in flight, not all sixteen: registers)
            }
        }
        const bool any_top = __builtin_amdgcn_ballot_w64((ntop[0] | ntop[1]) != 0) != 0;
        wave_lds_sync();

        // ---- CalculateHfrScale, hca.cpp:2676-2706: sequential sums over the scaled spectra of the bands below the HFR range
        if (hfr) {
#pragma unroll
            for (int sf = 0; sf < 8; sf++) *(f2*)(sp + sf * ENC_SP_ROW + b0) = xr[sf];
            wave_lds_sync();
            const int bpg = (int)F.bpg;
            const int hb = (int)(F.hfr_band_count < F.total - F.hfr_band_count ? F.hfr_band_count : F.total - F.hfr_band_count);
            if (lane < F.groups) {
                const int grp = (int)lane;
                float sum = 0.0f; int count = 0;
                for (int i = 0; i < bpg; i++) {
                    const int band = grp * bpg + i;
                    if (band >= hb) break;
                    for (int sf = 0; sf < 8; sf++) sum += fabsf(sp[sf * ENC_SP_ROW + (hfr_start - band - 1)]);
                    count += 8;
                }
                const float avg = sum / (float)count;
                float gs = havg[grp];
                if (avg > 0.0) {
                    const double m = 1.0 / (double)avg, r2 = sqrt(2.0);
                    gs = (float)((double)gs * (m < r2 ? m : r2));
                }
                hfrs[grp] = enc_find_scalefactor(T, gs);
            }
            wave_lds_sync();
        }

        ENC_MARK(4);
        // ---- rate loop: CalculateNoiseLevel, CalculateEvaluationBoundary (hca.cpp:2792-2866)
        // A band sits at curve position noise - kb; the table row of that position holds its resolution's thresholds and shortest code.
        int kb[2]; bool live[2];
        auto band_setup = [&]() {
#pragma unroll
            for (int b = 0; b < 2; b++) { kb[b] = 5 * sfr[b] / 2 - 2; live[b] = (b ? b1 : b0) < coded && sfr[b] != 0; }
        };
        auto row_of = [&](int noise, int b) -> uint2 {
            int cp = noise - kb[b];
            cp = cp < 0 ? 0 : (cp > 58 ? 58 : cp);
            return T.cp[live[b] ? cp : 59];
        };
        // the bits of band b's 8 spectra at one noise level (the inner part of CalculateUsedBits, hca.cpp:2771-2786): 8 times the
        // resolution's shortest code plus the spectra whose class reaches the resolution's rank, less what was counted for values the
        // quantiser pushes past its table (cri_host.cpp)
        auto band_bits = [&](int noise, int b) -> int { return enc_band_cost(row_of(noise, b), cl[b][0], cl[b][1], ntop[b], any_top); };
        // a search step's share of this lane: both bands at one level, as a raw sum (enc_band_cost_raw: the bits are its low 20 bits)
        auto lane_raw = [&](int noise, auto tops_c) -> uint32_t {
            const uint2 r0 = row_of(noise, 0), r1 = row_of(noise, 1);
            uint32_t n = enc_band_cost_raw(r0, cl[0][0], cl[0][1], r0.y);
            n = enc_band_cost_raw(r1, cl[1][0], cl[1][1], n + r1.y);
            if constexpr (decltype(tops_c)::value) {               // (the rare frame with values on the clamp)
                n -= (r0.y >> 28) & 1 ? ntop[0] * (((r0.y & 0xFF) >> 3) + 1) : 0u;
                n -= (r1.y >> 28) & 1 ? ntop[1] * (((r1.y & 0xFF) >> 3) + 1) : 0u;
            }
            return n;
        };
        int hbits_c = 0, dbits_c = 0, dd0 = 0, dd1 = 0;
        const int avail = (int)F.frame_size * 8;
        int noise_level = -1, eval_boundary = 0, status = 0, hbtot = 0;
        {
            // The workgroup's frames pass the same barriers: a frame whose search has ended keeps walking through the rounds the others
            // still need (nothing of its state changes), and the rounds end when no frame asks for another one
            const bool vote = XCH && a.frames_per_group > 1;
            const bool quad_last = (lane & 15) == 15;         // the last lane of a row of sixteen holds the row's sum
            int highest = (int)(F.base + F.stereo) - 1;
            bool done = false;
            for (uint32_t round = 0;; round++) {
                if (!done) { enc_header_length(F, sfr[0], sfr[1], c, lane, hbits_c, dbits_c, dd0, dd1); band_setup(); }
                if (XCH) {
                    if (lane == 0) X_hbits[c] = hbits_c;
                    if (c == 0 && lane < 8) X_step[((round + 1) & 1) * 8 + lane] = 0;      // the next round's sums (last read before the round before this one ended)
                    __syncthreads();
                    hbtot = 16 + 16 + 16;
#pragma unroll
                    for (uint32_t k = 0; k < C; k++) hbtot += X_hbits[k];
                } else hbtot = 16 + 16 + 16 + hbits_c;
                int low = 0, high = done ? 0 : 255;
                bool over = false;                             // "mid_value > available bits" of the last step (hca.cpp:2806-2815)
                // A step's bits of the whole frame: every lane's two bands, summed over the lanes and the channels.  Channels that share the
                // frame add into ONE word of LDS per step -- four DPP adds leave the sum of a row of sixteen in its last lane, four lanes per
                // wave add theirs with an LDS atomic (same-address atomics take an LDS cycle per lane: sixteen quads' worth cost more there
                // than the two further adds here) -- and read the total back behind the step's barrier
                auto steps = [&](auto tops_c) {
                    int* slot = X_step + (round & 1) * 8;
#pragma unroll
                    for (int step = 0; step < 8; step++) {     // 256 levels: always 8 steps
                        const int mid = (low + high) / 2;
                        int bits;
                        if (XCH) {
                            if (!done) {
                                uint32_t raw = lane_raw(mid, tops_c);
                                raw += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)raw, 0x111, 0xF, 0xF, true);
                                raw += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)raw, 0x112, 0xF, 0xF, true);
                                raw += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)raw, 0x114, 0xF, 0xF, true);
                                raw += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)raw, 0x118, 0xF, 0xF, true);
                                // (the instruction itself: the compiler's atomic optimiser would turn atomicAdd into a scalar loop over the lanes;
                                //  with its own wait, because the compiler does not count an LDS operation it cannot see before the barrier)
                                if (quad_last) asm volatile("ds_add_u32 %0, %1\n\ts_waitcnt lgkmcnt(0)" :: "v"(lds_address(&slot[step])), "v"(raw) : "memory");
                            }
                            __syncthreads();
                            bits = __builtin_amdgcn_readfirstlane(slot[step]) & (int)ENC_BITS_MASK;
                        } else bits = done ? 0 : (int)((uint32_t)wave_sum((int)lane_raw(mid, tops_c)) & ENC_BITS_MASK);
                        if (!done) { over = hbtot + bits > avail; if (over) low = mid + 1; else high = mid; }
                    }
                };
                if (any_top) steps(std::true_type{}); else steps(std::false_type{});
                if (!done) {
                    noise_level = (low == 255 && over) ? -1 : low;
                    if (noise_level >= 0) done = true;
                    else { highest -= 2; if (highest < 0) { status = CRI_ERR_HCA_ENCODE; done = true; } }
                }
                if (vote) {
                    if (tid == 0) wg_vote[(round + 1) & 1] = 0;
                    if (!done && lane == 0) wg_vote[round & 1] = 1;
                    __syncthreads();                           // (also: everyone has read this round's header bits)
                    if (!wg_vote[round & 1]) break;
                } else {
                    if (done) break;
                    if (XCH) __syncthreads();                  // everyone has read this round's header bits
                }
                if (!done) {
                    if (lane == 0) { sfac[highest + 1] = 0; sfac[highest + 2] = 0; }
                    wave_lds_sync();
                    sfr[0] = sfac[b0]; sfr[1] = sfac[b1];
                }
            }
        }
        ENC_MARK(5);
        uint32_t* P = (uint32_t*)sp;                           // the channel's spectra region is free from here on
        {
            // only two resolutions per band occur in this search (noise_level and noise_level - 1): cost them once, then the bits at an
            // evaluation boundary eb are (everything at noise_level) + (sum over the bands below eb of the difference)
            const bool search = status == 0 && noise_level != 0;
            int a0 = 0, a1 = 0, d0 = 0, d1 = 0;
            if (search) {
                a0 = band_bits(noise_level, 0); a1 = band_bits(noise_level, 1);
                d0 = band_bits(noise_level - 1, 0) - a0; d1 = band_bits(noise_level - 1, 1) - a1;
            }
            const uint32_t incl = wave_incl_scan_dpp((uint32_t)(d0 + d1));
            const int totA = wave_sum(a0 + a1);
            int v0, v1;                                        // bits at eb = b0, b1
            if (XCH) {
                *(uint2*)(P + b0) = uint2{incl - (uint32_t)(d0 + d1), incl - (uint32_t)d1};
                if (lane == 0) X_totA[c] = totA;
                __syncthreads();
                v0 = v1 = hbtot;
#pragma unroll
                for (uint32_t k = 0; k < C; k++) {
                    const uint2 p = *(const uint2*)((const uint32_t*)(ch0 + k * ENC_CH_BYTES + ENC_CH_SPEC) + b0);
                    const int t = X_totA[k];
                    v0 += t + (int)p.x; v1 += t + (int)p.y;
                }
            } else { v0 = hbtot + totA + (int)(incl - (uint32_t)(d0 + d1)); v1 = hbtot + totA + (int)(incl - (uint32_t)d1); }
            if (search) {
                const uint64_t over_even = __builtin_amdgcn_ballot_w64(v0 > avail), over_odd = __builtin_amdgcn_ballot_w64(v1 > avail);
                auto over_at = [&](int eb) -> bool { return (((eb & 1) ? over_odd : over_even) >> (eb >> 1)) & 1; };
                int low = 0, high = 127;
                while ((high - low > 1) || (low - high > 1)) {
                    const int mid = (low + high) / 2;
                    if (over_at(mid)) high = mid - 1; else low = mid;
                }
                int level;
                if (low == high) level = low < 127 ? low : -1;
                else level = over_at(high) ? low : high;
                if (level < 0) status = CRI_ERR_HCA_ENCODE; else eval_boundary = level;
            }
        }
        ENC_MARK(6);
        uint8_t* dst = a.out + st.dst_offset + (uint64_t)f * F.frame_size;
        if (status != 0) {                                     // (the same decision in every wave of the frame)
            if (c == 0) {
                if (lane == 0 && a.status) atomicMin(a.status + st.item, status);
                for (uint32_t i = lane; i < F.frame_size; i += 64) dst[i] = 0;
            }
        }

        // ---- CalculateFrameResolutions (hca.cpp:2868-2876), PackFrame (hca.cpp:2938-2963): sync word, 9+7 bit header, then per
        //      channel scalefactors + intensity / HFR scales
        int rb[2];
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int i = (int)(b ? b1 : b0);
            const uint2 t = row_of(i < eval_boundary ? noise_level - 1 : noise_level, b);
            rb[b] = (int)((t.y >> 20) & 15);                   // 0 for bands that are not coded or have no scalefactor
        }
        uint32_t pos = 32;
        if (XCH) { for (uint32_t k = 0; k < c; k++) pos += (uint32_t)X_hbits[k]; }
        if (status == 0) {
            if (c == 0 && lane == 0) atomicOr(&words[0], 0xFFFF0000u | (uint32_t)noise_level << 7 | (uint32_t)eval_boundary);   // sync, 9 + 7 bits
            const int db = dbits_c;
            if (lane == 0) put_bits(words, pos, (uint32_t)db, 3);
            pos += 3;
            if (db == 6) {
                for (int i = (int)lane; i < (int)coded; i += 64) put_bits(words, pos + 6 * i, sfac[i], 6);
                pos += 6 * coded;
            } else if (db != 0) {                              // WriteScalesFactors, hca.cpp:2894-2918
                const int maxd = (1 << (db - 1)) - 1, esc = (1 << db) - 1;
                // the lane's two bands' codes (a delta, or the escape value followed by the 6-bit scalefactor; band 0 is always the
                // plain 6 bits) leave as one write of at most 22 bits
                int len0 = 0, len1 = 0;
                uint32_t c0 = 0, c1 = 0;
                if (b0 < coded) {
                    const int dd = b0 == 0 ? 0 : dd0;
                    const bool e0 = (dd < 0 ? -dd : dd) > maxd;
                    len0 = b0 == 0 ? 6 : (e0 ? db + 6 : db);
                    c0 = b0 == 0 ? (uint32_t)sfr[0] : (e0 ? (((uint32_t)esc << 6) | (uint32_t)sfr[0]) : (uint32_t)(maxd + dd));
                }
                if (b1 < coded) {
                    const int dd = dd1;
                    const bool e1 = (dd < 0 ? -dd : dd) > maxd;
                    len1 = e1 ? db + 6 : db;
                    c1 = e1 ? (((uint32_t)esc << 6) | (uint32_t)sfr[1]) : (uint32_t)(maxd + dd);
                }
                const uint32_t incl01 = wave_incl_scan_dpp((uint32_t)(len0 + len1));
                put_bits(words, pos + (incl01 - (uint32_t)(len0 + len1)), (c0 << len1) | c1, (uint32_t)(len0 + len1));
                pos += (uint32_t)__builtin_amdgcn_readlane((int)incl01, 63);
            }
            if (mytype == CRI_CH_SECONDARY) {
                if (lane < 8) put_bits(words, pos + 4 * lane, X_inten[c * 8 + lane], 4);
            } else if (F.groups > 0) {
                if (lane < F.groups) put_bits(words, pos + 6 * lane, (uint32_t)hfrs[lane], 6);
            }
        }

        ENC_MARK(7);
        // ---- spectra: QuantizeSpectra (hca.cpp:2878-2892) + WriteSpectra (2920-2936).  A row (subframe, channel) of the stream is
        // this wave's 64 band pairs: the two codes of a lane as one word, their place by a prefix sum over the lanes; the rows' totals
        // go through LDS so that every wave knows where its rows start.
        uint32_t both[8], incl[8];                             // per subframe: the lane's two codes, lengths' inclusive prefix | own length << 16
        {
            // Resolutions 1 .. 7 take their code from the table (index = resolution * 16 + quantiser index); 8 .. 15 write sign and magnitude:
            // resolution - 4 bits for a zero, one more otherwise, (|q| << 1 | sign).  One formula serves both: the table's rows 8 .. 15 hold the
            // length of a zero and no code, `wide` adds the bit of a non-zero value and `wmask` lets the sign-magnitude code through.
            int downb[2]; float invb[2], upb[2]; uint32_t wmask[2], r16[2]; bool wide[2];
#pragma unroll
            for (int b = 0; b < 2; b++) {
                invb[b] = T.inv[rb[b]]; upb[b] = invb[b] + 1; downb[b] = (int)((double)invb[b] + 0.5);
                wide[b] = rb[b] >= 8; wmask[b] = wide[b] ? (2u << (rb[b] - 4)) - 1 : 0u; r16[b] = (uint32_t)rb[b] * 16;
            }
#pragma unroll
            for (int sf = 0; sf < 8; sf++) {
                uint32_t code[2], len[2];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int q = (int)((h ? xr[sf].y : xr[sf].x) * invb[h] + upb[h]) - downb[h];
                    const uint32_t ti = (((uint32_t)q + 8) & 15) | r16[h];
                    const uint32_t mag = (uint32_t)(q < 0 ? -q : q);
                    len[h] = (uint32_t)T.clen[ti] + ((wide[h] && q != 0) ? 1u : 0u);
                    code[h] = (((mag << 1) | ((uint32_t)q >> 31)) & wmask[h]) | T.code[ti];
                }
                const uint32_t tl = len[0] + len[1];                           // at most 13 + 13 bits
                both[sf] = (code[0] << len[1]) | code[1];
                incl[sf] = wave_incl_scan_dpp(tl) | tl << 16;                  // a row is at most 64 * 26 bits
            }
        }
        ENC_MARK(8);
        uint32_t rowbase[8];
        if (XCH) {
            if (lane == 63) {
#pragma unroll
                for (int sf = 0; sf < 8; sf++) X_rowtot[sf * C + c] = incl[sf] & 0xFFFF;
            }
            __syncthreads();
            const uint32_t mine = lane < 8 * C ? X_rowtot[lane] : 0u;          // stream order: subframe-major
            const uint32_t ex = wave_incl_scan_dpp(mine) - mine;
#pragma unroll
            for (int sf = 0; sf < 8; sf++) rowbase[sf] = (uint32_t)__builtin_amdgcn_readlane((int)ex, sf * (int)C + (int)c);
        } else {
            uint32_t acc = 0;
#pragma unroll
            for (int sf = 0; sf < 8; sf++) { rowbase[sf] = acc; acc += (uint32_t)__builtin_amdgcn_readlane((int)incl[sf], 63) & 0xFFFF; }
        }
        if (status == 0) {
            const uint32_t start = (uint32_t)hbtot - 16;       // sync + header + every channel's scalefactor part
#pragma unroll
            for (int sf = 0; sf < 8; sf++) {
                const uint32_t tl = incl[sf] >> 16;
                // BitWriter drops a write that does not fit (IO.cpp:131-134).  The rate loop's count can come out a few bits short of what the
                // pack writes (see the checksum below): a code the reference drops then starts inside the frame's last 16 bits, which the
                // checksum replaces, so writing it is harmless -- but a write that STARTS past the frame's end is skipped, whatever the
                // shortfall: the image has two spare words behind it, and a pair of codes is at most 26 bits, so nothing leaves `words`.
                const uint32_t at = start + rowbase[sf] + ((incl[sf] & 0xFFFF) - tl);
                if (at < F.frame_size * 8) put_bits(words, at, both[sf], tl);
            }
        }
        ENC_MARK(9);
        if (XCH) __syncthreads(); else wave_lds_sync();
        ENC_MARK(10);
        if (c == 0 && status == 0) {

            // ---- CRC16 over frame_size-2 bytes (hca.cpp:2961-2962): a chunk of whole words per lane, then one multiply per lane and an xor
            // across the wave.  The message is front-padded with zero bytes to 64 * crc_chunk bytes (leading zeros do not change a zero-init
            // CRC).  A chunk's remainder is taken 32 bits a step in the two factors of P = (x + 1)(x^15 + x + 1) -- the parity of its words,
            // and W ^ R << 4 ^ R << 2 with one fold (the decoder's intake, cri_hca_dec.hip) -- and put together again: the one polynomial
            // below x^16 with that remainder modulo x^15 + x + 1 and that parity is q, or q + (x^15 + x + 1).
            {
                const uint32_t n = F.frame_size - 2, mw = a.crc_chunk >> 2, pad = 64 * a.crc_chunk - n;
                const uint4 cw0 = ((const uint4*)(a.crc_mul + lane * 16))[0], cw1 = ((const uint4*)(a.crc_mul + lane * 16))[1];
                const uint32_t sh = ((0u - pad) & 3) * 8;          // the message's words against the image's (wave-uniform)
                uint32_t rq = 0, par = 0;
                for (uint32_t k = 0; k < mw; k++) {
                    const int q = (int)((lane * mw + k) * 4) - (int)pad;                   // first byte of the word in the message
                    const int wi = q >> 2;                                                 // (floor: -1 for the word that straddles the message's start)
                    const uint32_t hi = wi >= 0 ? words[wi] : 0u, lo = q > -4 ? words[wi + 1] : 0u;
                    const uint32_t w = sh ? __builtin_amdgcn_alignbit(hi, lo, 32 - sh) : hi;
                    rq = (w ^ (rq << 4) ^ (rq << 2));
                    { const uint32_t h = rq >> 15; rq = (rq & 0x7FFFu) ^ h ^ (h << 1); }
                    par ^= w;
                }
                { const uint32_t h = rq >> 15; rq = (rq & 0x7FFFu) ^ h ^ (h << 1); }      // below x^15 now
                uint32_t crc = rq ^ (((__builtin_popcount(rq) ^ __builtin_popcount(par)) & 1) ? 0x8003u : 0u);
                // lane l's chunk stands 8 * crc_chunk * (63 - l) bits above the end of the message, and the checksum is the remainder of the
                // message times x^16: multiply by x^that (mod P) -- the launch's table holds x^bit * x^(16 + 8 * crc_chunk * (63 - l)) for the
                // 16 bits of the chunk's remainder, a 32-byte row per lane -- and xor the 64 products together
                {
                    const uint32_t wr[8] = {cw0.x, cw0.y, cw0.z, cw0.w, cw1.x, cw1.y, cw1.z, cw1.w};
                    uint32_t acc = 0;
    #pragma unroll
                    for (uint32_t bit = 0; bit < 16; bit++) acc ^= (0u - ((crc >> bit) & 1u)) & (wr[bit >> 1] >> (16 * (bit & 1)));
                    acc &= 0xFFFFu;
                    acc ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0x111, 0xF, 0xF, true);      // the scan's pattern, with xor
                    acc ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0x112, 0xF, 0xF, true);
                    acc ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0x114, 0xF, 0xF, true);
                    acc ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0x118, 0xF, 0xF, true);
                    acc ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0x142, 0xA, 0xF, false);
                    acc ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0x143, 0xC, 0xF, false);
                    crc = (uint32_t)__builtin_amdgcn_readlane((int)acc, 63);
                }
                // The checksum REPLACES the frame's last two bytes (hca.cpp:2961-2962).  They are not always zero here: the bit writer's buffer
                // reaches to the frame's last byte (hca.cpp:2941) and the rate loop's count can come out short of what the pack writes
                // (hca.cpp:2771-2786 against 2920-2938), so a full frame's last code may end inside them -- cleared, then or-ed.
                if (lane == 0) {
                    const uint32_t p = (F.frame_size - 2) * 8;
                    const uint64_t m = (uint64_t)0xFFFFu << ((64u - (p & 31) - 16u) & 63);
                    words[p >> 5] &= ~(uint32_t)(m >> 32);
                    words[(p >> 5) + 1] &= ~(uint32_t)m;
                    put_bits(words, p, crc & 0xFFFF, 16);
                }
            }
            wave_lds_sync();
            // the frame image is big-endian words; whole dwords go out byte-swapped (unaligned dword stores), then the last bytes
            for (uint32_t i = lane; 4 * i + 4 <= F.frame_size; i += 64) { const uint32_t w = __builtin_bswap32(words[i]); __builtin_memcpy(dst + 4 * i, &w, 4); }
            if (lane < (F.frame_size & 3)) { const uint32_t i = (F.frame_size & ~3u) + lane; dst[i] = (uint8_t)(words[i >> 2] >> (24 - 8 * (i & 3))); }
            ENC_MARK(11);
        }
        ENC_PROF_FLUSH();
        __syncthreads();                                   // the frame's LDS is the next group's
    }
}

// LDS of one frame: exchange words, frame image, a region per channel
size_t hca_encode_lds_per_frame(uint32_t C, uint32_t frame_size) {
    const size_t nwords = (frame_size + 3) / 4 + 2;
    return ENC_X_BYTES(C) + ((nwords * 4 + 15) & ~(size_t)15) + (size_t)C * ENC_CH_BYTES;
}
// frames per workgroup
uint32_t hca_encode_frames_per_group(uint32_t C, uint32_t frame_size) {
    // (formats with intensity stereo ran one frame per workgroup while the tables were copied per workgroup and five waves shared a SIMD; with
    //  persistent workgroups at six waves per SIMD two frames per workgroup are what the LDS lets in: Middle 135 -> 151, Low 122 -> 133 M frames/s)
    uint32_t fpg = C >= ENC_MAX_WAVES ? 1 : ENC_MAX_WAVES / C;
    while (fpg > 1 && HCA_ET_LDS_BYTES + 16 + fpg * hca_encode_lds_per_frame(C, frame_size) > 160 * 1024) fpg--;
    return fpg;
}
size_t hca_encode_lds_bytes(uint32_t C, uint32_t frame_size) {
    return HCA_ET_LDS_BYTES + 16 + hca_encode_frames_per_group(C, frame_size) * hca_encode_lds_per_frame(C, frame_size);
}

void launch_hca_encode(const HcaEncArgs& a, hipStream_t s) {
    if (!a.frames || a.channels < 1 || a.channels > 8) return;
    HcaEncArgs b = a;
    b.lds_per_frame = (uint32_t)hca_encode_lds_per_frame(a.channels, a.frame_size);
    b.frames_per_group = hca_encode_frames_per_group(a.channels, a.frame_size);
    const size_t lds = HCA_ET_LDS_BYTES + 16 + b.frames_per_group * hca_encode_lds_per_frame(a.channels, a.frame_size);
    if (lds > 160 * 1024) return;
    b.groups = (a.frames + b.frames_per_group - 1) / b.frames_per_group;
    const uint32_t threads = 64 * a.channels * b.frames_per_group;
    // persistent workgroups: eight times what the chip holds at once (by wave slots and LDS), each walking its share of the groups -- the
    // set-up is still shared by hundreds of frames, and a workgroup that starts late (an estimate that is off, a slow compute unit) costs an
    // eighth of a share, not a whole one
    static std::atomic<int> cus_cached[64];                // per device (a box may hold devices of different sizes); zero-initialised
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    int cus = dev >= 0 && dev < 64 ? cus_cached[dev].load() : 0;
    if (cus <= 0) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) { (void)hipGetLastError(); cus = 256; }
        if (dev >= 0 && dev < 64) cus_cached[dev].store(cus);
    }
    const uint32_t by_waves = ENC_WAVES_PER_SIMD_OF(a.channels) * 4u * 64u / threads, by_lds = (uint32_t)((160u * 1024u) / ((lds + 511) & ~(size_t)511));
    const uint32_t per_cu = std::max(1u, std::min(by_waves, by_lds));
    const uint32_t resident = (uint32_t)cus * per_cu * 8u;
    const dim3 grid(b.groups < resident ? b.groups : resident), block(threads);
    switch (a.channels) {
        case 1: hipLaunchKernelGGL(k_hca_encode<1>, grid, block, lds, s, b); break;
        case 2: hipLaunchKernelGGL(k_hca_encode<2>, grid, block, lds, s, b); break;
        case 3: hipLaunchKernelGGL(k_hca_encode<3>, grid, block, lds, s, b); break;
        case 4: hipLaunchKernelGGL(k_hca_encode<4>, grid, block, lds, s, b); break;
        case 5: hipLaunchKernelGGL(k_hca_encode<5>, grid, block, lds, s, b); break;
        case 6: hipLaunchKernelGGL(k_hca_encode<6>, grid, block, lds, s, b); break;
        case 7: hipLaunchKernelGGL(k_hca_encode<7>, grid, block, lds, s, b); break;
        default: hipLaunchKernelGGL(k_hca_encode<8>, grid, block, lds, s, b); break;
    }
}

}  // namespace cri

#ifdef CRI_ENC_PROFILE
extern "C" int cri_debug_enc_profile(unsigned long long* out16, int reset) {
    static unsigned long long h[1024][16];
    if (out16) {
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(cri::g_enc_prof), sizeof h) != hipSuccess) return -1;
        for (int k = 0; k < 16; k++) { out16[k] = 0; for (int s = 0; s < 1024; s++) out16[k] += h[s][k]; }
    }
    if (reset) { memset(h, 0, sizeof h); if (hipMemcpyToSymbol(HIP_SYMBOL(cri::g_enc_prof), h, sizeof h) != hipSuccess) return -1; }
    return 0;
}
#endif
