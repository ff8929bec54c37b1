// cri_hca_enc.hip -- HCA encoder kernel for gfx950 (MI355X, wave64): one wave per frame, all channels.
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
// Up to four frames (waves) share a workgroup and its LDS copies of the tables; the MDCT runs in registers (packed fp32,
// DPP exchanges), the rate loop on register-resident bands.
//
// Everything the reference evaluates in floating point is evaluated here with the same single IEEE operations in the
// same order (sequential sums stay sequential, on one lane); bit allocation is integer work reduced across the wave;
// the bitstream is assembled with a wave prefix sum over code lengths and LDS atomic ORs.
#include <hip/hip_runtime.h>
#include <string.h>
#include "cri_kernels.h"
#include "cri_device.h"
#include "../../include/cricodecs_hip.h"

#define CRI_TABLE_QUAL static __device__ const
#include "cri_tables.h"

namespace cri {

// Developer instrumentation (-DCRI_ENC_PROFILE through CRI_HIPCC_EXTRA): cycles per phase of k_hca_encode, summed over frames
#ifdef CRI_ENC_PROFILE
__device__ unsigned long long g_enc_prof[1024][24];      // spread over 1024 slots so the atomics do not serialise on one address
#define ENC_MARK(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); prof_acc[k] += t_ - prof_t; prof_t = t_; } while (0)
#define ENC_PROF_FLUSH() do { if (lane == 0) for (int k_ = 0; k_ < 24; k_++) atomicAdd(&g_enc_prof[g & 1023][k_], prof_acc[k_]); } while (0)
#else
#define ENC_MARK(k) do {} while (0)
#define ENC_PROF_FLUSH() do {} while (0)
#endif
#ifdef CRI_ENC_PROFILE
#define ENC_COUNT(k, n) do { prof_acc[k] += (n); } while (0)
#define ENC_TIC() const unsigned long long tic_ = __builtin_readcyclecounter()
#define ENC_TOC(k) do { prof_acc[k] += __builtin_readcyclecounter() - tic_; } while (0)
#else
#define ENC_COUNT(k, n) do {} while (0)
#define ENC_TIC() do {} while (0)
#define ENC_TOC(k) do {} while (0)
#endif

__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t v);
__device__ __forceinline__ int wave_sum(int v) {          // total of the 64 lanes, wave-uniform (SGPR)
    return __builtin_amdgcn_readlane((int)wave_incl_scan_dpp((uint32_t)v), 63);
}
__device__ __forceinline__ int wave_excl_scan(int v, uint32_t) { return (int)wave_incl_scan_dpp((uint32_t)v) - v; }
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
__device__ __forceinline__ uint32_t crc16_step_enc(uint32_t crc, uint32_t b) {
    uint32_t t = (crc >> 8) ^ b;
    uint32_t tt = (t << 1) ^ (t << 2) ^ ((__builtin_popcount(t) & 1) ? 0x8003u : 0u);
    return ((crc << 8) & 0xFFFF) ^ tt;
}
// Workgroup-shared LDS copies of every table the per-frame code indexes with data (global-memory lookups were the
// encoder's bottleneck: ~940 dependent loads per frame)
struct EncTab {
    const float *win, *esin, *ecos, *deq, *escale, *dead, *inv, *ibounds;   // [128] [8][64][2] = etw (esin..ecos) [64] [64] [16] [16] [16]
    const f2* etw;                                                          // [8][64] {cos, sin}
    const uint16_t* crcmul;                                                 // (unused slot)
    const uint8_t* sfbase;                                                  // [32] entries of deq[0..62] that are <= 2^(j - 25)
    const uint8_t *curve, *clen, *code, *ishuf;                             // [64] [8][16] [8][16] [128] (ishuf[HCA_ENC_SHUFFLE[k]] = k)
    const uint32_t* bnd;                                                    // [16] per resolution: fewest | most << 16 bits one spectrum can take
    const uint32_t* gb;                                                     // [64] bnd[curve[position]]: the bounds straight from the curve position
    const uint8_t* runend;                                                  // [16] last curve position whose resolution is 15 - r (the curve falls from 15 to 1)
    const int16_t* rowoff;                                                  // [16] resolution r < 8: r * 16 - shiftDown(r), so that clen[rowoff + (int)t] is the code length
                                                                            // of the quantised spectrum t (hca.cpp:2778-2784 without its double arithmetic per band)
};
#define ENC_TAB_BYTES (512 + 2048 + 2048 + 288 + 256 + 64 + 64 + 64 + 64 + 128 + 128 + 128 + 192 + 32 + 64 + 256 + 32 + 16)
__device__ __forceinline__ EncTab enc_tables_to_lds(uint8_t* base, uint32_t tid, uint32_t nthreads, const uint16_t* crc_mul) {
    float* win = (float*)base; float* esin = win + 128; float* ecos = esin + 512; float* deq = ecos + 512; float* escale = deq + 72;
    float* dead = escale + 64; float* inv = dead + 16; float* ib = inv + 16;
    uint8_t* curve = (uint8_t*)(ib + 16); uint8_t* clen = curve + 64; uint8_t* code = clen + 128; uint8_t* shuf = code + 128;
    uint16_t* cm = (uint16_t*)(shuf + 128);
    uint8_t* sfb = (uint8_t*)(cm + 96);
    uint32_t* bnd = (uint32_t*)(sfb + 32);
    uint32_t* gbt = bnd + 16;
    int16_t* rowoff = (int16_t*)(gbt + 64);
    uint8_t* runend = (uint8_t*)(rowoff + 16);
    (void)crc_mul;
    if (tid < 16) {                                        // resolutions 15 .. 1 are runs of the curve's 59 positions
        uint32_t last = 0;
        for (uint32_t i = 0; i < 59; i++) if (HCA_ENC_CURVE_TO_RES[i] == 15 - tid) last = i;
        runend[tid] = (uint8_t)last;
    }                                                          // (read by the checksum step itself, a row per lane)
    for (uint32_t i = tid; i < 128; i += nthreads) { win[i] = HCA_WINDOW[i]; shuf[HCA_ENC_SHUFFLE[i]] = (uint8_t)i; clen[i] = HCA_ENC_CODE_LEN[i >> 4][i & 15]; code[i] = HCA_ENC_CODE[i >> 4][i & 15]; }
    for (uint32_t i = tid; i < 512; i += nthreads) { esin[2 * i] = HCA_ENC_COS[i >> 6][i & 63]; esin[2 * i + 1] = HCA_ENC_SIN[i >> 6][i & 63]; }   // etw[i] = {cos, sin}: a twiddle is one register pair
    for (uint32_t i = tid; i < 64; i += nthreads) { escale[i] = HCA_ENC_SCALE[i]; curve[i] = i < 59 ? HCA_ENC_CURVE_TO_RES[i] : 0; }
    for (uint32_t i = tid; i < 72; i += nthreads) deq[i] = i < 63 ? HCA_DEQ_SCALE[i] : __uint_as_float(0x7FC00000u);   // NaN padding never compares <=
    for (uint32_t j = tid; j < 32; j += nthreads) {
        const float thr = __uint_as_float((j + 102u) << 23);   // 2^(j - 25)
        uint32_t n = 0;
        for (uint32_t k = 0; k < 63; k++) n += HCA_DEQ_SCALE[k] <= thr ? 1u : 0u;
        sfb[j] = (uint8_t)n;
    }
    for (uint32_t i = tid; i < 16; i += nthreads) {
        dead[i] = HCA_ENC_DEAD_ZONE[i]; inv[i] = HCA_ENC_INV_STEP[i]; ib[i] = i < 14 ? HCA_ENC_INTENSITY_BOUNDS[i] : 0.0f;
        uint32_t lo, hi;                                   // code lengths of resolution i (hca.cpp:2771-2786)
        if (i >= 8) { hi = i - 3; lo = hi - 1; }           // sign-magnitude: max bits, one less for a zero
        else { lo = i ? 15 : 0; hi = 0; for (uint32_t q = 8 - i; q <= 8 + i; q++) { const uint32_t l = HCA_ENC_CODE_LEN[i][q]; lo = l < lo ? l : lo; hi = l > hi ? l : hi; } }
        bnd[i] = lo | (hi << 16);                          // per spectrum; a band has 8
        rowoff[i] = (int16_t)((int)i * 16 - (int)((double)HCA_ENC_INV_STEP[i] + 0.5 - 8));
    }
    for (uint32_t i = tid; i < 64; i += nthreads) {        // (the same arithmetic as above, per curve position)
        const uint32_t r = i < 59 ? HCA_ENC_CURVE_TO_RES[i] : 0;
        uint32_t lo, hi;
        if (r >= 8) { hi = r - 3; lo = hi - 1; }
        else { lo = r ? 15 : 0; hi = 0; for (uint32_t q = 8 - r; q <= 8 + r; q++) { const uint32_t l = HCA_ENC_CODE_LEN[r][q]; lo = l < lo ? l : lo; hi = l > hi ? l : hi; } }
        gbt[i] = lo | (hi << 16);
    }
    EncTab T; T.win = win; T.esin = esin; T.ecos = ecos; T.etw = (const f2*)esin; T.deq = deq; T.escale = escale; T.dead = dead; T.inv = inv; T.ibounds = ib;
    T.curve = curve; T.clen = clen; T.code = code; T.ishuf = shuf; T.crcmul = cm; T.sfbase = sfb; T.bnd = bnd; T.gb = gbt; T.rowoff = rowoff; T.runend = runend;
    return T;
}

// hca.cpp:2611-2623: the binary search over the ascending table returns the number of entries 0..62 that are <= v.  The
// table has 128/53 = 2.4 entries per octave, so that count is sfbase[exponent of v] (entries <= 2^exponent, built exactly
// at table-load time) plus at most three more compares -- two dependent LDS reads instead of six.
__device__ __forceinline__ int enc_find_scalefactor(const EncTab& T, float v) {
    int eb = (int)(__float_as_uint(v) >> 23) & 0xFF;
    eb = eb < 102 ? 102 : (eb > 133 ? 133 : eb);            // the table spans 2^-23 .. 2^3.5
    const int base = T.sfbase[eb - 102];
    const float e0 = T.deq[base], e1 = T.deq[base + 1], e2 = T.deq[base + 2];   // deq[] is padded past 63 with +inf
    return base + (e0 <= v ? 1 : 0) + (e1 <= v ? 1 : 0) + (e2 <= v ? 1 : 0);
}
__device__ __forceinline__ int enc_resolution(const EncTab& T, int sf, int noise) {   // hca.cpp:2752-2761
    int cp = noise - 5 * sf / 2 + 2;
    cp = cp < 0 ? 0 : (cp > 58 ? 58 : cp);
    const int r = T.curve[cp];
    return sf == 0 ? 0 : r;
}
__device__ __forceinline__ int enc_maxbits(int res) { return res > 7 ? res - 3 : (int)((0x44443320u >> (res * 4)) & 15); }

// (u.x*c + u.y*s, u.x*s - u.y*c): the rotation of hca.cpp:2515-2520 / 2493-2496 as three packed operations.  The twiddle is one
// register pair tw = {c, s}, and the operand selects of the packed multiplies pick u.x / u.y and c / s -- written out as
// instructions because the compiler builds the broadcast operands with moves first (68 of a pass's 274 instructions)
__device__ __forceinline__ f2 enc_rot(f2 u, f2 tw) {
    f2 p, q;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(p) : "v"(u), "v"(tw));      // {u.x*c, u.x*s}
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(q) : "v"(u), "v"(tw));      // {u.y*s, u.y*c}
    return pk_add_neg_hi(p, q);
}

struct EncLds {
    float* sp;        // [C][8][128] spectra
    float* sc;        // scaled spectra: the same buffer, scaled in place once the unscaled values are no longer needed
    float* tin;       // 1280 B: PCM staging of one MDCT pass (int16[640]); later the frame image `words` (same memory)
    uint32_t* words;  // frame as big-endian 32-bit words
    uint8_t* sfac;    // [C][128]
    uint8_t* res;     // [C][128]
    uint8_t* inten;   // [C][8]
    int* hfrs;        // [C][8] HFR scales
    float* havg;      // [C][8]
    float* ratio;     // [8]
    int* hbits;       // [C] header bits
    int* dbits;       // [C] delta bits
};

struct EncFmt {
    uint32_t C, frame_size, total, base, stereo, groups, bpg, hfr_band_count, types;
    __device__ __forceinline__ uint32_t type(uint32_t c) const { return (types >> (2 * c)) & 3u; }
    __device__ __forceinline__ uint32_t coded(uint32_t c) const { return type(c) == CRI_CH_SECONDARY ? base : base + stereo; }
};

// CalculateFrameHeaderLength, hca.cpp:2708-2750
__device__ __forceinline__ void enc_header_length(const EncFmt& F, const EncLds& L, uint32_t lane) {
    for (uint32_t c = 0; c < F.C; c++) {
        const int coded = (int)F.coded(c);
        const uint8_t* sf = L.sfac + c * 128;
        // a delta width db codes |delta| <= 2^(db-1) - 1 in db bits and the rest in db + 6: the length is
        // db * (coded - 1) + 6 * (deltas above the limit), so one pass counts the deltas above 0, 1, 3, 7, 15 (a byte each)
        uint32_t w0 = 0, w1 = 0;
        for (int b = (int)lane; b < coded; b += 64) {
            const int cur = sf[b];
            w1 |= cur != 0 ? 0x100u : 0u;                  // any scalefactor at all
            if (b >= 1) {
                int d = cur - (int)sf[b - 1]; d = d < 0 ? -d : d;
                w0 += (d > 0 ? 1u : 0u) | (d > 1 ? 0x100u : 0u) | (d > 3 ? 0x10000u : 0u) | (d > 7 ? 0x1000000u : 0u);
                w1 += d > 15 ? 1u : 0u;
            }
        }
        w0 = (uint32_t)wave_sum((int)w0);                  // every count is at most 127
        w1 = (uint32_t)wave_sum((int)w1);
        const int any = (int)(w1 >> 8);
        int min_len = 3, min_db = 0;
        if (any) {
            min_db = 6; min_len = 3 + 6 * coded;
            const int above[5] = {(int)(w0 & 0xFF), (int)((w0 >> 8) & 0xFF), (int)((w0 >> 16) & 0xFF), (int)(w0 >> 24), (int)(w1 & 0xFF)};
            for (int db = 1; db < 6; db++) {
                const int length = 3 + 6 + db * (coded > 0 ? coded - 1 : 0) + 6 * above[db - 1];
                if (length < min_len) { min_len = length; min_db = db; }
            }
        }
        if (F.type(c) == CRI_CH_SECONDARY) min_len += 32;
        else if (F.groups > 0) min_len += 6 * (int)F.groups;
        if (lane == 0) { L.hbits[c] = min_len; L.dbits[c] = min_db; }
    }
    wave_lds_sync();
}

// bits of the 8 spectra of one band at one resolution (the inner part of CalculateUsedBits, hca.cpp:2771-2786)
// (the spectra as four pairs: the multiply and the add of the quantiser are packed operations, the same two roundings per value)
__device__ __forceinline__ int enc_band_bits(const EncTab& T, const f2 x[4], int res) {
    int part = 0;
    if (res >= 8) {
        const int bits = enc_maxbits(res) - 1;
        const float dz = T.dead[res];
        part = 8 * bits;
#pragma unroll
        for (int j = 0; j < 4; j++) part += (fabsf(x[j].x) >= dz ? 1 : 0) + (fabsf(x[j].y) >= dz ? 1 : 0);
    } else {
        // (|x| < 1 by ScaleSpectra's clamp, so (int)t - shiftDown is 0 .. 2 * res: no mask; the row offset carries the shift)
        const float inv = T.inv[res], up = inv + 1;
        const uint8_t* row = T.clen + T.rowoff[res];
        const f2 invv = {inv, inv}, upv = {up, up};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const f2 t = x[j] * invv + upv;
            part += row[(int)t.x] + row[(int)t.y];
        }
    }
    return part;
}

// CalculateUsedBits, hca.cpp:2763-2790 (integer; reduced across the wave)
__device__ __forceinline__ int enc_used_bits(const EncFmt& F, const EncLds& L, const EncTab& T, uint32_t lane, int noise_level, int eval_boundary) {
    int part = 0;
    for (uint32_t c = 0; c < F.C; c++) {
        const int coded = (int)F.coded(c);
        for (int i = (int)lane; i < coded; i += 64) {
            const int noise = i < eval_boundary ? noise_level - 1 : noise_level;
            const int res = enc_resolution(T, L.sfac[c * 128 + i], noise);
            const float* x = L.sc + (c * 8) * 128 + i;
            if (res >= 8) {
                const int bits = enc_maxbits(res) - 1;
                const float dz = T.dead[res];
                for (int j = 0; j < 8; j++) { part += bits; if (fabsf(x[j * 128]) >= dz) part++; }
            } else {
                const float inv = T.inv[res], up = inv + 1;
                const int down = (int)((double)inv + 0.5 - 8);
                for (int j = 0; j < 8; j++) { const int q = (int)(x[j * 128] * inv + up) - down; part += T.clen[res * 16 + (q & 15)]; }
            }
        }
    }
    int length = 16 + 16 + 16 + wave_sum(part);
    for (uint32_t c = 0; c < F.C; c++) length += L.hbits[c];
    return length;
}

// MSB-first write of `len` bits of v at absolute bit position p of the frame (words are big-endian 32-bit)
__device__ __forceinline__ void put_bits(uint32_t* words, uint32_t p, uint32_t v, uint32_t len) {
    if (!len) return;
    v &= len >= 32 ? 0xFFFFFFFFu : ((1u << len) - 1);
    const uint32_t w = p >> 5;
    const int shift = 32 - (int)(p & 31) - (int)len;
    if (shift >= 0) atomicOr(&words[w], v << shift);
    else { atomicOr(&words[w], v >> (-shift)); atomicOr(&words[w + 1], v << (32 + shift)); }
}

#ifndef ENC_WAVES
#define ENC_WAVES 2     // most frames (waves) per workgroup; they share the LDS tables and are otherwise independent (measured, stereo
                        // High, ms per 469 k frames: 1 wave -, 2: 6.68, 3: 6.72-6.80, 4: 6.91-6.96, 7: 9.4; at 4 waves per SIMD it spills: 7.5)
#endif
#ifndef ENC_MIN_WAVES
#define ENC_MIN_WAVES 1
#endif
// CT = 1, 2, 4, 6, 8: channel count known at compile time, the rate loop keeps the lane's bands in registers; CT = 0: any count
template <int CT>
__global__ __launch_bounds__(64 * ENC_WAVES, ENC_MIN_WAVES) void k_hca_encode(HcaEncArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_all[];
    const EncTab T = enc_tables_to_lds(smem_all, threadIdx.x, blockDim.x, a.crc_mul);
    __syncthreads();                                       // the only workgroup barrier: tables are read-only from here on
    uint8_t* smem = smem_all + ENC_TAB_BYTES + (threadIdx.x >> 6) * a.lds_per_wave;
    const HcaFormat* Fp = a.formats + a.format;
    EncFmt F;
    F.C = Fp->channels; F.frame_size = Fp->frame_size; F.total = Fp->total_bands; F.base = Fp->base_bands; F.stereo = Fp->stereo_bands;
    F.groups = Fp->hfr_group_count; F.bpg = Fp->bands_per_hfr_group; F.hfr_band_count = Fp->hfr_band_count;
    { uint32_t t = 0; for (uint32_t c = 0; c < 16; c++) t |= (uint32_t)(Fp->type[c] & 3) << (2 * c); F.types = t; }
    const uint32_t C = CT ? (uint32_t)CT : F.C, lane = threadIdx.x & 63, g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (g >= a.frames) return;
    const uint32_t nwords = (F.frame_size + 3) / 4 + 1;
    EncLds L;
    // the MDCT's PCM staging buffer (tin) and the output frame image (words) are never live together
    L.sp = (float*)smem; L.sc = L.sp; L.tin = L.sp + C * 1024;
    L.words = (uint32_t*)L.tin; L.havg = (float*)(L.words + (nwords > 320 ? nwords : 320)); L.ratio = L.havg + C * 8;
    L.hfrs = (int*)(L.ratio + 8); L.hbits = L.hfrs + C * 8; L.dbits = L.hbits + C;
    L.sfac = (uint8_t*)(L.dbits + C); L.res = L.sfac + C * 128; L.inten = L.res + C * 128;

    // frame -> stream
    uint32_t lo = a.stream_begin, hi = a.stream_end;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (a.streams[mid].first_frame <= g) lo = mid; else hi = mid; }
    const HcaStream st = a.streams[lo];
    const uint32_t f = g - st.first_frame;
    const uint8_t* pcm = (st.src_in_scratch ? a.scratch : a.in) + st.src_offset;
    // PcmToFloat (hca.cpp:2470-2479) of the sample at position rel (-128 .. 1023) relative to the frame's first sample,
    // zero outside the stream.  Branch-free so that a lane's 16 loads of a pass issue back to back: the address is
    // clamped into the readable range and the value selected afterwards.
    const uint64_t F0 = (uint64_t)f * 1024;
    const int64_t nsamp = (int64_t)st.samples;
    // plain streams: readable rel range [rlo, rhi) of this frame and the address of rel = rlo
    const int rlo = F0 >= 128 ? -128 : -(int)F0;
    const int64_t hi64 = nsamp - (int64_t)F0;
    const int rhi = hi64 > 1024 ? 1024 : (hi64 < -128 ? -128 : (int)hi64);
    const uint8_t* fbase = pcm + (F0 + (int64_t)rlo) * C * 2;
    const bool any_plain = rhi > rlo;
    const bool have_any = st.enc_have > 0;
    // (the two stream kinds are separate straight-line code so that nothing but loads sits between the loads)
    auto sample_plain = [&](int rel, uint32_t c) -> float {
        const int rc = rel < rlo ? rlo : (rel > rhi - 1 ? rhi - 1 : rel);
        int16_t v; __builtin_memcpy(&v, fbase + ((uint32_t)(rc - rlo) * C + c) * 2, 2);
        const float x = (float)(int)v * (float)(1.0f / 32768.0f);
        return rc == rel ? x : 0.0f;
    };
    auto sample_loop = [&](int rel, uint32_t c) -> float {   // the feeding sequence of hca.cpp:2990-3107 (see cri_types.h)
        const int64_t n = (int64_t)F0 + rel, m = n - (int64_t)st.enc_pre, e = m - nsamp;
        const bool in_pre = m < 0, in_main = !in_pre && m < nsamp, in_post = !in_pre && !in_main && e < (int64_t)st.enc_post;
        int64_t src = in_pre ? 0 : (in_main ? m : (int64_t)st.enc_loop_src + e);
        const bool ok = n >= (int64_t)st.enc_pre_zero && (in_pre || in_main || (in_post && src < (int64_t)st.enc_loop_src_end)) && src < (int64_t)st.enc_have;
        src = ok ? src : 0;
        int16_t v; __builtin_memcpy(&v, pcm + ((uint64_t)src * C + c) * 2, 2);
        const float x = (float)(int)v * (float)(1.0f / 32768.0f);
        return ok ? x : 0.0f;
    };

#ifdef CRI_ENC_PROFILE
    unsigned long long prof_acc[24] = {0}; unsigned long long prof_t = __builtin_readcyclecounter();
#endif
    // ---- MDCT of every (channel, subframe): hca.cpp:2529-2553 (window + fold), 2481-2527 (DCT-IV), in registers.
    // Four transforms at a time: slot = lane >> 4 picks the transform, its 16 lanes hold the 64 complex points of the
    // reference's in-place radix-2 network, point j = 4 * lane16 + r in f2 z[r].  Stages on bits 5..2 of j exchange with
    // lane16 ^ 8, 4, 2, 1 (DPP), stages on bits 1, 0 pair registers.  The lane that holds the lower point of a pair keeps
    // the sum, the other one rotates the difference; the sum/difference is fma(z, +-1, partner) (exact product).
    {
        const uint32_t l16 = lane & 15, slot = lane >> 4;
        // window coefficients and sample positions (within the 256 samples [n0-128, n0+128)) of the lane's 8 folded inputs
        float wA[8], wB[8]; int mA[8], mB[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int r = q & 3, odd = q >> 2;
            const int k = odd ? 127 - 8 * (int)l16 - 2 * r : 8 * (int)l16 + 2 * r;
            const bool low = k < 64;
            mA[q] = low ? 192 + k : k - 64; mB[q] = 191 - k;
            const float a = T.win[low ? 63 - k : k - 64];
            wA[q] = low ? -a : a;                             // hca.cpp:2532: window * -sample
            wB[q] = T.win[low ? 64 + k : 191 - k];
        }
        const float sg8 = l16 & 8 ? -1.0f : 1.0f, sg4 = l16 & 4 ? -1.0f : 1.0f, sg2 = l16 & 2 ? -1.0f : 1.0f, sg1 = l16 & 1 ? -1.0f : 1.0f;
        uint32_t opos[8];                                      // where the lane's 8 outputs go in the spectrum (inverse of the final shuffle)
#pragma unroll
        for (int q = 0; q < 8; q++) opos[q] = T.ishuf[8 * l16 + q];
        // PCM of a pass (one channel, 4 subframes + the 128 samples before them = 640 samples) goes through LDS: coalesced
        // dword loads, every cache line requested once (lane-scattered 2-byte loads asked for each line 8 times and made
        // this phase L2-request bound).  The next pass's dwords are in flight during the current pass.
        int16_t* stg = (int16_t*)L.tin;                        // [640]
        uint32_t dw[10];
        bool fast_next = false;
        auto pass_fast = [&](uint32_t pass) -> bool {          // mono / stereo, no loop mapping, window entirely inside the stream
            const int first = (int)((pass * 4) & 7) * 128 - 128;
            return !st.enc_loop && (C == 1 || C == 2) && first >= rlo && first + 640 <= rhi;
        };
        auto issue = [&](uint32_t pass) {
            fast_next = pass_fast(pass);
            if (!fast_next) return;
            const int first = (int)((pass * 4) & 7) * 128 - 128;
            const uint8_t* src = fbase + (uint32_t)(first - rlo) * C * 2;
            const uint32_t nd = 320 * C;                       // dwords of the pass window
#pragma unroll
            for (int j = 0; j < 10; j++) { const uint32_t k = lane + 64 * j; dw[j] = 0; if (k < nd) dw[j] = ld_u32_unaligned(src + 4 * k); }
        };
        auto commit = [&](uint32_t pass) {
            const uint32_t c = (pass * 4) >> 3;
            if (fast_next) {
                if (C == 2) {
#pragma unroll
                    for (int j = 0; j < 10; j++) stg[lane + 64 * j] = (int16_t)(dw[j] >> (16 * c));
                } else {
#pragma unroll
                    for (int j = 0; j < 5; j++) ((uint32_t*)stg)[lane + 64 * j] = dw[j];
                }
            } else {                                           // stream edges, loop streams, more than two channels
                const int first = (int)((pass * 4) & 7) * 128 - 128;
                for (uint32_t i = lane; i < 640; i += 64) {
                    const float x = st.enc_loop ? (have_any ? sample_loop(first + (int)i, c) : 0.0f) : (any_plain ? sample_plain(first + (int)i, c) : 0.0f);
                    stg[i] = (int16_t)(int)(x * 32768.0f);     // exact round trip of the int16 sample
                }
            }
        };
        ENC_MARK(8);
        issue(0);
        ENC_MARK(9);
#pragma unroll 1
        for (uint32_t pass = 0; pass < 2 * C; pass++) {
            const uint32_t tr = pass * 4 + slot, c = tr >> 3, sf = tr & 7;
            wave_lds_sync();
            commit(pass);
            wave_lds_sync();
            ENC_MARK(10);
            if (pass + 1 < 2 * C) issue(pass + 1);
            ENC_MARK(11);
            float in[8];
            {
                const int16_t* sw = stg + slot * 128;
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const float xa = (float)(int)sw[mA[q]] * (float)(1.0f / 32768.0f), xb = (float)(int)sw[mB[q]] * (float)(1.0f / 32768.0f);   // PcmToFloat, hca.cpp:2470-2479
                    const float pa = wA[q] * xa, pb = wB[q] * xb;
                    in[q] = pa + pb;                           // a - b with b = -(w*x)
                }
            }
            f2 z[4];
            {
                const float4 ta = *(const float4*)(T.etw + 7 * 64 + 4 * l16), tb = *(const float4*)(T.etw + 7 * 64 + 4 * l16 + 2);
                const f2 tw[4] = {f2{ta.x, ta.y}, f2{ta.z, ta.w}, f2{tb.x, tb.y}, f2{tb.z, tb.w}};
#pragma unroll
                for (int r = 0; r < 4; r++) z[r] = enc_rot(f2{in[r], in[4 + r]}, tw[r]);
            }
#define ENC_CROSS(X, HB, SG) { \
                const uint32_t ti = HB * 64 + ((l16 & (X - 1)) << 2); \
                const float4 ta = *(const float4*)(T.etw + ti), tb = *(const float4*)(T.etw + ti + 2); \
                const f2 tw[4] = {f2{ta.x, ta.y}, f2{ta.z, ta.w}, f2{tb.x, tb.y}, f2{tb.z, tb.w}}; \
                const bool hi = (l16 & X) != 0; \
                _Pragma("unroll") for (int r = 0; r < 4; r++) { \
                    const f2 u = __builtin_elementwise_fma(z[r], f2{SG, SG}, lane16_xor2<X>(z[r])); \
                    const f2 w = enc_rot(u, tw[r]); \
                    z[r] = f2{hi ? w.x : u.x, hi ? w.y : u.y}; \
                } }
            ENC_CROSS(8, 5, sg8) ENC_CROSS(4, 4, sg4) ENC_CROSS(2, 3, sg2) ENC_CROSS(1, 2, sg1)
#undef ENC_CROSS
            {   // bit 1 of j: (z0, z2) with twiddle [1][0], (z1, z3) with [1][1]
                const float4 t1 = *(const float4*)(T.etw + 64);
                const f2 d0 = z[0] - z[2], d1 = z[1] - z[3];
                z[0] = z[0] + z[2]; z[1] = z[1] + z[3];
                z[2] = enc_rot(d0, f2{t1.x, t1.y}); z[3] = enc_rot(d1, f2{t1.z, t1.w});
            }
            {   // bit 0 of j: (z0, z1), (z2, z3) with twiddle [0][0]
                const f2 t0 = T.etw[0];
                const f2 d0 = z[0] - z[1], d1 = z[2] - z[3];
                z[0] = z[0] + z[1]; z[2] = z[2] + z[3];
                z[1] = enc_rot(d0, t0); z[3] = enc_rot(d1, t0);
            }
            float* out = L.sp + (c * 8 + sf) * 128;
#pragma unroll
            ENC_MARK(12);
#pragma unroll
            for (int r = 0; r < 4; r++) { const f2 o = z[r] * f2{0.125f, 0.125f}; out[opos[2 * r]] = o.x; out[opos[2 * r + 1]] = o.y; }
            ENC_MARK(13);
        }
        wave_lds_sync();
    }

    ENC_MARK(0);
    // ---- EncodeIntensityStereo, hca.cpp:2561-2609 (sequential sums: one lane per subframe)
    if (F.stereo > 0) {
        for (uint32_t c = 0; c + 1 < C; c++) {
            if (F.type(c) != CRI_CH_PRIMARY) continue;
            float* lsp = L.sp + (c * 8) * 128; float* rsp = L.sp + ((c + 1) * 8) * 128;
            if (lane < 8) {
                const float* l = lsp + lane * 128; const float* r = rsp + lane * 128;
                float el = 0, er = 0, et = 0;
                for (uint32_t b = F.base; b < F.total; b++) { el += fabsf(l[b]); er += fabsf(r[b]); et += fabsf(l[b] + r[b]); }
                et *= 2;
                const float elr = er + el;
                const float stored = 2 * el / elr;
                float ratio = elr / et;
                if (ratio < 0.5) ratio = 0.5f;
                else if ((double)ratio > sqrt(2.0) / 2) ratio = (float)(sqrt(2.0) / 2);
                int q = 1;
                if (er > 0 || el > 0) { while (q < 13 && T.ibounds[q] >= stored) q++; }
                else { q = 0; ratio = 1; }
                L.inten[(c + 1) * 8 + lane] = (uint8_t)q;
                L.ratio[lane] = ratio;
            }
            wave_lds_sync();
            for (uint32_t sf = 0; sf < 8; sf++) {
                const float ratio = L.ratio[sf];
                for (uint32_t b = F.base + lane; b < F.total; b += 64) {
                    const float s = lsp[sf * 128 + b] + rsp[sf * 128 + b];
                    lsp[sf * 128 + b] = s * ratio;
                    rsp[sf * 128 + b] = 0;
                }
            }
            wave_lds_sync();
        }
    }

    ENC_MARK(1);
    // ---- CalculateHfrGroupAverages, hca.cpp:2656-2674 (sequential sums: one lane per group).  It reads the unscaled
    //      spectra of the bands above the coded range, so it runs before they are scaled in place.
    const int hfr_start = (int)(F.stereo + F.base);
    if (F.groups > 0) {
        const int bpg = (int)F.bpg;
        for (uint32_t c = 0; c < C; c++) {
            if (F.type(c) == CRI_CH_SECONDARY) continue;
            if (lane < F.groups) {
                const int grp = (int)lane;
                float sum = 0.0f; int count = 0;
                for (int i = 0; i < bpg; i++) {
                    const int band = hfr_start + grp * bpg + i;
                    if (band >= 128) break;
                    for (int sf = 0; sf < 8; sf++) sum += fabsf(L.sp[(c * 8 + sf) * 128 + band]);
                    count += 8;
                }
                L.havg[c * 8 + grp] = sum / (float)count;
            }
        }
        wave_lds_sync();
    }

    // ---- CalculateScaleFactors + ScaleSpectra, hca.cpp:2625-2654 (scaled in place: L.sc == L.sp)
    for (uint32_t c = 0; c < C; c++) {
        const uint32_t coded = F.coded(c);
        for (uint32_t b = lane; b < 128; b += 64) {
            float x[8];
            float mx = 0;
#pragma unroll
            for (int sf = 0; sf < 8; sf++) { x[sf] = L.sp[(c * 8 + sf) * 128 + b]; const float v = fabsf(x[sf]); mx = (v < mx) ? mx : v; }
            uint32_t s = (uint32_t)enc_find_scalefactor(T, mx);
            s = b < coded ? s : 0u;
            L.sfac[c * 128 + b] = (uint8_t)s;
            const float es = T.escale[s];
#pragma unroll
            for (int sf = 0; sf < 8; sf++) {
                float v = x[sf] * es;
                if (v > 0.9999999f) v = 0.9999999f; else if (v < -0.9999999f) v = -0.9999999f;
                if (s == 0) v = 0;                         // also every band past the coded range
                L.sc[(c * 8 + sf) * 128 + b] = v;
            }
        }
    }
    wave_lds_sync();

    // ---- CalculateHfrScale, hca.cpp:2676-2706
    if (F.groups > 0) {
        const int bpg = (int)F.bpg;
        const int hb = (int)(F.hfr_band_count < F.total - F.hfr_band_count ? F.hfr_band_count : F.total - F.hfr_band_count);
        for (uint32_t c = 0; c < C; c++) {
            if (F.type(c) == CRI_CH_SECONDARY) continue;
            if (lane < F.groups) {
                const int grp = (int)lane;
                float sum = 0.0f; int count = 0;
                for (int i = 0; i < bpg; i++) {
                    const int band = grp * bpg + i;
                    if (band >= hb) break;
                    for (int sf = 0; sf < 8; sf++) sum += fabsf(L.sc[(c * 8 + sf) * 128 + (hfr_start - band - 1)]);
                    count += 8;
                }
                const float avg = sum / (float)count;
                float gs = L.havg[c * 8 + grp];
                if (avg > 0.0) {
                    const double m = 1.0 / (double)avg, r2 = sqrt(2.0);
                    gs = (float)((double)gs * (m < r2 ? m : r2));
                }
                L.havg[c * 8 + grp] = gs;
                L.hfrs[c * 8 + grp] = enc_find_scalefactor(T, gs);
            }
        }
        wave_lds_sync();
    }

    ENC_MARK(2);
    // ---- rate loop: CalculateNoiseLevel, CalculateEvaluationBoundary (hca.cpp:2792-2866)
    // With a compile-time channel count the lane's bands (i = lane, lane + 64 of every channel) sit in registers: their 8
    // scaled spectra, scalefactor and "is coded" flag.
    constexpr int NB = CT > 0 ? 2 * CT : 1;
    f2 xr[NB][4]; int sfr[NB]; bool inr[NB];
    int kb[NB]; bool live[NB];                             // curve position of band b at noise level n: n - kb[b]; live: coded and scalefactor != 0
    auto load_bands = [&]() {
        if constexpr (CT > 0) {
#pragma unroll
            for (int b = 0; b < NB; b++) {
                const uint32_t c = b >> 1, i = lane + 64 * (b & 1);
                inr[b] = i < F.coded(c);
                sfr[b] = L.sfac[c * 128 + i];
                kb[b] = 5 * sfr[b] / 2 - 2; live[b] = inr[b] && sfr[b] != 0;
#pragma unroll
                for (int j = 0; j < 4; j++) xr[b][j] = f2{L.sc[(c * 8 + 2 * j) * 128 + i], L.sc[(c * 8 + 2 * j + 1) * 128 + i]};
            }
        }
    };
    auto header_bits = [&]() { int h = 16 + 16 + 16; for (uint32_t c = 0; c < C; c++) h += L.hbits[c]; return h; };
    // The noise-level search ends with noise_level = the last level found to fit and noise_level - 1 = the last one found not to;
    // the boundary search that follows costs every band at exactly those two levels.  The bands' bits of the last exact
    // evaluation of either kind are kept (`last`, per lane and band), so that they are not quantised again.
    int fit_bits[NB], over_bits[NB], last[NB];
    int fit_noise = -1000, over_noise = -1000;             // (wave-uniform) levels the kept bits belong to; -1000: none
    auto used_bits = [&](int noise, int eb) -> int {
        if constexpr (CT > 0) {
            int part = 0;
#pragma unroll
            for (int b = 0; b < NB; b++) {
                const int i = (int)lane + 64 * (b & 1);
                const int bits = enc_band_bits(T, xr[b], enc_resolution(T, sfr[b], i < eb ? noise - 1 : noise));
                last[b] = inr[b] ? bits : 0;
                part += last[b];
            }
            return header_bits() + wave_sum(part);
        } else return enc_used_bits(F, L, T, lane, noise, eb);
    };
    enc_header_length(F, L, lane);
    ENC_MARK(14);
    load_bands();
    ENC_MARK(15);
    const int avail = (int)F.frame_size * 8;
    int noise_level = -1, eval_boundary = 0, status = 0;
    {
        int highest = (int)(F.base + F.stereo) - 1;
        for (;;) {
            int low = 0, high = 255;
            bool over = false;                             // "mid_value > available bits" of the last step (hca.cpp:2806-2815)
            const int hb = header_bits();
            // The first six levels of the search tree at once.  A band sits at curve position n - kb at noise level n, and the curve is
            // 15 runs of equal resolution, so the bands of one resolution at level n are those whose kb lies in a window that slides with
            // n: with H[x] = bands with kb + 2 <= x (a histogram of 5 * scalefactor / 2, prefix-summed), lane l < 63 adds up, for ITS node's
            // level, 15 window counts times that resolution's fewest / most bits.  What the loop below reads at depth < 6 is two ballots.
            uint64_t tree_over = 0, tree_fit = 0;
            if constexpr (CT > 0) {
                uint32_t* Hs = L.words;                    // [193]: Hs[x + 1] = H[x], x = -1 .. 191 (the frame image is not in use yet)
                wave_lds_sync();
                Hs[lane] = 0; Hs[lane + 64] = 0; Hs[lane + 128] = 0; if (lane == 0) Hs[192] = 0;
                wave_lds_sync();
#pragma unroll
                for (int b = 0; b < NB; b++) if (live[b]) atomicAdd(&Hs[kb[b] + 2 + 1], 1u);
                wave_lds_sync();
                {   // prefix sum over the 192 counters: three per lane, then across the lanes
                    const uint32_t c0 = Hs[3 * lane + 1], c1 = Hs[3 * lane + 2], c2 = Hs[3 * lane + 3];
                    const uint32_t incl = wave_incl_scan_dpp(c0 + c1 + c2), before = incl - (c0 + c1 + c2);
                    wave_lds_sync();
                    Hs[3 * lane + 1] = before + c0; Hs[3 * lane + 2] = before + c0 + c1; Hs[3 * lane + 3] = incl;
                }
                wave_lds_sync();
                const uint32_t total = Hs[192];
                const uint32_t node = lane < 63 ? lane : 0, d = 31 - (uint32_t)__clz((int)(node + 1)), jn = node + 1 - (1u << d);
                const int nmid = (int)(jn << (8 - d)) + (int)(1u << (7 - d)) - 1;
                uint32_t tot = 0, prevc = total;
#pragma unroll
                for (int r = 0; r < 15; r++) {             // resolution 15 - r: positions up to runend[r] that the run before did not take
                    uint32_t cur = 0;
                    if (r < 14) { int x = nmid + 1 - (int)T.runend[r]; x = x < -1 ? -1 : (x > 191 ? 191 : x); cur = Hs[x + 1]; }
                    tot += T.bnd[15 - r] * (prevc - cur);
                    prevc = cur;
                }
                const int least = hb + 8 * (int)(tot & 0xFFFF), most = hb + 8 * (int)(tot >> 16);
                tree_over = __ballot(lane < 63 && least > avail);
                tree_fit = __ballot(lane < 63 && most <= avail);
                wave_lds_sync();
            }
            uint32_t tnode = 0, depth = 0;
            while (low != high) {
                const int mid = (low + high) / 2;
                ENC_COUNT(16, 1);
                if constexpr (CT > 0) {
                    // the same decision from per-resolution bounds when they settle it (far from the answer they do): the
                    // fewest / most bits a band of that resolution can take, summed -- no quantisation of the spectra
                    bool settled = false;
                    if (depth < 6) {
                        if ((tree_over >> tnode) & 1) { over = true; settled = true; }
                        else if ((tree_fit >> tnode) & 1) { over = false; settled = true; }
                    } else {
                        uint32_t part = 0;
#pragma unroll
                        for (int b = 0; b < NB; b++) {                         // bnd[resolution] straight from the curve position
                            int cp = mid - kb[b];
                            cp = cp < 0 ? 0 : (cp > 58 ? 58 : cp);
                            part += live[b] ? T.gb[cp] : 0u;
                        }
                        const uint32_t tot = (uint32_t)wave_sum((int)part);                  // at most 2 * CT * 64 * 12 per half: no carry
                        const int least = hb + 8 * (int)(tot & 0xFFFF), most = hb + 8 * (int)(tot >> 16);
                        if (least > avail) { over = true; settled = true; }
                        else if (most <= avail) { over = false; settled = true; }
                    }
                    if (!settled) {
                        ENC_TIC();
                        over = used_bits(mid, 0) > avail;
                        ENC_TOC(18); ENC_COUNT(17, 1);
                        if (over) { over_noise = mid; _Pragma("unroll") for (int b = 0; b < NB; b++) over_bits[b] = last[b]; }
                        else { fit_noise = mid; _Pragma("unroll") for (int b = 0; b < NB; b++) fit_bits[b] = last[b]; }
                    }
                    tnode = 2 * tnode + 1 + (over ? 1u : 0u); depth++;
                } else over = used_bits(mid, 0) > avail;
                if (over) low = mid + 1; else high = mid;
            }
            noise_level = (low == 255 && over) ? -1 : low;
            if (noise_level >= 0) break;
            highest -= 2;
            if (highest < 0) { status = CRI_ERR_HCA_ENCODE; break; }
            wave_lds_sync();
            if (lane < C) { L.sfac[lane * 128 + highest + 1] = 0; L.sfac[lane * 128 + highest + 2] = 0; }
            wave_lds_sync();
            enc_header_length(F, L, lane);
            load_bands();
            fit_noise = over_noise = -1000;                // (the scalefactors changed)
        }
    }
    ENC_MARK(3);
    if (status == 0 && noise_level != 0) {
        // only two resolutions per band occur in this search (noise_level and noise_level - 1): cost them once
        int costA[NB], costB[NB];
        if constexpr (CT > 0) {
#pragma unroll
            for (int b = 0; b < NB; b++) {
                if (fit_noise == noise_level) costA[b] = fit_bits[b];
                else costA[b] = inr[b] ? enc_band_bits(T, xr[b], enc_resolution(T, sfr[b], noise_level)) : 0;
                if (over_noise == noise_level - 1) costB[b] = over_bits[b];
                else costB[b] = inr[b] ? enc_band_bits(T, xr[b], enc_resolution(T, sfr[b], noise_level - 1)) : 0;
            }
        }
        const int hb = header_bits();
        auto bits_at = [&](int eb) -> int {
            if constexpr (CT > 0) {
                int part = 0;
#pragma unroll
                for (int b = 0; b < NB; b++) part += ((int)lane + 64 * (b & 1)) < eb ? costB[b] : costA[b];
                return hb + wave_sum(part);
            } else return enc_used_bits(F, L, T, lane, noise_level, eb);
        };
        int low = 0, high = 127;
        while ((high - low > 1) || (low - high > 1)) {
            const int mid = (low + high) / 2;
            const int v = bits_at(mid);
            if (avail < v) high = mid - 1; else low = mid;
        }
        int level;
        if (low == high) level = low < 127 ? low : -1;
        else level = bits_at(high) > avail ? low : high;
        if (level < 0) status = CRI_ERR_HCA_ENCODE; else eval_boundary = level;
    }
    ENC_MARK(4);
    uint8_t* dst = a.out + st.dst_offset + (uint64_t)f * F.frame_size;
    if (status != 0) {
        if (lane == 0 && a.status) atomicMin(a.status + st.item, status);
        for (uint32_t i = lane; i < F.frame_size; i += 64) dst[i] = 0;
        return;
    }

    // ---- CalculateFrameResolutions (hca.cpp:2868-2876)
    for (uint32_t c = 0; c < C; c++)
        for (uint32_t i = lane; i < 128; i += 64) {
            int r = 0;
            if (i < F.coded(c)) r = enc_resolution(T, L.sfac[c * 128 + i], (int)i < eval_boundary ? noise_level - 1 : noise_level);
            L.res[c * 128 + i] = (uint8_t)r;
        }
    for (uint32_t i = lane; i < nwords; i += 64) L.words[i] = 0;
    wave_lds_sync();

    // ---- PackFrame (hca.cpp:2938-2963): sync word, 9+7 bit header, per channel scalefactors + intensity / HFR scales
    uint32_t pos = 16;
    if (lane == 0) { L.words[0] = 0xFFFF0000u; put_bits(L.words, 16, (uint32_t)noise_level, 9); put_bits(L.words, 25, (uint32_t)eval_boundary, 7); }
    pos += 16;
    for (uint32_t c = 0; c < C; c++) {
        const int db = L.dbits[c], coded = (int)F.coded(c);
        const uint8_t* sf = L.sfac + c * 128;
        if (lane == 0) put_bits(L.words, pos, (uint32_t)db, 3);
        pos += 3;
        if (db == 6) {
            for (int i = (int)lane; i < coded; i += 64) put_bits(L.words, pos + 6 * i, sf[i], 6);
            pos += 6 * coded;
        } else if (db != 0) {                                          // WriteScalesFactors, hca.cpp:2894-2918
            const int maxd = (1 << (db - 1)) - 1, esc = (1 << db) - 1;
            // bands 2*lane, 2*lane+1 per lane so that the prefix sum runs in band order; their codes (a delta, or the escape value
            // followed by the 6-bit scalefactor; band 0 is always the plain 6 bits) leave as one write of at most 22 bits
            int len0 = 0, len1 = 0;
            uint32_t c0 = 0, c1 = 0;
            const int b0 = 2 * (int)lane, b1 = b0 + 1;
            if (b0 < coded) {
                const int d0 = b0 == 0 ? 0 : (int)sf[b0] - (int)sf[b0 - 1];
                const bool e0 = (d0 < 0 ? -d0 : d0) > maxd;
                len0 = b0 == 0 ? 6 : (e0 ? db + 6 : db);
                c0 = b0 == 0 ? (uint32_t)sf[0] : (e0 ? (((uint32_t)esc << 6) | sf[b0]) : (uint32_t)(maxd + d0));
            }
            if (b1 < coded) {
                const int d1 = (int)sf[b1] - (int)sf[b1 - 1];
                const bool e1 = (d1 < 0 ? -d1 : d1) > maxd;
                len1 = e1 ? db + 6 : db;
                c1 = e1 ? (((uint32_t)esc << 6) | sf[b1]) : (uint32_t)(maxd + d1);
            }
            const uint32_t incl01 = wave_incl_scan_dpp((uint32_t)(len0 + len1));
            put_bits(L.words, pos + (incl01 - (uint32_t)(len0 + len1)), (c0 << len1) | c1, (uint32_t)(len0 + len1));
            pos += (uint32_t)__builtin_amdgcn_readlane((int)incl01, 63);
        }
        if (F.type(c) == CRI_CH_SECONDARY) {
            if (lane < 8) put_bits(L.words, pos + 4 * lane, L.inten[c * 8 + lane], 4);
            pos += 32;
        } else if (F.groups > 0) {
            if (lane < F.groups) put_bits(L.words, pos + 6 * lane, (uint32_t)L.hfrs[c * 8 + lane], 6);
            pos += 6 * F.groups;
        }
    }
    ENC_MARK(5);
    // spectra: QuantizeSpectra (hca.cpp:2878-2892) + WriteSpectra (2920-2936)
    if constexpr (CT > 0) {
        // bands 2*lane and 2*lane + 1 of each channel (neighbours in the bit stream: their two codes go out as one write);
        // per-band constants hoisted out of the subframe loop
        int rb[NB], downb[NB]; float invb[NB], upb[NB]; uint32_t mbb[NB];
#pragma unroll
        for (int b = 0; b < NB; b++) {
            rb[b] = L.res[(b >> 1) * 128 + 2 * lane + (b & 1)];           // 0 past the coded bands
            invb[b] = T.inv[rb[b]]; upb[b] = invb[b] + 1; downb[b] = (int)((double)invb[b] + 0.5);
            mbb[b] = rb[b] >= 8 ? (uint32_t)enc_maxbits(rb[b]) - 1 : 1u;
        }
#pragma unroll 1
        for (uint32_t sf = 0; sf < 8; sf++) {
#pragma unroll
            for (int c = 0; c < CT; c++) {
                uint32_t code[2], len[2];
                const float2 xv = *(const float2*)(L.sc + (c * 8 + sf) * 128 + 2 * lane);
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int b = 2 * c + h, r = rb[b];
                    const int q = (int)((h ? xv.y : xv.x) * invb[b] + upb[b]) - downb[b];
                    const uint32_t ti = (uint32_t)r * 16 + ((uint32_t)(q + 8) & 15);
                    const uint32_t lt = T.clen[ti & 127], ct = T.code[ti & 127];
                    const uint32_t mag = (uint32_t)(q < 0 ? -q : q) & ((1u << mbb[b]) - 1);
                    const uint32_t lb = q != 0 ? mbb[b] + 1 : mbb[b], cb = q != 0 ? ((mag << 1) | (q > 0 ? 0u : 1u)) : 0u;
                    len[h] = r == 0 ? 0u : (r < 8 ? lt : lb);
                    code[h] = r == 0 ? 0u : (r < 8 ? ct : cb);
                }
                const uint32_t tl = len[0] + len[1];                       // at most 13 + 13 bits
                const uint32_t both = (code[0] << len[1]) | code[1];
                const uint32_t incl = wave_incl_scan_dpp(tl);
                const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                // BitWriter drops writes that do not fit (IO.cpp:131-134); the rate loop guarantees they do
                put_bits(L.words, pos + (incl - tl), both, tl);
                pos += tot;
            }
        }
    } else {
    // bands 2*lane, 2*lane+1 per lane
    for (uint32_t sf = 0; sf < 8; sf++) {
        for (uint32_t c = 0; c < C; c++) {
            const uint32_t coded = F.coded(c);
            uint32_t code[2] = {0, 0}, len[2] = {0, 0};
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint32_t i = 2 * lane + h;
                if (i >= coded) continue;
                const int r = L.res[c * 128 + i];
                if (r == 0) continue;
                const float inv = T.inv[r], up = inv + 1;
                const int down = (int)((double)inv + 0.5);
                const int q = (int)(L.sc[(c * 8 + sf) * 128 + i] * inv + up) - down;
                if (r < 8) { len[h] = T.clen[r * 16 + ((q + 8) & 15)]; code[h] = T.code[r * 16 + ((q + 8) & 15)]; }
                else {
                    const uint32_t mb = (uint32_t)enc_maxbits(r) - 1, mag = (uint32_t)(q < 0 ? -q : q) & ((1u << mb) - 1);
                    if (q != 0) { code[h] = (mag << 1) | (q > 0 ? 0u : 1u); len[h] = mb + 1; } else { code[h] = 0; len[h] = mb; }
                }
            }
            const int off = wave_excl_scan((int)(len[0] + len[1]), lane);
            // BitWriter drops writes that do not fit (IO.cpp:131-134); the rate loop guarantees they do
            put_bits(L.words, pos + off, code[0], len[0]);
            put_bits(L.words, pos + off + len[0], code[1], len[1]);
            pos += (uint32_t)wave_sum((int)(len[0] + len[1]));
        }
    }
    }
    wave_lds_sync();

    ENC_MARK(6);
    // ---- CRC16 over frame_size-2 bytes (hca.cpp:2961-2962), chunk per lane, then one multiply per lane and an xor across the wave.
    // The message is front-padded with zero bytes to 64*m bytes (leading zeros do not change a zero-init CRC).
    {
        const uint32_t n = F.frame_size - 2, m = a.crc_chunk, pad = 64 * m - n;
        const uint4 cw0 = ((const uint4*)(a.crc_mul + lane * 16))[0], cw1 = ((const uint4*)(a.crc_mul + lane * 16))[1];
        uint32_t crc = 0;
        for (uint32_t k = 0; k < m; k++) {
            const uint32_t j = lane * m + k;
            uint32_t b = 0;
            if (j >= pad) { const uint32_t q = j - pad; b = (L.words[q >> 2] >> (24 - 8 * (q & 3))) & 0xFF; }
            crc = crc16_step_enc(crc, b);
        }
        // lane l's chunk stands 8*m*(63 - l) bits above the end of the message: multiply by x^that (mod P) -- the launch's table holds
        // x^bit * x^(8*m*(63 - l)) for the 16 bits of the chunk's remainder, a 32-byte row per lane -- and xor the 64 products together
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
        if (lane == 0) put_bits(L.words, (F.frame_size - 2) * 8, crc & 0xFFFF, 16);
    }
    wave_lds_sync();
    // the frame image is big-endian words; whole dwords go out byte-swapped (unaligned dword stores), then the last bytes
    for (uint32_t i = lane; 4 * i + 4 <= F.frame_size; i += 64) { const uint32_t w = __builtin_bswap32(L.words[i]); __builtin_memcpy(dst + 4 * i, &w, 4); }
    if (lane < (F.frame_size & 3)) { const uint32_t i = (F.frame_size & ~3u) + lane; dst[i] = (uint8_t)(L.words[i >> 2] >> (24 - 8 * (i & 3))); }
    ENC_MARK(7);
    ENC_PROF_FLUSH();
}

// LDS of one frame (wave): spectra, MDCT work buffers / frame image, small per-channel arrays
size_t hca_encode_lds_per_wave(uint32_t C, uint32_t frame_size) {
    size_t nwords = (frame_size + 3) / 4 + 1;
    if (nwords < 320) nwords = 320;                         // also the 640-sample PCM staging buffer of the MDCT
    const size_t n = (size_t)C * 1024 * 4 + nwords * 4 + C * 8 * 4 + 8 * 4 + C * 8 * 4 + C * 4 * 2 + C * 128 * 2 + C * 8 + 64;
    return (n + 15) & ~(size_t)15;
}
// frames per workgroup: as many as fit the 160 KB of LDS, at most ENC_WAVES (0: not even one fits)
uint32_t hca_encode_waves(uint32_t C, uint32_t frame_size) {
    const size_t room = 160 * 1024 - ENC_TAB_BYTES, per = hca_encode_lds_per_wave(C, frame_size);
    const size_t w = room / per;
    return (uint32_t)(w > ENC_WAVES ? ENC_WAVES : w);
}
size_t hca_encode_lds_bytes(uint32_t C, uint32_t frame_size) {
    const uint32_t w = hca_encode_waves(C, frame_size);
    return ENC_TAB_BYTES + (w ? w : 1) * hca_encode_lds_per_wave(C, frame_size);
}

void launch_hca_encode(const HcaEncArgs& a, hipStream_t s) {
    if (!a.frames) return;
    HcaEncArgs b = a;
    b.lds_per_wave = (uint32_t)hca_encode_lds_per_wave(a.channels, a.frame_size);
    const uint32_t w = hca_encode_waves(a.channels, a.frame_size);
    if (!w) return;
    const dim3 grid((a.frames + w - 1) / w), block(64 * w);
    const size_t lds = hca_encode_lds_bytes(a.channels, a.frame_size);
    switch (a.channels) {                                      // register-resident rate loop for the usual layouts
        case 1: hipLaunchKernelGGL(k_hca_encode<1>, grid, block, lds, s, b); break;
        case 2: hipLaunchKernelGGL(k_hca_encode<2>, grid, block, lds, s, b); break;
        case 4: hipLaunchKernelGGL(k_hca_encode<4>, grid, block, lds, s, b); break;
        case 6: hipLaunchKernelGGL(k_hca_encode<6>, grid, block, lds, s, b); break;
        case 8: hipLaunchKernelGGL(k_hca_encode<8>, grid, block, lds, s, b); break;
        default: hipLaunchKernelGGL(k_hca_encode<0>, grid, block, lds, s, b); break;
    }
}

}  // namespace cri

#ifdef CRI_ENC_PROFILE
extern "C" int cri_debug_enc_profile(unsigned long long* out16, int reset) {
    static unsigned long long h[1024][24];
    if (out16) {                                             // (24 values: 16 phases, then rate-loop detail)
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(cri::g_enc_prof), sizeof h) != hipSuccess) return -1;
        for (int k = 0; k < 24; k++) { out16[k] = 0; for (int s = 0; s < 1024; s++) out16[k] += h[s][k]; }
    }
    if (reset) { memset(h, 0, sizeof h); if (hipMemcpyToSymbol(HIP_SYMBOL(cri::g_enc_prof), h, sizeof h) != hipSuccess) return -1; }
    return 0;
}
#endif
