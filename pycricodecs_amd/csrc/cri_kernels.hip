// cri_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X, wave64) of the ADX / HCA hot path.
//
// Kernel map (reference functions replaced, /root/reference/CriCodecs/...):
//   k_hca_unpack     clHCA_DecodeBlock_unpack hca.cpp:1149-1205 = sync + CRC16 (186-211) + decipher (491-497) +
//                    unpack_scalefactors (1290-1358) + unpack_intensity (1361-1441) + calculate_resolution (1444-1494)
//                    + dequantize_coefficients' bit parse (1540-1571).  One LANE per frame (the bit parse is a
//                    serial chain per frame), 64 frames per wave, frame bytes staged transposed in LDS.
//   k_hca_transform  calculate_gain (1498-1507), the float half of dequantize (1566), reconstruct_high_frequency
//                    (1638-1683), apply_intensity_stereo (1696-1714), imdct_transform (1898-2019),
//                    clHCA_ReadSamples16 (339-360) and HcaDecode's delay/trim (3401-3452).  One WAVE per frame.
//   k_adx_decode     ChannelFrame::Decode adx.cpp:189-214 + block loop 404-413.  One lane per (file, channel) chain.
//   k_adx_encode     ChannelFrame::Encode adx.cpp:215-273 + loop 492-502.       One lane per (file, channel) chain.
//   k_hca_crypt      HcaCrypt frame loop hca.cpp:3322-3327.
// All float work is single IEEE binary32 operations in the reference's order (compiled with -ffp-contract=off).
#include <hip/hip_runtime.h>
#include "cri_kernels.h"
#include "../../include/cricodecs_hip.h"

#define CRI_TABLE_QUAL static __device__ const
#include "cri_tables.h"

namespace cri {

// ------------------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ld_u32_unaligned(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint32_t bswap32(uint32_t v) { return __builtin_bswap32(v); }
__device__ __forceinline__ uint64_t readlane64(uint64_t v, int l) {
    uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, l), hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), l);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ int32_t clamp_sym(int32_t v, int32_t limit) { return v > limit ? limit : (v < ~limit ? ~limit : v); }

// stream lookup: largest s in [lo, hi) with streams[s].first_frame <= g
__device__ __forceinline__ uint32_t find_stream(const HcaStream* streams, uint32_t lo, uint32_t hi, uint32_t g) {
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (streams[mid].first_frame <= g) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ void k_fill_i32(int32_t* p, int32_t v, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
void launch_fill_i32(int32_t* p, int32_t v, uint32_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_fill_i32, dim3((n + 255) / 256), dim3(256), 0, s, p, v, n);
}

__global__ void k_scatter_images(const uint8_t* img, const uint64_t* img_off, const uint64_t* dst_off, uint32_t n, uint8_t* out) {
    uint32_t i = blockIdx.x;
    if (i >= n) return;
    uint64_t b = img_off[i], e = img_off[i + 1];
    uint8_t* d = out + dst_off[i];
    for (uint64_t k = b + threadIdx.x; k < e; k += blockDim.x) d[k - b] = img[k];
}
void launch_scatter_images(const uint8_t* img, const uint64_t* img_off, const uint64_t* dst_off, uint32_t n, uint8_t* out, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_scatter_images, dim3(n), dim3(64), 0, s, img, img_off, dst_off, n, out);
}

// ------------------------------------------------------------------------------------------------------------
// HCA unpack: one lane per frame
// ------------------------------------------------------------------------------------------------------------
// LDS layout (per workgroup = one wave), W = frames per wave (power of two):
//   rows   uint32 [(R+1)][W]   frame bytes, row r = bytes 4r..4r+3 of each frame, big-endian after pass 1 (row R = 0)
//   ostage uint32 [64][W+1]    per-lane output words, transposed on flush
//   resb   uint8  [C*128][W]   resolution per coded band
//   tables: crc16 u16[256], curve->res u8[80], code len u8[128], code val i8[128], cipher u8[n][256] (n <= 16)
struct UnpackLds {
    uint32_t* rows; uint32_t* ostage; uint8_t* resb; uint16_t* crc; uint8_t* curve; uint8_t* clen; int8_t* cval; uint8_t* cipher;
};
__host__ __device__ inline size_t unpack_lds_carve(uint32_t rows, uint32_t C, uint32_t n_cipher, uint32_t W, size_t off[8]) {
    size_t o = 0;
    off[0] = o; o += (size_t)(rows + 1) * W * 4;
    off[1] = o; o += (size_t)64 * (W + 1) * 4;
    off[2] = o; o += (size_t)C * 128 * W;
    o = (o + 15) & ~(size_t)15;
    off[3] = o; o += 512;
    off[4] = o; o += 80;
    off[5] = o; o += 128;
    off[6] = o; o += 128;
    off[7] = o; o += (size_t)(n_cipher <= 16 ? n_cipher : 0) * 256;
    return (o + 15) & ~(size_t)15;
}
size_t hca_unpack_lds_bytes(uint32_t frame_size, uint32_t channels, uint32_t n_cipher, uint32_t fpw) {
    size_t off[8];
    return unpack_lds_carve((frame_size + 3) / 4, channels, n_cipher, fpw, off);
}

// MSB-first peek of n (0..16) bits at bit `pos` of the lane's frame, with the reference reader's end-of-frame
// behaviour (hca.cpp:225-281): 0 when the read crosses the end; and 0 when fewer than 24 (16) bits are left but
// the read spans more than 16 (8) bits from its byte start -- the reference then picks a window that is too narrow.
__device__ __forceinline__ uint32_t hca_peek(const uint32_t* rows, uint32_t W, uint32_t lane, int pos, int n, int size) {
    int left = size - pos;
    if (n > left || n == 0) return 0;
    uint32_t r = (uint32_t)pos >> 5, sh = (uint32_t)pos & 31;
    uint32_t hi = rows[r * W + lane], lo = rows[(r + 1) * W + lane];
    uint32_t w = (uint32_t)(((((uint64_t)hi << 32) | lo) << sh) >> 32);
    uint32_t v = w >> (32 - n);
    if (left < 24) {
        int off = n + (pos & 7);
        if (off >= 17 || (off >= 9 && left < 16)) v = 0;
    }
    return v;
}

__global__ __launch_bounds__(64) void k_hca_unpack(HcaDecArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const HcaFormat& F = a.formats[a.format];
    const uint32_t W = a.fpw, R = a.rows, C = F.channels, lane = threadIdx.x;
    const int fs = (int)F.frame_size, size_bits = fs * 8;
    size_t off[8];
    unpack_lds_carve(R, C, a.n_cipher, W, off);
    uint32_t* rows = (uint32_t*)(smem + off[0]);
    uint32_t* ostage = (uint32_t*)(smem + off[1]);
    uint8_t* resb = smem + off[2];
    uint16_t* crc_tab = (uint16_t*)(smem + off[3]);
    uint8_t* curve = smem + off[4];
    uint8_t* clen = smem + off[5];
    int8_t* cval = (int8_t*)(smem + off[6]);
    uint8_t* cipher_lds = smem + off[7];
    const bool cipher_in_lds = a.n_cipher <= 16;

    // tables -> LDS
    for (uint32_t i = lane; i < 256; i += 64) crc_tab[i] = CRI_CRC16_TAB[i];
    for (uint32_t i = lane; i < 66; i += 64) curve[i] = HCA_CURVE_TO_RES[i];
    for (uint32_t i = lane; i < 128; i += 64) { clen[i] = HCA_CODE_LEN[i]; cval[i] = HCA_CODE_VAL[i]; }
    if (cipher_in_lds) for (uint32_t i = lane; i < a.n_cipher * 256; i += 64) cipher_lds[i] = a.cipher_tables[i];

    // which frame does this lane own
    const uint32_t g = blockIdx.x * W + lane;
    const bool active = lane < W;          // lanes beyond the frames-per-wave count only help with staging / flushing
    const bool valid = active && g < a.frames;
    uint32_t si = a.stream_begin, f = 0;
    if (valid) { si = find_stream(a.streams, a.stream_begin, a.stream_end, g); f = g - a.streams[si].first_frame; }
    const HcaStream st = a.streams[si];
    const uint8_t* src = a.in + st.src_offset + (uint64_t)f * (uint32_t)fs;
    uint8_t* rec = a.scratch + st.scratch_offset + (uint64_t)f * F.record_bytes;

    // ---- phase A: cooperative, coalesced staging of W frames into the transposed row array
    for (uint32_t fr = 0; fr < W; fr++) {
        const bool fv = __builtin_amdgcn_readlane((int)valid, fr) != 0;
        if (!fv) { for (uint32_t r = lane; r < R; r += 64) rows[r * W + fr] = 0; continue; }
        const uint8_t* p = (const uint8_t*)readlane64((uint64_t)src, fr);
        for (uint32_t r = lane; r < R; r += 64) {
            uint32_t v;
            if ((int)(4 * r + 4) <= fs) v = ld_u32_unaligned(p + 4 * r);
            else { v = 0; for (int k = 0; 4 * (int)r + k < fs; k++) v |= (uint32_t)p[4 * r + k] << (8 * k); }
            rows[r * W + fr] = v;
        }
    }
    if (lane < W) rows[R * W + lane] = 0;
    __syncthreads();

    int status = 0;
    uint32_t packed = 0, flags = 0;
    int pos = 16;
    if (valid) {
        // ---- pass 1: sync word, CRC16 over the raw bytes, decipher, store big-endian words
        const uint8_t* ct = cipher_in_lds ? cipher_lds + st.cipher * 256 : a.cipher_tables + st.cipher * 256;
        uint32_t crc = 0;
        for (uint32_t r = 0; r < R; r++) {
            uint32_t raw = rows[r * W + lane], be = 0;
            int nb = fs - 4 * (int)r; nb = nb > 4 ? 4 : nb;
            if (r == 0 && (raw & 0xFFFF) != 0xFFFF) status = CRI_ERR_HCA_FRAME(4);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (k < nb) {
                    uint32_t b = (raw >> (8 * k)) & 0xFF;
                    crc = ((crc << 8) ^ crc_tab[(crc >> 8) ^ b]) & 0xFFFF;
                    uint32_t d = cipher_in_lds ? ct[b] : __ldg(ct + b);
                    be |= d << (24 - 8 * k);
                }
            }
            rows[r * W + lane] = be;
        }
        if (status == 0 && crc != 0) status = CRI_ERR_HCA_FRAME(3);
    }
    if (valid && status == 0) {
        // ---- frame header (hca.cpp:1175-1190)
        uint32_t nl = hca_peek(rows, W, lane, pos, 9, size_bits); pos += 9;
        uint32_t eb = hca_peek(rows, W, lane, pos, 7, size_bits); pos += 7;
        packed = (nl << 8) - eb;
    }
    const uint8_t* ath = a.ath_tables + F.ath_index * 128;
    for (uint32_t c = 0; c < C; c++) {
        const uint32_t coded = F.coded[c], type = F.type[c], groups = F.hfr_group_count;
        uint32_t cs = coded, extra = 0;
        if (!(type == CRI_CH_SECONDARY || groups == 0 || F.version <= 0x0200)) { extra = groups; cs += extra; }
        // zero the scalefactor staging rows (32 words per lane)
        if (active) for (uint32_t r = 0; r < 32; r++) ostage[r * (W + 1) + lane] = 0;
        uint8_t* sfst = (uint8_t*)ostage;
        const bool live = valid && status == 0;
        uint32_t db = 0, value = 0;
        if (live) { db = hca_peek(rows, W, lane, pos, 3, size_bits); pos += 3; }
        if (cs > 128) { if (live) status = CRI_ERR_HCA_FRAME(5); cs = 0; }
        const uint32_t expected = (1u << db) - 1;
        for (uint32_t i = 0; i < cs; i++) {                       // hca.cpp:1310-1350, all lanes in lock step
            uint32_t v = 0;
            if (live && status == 0 && db > 0) {
                const bool direct = db >= 6 || i == 0;
                int n1 = direct ? 6 : (int)db;
                uint32_t x = hca_peek(rows, W, lane, pos, n1, size_bits); pos += n1;
                if (direct) v = x;
                else if (x == expected) { v = hca_peek(rows, W, lane, pos, 6, size_bits); pos += 6; }
                else {
                    int t = (int)value + ((int)x - (int)(expected >> 1));
                    if (t < 0 || t >= 64) status = CRI_ERR_HCA_FRAME(5);
                    v = (value - (expected >> 1) + x) & 0x3F;
                }
                value = v;
            }
            if (active) sfst[((i >> 2) * (W + 1) + lane) * 4 + (i & 3)] = (uint8_t)v;
            if (i < coded && active) {                                      // calculate_resolution, hca.cpp:1450-1488
                uint32_t res = 0;
                if (v > 0) {
                    int noise = (int)ath[i] + (int)((packed + i) >> 8);
                    int cp = noise + 1 - (int)((5 * v) >> 1);
                    res = cp < 0 ? 15u : (cp <= 65 ? (uint32_t)curve[cp] : 0u);
                    res = res > F.max_res ? F.max_res : (res < F.min_res ? F.min_res : res);
                }
                resb[(c * 128 + i) * W + lane] = (uint8_t)res;
            }
        }
        // derived HFR scales of v3.0 (hca.cpp:1353-1355); the entry one past the coded+extra range reads as 0
        for (uint32_t i = 0; i < extra; i++) {
            uint32_t srci = cs - i;
            uint8_t sv = (srci < cs && active) ? sfst[((srci >> 2) * (W + 1) + lane) * 4 + (srci & 3)] : 0;
            uint32_t di = 127 - i;
            if (active) sfst[((di >> 2) * (W + 1) + lane) * 4 + (di & 3)] = sv;
        }
        // unpack_intensity, hca.cpp:1361-1441
        uint32_t inten_lo = 0, inten_hi = 0;
        if (type == CRI_CH_SECONDARY) {
            if (valid && status == 0) {
                uint8_t iv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                if (F.version <= 0x0200) {
                    uint32_t v = hca_peek(rows, W, lane, pos, 4, size_bits);
                    iv[0] = (uint8_t)v;
                    if (v < 15) { pos += 4; for (int k = 1; k < 8; k++) { iv[k] = (uint8_t)hca_peek(rows, W, lane, pos, 4, size_bits); pos += 4; } }
                    else flags |= 1u << c;                        // intensity[1..7] keep the previous frame's values
                } else {
                    uint32_t v = hca_peek(rows, W, lane, pos, 4, size_bits);
                    if (v < 15) {
                        pos += 4;
                        uint32_t dbi = hca_peek(rows, W, lane, pos, 2, size_bits); pos += 2;
                        iv[0] = (uint8_t)v;
                        if (dbi == 3) { for (int k = 1; k < 8; k++) { iv[k] = (uint8_t)hca_peek(rows, W, lane, pos, 4, size_bits); pos += 4; } }
                        else {
                            uint32_t bmax = (2u << dbi) - 1, bits = dbi + 1;
                            bool bad = false;
                            for (int k = 1; k < 8; k++) {
                                if (bad) { flags |= 1u << (16 + c); break; }   // reference returns early; later entries stay stale
                                uint32_t delta = hca_peek(rows, W, lane, pos, (int)bits, size_bits); pos += (int)bits;
                                if (delta == bmax) { v = hca_peek(rows, W, lane, pos, 4, size_bits); pos += 4; }
                                else { v = (v - (bmax >> 1) + delta) & 0xFF; if (v > 15) { bad = true; continue; } }
                                iv[k] = (uint8_t)v;
                            }
                        }
                    } else { pos += 4; for (int k = 0; k < 8; k++) iv[k] = 7; }
                }
                inten_lo = iv[0] | (iv[1] << 8) | (iv[2] << 16) | ((uint32_t)iv[3] << 24);
                inten_hi = iv[4] | (iv[5] << 8) | (iv[6] << 16) | ((uint32_t)iv[7] << 24);
            }
        } else if (F.version <= 0x0200) {
            for (uint32_t k = 0; k < groups; k++) {
                uint32_t v = 0;
                if (valid && status == 0) { v = hca_peek(rows, W, lane, pos, 6, size_bits); pos += 6; }
                uint32_t di = 128 - groups + k;
                if (active) sfst[((di >> 2) * (W + 1) + lane) * 4 + (di & 3)] = (uint8_t)v;
            }
        }
        if (valid) { uint32_t* ip = (uint32_t*)(rec + HCA_REC_INT(C, c)); ip[0] = inten_lo; ip[1] = inten_hi; }
        // flush the 128 scalefactor bytes of every frame, coalesced (lanes 0..31 carry one word each)
        __syncthreads();
        for (uint32_t fr = 0; fr < W; fr++) {
            if (!__builtin_amdgcn_readlane((int)valid, fr)) continue;
            uint8_t* rp = (uint8_t*)readlane64((uint64_t)rec, fr);
            if (lane < 32) ((uint32_t*)(rp + HCA_REC_SF(C, c)))[lane] = ostage[lane * (W + 1) + fr];
        }
        __syncthreads();
    }
    // ---- spectra: 8 subframes x C channels x coded symbols, serial per lane (hca.cpp:1194-1199, 1540-1571)
    const uint64_t maxbits_packed = 0xCBA9876544443320ull;
    for (uint32_t sf = 0; sf < 8; sf++) {
        for (uint32_t c = 0; c < C; c++) {
            const uint32_t coded = F.coded[c];
            const bool live = valid && status == 0;
            uint32_t word = 0;
            for (uint32_t i = 0; i < coded; i++) {
                int val = 0;
                if (live) {
                    uint32_t res = resb[(c * 128 + i) * W + lane];
                    int bits = (int)((maxbits_packed >> (res * 4)) & 15);
                    uint32_t code = hca_peek(rows, W, lane, pos, bits, size_bits);
                    if (res > 7) {
                        int mag = (int)(code >> 1);
                        val = (code & 1) ? -mag : mag;
                        pos += bits - (mag == 0 ? 1 : 0);
                    } else {
                        uint32_t idx = (res << 4) + code;
                        pos += (int)clen[idx];
                        val = cval[idx];
                    }
                }
                if (i & 1) { word |= (uint32_t)(uint16_t)(int16_t)val << 16; if (active) ostage[(i >> 1) * (W + 1) + lane] = word; }
                else word = (uint32_t)(uint16_t)(int16_t)val;
            }
            if ((coded & 1) && active) ostage[(coded >> 1) * (W + 1) + lane] = word;
            const uint32_t nwords = (coded + 1) >> 1;
            __syncthreads();
            for (uint32_t fr = 0; fr < W; fr++) {
                if (!__builtin_amdgcn_readlane((int)valid, fr)) continue;
                uint8_t* rp = (uint8_t*)readlane64((uint64_t)rec, fr);
                if (lane < nwords) ((uint32_t*)(rp + HCA_REC_QC(C, sf, c)))[lane] = ostage[lane * (W + 1) + fr];
            }
            __syncthreads();
        }
    }
    if (valid) {
        uint32_t* tail = (uint32_t*)(rec + HCA_REC_TAIL(C));
        tail[0] = packed; tail[1] = (uint32_t)status; tail[2] = flags; tail[3] = (uint32_t)pos;
    }
}

void launch_hca_unpack(const HcaDecArgs& a, hipStream_t s) {
    if (!a.frames) return;
    uint32_t blocks = (a.frames + a.fpw - 1) / a.fpw;
    hipLaunchKernelGGL(k_hca_unpack, dim3(blocks), dim3(64), a.unpack_lds, s, a);
}

// ------------------------------------------------------------------------------------------------------------
// HCA transform: one wave per frame
// ------------------------------------------------------------------------------------------------------------
// LDS (floats): S[C][128] spectra / dct, G[C][128] gains, P[C][128] overlap tail, T[128] ping-pong partner.
__device__ __forceinline__ int32_t cvt_trunc_x86(float v) {
    // (int)v as the x86-64 reference build evaluates it: out-of-range and NaN give INT_MIN
    return (v >= -2147483648.0f && v < 2147483648.0f) ? (int32_t)v : (int32_t)0x80000000;
}

// 128-point DCT-IV of hca.cpp:1898-1980 on LDS buffers x (in/out) and y (scratch); lane m owns pair m.
__device__ __forceinline__ void imdct_dct4(float* x, float* y, uint32_t m, const float tw_s[7], const float tw_c[7]) {
#pragma unroll
    for (int i = 0; i < 7; i++) {                     // sum / difference stages
        const uint32_t c = 64u >> i;
        const uint32_t j = m >> (6 - i), k = m & (c - 1);
        float p = x[2 * m], q = x[2 * m + 1];
        y[2 * c * j + k] = p + q;
        y[2 * c * j + c + k] = p - q;
        __syncthreads();
        float* t = x; x = y; y = t;
    }
#pragma unroll
    for (int i = 0; i < 7; i++) {                     // rotation stages
        const uint32_t c = 1u << i;
        const uint32_t j = m >> i, k = m & (c - 1);
        float p = x[2 * c * j + k], q = x[2 * c * j + c + k];
        float ps = p * tw_s[i], qc = q * tw_c[i], pc = p * tw_c[i], qs = q * tw_s[i];
        y[2 * c * j + k] = ps - qc;
        y[2 * c * j + 2 * c - 1 - k] = pc + qs;
        __syncthreads();
        float* t = x; x = y; y = t;
    }
}

struct TransformCtx {
    const HcaFormat* F; const uint8_t* ath; float* S; float* G; uint32_t C, lane;
};

// gains of one frame: calculate_resolution + calculate_gain (hca.cpp:1444-1507), two bands per lane
__device__ __forceinline__ void frame_gains(const TransformCtx& X, const uint8_t* rec) {
    const HcaFormat& F = *X.F;
    const uint32_t packed = ((const uint32_t*)(rec + HCA_REC_TAIL(X.C)))[0];
    for (uint32_t c = 0; c < X.C; c++) {
        const uint32_t sf2 = ((const uint16_t*)(rec + HCA_REC_SF(X.C, c)))[X.lane];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t i = 2 * X.lane + h, v = (sf2 >> (8 * h)) & 0xFF;
            float gain = 0.0f;
            if (i < F.coded[c]) {
                uint32_t res = 0;
                if (v > 0) {
                    int noise = (int)X.ath[i] + (int)((packed + i) >> 8);
                    int cp = noise + 1 - (int)((5 * v) >> 1);
                    res = cp < 0 ? 15u : (cp <= 65 ? (uint32_t)HCA_CURVE_TO_RES[cp] : 0u);
                    res = res > F.max_res ? F.max_res : (res < F.min_res ? F.min_res : res);
                }
                gain = HCA_DEQ_SCALE[v & 63] * HCA_DEQ_RANGE[res];
            }
            X.G[c * 128 + i] = gain;
        }
    }
}

// spectra of subframe sf of one frame into S: dequantise, HFR, intensity stereo (hca.cpp:1566, 1638-1683, 1696-1714)
__device__ __forceinline__ void frame_spectra(const TransformCtx& X, const uint8_t* rec, uint32_t sf, const uint8_t* inten /* [C][8] resolved */) {
    const HcaFormat& F = *X.F;
    const uint32_t C = X.C, lane = X.lane;
    for (uint32_t c = 0; c < C; c++) {
        const uint32_t q2 = ((const uint32_t*)(rec + HCA_REC_QC(C, sf, c)))[lane];
        const uint32_t i0 = 2 * lane;
        float q0 = (float)(int)(int16_t)(q2 & 0xFFFF), q1 = (float)(int)(int16_t)(q2 >> 16);
        X.S[c * 128 + i0] = i0 < F.coded[c] ? X.G[c * 128 + i0] * q0 : 0.0f;
        X.S[c * 128 + i0 + 1] = i0 + 1 < F.coded[c] ? X.G[c * 128 + i0 + 1] * q1 : 0.0f;
    }
    __syncthreads();
    if (F.bands_per_hfr_group > 0) {
        const int start = (int)(F.stereo_bands + F.base_bands), bpg = (int)F.bands_per_hfr_group, groups = (int)F.hfr_group_count;
        const int limit = F.version <= 0x0200 ? groups : (groups >> 1);
        const int total = (int)F.total_bands;
        // number of processed bands: stops at the first k with start+k >= total or low(k) < 0
        for (uint32_t c = 0; c < C; c++) {
            if (F.type[c] == CRI_CH_SECONDARY) continue;
            const uint8_t* sfb = rec + HCA_REC_SF(C, c);
            int nproc = groups * bpg;
            if (nproc > total - start) nproc = total - start;
            if (nproc < 0) nproc = 0;
            // low(k) = start-1 - min(k, limit*bpg) >= 0  <=>  k <= start-1 or limit*bpg <= start-1
            if (limit * bpg > start - 1) { if (nproc > start) nproc = start; }
            float vals[2]; int idx[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int k = (int)lane + 64 * h;
                idx[h] = -1; vals[h] = 0.0f;
                if (k < nproc) {
                    const int group = k / bpg;
                    int dec = k < limit * bpg ? k : limit * bpg;
                    const int low = start - 1 - dec;
                    int sc = (int)sfb[128 - groups + group] - (int)sfb[low] + 63;
                    sc = sc & ~(sc >> 31);
                    vals[h] = HCA_SCALE_CONV[sc & 127] * X.S[c * 128 + low];
                    idx[h] = start + k;
                }
            }
            __syncthreads();
#pragma unroll
            for (int h = 0; h < 2; h++) if (idx[h] >= 0) X.S[c * 128 + idx[h]] = vals[h];
            __syncthreads();
            if (lane == 0 && start + nproc - 1 >= 0) X.S[c * 128 + start + nproc - 1] = 0.0f;
            __syncthreads();
        }
    }
    if (F.stereo_bands > 0) {
        for (uint32_t c = 0; c + 1 < C; c++) {
            if (F.type[c] != CRI_CH_PRIMARY) continue;
            const float rl = HCA_INTENSITY_RATIO[inten[(c + 1) * 8 + sf] & 15];
            const float rr = 2.0f - rl;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint32_t b = lane + 64 * h;
                if (b >= F.base_bands && b < F.total_bands) {
                    float l = X.S[c * 128 + b];
                    X.S[c * 128 + b] = l * rl;
                    X.S[(c + 1) * 128 + b] = l * rr;
                }
            }
        }
        __syncthreads();
    }
}

// intensity indexes of frame f with the "nibble 15 keeps intensity[1..7]" rule resolved (hca.cpp:1367-1375):
// a flagged frame takes entries 1..7 from the nearest earlier unflagged frame of the stream (zeros if none).
__device__ __forceinline__ void resolve_intensity(const HcaFormat& F, const uint8_t* rec_stream0, uint32_t f, uint32_t C, uint32_t lane, uint8_t* inten) {
    if (lane < C * 8) {
        const uint32_t c = lane >> 3, k = lane & 7;
        const uint8_t* rec = rec_stream0 + (uint64_t)f * F.record_bytes;
        uint8_t v = rec[HCA_REC_INT(C, c) + k];
        if (k > 0) {
            uint32_t ff = f;
            while ((((const uint32_t*)(rec_stream0 + (uint64_t)ff * F.record_bytes + HCA_REC_TAIL(C)))[2] >> c) & 1u) {
                if (ff == 0) { v = 0; break; }
                ff--;
                v = rec_stream0[(uint64_t)ff * F.record_bytes + HCA_REC_INT(C, c) + k];
            }
        }
        inten[lane] = v;
    }
}

__global__ __launch_bounds__(64) void k_hca_transform(HcaDecArgs a) {
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    const HcaFormat& F = a.formats[a.format];
    const uint32_t C = F.channels, lane = threadIdx.x, g = blockIdx.x;
    float* S = fsm; float* G = S + C * 128; float* P = G + C * 128; float* T = P + C * 128;
    uint8_t* inten = (uint8_t*)(T + 128);           // [C][8]
    const uint32_t si = find_stream(a.streams, a.stream_begin, a.stream_end, g);
    const HcaStream st = a.streams[si];
    const uint32_t f = g - st.first_frame;
    const uint8_t* rec0 = a.scratch + st.scratch_offset;
    const uint8_t* rec = rec0 + (uint64_t)f * F.record_bytes;
    const int32_t status = (int32_t)((const uint32_t*)(rec + HCA_REC_TAIL(C)))[1];
    if (status != 0) { if (lane == 0 && a.status) atomicMin(a.status + st.item, status); return; }
    if (f > 0 && (int32_t)((const uint32_t*)(rec - F.record_bytes + HCA_REC_TAIL(C)))[1] != 0) return;

    float tw_s[7], tw_c[7];
#pragma unroll
    for (int i = 0; i < 7; i++) { tw_s[i] = HCA_IMDCT_SIN[i][lane]; tw_c[i] = HCA_IMDCT_COS[i][lane]; }
    const float w0 = HCA_WINDOW[lane], w1 = HCA_WINDOW[lane + 64], w2 = HCA_WINDOW[127 - lane], w3 = HCA_WINDOW[63 - lane];
    TransformCtx X; X.F = &F; X.ath = a.ath_tables + F.ath_index * 128; X.S = S; X.G = G; X.C = C; X.lane = lane;

    // overlap tail from the previous frame's last subframe (hca.cpp:1990-1991); zeros at stream start (hca.cpp:962)
    if (f > 0) {
        const uint8_t* prec = rec - F.record_bytes;
        resolve_intensity(F, rec0, f - 1, C, lane, inten);
        frame_gains(X, prec);
        __syncthreads();
        frame_spectra(X, prec, 7, inten);
        for (uint32_t c = 0; c < C; c++) {
            imdct_dct4(S + c * 128, T, lane, tw_s, tw_c);
            const float* dct = S + c * 128;
            float p0 = w2 * dct[63 - lane], p1 = w3 * dct[lane];
            P[c * 128 + lane] = p0; P[c * 128 + 64 + lane] = p1;
        }
        __syncthreads();
    } else {
        for (uint32_t c = 0; c < C; c++) { P[c * 128 + lane] = 0.0f; P[c * 128 + 64 + lane] = 0.0f; }
    }
    resolve_intensity(F, rec0, f, C, lane, inten);
    frame_gains(X, rec);
    __syncthreads();
    int16_t* pcm = (int16_t*)(a.out + st.dst_offset);
    for (uint32_t sf = 0; sf < 8; sf++) {
        frame_spectra(X, rec, sf, inten);
        for (uint32_t c = 0; c < C; c++) {
            imdct_dct4(S + c * 128, T, lane, tw_s, tw_c);
            const float* dct = S + c * 128;
            // window + overlap-add (hca.cpp:1987-1992)
            float a0 = w0 * dct[lane + 64] + P[c * 128 + lane];
            float a1 = w1 * dct[127 - lane] - P[c * 128 + 64 + lane];
            float p0 = w2 * dct[63 - lane], p1 = w3 * dct[lane];
            P[c * 128 + lane] = p0; P[c * 128 + 64 + lane] = p1;
            // PCM16 (hca.cpp:339-360) + delay / length trim (hca.cpp:3392-3425)
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint32_t n = f * 1024 + sf * 128 + lane + 64 * h;
                if (n >= st.delay && n - st.delay < st.samples) {
                    int32_t q = cvt_trunc_x86((h ? a1 : a0) * 32768.0f);
                    q = q > 32767 ? 32767 : (q < -32768 ? -32768 : q);
                    pcm[(uint64_t)(n - st.delay) * C + c] = (int16_t)q;
                }
            }
        }
        __syncthreads();
    }
}

void launch_hca_transform(const HcaDecArgs& a, hipStream_t s) {
    if (!a.frames) return;
    size_t lds = (size_t)(3 * a.channels + 1) * 128 * 4 + a.channels * 8 + 16;
    hipLaunchKernelGGL(k_hca_transform, dim3(a.frames), dim3(64), lds, s, a);
}

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

// ------------------------------------------------------------------------------------------------------------
// HCA crypt: byte substitution + CRC rewrite, one lane per frame (hca.cpp:3322-3327)
// ------------------------------------------------------------------------------------------------------------
__global__ void k_hca_crypt(CryptArgs a) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.frames) return;
    uint32_t lo = 0, hi = a.n_streams;
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (a.first_frame[mid] <= g) lo = mid; else hi = mid; }
    const HcaStream st = a.streams[lo];
    const uint32_t fs = a.frame_sizes[lo], f = g - a.first_frame[lo];
    const uint8_t* src = a.in + st.src_offset + (uint64_t)f * fs;
    uint8_t* dst = a.out + st.dst_offset + (uint64_t)f * fs;
    const uint8_t* t = a.cipher_tables + st.cipher * 256;
    uint32_t crc = 0;
    for (uint32_t i = 0; i + 2 < fs; i++) {
        uint32_t b = t[src[i]];
        dst[i] = (uint8_t)b;
        crc = ((crc << 8) ^ CRI_CRC16_TAB[(crc >> 8) ^ b]) & 0xFFFF;
    }
    dst[fs - 2] = (uint8_t)(crc >> 8); dst[fs - 1] = (uint8_t)crc;
}
void launch_hca_crypt(const CryptArgs& a, hipStream_t s) {
    if (a.frames) hipLaunchKernelGGL(k_hca_crypt, dim3((a.frames + 63) / 64), dim3(64), 0, s, a);
}

}  // namespace cri
