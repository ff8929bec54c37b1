// cri_hca_dec.hip -- HCA decode kernels for gfx950 (MI355X, wave64).
//
// Reference functions replaced (/root/reference/CriCodecs/hca.cpp):
//   k_hca_prepare    sync word + crc16_checksum (186-211) + cipher_decrypt (491-497): the head of
//                    clHCA_DecodeBlock_unpack (1159-1169).  One wave per tile of 64 frames: coalesced loads,
//                    LDS transpose, one LANE per frame for the (serial) CRC, coalesced store of the deciphered
//                    frames as big-endian words in a lane-interleaved tile [row][64 frames].
//   k_hca_parse      the rest of clHCA_DecodeBlock_unpack (1172-1204): unpack_scalefactors (1290-1358),
//                    unpack_intensity (1361-1441), calculate_resolution (1444-1494) and the bit parse of
//                    dequantize_coefficients (1540-1571).  The variable-length parse is a serial chain per frame,
//                    so it runs one LANE per frame (64 frames per wave) on a register bit buffer fed from the tile.
//   k_hca_transform  calculate_gain (1498-1507), the float half of dequantize (1566), reconstruct_high_frequency
//                    (1638-1683), apply_intensity_stereo (1696-1714), imdct_transform (1898-2019),
//                    clHCA_ReadSamples16 (339-360) and HcaDecode's delay/trim (3401-3452).  One WAVE per frame.
// All float work is single IEEE binary32 operations in the reference's order (compiled with -ffp-contract=off).
#include <hip/hip_runtime.h>
#include "cri_kernels.h"
#include "cri_device.h"
#include "../../include/cricodecs_hip.h"

#define CRI_TABLE_QUAL static __device__ const
#include "cri_tables.h"

namespace cri {

// ------------------------------------------------------------------------------------------------------------
// k_hca_prepare
// ------------------------------------------------------------------------------------------------------------
size_t hca_prepare_lds_bytes(uint32_t chunk_rows, uint32_t n_cipher) {
    return (size_t)chunk_rows * 256 + (size_t)(n_cipher <= 16 ? n_cipher : 0) * 256 + 16;
}

// CRC-16 (poly 0x8005, MSB first) byte step without a table: T[t] = t*x^16 mod P = parity(t)*0x8003 ^ (t<<1) ^ (t<<2)
__device__ __forceinline__ uint32_t crc16_step(uint32_t crc, uint32_t b) {
    uint32_t t = (crc >> 8) ^ b;
    uint32_t tt = (t << 1) ^ (t << 2) ^ ((__builtin_popcount(t) & 1) ? 0x8003u : 0u);
    return ((crc << 8) & 0xFFFF) ^ tt;
}

__global__ __launch_bounds__(64) void k_hca_prepare(HcaDecArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const HcaFormat& F = a.formats[a.format];
    const uint32_t R = a.rows, RC = a.prep_chunk_rows, lane = threadIdx.x, tile = blockIdx.x;
    const int fs = (int)F.frame_size;
    uint32_t* rows = (uint32_t*)smem;
    uint8_t* cipher_lds = smem + (size_t)RC * 256;
    const bool cipher_in_lds = a.n_cipher <= 16;
    if (cipher_in_lds) for (uint32_t i = lane; i < a.n_cipher * 256; i += 64) cipher_lds[i] = a.cipher_tables[i];

    const uint32_t g = tile * 64 + lane;
    const bool valid = g < a.frames;
    uint32_t si = a.stream_begin, f = 0;
    if (valid) { si = find_stream(a.streams, a.stream_begin, a.stream_end, g); f = g - a.streams[si].first_frame; }
    const HcaStream st = a.streams[si];
    const uint8_t* src = a.in + st.src_offset + (uint64_t)f * (uint32_t)fs;
    const uint8_t* ct = cipher_in_lds ? cipher_lds + st.cipher * 256 : a.cipher_tables + st.cipher * 256;
    uint32_t* tb = (uint32_t*)(a.scratch + a.tile_offset) + (uint64_t)tile * (R + 1) * 64;
    uint32_t crc = 0;
    int status = 0;
    __syncthreads();
    for (uint32_t r0 = 0; r0 < R; r0 += RC) {
        const uint32_t nr = R - r0 < RC ? R - r0 : RC;
        // coalesced staging: 256 contiguous bytes of one frame per wave load
        for (uint32_t fr = 0; fr < 64; fr++) {
            const bool fv = __builtin_amdgcn_readlane((int)valid, fr) != 0;
            const uint8_t* p = (const uint8_t*)readlane64((uint64_t)src, fr);
            for (uint32_t r = lane; r < nr; r += 64) {
                const uint32_t rr = r0 + r;
                uint32_t v = 0;
                if (fv) {
                    if ((int)(4 * rr + 4) <= fs) v = ld_u32_unaligned(p + 4 * rr);
                    else for (int k = 0; 4 * (int)rr + k < fs; k++) v |= (uint32_t)p[4 * rr + k] << (8 * k);
                }
                rows[r * 64 + fr] = v;
            }
        }
        __syncthreads();
        if (valid) {
            for (uint32_t r = 0; r < nr; r++) {
                const uint32_t raw = rows[r * 64 + lane];
                uint32_t be = 0;
                int nb = fs - 4 * (int)(r0 + r); nb = nb > 4 ? 4 : nb;
                if (r0 + r == 0 && (raw & 0xFFFF) != 0xFFFF) status = CRI_ERR_HCA_FRAME(4);   // hca.cpp:1162-1164
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (k < nb) {
                        const uint32_t b = (raw >> (8 * k)) & 0xFF;
                        crc = crc16_step(crc, b);
                        const uint32_t d = cipher_in_lds ? ct[b] : __ldg(ct + b);
                        be |= d << (24 - 8 * k);
                    }
                }
                rows[r * 64 + lane] = be;
            }
        }
        __syncthreads();
        for (uint32_t r = 0; r < nr; r++) tb[(uint64_t)(r0 + r) * 64 + lane] = rows[r * 64 + lane];
        __syncthreads();
    }
    tb[(uint64_t)R * 64 + lane] = 0;
    if (valid) {
        if (status == 0 && crc != 0) status = CRI_ERR_HCA_FRAME(3);                            // hca.cpp:1166-1167
        ((int32_t*)(a.scratch + a.fstat_offset))[g] = status;
    }
}

void launch_hca_prepare(const HcaDecArgs& a, hipStream_t s) {
    if (!a.frames) return;
    hipLaunchKernelGGL(k_hca_prepare, dim3((a.frames + 63) / 64), dim3(64), hca_prepare_lds_bytes(a.prep_chunk_rows, a.n_cipher), s, a);
}

// ------------------------------------------------------------------------------------------------------------
// k_hca_parse: one lane per frame
// ------------------------------------------------------------------------------------------------------------
// LDS per wave: ostage uint32[32][65] (per-lane output words, transposed on flush), resb uint8[C*64][64]
// (two 4-bit resolutions per byte), recptr uint64[64], curve->resolution table.
size_t hca_parse_lds_bytes(uint32_t channels) { return (size_t)32 * 65 * 4 + (size_t)channels * 64 * 64 + 64 * 8 + 80 + 16; }

struct BitBuf {
    uint64_t buf;            // next bits, MSB aligned
    int avail;               // valid bits in buf
    int pos;                 // absolute bit position in the frame (hca.cpp clData.bit)
    int size;                // frame size in bits
    const uint32_t* next;    // next word of this lane in the tile (stride 64 words)
    uint32_t pw;             // prefetched word
    int rows_left;
};
__device__ __forceinline__ void bb_refill(BitBuf& b) {
    if (b.avail <= 32) {
        b.buf |= (uint64_t)b.pw << (32 - b.avail);
        b.avail += 32;
        if (b.rows_left > 0) { b.pw = *b.next; b.next += 64; b.rows_left--; } else b.pw = 0;
    }
}
// MSB-first peek of n (0..12) bits with the reference reader's end-of-frame behaviour (hca.cpp:225-281): 0 when the
// read crosses the frame end; and 0 when fewer than 24 (16) bits are left but the read spans more than 16 (8) bits
// from its byte start -- the reference then serves it from a window that is too narrow (its shift count wraps).
__device__ __forceinline__ uint32_t bb_peek(const BitBuf& b, int n) {
    const int left = b.size - b.pos;
    uint32_t v = n ? (uint32_t)(b.buf >> (64 - n)) : 0u;
    if (n > left) v = 0;
    else if (left < 24) {
        const int off = n + (b.pos & 7);
        if (off >= 17 || (off >= 9 && left < 16)) v = 0;
    }
    return v;
}
__device__ __forceinline__ void bb_skip(BitBuf& b, int n) { b.buf <<= n; b.avail -= n; b.pos += n; }
__device__ __forceinline__ uint32_t bb_read(BitBuf& b, int n) { bb_refill(b); uint32_t v = bb_peek(b, n); bb_skip(b, n); return v; }

// transposed flush of the 32 staged words of every lane: frame fr's 32 words go to recptr[fr] + byte_off, 128 B per frame
__device__ __forceinline__ void flush32(const uint32_t* ostage, const uint64_t* recptr, uint32_t lane, uint32_t byte_off, uint32_t nwords) {
    __syncthreads();
    const uint32_t w = lane & 31;
    for (uint32_t it = 0; it < 32; it++) {
        const uint32_t fr = it * 2 + (lane >> 5);
        const uint64_t rp = recptr[fr];
        if (rp && w < nwords) ((uint32_t*)(rp + byte_off))[w] = ostage[w * 65 + fr];
    }
    __syncthreads();
}

__global__ __launch_bounds__(64) void k_hca_parse(HcaDecArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const HcaFormat& F = a.formats[a.format];
    const uint32_t R = a.rows, C = F.channels, lane = threadIdx.x, tile = blockIdx.x;
    uint32_t* ostage = (uint32_t*)smem;
    uint8_t* resb = smem + 32 * 65 * 4;
    uint64_t* recptr = (uint64_t*)(resb + (size_t)C * 64 * 64);
    uint8_t* curve = (uint8_t*)(recptr + 64);
    for (uint32_t i = lane; i < 66; i += 64) curve[i] = HCA_CURVE_TO_RES[i];

    const uint32_t g = tile * 64 + lane;
    const bool valid = g < a.frames;
    uint32_t si = a.stream_begin, f = 0;
    if (valid) { si = find_stream(a.streams, a.stream_begin, a.stream_end, g); f = g - a.streams[si].first_frame; }
    const HcaStream st = a.streams[si];
    uint8_t* rec = a.scratch + st.scratch_offset + (uint64_t)f * F.record_bytes;
    recptr[lane] = valid ? (uint64_t)rec : 0;
    int status = valid ? ((const int32_t*)(a.scratch + a.fstat_offset))[g] : 0;

    BitBuf bb;
    bb.buf = 0; bb.avail = 0; bb.pos = 0; bb.size = (int)F.frame_size * 8;
    bb.next = (const uint32_t*)(a.scratch + a.tile_offset) + (uint64_t)tile * (R + 1) * 64 + lane;
    bb.rows_left = (int)R + 1;
    bb.pw = *bb.next; bb.next += 64; bb.rows_left--;
    bb_refill(bb); bb_refill(bb);
    bb_skip(bb, 16);                                             // sync word, checked by k_hca_prepare
    uint32_t packed = 0, flags = 0;
    {
        uint32_t nl = bb_read(bb, 9), eb = bb_read(bb, 7);       // hca.cpp:1175-1178
        packed = (nl << 8) - eb;
    }
    const uint8_t* ath = a.ath_tables + F.ath_index * 128;
    uint8_t* sfst = (uint8_t*)ostage;
    __syncthreads();
    for (uint32_t c = 0; c < C; c++) {
        const uint32_t coded = F.coded[c], type = F.type[c], groups = F.hfr_group_count;
        uint32_t cs = coded, extra = 0;
        if (!(type == CRI_CH_SECONDARY || groups == 0 || F.version <= 0x0200)) { extra = groups; cs += extra; }
        for (uint32_t r = 0; r < 32; r++) ostage[r * 65 + lane] = 0;
        const bool live = valid && status == 0;
        uint32_t db = 0, value = 0, prev_res = 0;
        if (live) db = bb_read(bb, 3);
        if (cs > 128) { if (live) status = CRI_ERR_HCA_FRAME(5); cs = 0; }
        const uint32_t expected = (1u << db) - 1;
        for (uint32_t i = 0; i < cs; i++) {                       // hca.cpp:1310-1350, all lanes in lock step
            uint32_t v = 0;
            if (live && status == 0 && db > 0) {
                const bool direct = db >= 6 || i == 0;
                const uint32_t x = bb_read(bb, direct ? 6 : (int)db);
                if (direct) v = x;
                else if (x == expected) v = bb_read(bb, 6);
                else {
                    const int t = (int)value + ((int)x - (int)(expected >> 1));
                    if (t < 0 || t >= 64) status = CRI_ERR_HCA_FRAME(5);
                    v = (value - (expected >> 1) + x) & 0x3F;
                }
                value = v;
            }
            sfst[((i >> 2) * 65 + lane) * 4 + (i & 3)] = (uint8_t)v;
            if (i < coded) {                                      // calculate_resolution, hca.cpp:1450-1488
                uint32_t res = 0;
                if (v > 0) {
                    const int noise = (int)ath[i] + (int)((packed + i) >> 8);
                    const int cp = noise + 1 - (int)((5 * v) >> 1);
                    res = cp < 0 ? 15u : (cp <= 65 ? (uint32_t)curve[cp] : 0u);
                    res = res > F.max_res ? F.max_res : (res < F.min_res ? F.min_res : res);
                }
                if (i & 1) resb[(c * 64 + (i >> 1)) * 64 + lane] = (uint8_t)(prev_res | (res << 4));
                else if (i + 1 == coded) resb[(c * 64 + (i >> 1)) * 64 + lane] = (uint8_t)res;
                prev_res = res;
            }
        }
        // derived HFR scales of v3.0 (hca.cpp:1353-1355); the entry one past the decoded range reads as 0
        for (uint32_t i = 0; i < extra; i++) {
            const uint32_t srci = cs - i, di = 127 - i;
            const uint8_t sv = srci < cs ? sfst[((srci >> 2) * 65 + lane) * 4 + (srci & 3)] : 0;
            sfst[((di >> 2) * 65 + lane) * 4 + (di & 3)] = sv;
        }
        // unpack_intensity, hca.cpp:1361-1441
        uint32_t inten_lo = 0, inten_hi = 0;
        if (type == CRI_CH_SECONDARY) {
            if (valid && status == 0) {
                uint8_t iv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                bb_refill(bb);
                uint32_t v = bb_peek(bb, 4);
                if (F.version <= 0x0200) {
                    iv[0] = (uint8_t)v;
                    if (v < 15) { bb_skip(bb, 4); for (int k = 1; k < 8; k++) iv[k] = (uint8_t)bb_read(bb, 4); }
                    else flags |= 1u << c;                        // intensity[1..7] keep the previous frame's values
                } else if (v < 15) {
                    bb_skip(bb, 4);
                    const uint32_t dbi = bb_read(bb, 2);
                    iv[0] = (uint8_t)v;
                    if (dbi == 3) { for (int k = 1; k < 8; k++) iv[k] = (uint8_t)bb_read(bb, 4); }
                    else {
                        const uint32_t bmax = (2u << dbi) - 1, bits = dbi + 1;
                        bool bad = false;
                        for (int k = 1; k < 8 && !bad; k++) {
                            const uint32_t delta = bb_read(bb, (int)bits);
                            if (delta == bmax) v = bb_read(bb, 4);
                            else { v = (v - (bmax >> 1) + delta) & 0xFF; if (v > 15) { bad = true; break; } }
                            iv[k] = (uint8_t)v;
                        }
                        if (bad) flags |= 1u << (16 + c);          // reference returns early here; entries stay stale
                    }
                } else { bb_skip(bb, 4); for (int k = 0; k < 8; k++) iv[k] = 7; }
                inten_lo = iv[0] | (iv[1] << 8) | (iv[2] << 16) | ((uint32_t)iv[3] << 24);
                inten_hi = iv[4] | (iv[5] << 8) | (iv[6] << 16) | ((uint32_t)iv[7] << 24);
            }
        } else if (F.version <= 0x0200) {
            for (uint32_t k = 0; k < groups; k++) {
                uint32_t v = 0;
                if (valid && status == 0) v = bb_read(bb, 6);
                const uint32_t di = 128 - groups + k;
                sfst[((di >> 2) * 65 + lane) * 4 + (di & 3)] = (uint8_t)v;
            }
        }
        if (valid) { uint32_t* ip = (uint32_t*)(rec + HCA_REC_INT(C, c)); ip[0] = inten_lo; ip[1] = inten_hi; }
        flush32(ostage, recptr, lane, HCA_REC_SF(C, c), 32);
    }
    // ---- spectra: 8 subframes x C channels x coded symbols, serial per lane (hca.cpp:1194-1199, 1540-1571)
    const uint64_t maxbits_packed = 0xCBA9876544443320ull;
    for (uint32_t sf = 0; sf < 8; sf++) {
        for (uint32_t c = 0; c < C; c++) {
            const uint32_t coded = F.coded[c];
            const bool live = valid && status == 0;
            uint32_t word = 0, rb = 0;
            for (uint32_t i = 0; i < coded; i++) {
                if (!(i & 1)) rb = resb[(c * 64 + (i >> 1)) * 64 + lane];
                int val = 0;
                if (live) {
                    const uint32_t res = (i & 1) ? (rb >> 4) : (rb & 15);
                    const int bits = (int)((maxbits_packed >> (res * 4)) & 15);
                    bb_refill(bb);
                    const uint32_t code = bb_peek(bb, bits);
                    int len;
                    if (res > 7) {                                 // sign-magnitude, zero gives one bit back
                        const int mag = (int)(code >> 1);
                        val = (code & 1) ? -mag : mag;
                        len = bits - (mag == 0 ? 1 : 0);
                    } else {                                       // truncated-binary prefix code over 0,+1,-1,...,+res,-res
                        const uint32_t nshort = (1u << bits) - (2 * res + 1);
                        uint32_t sym;
                        if (code < 2 * nshort) { sym = code >> 1; len = bits - 1; } else { sym = code - nshort; len = bits; }
                        if (bits == 0) len = 0;
                        val = (sym & 1) ? (int)((sym + 1) >> 1) : -(int)(sym >> 1);
                    }
                    bb_skip(bb, len);
                }
                if (i & 1) { word |= (uint32_t)(uint16_t)(int16_t)val << 16; ostage[((i >> 1) & 31) * 65 + lane] = word; }
                else word = (uint32_t)(uint16_t)(int16_t)val;
                if ((i & 63) == 63) flush32(ostage, recptr, lane, HCA_REC_QC(C, sf, c) + (i >> 6) * 128, 32);
            }
            if (coded & 1) ostage[((coded >> 1) & 31) * 65 + lane] = word;
            if (coded & 63) flush32(ostage, recptr, lane, HCA_REC_QC(C, sf, c) + (coded >> 6) * 128, ((coded & 63) + 1) >> 1);
        }
    }
    if (valid) {
        uint32_t* tail = (uint32_t*)(rec + HCA_REC_TAIL(C));
        tail[0] = packed; tail[1] = (uint32_t)status; tail[2] = flags; tail[3] = (uint32_t)bb.pos;
    }
}

void launch_hca_parse(const HcaDecArgs& a, hipStream_t s) {
    if (!a.frames) return;
    hipLaunchKernelGGL(k_hca_parse, dim3((a.frames + 63) / 64), dim3(64), hca_parse_lds_bytes(a.channels), s, a);
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
    const int32_t status = (int32_t)((const uint32_t*)(rec + HCA_REC_TAIL(C)))[1];     // sync / CRC / unpack result of this frame
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

}  // namespace cri
