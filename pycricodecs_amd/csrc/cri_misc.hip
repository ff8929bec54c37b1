// cri_misc.hip -- header scatter, status fill and the HCA crypt kernel (gfx950).
//   k_hca_crypt      HcaCrypt frame loop /root/reference/CriCodecs/hca.cpp:3322-3327.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "cri_kernels.h"
#include "cri_device.h"
#include "../../include/cricodecs_hip.h"

#define CRI_TABLE_QUAL static __device__ const
#include "cri_tables.h"

namespace cri {
__global__ void k_fill_i32(int32_t* p, int32_t v, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
void launch_fill_i32(int32_t* p, int32_t v, uint32_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_fill_i32, dim3((n + 255) / 256), dim3(256), 0, s, p, v, n);
}

// test-only (called from the -DCRI_TESTING planner): a launch the runtime must refuse (more LDS than a compute unit has)
void launch_fill_i32_bad(int32_t* p, hipStream_t s) { hipLaunchKernelGGL(k_fill_i32, dim3(1), dim3(64), 1u << 20, s, p, 0, 0u); }

// Host memory -> HBM by the compute units: a few workgroups pull page-locked host memory across the link with 16-byte loads
// (source-aligned; the head and tail bytes singly).  Used by the pipelined host path for the uploads, so that the downloads have
// the DMA engines to themselves (cri_capi.cpp, run_host_core).
__global__ void __launch_bounds__(256) k_pull_host(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint64_t bytes) {
    const uint64_t head = bytes < 16 ? bytes : ((16 - ((uintptr_t)src & 15)) & 15);
    const uint64_t words = (bytes - head) >> 4, tail0 = head + (words << 4);
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (uint64_t)gridDim.x * blockDim.x;
    const uint4* s16 = (const uint4*)(src + head);
    for (uint64_t w = tid; w < words; w += nthr) {
        const uint4 v = s16[w];
        uint8_t* d = dst + head + (w << 4);
        if (((uintptr_t)d & 3) == 0) { uint32_t* d4 = (uint32_t*)d; d4[0] = v.x; d4[1] = v.y; d4[2] = v.z; d4[3] = v.w; }
        else { const uint32_t q[4] = {v.x, v.y, v.z, v.w}; for (int k = 0; k < 16; k++) d[k] = (uint8_t)(q[k >> 2] >> (8 * (k & 3))); }
    }
    if (tid < head) dst[tid] = src[tid];
    if (tid < bytes - tail0) dst[tail0 + tid] = src[tail0 + tid];
}
void launch_pull_host(uint8_t* dst, const uint8_t* src, uint64_t bytes, hipStream_t s, uint32_t max_wg) {
    if (!bytes) return;
    uint64_t wg = (bytes / 16 + 255) / 256 / 8;
    wg = wg < 1 ? 1 : (wg > max_wg ? max_wg : wg);    // 8 workgroups: 19 GB/s, and the downloads beside it keep their rate (cri_capi.cpp, run_host_core)
    hipLaunchKernelGGL(k_pull_host, dim3((uint32_t)wg), dim3(256), 0, s, dst, src, bytes);
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
// @SFA chunk payloads <-> contiguous audio streams (usm.py:263-277 reader, 313-322 / 1290-1300 AudioMask, 584-716 chunking):
// one wave per segment; bytes [mask_begin, mask_end) of a segment are XORed with mask[(j - mask_begin) % 32]
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_usm_segments(SegmentArgs a) {
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = blockIdx.x; i < a.n; i += gridDim.x) {
        const Segment g = a.segs[i];
        const uint8_t* src = a.in + g.src;
        uint8_t* dst = a.out + g.dst;
        if ((((uintptr_t)src | (uintptr_t)dst) & 3) == 0 && (g.mask_begin & 3) == 0) {
            const uint32_t nd = g.len >> 2;
            for (uint32_t d = lane; d < nd; d += 64) {
                uint32_t w = ((const uint32_t*)src)[d];
                const uint32_t j = d * 4;
                if (j + 4 > g.mask_begin && j < g.mask_end) {
                    uint32_t m = a.mask[((j - g.mask_begin) >> 2) & 7];
                    if (j + 4 > g.mask_end) m &= 0xFFFFFFFFu >> (8 * (j + 4 - g.mask_end));   // the mask stops inside this word
                    w ^= j >= g.mask_begin ? m : 0u;
                }
                ((uint32_t*)dst)[d] = w;
            }
            for (uint32_t j = (nd << 2) + lane; j < g.len; j += 64) {
                uint8_t b = src[j];
                if (j >= g.mask_begin && j < g.mask_end) b ^= (uint8_t)(a.mask[((j - g.mask_begin) >> 2) & 7] >> (8 * ((j - g.mask_begin) & 3)));
                dst[j] = b;
            }
        } else {
            for (uint32_t j = lane; j < g.len; j += 64) {
                uint8_t b = src[j];
                if (j >= g.mask_begin && j < g.mask_end) b ^= (uint8_t)(a.mask[((j - g.mask_begin) >> 2) & 7] >> (8 * ((j - g.mask_begin) & 3)));
                dst[j] = b;
            }
        }
    }
}
void launch_segments(const SegmentArgs& a, hipStream_t s) {
    if (!a.n) return;
    hipLaunchKernelGGL(k_usm_segments, dim3(a.n < (1u << 20) ? a.n : (1u << 20)), dim3(64), 0, s, a);
}

// ------------------------------------------------------------------------------------------------------------
// PCM::Get_PCM16 for WAV data that is not already 16-bit (pcm.cpp:455-545): one thread per sample, int16 into scratch
// ------------------------------------------------------------------------------------------------------------
__global__ void k_pcm_convert(ConvertArgs a) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.total) return;
    uint32_t lo = 0, hi = a.n_items;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (a.items[mid].first <= g) lo = mid; else hi = mid; }
    const ConvertItem it = a.items[lo];
    const uint64_t i = g - it.first;
    const uint8_t* p = a.in + it.src_offset + i * it.sample_size;
    int32_t v;
    if (it.bitdepth <= 8) v = ((int32_t)p[0] - (1 << (it.bitdepth - 1))) << 8;
    else if (it.mode == 3) {
        if (it.bitdepth == 32) {
            float f = __uint_as_float((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24)) * 32767.0f;
            v = (f >= -2147483648.0f && f < 2147483648.0f) ? (int32_t)f : (int32_t)0x80000000;
        } else {
            uint64_t u = 0;
            for (int k = 0; k < 8; k++) u |= (uint64_t)p[k] << (8 * k);
            double d = __longlong_as_double((long long)u) * 32767.0;
            v = (d >= -2147483648.0 && d < 2147483648.0) ? (int32_t)d : (int32_t)0x80000000;
        }
        v = v > 32767 ? 32767 : (v < -32768 ? -32768 : v);
    } else if (it.sample_size == 4) v = ((int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24))) >> (it.bitdepth - 16);
    else {
        v = (int32_t)(p[0] | (p[1] << 8) | (p[2] << 16));
        if (v & 0x800000) v |= (int32_t)0xFF000000;
        v >>= (it.bitdepth - 16);
    }
    ((int16_t*)(a.scratch + it.dst_offset))[i] = (int16_t)(v & 0xFFFF);
}
void launch_pcm_convert(const ConvertArgs& a, hipStream_t s) {
    if (a.total) hipLaunchKernelGGL(k_pcm_convert, dim3((uint32_t)((a.total + 255) / 256)), dim3(256), 0, s, a);
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
// The same work with one WAVE per frame (the kernel above reads and writes single bytes 682 B apart across the lanes):
// coalesced dword loads into an LDS image of the frame, byte substitution through an LDS copy of the table, CRC16 as one
// contiguous chunk per lane + a six-step combine  crc(A || B) = crc(A) * x^(8 |B|) + crc(B)  (table-free carry-less
// multiply modulo x^16 + x^15 + x^2 + 1), coalesced dword stores.  Used whenever the frame fits the LDS image.
__device__ __forceinline__ uint32_t crc16_step_tf(uint32_t crc, uint32_t b) {
    const uint32_t t = (crc >> 8) ^ b;
    return ((crc << 8) & 0xFFFF) ^ (t << 1) ^ (t << 2) ^ ((__builtin_popcount(t) & 1) ? 0x8003u : 0u);
}
__device__ __forceinline__ uint32_t gf16_mulmod(uint32_t a, uint32_t b) {      // a * b mod P over GF(2), 16-bit operands
    uint32_t r = 0;
#pragma unroll
    for (int bit = 15; bit >= 0; bit--) {
        r = ((r << 1) ^ ((r & 0x8000u) ? 0x8005u : 0u)) & 0xFFFFu;
        r ^= (0u - ((b >> bit) & 1u)) & a;
    }
    return r;
}
__global__ __launch_bounds__(64) void k_hca_crypt_wpf(CryptArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t g = blockIdx.x, lane = threadIdx.x;
    uint32_t lo = 0, hi = a.n_streams;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (a.first_frame[mid] <= g) lo = mid; else hi = mid; }
    const HcaStream st = a.streams[lo];
    const uint32_t fs = a.frame_sizes[lo], f = g - a.first_frame[lo];
    const uint8_t* src = a.in + st.src_offset + (uint64_t)f * fs;
    uint8_t* dst = a.out + st.dst_offset + (uint64_t)f * fs;
    uint8_t* lut = smem; uint8_t* img = smem + 256;
    const uint4 cw0 = ((const uint4*)(a.crc_pos + a.crc_pos_off[lo] + lane * 16))[0], cw1 = ((const uint4*)(a.crc_pos + a.crc_pos_off[lo] + lane * 16))[1];
    ((uint32_t*)lut)[lane] = ((const uint32_t*)(a.cipher_tables + st.cipher * 256))[lane];
    wave_lds_sync();
    const uint32_t n = fs - 2;                                  // bytes under the checksum (hca.cpp:3322-3331); fs >= 8
    for (uint32_t d = lane; 4 * d < fs; d += 64) {
        uint32_t v = 0;
        if (4 * d + 4 <= fs) v = ld_u32_unaligned(src + 4 * d);
        else for (uint32_t k = 0; 4 * d + k < fs; k++) v |= (uint32_t)src[4 * d + k] << (8 * k);
        uint32_t o = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            const uint32_t b = (v >> (8 * k)) & 0xFF;
            o |= (4 * d + k < n ? (uint32_t)lut[b] : b) << (8 * k);
        }
        ((uint32_t*)img)[d] = o;
    }
    wave_lds_sync();
    const uint32_t m = (n + 63) / 64, pad = 64 * m - n;         // front-padded with zero bytes (they do not change a zero-init CRC)
    uint32_t crc = 0;
    for (uint32_t k = 0; k < m; k++) {
        const uint32_t j = lane * m + k;
        crc = crc16_step_tf(crc, j >= pad ? (uint32_t)img[j - pad] : 0u);
    }
    // lane l's chunk stands 8*m*(63 - l) bits above the end of the message: multiply its remainder by x^that (mod P; the job's
    // table for this frame size holds x^bit * x^(8*m*(63 - l)), a 32-byte row per lane) and xor the 64 products together
    {
        const uint32_t wr[8] = {cw0.x, cw0.y, cw0.z, cw0.w, cw1.x, cw1.y, cw1.z, cw1.w};
        uint32_t acc = 0;
#pragma unroll
        for (uint32_t bit = 0; bit < 16; bit++) acc ^= (0u - ((crc >> bit) & 1u)) & (wr[bit >> 1] >> (16 * (bit & 1)));
        acc &= 0xFFFFu;
        acc ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0x111, 0xF, 0xF, true);          // a scan's pattern, with xor
        acc ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0x112, 0xF, 0xF, true);
        acc ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0x114, 0xF, 0xF, true);
        acc ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0x118, 0xF, 0xF, true);
        acc ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0x142, 0xA, 0xF, false);
        acc ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0x143, 0xC, 0xF, false);
        crc = (uint32_t)__builtin_amdgcn_readlane((int)acc, 63);
    }
    if (lane == 0) { img[fs - 2] = (uint8_t)(crc >> 8); img[fs - 1] = (uint8_t)crc; }
    wave_lds_sync();
    for (uint32_t d = lane; 4 * d + 4 <= fs; d += 64) { const uint32_t w = ((const uint32_t*)img)[d]; __builtin_memcpy(dst + 4 * d, &w, 4); }
    if (lane < (fs & 3)) { const uint32_t i = (fs & ~3u) + lane; dst[i] = img[i]; }
}
void launch_hca_crypt(const CryptArgs& a, hipStream_t s) {
    if (!a.frames) return;
    const size_t lds = 256 + (((size_t)a.max_frame_size + 3) & ~(size_t)3) + 16;
    if (a.max_frame_size >= 8 && lds <= 64 * 1024) hipLaunchKernelGGL(k_hca_crypt_wpf, dim3(a.frames), dim3(64), lds, s, a);
    else hipLaunchKernelGGL(k_hca_crypt, dim3((a.frames + 63) / 64), dim3(64), 0, s, a);
}

}  // namespace cri
