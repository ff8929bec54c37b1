// cri_misc.hip -- header scatter, status fill and the HCA crypt kernel (gfx950).
//   k_hca_crypt      HcaCrypt frame loop /root/reference/CriCodecs/hca.cpp:3322-3327.
#include <hip/hip_runtime.h>
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
void launch_hca_crypt(const CryptArgs& a, hipStream_t s) {
    if (a.frames) hipLaunchKernelGGL(k_hca_crypt, dim3((a.frames + 63) / 64), dim3(64), 0, s, a);
}

}  // namespace cri
