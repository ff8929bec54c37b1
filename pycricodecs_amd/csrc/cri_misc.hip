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
