// cri_testing.hip -- TEST BUILD ONLY: linked into pycricodecs_amd/lib/libcricodecs_hip_testing.so, never into libcricodecs_hip.so.
// Kernels that hold pieces of the product kernels (shared through headers) against the reference's rule over their whole domain.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "cri_adx_quant.h"
#include "cri_hca_enc_cost.h"
#include "../../include/cricodecs_hip.h"
#define CRI_TABLE_QUAL static __device__ const
#include "cri_tables.h"

namespace cri {

// The encoder's band cost (cri_hca_enc_cost.h: classes + ranks + the clamp anomaly) against the reference's rule (hca.cpp:2771-2786:
// quantise, look the length up; |x| >= dead zone from resolution 8 on), on "bands" of eight spectra with consecutive bit patterns:
// thread t takes magnitudes first + 8 * stride * t .. + 7 (and the same negated), at every resolution 1 .. 15.
__global__ __launch_bounds__(256) void k_test_enc_band_cost(const uint8_t* tables, uint32_t first, uint32_t stride, uint32_t bands, unsigned long long* counts, uint32_t* first_bad) {
    const uint8_t* cls = tables + HCA_ET_CLS;
    const uint2* cp = (const uint2*)(tables + HCA_ET_CP);
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= bands) return;
    unsigned long long bad = 0, n = 0;
    for (int sign = 0; sign < 2; sign++) {
        float x[8]; uint32_t cl0 = 0, cl1 = 0, ntop = 0;
        for (int j = 0; j < 8; j++) {
            uint32_t m = first + 8u * stride * t + (uint32_t)j;
            m = m > HCA_ENC_CLAMP_BITS ? HCA_ENC_CLAMP_BITS : m;           // ScaleSpectra never lets more through
            x[j] = __uint_as_float(m | (sign ? 0x80000000u : 0u));
            const uint32_t k = enc_class(cls, x[j]);
            if (j < 4) cl0 |= k << (8 * j); else cl1 |= k << (8 * (j - 4));
            ntop += enc_on_clamp(x[j]);
        }
        for (int pos = 0; pos < 59; pos++) {
            const int r = HCA_ENC_CURVE_TO_RES[pos];
            if (pos && HCA_ENC_CURVE_TO_RES[pos - 1] == r) continue;
            int want = 0;
            if (r >= 8) {
                const int bits = r - 3 - 1;
                for (int j = 0; j < 8; j++) want += bits + (fabsf(x[j]) >= HCA_ENC_DEAD_ZONE[r] ? 1 : 0);
            } else {
                const float inv = HCA_ENC_INV_STEP[r], up = inv + 1;
                const int down = (int)((double)inv + 0.5 - 8);
                for (int j = 0; j < 8; j++) { const int q = (int)(x[j] * inv + up) - down; want += HCA_ENC_CODE_LEN[r][q & 15]; }
            }
            const int got = enc_band_cost(cp[pos], cl0, cl1, ntop, ntop != 0);
            n++;
            if (got != want) { bad++; if (atomicCAS(&first_bad[0], 0u, 1u) == 0u) { first_bad[1] = __float_as_uint(x[0]); first_bad[2] = (uint32_t)r; first_bad[3] = (uint32_t)got; first_bad[4] = (uint32_t)want; } }
        }
    }
    atomicAdd(&counts[0], n);
    if (bad) atomicAdd(&counts[1], bad);
}

// Every (delta, scale) of one bit depth: the float quantisers of the ADX encoders (cri_adx_quant.h) against adx.cpp:256-261.
// scale index b: 1 .. 4096 are the scales modes 2 / 3 can write (adx.cpp:236-238), 4097 stands for 8192 (mode 4's largest power).
// FORM 0: AdxQuantSmall (k_adx_encode, bit depths 2 .. 8); FORM 1: AdxQuantLane (k_adx_lane_encode, bit depth 4)
template <int FORM>
__global__ __launch_bounds__(256) void k_test_adx_quant(int bitdepth, int d_min, int d_max, unsigned long long* counts, int32_t* first) {
    const int32_t scale = blockIdx.x + 1 <= 4096 ? (int32_t)blockIdx.x + 1 : 8192;
    const int32_t limit = (1 << (bitdepth - 1)) - 1;
    const AdxQuantSmall qs((uint32_t)scale, limit);
    const AdxQuantLane ql((uint32_t)scale);
    unsigned long long bad = 0, n = 0;
    for (int64_t d = (int64_t)d_min + threadIdx.x; d <= d_max; d += blockDim.x) {
        const int32_t want = adx_quant_reference((int32_t)d, scale, limit);
        const int32_t got = FORM == 0 ? qs((int32_t)d) : ql((int32_t)d);
        n++;
        if (got != want) {
            bad++;
            if (atomicCAS(&first[0], 0, 1) == 0) { first[1] = (int32_t)d; first[2] = scale; first[3] = got; first[4] = want; }
        }
    }
    atomicAdd(&counts[0], n);
    if (bad) atomicAdd(&counts[1], bad);
}

// Counter calibration (tools/prof_r04.sh): streams of a known byte count in the access widths the product kernels use, so that
// rocprofv3's FETCH_SIZE / WRITE_SIZE can be scaled per width before they are compared with byte counts (MI355X_MICROARCH.md, HBM:
// on gfx950 FETCH_SIZE reports half the bytes of 16-byte-per-lane streaming reads; other widths are "uncalibrated").
template <typename T> __global__ __launch_bounds__(256) void k_test_stream_read(const T* src, uint64_t n, uint32_t* sink) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const T v = src[i];
        const uint32_t* w = (const uint32_t*)&v;
        for (uint32_t k = 0; k < sizeof(T) / 4; k++) acc ^= w[k];
    }
    if (acc == 0x9E3779B9u) *sink = acc;                   // (keeps the loads)
}
template <typename T> __global__ __launch_bounds__(256) void k_test_stream_write(T* dst, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        T v; uint32_t* w = (uint32_t*)&v;
        for (uint32_t k = 0; k < sizeof(T) / 4; k++) w[k] = (uint32_t)i + k;
        dst[i] = v;
    }
}

}  // namespace cri

// reads (and, separately, writes) `bytes` of device memory `reps` times with 4-, 8- and 16-byte accesses per lane; the kernels are
// k_test_stream_read<unsigned int | uint2 | uint4> and k_test_stream_write<...> in the profiler's output
extern "C" int cri_test_stream(uint64_t bytes, int reps) {
    void* buf = nullptr; uint32_t* sink = nullptr;
    bytes &= ~(uint64_t)255;
    if (!bytes || hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return CRI_ERR_HIP;
    (void)hipMemset(buf, 1, bytes);
    const dim3 grid(256 * 32), block(256);
    for (int r = 0; r < reps; r++) {
        hipLaunchKernelGGL(cri::k_test_stream_read<uint32_t>, grid, block, 0, 0, (const uint32_t*)buf, bytes / 4, sink);
        hipLaunchKernelGGL(cri::k_test_stream_read<uint2>, grid, block, 0, 0, (const uint2*)buf, bytes / 8, sink);
        hipLaunchKernelGGL(cri::k_test_stream_read<uint4>, grid, block, 0, 0, (const uint4*)buf, bytes / 16, sink);
        hipLaunchKernelGGL(cri::k_test_stream_write<uint32_t>, grid, block, 0, 0, (uint32_t*)buf, bytes / 4);
        hipLaunchKernelGGL(cri::k_test_stream_write<uint2>, grid, block, 0, 0, (uint2*)buf, bytes / 8);
        hipLaunchKernelGGL(cri::k_test_stream_write<uint4>, grid, block, 0, 0, (uint4*)buf, bytes / 16);
    }
    const int rc = hipDeviceSynchronize() == hipSuccess ? 0 : CRI_ERR_HIP;
    (void)hipFree(buf); (void)hipFree(sink);
    return rc;
}

// form 0 / 1 as above; returns 0 and fills cases, mismatches and (when there is one) the first mismatch {delta, scale, got, want}
extern "C" int cri_test_adx_quantisers(int form, int bitdepth, int d_min, int d_max, unsigned long long* cases, unsigned long long* mismatches, int32_t first4[4]) {
    if (form < 0 || form > 1 || bitdepth < 2 || bitdepth > 8 || d_min > d_max || !cases || !mismatches || !first4) return CRI_ERR_INVALID_ARG;
    unsigned long long* d_counts = nullptr; int32_t* d_first = nullptr;
    if (hipMalloc(&d_counts, 16) != hipSuccess || hipMalloc(&d_first, 32) != hipSuccess) return CRI_ERR_HIP;
    (void)hipMemset(d_counts, 0, 16); (void)hipMemset(d_first, 0, 32);
    if (form == 0) hipLaunchKernelGGL(cri::k_test_adx_quant<0>, dim3(4097), dim3(256), 0, 0, bitdepth, d_min, d_max, d_counts, d_first);
    else hipLaunchKernelGGL(cri::k_test_adx_quant<1>, dim3(4097), dim3(256), 0, 0, bitdepth, d_min, d_max, d_counts, d_first);
    unsigned long long h[2] = {0, 0}; int32_t f[8] = {0};
    int rc = 0;
    if (hipMemcpy(h, d_counts, 16, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(f, d_first, 32, hipMemcpyDeviceToHost) != hipSuccess) rc = CRI_ERR_HIP;
    (void)hipFree(d_counts); (void)hipFree(d_first);
    *cases = h[0]; *mismatches = h[1];
    for (int k = 0; k < 4; k++) first4[k] = f[1 + k];
    return rc;
}

// tables = the HCA_ET_* blob (host memory, as hca_enc_build_tables makes it -- the test takes it from a job's planner through the
// same code path: cri_test_enc_tables below); bands of 8 consecutive magnitudes starting at first + 8 * stride * k, k < bands
extern "C" int cri_test_enc_band_cost(const uint8_t* tables_host, uint32_t first, uint32_t stride, uint32_t bands, unsigned long long* cases, unsigned long long* mismatches, uint32_t first4[4]) {
    if (!tables_host || !bands || !cases || !mismatches || !first4) return CRI_ERR_INVALID_ARG;
    uint8_t* d_tab = nullptr; unsigned long long* d_counts = nullptr; uint32_t* d_first = nullptr;
    if (hipMalloc(&d_tab, HCA_ET_BYTES) != hipSuccess || hipMalloc(&d_counts, 16) != hipSuccess || hipMalloc(&d_first, 32) != hipSuccess) return CRI_ERR_HIP;
    (void)hipMemcpy(d_tab, tables_host, HCA_ET_BYTES, hipMemcpyHostToDevice);
    (void)hipMemset(d_counts, 0, 16); (void)hipMemset(d_first, 0, 32);
    hipLaunchKernelGGL(cri::k_test_enc_band_cost, dim3((bands + 255) / 256), dim3(256), 0, 0, d_tab, first, stride, bands, d_counts, d_first);
    unsigned long long h[2] = {0, 0}; uint32_t f[8] = {0};
    int rc = 0;
    if (hipMemcpy(h, d_counts, 16, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(f, d_first, 32, hipMemcpyDeviceToHost) != hipSuccess) rc = CRI_ERR_HIP;
    (void)hipFree(d_tab); (void)hipFree(d_counts); (void)hipFree(d_first);
    *cases = h[0]; *mismatches = h[1];
    for (int k = 0; k < 4; k++) first4[k] = f[1 + k];
    return rc;
}
