// Developer probe (GPU box): latency of a dependent VALU chain in a lone wave per SIMD -- what bounds the serial ADX recurrences.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/chain_probe.hip -o /tmp/chain_probe && /tmp/chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ __launch_bounds__(64) void k(int32_t* out, int32_t c0, int32_t c1, int n) {
    int32_t v1 = threadIdx.x, v2 = 3, pre = 12345;
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k2 = 0; k2 < 32; k2++) {
            int32_t v;
            if (MODE == 0) { v = (__mul24(c0, v1) + pre) >> 12; v = v > 32767 ? 32767 : (v < -32768 ? -32768 : v); pre = ((k2 * 77 + (__mul24(c1, v1) >> 12)) << 12); }
            else if (MODE == 1) { v = v1 + pre; v = v > 32767 ? 32767 : (v < -32768 ? -32768 : v); v = v >> 1; }       // add, med3, shift: no multiply
            else { v = v1 + c0; }                                                                                  // one dependent add
            v2 = v1; v1 = v;
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = v1 + v2;
}
int main() {
    int32_t* d; hipMalloc(&d, 4096 * 64 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int n = 4000;
    for (int blocks : {1, 256, 1024, 4096}) {
        for (int mode = 0; mode < 3; mode++) {
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, d, 7400, -3342, n);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, d, 7400, -3342, n);
                else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, d, 7400, -3342, n);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (rep) printf("blocks %5d mode %d: %.3f ms -> %.2f ns per step (32*%d steps)\n", blocks, mode, ms, ms * 1e6 / (32.0 * n), n);
            }
        }
    }
    return 0;
}
