// Probe of the cross-lane primitives used by the IMDCT kernel: which lane does lane i read from?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL, int BANK> __device__ int dpp(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xF, BANK, false); }
__global__ void k(int* out) {
    int l = threadIdx.x;
    out[0 * 64 + l] = dpp<0xB1, 0xF>(-1, l);            // quad_perm [1,0,3,2]  -> expect l^1
    out[1 * 64 + l] = dpp<0x4E, 0xF>(-1, l);            // quad_perm [2,3,0,1]  -> expect l^2
    out[2 * 64 + l] = dpp<0x128, 0xF>(-1, l);           // row_ror:8            -> expect l^8
    int t = dpp<0x104, 0x5>(-1, l);                     // row_shl:4, banks 0,2
    out[3 * 64 + l] = dpp<0x114, 0xA>(t, l);            // row_shr:4, banks 1,3 -> expect l^4
    out[4 * 64 + l] = __builtin_amdgcn_ds_swizzle(l, 0x101F);   // xor 4 via swizzle
    out[5 * 64 + l] = __shfl_xor(l, 16);
    out[6 * 64 + l] = __shfl_xor(l, 32);
}
int main() {
    int* d; hipMalloc(&d, 7 * 64 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[7 * 64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const int want[7] = {1, 2, 8, 4, 4, 16, 32};
    for (int r = 0; r < 7; r++) {
        int ok = 1; for (int l = 0; l < 64; l++) ok &= h[r * 64 + l] == (l ^ want[r]);
        printf("probe %d (xor %d): %s :", r, want[r], ok ? "OK" : "MISMATCH");
        if (!ok) for (int l = 0; l < 16; l++) printf(" %d", h[r * 64 + l]);
        printf("\n");
    }
    return 0;
}
