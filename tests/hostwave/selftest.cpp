// hostwave self-test (TEST INFRASTRUCTURE): the emulator's cross-lane instructions against their closed forms (gfx9 ISA: DPP controls,
// ds_swizzle bit mode, ds_bpermute, v_readlane, v_mbcnt, the __shfl family of HIP), divergent exchanges, barriers between waves, LDS
// carving, early returns.  Built and run by tests/test_hostwave.py; prints "ok" or the first mismatch.
#include <hip/hip_runtime.h>
#include <stdio.h>

static int g_bad = 0;
#define CHECK(cond, ...) do { if (!(cond)) { if (!__atomic_fetch_add(&g_bad, 1, __ATOMIC_RELAXED)) { fprintf(stderr, "selftest line %d lane %u: ", __LINE__, threadIdx.x); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } } } while (0)

__global__ void k_cross_lane(uint32_t* out) {
    const uint32_t l = threadIdx.x & 63, v = 1000 + l;
    // DPP quad permutations and row rotation: the butterflies the DCT uses
    CHECK((uint32_t)__builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true) == 1000 + (l ^ 1), "quad_perm[1,0,3,2]");
    CHECK((uint32_t)__builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true) == 1000 + (l ^ 2), "quad_perm[2,3,0,1]");
    CHECK((uint32_t)__builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, true) == 1000 + (l ^ 8), "row_ror:8");
    CHECK((uint32_t)__builtin_amdgcn_update_dpp(0, v, 0x121, 0xF, 0xF, true) == 1000 + ((l & ~15u) | ((l - 1) & 15)), "row_ror:1");
    // row shifts: out of the row -> 0 with bound_ctrl, `old` without; bank masks keep `old`
    CHECK((uint32_t)__builtin_amdgcn_update_dpp(7, v, 0x104, 0xF, 0xF, true) == ((l & 15) + 4 <= 15 ? 1000 + l + 4 : 0), "row_shl:4 bound_ctrl");
    CHECK((uint32_t)__builtin_amdgcn_update_dpp(7, v, 0x104, 0xF, 0xF, false) == ((l & 15) + 4 <= 15 ? 1000 + l + 4 : 7), "row_shl:4");
    CHECK((uint32_t)__builtin_amdgcn_update_dpp(7, v, 0x114, 0xF, 0xA, false) == (((l >> 2) & 1) ? 1000 + l - 4 : 7), "row_shr:4 bank_mask 0xA");
    {   // lane ^ 4 out of two row shifts, as cri_device.h's alternative form builds it
        const int t = __builtin_amdgcn_update_dpp(0, v, 0x104, 0xF, 0xF, true);
        CHECK((uint32_t)__builtin_amdgcn_update_dpp(t, v, 0x114, 0xF, 0xA, false) == 1000 + (l ^ 4), "lane^4 by shl/shr");
    }
    CHECK((uint32_t)__builtin_amdgcn_update_dpp(7, v, 0x111, 0x5, 0xF, false) == (((l >> 4) & 1) ? 7 : ((l & 15) >= 1 ? 1000 + l - 1 : 7)), "row_shr:1 row_mask 0x5");
    CHECK((uint32_t)__builtin_amdgcn_update_dpp(7, v, 0x138, 0xF, 0xF, false) == (l >= 1 ? 1000 + l - 1 : 7), "wave_shr:1");
    CHECK((uint32_t)__builtin_amdgcn_update_dpp(7, v, 0x130, 0xF, 0xF, true) == (l <= 62 ? 1000 + l + 1 : 0), "wave_shl:1");
    CHECK((uint32_t)__builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true) == 1000 + ((l & ~15u) | (15 - (l & 15))), "row_mirror");
    CHECK((uint32_t)__builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true) == 1000 + ((l & ~7u) | (7 - (l & 7))), "row_half_mirror");
    CHECK((uint32_t)__builtin_amdgcn_update_dpp(7, v, 0x142, 0xA, 0xF, false) == (((l >> 4) & 1) ? 1000 + (l & ~15u) - 1 : 7), "row_bcast:15 into rows 1, 3");
    CHECK((uint32_t)__builtin_amdgcn_update_dpp(7, v, 0x143, 0xC, 0xF, false) == (l >= 32 ? 1031u : 7u), "row_bcast:31 into rows 2, 3");
    // ds_swizzle: bit mode (and | or ^ xor inside 32 lanes) and quad mode; ds_bpermute; readlane
    CHECK((uint32_t)__builtin_amdgcn_ds_swizzle(v, 0x101F) == 1000 + (l ^ 4), "swizzle xor 4");
    CHECK((uint32_t)__builtin_amdgcn_ds_swizzle(v, 0x201F) == 1000 + (l ^ 8), "swizzle xor 8");
    CHECK((uint32_t)__builtin_amdgcn_ds_swizzle(v, 0x0010) == 1000 + (l & 32) + (l & 16), "swizzle and 0x10 (broadcast of lanes 0 / 16 of a half)");
    CHECK((uint32_t)__builtin_amdgcn_ds_swizzle(v, 0x8000 | 0x1B) == 1000 + ((l & ~3u) | (3 - (l & 3))), "swizzle quad [3,2,1,0]");
    CHECK((uint32_t)__builtin_amdgcn_ds_bpermute((int)(((l * 7 + 3) & 63) << 2), v) == 1000 + ((l * 7 + 3) & 63), "bpermute");
    CHECK((uint32_t)__builtin_amdgcn_readlane(v, 37) == 1037, "readlane");
    CHECK((uint32_t)__builtin_amdgcn_readfirstlane(v) == 1000, "readfirstlane");
    // ballots and counts
    const uint64_t odd = __ballot(l & 1);
    CHECK(odd == 0xAAAAAAAAAAAAAAAAull, "ballot");
    CHECK(__builtin_amdgcn_ballot_w64(l < 40) == (1ull << 40) - 1, "ballot_w64");
    CHECK(__any(l == 63) && !__any(l == 64) && __all(l < 64) && !__all(l < 63), "any / all");
    CHECK(__builtin_amdgcn_mbcnt_hi((uint32_t)(odd >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)odd, 0)) == l / 2, "mbcnt");
    // HIP's shuffles with widths
    CHECK(__shfl(v, 5) == 1005u && __shfl(v, 5, 16) == 1000 + (l & ~15u) + 5, "shfl");
    CHECK(__shfl_xor(v, 32) == 1000 + (l ^ 32) && __shfl_xor(v, 3, 4) == 1000 + (l ^ 3), "shfl_xor");
    CHECK(__shfl_up(v, 1) == (l ? 1000 + l - 1 : 1000) && __shfl_up(v, 2, 8) == ((l & 7) >= 2 ? 1000 + l - 2 : 1000 + l), "shfl_up");
    CHECK(__shfl_down(v, 1) == (l < 63 ? 1000 + l + 1 : 1063) && __shfl_down(v, 3, 16) == ((l & 15) + 3 < 16 ? 1000 + l + 3 : 1000 + l), "shfl_down");
    CHECK(__shfl(0x0123456789ABCDEFull + l, 9) == 0x0123456789ABCDEFull + 9, "64-bit shfl");
    // a divergent exchange: only the odd lanes take part; a partner that is not in the branch reads as 0 (bound_ctrl) / old
    uint32_t d = 55;
    if (l & 1) {
        d = (uint32_t)__builtin_amdgcn_update_dpp(9, v, 0xB1, 0xF, 0xF, true);                     // lane ^ 1 is even: not here
        CHECK(d == 0, "divergent DPP, bound_ctrl");
        d = (uint32_t)__builtin_amdgcn_update_dpp(9, v, 0x4E, 0xF, 0xF, false);                    // lane ^ 2 is odd: here
        CHECK(d == 1000 + (l ^ 2), "divergent DPP among the branch's lanes");
        CHECK(__ballot(true) == 0xAAAAAAAAAAAAAAAAull, "ballot inside a branch counts the branch's lanes");
        CHECK((uint32_t)__builtin_amdgcn_readfirstlane(v) == 1001, "readfirstlane inside a branch");
    }
    CHECK(__ballot(true) == ~0ull, "reconverged");
    // scalar forms
    CHECK(__builtin_amdgcn_perm(0x44332211u, 0x88776655u, 0x07060100u) == 0x44336655u, "v_perm bytes");
    CHECK(__builtin_amdgcn_perm(0x80000000u, 0x00008000u, 0x0C0D0B09u) == 0x00FFFF00u, "v_perm constants and signs");
    CHECK(__builtin_amdgcn_alignbit(0x11223344u, 0xAABBCCDDu, 8) == 0x44AABBCCu, "v_alignbit");
    CHECK(__builtin_amdgcn_ubfe(0xABCD1234u, 12, 8) == 0xD1u && __builtin_amdgcn_sbfe((int)0xABCD1234u, 12, 8) == (int)0xFFFFFFD1, "bfe");
    CHECK(__builtin_amdgcn_sad_u8(0x10FF0005u, 0x20000105u, 3) == 3 + 0x10 + 0xFF + 1 + 0, "sad_u8");
    CHECK(__mul24(0x00FFFFFF, 5) == -5 && __mul24(-3, 7) == -21, "mul24 sign-extends 24 bits");
    CHECK(__builtin_amdgcn_fmed3f(3.0f, -1.0f, 2.0f) == 2.0f, "fmed3");
    { const auto p = __builtin_amdgcn_cvt_pk_i16(70000, -70000); CHECK(p.x == 32767 && p.y == -32768, "cvt_pk_i16 saturates"); }
    out[blockIdx.x * 64 + l] = v;
}

// four waves: static + dynamic LDS, __syncthreads between producers and consumers, LDS atomics, waves that leave early
__global__ void k_block(uint32_t* out, uint32_t n) {
    __shared__ uint32_t turn[4];
    __shared__ __attribute__((aligned(16))) uint32_t table[256];
    extern __shared__ __attribute__((aligned(16))) uint8_t dyn[];
    uint32_t* sums = (uint32_t*)dyn;
    const uint32_t t = threadIdx.x, w = t >> 6;
    CHECK((uint8_t*)table != (uint8_t*)turn && ((uintptr_t)table & 15) == 0 && ((uintptr_t)dyn & 15) == 0, "LDS carving");
    CHECK((uintptr_t)dyn < (1ull << 32), "LDS addresses fit 32 bits");
    if (t < 4) { turn[t] = 0; sums[t] = 0; }
    table[t] = t * t;
    __syncthreads();
    CHECK(table[255 - t] == (255 - t) * (255 - t), "another wave's LDS writes after the barrier");
    atomicAdd(&sums[w], t);
    atomicOr(&turn[w], 1u << (t & 31));
    __syncthreads();
    if (w >= 2) return;                                          // waves 2 and 3 are gone: the barrier below is for those that remain
    CHECK(sums[w] == (w * 64 + w * 64 + 63) * 64 / 2 && turn[w] == ~0u, "LDS atomics");
    __syncthreads();
    if (t == 0) out[blockIdx.x] = sums[0] + sums[1] + sums[2] + sums[3] + n;
}

// a read one byte past the launch's dynamic LDS: the guard page reports it (run in a child process)
__global__ void k_lds_overrun(uint32_t* out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t dyn[];
    out[0] = dyn[4096 + threadIdx.x * 64];
}

int main(int argc, char** argv) {
    uint32_t* out;
    if (hipMalloc(&out, 1 << 16) != hipSuccess) return 2;
    if (argc > 1 && !strcmp(argv[1], "overrun")) {
        hipLaunchKernelGGL(k_lds_overrun, dim3(1), dim3(64), 4096, 0, out);
        printf("no fault\n");
        return 0;
    }
    hipLaunchKernelGGL(k_cross_lane, dim3(3), dim3(64), 0, 0, out);
    for (int i = 0; i < 192; i++) if (out[i] != 1000u + (i & 63)) { fprintf(stderr, "k_cross_lane did not finish\n"); return 1; }
    hipLaunchKernelGGL(k_block, dim3(37), dim3(256), 64, 0, out, 5u);
    for (int b = 0; b < 37; b++) if (out[b] != 255u * 256 / 2 + 5) { fprintf(stderr, "k_block: block %d wrote %u\n", b, out[b]); return 1; }
    if (g_bad) { fprintf(stderr, "%d checks failed\n", g_bad); return 1; }
    printf("ok\n");
    return 0;
}
