// tests/shim/device_fn_host.cpp -- TEST INFRASTRUCTURE ONLY.
// The product's float shortcuts for integer rules live in headers shared by the kernels and by the device-side exhaustive tests
// (csrc/cri_adx_quant.h, csrc/cri_hca_enc_cost.h).  They are plain C++ over IEEE binary32 with explicit fused multiply-adds where a
// fused one is meant (-ffp-contract=off everywhere else), so what they compute does not depend on who executes them: this file compiles
// THE SAME HEADERS for the host (the HIP qualifiers and the two bit-cast intrinsics defined away below; hip_runtime.h skipped by its own
// include guard) and enumerates their whole domains against the reference's integer / table rule on the CPU.  It is what the CPU suite
// has when no GPU is (round 6); the device-side enumeration (cri_testing.hip) stays the check of the compiled kernels' arithmetic.
#define HIP_INCLUDE_HIP_HIP_RUNTIME_H        /* the real header is not needed: nothing below is a HIP API call */
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <atomic>
#include <thread>
#include <vector>
#define __device__
#define __host__
#define __forceinline__ inline
static inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
#include "../../pycricodecs_amd/csrc/cri_adx_quant.h"

// every delta in [d_min, d_max] x every scale 1 .. 4096 and 8192, form 0 = AdxQuantSmall (bit depths 2 .. 8), 1 = AdxQuantLane (bit depth 4)
extern "C" int host_adx_quantisers(int form, int bitdepth, int d_min, int d_max, unsigned long long* cases, unsigned long long* mismatches, int32_t first4[4]) {
    const int32_t limit = (1 << (bitdepth - 1)) - 1;
    std::atomic<unsigned long long> n{0}, bad{0};
    std::atomic<int> have_first{0};
    unsigned nt = std::thread::hardware_concurrency(); if (nt == 0) nt = 4; if (nt > 64) nt = 64;
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++) th.emplace_back([&, t] {
        unsigned long long ln = 0, lbad = 0;
        for (int si = (int)t; si < 4097; si += (int)nt) {
            const int32_t scale = si + 1 <= 4096 ? si + 1 : 8192;
            const cri::AdxQuantSmall qs((uint32_t)scale, limit);
            const cri::AdxQuantLane ql((uint32_t)scale);
            for (int64_t d = d_min; d <= d_max; d++) {
                const int32_t want = cri::adx_quant_reference((int32_t)d, scale, limit);
                const int32_t got = form == 0 ? qs((int32_t)d) : ql((int32_t)d);
                ln++;
                if (got != want) {
                    lbad++;
                    int z = 0;
                    if (have_first.compare_exchange_strong(z, 1)) { first4[0] = (int32_t)d; first4[1] = scale; first4[2] = got; first4[3] = want; }
                }
            }
        }
        n += ln; bad += lbad;
    });
    for (auto& x : th) x.join();
    *cases = n.load(); *mismatches = bad.load();
    return 0;
}

// ---- the HCA encoder's band cost (csrc/cri_hca_enc_cost.h: classes + ranks + the clamp anomaly) against CalculateUsedBits' inner loop
// (hca.cpp:2771-2786), as cri_testing.hip's k_test_enc_band_cost does on the device: bands of eight consecutive bit patterns starting at
// first + 8 * stride * k, both signs, every resolution 1 .. 15.  tables = the HCA_ET_* blob of hca_enc_build_tables (cri_host.cpp).
struct uint2 { uint32_t x, y; };
#include "../../pycricodecs_amd/csrc/cri_hca_enc_cost.h"
#define CRI_TABLE_QUAL static const
#include "../../pycricodecs_amd/csrc/cri_tables.h"

extern "C" int host_enc_band_cost(const uint8_t* tables, uint32_t first, uint32_t stride, uint32_t bands, unsigned long long* cases, unsigned long long* mismatches, uint32_t first4[4]) {
    using namespace cri;
    const uint8_t* cls = tables + HCA_ET_CLS;
    const uint2* cp = (const uint2*)(tables + HCA_ET_CP);
    std::atomic<unsigned long long> n{0}, bad{0};
    std::atomic<int> have_first{0};
    unsigned nt = std::thread::hardware_concurrency(); if (nt == 0) nt = 4; if (nt > 64) nt = 64;
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++) th.emplace_back([&, t] {
        unsigned long long ln = 0, lbad = 0;
        for (uint64_t b = t; b < bands; b += nt) {
            for (int sign = 0; sign < 2; sign++) {
                float x[8]; uint32_t cl0 = 0, cl1 = 0, ntop = 0;
                for (int j = 0; j < 8; j++) {
                    uint32_t m = first + 8u * stride * (uint32_t)b + (uint32_t)j;
                    m = m > HCA_ENC_CLAMP_BITS ? HCA_ENC_CLAMP_BITS : m;
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
                        for (int j = 0; j < 8; j++) { volatile float m1 = x[j] * inv; volatile float s1 = m1 + up; const int q = (int)s1 - down; want += HCA_ENC_CODE_LEN[r][q & 15]; }   // two roundings, as the reference's build (no FMA)
                    }
                    const int got = enc_band_cost(cp[pos], cl0, cl1, ntop, ntop != 0);
                    ln++;
                    if (got != want) {
                        lbad++;
                        int z = 0;
                        if (have_first.compare_exchange_strong(z, 1)) { first4[0] = __float_as_uint(x[0]); first4[1] = (uint32_t)r; first4[2] = (uint32_t)got; first4[3] = (uint32_t)want; }
                    }
                }
            }
        }
        n += ln; bad += lbad;
    });
    for (auto& x : th) x.join();
    *cases = n.load(); *mismatches = bad.load();
    return 0;
}

// ---- cri_bits.h: the frame intake's table-free CRC-16 and the noise generator's jump-ahead
#include "../../pycricodecs_amd/csrc/cri_bits.h"

// the parse's acceptance rule (cri_hca_dec.hip, feed_land + the check behind the last block): words of the frame big-endian through
// crcq_word, a tail of single bytes through crcq_byte, parity of everything; valid iff the folded remainder is 0 and the parity even.
// Returns 1 = accepted, 0 = rejected.
extern "C" int host_crc_accepts(const uint8_t* msg, size_t len) {
    uint32_t r = 0, par = 0;
    size_t i = 0;
    for (; i + 4 <= len; i += 4) {
        const uint32_t w_be = (uint32_t)msg[i] << 24 | (uint32_t)msg[i + 1] << 16 | (uint32_t)msg[i + 2] << 8 | msg[i + 3];
        r = cri::crcq_word(r, w_be);
        par ^= w_be;
    }
    for (; i < len; i++) { r = cri::crcq_byte(r, msg[i]); par ^= msg[i]; }
    return cri::crcq_fold(r) == 0 && (__builtin_popcount(par) & 1) == 0;
}

// lcg_jump(r, n) against n single steps of hca.cpp:1616, for n = 0 .. n_max from `seeds` start values; and the group law on large jumps
extern "C" unsigned long long host_lcg_jump_mismatches(uint32_t seeds, uint32_t n_max) {
    unsigned long long bad = 0;
    for (uint32_t s = 0; s < seeds; s++) {
        uint32_t r0 = s * 2654435761u + 1u, r = r0;
        for (uint32_t n = 0; n <= n_max; n++) {
            if (cri::lcg_jump(r0, n) != r) bad++;
            r = r * 0x343FDu + 0x269EC3u;
        }
        const uint32_t a = s * 40503u + 12345u, b = 0xFFFFFFFFu - s * 977u;          // (a + b wraps: the generator's period divides 2^32)
        if (cri::lcg_jump(cri::lcg_jump(r0, a), b) != cri::lcg_jump(r0, a + b)) bad++;
    }
    return bad;
}
