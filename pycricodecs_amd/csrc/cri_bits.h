// cri_bits.h -- integer pieces of the HCA decode kernels that are plain C++: the table-free CRC-16 steps of the frame intake and the
// jump-ahead of the v3.0 noise generator.  Shared by cri_hca_dec.hip and by the host-side enumeration of the same functions
// (tests/shim/device_fn_host.cpp, tests/test_device_functions_on_host.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cri {

// CRC-16 (poly 0x8005, init 0, MSB first; hca.cpp:186-211) without a table and 32 message bits per step.  The frame is valid
// iff its polynomial M(x) (CRC field included) is divisible by P = x^16 + x^15 + x^2 + 1 = (x + 1)(x^15 + x + 1):
//   modulo x + 1          the remainder is the parity of M: one xor per word, one popcount at the end;
//   modulo Q = x^15+x+1   x^15 = x + 1, hence x^32 = (x^15)^2 x^2 = (x^2 + 1) x^2 = x^4 + x^2: appending a word W to a running
//                         value R gives T = W ^ R<<4 ^ R<<2, and one fold of T's bits from 15 up (T>>15 times x + 1) brings it
//                         back under 18 bits.  R is only reduced completely at the end.
// (the checksum is over the bytes as stored, before the decipher.)
__device__ __forceinline__ uint32_t crcq_fold(uint32_t t) { const uint32_t h = t >> 15; return (t & 0x7FFFu) ^ h ^ (h << 1); }
__device__ __forceinline__ uint32_t crcq_word(uint32_t r, uint32_t w_be) { return crcq_fold(w_be ^ (r << 4) ^ (r << 2)); }
__device__ __forceinline__ uint32_t crcq_byte(uint32_t r, uint32_t b) { return crcq_fold((r << 8) ^ b); }   // x^8 needs no reduction: r < 2^18

// n steps of the decoder's generator r' = 0x343FD r + 0x269EC3 (hca.cpp:1616) in O(log n): affine maps composed by squaring
__device__ __forceinline__ uint32_t lcg_jump(uint32_t r, uint32_t n) {
    uint32_t am = 0x343FDu, ac = 0x269EC3u, rm = 1, rc = 0;
    while (n) {
        if (n & 1) { rm *= am; rc = rc * am + ac; }
        ac = (am + 1) * ac; am *= am; n >>= 1;
    }
    return rm * r + rc;
}

}  // namespace cri
