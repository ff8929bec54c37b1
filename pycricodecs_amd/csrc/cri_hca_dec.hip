// cri_hca_dec.hip -- HCA decode kernels for gfx950 (MI355X, wave64).
//
// Reference functions replaced (/root/reference/CriCodecs/hca.cpp):
//   k_hca_parse      all of clHCA_DecodeBlock_unpack (1159-1204): sync word + crc16_checksum (186-211) + cipher_decrypt (491-497)
//                    as the frame's bytes enter the kernel's bit feed, then unpack_scalefactors (1290-1358), unpack_intensity
//                    (1361-1441), calculate_resolution (1444-1494) and the bit parse of dequantize_coefficients (1540-1571).
//                    The variable-length parse is a serial chain per frame, so it runs one LANE per frame (64 frames per wave)
//                    on a sliding window fed from the lane's LDS ring; results leave as small frame records plus tile-major
//                    quantised lines (cri_types.h).
//   k_hca_transform<PLAIN, C>  calculate_gain (1498-1507), the float half of dequantize (1566),
//                    reconstruct_high_frequency (1638-1683), apply_intensity_stereo (1696-1714), imdct_transform
//                    (1898-2019), clHCA_ReadSamples16 (339-360) and HcaDecode's delay/trim (3401-3452).  One WAVE per run of
//                    8 frames, four transforms at a time in registers (1, 2, 4, 6 or 8 channels).
//   k_hca_transform_plain      formats without HFR / joint stereo / noise fill, 1, 2 or 4 channels (the usual case): each 16-lane
//                    slot follows one channel through consecutive subframes, so window + overlap-add stay in the DCT's lanes.
//                    Between k_hca_parse and it (mono, stereo) a frame's quantised lines travel as int8 when its tile allows
//                    it (HCA_REC_NARROW, cri_types.h), as int16 otherwise.
//   k_hca_transform_generic    the same for any other channel layout, one wave per frame, spectra assembled in LDS
//                    (the DCT is the same register network); k_hca_noise_scan gives it the generator state each frame starts from.
// All float work is single IEEE binary32 operations in the reference's order (compiled with -ffp-contract=off).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <algorithm>
#include "cri_kernels.h"
#include "cri_device.h"
#include "cri_bits.h"
#include "../../include/cricodecs_hip.h"

#define CRI_TABLE_QUAL static __device__ const
#include "cri_tables.h"

namespace cri {

// ------------------------------------------------------------------------------------------------------------
// frame intake: sync word, checksum, decipher (the head of clHCA_DecodeBlock_unpack, hca.cpp:1159-1169)
// ------------------------------------------------------------------------------------------------------------
// There is no separate pass over the compressed frames: k_hca_parse's bit feed takes its lane's frame from the input blob 16
// bytes at a time and checksums and deciphers the four words on their way into the lane's LDS ring (feed_land), so the
// deciphered words never exist in HBM.
//
// (the checksum arithmetic -- CRC-16 without a table, 32 message bits per step -- is in cri_bits.h: crcq_word / crcq_byte / crcq_fold)

// Format parameters held in registers (scalar) for the whole kernel: reading them through the HcaFormat pointer inside the
// loops would turn every use into a memory load.
struct Fmt {
    uint32_t channels, version, frame_size, min_res, max_res, total_bands, base_bands, stereo_bands, bands_per_hfr_group,
        hfr_group_count, ath_index, record_bytes, types;                     // types: 2 bits per channel
    __device__ __forceinline__ uint32_t type(uint32_t c) const { return (types >> (2 * c)) & 3u; }
    __device__ __forceinline__ uint32_t coded(uint32_t c) const { return type(c) == CRI_CH_SECONDARY ? base_bands : base_bands + stereo_bands; }
};
__device__ __forceinline__ Fmt load_fmt(const HcaFormat* f) {
    Fmt m;
    m.channels = f->channels; m.version = f->version; m.frame_size = f->frame_size; m.min_res = f->min_res; m.max_res = f->max_res;
    m.total_bands = f->total_bands; m.base_bands = f->base_bands; m.stereo_bands = f->stereo_bands;
    m.bands_per_hfr_group = f->bands_per_hfr_group; m.hfr_group_count = f->hfr_group_count; m.ath_index = f->ath_index;
    m.record_bytes = f->record_bytes;
    uint32_t t = 0;
    for (uint32_t c = 0; c < 16; c++) t |= (uint32_t)(f->type[c] & 3) << (2 * c);
    m.types = t;
    return m;
}

__device__ __forceinline__ uint4 ld_u128_unaligned(const uint8_t* p) { uint4 v; __builtin_memcpy(&v, p, 16); return v; }

// ------------------------------------------------------------------------------------------------------------
// k_hca_parse: one lane per frame
// ------------------------------------------------------------------------------------------------------------
// LDS per wave (8.4 KB): bit-feed ring uint32[RING_WORDS][64], ostage uint32[16][66] (per-lane output words, transposed
// on flush), curve->resolution table.  The per-band code descriptions (one byte each: max bits |
// short-code count << 4, see parse_symbol) live in the tile's `resg` area of scratch: one uint4 (16 bands) per lane and
// 16-band block, written once by the scalefactor pass and re-read (coalesced, L2-resident) by each of the 8 subframes.
//
// Bit feed.  HBM latency under load is microseconds and the parse is an in-order serial chain, so words travel
//   input blob (global) --16-byte chunk requested at a checkpoint--> VGPRs --checksummed, deciphered and landed one block of
//   parsing later--> per-lane LDS ring --one word prefetched per symbol--> 64-bit shift register.
// Checkpoints sit every 16 symbols (at most 16*12 bits = 6 words consumed in between).  A checkpoint asks for as many whole
// chunks (16 bytes of the lane's frame) as the ring has room for, at most two: with h words in the ring after landing,
// r = min(8, 4*floor((16-h)/4)) are requested and c <= 6 consumed before they land, so h' = h - c + r >= 7 whenever
// h >= 7, and h' <= 16.  The lanes of a wave parse the same block of their frames at the same time, so their requests
// mostly fall on the same checkpoints and a landing step works for most of the wave at once; the second chunk is rare (a
// lane that burned more than 4 words in one block) and sits behind a wave-uniform branch, and so do the frame's last,
// partial chunk and a chunk that would reach past the end of the input blob.
#define RING_WORDS 16       // (+ 4 spare rows: the sink of a lane that lands nothing)
size_t hca_parse_lds_bytes(uint32_t n_cipher) { return (size_t)(RING_WORDS + 4) * 256 + 16 * 66 * 4 + 96 + 128 + 128 + 16 + 256 + (size_t)(n_cipher <= 16 ? n_cipher : 0) * 256; }

struct BitFeed {
    const uint8_t* next;     // next chunk of this lane's frame in the input blob
    const uint8_t* in_end;   // end of the input blob
    const uint8_t* ct;       // this lane's cipher table (LDS, or global when the job has more than 16)
    int bytes_left;          // bytes of the frame not yet requested (requests past the frame land zeros)
    uint32_t* ring;          // LDS ring base of this lane (slot stride 64 words)
    uint32_t wr;             // words landed in the ring (a multiple of 4)
    uint32_t nfl;            // chunks in flight: 0, 1 or 2
    uint32_t nb0, nb1;       // bytes of the frame in each chunk in flight (16, less for the last one, 0 past the frame)
    uint32_t r, par;         // checksum state (crcq_*)
    uint4 fl0, fl1;
};
struct BitBuf {
    uint32_t hi, lo;         // frame words k and k+1 (big-endian): a 64-bit window the reader moves through
    uint32_t off;            // bits of the window already consumed: < 32 after a refill, and a refill precedes at most 24 more
    int pos;                 // absolute bit position in the frame (hca.cpp clData.bit)
    int size;                // frame size in bits
    uint32_t rd;             // ring index of nw (= k + 2)
    uint32_t nw;             // frame word k + 2 (LDS read issued one refill earlier)
};

// big-endian word of the deciphered bytes of a raw (little-endian) word
template <bool IDENTITY, bool CT_LDS>
__device__ __forceinline__ uint32_t decipher_word(const uint8_t* ct, uint32_t raw) {
    if (IDENTITY) return __builtin_bswap32(raw);
    uint32_t be = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t b = (raw >> (8 * k)) & 0xFF;
        be |= (CT_LDS ? (uint32_t)ct[b] : (uint32_t)__ldg(ct + b)) << (24 - 8 * k);
    }
    return be;
}
// 16 bytes at p; bytes at or past `end` read as 0
__device__ __forceinline__ uint4 load_chunk_guarded(const uint8_t* p, const uint8_t* end) {
    if (p + 16 <= end) return ld_u128_unaligned(p);
    uint32_t w[4] = {0, 0, 0, 0};
    for (int i = 0; i < 16; i++) if (p + i < end) w[i >> 2] |= (uint32_t)p[i] << (8 * (i & 3));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// A checkpoint is two halves with the pending record stores in between (see PendingFlush): the chunks asked for at the
// previous checkpoint land in the ring, the stores of the last two blocks go out, the next chunks are asked for.  vmcnt
// counts loads and stores together and in order, so the wait at the next checkpoint covers those stores too -- by then
// they have had a whole block of parsing to complete.
template <bool IDENTITY, bool CT_LDS>
__device__ __forceinline__ void feed_land_chunk(BitFeed& f, const uint4& c, uint32_t k, uint32_t nbs) {
    const bool on = k < f.nfl;
    uint32_t* slot = f.ring + (on ? ((f.wr + 4 * k) & (RING_WORDS - 1)) : (uint32_t)RING_WORDS) * 64;
    const uint32_t nb = nbs & 0xFF, sh = nbs >> 8;             // sh: feed_issue_wave loaded this chunk from sh bytes before its place
    uint32_t w[4] = {c.x, c.y, c.z, c.w};
    if (__any(on && sh != 0)) {                                // (only the last chunks of the input blob)
        const uint64_t lo = ((uint64_t)w[1] << 32) | w[0], hi = ((uint64_t)w[3] << 32) | w[2];
        const uint32_t bs = 8 * sh;                            // 0 .. 128 bits down; what comes in from past the blob's end is 0
        uint64_t nlo, nhi;
        if (bs == 0) { nlo = lo; nhi = hi; }
        else if (bs < 64) { nlo = (lo >> bs) | (hi << (64 - bs)); nhi = hi >> bs; }
        else if (bs < 128) { nlo = hi >> (bs - 64); nhi = 0; }
        else { nlo = 0; nhi = 0; }
        w[0] = (uint32_t)nlo; w[1] = (uint32_t)(nlo >> 32); w[2] = (uint32_t)nhi; w[3] = (uint32_t)(nhi >> 32);
    }
    uint32_t o[4] = {0, 0, 0, 0};
    if (__all(!on || nb == 16)) {                              // whole chunks (the usual case): 32 bits per checksum step
        if (on) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                f.r = crcq_word(f.r, __builtin_bswap32(w[j]));
                f.par ^= w[j];
                o[j] = decipher_word<IDENTITY, CT_LDS>(f.ct, w[j]);
            }
        }
    } else if (on) {                                           // a frame's last chunk, or nothing of the frame at all: byte by byte
        for (uint32_t i = 0; i < nb; i++) {
            const uint32_t b = (w[i >> 2] >> (8 * (i & 3))) & 0xFF;
            f.r = crcq_byte(f.r, b);
            f.par ^= b;
            const uint32_t d = IDENTITY ? b : (CT_LDS ? (uint32_t)f.ct[b] : (uint32_t)__ldg(f.ct + b));
            o[i >> 2] |= d << (24 - 8 * (i & 3));
        }
    }
    slot[0] = o[0]; slot[64] = o[1]; slot[128] = o[2]; slot[192] = o[3];
}
// (c0, c1: the registers the chunks in flight were loaded into -- BitFeed's own pair, or a pair that only lives inside one
//  iteration of the spectra loop, see there)
template <bool IDENTITY, bool CT_LDS>
__device__ __forceinline__ void feed_land(BitFeed& f, const uint4& c0, const uint4& c1) {
    if (__any(f.nfl > 0)) feed_land_chunk<IDENTITY, CT_LDS>(f, c0, 0, f.nb0);
    if (__any(f.nfl > 1)) feed_land_chunk<IDENTITY, CT_LDS>(f, c1, 1, f.nb1);
    f.wr += 4 * f.nfl;
    f.nfl = 0;
}
template <bool IDENTITY, bool CT_LDS>
__device__ __forceinline__ void feed_land(BitFeed& f) { feed_land<IDENTITY, CT_LDS>(f, f.fl0, f.fl1); }
// asks for n (0..2) chunks; a lane with `ask` false keeps what it has in flight
__device__ __forceinline__ void feed_issue(BitFeed& f, uint32_t n, bool ask, uint4& c0, uint4& c1) {
    const int left = f.bytes_left;
    const uint32_t nb0 = (uint32_t)(left > 16 ? 16 : left), nb1 = (uint32_t)(left > 32 ? 16 : (left > 16 ? left - 16 : 0));
    const bool want0 = ask && n > 0 && nb0 > 0, want1 = ask && n > 1 && nb1 > 0;
    if (__any((want0 && f.next + 16 > f.in_end) || (want1 && f.next + 32 > f.in_end))) {     // the blob's very last frame
        if (want0) c0 = load_chunk_guarded(f.next, f.in_end);
        if (want1) c1 = load_chunk_guarded(f.next + 16, f.in_end);
    } else {
        if (want0) c0 = ld_u128_unaligned(f.next);
        if (__any(want1)) { if (want1) c1 = ld_u128_unaligned(f.next + 16); }
    }
    if (ask) {
        const int adv = (int)(n > 1 ? nb0 + nb1 : (n > 0 ? nb0 : 0u));
        f.next += adv; f.bytes_left = left - adv;
        f.nb0 = nb0; f.nb1 = nb1; f.nfl = n;
    }
}
__device__ __forceinline__ void feed_issue(BitFeed& f, uint32_t n, bool ask = true) { feed_issue(f, n, ask, f.fl0, f.fl1); }
// The spectra loop's form: when any lane asks, every lane loads (no lane-masked definition of c0 / c1 for the compiler to keep
// copies of -- the registers are plain results of one load each), from an address pulled back inside the blob when the 16 bytes
// would reach past its end (the landing step shifts those down again: nb0 / nb1 carry the distance in their second byte).
__device__ __forceinline__ void feed_issue_wave(BitFeed& f, uint32_t n, bool ask, uint4& c0, uint4& c1) {
    const int left = f.bytes_left;
    const uint32_t nb0 = (uint32_t)(left > 16 ? 16 : left), nb1 = (uint32_t)(left > 32 ? 16 : (left > 16 ? left - 16 : 0));
    const bool want0 = ask && n > 0 && nb0 > 0, want1 = ask && n > 1 && nb1 > 0;
    const uint8_t* lim = f.in_end - 16;
    const uint8_t* p0 = f.next;
    const uint8_t* p1 = f.next + 16;
    const uint32_t sh0 = p0 > lim ? (uint32_t)(p0 - lim) : 0u, sh1 = p1 > lim ? (uint32_t)(p1 - lim) : 0u;
    // (a lane that does not ask has nothing in flight and lands nothing: what it loads is dead.  It loads the blob's last 16 bytes -- one
    //  line for all such lanes, hot after the first time -- not the next chunk of its own frame, which it would fetch again when it
    //  does ask: tools/traffic_census.py counted 45 % more chunk bytes requested than the frames hold, a line's worth of traffic each)
    if (__any(want0)) c0 = ld_u128_unaligned((!want0 || p0 > lim) ? lim : p0);
    if (__any(want1)) c1 = ld_u128_unaligned((!want1 || p1 > lim) ? lim : p1);
    if (ask) {
        const int adv = (int)(n > 1 ? nb0 + nb1 : (n > 0 ? nb0 : 0u));
        f.next += adv; f.bytes_left = left - adv;
        f.nb0 = nb0 | ((sh0 > 16 ? 16u : sh0) << 8); f.nb1 = nb1 | ((sh1 > 16 ? 16u : sh1) << 8); f.nfl = n;
    }
}
// `eager`: ask whenever there is room (always safe: see the invariant above).  Otherwise only a lane whose ring could run dry
// asks: `thresh` = the most words the block about to be parsed and the one after it can take, + 2 (the bounds are exact
// per block, see `needtab` in k_hca_parse; chunks asked for now land before the second of those blocks starts).
// In the spectra loop the lanes top up together every HCA_FEED_SYNC-th block (a landing step then serves most of the wave at
// once); in between only when a lane would run dry.  A chunk asked for at the top of a block's iteration lands at its bottom.
#ifndef HCA_FEED_SYNC
#define HCA_FEED_SYNC 4     // (measured, parse of 4.69 M frames: round 2: 1: 9.68 ms, 2: 9.38, 3: 9.07, 4: 9.00, 5: 9.33; round 5, alternating on one box: 2: 8.88, 3: 8.62, 4: 8.53, 5: 8.68; sparse material 3: 10.10, 4: 9.96)
#endif
__device__ __forceinline__ void feed_request(BitFeed& f, const BitBuf& b, bool eager, uint32_t thresh, uint4& c0, uint4& c1) {
    const uint32_t h = f.wr - b.rd, room = RING_WORDS - h;
    const bool ask = f.nfl == 0 && (eager || h < thresh);
    feed_issue_wave(f, room >> 2 > 2 ? 2u : room >> 2, ask, c0, c1);
}
__device__ __forceinline__ void feed_request(BitFeed& f, const BitBuf& b, bool eager = true, uint32_t thresh = 0) { feed_request(f, b, eager, thresh, f.fl0, f.fl1); }
// one refill opportunity per symbol; branch-free, the LDS read issued here is consumed by the NEXT call
__device__ __forceinline__ void bb_refill(BitBuf& b, const uint32_t* ring) {
    const bool need = b.off >= 32;                   // the window's first word is used up: slide by one word
    b.hi = need ? b.lo : b.hi;
    b.lo = need ? b.nw : b.lo;
    b.off &= 31;
    b.rd += need ? 1u : 0u;
    b.nw = ring[(b.rd & (RING_WORDS - 1)) * 64];
}
// MSB-first peek of n (0..12) bits.  CHECKED = the reference reader's end-of-frame behaviour (hca.cpp:225-281): 0 when
// the read crosses the frame end; and 0 when fewer than 24 (16) bits are left but the read spans more than 16 (8) bits
// from its byte start -- the reference then serves it from a window that is too narrow (its shift count wraps).
template <bool CHECKED>
__device__ __forceinline__ uint32_t bb_peek(const BitBuf& b, int n) {
    const uint32_t w = (uint32_t)(((((uint64_t)b.hi << 32) | b.lo) << b.off) >> 32);    // the next 32 bits
    const uint32_t v = __builtin_amdgcn_ubfe(w, 32 - n, n);                 // their top n (v_bfe_u32: n == 0 gives 0)
    if (!CHECKED) return v;
    const int left = b.size - b.pos;
    const int off = n + (b.pos & 7);
    const bool zero = (n > left) | ((left < 24) & ((off >= 17) | ((off >= 9) & (left < 16))));
    return zero ? 0u : v;
}
__device__ __forceinline__ void bb_skip(BitBuf& b, int n) { b.off += n; b.pos += n; }
__device__ __forceinline__ uint32_t bb_read(BitBuf& b, const uint32_t* ring, int n) { bb_refill(b, ring); uint32_t v = bb_peek<true>(b, n); bb_skip(b, n); return v; }

// One spectral symbol (hca.cpp:1546-1563).  `meta` = bits | T << 4 describes the band's code (band_meta): `bits` = most bits a
// symbol can take (hcatbdecoder_max_bit_table), nshort = nstab[T], and both code families reduce to one rule --
//   code < 2*nshort : symbol = code >> 1, one bit is given back          (prefix codes: the short codewords;
//   otherwise       : symbol = code - nshort                              sign-magnitude, nshort = 1: the zero)
//   value = +ceil(symbol/2) for odd symbols, -symbol/2 for even ones
// which reproduces read_bit/read_val for resolutions 1..7 (nshort = 2^bits - (2*res+1)) and the sign-magnitude form
// with its "zero gives the sign bit back" rule for resolutions 8..15.  The caller refills once per two symbols.
template <bool CHECKED>
__device__ __forceinline__ int parse_symbol(BitBuf& bb, uint32_t meta, const uint8_t* nstab) {
    const uint32_t bits = meta & 15, ns = nstab[(meta >> 4) & 15];
    const uint32_t code = bb_peek<CHECKED>(bb, (int)bits);
    const bool is_short = code < 2 * ns;
    const uint32_t sym = is_short ? (code >> 1) : (code - ns);
    const uint32_t len = bits - (is_short ? 1u : 0u);
    bb.off += len;
    if (CHECKED) bb.pos += (int)len;
    return (int)((sym >> 1) ^ (uint32_t)__builtin_amdgcn_sbfe(sym, 0, 1));       // = -(value): 0, -1, 1, -2, 2, ...
}
// two symbols' negated values as an int16 pair (still negated), and the two record forms made of such pairs:
//  int16 lines: the pair negated back (v_pk_sub_i16);  int8 lines (HCA_REC_NARROW frames): the low bytes of two pairs,
//  left negated -- the transform negates that frame's gains instead, which is exact
__device__ __forceinline__ uint32_t pair_negated(int n0, int n1) { return __builtin_amdgcn_perm((uint32_t)n1, (uint32_t)n0, 0x05040100u); }   // n1.lo16 : n0.lo16
__device__ __forceinline__ uint32_t pair_to_i16(uint32_t p) {
    typedef short s2 __attribute__((ext_vector_type(2)));
    const s2 z = {0, 0};
    return __builtin_bit_cast(uint32_t, z - __builtin_bit_cast(s2, p));
}
__device__ __forceinline__ uint32_t pairs_to_i8(uint32_t p01, uint32_t p23) { return __builtin_amdgcn_perm(p23, p01, 0x06040200u); }
// refill for up to two symbols (24 bits)
__device__ __forceinline__ void pair_refill(BitBuf& b, const uint32_t* ring) { bb_refill(b, ring); }
// code description of a band of resolution res (see parse_symbol)
__device__ __forceinline__ uint32_t band_bits(uint32_t res) { return res > 7 ? res - 3 : (0x44443320u >> (res * 4)) & 15; }   // 0,2,3,3,4,4,4,4,5,...,12
__device__ __forceinline__ uint32_t band_nshort(uint32_t res) { return res > 7 ? 1u : (0x13571310u >> (res * 4)) & 15; }    // 0,1,3,1,7,5,3,1
// Code description of a band: bits | T << 4.  For the prefix codes (resolutions 0..7, at most four bits) T is the short-code
// bound in terms of the NEXT FOUR stream bits: a symbol is short (takes bits - 1) iff those four bits, as a number, are below
// T = 2 * nshort << (4 - bits) -- 0, 8, 12, 4, 14, 10, 6, 2 for resolutions 0..7, all different, so T / 2 also names the
// resolution.  Resolutions 8..15 (sign-magnitude, nshort = 1) carry T = 1.
__device__ __forceinline__ uint32_t band_meta(uint32_t res) {
    const uint32_t bits = band_bits(res);
    const uint32_t t = res > 7 ? 1u : (2 * band_nshort(res)) << (4 - bits);
    return bits | (t << 4);
}

// The sixteen description bytes hashed into 32 table slots (as a byte offset into a float table): k_hca_transform_plain finds
// a band's dequantiser range by the description the parse left, not by computing the resolution again.
__device__ __forceinline__ uint32_t desc_key(uint32_t meta) { return ((meta * 69u) >> 4) & 0x7Cu; }

// Symbol values of the prefix codes: entry [T / 2][next four bits of the stream] = -value & 0xFF, the byte an int8 record line
// holds (HCA_REC_NARROW).  In a block of 16 bands in which no frame of the wave has a longer code (parse_block path below)
// a symbol's LENGTH is one compare against T -- the only thing the next symbol waits for -- and its value is a table read
// that nothing waits for until the bytes are packed.  nstab[T] = nshort for the arithmetic path (parse_symbol).
#define HCA_LUT_BYTES 128
__device__ __forceinline__ void build_symbol_lut(uint8_t* lut, uint16_t* lut16, uint8_t* nstab, uint32_t lane) {
#pragma unroll
    for (uint32_t e = lane; e < 128; e += 64) {
        const uint32_t res = e >> 4, idx = e & 15, bits = band_bits(res), ns = band_nshort(res);
        const uint32_t code = idx >> (4 - bits);
        const bool is_short = code < 2 * ns;
        const uint32_t sym = is_short ? (code >> 1) : (code - ns);
        const uint32_t negv = (sym >> 1) ^ (uint32_t)__builtin_amdgcn_sbfe(sym, 0, 1);
        lut[(band_meta(res) >> 5) * 16 + idx] = (uint8_t)negv;
        lut16[(band_meta(res) >> 5) * 16 + idx] = (uint16_t)(0u - negv);      // the value itself, for int16 lines
    }
    if (lane < 16) {
        uint32_t ns = 1;                                      // T = 1: resolutions 8..15
#pragma unroll
        for (uint32_t res = 0; res < 8; res++) if ((band_meta(res) >> 4) == lane) ns = band_nshort(res);
        nstab[lane] = (uint8_t)ns;
    }
}

// transposed flush of the 16 staged words of every lane: frame fr's words go to its record + byte_off, 64 B per frame.
// The records of a tile's frames are consecutive (cri_capi.cpp lays a format group's records out in frame order), so
// a lane owns a quarter (16 B) of one frame's 64 B per step: `recq` = tile's records + (lane >> 2) * record_bytes +
// (lane & 3) * 16, four steps of 16 frames.  OST = 66 keeps the four gathered LDS words of a half-wave on distinct banks.
#define OST 66
__device__ __forceinline__ void flush16(const uint32_t* ostage, uint8_t* recq, uint32_t rb16, uint32_t nvalid, uint32_t lane, uint32_t byte_off, uint32_t nwords) {
    wave_lds_sync();
    const uint32_t wq = (lane & 3) * 4, fq = lane >> 2;
    const bool all = nvalid == 64 && nwords == 16;                // (wave-uniform) the usual case: no per-lane masks
#pragma unroll
    for (uint32_t it = 0; it < 4; it++) {
        const uint32_t fr = it * 16 + fq;
        const uint32_t* src = ostage + wq * OST + fr;
        const uint4 v = make_uint4(src[0], src[OST], src[2 * OST], src[3 * OST]);
        uint4* dst = (uint4*)(recq + (size_t)it * rb16 + byte_off);
        if (all) *dst = v;
        else if (fr < nvalid && wq < nwords) *dst = v;
    }
    wave_lds_sync();
}
// the same for quantised lines: 64 B of each of the tile's 64 frames go to one quarter of a row (cri_types.h), which is 4 KB of
// contiguous memory -- lane l's 16 bytes of step `it` sit at qoff + it * 1 KB + l * 16
__device__ __forceinline__ void flush16_qc(const uint32_t* ostage, uint8_t* qc_tile, uint32_t nvalid, uint32_t lane, uint32_t qoff, uint32_t nwords) {
    wave_lds_sync();
    const uint32_t wq = (lane & 3) * 4, fq = lane >> 2;
    const bool all = nvalid == 64 && nwords == 16;
#pragma unroll
    for (uint32_t it = 0; it < 4; it++) {
        const uint32_t fr = it * 16 + fq;
        const uint32_t* src = ostage + wq * OST + fr;
        const uint4 v = make_uint4(src[0], src[OST], src[2 * OST], src[3 * OST]);
        uint4* dst = (uint4*)(qc_tile + qoff + it * 1024 + lane * 16);
        // (streaming stores: nobody reads the lines before the transform kernel, and the wave waits for its stores at the bottom
        //  of the next block together with its loads -- 9.17 -> 8.80 ms for the kernel)
        typedef uint32_t u4v __attribute__((ext_vector_type(4)));
        const u4v vv = {v.x, v.y, v.z, v.w};
        if (all) __builtin_nontemporal_store(vv, (u4v*)dst);
        else if (fr < nvalid && wq < nwords) __builtin_nontemporal_store(vv, (u4v*)dst);
    }
    wave_lds_sync();
}

// The staged words are stored one checkpoint later than they are complete: right after the feed's loads have landed and
// before the next ones are asked for (feed_land), so that no checkpoint waits on stores it has just issued.
struct PendingFlush {
    uint32_t byte_off, nwords;               // nwords == 0: nothing pending (wave-uniform)
    bool qc;                                 // byte_off is a quarter-row offset inside the tile's quantised lines, not a record offset
    __device__ __forceinline__ void set(uint32_t off, uint32_t n) { byte_off = off; nwords = n; qc = false; }
    __device__ __forceinline__ void set_qc(uint32_t off, uint32_t n) { byte_off = off; nwords = n; qc = true; }
    __device__ __forceinline__ void run(const uint32_t* ostage, uint8_t* recq, uint8_t* qc_tile, uint32_t rb16, uint32_t nvalid, uint32_t lane) {
        if (nwords) { if (qc) flush16_qc(ostage, qc_tile, nvalid, lane, byte_off, nwords); else flush16(ostage, recq, rb16, nvalid, lane, byte_off, nwords); }
        nwords = 0;
    }
};

template <bool IDENTITY, bool CT_LDS>
__global__ __launch_bounds__(64, 4) void k_hca_parse(HcaDecArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const Fmt F = load_fmt(a.formats + a.format);
    const uint32_t C = F.channels, lane = threadIdx.x, tile = blockIdx.x + a.tile_begin;
    uint32_t* ring = (uint32_t*)smem + lane;           // [RING_WORDS + 4][64]
    uint32_t* ostage = (uint32_t*)(smem + (RING_WORDS + 4) * 256);   // [16][OST]
    uint8_t* curve = (uint8_t*)(ostage + 16 * OST);    // 96 bytes reserved
    uint8_t* needtab = curve + 96;                     // [C][8] most ring words a block of 16 symbols can take over the wave's 64 frames; bit 7: no code of the block has more than four bits
    uint8_t* lut = needtab + 128;                      // [8][16] symbol values of the prefix codes (build_symbol_lut)
    uint8_t* nstab = lut + 128;                        // [16] short-code count by T
    uint16_t* lut16 = (uint16_t*)(nstab + 16);         // [8][16] the same values as int16, not negated (tiles whose lines are int16)
    uint8_t* cipher_lds = (uint8_t*)(lut16 + 128);     // [n_cipher][256] (jobs with up to 16 cipher tables)
    build_symbol_lut(lut, lut16, nstab, lane);
    for (uint32_t i = lane; i < 66; i += 64) curve[i] = HCA_CURVE_TO_RES[i];
    if (!IDENTITY && CT_LDS) for (uint32_t i = lane; i < a.n_cipher * 64; i += 64) ((uint32_t*)cipher_lds)[i] = ((const uint32_t*)a.cipher_tables)[i];

    const uint32_t g = tile * 64 + lane;
    const bool valid = g < a.frames;
    // the records of a format group are consecutive in frame order (cri_capi.cpp): frame g's starts at group base + g * record_bytes
    uint8_t* tile_rec = a.scratch + a.streams[a.stream_begin].scratch_offset + (uint64_t)tile * 64 * F.record_bytes;
    uint8_t* rec = tile_rec + (uint64_t)lane * F.record_bytes;
    uint8_t* recq = tile_rec + (uint64_t)(lane >> 2) * F.record_bytes + (lane & 3) * 16;
    const uint32_t rb16 = 16 * F.record_bytes;
    uint8_t* qc_tile = a.scratch + a.qc_offset + (uint64_t)tile * HCA_QC_TILE(C);
    const uint32_t nvalid = a.frames - tile * 64 < 64 ? a.frames - tile * 64 : 64;
    int status = 0;
    // every lane parses (frames that fail sync / checksum parse to ignored output; the padding lanes of the last tile read zeros)
    uint4* metag = (uint4*)(a.scratch + a.resg_offset) + (uint64_t)tile * C * 8 * 64 + lane;

    PendingFlush pend; pend.set(0, 0);
    BitFeed fd;
    {
        uint32_t si = a.stream_begin, f = 0;
        if (valid) { si = find_stream(a.streams, a.stream_begin, a.stream_end, g); f = g - a.streams[si].first_frame; }
        const uint64_t src_offset = a.streams[si].src_offset;
        const uint32_t cidx = a.streams[si].cipher;
        fd.next = a.in + src_offset + (uint64_t)f * F.frame_size;
        fd.in_end = a.in + a.in_bytes;
        fd.ct = (CT_LDS ? cipher_lds : a.cipher_tables) + cidx * 256;
        fd.bytes_left = valid ? (int)F.frame_size : 0;
    }
    fd.ring = ring; fd.wr = 0; fd.nfl = 0; fd.nb0 = fd.nb1 = 0; fd.r = 0; fd.par = 0;
    fd.fl0 = fd.fl1 = make_uint4(0, 0, 0, 0);
    BitBuf bb;
    bb.hi = 0; bb.lo = 0; bb.off = 0; bb.pos = 0; bb.size = (int)F.frame_size * 8; bb.rd = 0; bb.nw = 0;
    wave_lds_sync();
    // prime: the ring's 16 words (the bytes of a frame shorter than that are followed by zeros)
    feed_issue(fd, 2);
    const bool sync_bad = valid && (fd.fl0.x & 0xFFFF) != 0xFFFF;                 // hca.cpp:1162-1164 (frame_size >= 8)
    feed_land<IDENTITY, CT_LDS>(fd);
    feed_issue(fd, 2);
    feed_land<IDENTITY, CT_LDS>(fd);
    wave_lds_sync();
    bb.hi = ring[0]; bb.lo = ring[64]; bb.nw = ring[128]; bb.rd = 2;
    bb_skip(bb, 16);                                             // sync word
    uint32_t packed = 0, flags = 0, draws = 0, wide_bits = 0;
    {
        const uint32_t nl = bb_read(bb, ring, 9), eb = bb_read(bb, ring, 7);   // hca.cpp:1175-1178
        packed = (nl << 8) - eb;
    }
    const uint8_t* ath = a.ath_tables + F.ath_index * 128;
    uint8_t* sfst = (uint8_t*)ostage;
    wave_lds_sync();
    for (uint32_t c = 0; c < C; c++) {
        const uint32_t coded = F.coded(c), type = F.type(c), groups = F.hfr_group_count;
        uint32_t cs = coded, extra = 0;
        if (!(type == CRI_CH_SECONDARY || groups == 0 || F.version <= 0x0200)) { extra = groups; cs += extra; }
        uint32_t db = bb_read(bb, ring, 3), value = 0;
        uint32_t n_noise = 0, n_valid = 0;                        // bands reconstruct_noise fills / draws from (hca.cpp:1489-1497)
        if (cs > 128) { status = status ? status : CRI_ERR_HCA_FRAME(5); cs = 0; }
        const uint32_t expected = (1u << db) - 1;
        // scalefactors + resolutions in blocks of 16 bands (the last block is padded with zeros)
        for (uint32_t blk = 0; blk < 8; blk++) {
            // (as in the spectra loop below: stores and loads are issued here, the chunks land at the bottom of the iteration)
            pend.run(ostage, recq, qc_tile, rb16, nvalid, lane);
            uint4 c0, c1;
            feed_request(fd, bb, true, 0, c0, c1);
            uint32_t sfw[4] = {0, 0, 0, 0};
            uint32_t mw[4] = {0, 0, 0, 0};
            // far from the frame end (always, in a well-formed frame) the reader's end-of-frame rules cannot apply
            const bool sf_fast = __all(bb.size - bb.pos >= 16 * 12 + 24);
            if (blk * 16 < cs || blk == 0) {
#pragma unroll
                for (uint32_t k = 0; k < 16; k++) {               // hca.cpp:1310-1350, all lanes in lock step
                    const uint32_t i = blk * 16 + k;
                    // (with delta coding the first 6-bit value is read even when the channel codes no band at all, hca.cpp:1322-1324)
                    const bool in = i < cs || (i == 0 && db > 0 && db < 6);
                    const bool direct = db >= 6 || i == 0;
                    uint32_t x, y; bool esc;
                    if (sf_fast) {                                // delta and its possible 6-bit escape value in one peek
                        const bool none = !in || db == 0;
                        bb_refill(bb, ring);
                        const uint32_t both = bb_peek<false>(bb, none ? 0 : (direct ? 6 : (int)db + 6));
                        x = direct ? both : (both >> 6); y = both & 63;
                        esc = in && !direct && db > 0 && x == expected;
                        bb_skip(bb, none ? 0 : (direct ? 6 : (esc ? (int)db + 6 : (int)db)));
                    } else {
                        x = bb_read(bb, ring, (!in || db == 0) ? 0 : (direct ? 6 : (int)db));
                        esc = in && !direct && db > 0 && x == expected;
                        y = bb_read(bb, ring, esc ? 6 : 0);
                    }
                    const int t = (int)value + ((int)x - (int)(expected >> 1));
                    if (in && db > 0 && !direct && !esc && (t < 0 || t >= 64) && status == 0) status = CRI_ERR_HCA_FRAME(5);
                    uint32_t v = direct ? x : (esc ? y : ((value - (expected >> 1) + x) & 0x3F));
                    if (db == 0 || !in) v = 0;
                    value = in ? v : value;
                    sfw[k >> 2] |= v << (8 * (k & 3));
                    uint32_t res = 0;                             // calculate_resolution, hca.cpp:1450-1488
                    if (i < coded && v > 0) {
                        const int noise = (int)ath[i] + (int)((packed + i) >> 8);
                        const int cp = noise + 1 - (int)((5 * v) >> 1);
                        res = cp < 0 ? 15u : (cp <= 65 ? (uint32_t)curve[cp] : 0u);
                        res = res > F.max_res ? F.max_res : (res < F.min_res ? F.min_res : res);
                    }
                    mw[k >> 2] |= band_meta(res) << (8 * (k & 3));
                }
            }
            if (F.min_res == 0) {                                 // v3.0 only: count the noise / valid bands from the block's bytes
#pragma unroll
                for (uint32_t k = 0; k < 16; k++) {
                    const bool live = blk * 16 + k < coded && ((sfw[k >> 2] >> (8 * (k & 3))) & 0xFF) != 0;
                    const bool coded_res = ((mw[k >> 2] >> (8 * (k & 3))) & 0xFF) != 0;      // band_meta(0) == 0 (no bits, resolution 0)
                    n_noise += live && !coded_res ? 1u : 0u; n_valid += live && coded_res ? 1u : 0u;
                }
            }
            if (a.narrow) {                                       // a band whose symbols can take more than 8 bits (resolution >= 12) needs int16 lines
#pragma unroll
                for (uint32_t q = 0; q < 4; q++) wide_bits |= ((mw[q] & 0x0F0F0F0Fu) + 0x07070707u) & 0x10101010u;
            }
            {   // bits the block's 16 symbols can take (the low nibbles), the most over the wave, as ring words: a refill slides the
                // window by one word, and the window may already be up to 55 bits in when the block starts
                uint32_t sb = 0;
#pragma unroll
                for (uint32_t q = 0; q < 4; q++) sb = __builtin_amdgcn_sad_u8(mw[q] & 0x0F0F0F0Fu, 0u, sb);
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)sb, o); sb = t > sb ? t : sb; }
                uint32_t longc = 0;                              // a code of more than four bits (resolution >= 8)?
#pragma unroll
                for (uint32_t q = 0; q < 4; q++) longc |= ((mw[q] & 0x0F0F0F0Fu) + 0x0B0B0B0Bu) & 0x10101010u;
                const bool lut_ok = !__any(longc != 0);
                if (lane == 0) needtab[c * 8 + blk] = (uint8_t)(((sb + 55) >> 5) | (lut_ok ? 0x80u : 0u));
            }
#pragma unroll
            for (uint32_t q = 0; q < 4; q++) ostage[((blk & 3) * 4 + q) * OST + lane] = sfw[q];
            if ((blk & 3) == 3) {
                if (blk == 7) {                                   // the high end also carries the HFR scales
                    // derived HFR scales of v3.0 (hca.cpp:1353-1355); the entry one past the decoded range reads as 0.
                    // (v3.0 derived scales need scalefactors[cs - i]; they are re-read from the record after the flush.)
                    // v2.0: explicit HFR scales follow the scalefactors (hca.cpp:1427-1437)
                    if (type != CRI_CH_SECONDARY && F.version <= 0x0200) {
                        for (uint32_t k = 0; k < groups; k++) {
                            const uint32_t v = bb_read(bb, ring, 6);
                            const uint32_t di = 128 - groups + k;
                            sfst[(((di >> 2) & 15) * OST + lane) * 4 + (di & 3)] = (uint8_t)v;
                        }
                    }
                }
                pend.set(HCA_REC_SF(C, c) + (blk >> 2) * 64, 16);
            }
            feed_land<IDENTITY, CT_LDS>(fd, c0, c1);
            metag[(c * 8 + blk) * 64] = make_uint4(mw[0], mw[1], mw[2], mw[3]);      // (after the wait, not before it)
        }
        if (extra) {                                              // v3.0: scalefactors[127 - i] = scalefactors[cs - i]
            pend.run(ostage, recq, qc_tile, rb16, nvalid, lane);
            __syncthreads();                                      // (global data written by other lanes: drain the stores)
            if (valid) for (uint32_t i = 0; i < extra; i++) {
                const uint32_t srci = cs - i;
                rec[HCA_REC_SF(C, c) + 127 - i] = srci < cs ? rec[HCA_REC_SF(C, c) + srci] : 0;
            }
        }
        // unpack_intensity, hca.cpp:1361-1441.  Entries the reference leaves untouched (they keep the previous frame's
        // value: the v2.0 "index 15" form for entries 1..7, the v3.0 delta form from the first out-of-range value on) are
        // recorded as 0xFF; the transform resolves them by walking back through the stream's records.
        uint32_t inten_lo = 0, inten_hi = 0;
        if (type == CRI_CH_SECONDARY) {
            uint8_t iv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            bb_refill(bb, ring);
            uint32_t v = bb_peek<true>(bb, 4);
            if (F.version <= 0x0200) {
                iv[0] = (uint8_t)v;
                const bool take = v < 15;
                if (!take) flags |= 1u << c;
                bb_skip(bb, take ? 4 : 0);
                for (int k = 1; k < 8; k++) { const uint32_t t = bb_read(bb, ring, take ? 4 : 0); iv[k] = take ? (uint8_t)t : (uint8_t)0xFF; }
            } else if (v < 15) {                                  // v3.0 forms; divergent but rare
                bb_skip(bb, 4);
                const uint32_t dbi = bb_read(bb, ring, 2);
                iv[0] = (uint8_t)v;
                if (dbi == 3) { for (int k = 1; k < 8; k++) iv[k] = (uint8_t)bb_read(bb, ring, 4); }
                else {
                    const uint32_t bmax = (2u << dbi) - 1, bits = dbi + 1;
                    bool bad = false;
                    for (int k = 1; k < 8; k++) {
                        if (!bad) {
                            const uint32_t delta = bb_read(bb, ring, (int)bits);
                            if (delta == bmax) v = bb_read(bb, ring, 4);
                            else { v = (v - (bmax >> 1) + delta) & 0xFF; if (v > 15) bad = true; }
                        }
                        iv[k] = bad ? (uint8_t)0xFF : (uint8_t)v;  // the reference returns at the first bad value
                    }
                    if (bad) flags |= 1u << (16 + c);
                }
            } else { bb_skip(bb, 4); for (int k = 0; k < 8; k++) iv[k] = 7; }
            inten_lo = iv[0] | (iv[1] << 8) | (iv[2] << 16) | ((uint32_t)iv[3] << 24);
            inten_hi = iv[4] | (iv[5] << 8) | (iv[6] << 16) | ((uint32_t)iv[7] << 24);
        }
        if (valid) { uint32_t* ip = (uint32_t*)(rec + HCA_REC_INT(C, c)); ip[0] = inten_lo; ip[1] = inten_hi; }
        draws += (n_noise > 0 && n_valid > 0) ? 8 * n_noise : 0u;   // generator draws of this channel in the frame's 8 subframes (hca.cpp:1608-1633)
    }
    // ---- spectra: 8 subframes x C channels x coded symbols, serial per lane (hca.cpp:1194-1199, 1540-1571),
    //      in blocks of 16 symbols; bands past `coded` carry resolution 0 = no bits
    //      The code descriptions of a block are loaded one block ahead, before the pending stores: every vector memory
    //      operation a block waits for (feed_land) is then a whole block old.
    //      Formats the in-lane transform handles (a.narrow) store the lines as int8 -- half the record bytes, and this kernel is
    //      partly bound by the CU's store path -- when no band of the tile's 64 frames can exceed 8 bits (the usual case by far).
    const bool narrow = a.narrow != 0 && !__any(wide_bits != 0);
    //      (channels without coded bands -- a secondary channel of a format with base_band_count 0 -- have no blocks)
    uint32_t first_c = 0;
    while (first_c + 1 < C && F.coded(first_c) == 0) first_c++;
    // (c, blk) -> the block parsed after it: the next block of the channel, else block 0 of the next channel that has blocks
    auto block_after = [&](uint32_t c, uint32_t blk, uint32_t& nc, uint32_t& nb) {
        nb = blk + 1; nc = c;
        if (nb >= ((F.coded(c) + 15) >> 4)) {
            nb = 0;
            do { nc = nc + 1 == C ? 0 : nc + 1; } while (F.coded(nc) == 0 && nc != c);
        }
    };
    // One iteration = one block of 16 symbols:
    //   top     everything that goes to memory: the previous block's finished lines are stored, the code descriptions of the NEXT
    //           block are asked for, and the lanes' next chunks (together every HCA_FEED_SYNC-th block; in between only when a
    //           lane could run dry);
    //   middle  the block's symbols;
    //   bottom  the chunks land in the ring: the one place the iteration waits for memory, for operations a whole block old.
    // The chunk registers live inside one iteration on purpose: as loop-carried values the compiler kept copying them at the
    // back edge, which made every iteration wait for everything it had just issued, stores included (s_waitcnt vmcnt(0)).
    feed_land<IDENTITY, CT_LDS>(fd);                              // what the scalefactor pass left in flight
    pend.run(ostage, recq, qc_tile, rb16, nvalid, lane);
    uint4 mv_next = metag[first_c * 8 * 64];
    __builtin_amdgcn_s_waitcnt(0x0F70);                           // vmcnt(0): nothing is pending when the loop is entered, so the
                                                                  // waits the compiler places inside it stay exact counts
    wave_lds_sync();                                              // (needtab)
    uint32_t ck = 0;
    for (uint32_t sf = 0; sf < 8; sf++) {
        for (uint32_t c = 0; c < C; c++) {
            const uint32_t nblk = (F.coded(c) + 15) >> 4;
            for (uint32_t blk = 0; blk < nblk; blk++) {
                uint32_t nb, nc;
                block_after(c, blk, nc, nb);
                const uint32_t nt = needtab[c * 8 + blk];
                const uint32_t thresh = (nt & 0x7F) + ((uint32_t)needtab[nc * 8 + nb] & 0x7F) + 2;
                const uint4 mv = mv_next;
                pend.run(ostage, recq, qc_tile, rb16, nvalid, lane);      // the previous block's lines
                mv_next = metag[(nc * 8 + nb) * 64];
                uint4 c0, c1;
                if (ck == 0 || __any(fd.wr - bb.rd < thresh)) feed_request(fd, bb, ck == 0, thresh, c0, c1);   // (most blocks: nothing to ask for)
                ck = ck + 1 == HCA_FEED_SYNC ? 0 : ck + 1;
                const uint32_t mw[4] = {mv.x, mv.y, mv.z, mv.w};
                const bool fast = __all(bb.size - bb.pos >= 16 * 12 + 24);
                if (fast && !narrow && (nt & 0x80)) {
                    // the same for a tile whose lines are int16 (some other block of some frame has a long code): the table gives
                    // the 16-bit value, two of them make a line word
                    const uint32_t off0 = bb.off, rd0 = bb.rd;
#pragma unroll
                    for (uint32_t q = 0; q < 4; q++) {
                        bb_refill(bb, ring);
                        const uint64_t win = ((uint64_t)bb.hi << 32) | bb.lo;
                        uint32_t e[4];
#pragma unroll
                        for (uint32_t j = 0; j < 4; j++) {
                            const uint32_t idx = (uint32_t)((win << bb.off) >> 60);
                            const uint32_t t = __builtin_amdgcn_ubfe(mw[q], 8 * j + 4, 4), bits = __builtin_amdgcn_ubfe(mw[q], 8 * j, 4);
                            e[j] = lut16[(t << 3) + idx];
                            bb.off += bits - (idx < t ? 1u : 0u);
                        }
                        ostage[((blk & 1) * 8 + 2 * q) * OST + lane] = e[0] | (e[1] << 16);
                        ostage[((blk & 1) * 8 + 2 * q + 1) * OST + lane] = e[2] | (e[3] << 16);
                    }
                    bb.pos += (int)(bb.off - off0) + 32 * (int)(bb.rd - rd0);
                    if ((blk & 1) || blk + 1 == nblk)
                        pend.set_qc(HCA_QC_ROW(C, sf, c) + (blk >> 1) * HCA_QC_QUARTER, (blk & 1) ? 16 : 8);
                } else if (fast && narrow && (nt & 0x80)) {
                    // every code of the block has at most four bits, in all 64 frames.  Length of a symbol = bits - (next four
                    // bits < T): shift, compare, add-with-carry is all the next symbol waits for; the value byte comes from
                    // the table whenever it comes.  The window needs a refill only every four symbols (<= 16 bits).
                    const uint32_t off0 = bb.off, rd0 = bb.rd;
#pragma unroll
                    for (uint32_t q = 0; q < 4; q++) {
                        bb_refill(bb, ring);
                        const uint64_t win = ((uint64_t)bb.hi << 32) | bb.lo;
                        uint32_t e[4];
#pragma unroll
                        for (uint32_t j = 0; j < 4; j++) {
                            const uint32_t idx = (uint32_t)((win << bb.off) >> 60);                   // the next four bits
                            const uint32_t t = __builtin_amdgcn_ubfe(mw[q], 8 * j + 4, 4), bits = __builtin_amdgcn_ubfe(mw[q], 8 * j, 4);
                            e[j] = lut[(t << 3) + idx];
                            bb.off += bits - (idx < t ? 1u : 0u);
                        }
                        // the four line bytes, first band in the low byte
                        const uint32_t lo2 = __builtin_amdgcn_perm(e[1], e[0], 0x0c0c0400u), hi2 = __builtin_amdgcn_perm(e[3], e[2], 0x04000c0cu);
                        ostage[((blk & 3) * 4 + q) * OST + lane] = lo2 | hi2;
                    }
                    bb.pos += (int)(bb.off - off0) + 32 * (int)(bb.rd - rd0);
                    if ((blk & 3) == 3 || blk + 1 == nblk)
                        pend.set_qc(HCA_QC_ROW(C, sf, c) + (blk >> 2) * HCA_QC_QUARTER, 4 * ((blk & 3) + 1));
                } else {
                    uint32_t words[8];
                    if (fast) {
                        const uint32_t off0 = bb.off, rd0 = bb.rd;
#pragma unroll
                        for (uint32_t k = 0; k < 8; k++) {
                            pair_refill(bb, ring);
                            const int v0 = parse_symbol<false>(bb, (mw[k >> 1] >> (16 * (k & 1))) & 0xFF, nstab);
                            const int v1 = parse_symbol<false>(bb, (mw[k >> 1] >> (16 * (k & 1) + 8)) & 0xFF, nstab);
                            words[k] = pair_negated(v0, v1);
                        }
                        bb.pos += (int)(bb.off - off0) + 32 * (int)(bb.rd - rd0);
                    } else {
#pragma unroll
                        for (uint32_t k = 0; k < 8; k++) {
                            pair_refill(bb, ring);
                            const int v0 = parse_symbol<true>(bb, (mw[k >> 1] >> (16 * (k & 1))) & 0xFF, nstab);
                            const int v1 = parse_symbol<true>(bb, (mw[k >> 1] >> (16 * (k & 1) + 8)) & 0xFF, nstab);
                            words[k] = pair_negated(v0, v1);
                        }
                    }
                    if (narrow) {                              // (tile-uniform) int8 lines: 16 bytes per block, 64 B of a frame per four blocks
#pragma unroll
                        for (uint32_t k = 0; k < 4; k++) ostage[((blk & 3) * 4 + k) * OST + lane] = pairs_to_i8(words[2 * k], words[2 * k + 1]);
                        if ((blk & 3) == 3 || blk + 1 == nblk)
                            pend.set_qc(HCA_QC_ROW(C, sf, c) + (blk >> 2) * HCA_QC_QUARTER, 4 * ((blk & 3) + 1));
                    } else {
#pragma unroll
                        for (uint32_t k = 0; k < 8; k++) ostage[((blk & 1) * 8 + k) * OST + lane] = pair_to_i16(words[k]);
                        if ((blk & 1) || blk + 1 == nblk)
                            pend.set_qc(HCA_QC_ROW(C, sf, c) + (blk >> 1) * HCA_QC_QUARTER, (blk & 1) ? 16 : 8);
                    }
                }
                feed_land<IDENTITY, CT_LDS>(fd, c0, c1);
            }
        }
    }
    pend.run(ostage, recq, qc_tile, rb16, nvalid, lane);
    // whatever the parse left of the frame still goes through the checksum (the ring is no longer read: its room is its size)
    while (__any(fd.nfl > 0 || fd.bytes_left > 0)) {
        feed_land<IDENTITY, CT_LDS>(fd);
        feed_issue(fd, fd.bytes_left > 16 ? 2u : (fd.bytes_left > 0 ? 1u : 0u));
    }
    {   // hca.cpp:1162-1167: sync word, then checksum, come before anything the unpack finds
        const uint32_t rq = crcq_fold(fd.r);                  // < 2^15: the remainder modulo x^15 + x + 1
        const bool crc_bad = rq != 0 || (__builtin_popcount(fd.par) & 1);
        status = sync_bad ? CRI_ERR_HCA_FRAME(4) : (crc_bad ? CRI_ERR_HCA_FRAME(3) : status);
    }
    if (valid) {
        uint32_t* tail = (uint32_t*)(rec + HCA_REC_TAIL(C));
        tail[0] = packed; tail[1] = (uint32_t)status; tail[2] = flags | (narrow ? HCA_REC_NARROW : 0u); tail[3] = draws;      // k_hca_noise_scan turns tail[3] into a prefix
    }
    // The frames' code descriptions also go into their RECORDS (HCA_REC_DESC) for the in-lane transform: this lane's own stores of the
    // scalefactor pass come back (four blocks = 16 words a time) and leave transposed like the scalefactors, 64 B per frame and flush.
    // (here, after the last block, where nothing of the parse is live any more: placed between the scalefactor passes and the spectra
    //  loop the same lines cost three more spilled registers.)
    if (hca_transform_form(a) >= HCA_TR_INLANE_PLAIN) {
        for (uint32_t c = 0; c < C; c++) {
#pragma unroll
            for (uint32_t half = 0; half < 2; half++) {
#pragma unroll
                for (uint32_t q4 = 0; q4 < 4; q4++) {
                    const uint4 mv = metag[(c * 8 + half * 4 + q4) * 64];
                    ostage[(q4 * 4 + 0) * OST + lane] = mv.x; ostage[(q4 * 4 + 1) * OST + lane] = mv.y;
                    ostage[(q4 * 4 + 2) * OST + lane] = mv.z; ostage[(q4 * 4 + 3) * OST + lane] = mv.w;
                }
                flush16(ostage, recq, rb16, nvalid, lane, HCA_REC_DESC(C, c) + half * 64, 16);
            }
        }
    }
}

void launch_hca_parse(const HcaDecArgs& a, hipStream_t s) {
    if (!a.frames) return;
    const dim3 grid(a.tile_count ? a.tile_count : (a.frames + 63) / 64), block(64);
    const size_t lds = hca_parse_lds_bytes(a.cipher_identity ? 0 : a.n_cipher);
    if (a.cipher_identity) hipLaunchKernelGGL((k_hca_parse<true, true>), grid, block, lds, s, a);
    else if (a.n_cipher <= 16) hipLaunchKernelGGL((k_hca_parse<false, true>), grid, block, lds, s, a);
    else hipLaunchKernelGGL((k_hca_parse<false, false>), grid, block, lds, s, a);
}

// ------------------------------------------------------------------------------------------------------------
// HCA transform: one wave per frame
// ------------------------------------------------------------------------------------------------------------
// LDS (floats): S[C][128] spectra, G[C][128] gains, D[2][C][TR_DSTRIDE] DCT outputs of this and the previous subframe.
__device__ __forceinline__ int32_t cvt_trunc_x86(float v) {
    // (int)v as the x86-64 reference build evaluates it: out-of-range and NaN give INT_MIN
    return (v >= -2147483648.0f && v < 2147483648.0f) ? (int32_t)v : (int32_t)0x80000000;
}

#undef CRI_TABLE_QUAL
#define CRI_TABLE_QUAL static __device__ const
#include "cri_imdct_tables.h"
#define TR_DSTRIDE 144     // floats between the DCT-output ring's rows: the four slots of a pass touch the same columns of four
                           // rows, and a 128-float stride would put them all on the same LDS banks

#include "cri_dct_lane.h"

struct TransformCtx {
    const Fmt* F; const uint8_t* ath; float* S; float* G; uint32_t C, lane;
    // v3.0 noise fill (min_resolution == 0), hca.cpp:1602-1635
    bool noise; uint8_t* vlist; uint8_t* nrank; uint32_t* ncnt;      // [C][128] valid bands ascending, [C][128] rank of a noise band (0xFF: none), [C][2] = {noise_count, valid_count}
};

// calculate_resolution for one band (hca.cpp:1450-1488)
__device__ __forceinline__ uint32_t band_resolution(const Fmt& F, const uint8_t* ath, uint32_t packed, uint32_t i, uint32_t v) {
    if (v == 0) return 0;
    const int noise = (int)ath[i] + (int)((packed + i) >> 8);
    const int cp = noise + 1 - (int)((5 * v) >> 1);
    uint32_t res = cp < 0 ? 15u : (cp <= 65 ? (uint32_t)HCA_CURVE_TO_RES[cp] : 0u);
    return res > F.max_res ? F.max_res : (res < F.min_res ? F.min_res : res);
}

// (the generator's O(log n) jump-ahead, lcg_jump, is in cri_bits.h)

// noise / valid band lists of one frame and channel (the `noises` array of hca.cpp:1489-1497), two bands per lane.
// Returns the draws one subframe of this channel consumes (noise_count, or 0 when reconstruct_noise returns early).
__device__ __forceinline__ uint32_t noise_lists(const Fmt& F, const uint8_t* ath, const uint8_t* rec, uint32_t C, uint32_t c, uint32_t lane,
                                               uint8_t* vlist, uint8_t* nrank, uint32_t* ncnt) {
    const uint32_t packed = ((const uint32_t*)(rec + HCA_REC_TAIL(C)))[0];
    const uint32_t sf2 = ((const uint16_t*)(rec + HCA_REC_SF(C, c)))[lane];
    bool isn[2], isv[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const uint32_t i = 2 * lane + h, v = (sf2 >> (8 * h)) & 0xFF;
        const bool live = i < F.coded(c) && v > 0;
        const uint32_t res = live ? band_resolution(F, ath, packed, i, v) : 0;
        isn[h] = live && res < 1; isv[h] = live && res >= 1;
    }
    const uint64_t below = (1ull << lane) - 1;
    const uint64_t bn0 = __ballot(isn[0]), bn1 = __ballot(isn[1]), bv0 = __ballot(isv[0]), bv1 = __ballot(isv[1]);
    const uint32_t nc = __popcll(bn0) + __popcll(bn1), vc = __popcll(bv0) + __popcll(bv1);
    if (vlist) {
        const uint32_t rn = __popcll(bn0 & below) + __popcll(bn1 & below), rv = __popcll(bv0 & below) + __popcll(bv1 & below);
        nrank[c * 128 + 2 * lane] = isn[0] ? (uint8_t)rn : (uint8_t)0xFF;
        nrank[c * 128 + 2 * lane + 1] = isn[1] ? (uint8_t)(rn + (isn[0] ? 1 : 0)) : (uint8_t)0xFF;
        if (isv[0]) vlist[c * 128 + rv] = (uint8_t)(2 * lane);
        if (isv[1]) vlist[c * 128 + rv + (isv[0] ? 1 : 0)] = (uint8_t)(2 * lane + 1);
        if (lane == 0) { ncnt[2 * c] = nc; ncnt[2 * c + 1] = vc; }
    }
    return (nc > 0 && vc > 0) ? nc : 0;
}

// gains of one frame: calculate_resolution + calculate_gain (hca.cpp:1444-1507), two bands per lane
__device__ __forceinline__ void frame_gains(const TransformCtx& X, const uint8_t* rec) {
    const Fmt& F = *X.F;
    const uint32_t packed = ((const uint32_t*)(rec + HCA_REC_TAIL(X.C)))[0];
    for (uint32_t c = 0; c < X.C; c++) {
        const uint32_t sf2 = ((const uint16_t*)(rec + HCA_REC_SF(X.C, c)))[X.lane];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t i = 2 * X.lane + h, v = (sf2 >> (8 * h)) & 0xFF;
            float gain = 0.0f;
            if (i < F.coded(c)) {
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

// quantised lines of a frame (cri_types.h): the frame's 64-byte column of its tile, quarter 0 of row (0, 0); g = the frame's
// number within its format group
__device__ __forceinline__ const uint8_t* qc_frame(const HcaDecArgs& a, uint32_t C, uint32_t g) {
    return a.scratch + a.qc_offset + (uint64_t)(g >> 6) * HCA_QC_TILE(C) + (g & 63) * 64;
}

// spectra of subframe sf of one frame into S: dequantise, HFR, intensity stereo (hca.cpp:1566, 1638-1683, 1696-1714)
// `rnd` is the generator state before this subframe's first draw; it is advanced past the subframe's draws.
__device__ __forceinline__ void frame_spectra(const TransformCtx& X, const uint8_t* rec, const uint8_t* qcf, uint32_t sf, const uint8_t* inten /* [C][8] resolved */, uint32_t& rnd) {
    const Fmt& F = *X.F;
    const uint32_t C = X.C, lane = X.lane;
    for (uint32_t c = 0; c < C; c++) {
        const uint32_t q2 = *(const uint32_t*)(qcf + HCA_QC_ROW(C, sf, c) + (lane >> 4) * HCA_QC_QUARTER + (lane & 15) * 4);     // int16 lines 2*lane, 2*lane+1
        const uint32_t i0 = 2 * lane;
        float q0 = (float)(int)(int16_t)(q2 & 0xFFFF), q1 = (float)(int)(int16_t)(q2 >> 16);
        X.S[c * 128 + i0] = i0 < F.coded(c) ? X.G[c * 128 + i0] * q0 : 0.0f;
        X.S[c * 128 + i0 + 1] = i0 + 1 < F.coded(c) ? X.G[c * 128 + i0 + 1] * q1 : 0.0f;
    }
    wave_lds_sync();
    if (X.noise) {
        // reconstruct_noise (hca.cpp:1602-1635): noise band number k of (sf, c) takes draw k+1 after `rnd`; its source is a
        // valid band, which no noise band overwrites, so all noise bands of a channel are filled at once
        for (uint32_t c = 0; c < C; c++) {
            const uint32_t nc = X.ncnt[2 * c], vc = X.ncnt[2 * c + 1];
            if (nc == 0 || vc == 0) continue;
            const uint8_t* sfb = rec + HCA_REC_SF(C, c);
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint32_t b = 2 * lane + h, k = X.nrank[c * 128 + b];
                if (k != 0xFF) {
                    const uint32_t r = lcg_jump(rnd, k + 1);
                    const uint32_t vi = X.vlist[c * 128 + vc - 1 - (((r & 0x7FFF) * vc) >> 15)];
                    int sc = (int)sfb[b] - (int)sfb[vi] + 62;
                    sc = sc & ~(sc >> 31);
                    X.S[c * 128 + b] = HCA_SCALE_CONV[sc & 127] * X.S[c * 128 + vi];
                }
            }
            rnd = lcg_jump(rnd, nc);
        }
        wave_lds_sync();
    }
    if (F.bands_per_hfr_group > 0) {
        const int start = (int)(F.stereo_bands + F.base_bands), bpg = (int)F.bands_per_hfr_group, groups = (int)F.hfr_group_count;
        const int limit = F.version <= 0x0200 ? groups : (groups >> 1);
        const int total = (int)F.total_bands;
        // number of processed bands: stops at the first k with start+k >= total or low(k) < 0
        for (uint32_t c = 0; c < C; c++) {
            if (F.type(c) == CRI_CH_SECONDARY) continue;
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
            wave_lds_sync();
#pragma unroll
            for (int h = 0; h < 2; h++) if (idx[h] >= 0) X.S[c * 128 + idx[h]] = vals[h];
            wave_lds_sync();
            if (lane == 0 && start + nproc - 1 >= 0) X.S[c * 128 + start + nproc - 1] = 0.0f;
            wave_lds_sync();
        }
    }
    if (F.stereo_bands > 0) {
        for (uint32_t c = 0; c + 1 < C; c++) {
            if (F.type(c) != CRI_CH_PRIMARY) continue;
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
        wave_lds_sync();
    }
}

// intensity indexes of frame f with the "entry keeps its previous value" cases resolved (hca.cpp:1367-1375, 1405-1408):
// an entry recorded as 0xFF takes the value of the nearest earlier frame of the stream that set it (0 if none did).
__device__ __forceinline__ uint8_t intensity_walk_back(const Fmt& F, const uint8_t* rec_stream0, uint32_t f, uint32_t C, uint32_t c, uint32_t k, uint8_t v) {
    uint32_t ff = f;
    while (v == 0xFF) {
        if (ff == 0) { v = 0; break; }
        ff--;
        v = rec_stream0[(uint64_t)ff * F.record_bytes + HCA_REC_INT(C, c) + k];
    }
    return v;
}
__device__ __forceinline__ void resolve_intensity(const Fmt& F, const uint8_t* rec_stream0, uint32_t f, uint32_t C, uint32_t lane, uint8_t* inten) {
    if (lane < C * 8) {
        const uint32_t c = lane >> 3, k = lane & 7;
        const uint8_t* rec = rec_stream0 + (uint64_t)f * F.record_bytes;
        inten[lane] = intensity_walk_back(F, rec_stream0, f, C, c, k, rec[HCA_REC_INT(C, c) + k]);
    }
}

__global__ __launch_bounds__(64) void k_hca_transform_generic(HcaDecArgs a) {
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    const Fmt F = load_fmt(a.formats + a.format);
    const uint32_t C = F.channels, lane = threadIdx.x, g = blockIdx.x, slot = lane >> 4, l16 = lane & 15;
    float* S = fsm; float* G = S + C * 128; float* D = G + C * 128;           // D[2][C][TR_DSTRIDE]
    uint16_t* pcms = (uint16_t*)(D + 2 * C * TR_DSTRIDE);                      // [128][C] one subframe of interleaved PCM16
    uint8_t* inten = (uint8_t*)(pcms + 128 * C);      // [C][8]
    uint32_t* ncnt = (uint32_t*)(inten + ((C * 8 + 15) & ~15u));          // [C][2]
    uint8_t* vlist = (uint8_t*)(ncnt + 2 * C); uint8_t* nrank = vlist + C * 128;
    const uint32_t si = find_stream(a.streams, a.stream_begin, a.stream_end, g);
    const HcaStream st = a.streams[si];
    const uint32_t f = g - st.first_frame;
    const uint8_t* rec0 = a.scratch + st.scratch_offset;
    const uint8_t* rec = rec0 + (uint64_t)f * F.record_bytes;
    const int32_t status = (int32_t)((const uint32_t*)(rec + HCA_REC_TAIL(C)))[1];     // sync / CRC / unpack result of this frame
    if (status != 0) { if (lane == 0 && a.status) atomicMin(a.status + st.item, status); return; }
    if (f > 0 && (int32_t)((const uint32_t*)(rec - F.record_bytes + HCA_REC_TAIL(C)))[1] != 0) return;

    DctLane L;
    dct_lane_init(L, l16, HCA_DCT_LANE_SIN, HCA_DCT_LANE_COS);
    const uint2 dlogp = *(const uint2*)(HCA_DCT_LOGICAL + l16 * 8);          // 8 logical indexes, one byte each
    float w_lo[4], w_hi[4], w_rlo[4], w_rhi[4];                              // window (hca.cpp:1987-1992) at i, i+64, 63-i, 127-i
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const int i = (int)l16 + 16 * m;
        w_lo[m] = HCA_WINDOW[i]; w_hi[m] = HCA_WINDOW[i + 64]; w_rlo[m] = HCA_WINDOW[63 - i]; w_rhi[m] = HCA_WINDOW[127 - i];
    }
    TransformCtx X; X.F = &F; X.ath = a.ath_tables + F.ath_index * 128; X.S = S; X.G = G; X.C = C; X.lane = lane;
    X.noise = a.noise_fill != 0; X.vlist = vlist; X.nrank = nrank; X.ncnt = ncnt;
    uint32_t rnd = 1;                                // hca.cpp:961 (random = 1 at decoder reset)

    // DCT-IV of the C spectra in S, four channels at a time (one per 16 lanes), into row set `set` of D
    auto dct_rows = [&](uint32_t set) {
        for (uint32_t c0 = 0; c0 < C; c0 += 4) {
            const uint32_t c = c0 + slot, cc = c < C ? c : C - 1;
            const float4 s0 = *(const float4*)(S + cc * 128 + l16 * 8), s1 = *(const float4*)(S + cc * 128 + l16 * 8 + 4);
            f2 x[4] = {f2{s0.x, s0.y}, f2{s0.z, s0.w}, f2{s1.x, s1.y}, f2{s1.z, s1.w}};
            dct4_inplace(x, L);
            float* d = D + (set * C + cc) * TR_DSTRIDE;
            if (c < C) {
#pragma unroll
                for (int r = 0; r < 8; r++) d[((r < 4 ? dlogp.x : dlogp.y) >> (8 * (r & 3))) & 0xFF] = x[r >> 1][r & 1];
            }
        }
        wave_lds_sync();
    };

    // overlap tail: the previous frame's last subframe (hca.cpp:1990-1991); zeros at stream start (hca.cpp:962)
    if (f > 0) {
        const uint8_t* prec = rec - F.record_bytes;
        resolve_intensity(F, rec0, f - 1, C, lane, inten);
        frame_gains(X, prec);
        if (X.noise) {                               // generator state before the previous frame's subframe 7
            uint32_t per_sf = 0;
            for (uint32_t c = 0; c < C; c++) per_sf += noise_lists(F, X.ath, prec, C, c, lane, vlist, nrank, ncnt);
            rnd = lcg_jump(1, ((const uint32_t*)(prec + HCA_REC_TAIL(C)))[3] + 7 * per_sf);
        }
        wave_lds_sync();
        frame_spectra(X, prec, qc_frame(a, C, g - 1), 7, inten, rnd);
        dct_rows(1);
    }
    resolve_intensity(F, rec0, f, C, lane, inten);
    frame_gains(X, rec);
    if (X.noise) {
        for (uint32_t c = 0; c < C; c++) noise_lists(F, X.ath, rec, C, c, lane, vlist, nrank, ncnt);
        rnd = lcg_jump(1, ((const uint32_t*)(rec + HCA_REC_TAIL(C)))[3]);
    }
    wave_lds_sync();
    const bool dword_ok = ((st.delay * C * 2) & 3) == 0;
    uint8_t* dst = a.out + st.dst_offset;
    for (uint32_t sf = 0; sf < 8; sf++) {
        frame_spectra(X, rec, qc_frame(a, C, g), sf, inten, rnd);
        const uint32_t set = sf & 1;
        dct_rows(set);
        const bool have_prev = !(f == 0 && sf == 0);
        for (uint32_t c0 = 0; c0 < C; c0 += 4) {
            const uint32_t c = c0 + slot, cc = c < C ? c : C - 1;
            const float* d = D + (set * C + cc) * TR_DSTRIDE;
            const float* dp = D + ((set ^ 1) * C + cc) * TR_DSTRIDE;
            // window + overlap-add (hca.cpp:1987-1992), PCM16 (hca.cpp:339-360)
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const int i = (int)l16 + 16 * m;
                const float t0 = w_rhi[m] * dp[63 - i], t1 = w_rlo[m] * dp[i];
                const float p0 = have_prev ? t0 : 0.0f, p1 = have_prev ? t1 : 0.0f;
                const float v0 = w_lo[m] * d[i + 64] + p0, v1 = w_hi[m] * d[127 - i] - p1;     // wave[sf][i], wave[sf][i + 64]
                const float o0 = v0 * 32768.0f;
                const float o1 = v1 * 32768.0f;
                int32_t q0 = cvt_trunc_x86(o0), q1 = cvt_trunc_x86(o1);
                q0 = q0 > 32767 ? 32767 : (q0 < -32768 ? -32768 : q0);
                q1 = q1 > 32767 ? 32767 : (q1 < -32768 ? -32768 : q1);
                if (c < C) { pcms[i * C + c] = (uint16_t)(int16_t)q0; pcms[(i + 64) * C + c] = (uint16_t)(int16_t)q1; }
                if (a.float_out && c < C) {                    // validation hook: the samples before the int16 conversion
                    float* fo = a.float_out + st.float_offset + ((uint64_t)f * 1024 + sf * 128) * C + c;
                    fo[(uint64_t)i * C] = v0; fo[(uint64_t)(i + 64) * C] = v1;
                }
            }
        }
        wave_lds_sync();
        // 256*C contiguous bytes of interleaved PCM16 per subframe; delay / length trim of hca.cpp:3392-3425
        const uint32_t n0 = f * 1024 + sf * 128;
        if (dword_ok && n0 >= st.delay && n0 + 128 - st.delay <= st.samples) {
            uint32_t* q = (uint32_t*)(dst + (uint64_t)(n0 - st.delay) * C * 2);
            for (uint32_t k = 0; k < C; k++) q[k * 64 + lane] = ((const uint32_t*)pcms)[k * 64 + lane];
        } else {
            for (uint32_t e = lane; e < 128 * C; e += 64) {
                const uint32_t n = n0 + e / C;
                if (n >= st.delay && n - st.delay < st.samples) ((uint16_t*)dst)[(uint64_t)(n - st.delay) * C + e % C] = pcms[e];
            }
        }
        // (pcms is next written after the syncs of the next subframe's spectra and DCT)
    }
}

// Streams with min_resolution == 0 (v3.0): the noise generator runs on across the frames of a stream (hca.cpp:1616, 1633),
// so every frame needs the number of draws all earlier frames made.  k_hca_parse leaves each frame's own count in tail[3];
// one wave per stream turns them into exclusive prefixes, 64 frames per step (the transform then jumps the generator
// there with lcg_jump).
__global__ __launch_bounds__(64) void k_hca_noise_scan(HcaDecArgs a) {
    const Fmt F = load_fmt(a.formats + a.format);
    const uint32_t C = F.channels, lane = threadIdx.x;
    const HcaStream st = a.streams[a.stream_begin + blockIdx.x];
    uint32_t total = 0;
    for (uint32_t f0 = 0; f0 < st.frames; f0 += 64) {
        const uint32_t f = f0 + lane;
        uint32_t* tail = (uint32_t*)(a.scratch + st.scratch_offset + (uint64_t)(f < st.frames ? f : f0) * F.record_bytes + HCA_REC_TAIL(C));
        const uint32_t own = f < st.frames ? tail[3] : 0u;
        uint32_t incl = own;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, o); incl += (int)lane >= o ? t : 0u; }
        if (f < st.frames) tail[3] = total + incl - own;
        total += (uint32_t)__shfl((int)incl, 63);
    }
}

// ------------------------------------------------------------------------------------------------------------
// k_hca_transform: register-resident IMDCT, wavefront shuffles for the butterflies (1, 2, 4 channels; 6, 8 when every
// stereo pair starts on an even channel, so that a pair always shares a pass)
// ------------------------------------------------------------------------------------------------------------
// A wave owns a run of up to a.run_frames (8, 16 or 32: the planner, cri_capi.cpp) consecutive frames of one stream and walks its 8*C transforms per frame four at a
// time: transform slot j = lane >> 4, and the 16 lanes of a slot hold the 128 spectral lines of that transform, 8
// consecutive bands per lane (physical position p = lane16 * 8 + reg).  The 128-point DCT-IV of hca.cpp:1898-1980 is run
// in its in-place form (tools/gen_tables.py, imdct_inplace_maps): 14 butterfly stages between positions differing in one
// bit of p -- bits 0..2 are register pairs, bits 3..6 are lane16 ^ 1, 2, 4, 8 exchanged with DPP (quad_perm, row_shl/shr,
// row_ror) -- with exactly the reference's multiplies and adds (no FMA).  Window + overlap-add (hca.cpp:1987-1992) gathers
// the DCT outputs of a transform and of its predecessor (same channel, previous subframe) from a small LDS ring; the
// predecessor of a run's first subframe is recomputed from the previous frame's last subframe (a halo pass), so frames
// stay independent.  PCM16 is assembled in LDS and stored 256 contiguous bytes per wave instruction.

struct TrLds {
    float* G;          // [C][128] gains of the current frame
    // formats with high-frequency reconstruction / intensity stereo only (!PLAIN):
    float* hconv;      // [C][128] HFR scale of a reconstructed band (per frame)
    float* S;          // [4][128] dequantised coded lines of the pass's four transforms (HFR and intensity sources)
    float* conv;       // [128] HCA_SCALE_CONV
    float* iratio;     // [16] HCA_INTENSITY_RATIO
    uint8_t* sfb;      // [C][128] scalefactor bytes of the frame
    uint8_t* hlow;     // [128] source band of a reconstructed band (format constant)
    uint8_t* hgrp;     // [128] HFR group of a reconstructed band (format constant)
    uint8_t* inten;    // [C][8]
    // v3.0 noise reconstruction (min_resolution == 0) only:
    uint8_t* vlist;    // [C][128] valid bands in ascending order
    uint8_t* nrank;    // [C][128] rank of a noise band among the channel's noise bands (0xFF: not a noise band)
    uint32_t* nmeta;   // [C][4] = {noise_count, valid_count, draws of the lower channels in a subframe, draws of this channel}; then [per_sf, generator state at the frame's start]
    float* D;          // [8][128] ring of DCT outputs in logical order
    uint16_t* pcm;     // [512] int16 staging of one pass
    float* win;        // [128] synthesis window
    float* scale;      // [64] HCA_DEQ_SCALE
    float* range;      // [16] HCA_DEQ_RANGE
    uint8_t* curve;    // [66] HCA_CURVE_TO_RES
};

// what the per-frame setup needs from the frame record, fetched one frame ahead so its latency hides behind the
// previous frame's transforms
template <int C> struct FramePre { uint32_t packed; int32_t status; uint32_t flags; uint32_t ib; uint32_t noise0; uint32_t sf2[C]; };
template <bool PLAIN, int C>
__device__ __forceinline__ FramePre<C> tr_prefetch_frame(const uint8_t* rec, uint32_t lane) {
    FramePre<C> p;
    const uint32_t* tail = (const uint32_t*)(rec + HCA_REC_TAIL(C));
    p.packed = tail[0]; p.status = (int32_t)tail[1];
    p.flags = 0; p.ib = 0; p.noise0 = 0;
    if (!PLAIN) { p.flags = tail[2]; p.noise0 = tail[3]; p.ib = rec[HCA_REC_INT(C, 0) + (lane < C * 8 ? lane : 0)]; }   // intensity byte (channel lane >> 3, index lane & 7)
#pragma unroll
    for (int c = 0; c < C; c++) p.sf2[c] = ((const uint16_t*)(rec + HCA_REC_SF(C, c)))[lane];
    return p;
}

// per-frame setup: gains (hca.cpp:1444-1507), HFR source/scale per band (1638-1683), intensity indexes (1361-1441 + stale rule).
// Table lookups go to LDS copies and are unconditional (selects instead of branches around loads).
template <bool PLAIN, int C>
__device__ __forceinline__ void tr_setup_frame(const Fmt& F, const TrLds& T, const uint8_t* rec0, uint32_t f, uint32_t lane, int nproc,
                                               const FramePre<C>& pre, uint32_t ath2, bool narrow) {
    const uint32_t packed = __builtin_amdgcn_readfirstlane(pre.packed);
#pragma unroll
    for (int c = 0; c < C; c++) {
        const uint32_t sf2 = pre.sf2[c], coded = F.coded(c);
        float g[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t i = 2 * lane + h, v = (sf2 >> (8 * h)) & 0xFF;
            const int noise = (int)((ath2 >> (8 * h)) & 0xFF) + (int)((packed + i) >> 8);
            const int cp = noise + 1 - (int)((5 * v) >> 1);
            const int cpc = cp < 0 ? 0 : (cp > 65 ? 65 : cp);
            uint32_t res = T.curve[cpc];
            res = cp < 0 ? 15u : (cp > 65 ? 0u : res);
            res = res > F.max_res ? F.max_res : (res < F.min_res ? F.min_res : res);
            res = v > 0 ? res : 0u;
            const float gain = T.scale[v & 63] * T.range[res & 15];
            g[h] = i < coded ? (narrow ? -gain : gain) : 0.0f;      // (int8 lines are stored negated: negated gains, which is exact)
        }
        *(float2*)(T.G + c * 128 + 2 * lane) = make_float2(g[0], g[1]);
    }
    if (!PLAIN) {
        // scalefactor bytes to LDS: the HFR scale of band b needs the scalefactors of its group and of its source band
#pragma unroll
        for (int c = 0; c < C; c++) ((uint16_t*)T.sfb)[c * 64 + lane] = (uint16_t)pre.sf2[c];
        wave_lds_sync();
        if (F.bands_per_hfr_group > 0) {
            const int start = (int)(F.stereo_bands + F.base_bands), groups = (int)F.hfr_group_count;
#pragma unroll
            for (int c = 0; c < C; c++) {
                if (F.type(c) == CRI_CH_SECONDARY) continue;
                const uint8_t* sfb = T.sfb + c * 128;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int k = (int)lane + 64 * h, b = (start + k) & 127;
                    int sc = (int)sfb[(128 - groups + T.hgrp[b]) & 127] - (int)sfb[T.hlow[b]] + 63;
                    sc = sc & ~(sc >> 31);
                    if (k < nproc) T.hconv[c * 128 + b] = T.conv[sc & 127];
                }
            }
        }
        if (F.min_res == 0) {
            // v3.0 noise reconstruction (hca.cpp:1489-1497 lists, 1602-1635 use): per channel the noise bands (scalefactor > 0,
            // resolution 0) in band order and the valid bands; every (subframe, channel) draws noise_count numbers, in
            // subframe-major order, from a generator whose state at the frame's start is jumped to from k_hca_noise_scan's count
            uint32_t per_sf = 0;
#pragma unroll
            for (int c = 0; c < C; c++) {
                const uint32_t sf2 = pre.sf2[c], coded = F.coded(c);
                bool isn[2], isv[2];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const uint32_t i = 2 * lane + h, v = (sf2 >> (8 * h)) & 0xFF;
                    const int noise = (int)((ath2 >> (8 * h)) & 0xFF) + (int)((packed + i) >> 8);
                    const int cp = noise + 1 - (int)((5 * v) >> 1);
                    const int cpc = cp < 0 ? 0 : (cp > 65 ? 65 : cp);
                    uint32_t res = T.curve[cpc];
                    res = cp < 0 ? 15u : (cp > 65 ? 0u : res);
                    res = res > F.max_res ? F.max_res : res;
                    const bool live = i < coded && v > 0;
                    isn[h] = live && res < 1; isv[h] = live && res >= 1;
                }
                const uint64_t below = (1ull << lane) - 1;
                const uint64_t bn0 = __ballot(isn[0]), bn1 = __ballot(isn[1]), bv0 = __ballot(isv[0]), bv1 = __ballot(isv[1]);
                const uint32_t nc = __popcll(bn0) + __popcll(bn1), vc = __popcll(bv0) + __popcll(bv1);
                const uint32_t rn = __popcll(bn0 & below) + __popcll(bn1 & below), rv = __popcll(bv0 & below) + __popcll(bv1 & below);
                T.nrank[c * 128 + 2 * lane] = isn[0] ? (uint8_t)rn : (uint8_t)0xFF;
                T.nrank[c * 128 + 2 * lane + 1] = isn[1] ? (uint8_t)(rn + (isn[0] ? 1 : 0)) : (uint8_t)0xFF;
                if (isv[0]) T.vlist[c * 128 + rv] = (uint8_t)(2 * lane);
                if (isv[1]) T.vlist[c * 128 + rv + (isv[0] ? 1 : 0)] = (uint8_t)(2 * lane + 1);
                const uint32_t nd = (nc > 0 && vc > 0) ? nc : 0;
                if (lane == 0) { T.nmeta[c * 4] = nc; T.nmeta[c * 4 + 1] = vc; T.nmeta[c * 4 + 2] = per_sf; T.nmeta[c * 4 + 3] = nd; }
                per_sf += nd;
            }
            if (lane == 0) { T.nmeta[C * 4] = per_sf; T.nmeta[C * 4 + 1] = lcg_jump(1u, __builtin_amdgcn_readfirstlane(pre.noise0)); }
        }
        // intensity indexes (hca.cpp:1361-1441): the prefetched byte; 0xFF = "keeps its previous value" (rare), resolved by
        // walking back through the stream's records
        if (lane < C * 8) T.inten[lane] = intensity_walk_back(F, rec0, f, C, lane >> 3, lane & 7, (uint8_t)pre.ib);
    }
    wave_lds_sync();
}

// the 8 spectral lines b = lane16*8 + r of (frame record, subframe, channel) after dequantisation, high-frequency
// reconstruction and intensity stereo (hca.cpp:1566, 1638-1683, 1696-1714)
struct TrFetch { uint4 q; };         // quantised lines of (sf, c) for this lane's 8 bands
template <bool PLAIN, int C>
__device__ __forceinline__ TrFetch tr_fetch(const Fmt& F, const uint8_t* qcf, uint32_t sf, uint32_t c, uint32_t l16, bool narrow) {
    TrFetch t;
    if (narrow) {                                                                                         // int8 lines l16*8 .. +7 (wave-uniform)
        const uint2 h = *(const uint2*)(qcf + (HCA_QC_ROW(C, sf, c) + (l16 >> 3) * HCA_QC_QUARTER + (l16 & 7) * 8));
        t.q = make_uint4(h.x, h.y, 0, 0);
    } else t.q = *(const uint4*)(qcf + (HCA_QC_ROW(C, sf, c) + (l16 >> 2) * HCA_QC_QUARTER + (l16 & 3) * 16));    // int16 lines l16*8 .. +7
    return t;
}
template <bool PLAIN, int C>
__device__ __forceinline__ void tr_load_spectra(const Fmt& F, const TrLds& T, const TrFetch& ft, uint32_t sf, uint32_t c, uint32_t slot, uint32_t l16, int nproc, f2 x[4], bool narrow) {
    const bool secondary = F.type(c) == CRI_CH_SECONDARY;
    const uint32_t qw[4] = {ft.q.x, ft.q.y, ft.q.z, ft.q.w};
    if (PLAIN) {
        // gains are 0 past the coded bands, so no masking is needed
        const float4 g0 = *(const float4*)(T.G + c * 128 + l16 * 8), g1 = *(const float4*)(T.G + c * 128 + l16 * 8 + 4);
        const f2 g[4] = {f2{g0.x, g0.y}, f2{g0.z, g0.w}, f2{g1.x, g1.y}, f2{g1.z, g1.w}};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const f2 q = {(float)(int)(int16_t)(qw[k] & 0xFFFF), (float)((int)qw[k] >> 16)};
            x[k] = g[k] * q;
        }
        return;
    }
    // Dequantise the coded lines into registers and into the pass's LDS row; high-frequency reconstruction reads its
    // source band from that row, and a stereo secondary reads the primary's row (one slot down: same subframe, channel
    // c - 1) for the bands it shares with it.  Everything is selects over unconditional LDS reads.
    const float4 g0 = *(const float4*)(T.G + c * 128 + l16 * 8), g1 = *(const float4*)(T.G + c * 128 + l16 * 8 + 4);
    const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
    float own[8];
#pragma unroll
    for (int r = 0; r < 8; r++)                                                                         // gain is 0 past the coded bands
        own[r] = g[r] * (narrow ? (float)(int)(int8_t)(qw[r >> 2] >> (8 * (r & 3))) : (float)(int)(int16_t)(qw[r >> 1] >> (16 * (r & 1))));
    float* srow = T.S + slot * 128;
    *(float4*)(srow + l16 * 8) = make_float4(own[0], own[1], own[2], own[3]);
    *(float4*)(srow + l16 * 8 + 4) = make_float4(own[4], own[5], own[6], own[7]);
    wave_lds_sync();
    if (F.min_res == 0) {                                                        // reconstruct_noise, hca.cpp:1602-1635
        const uint32_t vc = T.nmeta[c * 4 + 1], nd = T.nmeta[c * 4 + 3];
        const uint2 rk = *(const uint2*)(T.nrank + c * 128 + l16 * 8);             // ranks of the lane's 8 bands
        uint32_t kfirst = 0xFF;
#pragma unroll
        for (int r = 7; r >= 0; r--) { const uint32_t k = ((r < 4 ? rk.x : rk.y) >> (8 * (r & 3))) & 0xFF; kfirst = k != 0xFF ? k : kfirst; }
        const bool mine = nd > 0 && kfirst != 0xFF;
        if (__any(mine)) {
            // draw k+1 of (subframe, channel) belongs to its noise band of rank k; a lane's noise bands have consecutive ranks
            const uint32_t base = lcg_jump(T.nmeta[C * 4 + 1], sf * T.nmeta[C * 4] + T.nmeta[c * 4 + 2]);
            uint32_t rcur = lcg_jump(base, (kfirst & 0x7F) + 1);
            bool started = false;
            const uint8_t* sfb = T.sfb + c * 128;
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint32_t b = l16 * 8 + r, k = ((r < 4 ? rk.x : rk.y) >> (8 * (r & 3))) & 0xFF;
                const bool isn = mine && k != 0xFF;
                const uint32_t rnext = rcur * 0x343FDu + 0x269EC3u;
                rcur = (isn && started) ? rnext : rcur;
                started = started || isn;
                const uint32_t vi = T.vlist[c * 128 + ((vc - 1 - (((rcur & 0x7FFF) * vc) >> 15)) & 127)];
                int sc = (int)sfb[b] - (int)sfb[vi & 127] + 62;
                sc = sc & ~(sc >> 31);
                const float nv = T.conv[sc & 127] * srow[vi & 127];
                own[r] = isn ? nv : own[r];
            }
        }
        wave_lds_sync();                                                         // (valid bands are never rewritten: reads above saw coded values)
        *(float4*)(srow + l16 * 8) = make_float4(own[0], own[1], own[2], own[3]);
        *(float4*)(srow + l16 * 8 + 4) = make_float4(own[4], own[5], own[6], own[7]);
        wave_lds_sync();
    }
    const bool stereo = F.stereo_bands > 0, hfr = F.bands_per_hfr_group > 0;
    const uint32_t cp = secondary && c > 0 ? c - 1 : c;                          // channel whose lines the shared bands come from
    const float* prow = secondary && slot > 0 ? srow - 128 : srow;
    const int start = (int)(F.stereo_bands + F.base_bands);
    const float rl = (stereo && (secondary || F.type(c) == CRI_CH_PRIMARY)) ? T.iratio[T.inten[(secondary ? c : c + 1) * 8 + sf] & 15] : 1.0f;
    const float rr = 2.0f - rl;
    const float4 p0 = *(const float4*)(prow + l16 * 8), p1 = *(const float4*)(prow + l16 * 8 + 4);
    const float pv[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
    const uint2 lowp = *(const uint2*)(T.hlow + l16 * 8);
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const uint32_t b = l16 * 8 + r;
        const bool from_prev = secondary && stereo && b >= F.base_bands && b < F.total_bands;   // intensity: R takes L's line (hca.cpp:1707-1711)
        const uint32_t cs = from_prev ? cp : c;
        const float* row = from_prev ? prow : srow;
        const bool cs_hfr = hfr && F.type(cs) != CRI_CH_SECONDARY;
        const uint32_t low = ((r < 4 ? lowp.x : lowp.y) >> (8 * (r & 3))) & 0xFF;
        const float hv = T.hconv[cs * 128 + b] * row[low & 127];                 // hca.cpp:1638-1683
        float v = from_prev ? pv[r] : own[r];
        v = b < F.coded(cs) ? v : ((cs_hfr && (int)b >= start && (int)b < start + nproc) ? hv : 0.0f);
        if (cs_hfr && (int)b == start + nproc - 1) v = 0.0f;                     // hca.cpp:1681
        if (stereo && b >= F.base_bands && b < F.total_bands) {
            if (from_prev) v = v * rr;
            else if (F.type(c) == CRI_CH_PRIMARY) v = v * rl;
        }
        x[r >> 1][r & 1] = v;
    }
}

// One loop over "steps": step -1 (only when the run does not start the stream) is the halo -- the DCT of the previous
// frame's last subframe, which only feeds the overlap ring -- and steps 0 .. nf*2C-1 are the passes of the run's frames.
// (the !PLAIN variants carry ~11 KB of LDS per wave, which already limits them to 3.5 waves per SIMD: give them the registers)
// FLT: validation instance that also stores the samples before the int16 conversion (HcaDecArgs::float_out)
template <bool PLAIN, int C, bool FLT>
__global__ __launch_bounds__(64, C > 4 ? 2 : (PLAIN ? 4 : 3)) void k_hca_transform(HcaDecArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const Fmt F = load_fmt(a.formats + a.format);
    const uint32_t lane = threadIdx.x, slot = lane >> 4, l16 = lane & 15;
    TrLds T;
    constexpr uint32_t RING = C > 4 ? 16 : 8;              // rows of the DCT-output ring: a pass's four plus the C predecessors
    constexpr uint32_t HALO_STEPS = (C + 3) / 4;
    T.G = (float*)smem; T.D = T.G + C * 128; T.pcm = (uint16_t*)(T.D + RING * TR_DSTRIDE);
    T.win = (float*)(T.pcm + (C > 4 ? 256 * C : 512)); T.scale = T.win + 128; T.range = T.scale + 64; T.curve = (uint8_t*)(T.range + 16);
    T.hconv = (float*)(T.curve + 80); T.S = T.hconv + C * 128; T.conv = T.S + 512; T.iratio = T.conv + 128;   // !PLAIN only from here
    T.sfb = (uint8_t*)(T.iratio + 16); T.hlow = T.sfb + C * 128; T.hgrp = T.hlow + 128; T.inten = T.hgrp + 128;
    T.nmeta = (uint32_t*)(T.inten + ((C * 8 + 15) & ~15)); T.vlist = (uint8_t*)(T.nmeta + C * 4 + 4); T.nrank = T.vlist + C * 128;

    // run -> stream, first frame
    uint32_t lo = a.stream_begin, hi = a.stream_end;
    const uint32_t run = blockIdx.x + a.run_begin;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (a.streams[mid].first_run <= run) lo = mid; else hi = mid; }
    const HcaStream st = a.streams[lo];
    const uint32_t f0 = (run - st.first_run) * a.run_frames;
    const uint32_t nf = st.frames - f0 < a.run_frames ? st.frames - f0 : a.run_frames;
    const uint8_t* rec0 = a.scratch + st.scratch_offset;

    // per-lane constants; small tables the setup indexes go to LDS
    DctLane L;
    dct_lane_init(L, l16, HCA_DCT_LANE_SIN, HCA_DCT_LANE_COS);
    const uint2 dlogp = *(const uint2*)(HCA_DCT_LOGICAL + l16 * 8);          // 8 logical indexes, one byte each
    T.win[lane] = HCA_WINDOW[lane]; T.win[lane + 64] = HCA_WINDOW[lane + 64];
    T.scale[lane] = HCA_DEQ_SCALE[lane];
    if (lane < 16) T.range[lane] = HCA_DEQ_RANGE[lane];
    T.curve[lane] = HCA_CURVE_TO_RES[lane]; if (lane < 2) T.curve[64 + lane] = HCA_CURVE_TO_RES[64 + lane];
    const uint32_t ath2 = ((const uint16_t*)(a.ath_tables + F.ath_index * 128))[lane];
    // bands reconstructed by HFR (format constant): stops when the high band leaves the spectrum or the low band index hits 0
    int nproc = 0;
    if (!PLAIN) {
        T.conv[lane] = HCA_SCALE_CONV[lane]; T.conv[lane + 64] = HCA_SCALE_CONV[lane + 64];
        if (lane < 16) T.iratio[lane] = HCA_INTENSITY_RATIO[lane];
        T.hlow[lane] = 0; T.hlow[lane + 64] = 0; T.hgrp[lane] = 0; T.hgrp[lane + 64] = 0;
        for (uint32_t i = lane; i < (uint32_t)C * 128; i += 64) T.hconv[i] = 0.0f;
        wave_lds_sync();
        if (F.bands_per_hfr_group > 0) {
            const int start = (int)(F.stereo_bands + F.base_bands), bpg = (int)F.bands_per_hfr_group, groups = (int)F.hfr_group_count;
            const int limit = F.version <= 0x0200 ? groups : (groups >> 1);
            nproc = groups * bpg;
            if (nproc > (int)F.total_bands - start) nproc = (int)F.total_bands - start;
            if (nproc < 0) nproc = 0;
            if (limit * bpg > start - 1 && nproc > start) nproc = start;
            for (int k = (int)lane; k < nproc; k += 64) {        // source band and group of every reconstructed band (hca.cpp:1650-1676)
                const int dec = k < limit * bpg ? k : limit * bpg;
                T.hlow[start + k] = (uint8_t)(start - 1 - dec);
                T.hgrp[start + k] = (uint8_t)(k / bpg);
            }
        }
    }
    wave_lds_sync();
    constexpr uint32_t PASSES = 2 * C;                     // passes per frame
    constexpr uint32_t SPAN = (4 / C) * 128;               // samples per channel completed by one pass
    const bool dword_ok = ((st.delay * C * 2) & 3) == 0;
    uint8_t* dst = a.out + st.dst_offset;

    // records are fetched ahead of use: the setup inputs one frame ahead, the quantised lines one step ahead
    const bool has_halo = f0 > 0;
    const uint32_t f_first = has_halo ? f0 - 1 : f0;
    const uint8_t* rec = rec0 + (uint64_t)f_first * F.record_bytes;         // record of the step's frame
    FramePre<C> pre = tr_prefetch_frame<PLAIN, C>(rec, lane);
    const uint32_t g0 = st.first_frame;                      // group frame number of the stream's frame 0
    // int8 (HCA_REC_NARROW) lines: the mono and stereo instances of the joint-stereo / HFR formats take them too (frames whose tile
    // allows it, see k_hca_parse); a frame's flag arrives with its setup inputs
    constexpr bool NWG = !PLAIN && C <= 2;
    bool ft_narrow = NWG && (__builtin_amdgcn_readfirstlane(pre.flags) & HCA_REC_NARROW) != 0;      // form of the lines in `ft`
    TrFetch ft = has_halo ? tr_fetch<PLAIN, C>(F, qc_frame(a, C, g0 + f_first), 7, slot < (uint32_t)C ? slot : 0, l16, ft_narrow)
                          : tr_fetch<PLAIN, C>(F, qc_frame(a, C, g0 + f_first), slot / C, slot % C, l16, ft_narrow);
    bool cur_narrow = ft_narrow;
    uint32_t f = f_first, pass = has_halo ? PASSES : 0;    // pass >= PASSES marks the halo steps (four channels each)
    uint32_t ring = RING;                                   // ring position of a normal step's slot 0
    const uint32_t f_end = f0 + nf;

#pragma unroll 1
    while (f < f_end) {
        const bool halo = pass >= PASSES;
        const uint32_t hc = (pass - PASSES) * 4 + slot;    // halo step: channel of this slot
        const bool hact = hc < (uint32_t)C;
        if ((halo && pass == PASSES) || pass == 0) {       // entering a frame
            const FramePre<C> cur_pre = pre;
            const int32_t status = __builtin_amdgcn_readfirstlane(cur_pre.status);
            if (status != 0) { if (!halo && lane == 0 && a.status) atomicMin(a.status + st.item, status); return; }
            if (f + 1 < f_end) pre = tr_prefetch_frame<PLAIN, C>(rec + F.record_bytes, lane);
            cur_narrow = NWG && (__builtin_amdgcn_readfirstlane(cur_pre.flags) & HCA_REC_NARROW) != 0;
            tr_setup_frame<PLAIN, C>(F, T, rec0, f, lane, nproc, cur_pre, ath2, cur_narrow);
        }
        const uint32_t t = halo ? 7 * C + (hact ? hc : 0) : pass * 4 + slot, sf = t / C, c = t % C;
        const TrFetch cur = ft;
        const bool last = halo ? pass + 1 == PASSES + HALO_STEPS : pass + 1 == PASSES;
        {   // next step's lines: same frame's next step, or the next frame's first pass
            if (!last) {
                const uint32_t tn = halo ? 7 * C + (hc + 4 < (uint32_t)C ? hc + 4 : 0) : t + 4;
                ft = tr_fetch<PLAIN, C>(F, qc_frame(a, C, g0 + f), tn / C, tn % C, l16, cur_narrow);
            } else if (f + 1 < f_end) {
                ft_narrow = NWG && (__builtin_amdgcn_readfirstlane(pre.flags) & HCA_REC_NARROW) != 0;
                ft = tr_fetch<PLAIN, C>(F, qc_frame(a, C, g0 + f + 1), slot / C, slot % C, l16, ft_narrow);
            }
        }
        f2 x[4];
        tr_load_spectra<PLAIN, C>(F, T, cur, sf, c, slot, l16, nproc, x, cur_narrow);
        dct4_inplace(x, L);
        const uint32_t dslot = halo ? ring - C + hc : ring + slot;
        float* d = T.D + (dslot & (RING - 1)) * TR_DSTRIDE;
        if (!halo || hact) {
#pragma unroll
            for (int r = 0; r < 8; r++) d[((r < 4 ? dlogp.x : dlogp.y) >> (8 * (r & 3))) & 0xFF] = x[r >> 1][r & 1];
        }
        wave_lds_sync();
        if (halo) {
            if (last) { pass = 0; f++; rec += F.record_bytes; } else pass++;
            continue;
        }

        // window + overlap-add (hca.cpp:1987-1992) against the predecessor (same channel, previous subframe)
        const float* dp = T.D + ((dslot - C) & (RING - 1)) * TR_DSTRIDE;
        const bool have_prev = !(f == 0 && sf == 0);                               // hca.cpp:962: tail starts as zeros
        const uint32_t sfl = C > 4 ? (sf & 1) : slot / C;                          // staging row set: subframe within this pass / subframe parity
        float o0[4], o1[4];
        uint32_t big = 0;                                  // largest |value| bit pattern: >= 2^31 or NaN needs x86 cvttss2si semantics
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int i = (int)l16 + 16 * m;
            const float t0 = T.win[127 - i] * dp[63 - i], t1 = T.win[63 - i] * dp[i];
            const float p0 = have_prev ? t0 : 0.0f, p1 = have_prev ? t1 : 0.0f;
            const float v0 = T.win[i] * d[i + 64] + p0, v1 = T.win[i + 64] * d[127 - i] - p1;        // wave[sf][i], wave[sf][i + 64]
            o0[m] = v0 * 32768.0f;
            o1[m] = v1 * 32768.0f;
            if (FLT) {
                float* fo = a.float_out + st.float_offset + ((uint64_t)f * 1024 + sf * 128) * C + c;
                fo[(uint64_t)i * C] = v0; fo[(uint64_t)(i + 64) * C] = v1;
            }
            const uint32_t u0 = __float_as_uint(o0[m]) & 0x7FFFFFFFu, u1 = __float_as_uint(o1[m]) & 0x7FFFFFFFu;
            big = max(big, max(u0, u1));
        }
        const bool slow_cvt = __any(big >= 0x4F000000u);
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int i = (int)l16 + 16 * m;
            int32_t q0, q1;                                                                  // hca.cpp:339-360
            if (slow_cvt) { q0 = cvt_trunc_x86(o0[m]); q1 = cvt_trunc_x86(o1[m]); }
            else { q0 = (int32_t)o0[m]; q1 = (int32_t)o1[m]; }
            q0 = q0 > 32767 ? 32767 : (q0 < -32768 ? -32768 : q0);
            q1 = q1 > 32767 ? 32767 : (q1 < -32768 ? -32768 : q1);
            T.pcm[((sfl * 128 + i) * C) + c] = (uint16_t)(int16_t)q0;
            T.pcm[((sfl * 128 + i + 64) * C) + c] = (uint16_t)(int16_t)q1;
        }
        wave_lds_sync();
        // interleaved PCM16 -- 1 KB per pass up to 4 channels, a whole subframe (256*C bytes) once its last channel is
        // done beyond; delay / length trim of hca.cpp:3392-3425
        const uint32_t sfd = C > 4 ? (pass * 4 + 4) / C - 1 : 0;                             // subframe completed by this pass
        if (C <= 4 || (pass * 4 + 4) / C != (pass * 4) / C) {
            constexpr uint32_t DW = C > 4 ? C : 4;                                           // dwords per lane
            const uint32_t n0 = C > 4 ? f * 1024 + sfd * 128 : f * 1024 + pass * SPAN;
            const uint32_t span = C > 4 ? 128 : SPAN;
            const uint32_t* src = (const uint32_t*)T.pcm + (C > 4 ? (sfd & 1) * 64 * C : 0);
            if (dword_ok && n0 >= st.delay && n0 + span - st.delay <= st.samples) {          // whole chunk inside the output
                uint32_t* q = (uint32_t*)(dst + (uint64_t)(n0 - st.delay) * C * 2);
#pragma unroll
                for (uint32_t k = 0; k < DW; k++) q[k * 64 + lane] = src[k * 64 + lane];
            } else {
#pragma unroll
                for (uint32_t q = 0; q < DW; q++) {
                    const uint32_t dw = q * 64 + lane, e0 = 2 * dw, e1 = e0 + 1;
                    const uint32_t na = n0 + e0 / C, nb = n0 + e1 / C;
                    const bool va = na >= st.delay && na - st.delay < st.samples, vb = nb >= st.delay && nb - st.delay < st.samples;
                    const uint32_t word = src[dw];
                    const uint64_t oa = ((uint64_t)(na - st.delay) * C + e0 % C) * 2, ob = ((uint64_t)(nb - st.delay) * C + e1 % C) * 2;
                    if (va && vb && dword_ok) *(uint32_t*)(dst + oa) = word;
                    else {
                        if (va) *(uint16_t*)(dst + oa) = (uint16_t)word;
                        if (vb) *(uint16_t*)(dst + ob) = (uint16_t)(word >> 16);
                    }
                }
            }
        }
        wave_lds_sync();
        ring += 4;
        if (++pass == PASSES) { pass = 0; f++; rec += F.record_bytes; }
    }
}

// ------------------------------------------------------------------------------------------------------------
// k_hca_transform_plain: formats without HFR / joint stereo / noise fill, 1, 2 or 4 channels -- overlap-add in the
// DCT's own lanes
// ------------------------------------------------------------------------------------------------------------
// In the in-place DCT-IV a lane's register pair j holds the outputs d[k] and d[127-k] (k < 64, HCA_DCT_LOGICAL), and the
// window/overlap-add of hca.cpp:1987-1992, written for those two, reads
//     out[63-k] = w[63-k]*d[127-k] + w[64+k]*prev[k]        out[64+k] = w[64+k]*d[127-k] - w[63-k]*prev[k]
// where prev[k] is the same position of the same channel's previous subframe.  So when a transform slot (16 lanes) always
// follows the same channel through consecutive subframes, the overlap state is four registers per lane, the window is
// eight per-lane constants, and nothing but the PCM staging goes through LDS.  The wave still owns a run of a.run_frames
// frames, but its 4 slots ("units") are 4/C groups x C channels: each group takes a contiguous part of the run and walks
// it frame by frame, subframe by subframe, after one halo pass (the subframe before its first one).
// k steps of the decoder's generator as one affine map, k = 0 .. 128: r_k = v[k].x * r + v[k].y
struct LcgMap { uint32_t x, y; };
struct LcgPow { LcgMap v[130]; };
constexpr LcgPow make_lcg_pow() {
    LcgPow t{};
    uint32_t m = 1, c = 0;
    for (int k = 0; k < 130; k++) { t.v[k].x = m; t.v[k].y = c; c = c * 0x343FDu + 0x269EC3u; m = m * 0x343FDu; }
    return t;
}
__device__ const LcgPow HCA_LCG_POW = make_lcg_pow();
#define HCA_PLAIN_LDS_BYTES (2048 + 1024 + 256 + 64 + 80 + 2048 + 2048)
#define HCA_PLAIN_JOINT_LDS_BYTES (HCA_PLAIN_LDS_BYTES + 2048 + 2048 + 512 + 64 + 128 + 16 + 512 + 256)
struct PlainPre { uint2 ps; uint32_t fl; uint32_t ib; uint32_t dr; uint32_t sf2[4]; };   // setup inputs of the four units' frames: lane v < 4 holds unit v's record tail {packed, status}, flags

#ifndef CRI_PLAIN_WAVES
#define CRI_PLAIN_WAVES 4
#endif
#ifndef CRI_JOINT_WAVES
#define CRI_JOINT_WAVES 3
#endif
// JOINT: the format has high-frequency reconstruction and / or intensity stereo (a stereo pair is an even channel and the next one,
// i.e. two neighbouring units of one group): each pass stages the units' dequantised lines in LDS, a reconstructed band reads its
// source band there, a secondary reads its primary's row.
// WIDE (C = 4 only): a plain format with CT = 3, 5, 6, 7 or 8 channels.  Plain channels do not depend on each other, so the
// workgroup is (CT + 3) / 4 waves that each take four channels of the SAME run through the same steps (channels past CT - 1 are
// dummies: they follow the last channel and their PCM goes to a dump slot), stage their PCM into one shared [128][CT] piece and,
// after a barrier, store it as whole sample frames -- 16 contiguous bytes per thread.  (A launch per channel group, each
// storing its 2C-byte pieces a sample frame apart, spent as long on those stores as on everything else: 7.8 ms against 3.9
// for 1000 eight-channel streams.)
// NOISE (with JOINT): v3.0 streams with min_resolution 0 -- a band whose resolution came out 0 is filled, subframe by subframe, with
// a randomly chosen coded band of its channel, scaled by the difference of their scalefactors (reconstruct_noise, hca.cpp:1602-1635),
// before HFR and intensity stereo read the spectrum.  The generator runs on through a stream's frames, channels and subframes:
// k_hca_noise_scan has left the draws before each frame in its record, a frame's start state is one O(log n) jump from the seed
// (on the scalar unit: it is the same for the whole unit), the step from one subframe to the next is one affine map per frame, and a
// lane reaches its first noise band of a subframe through a table of the generator's powers (HCA_LCG_POW).
template <int C, bool FLT, bool JOINT, bool WIDE = false, bool NOISE = false>
__global__ __launch_bounds__(WIDE ? 192 : 64, JOINT ? CRI_JOINT_WAVES : CRI_PLAIN_WAVES) void k_hca_transform_plain(HcaDecArgs a) {
    static_assert(!WIDE || C == 4, "the wide form is four channels per wave");
    static_assert(!NOISE || JOINT, "noise fill stages the spectrum like the joint form");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_all[];
    constexpr uint32_t NG = 4 / C;                         // groups = frames in flight
    const uint32_t CT = WIDE ? a.channels : (uint32_t)C;   // channels of the records, the line tiles and the PCM interleave
    const uint32_t wv = WIDE ? threadIdx.x >> 6 : 0u;
    constexpr uint32_t WAVE_LDS = NOISE ? HCA_PLAIN_JOINT_LDS_BYTES + 512 + 512 + 64 : (JOINT ? HCA_PLAIN_JOINT_LDS_BYTES : HCA_PLAIN_LDS_BYTES);
    uint8_t* smem = smem_all + wv * WAVE_LDS;
    uint16_t* pcmw = (uint16_t*)(smem_all + (WIDE ? (blockDim.x >> 6) * WAVE_LDS : 0u));      // WIDE: the shared [128][CT] piece, then 128 B of dump
    uint32_t* xnd = (uint32_t*)((uint8_t*)pcmw + 2048 + 128);     // WIDE + NOISE: [wave][unit] noise draws per subframe of every wave's channels (64 bytes)
    constexpr bool NW = true;                              // int8 lines (HCA_REC_NARROW) are read by all three instances
    const Fmt F = load_fmt(a.formats + a.format);
    const uint32_t lane = threadIdx.x & 63, u = lane >> 4, l16 = lane & 15, g = u / C, c = u % C;
    // WIDE: this wave's channels CB .. CB + CN - 1 -- groups of up to four, cut so that no stereo pair is split (5 channels: 3 + 2)
    uint32_t CB = 0, CN = C;
    if (WIDE) {
        for (uint32_t w = 0;; w++) {
            CN = CT - CB < 4 ? CT - CB : 4;
            if (CN == 4 && CB + 4 < CT && F.type(CB + 3) == CRI_CH_PRIMARY && F.type(CB + 4) == CRI_CH_SECONDARY) CN = 3;
            if (w == wv) break;
            CB += CN;
        }
    }
    auto chan_of = [&](uint32_t v) { return WIDE ? (v < CN ? CB + v : CB + CN - 1) : v % C; };      // channel of unit v
    const uint32_t cc = chan_of(u);
    const bool ch_live = !WIDE || c < CN;
    float* G = (float*)smem;                               // [4][128] gains of each unit's frame
    uint16_t* pcm = (uint16_t*)(G + 512);                  // [NG][128][C] one pass of PCM16
    float* scale = (float*)(pcm + 512); float* range = scale + 64; uint8_t* curve = (uint8_t*)(range + 16);   // 80 bytes

    uint32_t lo = a.stream_begin, hi = a.stream_end;
    const uint32_t run = blockIdx.x + a.run_begin;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (a.streams[mid].first_run <= run) lo = mid; else hi = mid; }
    const HcaStream st = a.streams[lo];
    const uint32_t f0 = (run - st.first_run) * a.run_frames;
    const uint32_t nf = st.frames - f0 < a.run_frames ? st.frames - f0 : a.run_frames;
    const uint8_t* rec0 = a.scratch + st.scratch_offset;
    const uint32_t h = (nf + NG - 1) / NG;                 // frames per group (the last groups may get fewer, or none)

    DctLane L;
    dct_lane_init(L, l16, HCA_DCT_LANE_SIN, HCA_DCT_LANE_COS);
    const uint2 dlogp = *(const uint2*)(HCA_DCT_LOGICAL + l16 * 8);
    // per-lane constants kept in LDS (12 registers otherwise): {w[63-k], w[64+k]} x 4 and the staging index of out[64+k] x 4
    // (as final LDS addresses: an offset would cost an add per store)
    float* wtab = (float*)(curve + 80) + lane * 8;         // [64][8]
    uint32_t* potab = (uint32_t*)(curve + 80 + 2048) + lane * 8;   // [64][8]: LDS addresses of out[64+k] x 4, of out[63-k] x 4
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t k = ((j < 2 ? dlogp.x : dlogp.y) >> (16 * (j & 1))) & 0xFF;      // register 2j: k < 64 (register 2j+1: 127 - k)
        wtab[2 * j] = HCA_WINDOW[63 - k]; wtab[2 * j + 1] = HCA_WINDOW[64 + k];
        if (WIDE) {
            uint8_t* dump = (uint8_t*)pcmw + 2048 + lane * 2;
            potab[j] = lds_address(ch_live ? (uint8_t*)pcmw + ((64 + k) * CT + CB + c) * 2 : dump);
            potab[4 + j] = lds_address(ch_live ? (uint8_t*)pcmw + ((63 - k) * CT + CB + c) * 2 : dump);
            continue;
        }
        const uint32_t po = ((g * 128 + 64 + k) * C + c) * 2;
        potab[j] = lds_address((uint8_t*)pcm + po); potab[4 + j] = lds_address((uint8_t*)pcm + (((2 * g * 128 + 127) * C + 2 * c) * 2 - po));
    }
    scale[lane] = HCA_DEQ_SCALE[lane];
    // A band's resolution is not computed again here (calculate_resolution, hca.cpp:1450-1488: ATH, noise level, curve, clamps): the
    // parse has left its code description byte in scratch (band_meta: one value per resolution), and desc_key() hashes the sixteen
    // values into 32 slots of the dequantiser's range table (the 144 bytes the range and curve tables used to take)
    if (lane < 32) range[lane] = 0.0f;
    wave_lds_sync();
    if (lane < 16) range[desc_key(band_meta(lane)) >> 2] = HCA_DEQ_RANGE[lane];
    // JOINT only: S[4][128] the pass's dequantised lines, hconv[4][128] HFR scale of a reconstructed band (per unit's frame),
    // conv[128], iratio[16], zero[4], ratio[4][8] intensity ratio of the unit's pair per subframe, sfb[4][128] scalefactor bytes,
    // hlow[128] / hgrp[128] source band and HFR group of a reconstructed band (format constants)
    float* S = (float*)(curve + 80 + 2048 + 2048); float* hconv = S + 512; float* conv = hconv + 512; float* iratio = conv + 128;
    float* ratio = iratio + 16; float* zero = ratio + 32; uint8_t* sfb = (uint8_t*)(zero + 4); uint8_t* hlow = sfb + 512; uint8_t* hgrp = hlow + 128;
    // NOISE only: nrank[4][128] rank of a band among its unit's noise bands (0xFF: not one), vlist[4][128] the unit's coded bands in
    // ascending order, nmeta[4][4] = {coded bands, draws per subframe, generator state of the unit's next subframe, -}
    uint8_t* nrank = hgrp + 128; uint8_t* vlist = nrank + 512; uint32_t* nmeta = (uint32_t*)(vlist + 512);
    constexpr uint32_t ZERO_IDX = 512 + 512 + 128 + 16 + 32;            // (index of `zero` from S)
    int nproc = 0;                                         // bands reconstructed by HFR (hca.cpp:1650-1676), as in k_hca_transform
    if (JOINT) {
        conv[lane] = HCA_SCALE_CONV[lane]; conv[lane + 64] = HCA_SCALE_CONV[lane + 64];
        if (lane < 16) iratio[lane] = HCA_INTENSITY_RATIO[lane];
        hlow[lane] = 0; hlow[lane + 64] = 0; hgrp[lane] = 0; hgrp[lane + 64] = 0;
        for (uint32_t i = lane; i < 512; i += 64) hconv[i] = 1.0f;       // (1: a band that is not reconstructed is taken as it is)
        if (lane < 4) zero[lane] = 0.0f;                                 // the line every band without a source reads
        wave_lds_sync();
        if (F.bands_per_hfr_group > 0) {
            const int start = (int)(F.stereo_bands + F.base_bands), bpg = (int)F.bands_per_hfr_group, groups = (int)F.hfr_group_count;
            const int limit = F.version <= 0x0200 ? groups : (groups >> 1);
            nproc = groups * bpg;
            if (nproc > (int)F.total_bands - start) nproc = (int)F.total_bands - start;
            if (nproc < 0) nproc = 0;
            if (limit * bpg > start - 1 && nproc > start) nproc = start;
            for (int k = (int)lane; k < nproc; k += 64) {
                const int dec = k < limit * bpg ? k : limit * bpg;
                hlow[start + k] = (uint8_t)(start - 1 - dec);
                hgrp[start + k] = (uint8_t)(k / bpg);
            }
        }
        wave_lds_sync();
    }
    // JOINT: where each of this lane's eight lines comes from -- format constants (tr_load_spectra states the same rules line by
    // line): its own coded line; for a secondary in the bands it shares, the primary's (hca.cpp:1707-1711); for a reconstructed
    // band the source band of the row it reads (hca.cpp:1638-1683; the last one is cleared, 1681); nothing.  The value is then
    // S[src] * hconv[unit][band] (1 unless reconstructed) * (the pair's ratio in the shared bands, else 1).
    uint32_t src_off[8];                                   // byte offsets into S
    uint32_t ratio_mask = 0;                               // bit r: line r takes the pair's ratio
    if (JOINT) {
        const uint32_t tc = F.type(cc);                    // (WIDE: a pair never straddles two waves -- pairs start on even channels, a wave on a multiple of four)
        const bool secondary = tc == CRI_CH_SECONDARY && c > 0, stereo = F.stereo_bands > 0, hfr = F.bands_per_hfr_group > 0;
        const int start = (int)(F.stereo_bands + F.base_bands);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t bnd = l16 * 8 + r;
            const bool shared = stereo && bnd >= F.base_bands && bnd < F.total_bands;
            const bool from_prev = secondary && shared;
            const uint32_t us = from_prev ? u - 1 : u, cs = from_prev ? cc - 1 : cc;
            const bool cs_hfr = hfr && F.type(cs) != CRI_CH_SECONDARY;
            uint32_t idx = ZERO_IDX;
            if (cs_hfr && (int)bnd == start + nproc - 1) idx = ZERO_IDX;                   // hca.cpp:1681 (with nothing reconstructed this is a coded band)
            else if (bnd < F.coded(cs)) idx = us * 128 + bnd;
            else if (cs_hfr && (int)bnd >= start && (int)bnd < start + nproc) idx = us * 128 + hlow[bnd];
            src_off[r] = idx * 4;
            if (shared && (from_prev || tc == CRI_CH_PRIMARY)) ratio_mask |= 1u << r;
        }
    }
    const bool dword_ok = ((st.delay * CT * 2) & 3) == 0;
    uint8_t* dst = a.out + st.dst_offset;

    // unit v (any lane can name it: the setup works on all four): first frame, frame count
    auto unit_first = [&](uint32_t v) { return f0 + (v / C) * h; };
    auto unit_count = [&](uint32_t v) { const uint32_t fb = unit_first(v); return fb < f0 + nf ? (f0 + nf - fb < h ? f0 + nf - fb : h) : 0u; };
    // frame of unit v in step s (s = -1: the halo frame); frames that do not exist are replaced by the run's first frame, which
    // is always there, and their results are dropped
    auto unit_frame = [&](uint32_t v, int s, bool& live) {
        const uint32_t fb = unit_first(v), n = unit_count(v);
        live = s < 0 ? (n > 0 && fb > 0) : ((uint32_t)s < n);
        return live ? fb + (uint32_t)s : f0;              // (fb + (uint32_t)-1 = fb - 1)
    };
    auto load_pre = [&](int s) {
        PlainPre p;
        {
            bool live; const uint32_t f = unit_frame(lane & 3, s, live);
            const uint32_t* tail = (const uint32_t*)(rec0 + (uint64_t)f * F.record_bytes + HCA_REC_TAIL(CT));
            p.ps = *(const uint2*)tail; p.fl = NW ? tail[2] : 0u;
            p.ps.y = live ? p.ps.y : 0u;
            p.dr = NOISE ? tail[3] : 0u;
        }
        p.ib = 0;
        if (JOINT) {                                       // lane < 32: intensity byte (lane & 7) of unit lane >> 3's pair (its secondary's entry)
            const uint32_t v = (lane >> 3) & 3, cv = chan_of(v), cs = (F.type(cv) == CRI_CH_SECONDARY || v % C + 1 >= CN) ? cv : cv + 1;
            bool live; const uint32_t f = unit_frame(v, s, live);
            p.ib = rec0[(uint64_t)f * F.record_bytes + HCA_REC_INT(CT, cs) + (lane & 7)];
        }
#pragma unroll
        for (uint32_t v = 0; v < 4; v++) {
            bool live; const uint32_t f = unit_frame(v, s, live);
            const uint8_t* rf = rec0 + (uint64_t)f * F.record_bytes;
            const uint32_t d2 = ((const uint16_t*)(rf + HCA_REC_DESC(CT, chan_of(v))))[lane];      // this lane's two bands (2 * lane, 2 * lane + 1)
            p.sf2[v] = ((const uint16_t*)(rf + HCA_REC_SF(CT, chan_of(v))))[lane] | d2 << 16;      // scalefactors | descriptions << 16
        }
        return p;
    };
    // HCA_REC_NARROW of this lane's unit's frame / of all four (the frame's lines are int8, negated: see k_hca_parse)
    auto my_narrow = [&](const PlainPre& p) {
        if (!NW) return false;
        const uint32_t z0 = __builtin_amdgcn_readlane(p.fl, 0), z1 = __builtin_amdgcn_readlane(p.fl, 1), z2 = __builtin_amdgcn_readlane(p.fl, 2), z3 = __builtin_amdgcn_readlane(p.fl, 3);
        return (((u & 2) ? ((u & 1) ? z3 : z2) : ((u & 1) ? z1 : z0)) & HCA_REC_NARROW) != 0;
    };
    auto none_narrow = [&](const PlainPre& p) {
        if (!NW) return true;
        return ((__builtin_amdgcn_readlane(p.fl, 0) | __builtin_amdgcn_readlane(p.fl, 1) | __builtin_amdgcn_readlane(p.fl, 2) | __builtin_amdgcn_readlane(p.fl, 3)) & HCA_REC_NARROW) == 0;
    };
    auto all_narrow = [&](const PlainPre& p) {
        if (!NW) return false;
        return (__builtin_amdgcn_readlane(p.fl, 0) & __builtin_amdgcn_readlane(p.fl, 1) & __builtin_amdgcn_readlane(p.fl, 2) & __builtin_amdgcn_readlane(p.fl, 3) & HCA_REC_NARROW) != 0;
    };
    // gains of the four units' frames (hca.cpp:1444-1507), two bands per lane; false if one of the frames is bad
    // NOISE: per-lane state of the unit's generator (set by setup for each step, advanced by every pass)
    uint32_t noise_vc[4] = {0, 0, 0, 0}, noise_nd[4] = {0, 0, 0, 0};
    uint32_t noise_am = 1, noise_ac = 0, noise_ja = 1, noise_jc = 0, noise_cur = 0, noise_vcu = 0;
    uint2 noise_rk = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
    bool noise_mine = false;
    auto setup = [&](const PlainPre& p, int cur_step) {
#pragma unroll
        for (uint32_t v = 0; v < 4; v++) {
            const int32_t status = (int32_t)__builtin_amdgcn_readlane(p.ps.y, v);
            if (status != 0) { if (lane == 0 && a.status) atomicMin(a.status + st.item, status); return false; }
        }
#pragma unroll
        for (uint32_t v = 0; v < 4; v++) {
            const uint32_t sf2 = p.sf2[v], coded = F.coded(chan_of(v));
            const bool neg = NW && (__builtin_amdgcn_readlane(p.fl, v) & HCA_REC_NARROW) != 0;      // negated lines: negated gains (exact)
            float gn[2];
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                const uint32_t i = 2 * lane + hh, sv = (sf2 >> (8 * hh)) & 0xFF, ds = (sf2 >> (16 + 8 * hh)) & 0xFF;
                const float gain = scale[sv & 63] * *(const float*)((const uint8_t*)range + desc_key(ds));
                gn[hh] = i < coded ? (neg ? -gain : gain) : 0.0f;
            }
            *(float2*)(G + v * 128 + 2 * lane) = make_float2(gn[0], gn[1]);
            if (JOINT) ((uint16_t*)sfb)[v * 64 + lane] = (uint16_t)sf2;
            if (NOISE) {                                   // the unit's noise / coded band lists (hca.cpp:1489-1497), two bands per lane
                bool isn[2], isv[2];
#pragma unroll
                for (int hh = 0; hh < 2; hh++) {
                    const uint32_t i = 2 * lane + hh, sv = (sf2 >> (8 * hh)) & 0xFF, ds = (sf2 >> (16 + 8 * hh)) & 0xFF;
                    const bool live = i < coded && sv > 0;
                    isn[hh] = live && ds == 0; isv[hh] = live && ds != 0;      // (description 0 = resolution 0)
                }
                const uint64_t below = (1ull << lane) - 1;
                const uint64_t bn0 = __ballot(isn[0]), bn1 = __ballot(isn[1]), bv0 = __ballot(isv[0]), bv1 = __ballot(isv[1]);
                const uint32_t nc = __popcll(bn0) + __popcll(bn1), vc = __popcll(bv0) + __popcll(bv1);
                const uint32_t rn = __popcll(bn0 & below) + __popcll(bn1 & below), rv = __popcll(bv0 & below) + __popcll(bv1 & below);
                nrank[v * 128 + 2 * lane] = isn[0] ? (uint8_t)rn : (uint8_t)0xFF;
                nrank[v * 128 + 2 * lane + 1] = isn[1] ? (uint8_t)(rn + (isn[0] ? 1 : 0)) : (uint8_t)0xFF;
                if (isv[0]) vlist[v * 128 + rv] = (uint8_t)(2 * lane);
                if (isv[1]) vlist[v * 128 + rv + (isv[0] ? 1 : 0)] = (uint8_t)(2 * lane + 1);
                noise_vc[v] = vc; noise_nd[v] = (nc > 0 && vc > 0) ? nc : 0u;
            }
        }
        if (NOISE) {
            // generator state at the start of each unit's next subframe: its frame's draws before it (k_hca_noise_scan), the whole
            // subframes before it (the halo step starts at subframe 7) and the channels below it -- all wave-uniform, scalar work
            const uint32_t sf0 = cur_step < 0 ? 7u : 0u;
            uint32_t others_all = 0, others_below = 0;       // WIDE: the frame's channels in the workgroup's other waves (all of them / the lower ones)
            if (WIDE) {
#pragma unroll
                for (uint32_t v = 0; v < 4; v++) if (v >= CN) noise_nd[v] = 0;      // (dummy units repeat the last channel: they draw nothing)
                if (lane < 4) xnd[wv * 4 + lane] = lane == 0 ? noise_nd[0] : (lane == 1 ? noise_nd[1] : (lane == 2 ? noise_nd[2] : noise_nd[3]));
                __syncthreads();
                for (uint32_t w = 0; w < (blockDim.x >> 6); w++) {
                    if (w == wv) continue;
                    const uint32_t t = xnd[w * 4] + xnd[w * 4 + 1] + xnd[w * 4 + 2] + xnd[w * 4 + 3];
                    others_all += t; others_below += w < wv ? t : 0u;
                }
                __syncthreads();                           // (the slots are written again in the next step's setup)
            }
#pragma unroll
            for (uint32_t v = 0; v < 4; v++) {
                const uint32_t gv = v / C;
                uint32_t per_sf = others_all, below_c = others_below;
#pragma unroll
                for (uint32_t w = 0; w < 4; w++) if (w / C == gv) { per_sf += noise_nd[w]; if (w < v) below_c += noise_nd[w]; }
                const uint32_t st = lcg_jump(1u, __builtin_amdgcn_readlane(p.dr, v) + sf0 * per_sf + below_c);
                uint32_t am = 0x343FDu, ac = 0x269EC3u, rm = 1, rc = 0;      // the map of per_sf steps (lcg_jump's loop, the map kept)
                for (uint32_t n = per_sf; n; n >>= 1) { if (n & 1) { rm *= am; rc = rc * am + ac; } ac = (am + 1) * ac; am *= am; }
                if (lane == 0) { nmeta[v * 4] = noise_vc[v]; nmeta[v * 4 + 1] = noise_nd[v]; nmeta[v * 4 + 2] = st; }
                if (u == v) { noise_am = rm; noise_ac = rc; }
            }
        }
        wave_lds_sync();
        if (NOISE) {                                       // this lane's bands: ranks, and the jump to its first noise band's draw
            noise_rk = *(const uint2*)(nrank + u * 128 + l16 * 8);
            uint32_t kfirst = 0xFF;
#pragma unroll
            for (int r = 7; r >= 0; r--) { const uint32_t kk = ((r < 4 ? noise_rk.x : noise_rk.y) >> (8 * (r & 3))) & 0xFF; kfirst = kk != 0xFF ? kk : kfirst; }
            noise_mine = nmeta[u * 4 + 1] > 0 && kfirst != 0xFF;
            const LcgMap jp = HCA_LCG_POW.v[(kfirst & 0x7F) + 1];
            noise_ja = jp.x; noise_jc = jp.y;
            noise_cur = nmeta[u * 4 + 2]; noise_vcu = nmeta[u * 4];
        }
        if (JOINT) {
            if (F.bands_per_hfr_group > 0) {               // HFR scale of every reconstructed band (hca.cpp:1638-1683), as tr_setup_frame
                const int start = (int)(F.stereo_bands + F.base_bands), groups = (int)F.hfr_group_count;
#pragma unroll
                for (uint32_t v = 0; v < 4; v++) {
                    if (F.type(chan_of(v)) == CRI_CH_SECONDARY) continue;
                    const bool pair = v % C + 1 < CN && F.type(chan_of(v) + 1) == CRI_CH_SECONDARY;      // the next unit takes these bands from this one
                    const uint8_t* sb = sfb + v * 128;
#pragma unroll
                    for (int hh = 0; hh < 2; hh++) {
                        const int k = (int)lane + 64 * hh, b = (start + k) & 127;
                        int sc = (int)sb[(128 - groups + hgrp[b]) & 127] - (int)sb[hlow[b]] + 63;
                        sc = sc & ~(sc >> 31);
                        if (k < nproc) {
                            const float hc = conv[sc & 127];
                            hconv[v * 128 + b] = hc;
                            if (pair && F.stereo_bands > 0 && (uint32_t)b < F.total_bands) hconv[(v + 1) * 128 + b] = hc;
                        }
                    }
                }
            }
            if (lane < 32) {                               // intensity ratio per (unit, subframe): hca.cpp:1361-1441, 1696-1714
                const uint32_t v = lane >> 3, cv = chan_of(v), tv = F.type(cv);
                const uint32_t cs = (tv == CRI_CH_SECONDARY || v % C + 1 >= CN) ? cv : cv + 1;
                bool live; const uint32_t f = unit_frame(v, cur_step, live);
                const uint8_t iv = intensity_walk_back(F, rec0, f, CT, cs, lane & 7, (uint8_t)p.ib);
                ratio[lane] = (F.stereo_bands > 0 && (tv == CRI_CH_SECONDARY || tv == CRI_CH_PRIMARY)) ? iratio[iv & 15] : 1.0f;
            }
            wave_lds_sync();
        }
        return true;
    };
    // row (0, c) of this lane's unit's quantised lines in the frame of step s (subframe sf is sf * C * 4 quarters on), and where
    // this lane's bands l16*8 .. +7 sit in a row: 16 bytes of int16 (quarter l16 / 4) or 8 bytes of int8 (quarter l16 / 8)
    // (as a 32-bit offset from the tile of the frame before the run's first one: the run and its halo span at most two tiles)
    const uint32_t g_run = st.first_frame + f0 - (f0 > 0 ? 1 : 0);
    const uint8_t* rec_run = a.scratch + a.qc_offset + (uint64_t)(g_run >> 6) * HCA_QC_TILE(CT);
    auto row0 = [&](int s) {
        bool live; const uint32_t f = unit_frame(u, s, live);
        const uint32_t g = st.first_frame + f;
        return ((g >> 6) - (g_run >> 6)) * HCA_QC_TILE(CT) + (g & 63) * 64 + HCA_QC_ROW(CT, 0, cc);
    };
    auto lane_off = [&](bool narrow) { return narrow ? (l16 >> 3) * HCA_QC_QUARTER + (l16 & 7) * 8 : (l16 >> 2) * HCA_QC_QUARTER + (l16 & 3) * 16; };
    // (step_narrow: all four units' frames are int8 -- wave-uniform; mine: this lane's is)
    auto lines_to_spectra = [&](const uint4& q, bool step_narrow, bool step_wide, bool mine, uint32_t sf, f2 x[4]) {
        const float4 g0 = *(const float4*)(G + u * 128 + l16 * 8), g1 = *(const float4*)(G + u * 128 + l16 * 8 + 4);
        const f2 gg[4] = {f2{g0.x, g0.y}, f2{g0.z, g0.w}, f2{g1.x, g1.y}, f2{g1.z, g1.w}};
        const uint32_t qw[4] = {q.x, q.y, q.z, q.w};
        if (NW && __builtin_expect(step_narrow, 1)) {
#pragma unroll
            for (int k = 0; k < 4; k++)                        // gains are 0 past the coded bands
                x[k] = gg[k] * (k & 1 ? f2{cvt_f32_i8<2>(qw[k >> 1]), cvt_f32_i8<3>(qw[k >> 1])} : f2{cvt_f32_i8<0>(qw[k >> 1]), cvt_f32_i8<1>(qw[k >> 1])});
        } else if (!NW || __builtin_expect(step_wide, 1)) {    // all four units' frames have int16 lines (material with high-resolution bands)
#pragma unroll
            for (int k = 0; k < 4; k++) x[k] = gg[k] * f2{cvt_f32_i16<0>(qw[k]), cvt_f32_i16<1>(qw[k])};
        } else {                                           // narrow and wide frames among the four (rare): either form per lane
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t w = qw[k >> 1] >> (16 * (k & 1));
                const int a0 = NW && mine ? (int)(int8_t)(w & 0xFF) : (int)(int16_t)(qw[k] & 0xFFFF), a1 = NW && mine ? (int)(int8_t)((w >> 8) & 0xFF) : ((int)qw[k] >> 16);
                x[k] = gg[k] * f2{(float)a0, (float)a1};
            }
        }
        if (JOINT) {
            float* srow = S + u * 128;
            *(float4*)(srow + l16 * 8) = make_float4(x[0].x, x[0].y, x[1].x, x[1].y);
            *(float4*)(srow + l16 * 8 + 4) = make_float4(x[2].x, x[2].y, x[3].x, x[3].y);
            wave_lds_sync();
            if (NOISE) {                                   // reconstruct_noise, hca.cpp:1602-1635: rank k of (subframe, channel) takes draw k + 1
                if (__any(noise_mine)) {
                    float own[8] = {x[0].x, x[0].y, x[1].x, x[1].y, x[2].x, x[2].y, x[3].x, x[3].y};
                    uint32_t rcur = noise_ja * noise_cur + noise_jc;
                    bool started = false;
                    const uint8_t* sfu = sfb + u * 128;
#pragma unroll
                    for (int r = 0; r < 8; r++) {
                        const uint32_t bnd = l16 * 8 + r, kk = ((r < 4 ? noise_rk.x : noise_rk.y) >> (8 * (r & 3))) & 0xFF;
                        const bool isn = noise_mine && kk != 0xFF;
                        const uint32_t rnext = rcur * 0x343FDu + 0x269EC3u;
                        rcur = (isn && started) ? rnext : rcur;
                        started = started || isn;
                        const uint32_t vi = vlist[u * 128 + ((noise_vcu - 1 - (((rcur & 0x7FFF) * noise_vcu) >> 15)) & 127)];
                        int sc = (int)sfu[bnd] - (int)sfu[vi & 127] + 62;
                        sc = sc & ~(sc >> 31);
                        const float nv = conv[sc & 127] * srow[vi & 127];
                        own[r] = isn ? nv : own[r];
                    }
                    wave_lds_sync();                       // (coded bands are never rewritten: the reads above saw coded values)
                    *(float4*)(srow + l16 * 8) = make_float4(own[0], own[1], own[2], own[3]);
                    *(float4*)(srow + l16 * 8 + 4) = make_float4(own[4], own[5], own[6], own[7]);
                    wave_lds_sync();
                }
                noise_cur = noise_am * noise_cur + noise_ac;           // the unit's next subframe
            }
            const float rl = ratio[u * 8 + sf];
            const float rm = (F.type(cc) == CRI_CH_SECONDARY) ? 2.0f - rl : rl;
            const float4 h0 = *(const float4*)(hconv + u * 128 + l16 * 8), h1 = *(const float4*)(hconv + u * 128 + l16 * 8 + 4);
            const f2 hm[4] = {f2{h0.x, h0.y}, f2{h0.z, h0.w}, f2{h1.x, h1.y}, f2{h1.z, h1.w}};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const f2 sv = {*(const float*)((const uint8_t*)S + src_off[2 * k]), *(const float*)((const uint8_t*)S + src_off[2 * k + 1])};
                const f2 mr = {(ratio_mask >> (2 * k)) & 1u ? rm : 1.0f, (ratio_mask >> (2 * k + 1)) & 1u ? rm : 1.0f};
                x[k] = (sv * hm[k]) * mr;
            }
        }
    };

    float prev[4] = {0.0f, 0.0f, 0.0f, 0.0f};              // hca.cpp:962: the overlap tail starts as zeros
    PlainPre pre = load_pre(-1);
    uint32_t rows = row0(-1);                              // rows of the current step's frame
    uint4 q = make_uint4(0, 0, 0, 0);
    {   // (the one load of a run that waits for a flag first)
        const uint8_t* p = rec_run + (rows + HCA_QC_ROW(CT, 7, 0) + lane_off(my_narrow(pre)));
        if (NW && all_narrow(pre)) { const uint2 t = *(const uint2*)p; q.x = t.x; q.y = t.y; } else q = *(const uint4*)p;
    }
    const uint32_t last_count = unit_count(3);             // frames of the last group: steps below it have every group at work
    // A pass's PCM leaves one pass late: its 16-byte store is issued right BEFORE the next pass's line load, so that the wait for
    // those lines (the counter retires in order) finds a store that has had a whole DCT to complete, not one issued just before it.
    uint32_t pend_n00 = 0xFFFFFFFFu;                       // first sample of the staged pass's group-0 subframe (wave-uniform), or none
    auto flush_pcm = [&]() {
        if (pend_n00 == 0xFFFFFFFFu) return;
        typedef uint32_t u4u __attribute__((ext_vector_type(4), aligned(4)));
        const uint32_t gk = lane / (16 * C), wi = 4 * lane - gk * 64 * C;
        if (WIDE) {                                        // the workgroup's 256 * CT bytes: 16 per thread
            if (threadIdx.x * 16 < 256 * CT) {
                const uint4 vw = ((const uint4*)pcmw)[threadIdx.x];
                *(u4u*)((uint32_t*)(dst + (uint64_t)(pend_n00 - st.delay) * CT * 2) + 4 * threadIdx.x) = u4u{vw.x, vw.y, vw.z, vw.w};
            }
            pend_n00 = 0xFFFFFFFFu;
            return;
        }
        const uint4 v = ((const uint4*)pcm)[lane];
        *(u4u*)((uint32_t*)(dst + (uint64_t)(pend_n00 - st.delay) * C * 2) + gk * h * 512 * C + wi) = u4u{v.x, v.y, v.z, v.w};
        pend_n00 = 0xFFFFFFFFu;
    };
    // step -1 is the halo: the subframe before each group's first one (its frame's subframe 7) only feeds the overlap state
#pragma unroll 1
    for (int s = -1; s < (int)h; s++) {
        const uint32_t next_rows = s + 1 < (int)h ? row0(s + 1) : rows;
        bool step_narrow, step_wide, mine; uint32_t loff;
        {
            const PlainPre cur = pre;
            step_narrow = all_narrow(cur); step_wide = none_narrow(cur); mine = my_narrow(cur); loff = lane_off(mine);
            if (s + 1 < (int)h) pre = load_pre(s + 1);
            wave_lds_sync();                               // (the previous pass has read G)
            if (!setup(cur, s)) { flush_pcm(); return; }                   // (a group's halo frame is one of the run's own, except the first group's)
        }
#pragma unroll 1
        for (uint32_t sf = s < 0 ? 7 : 0; sf < 8; sf++) {
            f2 x[4];
            lines_to_spectra(q, step_narrow, step_wide, mine, sf, x);
            // the next pass's lines, requested as soon as this pass's are in registers as floats, i.e. a whole DCT ahead of their use
            // (the next step's first row is laid out by that frame's own flag, which came with `pre` seven passes ago)
            flush_pcm();                                   // the previous pass's PCM (staged in LDS since)
            {   // (8 bytes per lane when the whole step reads int8 lines: 16 would reach into the unused half of the row's 256 B)
                const uint8_t* p; bool ld8;
                if (sf < 7) { p = rec_run + (rows + HCA_QC_ROW(CT, sf + 1, 0) + loff); ld8 = NW && step_narrow; }
                else {                                     // the next step's first row, laid out by that frame's own flag (it came with `pre` seven passes ago)
                    const bool more = s + 1 < (int)h;
                    p = rec_run + (next_rows + (more ? lane_off(my_narrow(pre)) : loff)); ld8 = NW && (more ? all_narrow(pre) : step_narrow);
                }
                if (ld8) { const uint2 t = *(const uint2*)p; q.x = t.x; q.y = t.y; } else q = *(const uint4*)p;
            }
            __builtin_amdgcn_sched_barrier(0);             // (keep the load here: the compiler would sink it behind most of the DCT)
            dct4_inplace(x, L);
            if (WIDE) __syncthreads();                     // (every wave has taken the previous pass's PCM out of the shared piece)
            if (s < 0) {
                bool live; unit_frame(u, -1, live);
#pragma unroll
                for (int j = 0; j < 4; j++) prev[j] = live ? x[j].x : 0.0f;
                continue;
            }
            // window + overlap-add (hca.cpp:1987-1992), PCM16 (hca.cpp:339-360)
            f2 o[4];
            float big = 0.0f;
            const float4 w0 = *(const float4*)wtab, w1 = *(const float4*)(wtab + 4);
            const float wa[4] = {w0.x, w0.z, w1.x, w1.z}, wb[4] = {w0.y, w0.w, w1.y, w1.w};
            const uint4 pov = *(const uint4*)potab, prv = *(const uint4*)(potab + 4);
            const uint32_t po[4] = {pov.x, pov.y, pov.z, pov.w}, pr[4] = {prv.x, prv.y, prv.z, prv.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float dy = x[j].y, pv = prev[j];
                const f2 t = f2{wa[j], wb[j]} * f2{dy, dy};          // w[63-k]*d[127-k], w[64+k]*d[127-k]
                const f2 r = f2{wb[j], wa[j]} * f2{pv, pv};          // w[64+k]*prev[k],  w[63-k]*prev[k]
                const f2 wv = pk_add_neg_hi(t, r);                 // wave[sf][63 - k], wave[sf][64 + k]
                o[j] = wv * f2{32768.0f, 32768.0f};
                if (FLT) {
                    bool live; const uint32_t ff = unit_frame(u, s, live);
                    if (live && ch_live) {
                        const uint32_t k = ((j < 2 ? dlogp.x : dlogp.y) >> (16 * (j & 1))) & 0xFF;
                        float* fo = a.float_out + st.float_offset + ((uint64_t)ff * 1024 + sf * 128) * CT + cc;
                        fo[(uint64_t)(63 - k) * CT] = wv.x; fo[(uint64_t)(64 + k) * CT] = wv.y;
                    }
                }
                prev[j] = x[j].x;
                big = __builtin_fmaxf(big, __builtin_fmaxf(__builtin_fabsf(o[j].x), __builtin_fabsf(o[j].y)));
            }
            // (|value| >= 2^31 needs x86 cvttss2si semantics; values here are finite: gains and lines are)
            if (__builtin_expect(__any(!(big < 2147483648.0f)), 0)) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    float ox = o[j].x, oy = o[j].y;
                    asm volatile("" : "+v"(ox), "+v"(oy));          // keep this path's arithmetic inside the branch
                    int32_t q0 = cvt_trunc_x86(ox), q1 = cvt_trunc_x86(oy);
                    q0 = q0 > 32767 ? 32767 : (q0 < -32768 ? -32768 : q0);
                    q1 = q1 > 32767 ? 32767 : (q1 < -32768 ? -32768 : q1);
                    *(lds_u16*)(uintptr_t)pr[j] = (uint16_t)(int16_t)q0;
                    *(lds_u16*)(uintptr_t)po[j] = (uint16_t)(int16_t)q1;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    typedef short s2 __attribute__((ext_vector_type(2)));
                    const s2 pk = __builtin_amdgcn_cvt_pk_i16((int32_t)o[j].x, (int32_t)o[j].y);     // saturating
                    *(lds_u16*)(uintptr_t)pr[j] = (uint16_t)pk.x;
                    *(lds_u16*)(uintptr_t)po[j] = (uint16_t)pk.y;
                }
            }
            if (WIDE) __syncthreads(); else wave_lds_sync();
            // 256*C contiguous bytes per group; delay / length trim of hca.cpp:3392-3425
            const uint32_t n00 = (f0 + (uint32_t)s) * 1024 + sf * 128;   // first sample (per channel) of group 0's subframe
            if (dword_ok && (uint32_t)s < last_count && n00 >= st.delay && n00 + (NG - 1) * h * 1024 + 128 - st.delay <= st.samples) {
                // one 16-byte store per lane, issued by the next pass (flush_pcm): the staging area is the groups' 256*C-byte pieces back
                // to back (the output address is only dword-aligned: the WAV header is 44 bytes)
                pend_n00 = n00;
            } else if (WIDE) {                             // (a frame cut by the delay or the length trim, or an odd alignment)
                const uint32_t n0 = (f0 + (uint32_t)s) * 1024 + sf * 128;
                for (uint32_t e = threadIdx.x; e < 128 * CT; e += blockDim.x) {
                    const uint32_t n = n0 + e / CT;
                    if (n >= st.delay && n - st.delay < st.samples) *(uint16_t*)(dst + ((uint64_t)(n - st.delay) * CT + e % CT) * 2) = pcmw[e];
                }
            } else {
#pragma unroll 1
                for (uint32_t k = 0; k < 4; k++) {
                    const uint32_t gk = k / C;                           // group of dwords k*64 .. k*64+63
                    bool live; const uint32_t f = unit_frame(gk * C, s, live);
                    if (!live) continue;
                    const uint32_t n0 = f * 1024 + sf * 128;
                    const uint32_t dw = (k - gk * C) * 64 + lane;        // dword within the group's 64*C
                    const uint32_t word = ((const uint32_t*)pcm)[k * 64 + lane];
                    const uint32_t e0 = 2 * dw, e1 = e0 + 1;
                    const uint32_t na = n0 + e0 / C, nb = n0 + e1 / C;
                    const bool va = na >= st.delay && na - st.delay < st.samples, vb = nb >= st.delay && nb - st.delay < st.samples;
                    const uint64_t oa = ((uint64_t)(na - st.delay) * C + e0 % C) * 2, ob = ((uint64_t)(nb - st.delay) * C + e1 % C) * 2;
                    if (va && vb && dword_ok) *(uint32_t*)(dst + oa) = word;
                    else {
                        if (va) *(uint16_t*)(dst + oa) = (uint16_t)word;
                        if (vb) *(uint16_t*)(dst + ob) = (uint16_t)(word >> 16);
                    }
                }
            }
            if (WIDE) __syncthreads(); else wave_lds_sync();
        }
        rows = next_rows;
    }
    flush_pcm();
}

#define HCA_PLAIN_LDS HCA_PLAIN_LDS_BYTES
#define HCA_PLAIN_JOINT_LDS HCA_PLAIN_JOINT_LDS_BYTES
#define HCA_PLAIN_NOISE_LDS (HCA_PLAIN_JOINT_LDS + 512 + 512 + 64)
size_t hca_transform_lds_bytes(uint32_t C, bool plain) {
    const size_t base = (size_t)C * 128 * 4 + (C > 4 ? 16 : 8) * TR_DSTRIDE * 4 + (C > 4 ? C * 512 : 1024) + 512 + 256 + 64 + 80;
    return plain ? base : base + (size_t)C * 128 * 4 + 2048 + 512 + 64 + C * 128 + 128 + 128 + ((C * 8 + 15) & ~15) + (C * 4 + 4) * 4 + 2 * C * 128 + 64;
}

// Which transform kernel a format group takes (also reported to callers: cri_job_hca_groups)
__host__ __device__ uint32_t hca_transform_form(const HcaDecArgs& a) {
    const bool in_regs = a.channels <= 8 && (a.plain || a.inlane || a.channels == 1 || a.channels == 2 || a.channels == 4 || ((a.channels == 6 || a.channels == 8) && a.pairs_even));
    if (!in_regs) return HCA_TR_GENERIC;
    const bool small = a.channels == 1 || a.channels == 2 || a.channels == 4;
    if (a.plain) return small ? HCA_TR_INLANE_PLAIN : (HCA_TR_INLANE_PLAIN | HCA_TR_WIDE);       // 3, 5, 6, 7, 8 channels: a wave per four channels, whole sample frames out
    if (a.inlane) {
        const bool wide = a.channels == 3 || a.channels > 4 || !a.pairs_even;                      // the wide joint form: a wave per group of channels
        return (a.noise_fill ? HCA_TR_INLANE_NOISE : HCA_TR_INLANE_JOINT) | (wide ? HCA_TR_WIDE : 0u);
    }
    return HCA_TR_GENERAL;
}

void launch_hca_transform(const HcaDecArgs& a, hipStream_t s) {
    if (!a.frames) return;
    if (a.noise_fill) hipLaunchKernelGGL(k_hca_noise_scan, dim3(a.stream_end - a.stream_begin), dim3(64), 0, s, a);
    const uint32_t form = hca_transform_form(a);
    if (form == HCA_TR_GENERIC) {
        size_t lds = (size_t)a.channels * (2 * 128 + 2 * TR_DSTRIDE) * 4 + a.channels * 256 + ((a.channels * 8 + 15) & ~15u) + a.channels * (8 + 256) + 16;
        hipLaunchKernelGGL(k_hca_transform_generic, dim3(a.frames), dim3(64), lds, s, a);
        return;
    }
    const size_t lds = hca_transform_lds_bytes(a.channels, a.plain != 0);
    const bool flt = a.float_out != nullptr;
    const uint32_t nruns = a.run_count ? a.run_count : a.runs;
    const uint32_t nw = a.wide_waves;
#define CRI_LAUNCH_TR(P, CH) do { if (flt) hipLaunchKernelGGL((k_hca_transform<P, CH, true>), dim3(nruns), dim3(64), lds, s, a); \
                                  else hipLaunchKernelGGL((k_hca_transform<P, CH, false>), dim3(nruns), dim3(64), lds, s, a); } while (0)
#define CRI_LAUNCH_IL(CH, JOINT, WIDE, NOISE, THREADS, BYTES) do { if (flt) hipLaunchKernelGGL((k_hca_transform_plain<CH, true, JOINT, WIDE, NOISE>), dim3(nruns), dim3(THREADS), BYTES, s, a); \
                                                                   else hipLaunchKernelGGL((k_hca_transform_plain<CH, false, JOINT, WIDE, NOISE>), dim3(nruns), dim3(THREADS), BYTES, s, a); } while (0)
#define CRI_LAUNCH_124(JOINT, NOISE, BYTES) switch (a.channels) { case 1: CRI_LAUNCH_IL(1, JOINT, false, NOISE, 64, BYTES); break; \
                                                                  case 2: CRI_LAUNCH_IL(2, JOINT, false, NOISE, 64, BYTES); break; default: CRI_LAUNCH_IL(4, JOINT, false, NOISE, 64, BYTES); break; }
    switch (form) {
        case HCA_TR_INLANE_PLAIN: CRI_LAUNCH_124(false, false, HCA_PLAIN_LDS); break;
        case HCA_TR_INLANE_JOINT: CRI_LAUNCH_124(true, false, HCA_PLAIN_JOINT_LDS); break;
        case HCA_TR_INLANE_NOISE: CRI_LAUNCH_124(true, true, HCA_PLAIN_NOISE_LDS); break;
        case HCA_TR_INLANE_PLAIN | HCA_TR_WIDE: CRI_LAUNCH_IL(4, false, true, false, 64 * nw, nw * HCA_PLAIN_LDS + 2048 + 128 + 64); break;
        case HCA_TR_INLANE_JOINT | HCA_TR_WIDE: CRI_LAUNCH_IL(4, true, true, false, 64 * nw, nw * HCA_PLAIN_JOINT_LDS + 2048 + 128 + 64); break;
        case HCA_TR_INLANE_NOISE | HCA_TR_WIDE: CRI_LAUNCH_IL(4, true, true, true, 64 * nw, nw * HCA_PLAIN_NOISE_LDS + 2048 + 128 + 64); break;
        default:                                           // the general in-register transform: spectra assembled through LDS
            switch (a.channels) {
                case 1: CRI_LAUNCH_TR(false, 1); break; case 2: CRI_LAUNCH_TR(false, 2); break; case 4: CRI_LAUNCH_TR(false, 4); break;
                case 6: CRI_LAUNCH_TR(false, 6); break; default: CRI_LAUNCH_TR(false, 8); break;
            }
    }
#undef CRI_LAUNCH_TR
#undef CRI_LAUNCH_IL
#undef CRI_LAUNCH_124
}

}  // namespace cri
