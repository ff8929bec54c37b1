// cri_types.h -- plain structs shared by the host planner (cri_host.cpp / cri_capi.cpp) and the HIP kernels
// (cri_kernels.hip).  Everything here is POD and is uploaded to HBM verbatim.
#pragma once

#include <stdint.h>

enum { CRI_CH_DISCRETE = 0, CRI_CH_PRIMARY = 1, CRI_CH_SECONDARY = 2 };

// ---- HCA ---------------------------------------------------------------------------------------------------
// A "format" is everything in an HCA header that shapes the per-frame work (hca.cpp:80-174 clHCA header config).
// Streams of one batch that share a format are decoded/encoded by the same launches.
struct HcaFormat {
    uint32_t channels, version, frame_size, min_res, max_res;
    uint32_t total_bands, base_bands, stereo_bands, bands_per_hfr_group, hfr_group_count;
    uint32_t ath_index;            // row of the ATH table array (row 0 is all zero = ath type 0)
    uint32_t record_bytes;         // bytes of one frame's intermediate record in scratch (decode)
    uint32_t hfr_band_count;       // encoder only (hca.cpp:2275)
    uint32_t pad0;
    uint8_t type[16];              // channel types (hca.cpp:887-970)
    uint8_t coded[16];             // coded band count per channel (<=128)
};

struct HcaStream {
    uint64_t src_offset;           // decode: first frame byte in the input blob; encode: first PCM byte
    uint64_t dst_offset;           // decode: first PCM byte in the output blob; encode: first frame byte
    uint64_t scratch_offset;       // decode: first frame record of this stream in scratch
    uint32_t format;               // index into the format array
    uint32_t cipher;               // index into the cipher table array
    uint32_t frames;               // frames to process
    uint32_t delay;                // decode: leading samples to drop (encoder_delay)
    uint32_t samples;              // decode: samples per channel to emit; encode: input samples per channel
    uint32_t item;                 // index of the batch item (for status reporting)
    uint32_t src_in_scratch;       // encode: src_offset is relative to the job scratch (converted PCM16)
    uint32_t first_frame;          // global frame number of this stream's frame 0 within its format group
    uint32_t first_run;            // global number of this stream's first run (HcaDecArgs.run_frames frames) within its format group
    // encode, looping input (hca.cpp:2990-3107): the encoder's input is the sequence
    //   enc_pre_zero zeros | first sample up to enc_pre | `samples` of main audio | enc_post samples from enc_loop_src | zeros
    uint32_t enc_loop;             // 1: use the sequence above (0: main audio then zeros)
    uint32_t enc_pre_zero, enc_pre, enc_post;
    uint32_t enc_loop_src;         // first source sample of the post audio (the loop start)
    uint32_t enc_loop_src_end;     // source samples at or past this index read as zero in the post audio
    uint32_t enc_have;             // samples per channel actually present in the WAV data
    uint64_t float_offset;         // decode: first float of this stream in the validation output (HcaDecArgs::float_out), in floats
    uint32_t pad3;
};

// Between k_hca_parse and the transform kernels a frame's unpacked state travels through scratch in two places.
//
// 1. Frame record (per frame, consecutive in frame order within a format group; HcaStream::scratch_offset = the stream's first):
//      uint8 scalefactors[C][128] | uint8 intensity[C][8] | uint32 tail[4] | (from the next 64-byte boundary) uint8 code descriptions[C][128]
//      tail = { packed_noise_level, status (0 or CRI_ERR_HCA_FRAME), flags (bit c: channel c reuses intensity[1..7];
//               HCA_REC_NARROW), generator draws (v3.0 noise fill; k_hca_noise_scan turns it into a prefix) }
//      code descriptions = band_meta of every band (cri_hca_dec.hip), for formats the in-lane transform takes: the parse keeps them
//      tile-major for its own eight passes over them (HcaDecArgs::resg_offset) and copies them here, so that a transform wave finds a
//      frame's 128 bytes per channel in one line instead of sixteen 16-byte pieces of sixteen lines it shares with seven other frames
//      (tools/traffic_census.py: 2.6 KB of lines per stereo frame for 0.33 KB used -> 0.4 KB)
//    (an odd number of 64-byte lines: k_hca_parse stores the same 64 B of 16 consecutive records per instruction, and with an
//     even line stride those would land on a fraction of the L2 channels)
#define HCA_REC_SF(C, c) ((c) * 128u)
#define HCA_REC_INT(C, c) ((C) * 128u + (c) * 8u)
#define HCA_REC_TAIL(C) ((C) * 136u)
#define HCA_REC_DESC(C, c) ((((C) * 136u + 16u + 63u) & ~63u) + (c) * 128u)
static inline uint32_t hca_record_bytes(uint32_t channels) { return (((HCA_REC_DESC(channels, channels) + 63u) >> 6) | 1u) << 6; }
#define HCA_REC_NARROW 0x40000000u   // tail flags: the frame's quantised lines are int8 and NEGATED (formats with HcaDecArgs::narrow
                                     // only: k_hca_parse -> k_hca_transform_plain)
//
// 2. Quantised lines, tile-major (a tile = the 64 consecutive frames one parse wave owns; HcaDecArgs::qc_offset = the group's
//    first tile):  [tile][subframe 8][channel C][quarter 4][frame 64][64 B]
//    A row of 128 lines is 256 B of int16 (four quarters of 32 lines) or, for HCA_REC_NARROW frames, 128 B of int8 (quarters 0
//    and 1, 64 lines each; 2 and 3 unused).  What a parse wave has ready at a flush -- the same 64 B of a row for each of its
//    64 frames -- is therefore 4 KB of contiguous memory (four fully coalesced 1 KB stores) instead of 64 pieces a record apart,
//    and a transform lane still finds its 8 lines in one 8- or 16-byte piece.
#define HCA_QC_QUARTER 4096u                                          /* 64 frames x 64 B */
#define HCA_QC_ROW(C, sf, c) ((((sf) * (C)) + (c)) * (4u * HCA_QC_QUARTER))   /* row (sf, c), quarter 0, frame 0, inside a tile */
#define HCA_QC_TILE(C) (8u * (C) * 4u * HCA_QC_QUARTER)              /* bytes of a tile: 2 KB per frame and channel */

// Encoder tables as k_hca_encode keeps them in LDS: one blob, built on the host (hca_enc_build_tables), byte offsets below
#define HCA_ET_TW 0           // float2[128]  {cos, sin} twiddles, only the entries the lane mapping reads: row 7 [0,64), row 5 [64,96), 4 [96,112), 3 [112,120), 2 [120,124), 1 [124,126), 0 [126]
#define HCA_ET_DEQ 1024       // float[72]    scale-factor table padded with NaN
#define HCA_ET_ESCALE 1312    // float[64]
#define HCA_ET_CP 1568        // uint2[60]    per curve position {(16 - rank) in every byte, 8 * shortest | resolution << 20 | anomaly << 28}; [59] = a band that costs nothing.  (The second word is
                              //              summed as it is over bands, lanes and channels by the rate loop: the bits are the low 20 bits of that sum -- at most 8 x 128 x 96)
#define HCA_ET_INV 2048       // float[16]    quantiser inverse step per resolution
#define HCA_ET_IBOUNDS 2112   // float[16]
#define HCA_ET_SFBASE 2176    // uint8[32]
#define HCA_ET_ISHUF 2208     // uint8[128]
#define HCA_ET_CLEN 2336      // uint8[256]   code length by resolution * 16 + quantiser index; rows 8 .. 15 (sign-magnitude codes): resolution - 4, the length of a zero
#define HCA_ET_CODE 2592      // uint8[256]   code; rows 8 .. 15: 0
#define HCA_ET_WIN4 2848      // float4[8][8] the MDCT window: [register r][lane & 7] = the four factors of the point's two folded inputs (hca.cpp:2529-2553), times 2^-15, with
                              //              the fold's signs: {even a, odd a, even b, odd b}
#define HCA_ET_LDS_BYTES 3872 // ... up to here the blob is copied into LDS; what follows is read from memory (every lane its own row: the L1 holds the 0.5 KB that are ever hit)
#define HCA_ET_CLS 3872       // uint2[768]   row (x >> 22: sign, exponent field, top mantissa bit -- a half-binade), |x| < 1: {A, classes below}: class = base + (|x| >= A).  Rows
                              //              256 .. 511 (exponents no |x| < 1 has) are never read
#define HCA_ET_BYTES 10016    // HCA_ET_CLS + 768 * 8
#define HCA_ENC_CLAMP_BITS 0x3F7FFFFEu   // ScaleSpectra's clamp 0.9999999f (hca.cpp:2639-2654): the one value the quantiser can push past its table


// ---- ADX ---------------------------------------------------------------------------------------------------
struct AdxStream {
    uint64_t src_offset;           // decode: first block byte; encode: first PCM byte
    uint64_t dst_offset;           // decode: first PCM byte; encode: first block byte
    uint64_t src_end;              // decode: one past the last readable input byte of this item
    uint32_t frames;               // block rows (one block per channel each)
    uint32_t channels, blocksize, bitdepth, mode, samples_per_block;
    int32_t coef0, coef1;
    uint32_t samples;              // decode: samples per channel to emit; encode: valid input samples per channel
    uint32_t filter_bits;          // encode, mode 2: filter << 13
    uint32_t hist_offset;          // index of this stream's first entry in the per-chain history array
    uint32_t item;
    uint32_t first_chain;          // global chain number of channel 0
    uint32_t stale_offset, stale_len; // encode: header image bytes that overlap the block area (OR-ed into first block bytes)
    uint32_t src_in_scratch;       // encode: src_offset is relative to the job scratch (converted PCM16), not to the input blob
    // segmented chains (k_adx_seg_*, cri_adx.hip): the stream's block rows are cut into seg_count segments of seg_rows rows; a
    // segment is decoded speculatively from a warm-up of warm_rows rows before it.  first_seg = global number of the stream's
    // first (segment, channel) lane; rows_avail = block rows the input fully contains (decode)
    uint32_t seg_rows, seg_count, warm_rows, first_seg, rows_avail;
};
