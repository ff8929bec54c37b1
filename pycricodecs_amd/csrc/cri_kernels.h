// cri_kernels.h -- launch interface between the job planner (cri_capi.cpp) and the gfx950 kernels (cri_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "cri_types.h"

namespace cri {

struct HcaDecArgs {
    const uint8_t* in;             // input blob (device)
    uint8_t* out;                  // output blob (device)
    uint8_t* scratch;              // frame records, quantised lines, band code descriptions
    int32_t* status;               // per item, may be null
    const HcaFormat* formats;
    const HcaStream* streams;      // sorted by format
    const uint8_t* cipher_tables;  // [n_cipher][256]
    const uint8_t* ath_tables;     // [n_ath][128]
    uint32_t format;               // format index of this launch
    uint32_t stream_begin, stream_end;   // streams of this format
    uint32_t frames;               // total frames of this format group
    uint32_t runs;                 // total runs (run_frames consecutive frames of one stream) of this format group
    uint32_t run_frames;           // 8, 16 or 32: frames of a run (the planner's choice by the group's size)
    uint32_t n_cipher;
    uint32_t cipher_identity;      // 1: every stream of the job is unencrypted (one identity table): the intake skips the lookups
    uint32_t pad0;
    // a launch may cover a slice of the group (the pipelined host path, cri_capi.cpp): tiles [tile_begin, tile_begin + tile_count) of the
    // parse, runs [run_begin, run_begin + run_count) of the run-per-wave transforms; counts of 0 = the whole group
    uint32_t tile_begin, tile_count, run_begin, run_count;
    uint32_t channels;             // channel count of this format
    uint32_t plain;                // 1: no HFR and no joint stereo in this format (spectra need dequantisation only)
    uint32_t noise_fill;           // 1: min_resolution == 0 (v3.0): k_hca_noise_scan + noise reconstruction in the transform
    uint32_t pairs_even;           // 1: every stereo pair starts on an even channel (a pair then shares a transform pass)
    uint32_t narrow;               // 1: k_hca_transform_plain reads this format's records: int8 lines where the values allow it
    uint32_t inlane;               // 1: joint-stereo / HFR / noise-fill format that k_hca_transform_plain's joint, noise or wide joint instance takes
    uint32_t wide_waves;           // wide forms: waves per workgroup = groups of up to four consecutive channels that keep every stereo pair together
    uint64_t in_bytes;             // size of the input blob (the intake's 16-byte loads stop there)
    uint64_t qc_offset;            // scratch byte offset of this group's quantised lines (tile-major, cri_types.h)
    uint64_t resg_offset;          // scratch byte offset of this group's band code descriptions: [tile][C][8 blocks][64 lanes] uint4 (16 bands x 1 byte)
    float* float_out;              // validation runs only (cri_job_run_floats), else null: every frame's samples before the int16
                                   // conversion (hca.cpp:1987-1992 wave[][]), [frame][1024][C] floats from HcaStream::float_offset on
};
// transform kernel of a format group: k_hca_transform_generic, k_hca_transform<.> (general, in registers), or an instance of the
// in-lane kernel k_hca_transform_plain (plain / joint stereo + HFR / + v3.0 noise fill), | HCA_TR_WIDE = a wave per four channels
enum { HCA_TR_GENERIC = 0, HCA_TR_GENERAL = 1, HCA_TR_INLANE_PLAIN = 2, HCA_TR_INLANE_JOINT = 3, HCA_TR_INLANE_NOISE = 4, HCA_TR_WIDE = 8 };
__host__ __device__ uint32_t hca_transform_form(const HcaDecArgs& a);
size_t hca_parse_lds_bytes(uint32_t n_cipher);
void launch_hca_parse(const HcaDecArgs& a, hipStream_t s);
void launch_hca_transform(const HcaDecArgs& a, hipStream_t s);

struct HcaEncArgs {
    const uint8_t* in; uint8_t* out; int32_t* status;
    const uint8_t* tables;         // HCA_ET_* blob
    const uint8_t* scratch;        // converted PCM16 (HcaStream::pad0 != 0 -> src_offset is relative to scratch)
    const HcaFormat* formats;
    const HcaStream* streams;      // sorted by format; src_offset = first PCM byte, dst_offset = first frame byte
    const uint16_t* crc_mul;       // [64][16]: (x^bit * x^(16 + 8 * crc_chunk * (63 - lane))) mod P: a lane's chunk remainder times its place in the frame
    const uint32_t* stream_hint;   // stream (index into `streams`) of every 16th frame of the launch
    uint32_t format, stream_begin, stream_end, frames, channels, frame_size;
    uint32_t crc_chunk;            // bytes of the (front-padded) frame each lane checksums: whole words
    uint32_t lds_per_frame;        // LDS bytes of one frame's working set (set by launch_hca_encode)
    uint32_t frames_per_group;     // frames per workgroup (set by launch_hca_encode)
    uint32_t groups;               // frame groups of the launch: the (persistent) workgroups walk them (set by launch_hca_encode)
};
size_t hca_encode_lds_bytes(uint32_t channels, uint32_t frame_size);
void launch_hca_encode(const HcaEncArgs& a, hipStream_t s);

struct AdxArgs {
    const uint8_t* in; uint8_t* out; int32_t* status;
    const uint8_t* scratch;        // converted PCM16 of items whose WAV is not 16-bit (AdxStream::src_in_scratch)
    const AdxStream* streams;
    const uint32_t* chain_stream;  // per chain: stream index
    const int16_t* history;        // 2 per chain
    const uint8_t* stale;          // encode: header-image bytes that spill into the block area
    uint32_t chains;               // incl. padding entries (chain_stream == 0xFFFFFFFF) that keep a file inside one wave
    uint32_t rows_per_round;       // block rows staged in LDS per round (T)
    uint32_t lds_in_bytes, lds_out_bytes;
    const uint32_t* wpf_order;     // wave-per-file kernels: stream of workgroup b (longest files first), or null = b
    // segmented chains: first (segment, channel) lane of every stream (n_streams + 1 entries), the lanes' records
    // {speculative start state, end state, end state after repair, stop row} and one flag word per (stream, channel) chain
    const uint32_t* seg_first; uint32_t n_streams, seg_lanes; uint32_t* seg_state; uint32_t* seg_flags;
    uint32_t seg_max_count;        // decode: the most segments any chain has (long chains get k_adx_seg_runs)
    uint32_t* seg_ckpt;            // encode: the history after every round of four rows, per channel (the output holds codes, not histories)
};
void launch_adx_decode(const AdxArgs& a, hipStream_t s);
void launch_adx_encode(const AdxArgs& a, hipStream_t s);
// wave-per-file variants (blocksize 18, bitdepth 4, <= 2 channels, no header spill); one block per stream
void launch_adx_decode_wpf(const AdxArgs& a, uint32_t n_streams, hipStream_t s);
void launch_adx_encode_wpf(const AdxArgs& a, uint32_t n_streams, hipStream_t s);
// segmented chains (standard layout): speculative decode of all segments, parallel repair, serial repair of flagged chains
void launch_adx_decode_seg(const AdxArgs& a, hipStream_t s);
void launch_adx_encode_seg(const AdxArgs& a, hipStream_t s);
void launch_adx_encode_lane(const AdxArgs& a, hipStream_t s);   // lane per (file, channel, segment): batches of many files

struct CryptArgs {
    const uint8_t* in; uint8_t* out;
    const HcaStream* streams;      // src_offset/dst_offset = first frame, frames, cipher index; format -> frame size table
    const uint32_t* frame_sizes;   // per stream
    const uint8_t* cipher_tables;
    const uint32_t* first_frame;   // prefix over streams, n_streams + 1
    uint32_t n_streams, frames;
    uint32_t max_frame_size;       // largest frame of the job (sizes the wave-per-frame kernel's LDS image)
    const uint16_t* crc_pos;       // per distinct frame size [64][16]: (x^bit * x^(8 * m * (63 - lane))) mod P, m = bytes per lane of the checksum
    const uint32_t* crc_pos_off;   // per stream: its table's first entry
};
void launch_hca_crypt(const CryptArgs& a, hipStream_t s);

struct ConvertItem { uint64_t first, src_offset, dst_offset; uint32_t sample_size, bitdepth, mode, pad; };
struct ConvertArgs { const uint8_t* in; uint8_t* scratch; const ConvertItem* items; uint32_t n_items; uint64_t total; };
void launch_pcm_convert(const ConvertArgs& a, hipStream_t s);

// copies n small byte images (headers) into the output blob: image i = img[img_off[i], img_off[i+1]) -> out + dst_off[i]
void launch_scatter_images(const uint8_t* img, const uint64_t* img_off, const uint64_t* dst_off, uint32_t n, uint8_t* out, hipStream_t s);
// copies item bodies verbatim (crypt: headers are patched on the host image, frames by the kernel)
void launch_fill_i32(int32_t* p, int32_t v, uint32_t n, hipStream_t s);
void launch_fill_i32_bad(int32_t* p, hipStream_t s);
void launch_pull_host(uint8_t* dst, const uint8_t* src_host, uint64_t bytes, hipStream_t s, uint32_t max_wg = 8);

// USM audio (@SFA) chunk streams, usm.py: byte segments copied between a container and contiguous streams, with the audio
// mask (32 bytes, usm.py:112-117) XORed over bytes [mask_begin, mask_end) of a segment
struct Segment { uint64_t src, dst; uint32_t len, mask_begin, mask_end, pad; };
struct SegmentArgs { const uint8_t* in; uint8_t* out; const Segment* segs; uint32_t n; uint32_t mask[8]; };
void launch_segments(const SegmentArgs& a, hipStream_t s);

}  // namespace cri
