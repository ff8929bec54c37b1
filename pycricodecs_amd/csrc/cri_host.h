// cri_host.h -- host-side (per-file, once) parts of the ADX / HCA path: container headers, WAV glue,
// coefficient / cipher-table / encoder-parameter derivation.  The per-frame and per-block work is NOT here;
// it lives in cri_kernels.hip.  Reference locations are cited at each definition in cri_host.cpp.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <vector>
#include "cri_types.h"

namespace cri {

// ---- byte order helpers
inline uint32_t be16(const uint8_t* p) { return ((uint32_t)p[0] << 8) | p[1]; }
inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline uint32_t le16(const uint8_t* p) { return p[0] | ((uint32_t)p[1] << 8); }
inline uint32_t le32(const uint8_t* p) { return p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline void put_be16(uint8_t* p, uint32_t v) { p[0] = (uint8_t)(v >> 8); p[1] = (uint8_t)v; }
inline void put_be32(uint8_t* p, uint32_t v) { put_be16(p, v >> 16); put_be16(p + 2, v & 0xFFFF); }
inline void put_le16(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
inline void put_le32(uint8_t* p, uint32_t v) { put_le16(p, v & 0xFFFF); put_le16(p + 2, v >> 16); }

uint16_t crc16(const uint8_t* p, size_t n);

// ---- WAV
struct WavInfo {
    uint32_t channels = 0, rate = 0, block_align = 0, bitdepth = 0, mode = 0;  // mode 1 = PCM, 3 = IEEE float
    uint64_t data_offset = 0;      // offset of the sample data inside the file
    uint32_t data_size = 0;
    bool looping = false;
    uint32_t num_loops = 0;
    std::vector<uint32_t> loop_start, loop_end;
    uint32_t column_size = 0;      // total interleaved samples
    uint32_t sample_size = 0;      // bytes per sample
};
int wav_parse(const uint8_t* w, size_t len, WavInfo& o);
// true when the sample data can be consumed as-is as little-endian int16 (pcm.cpp:533-534)
bool wav_is_pcm16(const WavInfo& w);
// 0 when the sample format is one the reference converts to PCM16 (pcm.cpp:530-545), else the PCM error code
int wav_convertible(const WavInfo& w);
// one sample converted the way PCM::Get_PCM16 does (used for the ADX header's initial history only)
int16_t wav_sample16(const WavInfo& w, const uint8_t* file, uint64_t index);
uint32_t wav_write_header(uint8_t* d, uint32_t channels, uint32_t rate, uint32_t samples_per_channel,
                          bool looping, uint32_t loop_start, uint32_t loop_end);

// ---- ADX
void adx_coefficients(uint32_t highpass, uint32_t rate, int32_t coef[2]);
struct AdxHeader {
    uint32_t data_offset, mode, blocksize, bitdepth, channels, rate, sample_count, highpass, version;
    bool looping; uint32_t loop_start, loop_end;
    std::vector<int16_t> history;  // 2 per channel
    uint32_t samples_per_block, blocks; int32_t coef[2];
};
int adx_parse_header(const uint8_t* d, size_t len, AdxHeader& h);
struct AdxEncodePlan {
    uint32_t channels, samples_per_channel, samples_per_block, frames, header_size;
    int32_t coef[2];
    std::vector<int16_t> history;        // initial per-channel history (2 per channel)
    std::vector<uint8_t> image;          // header bytes + any bytes the reference writes past it (see cri_host.cpp)
    uint64_t total_size;
};
int adx_plan_encode(const uint8_t* wav, size_t len, const WavInfo& w, uint32_t bitdepth, uint32_t blocksize, uint32_t mode,
                    uint32_t highpass, uint32_t filter, uint32_t version, bool force_no_loop, AdxEncodePlan& p);

// ---- HCA
struct HcaHeader {
    uint32_t version, header_size, channels, rate, frame_count, delay, padding;
    uint32_t frame_size, min_res, max_res, track_count, channel_config, stereo_type;
    uint32_t total_bands, base_bands, stereo_bands, bands_per_hfr_group, ms_stereo;
    uint32_t ath_type, loop_start_frame, loop_end_frame, loop_start_delay, loop_end_padding, loop_flag;
    uint32_t ciph_type, comment_len, hfr_group_count;
    uint8_t ath[128];
    uint8_t type[16]; uint32_t coded[16];
};
int hca_parse_header(const uint8_t* d, size_t len, uint32_t header_size, HcaHeader& h);
void hca_channel_types(uint32_t channels, uint32_t track_count, uint32_t stereo_bands, uint32_t config, uint8_t t[16]);
int hca_cipher_table(uint32_t type, uint64_t key, uint8_t t[256]);
uint64_t hca_mix_key(uint64_t key, uint16_t subkey);
void hca_crypt_header(uint8_t* d, uint32_t header_size, uint32_t encrypt, uint32_t type);
struct HcaEncSetup {
    uint32_t channels, rate, frame_size, frame_count, delay, padding, channel_config;
    uint32_t total_bands, base_bands, stereo_bands, hfr_group_count, bands_per_hfr_group, hfr_band_count;
    uint32_t header_size, samples_per_channel;     // samples_per_channel: main audio fed to the encoder (clamped to the loop end)
    uint32_t loop_flag, loop_start_frame, loop_end_frame, loop_start_delay, loop_end_padding;
    uint32_t pre_samples, post_samples, loop_start; // BufferPreSamples, PostSamples, first sample of the post audio
    uint8_t type[16]; uint32_t coded[16];
};
void hca_enc_setup_loop(HcaEncSetup& e, uint32_t loop_start, uint32_t loop_end, uint32_t column_size);
int hca_enc_setup(uint32_t channels, uint32_t rate, uint32_t samples_per_channel, uint32_t quality, HcaEncSetup& e);
void hca_pack_header(const HcaEncSetup& e, uint8_t* out);
// the table blob k_hca_encode copies into LDS (layout: HCA_ET_* in cri_kernels.h); 0 or CRI_ERR_INVALID_ARG when the
// code-length tables do not have the one-threshold shape the kernel's rate loop relies on (they do; checked, not assumed)
int hca_enc_build_tables(std::vector<uint8_t>& blob);

}  // namespace cri
