// tests/shim/host_shim.cpp -- TEST INFRASTRUCTURE ONLY.
// Exposes the product's host-side header logic (pycricodecs_amd/csrc/cri_host.cpp: no HIP in it) through a C ABI so that the
// CPU test suite can drive it without a device: the job planner that normally calls these functions (cri_capi.cpp) refuses
// to build a job when no gfx950 device is present.  Built on demand by tests/test_host_logic.py with plain g++.
#include <string.h>
#include "../../pycricodecs_amd/csrc/cri_host.h"

using namespace cri;

extern "C" int shim_wav_parse(const uint8_t* w, size_t len, uint32_t out[10]) {
    WavInfo o;
    const int rc = wav_parse(w, len, o);
    out[0] = o.channels; out[1] = o.rate; out[2] = o.block_align; out[3] = o.bitdepth; out[4] = o.mode;
    out[5] = (uint32_t)o.data_offset; out[6] = o.data_size; out[7] = o.looping ? 1u : 0u; out[8] = o.num_loops; out[9] = o.column_size;
    return rc;
}

extern "C" int shim_hca_parse_header(const uint8_t* d, size_t len, uint32_t size_arg, uint32_t out[32]) {
    HcaHeader h;
    const int rc = hca_parse_header(d, len, size_arg, h);
    const uint32_t f[] = {h.version, h.header_size, h.channels, h.rate, h.frame_count, h.delay, h.padding, h.frame_size, h.min_res, h.max_res,
                          h.track_count, h.channel_config, h.stereo_type, h.total_bands, h.base_bands, h.stereo_bands, h.bands_per_hfr_group,
                          h.ms_stereo, h.ath_type, h.loop_start_frame, h.loop_end_frame, h.loop_start_delay, h.loop_end_padding, h.loop_flag,
                          h.ciph_type, h.comment_len, h.hfr_group_count};
    memset(out, 0, 32 * sizeof(uint32_t));
    memcpy(out, f, sizeof f);
    return rc;
}

extern "C" void shim_hca_crypt_header(uint8_t* d, uint32_t hs, uint32_t encrypt, uint32_t type) { hca_crypt_header(d, hs, encrypt, type); }

extern "C" int shim_adx_parse_header(const uint8_t* d, size_t len, uint32_t out[12]) {
    AdxHeader h;
    const int rc = adx_parse_header(d, len, h);
    if (rc) return rc;
    const uint32_t f[] = {h.data_offset, h.mode, h.blocksize, h.bitdepth, h.channels, h.rate, h.sample_count, h.highpass, h.version,
                          h.samples_per_block, h.blocks, h.looping ? 1u : 0u};
    memcpy(out, f, sizeof f);
    return 0;
}

extern "C" int shim_hca_enc_tables(uint8_t* out, size_t cap) {   // the encoder's LDS table blob (HCA_ET_* layout, cri_types.h)
    std::vector<uint8_t> blob;
    const int rc = hca_enc_build_tables(blob);
    if (rc) return rc;
    if (blob.size() > cap) return -1;
    memcpy(out, blob.data(), blob.size());
    return (int)blob.size();
}
