// cri_host.cpp -- per-file host logic of the ADX / HCA path (headers, WAV glue, parameter derivation).
// Nothing here touches sample data; all per-frame / per-block work is in cri_kernels.hip.
// Citations are to /root/reference/CriCodecs/<file>:<lines>.
#include "cri_host.h"
#include <math.h>
#include <string.h>
#include "../../include/cricodecs_hip.h"
#define CRI_TABLE_QUAL static const
#include "cri_tables.h"

namespace cri {

static int next_multiple(int value, int multiple) {   // IO.hpp:26-32
    if (multiple <= 0) return value;
    if (value % multiple == 0) return value;
    return value + multiple - value % multiple;
}

uint16_t crc16(const uint8_t* p, size_t n) {           // hca.cpp:205-211
    uint16_t s = 0;
    for (size_t i = 0; i < n; i++) s = (uint16_t)((s << 8) ^ CRI_CRC16_TAB[(s >> 8) ^ p[i]]);
    return s;
}

// ------------------------------------------------------------------------------------------------ WAV
// RIFF chunk walk: pcm.cpp:335-342 (riff), 291-327 (chunks), 177-201 (fmt), 243-261 (smpl), 276-283 (data),
// 419-444 (sample layout).  Every access is bounds-checked against `len` (the reference trusts the RIFF size) and every
// iteration advances by at least 8 bytes (a chunk length of 0xFFFFFFF8..0xFFFFFFFF wraps the reference's 32-bit size).
int wav_parse(const uint8_t* w, size_t len, WavInfo& o) {
    o = WavInfo();
    if (len < 12) return CRI_ERR_PCM(1);
    if (le32(w) != 0x46464952u || le32(w + 8) != 0x45564157u) return CRI_ERR_PCM(1);
    uint32_t fullsize = le32(w + 4), sum = 4, raw_mode = 0, ext_bits = 0, subfmt = 0;
    bool have_fmt = false, have_data = false;
    size_t cur = 12;
    while (sum < fullsize) {
        if (cur + 8 > len) return CRI_ERR_PCM(7);
        uint32_t sig = le32(w + cur), size = le32(w + cur + 4) + 8;
        if (size < 8) return CRI_ERR_PCM(7);                  // the 32-bit add wrapped (the reference then stops advancing: size 0 loops forever)
        size += ((size & 1) && (uint64_t)size + sum + (size & 1) <= fullsize);
        if (sig == 0x20746D66u) {
            uint32_t fsz = le32(w + cur + 4);
            if (fsz < 16) return CRI_ERR_PCM(2);
            if (cur + 24 > len) return CRI_ERR_PCM(7);
            raw_mode = le16(w + cur + 8);
            o.channels = le16(w + cur + 10);
            o.rate = le32(w + cur + 12);
            o.block_align = le16(w + cur + 20);
            o.bitdepth = le16(w + cur + 22);
            if (fsz > 18 && raw_mode == 0xFFFE) {
                if (cur + 48 > len) return CRI_ERR_PCM(7);
                ext_bits = le16(w + cur + 26);
                subfmt = le32(w + cur + 32);
                if (subfmt != 1 && subfmt != 0xFFFE && subfmt != 3) return CRI_ERR_PCM(3);
            }
            if (raw_mode != 1 && raw_mode != 0xFFFE && raw_mode != 3) return CRI_ERR_PCM(3);
            have_fmt = true;
        } else if (sig == 0x6C706D73u) {
            uint32_t ssz = le32(w + cur + 4);
            if (ssz < 36) return CRI_ERR_PCM(4);
            if (cur + 44 > len) return CRI_ERR_PCM(7);
            uint32_t nl = le32(w + cur + 36), sd = le32(w + cur + 40);
            if ((uint64_t)ssz < (uint64_t)nl * 24 + sd + 36) return CRI_ERR_PCM(5);
            if (cur + 44 + (uint64_t)nl * 24 > len) return CRI_ERR_PCM(7);
            o.num_loops = nl;
            o.loop_start.clear(); o.loop_end.clear();
            for (uint32_t i = 0; i < nl; i++) {
                o.loop_start.push_back(le32(w + cur + 44 + 24 * (size_t)i + 8));
                o.loop_end.push_back(le32(w + cur + 44 + 24 * (size_t)i + 12));
            }
            o.looping = true;
        } else if (sig == 0x61746164u) {
            o.data_offset = cur + 8;
            o.data_size = le32(w + cur + 4);
            have_data = true;
        }
        cur += size;
        if ((uint64_t)sum + size > fullsize) return CRI_ERR_PCM(7);
        sum += size;
    }
    if (!have_fmt) return CRI_ERR_PCM(2);
    if (!have_data) return CRI_ERR_PCM(6);
    if (o.data_offset + o.data_size > len) return CRI_ERR_PCM(7);
    if (raw_mode == 0xFFFE) { o.bitdepth = ext_bits; o.mode = subfmt; } else o.mode = raw_mode;
    if (o.channels == 0 || o.block_align / o.channels == 0) return CRI_ERR_PCM(8);
    o.sample_size = o.block_align / o.channels;
    o.column_size = o.data_size / o.sample_size;
    if (o.mode == 3) { if (o.bitdepth != 32 && o.bitdepth != 64) return CRI_ERR_PCM(8); }
    else if (o.bitdepth < 1 || o.bitdepth > 32 || o.sample_size > 4 || o.sample_size < 1) return CRI_ERR_PCM(8);
    return 0;
}

bool wav_is_pcm16(const WavInfo& w) { return w.mode != 3 && w.bitdepth > 8 && w.bitdepth <= 16 && w.sample_size == 2; }

int wav_convertible(const WavInfo& w) {
    if (wav_is_pcm16(w)) return 0;
    if (w.bitdepth <= 8) return w.sample_size == 1 ? 0 : CRI_ERR_PCM(8);
    if (w.mode == 3) return (w.sample_size == 4 && w.bitdepth == 32) || (w.sample_size == 8 && w.bitdepth == 64) ? 0 : CRI_ERR_PCM(8);
    if (w.sample_size == 4 || w.sample_size == 3) return w.bitdepth >= 16 ? 0 : CRI_ERR_PCM(8);
    return CRI_ERR_PCM(8);
}

static int32_t trunc_x86(double v) { return (v >= -2147483648.0 && v < 2147483648.0) ? (int32_t)v : INT32_MIN; }

// pcm.cpp:455-461 (float -> PCM16, midpoint 32767), 498-504 (wide ints), 517-522 (8-bit)
int16_t wav_sample16(const WavInfo& w, const uint8_t* file, uint64_t index) {
    const uint8_t* p = file + w.data_offset + index * w.sample_size;
    if (wav_is_pcm16(w)) return (int16_t)le16(p);
    if (w.bitdepth <= 8) return (int16_t)(((int32_t)p[0] - (1 << (w.bitdepth - 1))) << 8);
    int32_t v;
    if (w.mode == 3) {
        if (w.bitdepth == 32) { float f; uint32_t u = le32(p); memcpy(&f, &u, 4); f = f * 32767.0f; v = trunc_x86((double)f); }
        else { double d; uint64_t u = (uint64_t)le32(p) | ((uint64_t)le32(p + 4) << 32); memcpy(&d, &u, 8); v = trunc_x86(d * 32767.0); }
        v = v > 32767 ? 32767 : (v < -32768 ? -32768 : v);
        return (int16_t)v;
    }
    if (w.sample_size == 4) return (int16_t)((((int32_t)le32(p)) >> (w.bitdepth - 16)) & 0xFFFF);
    v = (int32_t)(p[0] | (p[1] << 8) | (p[2] << 16));
    if (v & 0x800000) v |= (int32_t)0xFF000000;
    return (int16_t)((v >> (w.bitdepth - 16)) & 0xFFFF);
}

// pcm.cpp:350-375 (riff header), 262-269 (smpl), 547-556 (sizes)
uint32_t wav_write_header(uint8_t* d, uint32_t channels, uint32_t rate, uint32_t spc, bool looping, uint32_t ls, uint32_t le) {
    uint32_t hs = looping ? 0x70 : 0x2C, pos = 36, datasize = spc * channels * 2;
    put_le32(d, 0x46464952u); put_le32(d + 4, hs + datasize - 8); put_le32(d + 8, 0x45564157u);
    put_le32(d + 12, 0x20746D66u); put_le32(d + 16, 16); put_le16(d + 20, 1); put_le16(d + 22, channels);
    put_le32(d + 24, rate); put_le32(d + 28, 2 * channels * rate); put_le16(d + 32, 2 * channels); put_le16(d + 34, 16);
    if (looping) {
        put_le32(d + 36, 0x6C706D73u); put_le32(d + 40, 0x3C);
        memset(d + 44, 0, 0x3C);
        put_le32(d + 36 + 0x24, 1); put_le32(d + 36 + 0x34, ls); put_le32(d + 36 + 0x38, le);
        pos = 104;
    }
    put_le32(d + pos, 0x61746164u); put_le32(d + pos + 4, datasize);
    return hs;
}

// ------------------------------------------------------------------------------------------------ ADX
void adx_coefficients(uint32_t highpass, uint32_t rate, int32_t coef[2]) {   // adx.cpp:58-64 (+ macros 6-7)
    double a = 1.414213562373095 - cos(2.0 * 3.141592653589793 * (uint16_t)highpass / rate);
    double b = 1.414213562373095 - 1;
    double c = (a - sqrt((a + b) * (a - b))) / b;
    coef[0] = (int32_t)(c * 8192);
    coef[1] = (int32_t)(c * c * -4096);
}

// adx.cpp:298-358 (ADX::loadHeader) + 145-183 (field layout) + 117-129 (loop table) + 385-391
int adx_parse_header(const uint8_t* d, size_t len, AdxHeader& h) {
    if (!d || len < 20) return CRI_ERR_ADX(1);
    uint32_t sig = be16(d), flag = d[19];
    h.data_offset = be16(d + 2); h.mode = d[4]; h.blocksize = d[5]; h.bitdepth = d[6]; h.channels = d[7];
    h.rate = be32(d + 8); h.sample_count = be32(d + 12); h.highpass = be16(d + 16); h.version = d[18];
    h.looping = false; h.loop_start = h.loop_end = 0;
    if (sig != 0x8000) return CRI_ERR_ADX(1);
    if (h.mode == 0x11 || h.mode == 0x10 || h.version == 6 || h.blocksize == 0 || h.bitdepth == 0) return CRI_ERR_ADX(2);
    if (flag == 8 || flag == 9) return CRI_ERR_ADX(3);
    if (h.mode != 2 && h.mode != 3 && h.mode != 4) return CRI_ERR_ADX(4);
    if (h.version != 3 && h.version != 4 && h.version != 5) return CRI_ERR_ADX(5);
    if (((int)(h.blocksize - 2) * 8) % (int)h.bitdepth != 0 || h.bitdepth >= 16) return CRI_ERR_ADX(6);
    if (h.blocksize <= 2) return CRI_ERR_ADX(6);              // no samples per block: the reference divides by zero (2) or sizes its buffers negative (1)
    if (h.channels == 0) return CRI_ERR_ADX(7);
    h.history.assign((size_t)h.channels * 2, 0);
    uint32_t base = 20;
    bool looping = false;
    if (h.version == 4) {
        base += 4;
        for (uint32_t i = 0; i < h.channels; i++) {
            size_t p = base + 4 * (size_t)i;
            if (p + 4 <= len) { h.history[2 * i] = (int16_t)be16(d + p); h.history[2 * i + 1] = (int16_t)be16(d + p + 2); }
        }
        base += 4 * (h.channels > 1 ? h.channels : 2);
        if (base + 24 <= (uint32_t)((int32_t)h.data_offset - 2)) looping = true;
    } else if (h.version == 3) {
        if (base + 24 <= (uint32_t)((int32_t)h.data_offset - 2)) looping = true;
    }
    if (looping) {
        if ((size_t)base + 4 > len) return CRI_ERR_ADX(1);
        uint32_t lc = be16(d + base + 2);
        if (!lc) looping = false;
        else {
            if ((uint64_t)base + 4 + (uint64_t)lc * 20 >= (uint64_t)(int64_t)((int32_t)h.data_offset - 2)) return CRI_ERR_ADX(8);
            if ((size_t)base + 24 > len) return CRI_ERR_ADX(1);
            h.loop_start = be32(d + base + 8);
            h.loop_end = be32(d + base + 16);
        }
    }
    h.looping = looping;
    static const char cri_str[7] = "(c)CRI";
    for (uint32_t i = 0; i < 7; i++) {                        // 7 bytes incl. NUL: adx.cpp:345-348
        size_t p = (size_t)h.data_offset - 2 + i;
        if (h.data_offset < 2 || p >= len || (char)d[p] != cri_str[i]) return CRI_ERR_ADX(9);
    }
    h.samples_per_block = (h.blocksize - 2) * 8 / h.bitdepth;
    adx_coefficients(h.highpass, h.rate, h.coef);
    h.blocks = (uint32_t)ceilf((float)h.sample_count / (float)h.samples_per_block);
    return 0;
}

// adx.cpp:416-489 (validation, padding rule, header size, initial history) + 359-379 (header bytes) +
// 131-142 / 94-105 (loop table).  `image` holds every byte the reference writes before the block loop starts:
// the 16-aligned header, plus -- for more than 6 channels -- the per-channel history entries that spill past
// it into the first blocks (the block writer ORs into the first byte of a block, IO.cpp:139).
int adx_plan_encode(const uint8_t* wav, size_t len, const WavInfo& w, uint32_t bd, uint32_t bs, uint32_t mode, uint32_t highpass,
                    uint32_t filter, uint32_t ver, bool force_no_loop, AdxEncodePlan& p) {
    (void)len;
    uint32_t ch = w.channels & 0xFF;
    bool looping = (force_no_loop && ver == 5) ? false : w.looping;
    if (ch < 1) return CRI_ERR_ADX(10);
    if (bd <= 1 || bd >= 16) return CRI_ERR_ADX(11);
    if (bs <= 2 || bs > 255) return CRI_ERR_ADX(12);
    if (mode != 2 && mode != 3 && mode != 4) return CRI_ERR_ADX(13);
    if (filter > 3) return CRI_ERR_ADX(15);
    if (ver != 3 && ver != 4 && ver != 5) return CRI_ERR_ADX(16);
    if ((8 * (bs - 2)) % bd != 0) return CRI_ERR_ADX(17);
    if (w.column_size < ch || w.column_size % ch != 0) return CRI_ERR_ADX(18);
    if (looping && w.num_loops == 0) return CRI_ERR_UNSUPPORTED;
    uint32_t dbs = bs - 2, spb = dbs * 8 / bd, spc = w.column_size / ch, frames;
    if (spc % spb != 0) frames = ((uint32_t)next_multiple((int)spc, (int)dbs)) / spb;   // adx.cpp:450-452
    else frames = spc / spb;
    p.channels = ch; p.samples_per_channel = spc; p.samples_per_block = spb; p.frames = frames;
    if (mode == 2) { p.coef[0] = ADX_STATIC_COEFS[filter * 2]; p.coef[1] = ADX_STATIC_COEFS[filter * 2 + 1]; }
    else adx_coefficients((uint16_t)highpass, w.rate, p.coef);
    p.history.assign((size_t)ch * 2, 0);
    if (ver == 4 || ver == 5) {
        for (uint32_t i = 0; i < ch; i++) {                    // first sample of each channel (adx.cpp:473-476); 16-bit input only
            int16_t s = 0;
            if (wav_convertible(w) == 0 && (uint64_t)(i + 1) * w.sample_size <= w.data_size) s = wav_sample16(w, wav, i);
            p.history[2 * i] = p.history[2 * i + 1] = s;
        }
    }
    uint32_t hs = 20 + 6;
    if (ver == 4 || ver == 5) hs += 8;                          // adx.cpp:482 with a zeroed Header.Channels
    if (looping) hs += 4 + w.num_loops * 20;
    hs = hs % 16 == 0 ? hs : hs + (16 - hs % 16);
    p.header_size = hs;
    p.total_size = (uint64_t)hs + (uint64_t)frames * ch * bs + bs;
    size_t img = hs + 1;
    if (ver == 4 || ver == 5) { size_t e = 24 + 4 * (size_t)ch; if (e > img) img = e; }
    if (img > p.total_size) img = (size_t)p.total_size;
    p.image.assign(img, 0);
    uint8_t* o = p.image.data();
    auto P8 = [&](size_t pos, uint32_t v) { if (pos < img) o[pos] = (uint8_t)v; };
    auto P16 = [&](size_t pos, uint32_t v) { P8(pos, v >> 8); P8(pos + 1, v); };
    auto P32 = [&](size_t pos, uint32_t v) { P16(pos, v >> 16); P16(pos + 2, v & 0xFFFF); };
    P16(0, 0x8000); P16(2, hs - 4); P8(4, mode); P8(5, bs); P8(6, bd); P8(7, w.channels);
    P32(8, w.rate); P32(12, spc); P16(16, mode == 2 ? 0 : (uint16_t)highpass); P8(18, ver); P8(19, 0);
    size_t off = 20;
    if (ver == 4 || ver == 5) {
        P32(off, 0);
        for (uint32_t i = 0; i < ch; i++) { P16(off + 4 + 4 * (size_t)i, (uint16_t)p.history[2 * i]); P16(off + 6 + 4 * (size_t)i, (uint16_t)p.history[2 * i + 1]); }
        off += 4 + (ch > 1 ? 4 * ch : 8);
    }
    if (looping) {
        uint32_t sif = (bs - 2) * 2;
        uint16_t align = (uint16_t)next_multiple((int)w.loop_start[0], (int)(ch == 1 ? sif * 2 : sif));
        P16(off, align); P16(off + 2, w.num_loops);
        for (uint32_t i = 0; i < w.num_loops; i++) {
            uint32_t st = w.loop_start[i] + align, en = w.loop_end[i] + align;
            uint32_t sb = hs + ((st / spb) * bs) * ch;
            uint32_t eb = hs + (uint32_t)next_multiple((int)((en / spb) * bs + (en % spb) / bs), (int)bs) * ch;
            size_t q = off + 4 + 20 * (size_t)i;
            P16(q, i); P16(q + 2, 1); P32(q + 4, st); P32(q + 8, sb); P32(q + 12, en); P32(q + 16, eb);
        }
    }
    static const char cri_str[7] = "(c)CRI";
    for (uint32_t i = 0; i < 7; i++) P8((size_t)hs + i - 6, (uint8_t)cri_str[i]);
    return 0;
}

// ------------------------------------------------------------------------------------------------ HCA
void hca_channel_types(uint32_t channels, uint32_t track_count, uint32_t stereo_bands, uint32_t config, uint8_t t[16]) {
    // hca.cpp:887-958 (decoder) and 2323-2401 (encoder) build the same table
    memset(t, CRI_CH_DISCRETE, 16);
    if (track_count == 0) return;
    uint32_t cpt = channels / track_count;
    if (stereo_bands == 0 || cpt <= 1) return;
    for (uint32_t i = 0; i + cpt <= channels && i / cpt < track_count; i += cpt) {
        uint8_t* c = t + i;
        if (cpt >= 2 && cpt <= 8) { c[0] = CRI_CH_PRIMARY; c[1] = CRI_CH_SECONDARY; }
        if (cpt == 4 && config == 0) { c[2] = CRI_CH_PRIMARY; c[3] = CRI_CH_SECONDARY; }
        if (cpt == 5 && config <= 2) { c[3] = CRI_CH_PRIMARY; c[4] = CRI_CH_SECONDARY; }
        if (cpt >= 6 && cpt <= 8) { c[4] = CRI_CH_PRIMARY; c[5] = CRI_CH_SECONDARY; }
        if (cpt == 8) { c[6] = CRI_CH_PRIMARY; c[7] = CRI_CH_SECONDARY; }
    }
}

// hca.cpp:628-984 (clHCA_DecodeHeader).  Returns 0 or CRI_ERR_HCA_HEADER.
// The reference reads the chunks through its bit reader, initialised with the header size the caller passed (hca.cpp:639):
// a read that would cross that bound returns 0 (hca.cpp:231-232), so a chunk magic is only ever matched inside it and
// fields cut off by it read as zero.  `rd` is that rule (and never reads past `len` either).  The `ath` chunk does not
// reduce `size` (hca.cpp:750-753) -- kept, it changes which later chunks a short header still accepts.
int hca_parse_header(const uint8_t* d, size_t len, uint32_t size_arg, HcaHeader& h) {
    memset(&h, 0, sizeof h);
    uint32_t size = size_arg, pos = 0;
    const size_t bound = len < (size_t)size_arg ? len : (size_t)size_arg;
    auto rd = [&](uint32_t at, uint32_t n) -> uint32_t {
        if ((size_t)at + n > bound) return 0;
        uint32_t v = 0;
        for (uint32_t k = 0; k < n; k++) v = (v << 8) | d[at + k];
        return v;
    };
    auto magic = [&](uint32_t at) { return rd(at, 4) & 0x7F7F7F7Fu; };
    if (!d || size < 8 || len < 8) return CRI_ERR_HCA_HEADER;
    if (magic(0) != 0x48434100u) return CRI_ERR_HCA_HEADER;
    h.version = rd(4, 2); h.header_size = rd(6, 2);
    if (h.version != 0x0101 && h.version != 0x0102 && h.version != 0x0103 && h.version != 0x0200 && h.version != 0x0300) return CRI_ERR_HCA_HEADER;
    if (size < h.header_size || len < h.header_size) return CRI_ERR_HCA_HEADER;
    if (crc16(d, h.header_size)) return CRI_ERR_HCA_HEADER;
    size -= 8; pos = 8;
    if (size >= 0x10 && magic(pos) == 0x666D7400u) {                                   // fmt, hca.cpp:667-688
        h.channels = rd(pos + 4, 1); h.rate = rd(pos + 5, 3); h.frame_count = rd(pos + 8, 4);
        h.delay = rd(pos + 12, 2); h.padding = rd(pos + 14, 2);
        if (!(h.channels >= 1 && h.channels <= 16) || h.frame_count == 0 || !(h.rate >= 1 && h.rate <= 0x7FFFFF)) return CRI_ERR_HCA_HEADER;
        size -= 0x10; pos += 0x10;
    } else return CRI_ERR_HCA_HEADER;
    if (size >= 0x10 && magic(pos) == 0x636F6D70u) {                                   // comp, hca.cpp:691-709
        h.frame_size = rd(pos + 4, 2); h.min_res = rd(pos + 6, 1); h.max_res = rd(pos + 7, 1); h.track_count = rd(pos + 8, 1);
        h.channel_config = rd(pos + 9, 1); h.total_bands = rd(pos + 10, 1); h.base_bands = rd(pos + 11, 1); h.stereo_bands = rd(pos + 12, 1);
        h.bands_per_hfr_group = rd(pos + 13, 1); h.ms_stereo = rd(pos + 14, 1);
        size -= 0x10; pos += 0x10;
    } else if (size >= 0x0c && magic(pos) == 0x64656300u) {                            // dec (v1.x), hca.cpp:710-727
        h.frame_size = rd(pos + 4, 2); h.min_res = rd(pos + 6, 1); h.max_res = rd(pos + 7, 1);
        h.total_bands = rd(pos + 8, 1) + 1u; h.base_bands = rd(pos + 9, 1) + 1u;
        h.track_count = rd(pos + 10, 1) >> 4; h.channel_config = rd(pos + 10, 1) & 0xF; h.stereo_type = rd(pos + 11, 1);
        if (h.stereo_type == 0) h.base_bands = h.total_bands;
        h.stereo_bands = h.total_bands - h.base_bands;
        h.bands_per_hfr_group = 0;
        size -= 0x0c; pos += 0x0c;
    } else return CRI_ERR_HCA_HEADER;
    if (size >= 8 && magic(pos) == 0x76627200u) {                                      // vbr, hca.cpp:733-748
        uint32_t mx = rd(pos + 4, 2);
        if (!(h.frame_size == 0 && mx > 8 && mx <= 0x1FF)) return CRI_ERR_HCA_HEADER;
        size -= 8; pos += 8;
    }
    if (size >= 6 && magic(pos) == 0x61746800u) { h.ath_type = rd(pos + 4, 2); pos += 6; } // ath, hca.cpp:750-753: `size` is NOT reduced
    else h.ath_type = h.version < 0x0200 ? 1 : 0;
    if (size >= 0x10 && magic(pos) == 0x6C6F6F70u) {                                   // loop, hca.cpp:760-774
        h.loop_start_frame = rd(pos + 4, 4); h.loop_end_frame = rd(pos + 8, 4);
        h.loop_start_delay = rd(pos + 12, 2); h.loop_end_padding = rd(pos + 14, 2);
        h.loop_flag = 1;
        if (!(h.loop_start_frame <= h.loop_end_frame && h.loop_end_frame < h.frame_count)) return CRI_ERR_HCA_HEADER;
        size -= 0x10; pos += 0x10;
    }
    if (size >= 6 && magic(pos) == 0x63697068u) {                                      // ciph, hca.cpp:786-794
        h.ciph_type = rd(pos + 4, 2);
        if (!(h.ciph_type == 0 || h.ciph_type == 1 || h.ciph_type == 56)) return CRI_ERR_HCA_HEADER;
        size -= 6; pos += 6;
    }
    if (size >= 8 && magic(pos) == 0x72766100u) { size -= 8; pos += 8; }               // rva, hca.cpp:799-812 (volume unused by the decoder)
    if (size >= 5 && magic(pos) == 0x636F6D6Du) {                                      // comm, hca.cpp:814-830
        h.comment_len = rd(pos + 4, 1);
        if (h.comment_len > size) return CRI_ERR_HCA_HEADER;
        size -= 5 + h.comment_len; pos += 5 + h.comment_len;
    }
    if (!(h.frame_size >= 8 && h.frame_size <= 0xFFFF)) return CRI_ERR_HCA_HEADER;
    if (h.version <= 0x0200) { if (h.min_res != 1 || h.max_res != 15) return CRI_ERR_HCA_HEADER; }
    else if (h.min_res > h.max_res || h.max_res > 15) return CRI_ERR_HCA_HEADER;
    if (h.track_count == 0) h.track_count = 1;
    if (h.track_count > h.channels) return CRI_ERR_HCA_HEADER;
    if (h.total_bands > 128 || h.base_bands > 128 || h.stereo_bands > 128 || h.base_bands + h.stereo_bands > 128 ||
        h.bands_per_hfr_group > 128) return CRI_ERR_HCA_HEADER;
    {
        uint32_t a = h.total_bands - h.base_bands - h.stereo_bands, b = h.bands_per_hfr_group;   // hca.cpp:619-623, 872-874
        h.hfr_group_count = b < 1 ? 0 : (a / b + ((a % b) ? 1 : 0));
    }
    if (h.hfr_group_count > 128) return CRI_ERR_HCA_HEADER;   // total < base + stereo wraps the count: the reference then indexes scalefactors[128 - count] out of bounds
    if (h.ath_type == 0) memset(h.ath, 0, 128);                                               // hca.cpp:451-485
    else if (h.ath_type == 1) {
        uint32_t acc = 0;
        for (uint32_t i = 0; i < 128; i++) {
            acc += h.rate;
            uint32_t index = acc >> 13;
            if (index >= 654) { memset(h.ath + i, 0xFF, 128 - i); break; }
            h.ath[i] = HCA_ATH_BASE[index];
        }
    } else return CRI_ERR_HCA_HEADER;
    hca_channel_types(h.channels, h.track_count, h.stereo_bands, h.channel_config, h.type);
    for (uint32_t i = 0; i < h.channels; i++)
        h.coded[i] = h.type[i] != CRI_CH_SECONDARY ? h.base_bands + h.stereo_bands : h.base_bands;
    if (h.ms_stereo) return CRI_ERR_HCA_HEADER;
    return 0;
}

static void cipher56_row(uint8_t* r, uint8_t key) {          // hca.cpp:524-534
    int mul = ((key & 1) << 3) | 5, add = (key & 0xE) | 1;
    key >>= 4;
    for (int i = 0; i < 16; i++) { key = (uint8_t)((key * mul + add) & 0xF); r[i] = key; }
}

int hca_cipher_table(uint32_t type, uint64_t key, uint8_t t[256]) {   // hca.cpp:499-617
    if (type == 56 && !key) type = 0;
    if (type == 0) { for (uint32_t i = 0; i < 256; i++) t[i] = (uint8_t)i; return 0; }
    if (type == 1) {
        uint32_t v = 0;
        for (uint32_t i = 1; i < 255; i++) {
            v = (v * 13 + 11) & 0xFF;
            if (v == 0 || v == 0xFF) v = (v * 13 + 11) & 0xFF;
            t[i] = (uint8_t)v;
        }
        t[0] = 0; t[255] = 0xFF;
        return 0;
    }
    if (type != 56) return CRI_ERR_HCA_HEADER;
    uint8_t kc[8], seed[16], base[256], br[16], bc[16];
    key--;
    for (uint32_t r = 0; r < 7; r++) { kc[r] = (uint8_t)key; key >>= 8; }
    const uint8_t pick[16][2] = {{1, 0}, {1, 6}, {2, 3}, {2, 0}, {2, 1}, {3, 4}, {3, 0}, {3, 2},
                                 {4, 5}, {4, 0}, {4, 3}, {5, 6}, {5, 0}, {5, 4}, {6, 1}, {6, 0}};
    for (uint32_t i = 0; i < 16; i++) seed[i] = (uint8_t)(kc[pick[i][0]] ^ (pick[i][1] ? kc[pick[i][1]] : 0));
    cipher56_row(br, kc[0]);
    for (uint32_t r = 0; r < 16; r++) {
        cipher56_row(bc, seed[r]);
        for (uint32_t c = 0; c < 16; c++) base[r * 16 + c] = (uint8_t)((br[r] << 4) | bc[c]);
    }
    uint32_t x = 0, pos = 1;
    for (uint32_t i = 0; i < 256; i++) {
        x = (x + 17) & 0xFF;
        if (base[x] != 0 && base[x] != 0xFF) t[pos++] = base[x];
    }
    t[0] = 0; t[255] = 0xFF;
    return 0;
}

uint64_t hca_mix_key(uint64_t key, uint16_t subkey) {        // hca.cpp:3381-3383 / 3309-3311
    if (subkey) key = key * (((uint64_t)subkey << 16) | (uint64_t)((uint16_t)~subkey + 2u));
    return key;
}

// hca.cpp:3166-3250 (CryptHeader): toggle bit 7 of the chunk magics, rewrite the ciph type, refresh the header CRC.
// Same walk as hca_parse_header, including the `ath` chunk that leaves `size` alone (hca.cpp:3203-3206): every magic is
// matched through the reference's bounded reader (bound = the header size, hca.cpp:3171), so nothing past the header is
// read; the reference's one unguarded write -- the cipher type of a `ciph` chunk whose last bytes lie past the header --
// is dropped instead of written out of bounds.
void hca_crypt_header(uint8_t* d, uint32_t hs, uint32_t encrypt, uint32_t type) {
    uint32_t size = hs, pos = 0;
    auto rd = [&](uint32_t at, uint32_t n) -> uint32_t {
        if ((uint64_t)at + n > hs) return 0;
        uint32_t v = 0;
        for (uint32_t k = 0; k < n; k++) v = (v << 8) | d[at + k];
        return v;
    };
    auto magic = [&](uint32_t at) { return rd(at, 4) & 0x7F7F7F7Fu; };
    auto flip = [&](uint32_t at, int n) { for (int i = 0; i < n; i++) d[at + i] ^= 0x80; };   // only after magic(at) matched: at + 4 <= hs
    if (hs < 2) return;
    if (magic(pos) == 0x48434100u) { flip(pos, 3); pos += 8; size -= 8; }
    if (size >= 0x10 && magic(pos) == 0x666D7400u) { flip(pos, 3); pos += 16; size -= 16; }
    if (size >= 0x10 && magic(pos) == 0x636F6D70u) { flip(pos, 4); pos += 16; size -= 16; }
    else if (size >= 0x0c && magic(pos) == 0x64656300u) { flip(pos, 3); pos += 12; size -= 12; }
    if (size >= 8 && magic(pos) == 0x76627200u) { flip(pos, 3); pos += 8; size -= 8; }
    if (size >= 6 && magic(pos) == 0x61746800u) { flip(pos, 3); pos += 6; }
    if (size >= 0x10 && magic(pos) == 0x6C6F6F70u) { flip(pos, 4); pos += 16; size -= 16; }
    if (size >= 6 && magic(pos) == 0x63697068u) {
        flip(pos, 4);
        if ((uint64_t)pos + 6 <= hs) put_be16(d + pos + 4, encrypt == 1 ? (type & 0xFFFF) : 0);
        pos += 6; size -= 6;
    }
    if (size >= 8 && magic(pos) == 0x72766100u) { flip(pos, 3); pos += 8; size -= 8; }
    if (size >= 5 && magic(pos) == 0x636F6D6Du) { uint32_t cl = rd(pos + 4, 1); flip(pos, 4); pos += 5 + cl; size -= 5 + cl; }
    if (size >= 4 && magic(pos) == 0x70616400u) flip(pos, 3);
    put_be16(d + hs - 2, crc16(d, hs - 2));
}

static int div_round_up(int v, int d) { return (int)ceilf((float)v / d); }   // hca.cpp:182-184
static uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }

// hca.cpp:2206-2234 (bitrate), 2236-2270 (band counts), 2272-2277 (hfr), 2279-2290 (channel config),
// 2307-2321 (header size), 2414-2462 (initHCAEncode), non-looping path.
int hca_enc_setup(uint32_t channels, uint32_t rate, uint32_t spc, uint32_t quality, HcaEncSetup& e) {
    memset(&e, 0, sizeof e);
    if (channels == 0 || rate == 0) return CRI_ERR_HCA_CHANNEL_CONFIG;
    uint32_t pcm_bitrate = rate * channels * 16, max_bitrate = pcm_bitrate / 4, cutoff = rate / 2;
    int ratio = 6;
    switch (quality) {
        case 0: ratio = 4; break; case 1: ratio = 6; break; case 2: ratio = 8; break;
        case 3: ratio = channels == 1 ? 10 : 12; break; case 4: ratio = channels == 1 ? 12 : 16; break;
        default: break;
    }
    uint32_t bitrate = pcm_bitrate / (uint32_t)ratio;
    if (bitrate > max_bitrate) bitrate = max_bitrate;
    if (bitrate == 0) return CRI_ERR_HCA_CHANNEL_CONFIG;
    e.channels = channels; e.rate = rate; e.samples_per_channel = spc; e.delay = 128;
    e.frame_size = bitrate * 1024 / rate / 8;
    uint32_t hfr_ratio, cutoff_ratio;
    if (channels <= 1 || pcm_bitrate / bitrate <= 6) { hfr_ratio = 6; cutoff_ratio = 12; } else { hfr_ratio = 8; cutoff_ratio = 16; }
    if (bitrate < pcm_bitrate / cutoff_ratio) cutoff = umin(cutoff, cutoff_ratio * bitrate / (32 * channels));
    uint32_t total = (uint32_t)round((double)cutoff * 256.0 / rate);
    uint32_t hfr_start = umin(total, (uint32_t)round(((double)hfr_ratio * bitrate * 128.0) / pcm_bitrate));
    uint32_t stereo_start = hfr_ratio == 6 ? hfr_start : (hfr_start + 1) / 2;
    uint32_t hfr_bands = total - hfr_start, groups = 0;
    uint32_t bpg = (uint32_t)div_round_up((int)hfr_bands, 8);
    if (bpg > 0) groups = (uint32_t)div_round_up((int)hfr_bands, (int)bpg);
    e.total_bands = total; e.base_bands = stereo_start; e.stereo_bands = hfr_start - stereo_start;
    e.hfr_group_count = groups; e.bands_per_hfr_group = bpg;
    if (bpg > 0) {
        e.hfr_band_count = e.total_bands - e.base_bands - e.stereo_bands;
        e.hfr_group_count = (uint32_t)div_round_up((int)e.hfr_band_count, (int)bpg);
    }
    if (channels > 8) return CRI_ERR_HCA_CHANNEL_CONFIG;
    e.channel_config = HCA_DEFAULT_CHANNEL_CONFIG[channels];
    if (HCA_VALID_CHANNEL_CONFIG[channels - 1][e.channel_config] != 1) return CRI_ERR_HCA_CHANNEL_CONFIG;
    e.header_size = 96;
    e.frame_count = (uint32_t)div_round_up((int)(spc + e.delay), 1024);
    e.padding = e.frame_count * 1024 - e.delay - spc;
    hca_channel_types(channels, 1, e.stereo_bands, e.channel_config, e.type);
    for (uint32_t i = 0; i < channels; i++)
        e.coded[i] = e.type[i] == CRI_CH_SECONDARY ? e.base_bands : e.base_bands + e.stereo_bands;
    return 0;
}

// Loop branch of initHCAEncode (hca.cpp:2439-2462) + CalculateLoopInfo (2292-2306) + CalculateHeaderSize (2308-2321);
// runs after hca_enc_setup.  column_size is the WAV's total interleaved sample count (the reference clamps with it as is).
void hca_enc_setup_loop(HcaEncSetup& e, uint32_t loop_start, uint32_t loop_end, uint32_t column_size) {
    e.loop_flag = 1;
    e.samples_per_channel = loop_end < column_size ? loop_end : column_size;
    e.delay += (uint32_t)next_multiple((int)loop_start, 1024) - loop_start;
    const uint32_t ls = loop_start + e.delay, le = loop_end + e.delay;
    e.loop_start_frame = ls / 1024; e.loop_start_delay = ls % 1024;
    e.loop_end_frame = le / 1024; e.loop_end_padding = 1024 - le % 1024;
    if (e.loop_end_padding == 1024) { e.loop_end_frame--; e.loop_end_padding = 0; }
    uint32_t input = (uint32_t)next_multiple((int)e.samples_per_channel, 128);
    input = (input < column_size ? input : column_size) + 256;
    e.post_samples = input - e.samples_per_channel;
    const uint32_t off = e.header_size + e.frame_size * e.loop_start_frame;
    const uint32_t pad_bytes = (uint32_t)next_multiple((int)off, 2048) - off, pad_frames = pad_bytes / e.frame_size;
    e.delay += pad_frames * 1024;
    e.loop_start_frame += pad_frames; e.loop_end_frame += pad_frames;
    e.header_size += pad_bytes % e.frame_size;
    e.frame_count = (uint32_t)div_round_up((int)(input + e.delay), 1024);
    e.padding = e.frame_count * 1024 - e.delay - input;
    e.pre_samples = e.delay - 128;
    e.loop_start = loop_start;
}

void hca_pack_header(const HcaEncSetup& e, uint8_t* o) {      // hca.cpp:3109-3164
    memset(o, 0, e.header_size);
    put_be32(o, 0x48434100u); put_be16(o + 4, 0x0200); put_be16(o + 6, e.header_size);
    put_be32(o + 8, 0x666D7400u); put_be32(o + 12, e.rate); o[12] = (uint8_t)e.channels;
    put_be32(o + 16, e.frame_count); put_be16(o + 20, e.delay); put_be16(o + 22, e.padding);
    put_be32(o + 24, 0x636F6D70u); put_be16(o + 28, e.frame_size); o[30] = 1; o[31] = 15; o[32] = 1;
    o[33] = (uint8_t)e.channel_config; o[34] = (uint8_t)e.total_bands; o[35] = (uint8_t)e.base_bands;
    o[36] = (uint8_t)e.stereo_bands; o[37] = (uint8_t)e.bands_per_hfr_group;
    uint32_t pos = 40;
    if (e.loop_flag) {
        put_be32(o + 40, 0x6C6F6F70u); put_be32(o + 44, e.loop_start_frame); put_be32(o + 48, e.loop_end_frame);
        put_be16(o + 52, e.loop_start_delay); put_be16(o + 54, e.loop_end_padding);
        pos = 56;
    }
    put_be32(o + pos, 0x63697068u); put_be16(o + pos + 4, 0);
    put_be32(o + pos + 6, 0x70616400u);
    put_be16(o + e.header_size - 2, crc16(o, e.header_size - 2));
}

// ------------------------------------------------------------------------------------------------ encoder tables
// The rate loop (CalculateUsedBits, hca.cpp:2763-2790) costs a spectrum x of a band with resolution r as
//   r >= 8:  (r - 3) - 1 bits, one more when |x| >= QuantizerDeadZone[r]
//   r <  8:  QuantizeSpectrumBits[r][(int)(x * inv + (inv + 1)) - (int)(inv + 0.5 - 8)]
// and every row of QuantizeSpectrumBits is "shortest code near zero, one bit more from some |q| on" (plus entries of 0 past
// the row's ends).  (int)(fl(fl(x * inv) + up)) is monotone in x, so the bits of x are
//   shortest[r] + (x >= t_plus[r]) + (x <= -t_minus[r])
// for two floats found here by bisection over the float bit patterns with the reference's own expression (they differ by a few
// ulps: the sum rounds differently on the two sides).  The fifteen thresholds of either sign are well apart and in the same order,
// so "x passes resolution r's threshold" is "at least rank[r] of its sign's thresholds are <= |x|": the kernel computes that count
// -- the CLASS of x, 0 .. 15 -- once per spectrum (a table row per binade of |x| and sign holds the at most two thresholds inside
// it), keeps a band's eight classes as eight bytes, and a search step then costs a band
//   8 * shortest[r] + (number of its bytes with class + (16 - rank[r]) >= 16)            (two adds, two ands, two popcounts).
// The one exception is x = 0.9999999f (ScaleSpectra's clamp): at resolutions 2, 4 and 5 the sum rounds up to the next integer,
// the index runs one past the row's codes and the entry there is 0 -- no bits at all.  The kernel counts those values per
// band once and takes their bits back out ("anomaly").  Everything is derived from the tables and checked: a table of
// another shape makes this function fail instead of producing wrong bytes.
static float f32_from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static int enc_len_of(int r, float x) {
    const float inv = HCA_ENC_INV_STEP[r], up = inv + 1;
    const int down = (int)((double)inv + 0.5 - 8);
    volatile float m = x * inv;                            // two roundings, as the reference's build (no FMA)
    volatile float t = m + up;
    const int idx = (int)t - down;
    return (idx >= 0 && idx < 16) ? HCA_ENC_CODE_LEN[r][idx] : 0;
}
int hca_enc_build_tables(std::vector<uint8_t>& blob) {
    blob.assign(HCA_ET_BYTES, 0);
    float* win4 = (float*)(blob.data() + HCA_ET_WIN4);
    float* tw = (float*)(blob.data() + HCA_ET_TW);
    float* deq = (float*)(blob.data() + HCA_ET_DEQ);
    float* escale = (float*)(blob.data() + HCA_ET_ESCALE);
    uint32_t* cp = (uint32_t*)(blob.data() + HCA_ET_CP);
    uint32_t* cls = (uint32_t*)(blob.data() + HCA_ET_CLS);
    float* inv = (float*)(blob.data() + HCA_ET_INV);
    float* ib = (float*)(blob.data() + HCA_ET_IBOUNDS);
    uint8_t* sfbase = blob.data() + HCA_ET_SFBASE;
    uint8_t* clen = blob.data() + HCA_ET_CLEN;
    uint8_t* code = blob.data() + HCA_ET_CODE;
    uint8_t* ishuf = blob.data() + HCA_ET_ISHUF;
    for (int i = 0; i < 128; i++) { ishuf[HCA_ENC_SHUFFLE[i]] = (uint8_t)i; clen[i] = HCA_ENC_CODE_LEN[i >> 4][i & 15]; code[i] = HCA_ENC_CODE[i >> 4][i & 15]; }
    for (int i = 128; i < 256; i++) { clen[i] = (uint8_t)((i >> 4) - 4); code[i] = 0; }      // sign-magnitude codes: the length of a zero (a non-zero value takes one bit more)
    // the MDCT's window + fold (hca.cpp:2532-2547) as k_hca_encode's lanes use it: point j = l8 + 8 r of a subframe takes the folded
    // inputs k = 2 j (even) and 127 - 2 j (odd); input k < 64 is  -w[63 - k] x[192 + k] + w[64 + k] x[191 - k],  k >= 64 is
    // w[k - 64] x[k - 64] + w[191 - k] x[191 - k]  (x = the subframe's 256 samples, previous half first).  The window is held times
    // 2^-15: PcmToFloat's scale (hca.cpp:2470-2479), exact -- a power of two commutes with the rounding of the product
    for (int r = 0; r < 8; r++)
        for (int l8 = 0; l8 < 8; l8++)
            for (int odd = 0; odd < 2; odd++) {
                const int k = odd ? 127 - 2 * l8 - 16 * r : 2 * l8 + 16 * r;
                const bool low = k < 64;
                const float wa = HCA_WINDOW[low ? 63 - k : k - 64] * (1.0f / 32768.0f), wb = HCA_WINDOW[low ? 64 + k : 191 - k] * (1.0f / 32768.0f);
                win4[4 * (8 * r + l8) + odd] = low ? -wa : wa;
                win4[4 * (8 * r + l8) + 2 + odd] = wb;
            }
    {
        int k = 0;
        const int rows[7] = {7, 5, 4, 3, 2, 1, 0}, count[7] = {64, 32, 16, 8, 4, 2, 1};
        for (int j = 0; j < 7; j++) for (int i = 0; i < count[j]; i++, k++) { tw[2 * k] = HCA_ENC_COS[rows[j]][i]; tw[2 * k + 1] = HCA_ENC_SIN[rows[j]][i]; }
    }
    for (int i = 0; i < 72; i++) deq[i] = i < 63 ? HCA_DEQ_SCALE[i] : f32_from_bits(0x7FC00000u);   // NaN padding never compares <=
    for (int i = 0; i < 64; i++) escale[i] = i ? HCA_ENC_SCALE[i] : 0.0f;     // (entry 0 is never used as a factor: scalefactor 0 means zeros, hca.cpp:2641-2644)
    for (int j = 0; j < 32; j++) {                         // entries 0..62 that are <= 2^(j - 25) (hca.cpp:2611-2623 by exponent)
        const float thr = f32_from_bits((uint32_t)(j + 102) << 23);
        int n = 0;
        for (int k = 0; k < 63; k++) n += HCA_DEQ_SCALE[k] <= thr ? 1 : 0;
        sfbase[j] = (uint8_t)n;
    }
    for (int i = 0; i < 16; i++) { inv[i] = HCA_ENC_INV_STEP[i]; ib[i] = i < 14 ? HCA_ENC_INTENSITY_BOUNDS[i] : 0.0f; }
    // per resolution: shortest code, thresholds, anomaly
    uint32_t tplus[16], tminus[16], shortest[16], anomaly[16];
    tplus[0] = tminus[0] = 0x7F800000u; shortest[0] = 0; anomaly[0] = 0;   // resolution 0: nothing is written
    for (int r = 1; r < 16; r++) {
        anomaly[r] = 0;
        if (r >= 8) { uint32_t u; memcpy(&u, &HCA_ENC_DEAD_ZONE[r], 4); tplus[r] = tminus[r] = u; shortest[r] = (uint32_t)r - 4; continue; }
        const int l0 = enc_len_of(r, 0.0f);
        shortest[r] = (uint32_t)l0;
        for (int sign = 0; sign < 2; sign++) {
            // smallest magnitude whose code is not the shortest one
            uint32_t lo = 0, hi = HCA_ENC_CLAMP_BITS;
            const float s = sign ? -1.0f : 1.0f;
            if (enc_len_of(r, s * f32_from_bits(hi)) == l0) return CRI_ERR_INVALID_ARG;
            while (lo < hi) { const uint32_t mid = lo + (hi - lo) / 2; if (enc_len_of(r, s * f32_from_bits(mid)) != l0) hi = mid; else lo = mid + 1; }
            (sign ? tminus : tplus)[r] = lo;
            // from there on: one bit more, up to the clamp -- except (positive side only) the clamp value itself, which may cost nothing
            const uint32_t probes[6] = {lo, lo + 1, lo + (HCA_ENC_CLAMP_BITS - lo) / 3, lo + (HCA_ENC_CLAMP_BITS - lo) / 2, HCA_ENC_CLAMP_BITS - 1, HCA_ENC_CLAMP_BITS};
            for (int k = 0; k < 6; k++) {
                const int l = enc_len_of(r, s * f32_from_bits(probes[k]));
                if (l == l0 + 1) continue;
                if (!sign && probes[k] == HCA_ENC_CLAMP_BITS && l == 0) { anomaly[r] = 1; continue; }
                return CRI_ERR_INVALID_ARG;
            }
        }
        // exact check of the step structure: every index step of the quantiser (at most 16 per side) changes the length as the rule says
        for (int sign = 0; sign < 2; sign++) {
            const float s = sign ? -1.0f : 1.0f;
            uint32_t at = 0;
            while (at < HCA_ENC_CLAMP_BITS) {                // next magnitude where the length changes
                const int l = enc_len_of(r, s * f32_from_bits(at));
                uint32_t lo = at, hi = HCA_ENC_CLAMP_BITS;
                if (enc_len_of(r, s * f32_from_bits(hi)) == l) {
                    // no length change up to the clamp -- but the index may step without changing the length; walk the steps
                    break;
                }
                while (lo < hi) { const uint32_t mid = lo + (hi - lo) / 2; if (enc_len_of(r, s * f32_from_bits(mid)) != l) hi = mid; else lo = mid + 1; }
                const uint32_t expect = (sign ? tminus : tplus)[r];
                if (lo != expect && !(lo == HCA_ENC_CLAMP_BITS && !sign && anomaly[r])) return CRI_ERR_INVALID_ARG;
                at = lo;
            }
        }
    }
    // ranks: the thresholds of either sign in ascending order are the same sequence of resolutions, strictly ascending
    uint32_t rank[16]; rank[0] = 0;
    for (int r = 1; r < 16; r++) {
        uint32_t below_p = 0, below_m = 0;
        for (int q = 1; q < 16; q++) {
            if (q == r) continue;
            if (tplus[q] == tplus[r] || tminus[q] == tminus[r]) return CRI_ERR_INVALID_ARG;
            below_p += tplus[q] < tplus[r]; below_m += tminus[q] < tminus[r];
        }
        if (below_p != below_m) return CRI_ERR_INVALID_ARG;
        rank[r] = below_p + 1;
    }
    // classes by half-binade: a row per (sign, exponent field, top mantissa bit) = the float's top ten bits -- everything below 2^-13
    // is under every threshold, no half-binade above holds more than one (checked: a table of another shape makes the job fail)
    for (int sign = 0; sign < 2; sign++)
        for (uint32_t h = 0; h < 256; h++) {
            const uint32_t* t = sign ? tminus : tplus;
            uint32_t in_row = 0x7F800000u, n = 0, base = 0;
            for (int r = 1; r < 16; r++) {
                const uint32_t th = t[r] >> 22;
                if (th <= 2 * 114 + 1 || th > 2 * 126 + 1) return CRI_ERR_INVALID_ARG;    // the rows up to there stand for everything smaller: they must be empty
                if (th < h) base++;
                else if (th == h) { in_row = t[r]; n++; }
            }
            if (n > 1) return CRI_ERR_INVALID_ARG;
            uint32_t* row = cls + 2 * (512 * sign + h);
            row[0] = in_row; row[1] = base;
        }
    for (int i = 0; i < 60; i++) {
        const int r = i < 59 ? HCA_ENC_CURVE_TO_RES[i] : 0;
        cp[2 * i + 0] = r ? (16 - rank[r]) * 0x01010101u : 0u;
        cp[2 * i + 1] = 8 * shortest[r] | (uint32_t)r << 20 | anomaly[r] << 28;
    }
    return 0;
}

}  // namespace cri
