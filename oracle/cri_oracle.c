/* oracle/cri_oracle.c -- TEST INFRASTRUCTURE ONLY (see cri_oracle.h for the rules and the parity status).
 *
 * Plain-C restatement of the reference's ADX and HCA algorithms.  Each function cites the reference
 * lines (relative to /root/reference/CriCodecs/) whose behaviour it restates.  Semantics are "reference
 * with zero-initialised buffers and in-bounds accesses".  Build: oracle/Makefile (gcc -O2 -fwrapv
 * -ffp-contract=off; the reference's float work is single IEEE binary32 operations, no FMA).
 */
#include "cri_oracle.h"
#include "cri_tables.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define E_ADX(n) (-(n))
#define E_PCM(n) (-(100 + (n)))
#define E_HCA_HEADER (-201)
#define E_HCA_DECODE (-202)
#define E_HCA_CHCONF (-203)
#define E_HCA_ENCODE (-204)
#define E_ARG (-301)
#define E_NOMEM (-302)
#define E_UNSUPPORTED (-304)

void ora_free(void* p) { free(p); }

/* ------------------------------------------------------------------------------------------------
 * Byte / bit IO (IO.hpp:1-32, IO.cpp:11-182)
 * ---------------------------------------------------------------------------------------------- */
static uint32_t be16(const uint8_t* p) { return ((uint32_t)p[0] << 8) | p[1]; }
static uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
static uint32_t le16(const uint8_t* p) { return p[0] | ((uint32_t)p[1] << 8); }
static uint32_t le32(const uint8_t* p) { return p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static void put_be16(uint8_t* p, uint32_t v) { p[0] = (uint8_t)(v >> 8); p[1] = (uint8_t)v; }
static void put_be32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; }
static void put_le16(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
static void put_le32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }

static int next_multiple(int value, int multiple) { /* IO.hpp:26-32 */
    if (multiple <= 0) return value;
    if (value % multiple == 0) return value;
    return value + multiple - value % multiple;
}

/* MSB-first bit extraction; out-of-range bits read as 0 (IO.cpp:77-117 / hca.cpp:225-281 reduce to this
 * inside the buffer; hca.cpp:232-233 returns 0 when the read would cross the end). */
typedef struct { const uint8_t* p; uint32_t nbits; uint32_t pos; } bitrd;
static uint32_t br_peek(const bitrd* b, uint32_t n) {   /* plain extraction: ADX blocks (IO.cpp:77-117) */
    uint32_t v = 0, i;
    if (n == 0 || b->pos + n > b->nbits) return 0;
    for (i = 0; i < n; i++) { uint32_t q = b->pos + i; v = (v << 1) | ((b->p[q >> 3] >> (7 - (q & 7))) & 1u); }
    return v;
}
/* HCA frame reader, hca.cpp:225-281.  The reference picks a 32/24/16/8-bit window from how many bits are LEFT
 * in the frame, not from how many the read spans, so within the last 3 bytes of a frame a read can be served
 * from a window that is too narrow (shift count goes negative; x86 masks it to 5 bits).  Valid streams never
 * read there (the last 16 bits are the CRC); wrong-key garbage does, and decodes "successfully" in the
 * reference, so the quirk is restated to keep garbage-in/garbage-out identical. */
static uint32_t hca_peek(const bitrd* b, uint32_t n) {
    uint32_t bit = b->pos, rem = bit & 7, size = b->nbits, v, off, left;
    const uint8_t* d;
    if (!((uint64_t)bit + n <= size)) return 0;
    d = b->p + (bit >> 3);
    off = n + rem; left = size - bit;
    if (left >= 32 && off >= 25) { v = be32(d) & (0xFFFFFFFFu >> rem); v >>= (32 - rem - n) & 31; }
    else if (left >= 24 && off >= 17) { v = ((uint32_t)d[0] << 16 | (uint32_t)d[1] << 8 | d[2]) & (0xFFFFFFu >> rem); v >>= (24 - rem - n) & 31; }
    else if (left >= 16 && off >= 9) { v = ((uint32_t)d[0] << 8 | d[1]) & (0xFFFFu >> rem); v >>= (16 - rem - n) & 31; }
    else { v = d[0] & (0xFFu >> rem); v >>= (8 - rem - n) & 31; }
    return v;
}
static uint32_t hca_read(bitrd* b, uint32_t n) { uint32_t v = hca_peek(b, n); b->pos += n; return v; }
static uint32_t br_read(bitrd* b, uint32_t n) { uint32_t v = br_peek(b, n); b->pos += n; return v; }

/* MSB-first bit packer writing into a zeroed buffer.  The reference writer ORs into the first touched byte
 * and overwrites the following ones (IO.cpp:129-156); on sequential writes into a zeroed block that equals
 * plain OR-packing.  Bits that do not fit are dropped without advancing (IO.cpp:131-134). */
typedef struct { uint8_t* p; uint32_t nbits; uint32_t pos; } bitwr;
static void bw_write(bitwr* b, int32_t value, uint32_t n) {
    uint32_t i;
    if (n > 32 || n > b->nbits - b->pos) return;
    for (i = 0; i < n; i++) {
        uint32_t q = b->pos + i;
        if (((uint32_t)value >> (n - 1 - i)) & 1u) b->p[q >> 3] |= (uint8_t)(0x80u >> (q & 7));
    }
    b->pos += n;
}

/* hca.cpp:205-211 (CRC-16, poly 0x8005, MSB first, init 0) */
uint16_t ora_crc16(const uint8_t* p, size_t n) {
    uint16_t sum = 0; size_t i;
    for (i = 0; i < n; i++) sum = (uint16_t)((sum << 8) ^ CRI_CRC16_TAB[(sum >> 8) ^ p[i]]);
    return sum;
}

static int32_t clamp_i(int32_t v, int32_t limit) { /* pcm.cpp:155-161: [~limit, limit] */
    if (v > limit) return limit;
    if (v < ~limit) return ~limit;
    return v;
}

/* ------------------------------------------------------------------------------------------------
 * WAV in (pcm.cpp:163-342, 411-444, 455-545) and WAV out (pcm.cpp:350-375, 547-556, 262-269)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t channels, rate, block_align, bitdepth, mode;   /* mode: 1 PCM, 3 IEEE float (after EXTENSIBLE unwrap) */
    const uint8_t* data; uint32_t data_size;
    int looping; uint32_t num_loops; const uint8_t* loops;  /* 24-byte smplloop records */
    uint32_t column_size;                                   /* total interleaved samples */
    int16_t* pcm; int owned;
} wavin;

static int wav_parse(const uint8_t* w, size_t len, wavin* o) {
    uint32_t fullsize, sum = 4, have_fmt = 0, have_data = 0, ext_bits = 0, subfmt = 0, raw_mode = 0;
    size_t cur = 12;
    memset(o, 0, sizeof *o);
    if (len < 12) return E_PCM(1);
    if (le32(w) != 0x46464952u || le32(w + 8) != 0x45564157u) return E_PCM(1);       /* pcm.cpp:335-341 */
    fullsize = le32(w + 4);
    while (sum < fullsize) {                                                          /* pcm.cpp:291-327 */
        uint32_t sig, size;
        if (cur + 8 > len) return E_PCM(7);
        sig = le32(w + cur);
        size = le32(w + cur + 4) + 8;
        if (size < 8) return E_PCM(7);                       /* wrapped 32-bit add: the reference stops advancing here */
        size += ((size & 1) && (uint64_t)size + sum + (size & 1) <= fullsize);
        if (sig == 0x20746D66u) {                                                     /* "fmt " pcm.cpp:177-201 */
            uint32_t fsz = le32(w + cur + 4);
            if (fsz < 16) return E_PCM(2);
            if (cur + 24 > len) return E_PCM(7);
            raw_mode = le16(w + cur + 8);
            o->channels = le16(w + cur + 10);
            o->rate = le32(w + cur + 12);
            o->block_align = le16(w + cur + 20);
            o->bitdepth = le16(w + cur + 22);
            if (fsz > 18 && raw_mode == 0xFFFE) {
                if (cur + 48 > len) return E_PCM(7);
                ext_bits = le16(w + cur + 26);
                subfmt = le32(w + cur + 32);
                if (subfmt != 1 && subfmt != 0xFFFE && subfmt != 3) return E_PCM(3);
            }
            if (raw_mode != 1 && raw_mode != 0xFFFE && raw_mode != 3) return E_PCM(3);
            have_fmt = 1;
        } else if (sig == 0x6C706D73u) {                                              /* "smpl" pcm.cpp:243-261 */
            uint32_t ssz = le32(w + cur + 4), nl, sd;
            if (ssz < 36) return E_PCM(4);
            if (cur + 44 > len) return E_PCM(7);
            nl = le32(w + cur + 36);
            sd = le32(w + cur + 40);
            if ((uint64_t)ssz < (uint64_t)nl * 24 + sd + 36) return E_PCM(5);
            if (cur + 44 + (uint64_t)nl * 24 > len) return E_PCM(7);
            o->num_loops = nl;
            o->loops = w + cur + 44;
            o->looping = 1;
        } else if (sig == 0x61746164u) {                                              /* "data" pcm.cpp:276-283 */
            o->data = w + cur + 8;
            o->data_size = le32(w + cur + 4);
            have_data = 1;
        }
        cur += size;
        if ((uint64_t)sum + size > fullsize) return E_PCM(7);
        sum += size;
    }
    if (!have_fmt) return E_PCM(2);
    if (!have_data) return E_PCM(6);
    if ((size_t)(o->data - w) + o->data_size > len) return E_PCM(7);
    /* pcm.cpp:419-444 */
    if (raw_mode == 0xFFFE) { o->bitdepth = ext_bits; o->mode = subfmt; } else o->mode = raw_mode;
    if (o->channels == 0 || o->block_align / o->channels == 0) return E_PCM(8);
    o->column_size = o->data_size / (o->block_align / o->channels);
    {
        uint32_t ss = o->block_align / o->channels, bd = o->bitdepth;
        if (o->mode == 3) { if (bd != 32 && bd != 64) return E_PCM(8); }
        else if (bd < 1 || bd > 32 || ss > 4 || ss < 1) return E_PCM(8);
    }
    return 0;
}

/* pcm.cpp:530-545 and the converters 455-523 */
static int wav_pcm16(wavin* o) {
    uint32_t ss = o->block_align / o->channels, bd = o->bitdepth, n = o->column_size, i;
    if (bd > 8 && bd <= 16 && ss == 2 && o->mode != 3) {
        o->pcm = (int16_t*)malloc((size_t)n * 2 + 2);
        if (!o->pcm) return E_NOMEM;
        for (i = 0; i < n; i++) o->pcm[i] = (int16_t)le16(o->data + 2 * (size_t)i);
        o->owned = 1;
        return 0;
    }
    o->pcm = (int16_t*)malloc((size_t)n * 2 + 2);
    if (!o->pcm) return E_NOMEM;
    o->owned = 1;
    if (bd <= 8) {                                   /* pcm.cpp:517-522 */
        int32_t mid = 1 << (bd - 1);
        if (ss != 1) return E_PCM(8);
        for (i = 0; i < n; i++) o->pcm[i] = (int16_t)(((int32_t)o->data[i] - mid) << 8);
    } else if (o->mode == 3) {                       /* pcm.cpp:455-461, target depth 16 => midpoint 32767 */
        for (i = 0; i < n; i++) {
            int32_t v;
            if (bd == 32) { float f; uint32_t u = le32(o->data + 4 * (size_t)i); memcpy(&f, &u, 4); f = f * 32767.0f;
                            v = (f >= 2147483648.0f || f < -2147483648.0f || f != f) ? INT32_MIN : (int32_t)f; }
            else { double d; uint64_t u = (uint64_t)le32(o->data + 8 * (size_t)i) | ((uint64_t)le32(o->data + 8 * (size_t)i + 4) << 32);
                   memcpy(&d, &u, 8); d = d * 32767.0;
                   v = (d >= 2147483648.0 || d < -2147483648.0 || d != d) ? INT32_MIN : (int32_t)d; }
            o->pcm[i] = (int16_t)clamp_i(v, 32767);
        }
    } else if (ss == 4) {                            /* pcm.cpp:498-504 */
        for (i = 0; i < n; i++) o->pcm[i] = (int16_t)((((int32_t)le32(o->data + 4 * (size_t)i)) >> (bd - 16)) & 0xFFFF);
    } else if (ss == 3) {
        for (i = 0; i < n; i++) {
            const uint8_t* p = o->data + 3 * (size_t)i;
            int32_t v = (int32_t)(p[0] | (p[1] << 8) | (p[2] << 16));
            if (v & 0x800000) v |= (int32_t)0xFF000000;
            o->pcm[i] = (int16_t)((v >> (bd - 16)) & 0xFFFF);
        }
    } else return E_PCM(8);
    return 0;
}

static void wav_release(wavin* o) { if (o->owned) free(o->pcm); o->pcm = 0; o->owned = 0; }

/* pcm.cpp:350-375 + 262-269 + 547-556.  Returns header size (44 or 0x70). */
static uint32_t wav_write_header(uint8_t* d, uint32_t channels, uint32_t rate, uint32_t samples_per_channel,
                                 int looping, uint32_t loop_start, uint32_t loop_end) {
    uint32_t hs = looping ? 0x70 : 0x2C, pos = 36;
    uint32_t datasize = samples_per_channel * channels * 2;
    put_le32(d + 0, 0x46464952u);
    put_le32(d + 4, hs + datasize - 8);
    put_le32(d + 8, 0x45564157u);
    put_le32(d + 12, 0x20746D66u);
    put_le32(d + 16, 16);
    put_le16(d + 20, 1);
    put_le16(d + 22, channels);
    put_le32(d + 24, rate);
    put_le32(d + 28, 2 * channels * rate);
    put_le16(d + 32, 2 * channels);
    put_le16(d + 34, 16);
    if (looping) {
        put_le32(d + 36, 0x6C706D73u);
        put_le32(d + 40, 0x3C);
        memset(d + 44, 0, 0x3C);
        put_le32(d + 36 + 0x24, 1);
        put_le32(d + 36 + 0x34, loop_start);
        put_le32(d + 36 + 0x38, loop_end);
        pos = 104;
    }
    put_le32(d + pos, 0x61746164u);
    put_le32(d + pos + 4, datasize);
    return hs;
}

/* ------------------------------------------------------------------------------------------------
 * ADX
 * ---------------------------------------------------------------------------------------------- */
void ora_adx_coefficients(uint32_t highpass, uint32_t rate, int32_t coef[2]) { /* adx.cpp:58-64 */
    double a = 1.414213562373095 - cos(2.0 * 3.141592653589793 * (uint16_t)highpass / rate);
    double b = 1.414213562373095 - 1;
    double c = (a - sqrt((a + b) * (a - b))) / b;
    coef[0] = (int32_t)(c * 8192);
    coef[1] = (int32_t)(c * c * -4096);
}

static int32_t adx_static_coef(uint32_t predictor, int k) { /* adx.cpp:45; predictors 4..7 index past the table */
    uint32_t idx = predictor * 2 + (uint32_t)k;
    return idx < 8 ? ADX_STATIC_COEFS[idx] : 0;
}

/* adx.cpp:298-358 (header), 380-415 (driver), 189-214 (block) */
int ora_adx_decode(const uint8_t* d, size_t len, uint8_t** out, size_t* out_len) {
    uint32_t sig, data_offset, mode, bs, bd, ch, rate, count, hp, ver, flag;
    uint32_t base = 20, looping = 0, loop_start = 0, loop_end = 0, i, j, s;
    uint32_t dbs, spb, blocks, hs;
    int32_t coef[2];
    int16_t (*hist)[2];
    uint8_t* o; size_t total, off;
    if (!d || !out || !out_len || len < 20) return E_ADX(1);
    sig = be16(d); data_offset = be16(d + 2); mode = d[4]; bs = d[5]; bd = d[6]; ch = d[7];
    rate = be32(d + 8); count = be32(d + 12); hp = be16(d + 16); ver = d[18]; flag = d[19];
    if (sig != 0x8000) return E_ADX(1);
    if (mode == 0x11 || mode == 0x10 || ver == 6 || bs == 0 || bd == 0) return E_ADX(2);
    if (flag == 8 || flag == 9) return E_ADX(3);
    if (mode != 2 && mode != 3 && mode != 4) return E_ADX(4);
    if (ver != 3 && ver != 4 && ver != 5) return E_ADX(5);
    if (((int)(bs - 2) * 8) % (int)bd != 0 || bd >= 16) return E_ADX(6);
    if (bs <= 2) return E_ADX(6);        /* no samples per block: the reference divides by zero (2) or sizes its buffers negative (1) */
    if (ch == 0) return E_ADX(7);
    hist = (int16_t(*)[2])calloc(ch, sizeof *hist);
    if (!hist) return E_NOMEM;
    if (ver == 4) {
        base += 4;
        for (i = 0; i < ch; i++) {
            size_t p = base + 4 * (size_t)i;
            if (p + 4 <= len) { hist[i][0] = (int16_t)be16(d + p); hist[i][1] = (int16_t)be16(d + p + 2); }
        }
        base += 4 * (ch > 1 ? ch : 2);
        if (base + 24 <= (uint32_t)((int32_t)data_offset - 2)) looping = 1;
    } else if (ver == 3) {
        if (base + 24 <= (uint32_t)((int32_t)data_offset - 2)) looping = 1;
    }
    if (looping) {                                                         /* adx.cpp:117-129 */
        uint32_t lc;
        if ((size_t)base + 4 > len) { free(hist); return E_ADX(1); }
        lc = be16(d + base + 2);
        if (!lc) looping = 0;
        else {
            if ((uint64_t)base + 4 + (uint64_t)lc * 20 >= (uint64_t)(int64_t)((int32_t)data_offset - 2)) { free(hist); return E_ADX(8); }
            if ((size_t)base + 4 + 20 > len) { free(hist); return E_ADX(1); }
            loop_start = be32(d + base + 4 + 4);
            loop_end = be32(d + base + 4 + 12);
        }
    }
    dbs = bs - 2; spb = dbs * 8 / bd;
    for (i = 0; i < 7; i++) {                                              /* adx.cpp:345-348 */
        size_t p = (size_t)data_offset - 2 + i;
        static const char cri[7] = "(c)CRI";
        if (data_offset < 2 || p >= len || (char)d[p] != cri[i]) { free(hist); return E_ADX(9); }
    }
    ora_adx_coefficients(hp, rate, coef);
    blocks = (uint32_t)ceilf((float)count / (float)spb);
    if ((uint64_t)count * ch * 2 > 0x7FFFFF00ull) { free(hist); return E_ARG; }
    total = (looping ? 0x70 : 0x2C) + (size_t)count * ch * 2;
    o = (uint8_t*)calloc(1, total ? total : 1);
    if (!o) { free(hist); return E_NOMEM; }
    hs = wav_write_header(o, ch, rate, count, (int)looping, loop_start, loop_end);
    off = (size_t)data_offset + 4;
    for (i = 0; i < blocks; i++) {
        if (off + 2 > len) break;
        if (d[off] == 0x80 && d[off + 1] == 0x01) break;                   /* EOF scale, adx.cpp:405-406 */
        if (off + (size_t)bs * ch > len) break;
        for (j = 0; j < ch; j++, off += bs) {
            int32_t scale = (int32_t)be16(d + off), c0 = coef[0], c1 = coef[1];
            bitrd r; r.p = d + off + 2; r.nbits = dbs * 8; r.pos = 0;
            if (mode == 4) scale = (int32_t)(1u << ((12 - scale) & 31));
            else if (mode == 2) {
                uint32_t pred = ((uint32_t)scale >> 13) & 7;
                scale = (scale & 0x1FFF) + 1;
                c0 = coef[0] = adx_static_coef(pred, 0);
                c1 = coef[1] = adx_static_coef(pred, 1);
            } else scale += 1;
            for (s = 0; s < spb; s++) {
                uint32_t raw = br_read(&r, bd);
                int32_t v = (int32_t)(raw << (32 - bd)) >> (32 - bd);
                uint64_t idx = (uint64_t)i * spb + s;
                v = v * scale + ((c0 * (int32_t)hist[j][0]) >> 12) + ((c1 * (int32_t)hist[j][1]) >> 12);
                v = clamp_i(v, 0x7FFF);
                if (idx < count) put_le16(o + hs + (idx * ch + j) * 2, (uint32_t)v & 0xFFFF);
                hist[j][1] = hist[j][0];
                hist[j][0] = (int16_t)v;
            }
        }
    }
    free(hist);
    *out = o; *out_len = total;
    return 0;
}

/* adx.cpp:215-273 */
static void adx_encode_block(uint8_t* blk, uint32_t bs, uint32_t spb, const int16_t* pcm, uint32_t stride,
                             const int32_t coef[2], uint32_t bd, uint32_t mode, uint32_t filter_bits, int16_t hist[2]) {
    int32_t mn = 0, mx = 0, limit = (1 << (bd - 1)) - 1;
    int16_t o1 = hist[0], o2 = hist[1];
    uint32_t i;
    uint16_t scale;
    bitwr w; w.p = blk; w.nbits = bs * 8; w.pos = 0;
    for (i = 0; i < spb; i++) {
        int32_t x = pcm[(size_t)i * stride];
        int32_t r = ((x << 12) - coef[0] * hist[0] - coef[1] * hist[1]) >> 12;
        if (r < mn) mn = r; else if (r > mx) mx = r;
        hist[1] = hist[0]; hist[0] = (int16_t)x;
    }
    if (!mn && !mx) { memset(blk, 0, bs); return; }      /* silent block: history stays raw (adx.cpp:231-234) */
    scale = (uint16_t)(mx / limit > mn / ~limit ? mx / limit : mn / ~limit);
    if (scale > 0x1000) scale = 0x1000;
    if (mode == 4) {
        uint32_t power = 0;
        if (scale) { uint32_t v = scale; power = 0; while (v >>= 1) power++; power += 1; }
        scale = (uint16_t)(1u << power);
        bw_write(&w, (int32_t)(12 - (int32_t)power), 16);
    } else if (mode == 2) bw_write(&w, (int32_t)(filter_bits | (scale & 0x1FFF)), 16);
    else bw_write(&w, scale, 16);
    hist[0] = o1; hist[1] = o2;
    for (i = 0; i < spb; i++) {
        int32_t x = pcm[(size_t)i * stride], delta, sim;
        delta = ((x << 12) - coef[0] * hist[0] - coef[1] * hist[1]) >> 12;
        if (!scale) scale = 1;
        delta = delta > 0 ? delta + (scale >> 1) : delta - (scale >> 1);
        delta /= scale;
        delta = clamp_i(delta, limit);
        sim = ((delta << 12) * (int32_t)scale + coef[0] * hist[0] + coef[1] * hist[1]) >> 12;
        sim = clamp_i(sim, 0x7FFF);
        hist[1] = hist[0]; hist[0] = (int16_t)sim;
        bw_write(&w, delta, bd);
    }
}

/* adx.cpp:416-506 (driver), 359-379 (header), 94-105 / 131-142 (loop table) */
int ora_adx_encode(const uint8_t* wav, size_t len, uint32_t bd, uint32_t bs, uint32_t mode, uint32_t highpass,
                   uint32_t filter, uint32_t ver, int force_no_loop, uint8_t** out, size_t* out_len) {
    wavin w; int rc; uint32_t ch, spb, dbs, spc, frames, hs, i, j, looping;
    int32_t coef[2]; int16_t (*hist)[2] = 0; int16_t* pcm = 0; int own_pcm = 0;
    uint8_t* o; size_t total, off; uint16_t hp = (uint16_t)highpass;
    if (!wav || !out || !out_len) return E_ARG;
    rc = wav_parse(wav, len, &w);
    if (rc) return rc;
    ch = w.channels & 0xFF;                                 /* unsigned char ChannelCount, adx.cpp:418 */
    looping = (force_no_loop && ver == 5) ? 0 : (uint32_t)w.looping;
    if (ch < 1) return E_ADX(10);
    if (bd <= 1 || bd >= 16) return E_ADX(11);
    if (bs <= 2 || bs > 255) return E_ADX(12);
    if (mode != 2 && mode != 3 && mode != 4) return E_ADX(13);
    if (filter > 3) return E_ADX(15);
    if (ver != 3 && ver != 4 && ver != 5) return E_ADX(16);
    if ((8 * (bs - 2)) % bd != 0) return E_ADX(17);
    if (w.column_size < ch || w.column_size % ch != 0) return E_ADX(18);
    if (looping && w.num_loops == 0) return E_UNSUPPORTED;  /* reference reads an empty loop array here */
    rc = wav_pcm16(&w);
    if (rc) { wav_release(&w); return rc; }
    dbs = bs - 2; spb = dbs * 8 / bd; spc = w.column_size / ch;
    if (spc % spb != 0) {                                   /* adx.cpp:450-456 */
        uint32_t needed = (uint32_t)next_multiple((int)spc, (int)dbs) * ch;
        frames = (needed / ch) / spb;
        pcm = (int16_t*)calloc(needed ? needed : 1, 2);
        if (!pcm) { wav_release(&w); return E_NOMEM; }
        memcpy(pcm, w.pcm, (size_t)w.column_size * 2);
        own_pcm = 1;
    } else { pcm = w.pcm; frames = spc / spb; }
    if (mode == 2) { coef[0] = ADX_STATIC_COEFS[filter * 2]; coef[1] = ADX_STATIC_COEFS[filter * 2 + 1]; }
    else ora_adx_coefficients(hp, w.rate, coef);
    hist = (int16_t(*)[2])calloc(ch, sizeof *hist);
    for (i = 0; i < ch; i++) if (ver == 4 || ver == 5) hist[i][0] = hist[i][1] = pcm[i];
    hs = 20 + 6;
    if (ver == 4 || ver == 5) hs += 8;                      /* adx.cpp:482 reads a zeroed Header.Channels */
    if (looping) hs += 4 + w.num_loops * 20;
    hs = hs % 16 == 0 ? hs : hs + (16 - hs % 16);
    total = (size_t)hs + (size_t)frames * ch * bs + bs;
    o = (uint8_t*)calloc(1, total);
    if (!o) { free(hist); if (own_pcm) free(pcm); wav_release(&w); return E_NOMEM; }
#define PUT8(pos, v) do { size_t p_ = (pos); if (p_ < total) o[p_] = (uint8_t)(v); } while (0)
#define PUT16(pos, v) do { PUT8((pos), (uint32_t)(v) >> 8); PUT8((pos) + 1, (v)); } while (0)
#define PUT32(pos, v) do { PUT16((pos), (uint32_t)(v) >> 16); PUT16((pos) + 2, (uint32_t)(v) & 0xFFFF); } while (0)
    PUT16(0, 0x8000); PUT16(2, hs - 4); PUT8(4, mode); PUT8(5, bs); PUT8(6, bd); PUT8(7, w.channels);
    PUT32(8, w.rate); PUT32(12, spc); PUT16(16, mode == 2 ? 0 : hp); PUT8(18, ver); PUT8(19, 0);
    off = 20;
    if (ver == 4 || ver == 5) {
        PUT32(off, 0);
        for (i = 0; i < ch; i++) { PUT16(off + 4 + 4 * (size_t)i, (uint16_t)hist[i][0]); PUT16(off + 6 + 4 * (size_t)i, (uint16_t)hist[i][1]); }
        off += 4 + (ch > 1 ? 4 * ch : 8);
    }
    if (looping) {                                          /* adx.cpp:131-142, 94-105 */
        uint32_t start0 = le32(w.loops + 8), sif = (bs - 2) * 2;
        uint16_t align = (uint16_t)next_multiple((int)start0, (int)(ch == 1 ? sif * 2 : sif));
        PUT16(off, align); PUT16(off + 2, w.num_loops);
        for (i = 0; i < w.num_loops; i++) {
            uint32_t ls = le32(w.loops + 24 * (size_t)i + 8), le = le32(w.loops + 24 * (size_t)i + 12);
            uint32_t st = ls + align, en = le + align, sb, eb;
            size_t p = off + 4 + 20 * (size_t)i;
            sb = hs + ((st / spb) * bs) * ch;
            eb = hs + (uint32_t)next_multiple((int)((en / spb) * bs + (en % spb) / bs), (int)bs) * ch;
            PUT16(p, i); PUT16(p + 2, 1); PUT32(p + 4, ls + align); PUT32(p + 8, sb); PUT32(p + 12, le + align); PUT32(p + 16, eb);
        }
    }
    { static const char cri[7] = "(c)CRI"; for (i = 0; i < 7; i++) PUT8((size_t)hs + i - 6, cri[i]); }
    off = hs;
    {
        uint8_t* blk = (uint8_t*)malloc(bs);
        for (i = 0; i < frames; i++)
            for (j = 0; j < ch; j++, off += bs) {
                uint8_t stale = o[off];                    /* writer ORs into the first byte (IO.cpp:139) */
                memset(blk, 0, bs);
                blk[0] = stale;
                adx_encode_block(blk, bs, spb, pcm + (size_t)i * spb * ch + j, ch, coef, bd, mode, filter << 13, hist[j]);
                memcpy(o + off, blk, bs);
            }
        free(blk);
    }
    memset(o + off, 0, bs);
    PUT16(off, 0x8001);
    PUT16(off + 2, (bs - 4) & 0xFFFF);
    free(hist); if (own_pcm) free(pcm); wav_release(&w);
    *out = o; *out_len = total;
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * HCA common: cipher tables (hca.cpp:499-617), ATH (451-485)
 * ---------------------------------------------------------------------------------------------- */
static void cipher56_row(uint8_t* r, uint8_t key) {          /* hca.cpp:524-534 */
    int mul = ((key & 1) << 3) | 5, add = (key & 0xE) | 1, i;
    key >>= 4;
    for (i = 0; i < 16; i++) { key = (uint8_t)((key * mul + add) & 0xF); r[i] = key; }
}

int ora_cipher_table(uint32_t type, uint64_t key, uint8_t t[256]) {
    uint32_t i;
    if (type == 56 && !key) type = 0;
    if (type == 0) { for (i = 0; i < 256; i++) t[i] = (uint8_t)i; return 0; }
    if (type == 1) {                                         /* hca.cpp:508-522 */
        uint32_t v = 0;
        for (i = 1; i < 255; i++) {
            v = (v * 13 + 11) & 0xFF;
            if (v == 0 || v == 0xFF) v = (v * 13 + 11) & 0xFF;
            t[i] = (uint8_t)v;
        }
        t[0] = 0; t[255] = 0xFF;
        return 0;
    }
    if (type == 56) {                                        /* hca.cpp:536-597 */
        uint8_t kc[8], seed[16], base[256], br[16], bc[16];
        uint32_t r, c, x = 0, pos = 1;
        key--;
        for (r = 0; r < 7; r++) { kc[r] = (uint8_t)key; key >>= 8; }
        seed[0] = kc[1]; seed[1] = kc[1] ^ kc[6]; seed[2] = kc[2] ^ kc[3]; seed[3] = kc[2];
        seed[4] = kc[2] ^ kc[1]; seed[5] = kc[3] ^ kc[4]; seed[6] = kc[3]; seed[7] = kc[3] ^ kc[2];
        seed[8] = kc[4] ^ kc[5]; seed[9] = kc[4]; seed[10] = kc[4] ^ kc[3]; seed[11] = kc[5] ^ kc[6];
        seed[12] = kc[5]; seed[13] = kc[5] ^ kc[4]; seed[14] = kc[6] ^ kc[1]; seed[15] = kc[6];
        cipher56_row(br, kc[0]);
        for (r = 0; r < 16; r++) {
            cipher56_row(bc, seed[r]);
            for (c = 0; c < 16; c++) base[r * 16 + c] = (uint8_t)((br[r] << 4) | bc[c]);
        }
        for (i = 0; i < 256; i++) {
            x = (x + 17) & 0xFF;
            if (base[x] != 0 && base[x] != 0xFF) t[pos++] = base[x];
        }
        t[0] = 0; t[255] = 0xFF;
        return 0;
    }
    return E_HCA_HEADER;
}

static uint64_t mix_key(uint64_t key, uint16_t subkey) {     /* hca.cpp:3381-3383 */
    if (subkey) key = key * (((uint64_t)subkey << 16) | (uint64_t)((uint16_t)~subkey + 2u));
    return key;
}

/* ------------------------------------------------------------------------------------------------
 * HCA header (hca.cpp:628-984)
 * ---------------------------------------------------------------------------------------------- */
enum { CH_DISCRETE = 0, CH_PRIMARY = 1, CH_SECONDARY = 2 };

typedef struct {
    uint32_t version, header_size, channels, rate, frame_count, delay, padding;
    uint32_t frame_size, min_res, max_res, track_count, channel_config, stereo_type;
    uint32_t total_bands, base_bands, stereo_bands, bands_per_hfr_group, ms_stereo;
    uint32_t ath_type, loop_start_frame, loop_end_frame, loop_start_delay, loop_end_padding, loop_flag;
    uint32_t ciph_type, comment_len, hfr_group_count;
    uint32_t ciph_pos;                      /* byte offset of the ciph chunk (0 if none) */
    uint8_t ath[128];
    uint8_t type[16]; uint32_t coded[16];
} hca_info;

static void channel_types(uint32_t channels, uint32_t track_count, uint32_t stereo_bands, uint32_t config, uint8_t* t) {
    /* hca.cpp:887-958 (decoder) == 2323-2401 (encoder) */
    uint32_t cpt = channels / track_count, i;
    memset(t, CH_DISCRETE, 16);
    if (stereo_bands == 0 || cpt <= 1) return;
    for (i = 0; i + cpt <= channels && i / cpt < track_count; i += cpt) {
        uint8_t* c = t + i;
        if (cpt >= 2 && cpt <= 8) { c[0] = CH_PRIMARY; c[1] = CH_SECONDARY; }
        if (cpt == 4 && config == 0) { c[2] = CH_PRIMARY; c[3] = CH_SECONDARY; }
        if (cpt == 5 && config <= 2) { c[3] = CH_PRIMARY; c[4] = CH_SECONDARY; }
        if (cpt >= 6 && cpt <= 8) { c[4] = CH_PRIMARY; c[5] = CH_SECONDARY; }
        if (cpt == 8) { c[6] = CH_PRIMARY; c[7] = CH_SECONDARY; }
    }
}

/* the reference's bit reader at byte granularity (hca.cpp:225-232): a read crossing the bound returns 0 */
static uint32_t rd_be(const uint8_t* d, size_t bound, uint32_t at, uint32_t n) {
    uint32_t v = 0, k;
    if ((size_t)at + n > bound) return 0;
    for (k = 0; k < n; k++) v = (v << 8) | d[at + k];
    return v;
}

static int hca_parse_header(const uint8_t* d, size_t len, uint32_t size_arg, hca_info* h) {
    uint32_t size = size_arg, pos = 0, i;
    const size_t bound = len < (size_t)size_arg ? len : (size_t)size_arg;   /* bitreader_init(&br, data, size), hca.cpp:639 */
#define RD(at, n) rd_be(d, bound, (at), (n))
#define MAGIC(at) (RD((at), 4) & 0x7F7F7F7Fu)
    memset(h, 0, sizeof *h);
    if (size < 8 || len < 8) return E_HCA_HEADER;
    if (MAGIC(0) != 0x48434100u) return E_HCA_HEADER;
    h->version = RD(4, 2); h->header_size = RD(6, 2);
    if (h->version != 0x0101 && h->version != 0x0102 && h->version != 0x0103 && h->version != 0x0200 && h->version != 0x0300) return E_HCA_HEADER;
    if (size < h->header_size || len < h->header_size) return E_HCA_HEADER;
    if (ora_crc16(d, h->header_size)) return E_HCA_HEADER;
    size -= 8; pos = 8;
    if (size >= 0x10 && MAGIC(pos) == 0x666D7400u) {                                /* fmt, hca.cpp:667-688 */
        h->channels = RD(pos + 4, 1); h->rate = RD(pos + 5, 3); h->frame_count = RD(pos + 8, 4);
        h->delay = RD(pos + 12, 2); h->padding = RD(pos + 14, 2);
        if (!(h->channels >= 1 && h->channels <= 16)) return E_HCA_HEADER;
        if (h->frame_count == 0) return E_HCA_HEADER;
        if (!(h->rate >= 1 && h->rate <= 0x7FFFFF)) return E_HCA_HEADER;
        size -= 0x10; pos += 0x10;
    } else return E_HCA_HEADER;
    if (size >= 0x10 && MAGIC(pos) == 0x636F6D70u) {                                /* comp, hca.cpp:691-709 */
        h->frame_size = RD(pos + 4, 2); h->min_res = RD(pos + 6, 1); h->max_res = RD(pos + 7, 1);
        h->track_count = RD(pos + 8, 1); h->channel_config = RD(pos + 9, 1); h->total_bands = RD(pos + 10, 1);
        h->base_bands = RD(pos + 11, 1); h->stereo_bands = RD(pos + 12, 1); h->bands_per_hfr_group = RD(pos + 13, 1);
        h->ms_stereo = RD(pos + 14, 1);
        size -= 0x10; pos += 0x10;
    } else if (size >= 0x0c && MAGIC(pos) == 0x64656300u) {                         /* dec, hca.cpp:710-727 */
        h->frame_size = RD(pos + 4, 2); h->min_res = RD(pos + 6, 1); h->max_res = RD(pos + 7, 1);
        h->total_bands = RD(pos + 8, 1) + 1u; h->base_bands = RD(pos + 9, 1) + 1u;
        h->track_count = RD(pos + 10, 1) >> 4; h->channel_config = RD(pos + 10, 1) & 0xF; h->stereo_type = RD(pos + 11, 1);
        if (h->stereo_type == 0) h->base_bands = h->total_bands;
        h->stereo_bands = h->total_bands - h->base_bands;
        h->bands_per_hfr_group = 0;
        size -= 0x0c; pos += 0x0c;
    } else return E_HCA_HEADER;
    if (size >= 8 && MAGIC(pos) == 0x76627200u) {                                   /* vbr, hca.cpp:733-748 */
        uint32_t mx = RD(pos + 4, 2);
        if (!(h->frame_size == 0 && mx > 8 && mx <= 0x1FF)) return E_HCA_HEADER;
        size -= 8; pos += 8;
    }
    if (size >= 6 && MAGIC(pos) == 0x61746800u) { h->ath_type = RD(pos + 4, 2); pos += 6; } /* ath: size not reduced (hca.cpp:750-753) */
    else h->ath_type = h->version < 0x0200 ? 1 : 0;
    if (size >= 0x10 && MAGIC(pos) == 0x6C6F6F70u) {                                /* loop, hca.cpp:760-774 */
        h->loop_start_frame = RD(pos + 4, 4); h->loop_end_frame = RD(pos + 8, 4);
        h->loop_start_delay = RD(pos + 12, 2); h->loop_end_padding = RD(pos + 14, 2);
        h->loop_flag = 1;
        if (!(h->loop_start_frame <= h->loop_end_frame && h->loop_end_frame < h->frame_count)) return E_HCA_HEADER;
        size -= 0x10; pos += 0x10;
    }
    if (size >= 6 && MAGIC(pos) == 0x63697068u) {                                   /* ciph, hca.cpp:786-794 */
        h->ciph_pos = pos;
        h->ciph_type = RD(pos + 4, 2);
        if (!(h->ciph_type == 0 || h->ciph_type == 1 || h->ciph_type == 56)) return E_HCA_HEADER;
        size -= 6; pos += 6;
    }
    if (size >= 8 && MAGIC(pos) == 0x72766100u) { size -= 8; pos += 8; }            /* rva, hca.cpp:799-812: volume unused */
    if (size >= 5 && MAGIC(pos) == 0x636F6D6Du) {                                   /* comm, hca.cpp:814-830 */
        h->comment_len = RD(pos + 4, 1);
        if (h->comment_len > size) return E_HCA_HEADER;
        size -= 5 + h->comment_len; pos += 5 + h->comment_len;
    }
    if (!(h->frame_size >= 8 && h->frame_size <= 0xFFFF)) return E_HCA_HEADER;
    if (h->version <= 0x0200) { if (h->min_res != 1 || h->max_res != 15) return E_HCA_HEADER; }
    else if (h->min_res > h->max_res || h->max_res > 15) return E_HCA_HEADER;
    if (h->track_count == 0) h->track_count = 1;
    if (h->track_count > h->channels) return E_HCA_HEADER;
    if (h->total_bands > 128 || h->base_bands > 128 || h->stereo_bands > 128 || h->base_bands + h->stereo_bands > 128 ||
        h->bands_per_hfr_group > 128) return E_HCA_HEADER;
    {   /* hca.cpp:872-874, header_ceil2 */
        uint32_t a = h->total_bands - h->base_bands - h->stereo_bands, b = h->bands_per_hfr_group;
        h->hfr_group_count = b < 1 ? 0 : (a / b + ((a % b) ? 1 : 0));
    }
    if (h->hfr_group_count > 128) return E_HCA_HEADER;   /* total < base + stereo wraps the count: the reference then indexes scalefactors[128 - count] out of bounds */
    if (h->ath_type == 0) memset(h->ath, 0, 128);                                 /* hca.cpp:451-485 */
    else if (h->ath_type == 1) {
        uint32_t acc = 0;
        for (i = 0; i < 128; i++) {
            uint32_t index;
            acc += h->rate; index = acc >> 13;
            if (index >= 654) { memset(h->ath + i, 0xFF, 128 - i); break; }
            h->ath[i] = HCA_ATH_BASE[index];
        }
    } else return E_HCA_HEADER;
    channel_types(h->channels, h->track_count, h->stereo_bands, h->channel_config, h->type);
    for (i = 0; i < h->channels; i++)
        h->coded[i] = h->type[i] != CH_SECONDARY ? h->base_bands + h->stereo_bands : h->base_bands;
    if (h->ms_stereo) return E_HCA_HEADER;
    return 0;
#undef MAGIC
#undef RD
}

/* ------------------------------------------------------------------------------------------------
 * HCA frame decode (hca.cpp:1149-1254, 1290-1571, 1602-1736, 1898-2019, 339-360)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    uint8_t intensity[8], scalefactors[128], resolution[128], noises[128];
    uint32_t noise_count, valid_count;
    float gain[128], spectra[8][128], prev[128], wave[8][128];
} hca_chan;

typedef struct { hca_info h; uint8_t cipher[256]; uint32_t random; hca_chan ch[16]; } hca_dec;

static int unpack_scalefactors(hca_dec* D, uint32_t c, bitrd* br) {       /* hca.cpp:1290-1358 */
    hca_chan* ch = &D->ch[c];
    uint32_t cs = D->h.coded[c], extra = 0, i;
    uint32_t delta_bits = hca_read(br, 3);
    if (!(D->h.type[c] == CH_SECONDARY || D->h.hfr_group_count == 0 || D->h.version <= 0x0200)) {
        extra = D->h.hfr_group_count; cs += extra;
        if (cs > 128) return -5;
    }
    if (delta_bits >= 6) { for (i = 0; i < cs; i++) ch->scalefactors[i] = (uint8_t)hca_read(br, 6); }
    else if (delta_bits > 0) {
        uint8_t expected = (uint8_t)((1 << delta_bits) - 1), value = (uint8_t)hca_read(br, 6);
        ch->scalefactors[0] = value;
        for (i = 1; i < cs; i++) {
            uint8_t delta = (uint8_t)hca_read(br, delta_bits);
            if (delta == expected) value = (uint8_t)hca_read(br, 6);
            else {
                int t = (int)value + ((int)delta - (int)(expected >> 1));
                if (t < 0 || t >= 64) return -5;
                value = (uint8_t)(value - (expected >> 1) + delta);
                value &= 0x3F;
            }
            ch->scalefactors[i] = value;
        }
    } else memset(ch->scalefactors, 0, 128);
    for (i = 0; i < extra; i++) ch->scalefactors[127 - i] = ch->scalefactors[cs - i];
    return 0;
}

static int unpack_intensity(hca_dec* D, uint32_t c, bitrd* br) {          /* hca.cpp:1361-1441 */
    hca_chan* ch = &D->ch[c];
    uint32_t i, g = D->h.hfr_group_count;
    if (D->h.type[c] == CH_SECONDARY) {
        if (D->h.version <= 0x0200) {
            uint8_t v = (uint8_t)hca_peek(br, 4);
            ch->intensity[0] = v;
            if (v < 15) { br->pos += 4; for (i = 1; i < 8; i++) ch->intensity[i] = (uint8_t)hca_read(br, 4); }
        } else {
            uint8_t v = (uint8_t)hca_peek(br, 4);
            if (v < 15) {
                uint8_t db;
                br->pos += 4;
                db = (uint8_t)hca_read(br, 2);
                ch->intensity[0] = v;
                if (db == 3) { for (i = 1; i < 8; i++) ch->intensity[i] = (uint8_t)hca_read(br, 4); }
                else {
                    uint8_t bmax = (uint8_t)((2 << db) - 1), bits = (uint8_t)(db + 1);
                    for (i = 1; i < 8; i++) {
                        uint8_t delta = (uint8_t)hca_read(br, bits);
                        if (delta == bmax) v = (uint8_t)hca_read(br, 4);
                        else { v = (uint8_t)(v - (bmax >> 1) + delta); if (v > 15) return -5; }
                        ch->intensity[i] = v;
                    }
                }
            } else { br->pos += 4; for (i = 0; i < 8; i++) ch->intensity[i] = 7; }
        }
    } else if (D->h.version <= 0x0200) {
        for (i = 0; i < g; i++) ch->scalefactors[128 - g + i] = (uint8_t)hca_read(br, 6);
    }
    return 0;
}

static void calc_resolution_gain(hca_dec* D, uint32_t c, uint32_t packed_noise) { /* hca.cpp:1444-1507 */
    hca_chan* ch = &D->ch[c];
    uint32_t n = D->h.coded[c], i, nc = 0, vc = 0;
    for (i = 0; i < n; i++) {
        uint8_t res = 0, sf = ch->scalefactors[i];
        if (sf > 0) {
            int noise = D->h.ath[i] + (int)((packed_noise + i) >> 8);
            int cp = noise + 1 - ((5 * sf) >> 1);
            if (cp < 0) res = 15; else if (cp <= 65) res = HCA_CURVE_TO_RES[cp]; else res = 0;
            if (res > D->h.max_res) res = (uint8_t)D->h.max_res; else if (res < D->h.min_res) res = (uint8_t)D->h.min_res;
            if (res < 1) ch->noises[nc++] = (uint8_t)i; else ch->noises[127 - vc++] = (uint8_t)i;
        }
        ch->resolution[i] = res;
    }
    ch->noise_count = nc; ch->valid_count = vc;
    memset(ch->resolution + n, 0, 128 - n);
    for (i = 0; i < n; i++) ch->gain[i] = HCA_DEQ_SCALE[ch->scalefactors[i]] * HCA_DEQ_RANGE[ch->resolution[i]];
}

static void dequantize(hca_dec* D, uint32_t c, bitrd* br, uint32_t sf) {  /* hca.cpp:1540-1571 */
    hca_chan* ch = &D->ch[c];
    uint32_t n = D->h.coded[c], i;
    for (i = 0; i < n; i++) {
        float qc;
        uint8_t res = ch->resolution[i], bits = HCA_MAX_BITS[res];
        uint32_t code = hca_read(br, bits);
        if (res > 7) {
            int sc = (1 - (int)((code & 1) << 1)) * (int)(code >> 1);
            if (sc == 0) br->pos -= 1;
            qc = (float)sc;
        } else {
            uint32_t idx = ((uint32_t)res << 4) + code;
            br->pos += (uint32_t)((int)HCA_CODE_LEN[idx] - (int)bits);
            qc = (float)HCA_CODE_VAL[idx];
        }
        ch->spectra[sf][i] = ch->gain[i] * qc;
    }
    memset(&ch->spectra[sf][n], 0, sizeof(float) * (128 - n));
}

static int hca_unpack(hca_dec* D, uint8_t* frame) {                      /* hca.cpp:1149-1205 */
    bitrd br; uint32_t c, sf, i, nl, eb, packed;
    br.p = frame; br.nbits = D->h.frame_size * 8; br.pos = 0;
    if (hca_read(&br, 16) != 0xFFFF) return -4;
    if (ora_crc16(frame, D->h.frame_size)) return -3;
    for (i = 0; i < D->h.frame_size; i++) frame[i] = D->cipher[frame[i]];
    nl = hca_read(&br, 9); eb = hca_read(&br, 7);
    packed = (nl << 8) - eb;
    for (c = 0; c < D->h.channels; c++) {
        int e = unpack_scalefactors(D, c, &br);
        if (e < 0) return e;
        e = unpack_intensity(D, c, &br);      /* reference ignores this result (hca.cpp:1185) */
        (void)e;
        calc_resolution_gain(D, c, packed);
    }
    for (sf = 0; sf < 8; sf++) for (c = 0; c < D->h.channels; c++) dequantize(D, c, &br, sf);
    return (int)br.pos;
}

static void reconstruct_noise(hca_dec* D, uint32_t c, uint32_t sf) {       /* hca.cpp:1602-1635 */
    hca_chan* ch = &D->ch[c];
    uint32_t i, r = D->random;
    if (D->h.min_res > 0) return;
    if (ch->valid_count == 0 || ch->noise_count == 0) return;
    if (!(!D->h.ms_stereo || D->h.type[c] == CH_PRIMARY)) return;
    for (i = 0; i < ch->noise_count; i++) {
        uint32_t ri, ni, vi; int sc;
        r = 0x343FD * r + 0x269EC3;
        ri = 128 - ch->valid_count + (((r & 0x7FFF) * ch->valid_count) >> 15);
        ni = ch->noises[i]; vi = ch->noises[ri];
        sc = (int)ch->scalefactors[ni] - (int)ch->scalefactors[vi] + 62;
        sc = sc & ~(sc >> 31);
        ch->spectra[sf][ni] = HCA_SCALE_CONV[sc] * ch->spectra[sf][vi];
    }
    D->random = r;
}

static void reconstruct_hfr(hca_dec* D, uint32_t c, uint32_t sf) {         /* hca.cpp:1638-1683 */
    hca_chan* ch = &D->ch[c];
    const hca_info* h = &D->h;
    int start = (int)(h->stereo_bands + h->base_bands), high = start, low = start - 1, group, limit, i;
    const uint8_t* hfr_scales = &ch->scalefactors[128 - h->hfr_group_count];
    if (h->bands_per_hfr_group == 0 || h->type[c] == CH_SECONDARY) return;
    limit = h->version <= 0x0200 ? (int)h->hfr_group_count : (int)(h->hfr_group_count >> 1);
    for (group = 0; group < (int)h->hfr_group_count; group++) {
        int sub = group < limit ? 1 : 0;
        for (i = 0; i < (int)h->bands_per_hfr_group; i++) {
            int sc;
            if (high >= (int)h->total_bands || low < 0) break;
            sc = (int)hfr_scales[group] - (int)ch->scalefactors[low] + 63;
            sc = sc & ~(sc >> 31);
            ch->spectra[sf][high] = HCA_SCALE_CONV[sc] * ch->spectra[sf][low];
            high += 1; low -= sub;
        }
    }
    if (high >= 1) ch->spectra[sf][high - 1] = 0.0f;
}

static void intensity_stereo(hca_dec* D, uint32_t c, uint32_t sf) {        /* hca.cpp:1696-1714 */
    const hca_info* h = &D->h;
    float rl, rr; uint32_t b;
    if (h->type[c] != CH_PRIMARY) return;
    rl = HCA_INTENSITY_RATIO[D->ch[c + 1].intensity[sf]];
    rr = 2.0f - rl;
    for (b = h->base_bands; b < h->total_bands; b++) {
        float l = D->ch[c].spectra[sf][b];
        D->ch[c].spectra[sf][b] = l * rl;
        D->ch[c + 1].spectra[sf][b] = l * rr;
    }
}

static void imdct(hca_chan* ch, uint32_t sf) {                             /* hca.cpp:1898-2019; SURVEY Appendix B */
    float a[128], b[128], *x = ch->spectra[sf], *y = a;
    uint32_t i, j, k;
    memcpy(b, x, sizeof b); x = b;
    for (i = 0; i < 7; i++) {                       /* sum / difference stages */
        uint32_t c = 64u >> i; float* t;
        for (j = 0; j < (1u << i); j++)
            for (k = 0; k < c; k++) {
                float p = x[2 * (c * j + k)], q = x[2 * (c * j + k) + 1];
                y[2 * c * j + k] = p + q;
                y[2 * c * j + c + k] = p - q;
            }
        t = x; x = y; y = t;
    }
    for (i = 0; i < 7; i++) {                       /* rotation stages */
        uint32_t c = 1u << i; float* t;
        for (j = 0; j < (64u >> i); j++)
            for (k = 0; k < c; k++) {
                uint32_t tw = c * j + k;
                float p = x[2 * c * j + k], q = x[2 * c * j + c + k];
                float s = HCA_IMDCT_SIN[i][tw], co = HCA_IMDCT_COS[i][tw];
                y[2 * c * j + k] = p * s - q * co;
                y[2 * c * j + 2 * c - 1 - k] = p * co + q * s;
            }
        t = x; x = y; y = t;
    }
    for (i = 0; i < 64; i++) {                      /* window + overlap-add */
        ch->wave[sf][i] = HCA_WINDOW[i] * x[i + 64] + ch->prev[i];
        ch->wave[sf][i + 64] = HCA_WINDOW[i + 64] * x[127 - i] - ch->prev[i + 64];
        ch->prev[i] = HCA_WINDOW[127 - i] * x[63 - i];
        ch->prev[i + 64] = HCA_WINDOW[63 - i] * x[i];
    }
}

static void hca_transform(hca_dec* D) {                                    /* hca.cpp:1207-1233 */
    uint32_t sf, c;
    for (sf = 0; sf < 8; sf++) {
        for (c = 0; c < D->h.channels; c++) { reconstruct_noise(D, c, sf); reconstruct_hfr(D, c, sf); }
        if (D->h.stereo_bands > 0) for (c = 0; c + 1 < D->h.channels; c++) intensity_stereo(D, c, sf);
        for (c = 0; c < D->h.channels; c++) imdct(&D->ch[c], sf);
    }
}

static int32_t x86_cvtt(float f) {  /* (int)f as the x86-64 reference build evaluates it */
    if (!(f > -2147483904.0f && f < 2147483648.0f)) return INT32_MIN;
    return (int32_t)f;
}

static int hca_decode_impl(const uint8_t* d, size_t len, uint64_t key, uint16_t subkey, uint8_t** out, size_t* out_len,
                           float** fout, size_t* fcount) {
    hca_dec* D; int rc; uint32_t f, sf, s, c, hs, spc, total_samples; uint8_t* frame; uint8_t* o = 0; float* fo = 0;
    size_t wsize = 0; uint32_t wh = 0;
    if (!d || len < 8) return E_HCA_HEADER;
    D = (hca_dec*)calloc(1, sizeof *D);
    if (!D) return E_NOMEM;
    hs = be16(d + 6);
    rc = hca_parse_header(d, len, hs, &D->h);
    if (rc) { free(D); return rc; }
    ora_cipher_table(D->h.ciph_type, mix_key(key, subkey), D->cipher);
    D->random = 1;
    total_samples = D->h.frame_count * 1024;
    if (total_samples < D->h.delay + D->h.padding) { free(D); return E_HCA_HEADER; }
    spc = total_samples - D->h.delay - D->h.padding;
    if (fout) { fo = (float*)calloc((size_t)total_samples * D->h.channels + 1, sizeof(float)); if (!fo) { free(D); return E_NOMEM; } }
    else {
        uint32_t ls = D->h.loop_start_frame * 1024 + D->h.loop_start_delay - D->h.delay;
        uint32_t le = D->h.loop_end_frame * 1024 + (1024 - D->h.loop_end_padding) - D->h.delay;
        wsize = (D->h.loop_flag ? 0x70 : 0x2C) + (size_t)spc * D->h.channels * 2;
        o = (uint8_t*)calloc(1, wsize);
        if (!o) { free(D); return E_NOMEM; }
        wh = wav_write_header(o, D->h.channels, D->h.rate, spc, (int)D->h.loop_flag, ls, le);
    }
    frame = (uint8_t*)malloc(D->h.frame_size);
    for (f = 0; f < D->h.frame_count; f++) {
        size_t off = (size_t)D->h.header_size + (size_t)f * D->h.frame_size;
        if (!fout && (spc == 0 || (uint64_t)f * 1024 >= (uint64_t)D->h.delay + spc)) break;     /* driver stops once samples_to_do are out */
        if (off + D->h.frame_size > len) { rc = E_HCA_DECODE; break; }
        memcpy(frame, d + off, D->h.frame_size);
        rc = hca_unpack(D, frame);
        if (rc < 0) { rc = E_HCA_DECODE; break; }
        rc = 0;
        hca_transform(D);
        for (sf = 0; sf < 8; sf++) for (s = 0; s < 128; s++) for (c = 0; c < D->h.channels; c++) {
            uint32_t n = f * 1024 + sf * 128 + s;
            float v = D->ch[c].wave[sf][s];
            if (fout) fo[(size_t)n * D->h.channels + c] = v;
            else if (n >= D->h.delay && n - D->h.delay < spc) {
                int32_t q = x86_cvtt(v * 32768.0f);
                if (q > 32767) q = 32767; else if (q < -32768) q = -32768;
                put_le16(o + wh + ((size_t)(n - D->h.delay) * D->h.channels + c) * 2, (uint32_t)q & 0xFFFF);
            }
        }
    }
    free(frame);
    if (rc) { free(o); free(fo); free(D); return rc; }
    if (fout) { *fout = fo; *fcount = (size_t)total_samples * D->h.channels; }
    else { *out = o; *out_len = wsize; }
    free(D);
    return 0;
}

int ora_hca_decode(const uint8_t* d, size_t len, uint64_t key, uint16_t subkey, uint8_t** out, size_t* out_len) {
    if (!out || !out_len) return E_ARG;
    return hca_decode_impl(d, len, key, subkey, out, out_len, 0, 0);
}
int ora_hca_decode_float(const uint8_t* d, size_t len, uint64_t key, uint16_t subkey, float** out, size_t* out_count) {
    if (!out || !out_count) return E_ARG;
    return hca_decode_impl(d, len, key, subkey, 0, 0, out, out_count);
}

/* ------------------------------------------------------------------------------------------------
 * HCA crypt (hca.cpp:3166-3250, 3271-3337)
 * ---------------------------------------------------------------------------------------------- */
int ora_hca_crypt(uint8_t* d, size_t len, uint32_t encrypt, uint32_t type, uint64_t key, uint16_t subkey) {
    hca_info h; uint8_t t[256], inv[256]; uint32_t i, f, hs, size, pos; int rc;
    if (!d || len < 8) return E_HCA_HEADER;
    hs = be16(d + 6);
    rc = hca_parse_header(d, len, hs, &h);
    if (rc) return E_HCA_HEADER;
    if (encrypt == 1) h.ciph_type = type;
    if (ora_cipher_table(h.ciph_type, mix_key(key, subkey), t)) return E_HCA_HEADER;
    if (encrypt) { for (i = 0; i < 256; i++) inv[t[i]] = (uint8_t)i; memcpy(t, inv, 256); }
    if ((size_t)hs + (size_t)h.frame_count * h.frame_size > len) return E_HCA_HEADER;
    for (f = 0; f < h.frame_count; f++) {
        uint8_t* fr = d + hs + (size_t)f * h.frame_size;
        for (i = 0; i < h.frame_size; i++) fr[i] = t[fr[i]];
        put_be16(fr + h.frame_size - 2, ora_crc16(fr, h.frame_size - 2));
    }
    /* chunk magic masks, hca.cpp:3166-3250; bounded reader over the header (hca.cpp:3171), `ath` keeps `size` (3203-3206) */
    size = hs; pos = 0;
#define MAGIC(at) (rd_be(d, hs, (at), 4) & 0x7F7F7F7Fu)
#define FLIP3(at) do { d[(at)] ^= 0x80; d[(at) + 1] ^= 0x80; d[(at) + 2] ^= 0x80; } while (0)
#define FLIP4(at) do { FLIP3(at); d[(at) + 3] ^= 0x80; } while (0)
    if (MAGIC(pos) == 0x48434100u) { FLIP3(pos); pos += 8; size -= 8; }
    if (size >= 0x10 && MAGIC(pos) == 0x666D7400u) { FLIP3(pos); pos += 16; size -= 16; }
    if (size >= 0x10 && MAGIC(pos) == 0x636F6D70u) { FLIP4(pos); pos += 16; size -= 16; }
    else if (size >= 0x0c && MAGIC(pos) == 0x64656300u) { FLIP3(pos); pos += 12; size -= 12; }
    if (size >= 8 && MAGIC(pos) == 0x76627200u) { FLIP3(pos); pos += 8; size -= 8; }
    if (size >= 6 && MAGIC(pos) == 0x61746800u) { FLIP3(pos); pos += 6; }
    if (size >= 0x10 && MAGIC(pos) == 0x6C6F6F70u) { FLIP4(pos); pos += 16; size -= 16; }
    if (size >= 6 && MAGIC(pos) == 0x63697068u) {
        FLIP4(pos);
        if ((size_t)pos + 6 <= hs) put_be16(d + pos + 4, encrypt == 1 ? (type & 0xFFFF) : 0);   /* (the reference writes it out of bounds otherwise) */
        pos += 6; size -= 6;
    }
    if (size >= 8 && MAGIC(pos) == 0x72766100u) { FLIP3(pos); pos += 8; size -= 8; }
    if (size >= 5 && MAGIC(pos) == 0x636F6D6Du) { uint32_t cl = rd_be(d, hs, pos + 4, 1); FLIP4(pos); pos += 5 + cl; size -= 5 + cl; }
    if (size >= 4 && MAGIC(pos) == 0x70616400u) { FLIP3(pos); }
    put_be16(d + hs - 2, ora_crc16(d, hs - 2));
#undef MAGIC
#undef FLIP3
#undef FLIP4
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * HCA encoder (hca.cpp:2206-2462 setup, 2470-2988 frame, 2990-3107 feeding, 3109-3164 header)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    uint8_t type; uint32_t coded;
    uint8_t intensity[8], scalefactors[128], resolution[128];
    float spectra[8][128], scaled[128][8], prev[128], wave[8][128], hfr_avg[8];
    int quant[8][128];
    int hfr_scales[8], header_bits, delta_bits;
} enc_chan;

typedef struct {
    uint32_t channels, rate, frame_size, frame_count, delay, padding, channel_config;
    uint32_t total_bands, base_bands, stereo_bands, hfr_group_count, bands_per_hfr_group, hfr_band_count;
    uint32_t header_size, spc;
    uint32_t loop_flag, loop_start_frame, loop_end_frame, loop_start_delay, loop_end_padding;
    uint32_t pre_samples, post_samples;                      /* BufferPreSamples, PostSamples */
    int noise_level, eval_boundary;
    enc_chan ch[16];
} hca_enc;

static int div_round_up(int v, int d) { return (int)ceilf((float)v / d); }   /* hca.cpp:182-184 */
static uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }

static int enc_setup(hca_enc* E, uint32_t channels, uint32_t rate, uint32_t spc, uint32_t quality) {
    uint32_t pcm_bitrate = rate * channels * 16, max_bitrate = pcm_bitrate / 4, bitrate, cutoff = rate / 2;
    uint32_t hfr_ratio, cutoff_ratio, total, hfr_start, stereo_start, hfr_bands, bpg, groups = 0, i;
    int ratio = 6;
    uint8_t types[16];
    memset(E, 0, sizeof *E);
    E->channels = channels; E->rate = rate; E->spc = spc; E->delay = 128;
    switch (quality) {                                      /* hca.cpp:2206-2234 */
        case 0: ratio = 4; break; case 1: ratio = 6; break; case 2: ratio = 8; break;
        case 3: ratio = channels == 1 ? 10 : 12; break; case 4: ratio = channels == 1 ? 12 : 16; break;
        default: break;
    }
    if (pcm_bitrate == 0) return E_HCA_CHCONF;
    bitrate = pcm_bitrate / (uint32_t)ratio;
    if (bitrate > max_bitrate) bitrate = max_bitrate;
    if (bitrate == 0) return E_HCA_CHCONF;
    E->frame_size = bitrate * 1024 / rate / 8;              /* hca.cpp:2236-2270 */
    if (channels <= 1 || pcm_bitrate / bitrate <= 6) { hfr_ratio = 6; cutoff_ratio = 12; } else { hfr_ratio = 8; cutoff_ratio = 16; }
    if (bitrate < pcm_bitrate / cutoff_ratio) cutoff = umin(cutoff, cutoff_ratio * bitrate / (32 * channels));
    total = (uint32_t)round((double)cutoff * 256.0 / rate);
    hfr_start = umin(total, (uint32_t)round(((double)hfr_ratio * bitrate * 128.0) / pcm_bitrate));
    stereo_start = hfr_ratio == 6 ? hfr_start : (hfr_start + 1) / 2;
    hfr_bands = total - hfr_start;
    bpg = (uint32_t)div_round_up((int)hfr_bands, 8);
    if (bpg > 0) groups = (uint32_t)div_round_up((int)hfr_bands, (int)bpg);
    E->total_bands = total; E->base_bands = stereo_start; E->stereo_bands = hfr_start - stereo_start;
    E->hfr_group_count = groups; E->bands_per_hfr_group = bpg;
    if (bpg > 0) {                                          /* hca.cpp:2272-2277 */
        E->hfr_band_count = E->total_bands - E->base_bands - E->stereo_bands;
        E->hfr_group_count = (uint32_t)div_round_up((int)E->hfr_band_count, (int)bpg);
    }
    if (channels > 8) return E_HCA_CHCONF;                   /* hca.cpp:2279-2290 */
    E->channel_config = HCA_DEFAULT_CHANNEL_CONFIG[channels];
    if (HCA_VALID_CHANNEL_CONFIG[channels - 1][E->channel_config] != 1) return E_HCA_CHCONF;
    E->header_size = 96;                                     /* hca.cpp:2307-2321, no comment, no loop */
    E->frame_count = (uint32_t)div_round_up((int)(spc + E->delay), 1024);
    E->padding = E->frame_count * 1024 - E->delay - spc;
    channel_types(channels, 1, E->stereo_bands, E->channel_config, types);
    for (i = 0; i < channels; i++) {
        E->ch[i].type = types[i];
        E->ch[i].coded = types[i] == CH_SECONDARY ? E->base_bands : E->base_bands + E->stereo_bands;
    }
    return 0;
}

/* Loop branch of initHCAEncode (hca.cpp:2439-2462) with CalculateLoopInfo (2292-2306) and CalculateHeaderSize (2308-2321).
 * Runs after enc_setup; `column_size` is the WAV's total interleaved sample count (the reference clamps with it as is). */
static void enc_setup_loop(hca_enc* E, uint32_t loop_start, uint32_t loop_end, uint32_t column_size) {
    uint32_t input, ls, le, off, pad_bytes, pad_frames;
    E->loop_flag = 1;
    E->spc = umin(loop_end, column_size);
    E->delay += (uint32_t)next_multiple((int)loop_start, 1024) - loop_start;
    ls = loop_start + E->delay; le = loop_end + E->delay;
    E->loop_start_frame = ls / 1024; E->loop_start_delay = ls % 1024;
    E->loop_end_frame = le / 1024; E->loop_end_padding = 1024 - le % 1024;
    if (E->loop_end_padding == 1024) { E->loop_end_frame--; E->loop_end_padding = 0; }
    input = umin((uint32_t)next_multiple((int)E->spc, 128), column_size) + 256;
    E->post_samples = input - E->spc;
    off = E->header_size + E->frame_size * E->loop_start_frame;
    pad_bytes = (uint32_t)next_multiple((int)off, 2048) - off;
    pad_frames = pad_bytes / E->frame_size;
    E->delay += pad_frames * 1024;
    E->loop_start_frame += pad_frames; E->loop_end_frame += pad_frames;
    E->header_size += pad_bytes % E->frame_size;
    E->frame_count = (uint32_t)div_round_up((int)(input + E->delay), 1024);
    E->padding = E->frame_count * 1024 - E->delay - input;
    E->pre_samples = E->delay - 128;
}

static void enc_dct4(enc_chan* ch, const float* in, uint32_t sf) {          /* hca.cpp:2481-2527 */
    float t[128]; int i, stage;
    for (i = 0; i < 64; i++) {
        float a = in[2 * i], b = in[127 - 2 * i], s = HCA_ENC_SIN[7][i], c = HCA_ENC_COS[7][i];
        t[2 * i] = a * c + b * s;
        t[2 * i + 1] = a * s - b * c;
    }
    for (stage = 0; stage < 6; stage++) {
        int block_count = 1 << stage, bits = 6 - stage, half_bits = bits - 1, bsz = 1 << bits, bh = 1 << half_bits, block;
        for (block = 0; block < block_count; block++)
            for (i = 0; i < bh; i++) {
                int fp = (block * bsz + i) * 2, bp = fp + bsz;
                float a = t[fp] - t[bp], b = t[fp + 1] - t[bp + 1];
                float s = HCA_ENC_SIN[half_bits][i], c = HCA_ENC_COS[half_bits][i];
                t[fp] += t[bp];
                t[fp + 1] += t[bp + 1];
                t[bp] = a * c + b * s;
                t[bp + 1] = a * s - b * c;
            }
    }
    for (i = 0; i < 128; i++) ch->spectra[sf][i] = t[HCA_ENC_SHUFFLE[i]] * 0.125f;
}

static void enc_mdct(enc_chan* ch, uint32_t sf) {                            /* hca.cpp:2529-2553 */
    float s[128]; int i;
    for (i = 0; i < 64; i++) {
        float a = HCA_WINDOW[63 - i] * -ch->wave[sf][64 + i];
        float b = -HCA_WINDOW[64 + i] * ch->wave[sf][63 - i];
        float c = HCA_WINDOW[i] * ch->prev[i];
        float d = -HCA_WINDOW[127 - i] * ch->prev[127 - i];
        s[i] = a - b;
        s[64 + i] = c - d;
    }
    enc_dct4(ch, s, sf);
    memcpy(ch->prev, ch->wave[sf], sizeof ch->prev);
}

static void enc_intensity(hca_enc* E) {                                      /* hca.cpp:2561-2609 */
    uint32_t c, sf, b;
    if (E->stereo_bands == 0) return;
    for (c = 0; c < E->channels; c++) {
        if (E->ch[c].type != CH_PRIMARY) continue;
        for (sf = 0; sf < 8; sf++) {
            float* l = E->ch[c].spectra[sf]; float* r = E->ch[c + 1].spectra[sf];
            float el = 0, er = 0, et = 0, elr, stored, ratio; int q = 1;
            for (b = E->base_bands; b < E->total_bands; b++) { el += fabsf(l[b]); er += fabsf(r[b]); et += fabsf(l[b] + r[b]); }
            et *= 2;
            elr = er + el;
            stored = 2 * el / elr;
            ratio = elr / et;
            if (ratio < 0.5) ratio = 0.5f;
            else if ((double)ratio > sqrt(2) / 2) ratio = (float)(sqrt(2) / 2);
            if (er > 0 || el > 0) { while (q < 13 && HCA_ENC_INTENSITY_BOUNDS[q] >= stored) q++; }
            else { q = 0; ratio = 1; }
            E->ch[c + 1].intensity[sf] = (uint8_t)q;
            for (b = E->base_bands; b < E->total_bands; b++) { l[b] = (l[b] + r[b]) * ratio; r[b] = 0; }
        }
    }
}

static int find_scalefactor(float v) {                                       /* hca.cpp:2611-2623 */
    uint32_t low = 0, high = 63;
    while (low < high) { uint32_t mid = (low + high) / 2; if (HCA_DEQ_SCALE[mid] <= v) low = mid + 1; else high = mid; }
    return (int)low;
}

static void enc_scalefactors_and_scale(hca_enc* E) {                         /* hca.cpp:2625-2654 */
    uint32_t c, b, sf;
    for (c = 0; c < E->channels; c++) {
        enc_chan* ch = &E->ch[c];
        for (b = 0; b < ch->coded; b++) {
            float mx = 0;
            for (sf = 0; sf < 8; sf++) { float a = fabsf(ch->spectra[sf][b]); mx = a > mx ? a : mx; }   /* std::max(coeff, max) */
            ch->scalefactors[b] = (uint8_t)find_scalefactor(mx);
        }
        memset(ch->scalefactors + ch->coded, 0, 128 - ch->coded);
        for (b = 0; b < ch->coded; b++) {
            int s = ch->scalefactors[b];
            for (sf = 0; sf < 8; sf++) {
                float a = ch->spectra[sf][b] * HCA_ENC_SCALE[s];
                if (a > 0.9999999f) a = 0.9999999f; else if (a < -0.9999999f) a = -0.9999999f;
                ch->scaled[b][sf] = s == 0 ? 0 : a;
            }
        }
    }
}

static void enc_hfr(hca_enc* E) {                                            /* hca.cpp:2656-2706 */
    uint32_t c; int start = (int)(E->stereo_bands + E->base_bands), group, i, sf;
    if (E->hfr_group_count == 0) return;
    for (c = 0; c < E->channels; c++) {
        enc_chan* ch = &E->ch[c]; int band;
        if (ch->type == CH_SECONDARY) continue;
        for (group = 0, band = start; group < (int)E->hfr_group_count; group++) {
            float sum = 0.0f; int count = 0;
            for (i = 0; i < (int)E->bands_per_hfr_group && band < 128; band++, i++) {
                for (sf = 0; sf < 8; sf++) sum += fabsf(ch->spectra[sf][band]);
                count += 8;
            }
            ch->hfr_avg[group] = sum / count;
        }
    }
    {
        int hb = (int)umin(E->hfr_band_count, E->total_bands - E->hfr_band_count);
        for (c = 0; c < E->channels; c++) {
            enc_chan* ch = &E->ch[c]; int band;
            if (ch->type == CH_SECONDARY) continue;
            for (group = 0, band = 0; group < (int)E->hfr_group_count; group++) {
                float sum = 0.0f, avg; int count = 0;
                for (i = 0; i < (int)E->bands_per_hfr_group && band < hb; band++, i++) {
                    for (sf = 0; sf < 8; sf++) sum += fabsf(ch->scaled[start - band - 1][sf]);
                    count += 8;
                }
                avg = sum / count;
                if (avg > 0.0) {
                    double m = 1.0 / avg, r2 = sqrt(2);
                    ch->hfr_avg[group] = (float)(ch->hfr_avg[group] * (m < r2 ? m : r2));
                }
                ch->hfr_scales[group] = find_scalefactor(ch->hfr_avg[group]);
            }
        }
    }
}

static void enc_header_length(hca_enc* E) {                                  /* hca.cpp:2708-2750 */
    uint32_t c;
    for (c = 0; c < E->channels; c++) {
        enc_chan* ch = &E->ch[c]; int empty = 1, db, band; uint32_t i;
        for (i = 0; i < ch->coded; i++) if (ch->scalefactors[i] != 0) { empty = 0; break; }
        if (empty) { ch->header_bits = 3; ch->delta_bits = 0; }
        else {
            int min_db = 6, min_len = 3 + 6 * (int)ch->coded;
            for (db = 1; db < 6; db++) {
                int maxd = (1 << (db - 1)) - 1, length = 3 + 6;
                for (band = 1; band < (int)ch->coded; band++) {
                    int delta = (int)ch->scalefactors[band] - (int)ch->scalefactors[band - 1];
                    length += abs(delta) > maxd ? db + 6 : db;
                }
                if (length < min_len) { min_len = length; min_db = db; }
            }
            ch->header_bits = min_len; ch->delta_bits = min_db;
        }
        if (ch->type == CH_SECONDARY) ch->header_bits += 32;
        else if (E->hfr_group_count > 0) ch->header_bits += 6 * (int)E->hfr_group_count;
    }
}

static int enc_resolution(int sf, int noise) {                               /* hca.cpp:2752-2761 */
    int cp;
    if (sf == 0) return 0;
    cp = noise - 5 * sf / 2 + 2;
    if (cp < 0) cp = 0; else if (cp > 58) cp = 58;
    return HCA_ENC_CURVE_TO_RES[cp];
}

static int enc_used_bits(hca_enc* E, int noise_level, int eval_boundary) {   /* hca.cpp:2763-2790 */
    int length = 16 + 16 + 16; uint32_t c, i; int j;
    for (c = 0; c < E->channels; c++) {
        enc_chan* ch = &E->ch[c];
        length += ch->header_bits;
        for (i = 0; i < ch->coded; i++) {
            int noise = (int)i < eval_boundary ? noise_level - 1 : noise_level;
            int res = enc_resolution(ch->scalefactors[i], noise);
            if (res >= 8) {
                int bits = HCA_MAX_BITS[res] - 1; float dz = HCA_ENC_DEAD_ZONE[res];
                for (j = 0; j < 8; j++) { length += bits; if (fabsf(ch->scaled[i][j]) >= dz) length++; }
            } else {
                float inv = HCA_ENC_INV_STEP[res], up = inv + 1; int down = (int)(inv + 0.5 - 8);
                for (j = 0; j < 8; j++) { int q = (int)(ch->scaled[i][j] * inv + up) - down; length += HCA_ENC_CODE_LEN[res][q]; }
            }
        }
    }
    return length;
}

static int enc_noise_level(hca_enc* E) {                                     /* hca.cpp:2792-2832 */
    int highest = (int)(E->base_bands + E->stereo_bands) - 1, avail = (int)E->frame_size * 8, level;
    for (;;) {
        int low = 0, high = 255, mid_value = 0;
        while (low != high) {
            int mid = (low + high) / 2;
            mid_value = enc_used_bits(E, mid, 0);
            if (mid_value > avail) low = mid + 1; else high = mid;
        }
        level = (low == 255 && mid_value > avail) ? -1 : low;
        if (level >= 0) break;
        highest -= 2;
        if (highest < 0) return -3;
        { uint32_t c; for (c = 0; c < E->channels; c++) { E->ch[c].scalefactors[highest + 1] = 0; E->ch[c].scalefactors[highest + 2] = 0; } }
        enc_header_length(E);
    }
    E->noise_level = level;
    return 0;
}

static int enc_eval_boundary(hca_enc* E) {                                   /* hca.cpp:2834-2866 */
    int avail = (int)E->frame_size * 8, low = 0, high = 127, level;
    if (E->noise_level == 0) { E->eval_boundary = 0; return 0; }
    while (abs(high - low) > 1) {
        int mid = (low + high) / 2, v = enc_used_bits(E, E->noise_level, mid);
        if (avail < v) high = mid - 1; else low = mid;
    }
    if (low == high) level = low < 127 ? low : -1;
    else level = enc_used_bits(E, E->noise_level, high) > avail ? low : high;
    if (level < 0) return -4;
    E->eval_boundary = level;
    return 0;
}

static void enc_quantize(hca_enc* E) {                                       /* hca.cpp:2868-2892 */
    uint32_t c, i, sf;
    for (c = 0; c < E->channels; c++) {
        enc_chan* ch = &E->ch[c];
        for (i = 0; i < ch->coded; i++)
            ch->resolution[i] = (uint8_t)enc_resolution(ch->scalefactors[i], (int)i < E->eval_boundary ? E->noise_level - 1 : E->noise_level);
        memset(ch->resolution + ch->coded, 0, 128 - ch->coded);
        for (i = 0; i < ch->coded; i++) {
            float inv = HCA_ENC_INV_STEP[ch->resolution[i]], up = inv + 1; int down = (int)(inv + 0.5);
            for (sf = 0; sf < 8; sf++) ch->quant[sf][i] = (int)(ch->scaled[i][sf] * inv + up) - down;
        }
    }
}

static void enc_pack(hca_enc* E, uint8_t* out) {                             /* hca.cpp:2894-2963 */
    bitwr w; uint32_t c, i, sf, k; uint16_t crc;
    memset(out, 0, E->frame_size);
    put_be16(out, 0xFFFF);
    w.p = out + 2; w.nbits = (E->frame_size - 2) * 8; w.pos = 0;
    bw_write(&w, E->noise_level, 9);
    bw_write(&w, E->eval_boundary, 7);
    for (c = 0; c < E->channels; c++) {
        enc_chan* ch = &E->ch[c]; int db = ch->delta_bits;
        bw_write(&w, db, 3);
        if (db == 6) { for (i = 0; i < ch->coded; i++) bw_write(&w, ch->scalefactors[i], 6); }
        else if (db != 0) {
            int maxd = (1 << (db - 1)) - 1, esc = (1 << db) - 1;
            bw_write(&w, ch->scalefactors[0], 6);
            for (i = 1; i < ch->coded; i++) {
                int delta = (int)ch->scalefactors[i] - (int)ch->scalefactors[i - 1];
                if (abs(delta) > maxd) { bw_write(&w, esc, (uint32_t)db); bw_write(&w, ch->scalefactors[i], 6); }
                else bw_write(&w, maxd + delta, (uint32_t)db);
            }
        }
        if (ch->type == CH_SECONDARY) { for (k = 0; k < 8; k++) bw_write(&w, ch->intensity[k], 4); }
        else if (E->hfr_group_count > 0) { for (k = 0; k < E->hfr_group_count; k++) bw_write(&w, ch->hfr_scales[k], 6); }
    }
    for (sf = 0; sf < 8; sf++)
        for (c = 0; c < E->channels; c++) {
            enc_chan* ch = &E->ch[c];
            for (i = 0; i < ch->coded; i++) {
                int res = ch->resolution[i], q = ch->quant[sf][i];
                if (res == 0) continue;
                if (res < 8) bw_write(&w, HCA_ENC_CODE[res][q + 8], HCA_ENC_CODE_LEN[res][q + 8]);
                else { bw_write(&w, abs(q), (uint32_t)HCA_MAX_BITS[res] - 1); if (q != 0) bw_write(&w, q > 0 ? 0 : 1, 1); }
            }
        }
    crc = ora_crc16(out, E->frame_size - 2);
    put_be16(out + E->frame_size - 2, crc);
}

static int enc_frame(hca_enc* E, const int16_t* pcm /* 1024*channels interleaved */, uint8_t* out) { /* hca.cpp:2965-2988 */
    uint32_t c, sf, i;
    for (c = 0; c < E->channels; c++)
        for (sf = 0; sf < 8; sf++) for (i = 0; i < 128; i++)
            E->ch[c].wave[sf][i] = (float)(pcm[(sf * 128 + i) * E->channels + c] * (float)(1.0f / 32768.0f));
    for (c = 0; c < E->channels; c++) for (sf = 0; sf < 8; sf++) enc_mdct(&E->ch[c], sf);
    enc_intensity(E);
    enc_scalefactors_and_scale(E);
    enc_hfr(E);
    enc_header_length(E);
    if (enc_noise_level(E) < 0) return E_HCA_ENCODE;
    if (enc_eval_boundary(E) < 0) return E_HCA_ENCODE;
    enc_quantize(E);
    enc_pack(E, out);
    return 0;
}

int ora_hca_encode(const uint8_t* wav, size_t len, uint32_t force_no_loop, uint32_t quality, uint8_t** out, size_t* out_len) {
    wavin w; hca_enc* E; int rc, looping; uint8_t* o; int16_t* seq; uint32_t f, loop_start = 0, have_spc, pos; size_t total;
    if (!wav || !out || !out_len) return E_ARG;
    rc = wav_parse(wav, len, &w);
    if (rc) return rc;
    looping = w.looping && !force_no_loop;
    if (looping && w.num_loops == 0) return E_UNSUPPORTED;   /* reference reads an empty loop array here */
    rc = wav_pcm16(&w);
    if (rc) { wav_release(&w); return rc; }
    E = (hca_enc*)calloc(1, sizeof *E);
    if (!E) { wav_release(&w); return E_NOMEM; }
    rc = enc_setup(E, w.channels, w.rate, w.column_size / w.channels, quality);
    if (rc) { free(E); wav_release(&w); return rc; }
    have_spc = w.column_size / w.channels;
    if (looping) { loop_start = le32(w.loops + 8); enc_setup_loop(E, loop_start, le32(w.loops + 12), w.column_size); }
    else E->post_samples = 128;
    total = (size_t)E->header_size + (size_t)E->frame_count * E->frame_size;
    o = (uint8_t*)calloc(1, total);
    /* The feeding state machine (Encode / HcaEncode / PreEncode / SaveLoopAudio / EncodeMainAudio / EncodePostAudio,
     * hca.cpp:2990-3107) amounts to encoding this sample sequence frame by frame: whole zero frames while more than 1024
     * pre-samples remain, the remaining pre-samples as copies of the first sample, the main audio [0, spc), PostSamples
     * of audio from the loop start (only what SaveLoopAudio saw: chunks of 1024 up to the one where the main audio ends),
     * zeros to the end.  Samples past the end of the WAV data read as zero (the reference reads out of bounds there). */
    seq = (int16_t*)calloc((size_t)E->frame_count * 1024 * E->channels + 1, 2);
    if (!o || !seq) { free(o); free(seq); free(E); wav_release(&w); return E_NOMEM; }
    {
        uint32_t pre = E->pre_samples, pos = 0, k, c, seen_end, cap = E->frame_count * 1024;
        while (pre > 1024) { pos += 1024; pre -= 1024; }
        for (k = 0; k < pre && pos + k < cap; k++) for (c = 0; c < E->channels; c++) seq[(size_t)(pos + k) * E->channels + c] = have_spc ? w.pcm[c] : 0;
        pos += pre;
        for (k = 0; k < E->spc && pos + k < cap; k++) if (k < have_spc)
            memcpy(seq + (size_t)(pos + k) * E->channels, w.pcm + (size_t)k * E->channels, (size_t)E->channels * 2);
        pos += E->spc;
        seen_end = E->spc == 0 ? 1024 : ((E->spc - 1) / 1024 + 1) * 1024;
        if (looping) for (k = 0; k < E->post_samples && pos + k < cap; k++) {
            uint64_t src = (uint64_t)loop_start + k;
            if (src < seen_end && src < have_spc) memcpy(seq + (size_t)(pos + k) * E->channels, w.pcm + (size_t)src * E->channels, (size_t)E->channels * 2);
        }
    }
    for (f = 0; f < E->frame_count && !rc; f++)
        rc = enc_frame(E, seq + (size_t)f * 1024 * E->channels, o + E->header_size + (size_t)f * E->frame_size);
    free(seq);
    if (rc) { free(o); free(E); wav_release(&w); return rc; }
    /* hca.cpp:3109-3164 */
    put_be32(o, 0x48434100u); put_be16(o + 4, 0x0200); put_be16(o + 6, E->header_size);
    put_be32(o + 8, 0x666D7400u); put_be32(o + 12, E->rate); o[12] = (uint8_t)E->channels;
    put_be32(o + 16, E->frame_count); put_be16(o + 20, E->delay); put_be16(o + 22, E->padding);
    put_be32(o + 24, 0x636F6D70u); put_be16(o + 28, E->frame_size); o[30] = 1; o[31] = 15; o[32] = 1;
    o[33] = (uint8_t)E->channel_config; o[34] = (uint8_t)E->total_bands; o[35] = (uint8_t)E->base_bands;
    o[36] = (uint8_t)E->stereo_bands; o[37] = (uint8_t)E->bands_per_hfr_group;
    pos = 40;
    if (E->loop_flag) {
        put_be32(o + 40, 0x6C6F6F70u); put_be32(o + 44, E->loop_start_frame); put_be32(o + 48, E->loop_end_frame);
        put_be16(o + 52, E->loop_start_delay); put_be16(o + 54, E->loop_end_padding);
        pos = 56;
    }
    put_be32(o + pos, 0x63697068u); put_be16(o + pos + 4, 0);
    put_be32(o + pos + 6, 0x70616400u);
    put_be16(o + E->header_size - 2, ora_crc16(o, E->header_size - 2));
    free(E); wav_release(&w);
    *out = o; *out_len = total;
    return 0;
}
