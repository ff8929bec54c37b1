// tools/asan_fuzz.cpp -- malformed-input driver for the AddressSanitizer build of the library (tools/asan_gpu.sh), plain C ABI, no Python:
// this ROCm's sanitizer hooks fail inside a python process (hsa_amd_memory_pool_allocate "out of memory" at runtime start-up), and
// parity is not the question here -- only whether a kernel or the host planner ever touches memory it does not own while it eats
// untrusted bytes.  Inputs: the golden files of tests/golden (valid HCA / ADX / WAV), mutated by a seeded generator:
//   header edits (bit flips / random bytes in the first 160 bytes, half of the HCA ones with the header checksum renewed), truncations,
//   HCA frames overwritten with random bytes at several densities WITH a valid frame checksum (so that the unpack runs: escape codes,
//   out-of-range deltas, reads past the frame end -- hca.cpp:1149-1205), ADX blocks overwritten, end-of-stream markers planted on random
//   rows, sample counts rewritten (adx.cpp:380-415) -- through the five single-file calls and, every few rounds, as batches through the
//   job API (cri_job_create_*_items + cri_job_run_host_items: the segmented ADX kernels, the lane-per-frame parse, the encoders).
// Every buffer handed to the library is an exact-size malloc block (host red zones); device buffers are the library's own hipMalloc'ed
// arena (device red zones).  A report ends the process; the driver prints a count line per round so that the log shows how far it got.
//   usage: asan_fuzz ROUNDS SEED file...
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "../include/cricodecs_hip.h"

static uint64_t rng_state = 1;
static uint32_t rnd() { rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(rng_state >> 33); }
static uint32_t rnd_below(uint32_t n) { return n ? rnd() % n : 0; }
static uint16_t crc16(const uint8_t* p, size_t n) {
    uint32_t c = 0;
    for (size_t i = 0; i < n; i++) { c ^= (uint32_t)p[i] << 8; for (int k = 0; k < 8; k++) c = (c & 0x8000) ? ((c << 1) ^ 0x8005) & 0xFFFF : (c << 1) & 0xFFFF; }
    return (uint16_t)c;
}
struct File { std::string name; std::vector<uint8_t> data; int kind; };   // 0 hca, 1 adx, 2 wav
static std::vector<uint8_t> exact(const std::vector<uint8_t>& v) { return v; }

static void mutate(const File& f, std::vector<uint8_t>& b) {
    b = f.data;
    const uint32_t how = rnd_below(8);
    if (how == 0 && b.size() > 1) { b.resize(rnd_below((uint32_t)b.size())); return; }            // truncation
    if (how <= 3) {                                                                                 // header edits
        const uint32_t region = (uint32_t)std::min<size_t>(b.size(), 160);
        for (uint32_t k = 0, n = 1 + rnd_below(3); k < n && region; k++) {
            const uint32_t p = rnd_below(region);
            b[p] = (rnd() & 1) ? (uint8_t)rnd() : (uint8_t)(b[p] ^ (1u << rnd_below(8)));
        }
        if (f.kind == 0 && (rnd() & 1) && b.size() >= 8) {
            const uint32_t hs = ((uint32_t)f.data[6] << 8) | f.data[7];
            if (hs >= 2 && hs <= b.size()) { b[6] = f.data[6]; b[7] = f.data[7]; const uint16_t c = crc16(b.data(), hs - 2); b[hs - 2] = (uint8_t)(c >> 8); b[hs - 1] = (uint8_t)c; }
        }
        return;
    }
    if (f.kind == 0 && b.size() >= 0x20) {                                                          // frames of random bytes, checksums valid
        const uint32_t hs = ((uint32_t)b[6] << 8) | b[7], fs = ((uint32_t)b[0x1C] << 8) | b[0x1D];
        if (fs >= 8 && hs < b.size()) {
            const uint32_t dens = 1 + rnd_below(100);
            for (size_t at = hs; at + fs <= b.size(); at += fs) {
                if (rnd_below(100) >= dens && how != 7) continue;
                const uint32_t from = how == 4 ? 2 : 2 + rnd_below(fs - 4);
                for (uint32_t i = from; i < fs - 2; i++) if (how == 7 || rnd_below(100) < dens) b[at + i] = (uint8_t)rnd();
                b[at] = 0xFF; b[at + 1] = 0xFF;
                const uint16_t c = crc16(b.data() + at, fs - 2); b[at + fs - 2] = (uint8_t)(c >> 8); b[at + fs - 1] = (uint8_t)c;
            }
        }
        return;
    }
    if (f.kind == 1 && b.size() >= 0x14) {                                                          // ADX: blocks, end markers, counts
        const uint32_t off = (((uint32_t)b[2] << 8) | b[3]) + 4, bs = b[5], ch = b[7];
        const uint32_t rowb = bs * (ch ? ch : 1);
        if (off < b.size() && rowb) {
            const uint32_t rows = (uint32_t)((b.size() - off) / rowb);
            for (uint32_t k = 0, n = rnd_below(6); k < n && rows; k++) { const size_t at = off + (size_t)rnd_below(rows) * rowb + (rnd() & 1 ? 0 : bs * rnd_below(ch ? ch : 1)); if (at + 2 <= b.size()) { b[at] = 0x80; b[at + 1] = 0x01; } }
            for (uint32_t k = 0, n = rnd_below(40); k < n; k++) b[off + rnd_below((uint32_t)(b.size() - off))] = (uint8_t)rnd();
            if (rnd() & 1) { const uint32_t cnt = rnd_below(3) == 0 ? rnd() : rnd_below(rows * 32 + 64); b[12] = (uint8_t)(cnt >> 24); b[13] = (uint8_t)(cnt >> 16); b[14] = (uint8_t)(cnt >> 8); b[15] = (uint8_t)cnt; }
        }
        return;
    }
    for (uint32_t k = 0, n = 1 + rnd_below(8); k < n && !b.empty(); k++) b[rnd_below((uint32_t)b.size())] = (uint8_t)rnd();   // WAV: anywhere
}

static void one_call(const File& f, const std::vector<uint8_t>& in, long* counts) {
    uint8_t* heap = (uint8_t*)malloc(in.size() ? in.size() : 1);                                    // exact size: host red zones right behind it
    memcpy(heap, in.data(), in.size());
    uint8_t* out = nullptr; size_t n = 0; int rc = 0;
    if (f.kind == 0) {
        const uint32_t hs = in.size() >= 8 ? ((uint32_t)heap[6] << 8) | heap[7] : 0;
        rc = cri_hca_decode(heap, in.size(), hs, (rnd() & 3) ? 0 : 0xCF222F1FE0748978ull, 0, &out, &n);
        if (out) cri_free(out);
        if ((rnd() & 7) == 0) (void)cri_hca_crypt(heap, in.size(), rnd() & 1, hs, 56, 0xCF222F1FE0748978ull, (uint16_t)rnd());
    } else if (f.kind == 1) {
        rc = cri_adx_decode(heap, in.size(), &out, &n);
        if (out) cri_free(out);
    } else {
        if (rnd() & 1) rc = cri_adx_encode(heap, in.size(), 4, 18, 2 + rnd_below(3), rnd_below(3) ? 500 : rnd_below(65536), 0, 3 + rnd_below(3), rnd() & 1, &out, &n);
        else rc = cri_hca_encode(heap, in.size(), rnd() & 1, rnd_below(6), &out, &n);
        if (out) cri_free(out);
    }
    counts[rc == 0 ? 0 : 1]++;
    free(heap);
}

static void batch(const std::vector<File>& files, int kind, long* counts) {
    std::vector<std::vector<uint8_t>> items;
    for (uint32_t k = 0, n = 8 + rnd_below(40); k < n; k++) {
        const File* f;
        do f = &files[rnd_below((uint32_t)files.size())]; while (f->kind != kind);
        std::vector<uint8_t> b;
        if (rnd_below(4) == 0) b = f->data; else mutate(*f, b);
        items.push_back(b);
    }
    std::vector<uint8_t*> heaps; std::vector<const uint8_t*> ptrs; std::vector<uint64_t> lens, keys; std::vector<uint16_t> sub;
    for (auto& b : items) { uint8_t* h = (uint8_t*)malloc(b.size() ? b.size() : 1); memcpy(h, b.data(), b.size()); heaps.push_back(h); ptrs.push_back(h); lens.push_back(b.size()); keys.push_back((rnd() & 1) ? 0xCF222F1FE0748978ull : 0); sub.push_back(0); }
    cri_items it{ptrs.data(), lens.data(), nullptr, (uint32_t)items.size()};
    cri_job* job = nullptr;
    int rc;
    if (kind == 0) rc = (rnd() & 3) ? cri_job_create_hca_decode_items(&it, keys.data(), sub.data(), &job) : cri_job_create_hca_crypt_items(&it, rnd() & 1, 56, keys.data(), sub.data(), &job);
    else if (kind == 1) rc = cri_job_create_adx_decode_items(&it, &job);
    else if (rnd() & 1) { cri_adx_encode_params p; memset(&p, 0, sizeof p); p.bitdepth = 4; p.blocksize = 18; p.encoding_mode = 2 + rnd_below(3); p.highpass_frequency = 500; p.adx_version = 4; rc = cri_job_create_adx_encode_items(&it, &p, &job); }
    else rc = cri_job_create_hca_encode_items(&it, 0, rnd_below(5), &job);
    if (rc == 0 && job) {
        const uint64_t ob = cri_job_output_bytes(job);
        uint8_t* out = (uint8_t*)malloc(ob ? ob : 1);
        std::vector<int32_t> st(items.size() + 1);
        rc = cri_job_run_host_items(job, &it, out, st.data());
        counts[rc == 0 ? 2 : 3]++;
        free(out);
        cri_job_destroy(job);
    } else counts[3]++;
    for (uint8_t* h : heaps) free(h);
}

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s ROUNDS SEED file...\n", argv[0]); return 2; }
    const long rounds = atol(argv[1]);
    rng_state = (uint64_t)atoll(argv[2]) * 2654435761u + 12345;
    std::vector<File> files;
    for (int i = 3; i < argc; i++) {
        FILE* fp = fopen(argv[i], "rb");
        if (!fp) continue;
        File f; f.name = argv[i];
        uint8_t buf[65536]; size_t n;
        while ((n = fread(buf, 1, sizeof buf, fp)) > 0) f.data.insert(f.data.end(), buf, buf + n);
        fclose(fp);
        if (f.data.size() < 16) continue;
        f.kind = (f.data[0] & 0x7F) == 'H' && (f.data[1] & 0x7F) == 'C' ? 0 : (f.data[0] == 0x80 && f.data[1] == 0x00 ? 1 : (memcmp(f.data.data(), "RIFF", 4) == 0 ? 2 : -1));
        if (f.kind >= 0) files.push_back(f);
    }
    if (!cri_device_available()) { fprintf(stderr, "no device\n"); return 3; }
    // longer material, made here (the golden files are a second at most: too short for the segmented ADX kernels and for transform runs):
    // 6 s of a tone over noise, mono and stereo, as WAV, as ADX (the library's own encoder) and as HCA of three qualities, one enciphered
    for (uint32_t ch = 1; ch <= 2; ch++) {
        const uint32_t n = 48000 * 6;
        File w; w.kind = 2; w.name = "generated.wav";
        w.data.resize(44 + (size_t)n * ch * 2);
        uint8_t* h = w.data.data();
        auto put32 = [&](size_t at, uint32_t v) { h[at] = (uint8_t)v; h[at + 1] = (uint8_t)(v >> 8); h[at + 2] = (uint8_t)(v >> 16); h[at + 3] = (uint8_t)(v >> 24); };
        memcpy(h, "RIFF", 4); put32(4, (uint32_t)w.data.size() - 8); memcpy(h + 8, "WAVEfmt ", 8); put32(16, 16); h[20] = 1; h[21] = 0; h[22] = (uint8_t)ch; h[23] = 0;
        put32(24, 48000); put32(28, 48000 * ch * 2); h[32] = (uint8_t)(ch * 2); h[33] = 0; h[34] = 16; h[35] = 0; memcpy(h + 36, "data", 4); put32(40, n * ch * 2);
        int32_t ph = 0;
        for (uint32_t i = 0; i < n; i++) {
            ph = (ph + 1800 + (int32_t)(i >> 9)) & 0xFFFF;
            const int32_t tri = (ph < 0x8000 ? ph : 0xFFFF - ph) - 0x4000;                           // a sweeping triangle
            const int32_t amp = i < 512 ? (int32_t)i : 512;
            for (uint32_t c = 0; c < ch; c++) {
                int32_t v = (tri * amp / 1024) + (int32_t)(rnd() % 600) - 300;
                if (i > 100000 && i < 130000) v = 0;                                                 // a stretch of digital silence
                h[44 + ((size_t)i * ch + c) * 2] = (uint8_t)v; h[45 + ((size_t)i * ch + c) * 2] = (uint8_t)(v >> 8);
            }
        }
        files.push_back(w);
        uint8_t* out = nullptr; size_t on = 0;
        if (cri_adx_encode(w.data.data(), w.data.size(), 4, 18, 3, 500, 0, 4, 0, &out, &on) == 0) { File a; a.kind = 1; a.name = "generated.adx"; a.data.assign(out, out + on); files.push_back(a); cri_free(out); }
        for (uint32_t q = 1; q <= 3; q++)
            if (cri_hca_encode(w.data.data(), w.data.size(), 0, q, &out, &on) == 0) {
                File a; a.kind = 0; a.name = "generated.hca"; a.data.assign(out, out + on); cri_free(out);
                if (q == 1) (void)cri_hca_crypt(a.data.data(), a.data.size(), 1, ((uint32_t)a.data[6] << 8) | a.data[7], 56, 0xCF222F1FE0748978ull, 0);
                files.push_back(a);
            }
    }
    int have[3] = {0, 0, 0};
    for (auto& f : files) have[f.kind]++;
    printf("build %s; %zu files (%d hca, %d adx, %d wav); %ld rounds\n", cri_build_id(), files.size(), have[0], have[1], have[2], rounds);
    long counts[4] = {0, 0, 0, 0};
    for (long r = 0; r < rounds; r++) {
        for (int k = 0; k < 64; k++) {
            const File& f = files[rnd_below((uint32_t)files.size())];
            std::vector<uint8_t> b;
            mutate(f, b);
            one_call(f, b, counts);
        }
        for (int kind = 0; kind < 3; kind++) if (have[kind]) batch(files, kind, counts);
        printf("round %ld: single calls accepted %ld rejected %ld; batches run %ld refused %ld\n", r, counts[0], counts[1], counts[2], counts[3]);
        fflush(stdout);
    }
    printf("done: no report\n");
    return 0;
}
