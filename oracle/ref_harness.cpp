// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
//
// Builds the *unmodified* reference codec sources (where they lie under /root/reference) into a small
// command-line tool, oracle/_ref/criref, used to (a) pin oracle/cri_oracle.c against the real reference,
// (b) generate the golden vectors under tests/golden/, (c) dump the reference's constant tables so that
// tools/gen_tables.py can verify the tables it regenerates, and (d) time the reference's single-thread
// CPU path (bench.py cpu_baseline kind="reference").
//
// It contains no reference code: the reference translation unit is pulled in by absolute path below.
// Determinism recipe: zero-initialised heap (calloc-backed operator new[]),
// zeroed ADX / clHCA structs.  Entry points exercised:
//   ADX::GetADX / ADX::GetWAVE            /root/reference/CriCodecs/adx.cpp:507-514
//   initHCAEncode / Encode / PackHeader   /root/reference/CriCodecs/hca.cpp:2414, 3072, 3109
//   clHCA_DecodeHeader / clHCA_DecodeBlock / clHCA_ReadSamples16   hca.cpp:628, 1238, 339
//   cipher_init / cipher_decrypt / CryptHeader                     hca.cpp:599, 491, 3166
// The wrapper loops around them mirror AdxEncode/AdxDecode (adx.cpp:517-558), HcaDecode (hca.cpp:3340-3457),
// HcaEncode (3459-3489) and HcaCrypt (3271-3337) minus the CPython argument plumbing.
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <new>
#include <chrono>
#include <string>
#include <vector>

void* operator new[](std::size_t n) { void* p = calloc(1, n ? n : 1); if (!p) abort(); return p; }
void operator delete[](void* p) noexcept { free(p); }
void operator delete[](void* p, std::size_t) noexcept { free(p); }

#include "/root/reference/CriCodecs/CriCodecs.cpp"

typedef std::vector<unsigned char> bytes_t;

static bytes_t slurp(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    bytes_t b((size_t)n + 64, 0);           // 64 bytes of zero slack: the reference reads past the end
    if (n && fread(b.data(), 1, (size_t)n, f) != (size_t)n) { exit(2); }
    fclose(f);
    b.resize((size_t)n);                     // capacity keeps the slack
    return b;
}
static void spit(const char* path, const unsigned char* p, size_t n) {
    FILE* f = fopen(path, "wb");
    if (!f) { fprintf(stderr, "cannot write %s\n", path); exit(2); }
    if (n) fwrite(p, 1, n, f);
    fclose(f);
}

// ---- ADX ------------------------------------------------------------------------------------------
static int do_adx_encode(bytes_t& wav, unsigned bd, unsigned bs, unsigned mode, unsigned hp, unsigned filter,
                         unsigned ver, int force, bytes_t& out) {
    AdxErrorCode = 0;
    ADX* a = new ADX();
    memset(&a->Header, 0, sizeof a->Header);
    a->Looping = 0;
    PCM w;
    char res = w.LoadDirect(wav.data());
    if (res < 0) return 100 + (-res);        // PCM error domain
    unsigned char* o = a->GetADX(w, bd, bs, mode, (unsigned short)hp, filter, ver, force != 0);
    if (AdxErrorCode) return -AdxErrorCode;
    out.assign(o, o + a->size);
    delete[] o;
    delete a;
    return 0;
}
static int do_adx_decode(bytes_t& adx, bytes_t& out) {
    AdxErrorCode = 0;
    ADX* b = new ADX();
    b->Looping = 0;
    memset(&b->Header, 0, sizeof b->Header);
    PCM w2;
    b->GetWAVE(adx.data(), w2);
    if (AdxErrorCode) return -AdxErrorCode;
    out.assign(w2.WAVEBuffer, w2.WAVEBuffer + w2.wav.size + 8);
    delete b;
    return 0;
}

// ---- HCA ------------------------------------------------------------------------------------------
static int do_hca_encode(bytes_t& wav, unsigned quality, unsigned force_noloop, bytes_t& out) {
    HcaErrorCode = 0;
    PCM w;
    clHCA* hca = (clHCA*)calloc(1, sizeof(clHCA));
    char res = w.LoadDirect(wav.data());
    if (res < 0) return 100 + (-res);
    hca->loop_flag = w.wav.chunks.Looping && !force_noloop;
    res = initHCAEncode(w, *hca, (CriHcaQuality)quality);
    if (res < 0) return 3;
    size_t total = hca->header_size + (size_t)hca->frame_count * hca->frame_size;
    out.assign(total, 0);
    Encode(*hca, w, out.data() + hca->header_size);
    if (HcaErrorCode < 0) return 4;
    PackHeader(*hca, out.data());
    free(hca);
    return 0;
}

static unsigned long long mix_key(unsigned long long keycode, unsigned short subkey) {
    if (subkey) keycode = keycode * (((uint64_t)subkey << 16u) | ((uint16_t)~subkey + 2u));
    return keycode;
}

// mirrors HcaDecode (hca.cpp:3340-3457); float_out != NULL additionally receives the pre-clamp wave floats
static int do_hca_decode(bytes_t& in, unsigned long long keycode, unsigned short subkey, bytes_t& out,
                         std::vector<float>* float_out) {
    unsigned char* data = in.data();
    int hs = clHCA_isOurFile(data, (unsigned)in.size());
    if (hs < 0) return 1;
    unsigned header_size = (unsigned)hs;
    clHCA* hca = (clHCA*)calloc(1, sizeof(clHCA));
    if (clHCA_DecodeHeader(hca, data, header_size) < 0) return 1;
    PCM wav;
    wav.wav.chunks.WAVEfmt.Channels = hca->channels;
    wav.wav.chunks.WAVEfmt.SampleRate = hca->sample_rate;
    wav.wav.chunks.WAVEfmt.BlockAlign = hca->channels * 2;
    if (hca->loop_flag) {
        wav.wav.chunks.Looping = 1;
        wav.wav.chunks.WAVEsmpl.Loops = new smplloop[1];
        wav.wav.chunks.WAVEsmpl.Loops[0].Start = hca->loop_start_frame * HCA_SAMPLES_PER_FRAME + hca->loop_start_delay - hca->encoder_delay;
        wav.wav.chunks.WAVEsmpl.Loops[0].End = hca->loop_end_frame * HCA_SAMPLES_PER_FRAME + (HCA_SAMPLES_PER_FRAME - hca->loop_end_padding) - hca->encoder_delay;
    }
    unsigned datasize = (hca->frame_count * HCA_SAMPLES_PER_FRAME - hca->encoder_delay - hca->encoder_padding) * hca->channels * sizeof(short);
    wav.wav.chunks.WAVEdata.size = datasize;
    data += header_size;
    clHCA_SetKey(hca, mix_key(keycode, subkey));
    std::vector<unsigned char> buf(hca->frame_size, 0);
    wav.GetWaveBuffer(datasize / (hca->channels * 2), hca->channels, hca->sample_rate, wav.wav.chunks.Looping);
    signed short* outbuf = wav.PCM_16;
    const unsigned samples_to_do = (datasize >> 1) / hca->channels;
    unsigned samples_filled = 0, samples_to_discard = hca->encoder_delay, samples_consumed = 0, current_block = 0;
    int samples_done = 0;
    std::vector<short> sample_buffer(hca->channels * HCA_SAMPLES_PER_FRAME, 0);
    while (samples_done < (int)samples_to_do) {
        if (samples_filled) {
            int samples_to_get = samples_filled;
            if (samples_to_discard) {
                if (samples_to_get > (int)samples_to_discard) samples_to_get = samples_to_discard;
                samples_to_discard -= samples_to_get;
            } else {
                if (samples_to_get > (int)(samples_to_do - samples_done)) samples_to_get = samples_to_do - samples_done;
                memcpy(outbuf + samples_done * hca->channels, sample_buffer.data() + samples_consumed * hca->channels,
                       samples_to_get * hca->channels * sizeof(short));
                samples_done += samples_to_get;
            }
            samples_consumed += samples_to_get;
            samples_filled -= samples_to_get;
        } else {
            if (current_block >= hca->frame_count) break;
            if ((size_t)(data - in.data()) + hca->frame_size > in.size()) return 2;
            memcpy(buf.data(), data, hca->frame_size);
            data += hca->frame_size;
            current_block++;
            int res = clHCA_DecodeBlock(hca, buf.data(), hca->frame_size);
            if (res < 0) return 2;
            clHCA_ReadSamples16(hca, sample_buffer.data());
            if (float_out) {
                for (unsigned sf = 0; sf < HCA_SUBFRAMES; sf++)
                    for (unsigned s = 0; s < HCA_SAMPLES_PER_SUBFRAME; s++)
                        for (unsigned c = 0; c < hca->channels; c++)
                            float_out->push_back(hca->channel[c].wave[sf][s]);
            }
            samples_consumed = 0;
            samples_filled += HCA_SAMPLES_PER_FRAME;
        }
    }
    out.assign(wav.WAVEBuffer, wav.WAVEBuffer + wav.wav.size + 8);   // GetWaveBuffer overwrote riff.size (pcm.cpp:549-552)
    free(hca);
    return 0;
}

// mirrors HcaCrypt (hca.cpp:3271-3337), on a private copy
static int do_hca_crypt(bytes_t& io, unsigned crypt, unsigned type, unsigned long long keycode, unsigned short subkey) {
    unsigned char* buffer = io.data();
    int hs = clHCA_isOurFile(buffer, (unsigned)io.size());
    if (hs < 0) return 1;
    unsigned header_size = (unsigned)hs;
    clHCA* hca = (clHCA*)calloc(1, sizeof(clHCA));
    if (clHCA_DecodeHeader(hca, buffer, header_size) < 0) return 1;
    hca->ciph_type = crypt == 1 ? type : hca->ciph_type;
    keycode = mix_key(keycode, subkey);
    cipher_init(hca->cipher_table, hca->ciph_type, keycode);
    if (crypt) {
        unsigned char o[256];
        for (int i = 0; i < 256; i++) o[hca->cipher_table[i]] = (unsigned char)i;
        memcpy(hca->cipher_table, o, 256);
    }
    unsigned char* frames = buffer + header_size;
    for (unsigned i = 0; i < hca->frame_count; i++, frames += hca->frame_size) {
        cipher_decrypt(hca->cipher_table, frames, hca->frame_size);
        WriteShortBE(frames + hca->frame_size - 2, crc16_checksum(frames, hca->frame_size - 2));
    }
    type = crypt == 1 ? type : 0;
    CryptHeader(hca, buffer, header_size, type);
    free(hca);
    return 0;
}

// ---- table dump (verification input of tools/gen_tables.py) --------------------------------------------
static void dump_u8(FILE* f, const char* name, const unsigned char* p, size_t n) {
    fprintf(f, "%s u8 %zu", name, n); for (size_t i = 0; i < n; i++) fprintf(f, " %u", p[i]); fprintf(f, "\n");
}
static void dump_u32(FILE* f, const char* name, const unsigned int* p, size_t n) {
    fprintf(f, "%s u32 %zu", name, n); for (size_t i = 0; i < n; i++) fprintf(f, " %u", p[i]); fprintf(f, "\n");
}
static void dump_i32(FILE* f, const char* name, const int* p, size_t n) {
    fprintf(f, "%s i32 %zu", name, n); for (size_t i = 0; i < n; i++) fprintf(f, " %d", p[i]); fprintf(f, "\n");
}
static void dump_tables(const char* path) {
    FILE* f = fopen(path, "w");
    unsigned int tmp[1024];
    for (int i = 0; i < 256; i++) tmp[i] = hcacommon_crc_mask_table[i];
    dump_u32(f, "crc16", tmp, 256);
    dump_u8(f, "ath_base_curve", ath_base_curve, 656);
    dump_u8(f, "invert_table", hcadecoder_invert_table, 66);
    dump_u32(f, "dequant_scaling", hcadequantizer_scaling_table_float_hex, 64);
    dump_u32(f, "dequant_range", hcadequantizer_range_table_float_hex, 16);
    dump_u8(f, "max_bit", hcatbdecoder_max_bit_table, 16);
    dump_u8(f, "read_bit", hcatbdecoder_read_bit_table, 128);
    for (int i = 0; i < 128; i++) { float v = hcatbdecoder_read_val_table[i]; memcpy(&tmp[i], &v, 4); }
    dump_u32(f, "read_val", tmp, 128);
    dump_u32(f, "scale_conversion", hcadecoder_scale_conversion_table_hex, 128);
    dump_u32(f, "intensity_ratio", hcadecoder_intensity_ratio_table_hex, 16);
    dump_u32(f, "dec_sin", &sin_tables_hex[0][0], 7 * 64);
    dump_u32(f, "dec_cos", &cos_tables_hex[0][0], 7 * 64);
    dump_u32(f, "imdct_window", hcaimdct_window_float_hex, 128);
    dump_u8(f, "default_channel_mapping", DefaultChannelMapping, 9);
    dump_u8(f, "valid_channel_mappings", &ValidChannelMappings[0][0], 64);
    dump_u8(f, "enc_max_bits", QuantizedSpectrumMaxBits, 16);
    for (int i = 0; i < 16; i++) { float v = QuantizerInverseStepSize[i]; memcpy(&tmp[i], &v, 4); }
    dump_u32(f, "enc_inv_step", tmp, 16);
    dump_i32(f, "enc_scale_to_res", ScaleToResolutionCurve, 59);
    dump_i32(f, "enc_spectrum_bits", &QuantizeSpectrumBits[0][0], 128);
    { int t2[128]; for (int i = 0; i < 128; i++) t2[i] = (&QuantizeSpectrumValue[0][0])[i]; dump_i32(f, "enc_spectrum_value", t2, 128); }
    dump_u32(f, "enc_intensity_bounds", IntensityRatioBoundsTableHex, 14);
    dump_u32(f, "enc_dead_zone", QuantizerDeadZoneHex, 16);
    dump_u8(f, "enc_shuffle", ShuffleTable, 128);
    dump_u32(f, "enc_quant_scaling", QuantizerScalingTableHex, 64);
    dump_u32(f, "enc_sin", &SinTablesHex[0][0], 8 * 128);
    dump_u32(f, "enc_cos", &CosTablesHex[0][0], 8 * 128);
    { int t2[8]; for (int i = 0; i < 8; i++) t2[i] = StaticCoefficients[i]; dump_i32(f, "adx_static_coefs", t2, 8); }
    fclose(f);
}

static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: criref <cmd> ...\n"); return 2; }
    std::string cmd = argv[1];
    bytes_t in, out;
    int rc = 0;
    if (cmd == "dump-tables" && argc == 3) { dump_tables(argv[2]); return 0; }
    if (cmd == "adxenc" && argc == 11) {
        in = slurp(argv[2]);
        rc = do_adx_encode(in, atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), atoi(argv[7]), atoi(argv[8]), atoi(argv[9]), atoi(argv[10]), out);
        if (!rc) spit(argv[3], out.data(), out.size());
    } else if (cmd == "adxdec" && argc == 4) {
        in = slurp(argv[2]);
        rc = do_adx_decode(in, out);
        if (!rc) spit(argv[3], out.data(), out.size());
    } else if (cmd == "hcaenc" && argc == 6) {
        in = slurp(argv[2]);
        rc = do_hca_encode(in, atoi(argv[4]), atoi(argv[5]), out);
        if (!rc) spit(argv[3], out.data(), out.size());
    } else if ((cmd == "hcadec" || cmd == "hcadecf") && argc == 6) {
        in = slurp(argv[2]);
        std::vector<float> fl;
        rc = do_hca_decode(in, strtoull(argv[4], 0, 0), (unsigned short)strtoul(argv[5], 0, 0), out, cmd == "hcadecf" ? &fl : 0);
        if (!rc) {
            if (cmd == "hcadecf") spit(argv[3], (const unsigned char*)fl.data(), fl.size() * 4);  // raw pre-clamp floats, all frames
            else spit(argv[3], out.data(), out.size());
        }
    } else if (cmd == "hcacrypt" && argc == 8) {
        in = slurp(argv[2]);
        rc = do_hca_crypt(in, atoi(argv[4]), atoi(argv[5]), strtoull(argv[6], 0, 0), (unsigned short)strtoul(argv[7], 0, 0));
        if (!rc) spit(argv[3], in.data(), in.size());
    } else if (cmd == "bench" && argc >= 5) {
        // criref bench <hcadec|hcaenc|adxdec|adxenc> <file> <min_seconds> [key]  -> prints "units seconds"
        std::string what = argv[2];
        in = slurp(argv[3]);
        double min_s = atof(argv[4]);
        unsigned long long key = argc > 5 ? strtoull(argv[5], 0, 0) : 0;
        double t0 = now_s(), t1 = t0; long reps = 0;
        do {
            bytes_t tmp = in; tmp.reserve(in.size() + 64);
            if (what == "hcadec") rc = do_hca_decode(tmp, key, 0, out, 0);
            else if (what == "hcaenc") rc = do_hca_encode(tmp, 1, 1, out);
            else if (what == "adxdec") rc = do_adx_decode(tmp, out);
            else if (what == "adxenc") rc = do_adx_encode(tmp, 4, 18, 3, 500, 0, 4, 0, out);
            else return 2;
            if (rc) break;
            reps++; t1 = now_s();
        } while (t1 - t0 < min_s);
        printf("%ld %.6f\n", reps, t1 - t0);
    } else {
        fprintf(stderr, "bad command line\n");
        return 2;
    }
    if (rc) fprintf(stderr, "criref: error %d\n", rc);
    return rc ? 10 + (rc & 0x7f) : 0;
}
