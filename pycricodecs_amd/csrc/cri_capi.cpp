// cri_capi.cpp -- job planner + the extern "C" boundary declared in include/cricodecs_hip.h.
//
// A job is planned on the host from the items' headers (cri_host.cpp), its small metadata (stream descriptors,
// cipher / ATH tables, header images) is uploaded once, and cri_job_run only enqueues kernels on the caller's
// stream.  There is no CPU implementation of the per-frame / per-block work in this library.
#include <hip/hip_runtime.h>
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>
#include "../../include/cricodecs_hip.h"
#include "cri_host.h"
#include "cri_kernels.h"

using namespace cri;

namespace {

// Job metadata (stream tables, cipher / ATH tables, header images ...) is staged on the host while a job is planned and goes
// to the device as ONE allocation and ONE copy (MetaPool::commit): a single-file call used to pay eight hipMalloc + hipMemcpy
// pairs and as many hipFree (each a device-wide wait).  Small allocations are recycled through a per-device cache.
struct DevBuf;
struct MetaPool {
    std::vector<uint8_t> host; void* dev = nullptr; size_t cap = 0; int device = -1;
    std::vector<DevBuf*> bufs;
    int commit();
    ~MetaPool();
};
struct DevBuf {
    void* p = nullptr; size_t n = 0, off = 0; MetaPool* pool = nullptr;
    int upload(const void* src, size_t bytes) {
        n = bytes;
        if (!bytes) return 0;
        off = (pool->host.size() + 255) & ~(size_t)255;
        pool->host.resize(off + bytes);
        memcpy(pool->host.data() + off, src, bytes);
        pool->bufs.push_back(this);
        return 0;
    }
    template <class T> int upload(const std::vector<T>& v) { return upload(v.data(), v.size() * sizeof(T)); }
};
struct MetaCache {                                            // recycled metadata allocations of one device (each at most 4 MB)
    std::mutex mu; std::vector<std::pair<void*, size_t>> free_list;
};
static std::mutex g_meta_mu;
static std::map<int, MetaCache*> g_meta_caches;
static MetaCache* meta_cache_of(int device) {
    std::lock_guard<std::mutex> lk(g_meta_mu);
    auto it = g_meta_caches.find(device);
    if (it == g_meta_caches.end()) it = g_meta_caches.emplace(device, new MetaCache()).first;
    return it->second;
}
int MetaPool::commit() {
    if (host.empty()) return 0;
    const size_t need = host.size();
    if (need <= (4u << 20)) {
        MetaCache* c = meta_cache_of(device);
        std::lock_guard<std::mutex> lk(c->mu);
        for (size_t k = 0; k < c->free_list.size(); k++)
            if (c->free_list[k].second >= need) { dev = c->free_list[k].first; cap = c->free_list[k].second; c->free_list.erase(c->free_list.begin() + k); break; }
    }
    if (!dev) {
        cap = need <= (4u << 20) ? ((need + 65535) & ~(size_t)65535) : need;
        if (hipMalloc(&dev, cap) != hipSuccess) { dev = nullptr; return CRI_ERR_HIP; }
    }
    if (hipMemcpy(dev, host.data(), need, hipMemcpyHostToDevice) != hipSuccess) return CRI_ERR_HIP;
    for (DevBuf* b : bufs) b->p = (uint8_t*)dev + b->off;
    std::vector<uint8_t>().swap(host);
    return 0;
}
MetaPool::~MetaPool() {
    if (!dev) return;
    if (cap <= (4u << 20)) {
        MetaCache* c = meta_cache_of(device);
        std::lock_guard<std::mutex> lk(c->mu);
        if (c->free_list.size() < 8) { c->free_list.emplace_back(dev, cap); return; }
    }
    (void)hipFree(dev);
}

struct Image { uint64_t dst; std::vector<uint8_t> bytes; };

inline uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

// Tuning knobs.  The shipped library reads three documented environment variables ONCE ("Knobs"):
//   CRICODECS_ADX_MAPPING = chain | file | seg | lane | wave   which ADX kernels take a job (default: chosen by job shape)
//   CRICODECS_HOST_SLICE_MIN = bytes                           host-memory jobs moving at least this much are pipelined
//   CRICODECS_HOST_STAGE_PIECE = bytes                         largest piece of the page-locked staging ring used per copy
// Everything else here exists for the parity tests, which must be able to push work onto the kernels that normally take little of
// it (repair passes, general transforms): those fields can only be changed through cri_test_set, and that entry point exists only
// in the -DCRI_TESTING build of this file (pycricodecs_amd/lib/libcricodecs_hip_testing.so; never in libcricodecs_hip.so).
enum { ADX_MAP_AUTO = 0, ADX_MAP_CHAIN, ADX_MAP_FILE, ADX_MAP_SEG, ADX_MAP_LANE, ADX_MAP_WAVE };
struct Knobs {
    int adx_mapping = ADX_MAP_AUTO;
    uint64_t host_slice_min = 64ull << 20;       // jobs moving at least this much (in + out) are pipelined
    uint64_t host_stage_piece = 0;               // 0: the staging slot size
    // test-only
    int no_inlane = 0;                           // joint / wide / noise formats on the general transform kernels instead of the in-lane ones
    uint64_t adx_warm_pct = 100;                 // segmented ADX chains: warm-up length in per cent of the planner's
    uint64_t host_pull_wgs = 0;                  // the pipelined host path's upload kernel: workgroups (0: the planner's choice)
    uint64_t hca_run = 0;                        // HCA decode: frames of a transform run (0: the planner's choice of 8 / 16 / 32)
    uint64_t adx_seglen = 0;                     // ... least segment length (decode: in warm-ups, default 3; lane encode: per cent of the warm-up, default 50)
    int bad_launch = 0;                          // every run also launches a kernel with an impossible configuration (what a run reports then is under test)
};
// SIMDs of the calling thread's device (CUs x 4): the ADX lane kernels run one row after the other in a lane and are bound by instruction
// issue per SIMD, so what a dispatch costs is (waves per SIMD, rounded UP) x (rows of its longest lane) -- 1.4 waves per SIMD cost what 2 do
static uint32_t device_simds() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 1024;
    return (uint32_t)cus * 4;
}
static int adx_mapping_of(const char* e) {
    if (!e || !*e) return ADX_MAP_AUTO;
    const char* names[] = {"auto", "chain", "file", "seg", "lane", "wave"};
    for (int k = 0; k < 6; k++) if (!strcmp(e, names[k])) return k;
    fprintf(stderr, "cricodecs_hip: CRICODECS_ADX_MAPPING=%s is not one of auto|chain|file|seg|lane|wave: ignored\n", e);
    return ADX_MAP_AUTO;
}
// a byte count from the environment: decimal digits only (an optional k / m / g suffix); anything else leaves the default in place
static void env_bytes(const char* name, uint64_t& field) {
    const char* e = getenv(name);
    if (!e) return;
    char* end = nullptr;
    errno = 0;
    const unsigned long long v = strtoull(e, &end, 10);
    uint64_t mul = 1;
    if (end && (*end == 'k' || *end == 'K')) { mul = 1ull << 10; end++; }
    else if (end && (*end == 'm' || *end == 'M')) { mul = 1ull << 20; end++; }
    else if (end && (*end == 'g' || *end == 'G')) { mul = 1ull << 30; end++; }
    if (end == e || *end != 0 || errno != 0 || e[0] == '-' || e[0] == '+' || v > (~0ull >> 1) / mul) {
        fprintf(stderr, "cricodecs_hip: %s=%s is not a byte count: ignored\n", name, e);
        return;
    }
    field = v * mul;
}
static Knobs& knobs_mut() {
    static Knobs k;
    static std::once_flag once;
    std::call_once(once, [] {
        k.adx_mapping = adx_mapping_of(getenv("CRICODECS_ADX_MAPPING"));
        env_bytes("CRICODECS_HOST_SLICE_MIN", k.host_slice_min);
        env_bytes("CRICODECS_HOST_STAGE_PIECE", k.host_stage_piece);
    });
    return k;
}
static const Knobs& knobs() { return knobs_mut(); }

}  // namespace

struct cri_job {
    uint32_t kind = 0, n = 0;
    int device = -1;                             // HIP device the job's metadata lives on (current device of the creating thread)
    bool items_form = false;                     // created from a cri_items list (not one host blob) ...
    bool items_packed = true;                    // ... whose device layout is the items back to back (no caller offsets)
    std::vector<HcaStream> hca_streams_host;     // HCA decode: the sorted stream table (the pipelined host path slices it)
    std::vector<uint64_t> in_offsets, out_offsets;
    std::vector<int32_t> host_status;
    uint64_t in_bytes = 0, out_bytes = 0, scratch_bytes = 0, units = 0, units2 = 0, alg_bytes = 0;
    std::string dominant;
    // device metadata (one allocation: MetaPool)
    MetaPool meta;
    DevBuf d_formats, d_streams, d_cipher, d_ath, d_img, d_img_off, d_img_dst, d_chain_stream, d_history, d_stale,
        d_frame_sizes, d_first_frame, d_adx_streams, d_adx_order, d_crc_off, d_convert, d_segs, d_crcmul, d_seg_chain, d_enctab, d_enc_hint;
    cri_job() {
        for (DevBuf* b : {&d_formats, &d_streams, &d_cipher, &d_ath, &d_img, &d_img_off, &d_img_dst, &d_chain_stream, &d_history, &d_stale,
                          &d_frame_sizes, &d_first_frame, &d_adx_streams, &d_adx_order, &d_crc_off, &d_convert, &d_segs, &d_crcmul, &d_seg_chain, &d_enctab, &d_enc_hint}) b->pool = &meta;
    }
    std::vector<ConvertItem> convert;            // WAV items whose samples are converted to PCM16 in scratch before encoding
    uint64_t convert_total = 0;
    // registers item data for conversion; returns the scratch offset its PCM16 will be at
    uint64_t add_convert(uint64_t src_offset, const WavInfo& w) {
        ConvertItem c; c.first = convert_total; c.src_offset = src_offset; c.dst_offset = scratch_bytes;
        c.sample_size = w.sample_size; c.bitdepth = w.bitdepth; c.mode = w.mode; c.pad = 0;
        convert.push_back(c); convert_total += w.column_size;
        scratch_bytes = (scratch_bytes + 2ull * w.column_size + 255) & ~255ull;
        return c.dst_offset;
    }
    uint32_t n_images = 0;
    std::vector<Image> images;
    // launch plans (device pointers for in/out/scratch/status are filled at run time)
    std::vector<HcaDecArgs> hca_dec;
    std::vector<uint64_t> hca_group_first_record; // per launch set: scratch offset of its first frame record
    AdxArgs adx{};
    uint32_t adx_streams = 0;                    // number of valid ADX streams
    bool adx_wave_per_file = false;              // few chains, standard layout: use the wave-per-file kernels
    bool adx_seg = false;                        // segmented chains (k_adx_seg_*): long files of the standard layout
    bool adx_lane = false;                       // encode: a lane per (file, channel, segment) (k_adx_lane_encode) instead of a wave per (file, segment)
    uint64_t adx_seg_flags_offset = 0;           // scratch offset of the per-chain flag words (the lanes' records are at 0)
    uint64_t adx_seg_state_offset = 0, adx_seg_ckpt_offset = 0;   // encode: records / checkpoints (after the converted PCM)
    CryptArgs crypt{};
    SegmentArgs seg{};                           // USM demux / SFA pack: segment copies (+ audio mask)
    std::vector<uint32_t> item_tags;
    std::vector<uint64_t> item_sizes;             // true byte length of every output item (jobs whose items carry no length of their own)
    std::vector<uint64_t> float_offsets;          // HCA decode: n + 1 offsets (in floats) of the items in the validation output
    struct EncLaunch { uint32_t format, stream_begin, stream_end, frames, channels; };
    std::vector<HcaEncArgs> hca_enc;
    std::vector<uint32_t> hca_enc_crc_off;       // per launch: offset into d_crcmul
    std::vector<uint32_t> hca_enc_hint_off;      // per launch: offset into d_enc_hint (the stream of every 16th frame)
    uint32_t n_cipher = 0;
    // optional per-kernel-class event timing
    bool events_on = false;
    std::vector<std::string> class_names;
    std::vector<std::vector<std::pair<hipEvent_t, hipEvent_t>>> class_events;   // [class][launch]
    std::vector<size_t> class_used;
    void begin_run() { class_used.assign(class_names.size(), 0); }
    hipEvent_t mark(size_t cls, bool start, hipStream_t s) {
        if (!events_on) return nullptr;
        auto& v = class_events[cls];
        if (start) {
            if (class_used[cls] == v.size()) { hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b); v.push_back({a, b}); }
            (void)hipEventRecord(v[class_used[cls]].first, s);
            return v[class_used[cls]].first;
        }
        (void)hipEventRecord(v[class_used[cls]].second, s);
        class_used[cls]++;
        return nullptr;
    }
    // Lifetime: the kernels of a run read the job's metadata, and a destroyed job's metadata allocation is handed to the next job
    // (MetaPool).  Every cri_job_run leaves an event behind its last kernel; the destructor waits for it before anything is
    // recycled.  A run that is being CAPTURED into a hipGraph records nothing (a captured event cannot be waited for): a job must
    // outlive the launches of a graph that holds it (include/cricodecs_hip.h).
    // The pipelined host path of jobs that cannot be cut inside (run_host_core): the same work planned again as a few jobs over
    // consecutive ranges of the items, made on the first host run that wants them and kept.
    bool partable = false, parts_tried = false;
    std::mutex parts_mu;                         // (two threads may make their first large host run on one job at the same time)
    std::vector<uint64_t> part_keys;             // HCA decode: the caller's keys / subkeys, for the parts
    std::vector<uint16_t> part_subkeys;
    std::vector<cri_job*> host_parts;
    std::vector<uint32_t> host_part_first;       // [parts + 1] first item of each part
    // One event per stream the job has been run on (two threads may run one job on their own streams; the host path runs its
    // parts on private streams): the destructor waits for all of them.  Past 16 streams finished entries are reused (never waited for).
    std::mutex run_mu;
    std::vector<std::pair<hipStream_t, hipEvent_t>> run_events;
    void note_run(hipStream_t s) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) != hipSuccess) { (void)hipGetLastError(); return; }   // (only the query's own error is dropped)
        if (cs != hipStreamCaptureStatusNone) return;
        std::lock_guard<std::mutex> lk(run_mu);
        hipEvent_t ev = nullptr;
        for (auto& e : run_events) if (e.first == s) { ev = e.second; break; }
        if (!ev) {
            // Past 16 streams an entry whose event has COMPLETED is reused (a query, never a wait: cri_job_run does not block, and a
            // blocking call here could also break another thread's global-mode capture); while all are pending the list grows.
            if (run_events.size() >= 16)
                for (size_t i = 0; i < run_events.size(); i++)
                    if (hipEventQuery(run_events[i].second) == hipSuccess) { ev = run_events[i].second; run_events.erase(run_events.begin() + (long)i); break; }
                    else (void)hipGetLastError();              // (hipErrorNotReady is the answer, not a fault)
            if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return; }
            run_events.emplace_back(s, ev);
        }
        (void)hipEventRecord(ev, s);
    }
    ~cri_job() {
        for (cri_job* part : host_parts) delete part;
        for (auto& e : run_events) { (void)hipEventSynchronize(e.second); (void)hipEventDestroy(e.second); }
        for (auto& v : class_events) for (auto& e : v) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    }

    int upload_images() {
        std::vector<uint8_t> blob; std::vector<uint64_t> off{0}, dst;
        for (auto& im : images) { blob.insert(blob.end(), im.bytes.begin(), im.bytes.end()); off.push_back(blob.size()); dst.push_back(im.dst); }
        n_images = (uint32_t)images.size();
        if (!n_images) return 0;
        if (blob.empty()) blob.push_back(0);
        int rc = d_img.upload(blob); if (rc) return rc;
        rc = d_img_off.upload(off); if (rc) return rc;
        return d_img_dst.upload(dst);
    }
    int upload_convert() { return convert.empty() ? 0 : d_convert.upload(convert); }
};

// Device selection.  HIP's current device is a per-thread setting; a job is bound to the device that was current in the
// thread that created it (its metadata lives there), and every later call on the job -- from any thread -- switches to that
// device for its duration (DeviceGuard).  One process per GPU never notices; a threaded host may drive several GPUs.
static std::once_flag g_dev_once;
static std::atomic<int> g_dev_count{0};
static int device_count() {
    std::call_once(g_dev_once, [] { int n = 0; g_dev_count.store(hipGetDeviceCount(&n) == hipSuccess && n > 0 ? n : 0); });
    return g_dev_count.load();
}
extern "C" int cri_device_available(void) { return device_count() > 0 ? 1 : 0; }
extern "C" int cri_device_count(void) { return device_count(); }
extern "C" int cri_set_device(int device) {
    if (device < 0 || device >= device_count()) return device_count() ? CRI_ERR_INVALID_ARG : CRI_ERR_HIP;
    return hipSetDevice(device) == hipSuccess ? 0 : CRI_ERR_HIP;
}
extern "C" int cri_get_device(void) {
    int d = -1;
    if (!device_count() || hipGetDevice(&d) != hipSuccess) return -1;
    return d;
}
namespace {
struct DeviceGuard {
    int prev = -1; bool switched = false, good = false;
    explicit DeviceGuard(int device) {
        if (device < 0 || hipGetDevice(&prev) != hipSuccess) return;        // a job without a device never runs
        if (prev == device) { good = true; return; }
        switched = hipSetDevice(device) == hipSuccess;
        good = switched;
    }
    bool ok() const { return good; }                                          // false: the calling thread is NOT on the job's device
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};
}  // namespace

extern "C" void cri_free(void* p) { free(p); }

// Identity of the sources this library was built from (pycricodecs_amd/build.py: source_id()).  The string literal carries a marker in
// front so that the build script can also find the id in the FILE of a prebuilt library.
#ifndef CRI_BUILD_ID_STRING
#define CRI_BUILD_ID_STRING "CRI_BUILD_ID=unstamped"
#endif
extern "C" const char* cri_build_id(void) {
    static const char id[] = CRI_BUILD_ID_STRING;
    return id + 13;
}
#ifdef CRI_TESTING
extern "C" int cri_is_testing_build(void) { return 1; }
#endif

extern "C" const char* cri_strerror(int code) {
    static const char* adx[] = {                      // adx.cpp:11-30
        "Invalid ADX file header.", "AHX file provided, unsopported.", "Encrypted ADX detected, unsupported.",
        "Invalid/Unknown encoding mode found.", "Unknown ADX version provided.", "Invalid Bitdepth found on the provided ADX.",
        "ADX does not contain any channels info.", "Invalid ADX header, loop information size is bigger than the header.",
        "Inavlid ADX header, Criware copyright string not found.", "Numbers of Channel cannot exceed 255 or go below 0.",
        "Bitdepth must be between 2 and 15 inclusive.", "Blocksize must be between 3 and 255 inclusive.",
        "EncodingMode must be either 2, 3, or 4.", "HighpassFrequency must be between 0 and 65535 inclusive.",
        "Filter is used with EncodingMode == 2 and must be between 0 and 4 inclusive.", "AdxVersion must be either 3, 4 or 5.",
        "Provided Bitdepth does not fit correctly with the provided BlockSize", "Given WAVE file is not valid for ADX encoding."};
    static const char* pcm[] = {                      // pcm.cpp:22-33
        "Invalid WAVE file header.", "Invalid WAVE file header. Format info is not present.",
        "Unsupported/Unknown WAVE compression mode.", "Invalid looping sample info data.",
        "Invalid looping sample info data, Number of loops/loop data is larger than the available size.",
        "Data tag is not present.", "Header is not valid.", "PCM Bitdepth does not match compression type.",
        "Filesize exceeds 2GB use python to load in with buffer.", "Filesize is too low to be viable for loading."};
    if (code == 0) return "OK";
    if (code <= -1 && code >= -18) return adx[-code - 1];
    if (code <= -101 && code >= -110) return pcm[-code - 101];
    switch (code) {                                   // hca.cpp:3255-3264
        case CRI_ERR_HCA_HEADER: return "Header decoding error, the header is not a valid HCA header.";
        case CRI_ERR_HCA_DECODE: return "Decoding error, either an incorrect key or an unknown exception.";
        case CRI_ERR_HCA_CHANNEL_CONFIG: return "Error setting up channel configuration.";
        case CRI_ERR_HCA_ENCODE: return "Unknown Encoding error.";
        case CRI_ERR_INVALID_ARG: return "Invalid argument.";
        case CRI_ERR_NOMEM: return "Out of memory.";
        case CRI_ERR_HIP: return "HIP runtime error or no gfx950 device available (this library has no CPU fallback).";
        case CRI_ERR_UNSUPPORTED: return "Input is valid for the reference but not yet supported by the device path.";
        case CRI_ERR_AWB_HEADER: return "Invalid AWB header.";                     // awb.py:38
        case CRI_ERR_AWB_INTSIZE: return "Unknown int size.";                      // awb.py:106
        case CRI_ERR_USM_HEADER: return "Unsupported file type";                   // usm.py:130
        case CRI_ERR_USM_CHUNK: return "Unsupported chunk type";                   // usm.py:189
        case CRI_ITEM_SKIPPED: return "Item is not handled by this job.";
    }
    if (code <= -211 && code >= -216) return "Decoding error, either an incorrect key or an unknown exception.";
    return "Unknown error.";
}

// ------------------------------------------------------------------------------------------------ accessors
extern "C" uint32_t cri_job_kind(const cri_job* j) { return j ? j->kind : 0; }
extern "C" uint32_t cri_job_items(const cri_job* j) { return j ? j->n : 0; }
extern "C" uint64_t cri_job_input_bytes(const cri_job* j) { return j ? j->in_bytes : 0; }
extern "C" uint64_t cri_job_output_bytes(const cri_job* j) { return j ? j->out_bytes : 0; }
extern "C" const uint64_t* cri_job_output_offsets(const cri_job* j) { return j ? j->out_offsets.data() : nullptr; }
extern "C" const uint64_t* cri_job_input_offsets(const cri_job* j) { return j ? j->in_offsets.data() : nullptr; }
extern "C" const int32_t* cri_job_host_status(const cri_job* j) { return j ? j->host_status.data() : nullptr; }
extern "C" uint64_t cri_job_scratch_bytes(const cri_job* j) { return j ? j->scratch_bytes : 0; }
extern "C" uint64_t cri_job_units(const cri_job* j) { return j ? j->units : 0; }
extern "C" uint64_t cri_job_units2(const cri_job* j) { return j ? j->units2 : 0; }
extern "C" uint64_t cri_job_algorithmic_bytes(const cri_job* j) { return j ? j->alg_bytes : 0; }
extern "C" const char* cri_job_dominant_kernel(const cri_job* j) { return j ? j->dominant.c_str() : ""; }
extern "C" void cri_job_destroy(cri_job* j) { if (!j) return; DeviceGuard g(j->device); delete j; }
extern "C" int cri_job_device(const cri_job* j) { return j ? j->device : -1; }

// Where a job's items are on the host while it is planned, and where they will be on the device when it runs.
//  blob form    one host blob + offsets[n+1]; the device input is a byte-identical copy of the blob
//  items form   n host pointers + lengths (several items may share one host buffer: a tiled batch costs no host copy),
//               device offsets[n+1] (offsets[n] = device input size) or NULL for "packed back to back"
struct ItemSrc {
    const uint8_t* blob = nullptr; const uint64_t* offsets = nullptr;
    const uint8_t* const* ptrs = nullptr; const uint64_t* lens = nullptr;
    std::vector<uint64_t> packed;                // items form without offsets: prefix sums of lens
    uint32_t n = 0;
    static ItemSrc from_blob(const uint8_t* b, const uint64_t* o, uint32_t n) { ItemSrc s; s.blob = b; s.offsets = o; s.n = n; return s; }
    static int from_items(const cri_items* it, ItemSrc& s) {
        if (!it || (it->n && (!it->ptrs || !it->lens))) return CRI_ERR_INVALID_ARG;
        s.ptrs = it->ptrs; s.lens = it->lens; s.n = it->n; s.offsets = it->offsets;
        if (!s.offsets) {
            s.packed.assign((size_t)it->n + 1, 0);
            for (uint32_t i = 0; i < it->n; i++) s.packed[i + 1] = s.packed[i] + it->lens[i];
            s.offsets = s.packed.data();
        } else {
            for (uint32_t i = 0; i < it->n; i++)           // items may not overlap or run past the next one's start
                if (it->offsets[i] > it->offsets[i + 1] || it->offsets[i + 1] - it->offsets[i] < it->lens[i]) return CRI_ERR_INVALID_ARG;
        }
        for (uint32_t i = 0; i < it->n; i++) if (!it->ptrs[i] && it->lens[i]) return CRI_ERR_INVALID_ARG;
        return 0;
    }
    // device offsets never decrease with the item index (the planner's upload ranges and per-item lengths rely on it), and an item
    // given by pointer + length ends before the next one starts
    bool ok() const {
        if (!offsets || !(blob || ptrs || n == 0)) return false;
        for (uint32_t i = 0; i < n; i++) {
            if (offsets[i + 1] < offsets[i]) return false;
            if (lens && lens[i] > offsets[i + 1] - offsets[i]) return false;
        }
        return true;
    }
    const uint8_t* ptr(uint32_t i) const { return ptrs ? ptrs[i] : blob + offsets[i]; }
    size_t len(uint32_t i) const { return (size_t)(lens ? lens[i] : offsets[i + 1] - offsets[i]); }
    uint64_t off(uint32_t i) const { return offsets[i]; }   // device offset of item i
};

// Where a decoded WAV goes in the device output: the first free byte moved up so that the SAMPLES behind the header start a 128-byte
// line -- the decoders store PCM in pieces of whole sample rows, and with the items themselves aligned every piece began 44 bytes
// into a line: each line of the output was written in two parts by two store instructions a row apart (k_adx_seg_decode: 1.27 ms
// per 1000 x 10 s with those stores, 0.71 with stores that hit the cache).  The gap before an item is zero like every byte no kernel writes.
static uint64_t wav_item_start(uint64_t free_from, uint32_t header_bytes) { return align_up(free_from + header_bytes, 128) - header_bytes; }
static cri_job* new_job(uint32_t kind, const uint64_t* offsets, uint32_t n) {
    int device = -1;
    if (hipGetDevice(&device) != hipSuccess || device < 0) return nullptr;    // (callers return CRI_ERR_HIP)
    cri_job* j = new cri_job();
    j->kind = kind; j->n = n;
    j->device = device; j->meta.device = device;
    j->in_offsets.assign(offsets, offsets + n + 1);
    j->in_bytes = offsets[n];
    j->host_status.assign(n, 0);
    j->out_offsets.assign(n + 1, 0);
    return j;
}
static cri_job* new_job(uint32_t kind, const ItemSrc& it) {
    cri_job* j = new_job(kind, it.offsets, it.n);
    if (j) { j->items_form = it.ptrs != nullptr; j->items_packed = it.ptrs == nullptr || !it.packed.empty(); }
    return j;
}

// ------------------------------------------------------------------------------------------------ HCA decode
static int create_hca_decode(const ItemSrc& it, const uint64_t* keys, const uint16_t* subkeys,
                             const uint32_t* header_sizes, cri_job** out, const uint8_t* take = nullptr, uint8_t take_kind = 0) {
    const uint32_t n = it.n;
    if (!it.ok() || !out) return CRI_ERR_INVALID_ARG;
    if (!cri_device_available()) return CRI_ERR_HIP;
    cri_job* j = new_job(CRI_JOB_HCA_DECODE, it);
    if (!j) return CRI_ERR_HIP;
    j->dominant = "k_hca_transform";
    std::vector<HcaFormat> formats; std::vector<HcaStream> streams;
    std::vector<uint8_t> cipher, ath(128, 0);
    std::map<std::vector<uint32_t>, uint32_t> fmt_index;
    std::map<std::pair<uint32_t, uint64_t>, uint32_t> cipher_index;
    std::map<std::vector<uint8_t>, uint32_t> ath_index;
    ath_index[std::vector<uint8_t>(128, 0)] = 0;
    uint64_t out_pos = 0, float_pos = 0;
    j->float_offsets.assign((size_t)n + 1, 0);
    for (uint32_t i = 0; i < n; i++) {
        j->out_offsets[i] = out_pos;
        j->float_offsets[i] = float_pos;
        if (take && take[i] != take_kind) { j->host_status[i] = CRI_ITEM_SKIPPED; continue; }
        const uint8_t* d = it.ptr(i);
        size_t len = it.len(i);
        HcaHeader h;
        uint32_t hs_arg = header_sizes ? header_sizes[i] : (len >= 8 ? be16(d + 6) : 0);
        int rc = hca_parse_header(d, len, hs_arg, h);
        if (rc) { j->host_status[i] = rc; continue; }
        const uint32_t total = h.frame_count * 1024u;           // 32-bit like the reference's sample count (hca.cpp:3369-3391): a huge frame count wraps
        if (total < h.delay + h.padding) { j->host_status[i] = CRI_ERR_HCA_HEADER; continue; }
        uint32_t spc = total - h.delay - h.padding;
        uint32_t frames = spc == 0 ? 0 : (uint32_t)std::min<uint64_t>(h.frame_count, ((uint64_t)h.delay + spc + 1023) / 1024);
        if ((uint64_t)hs_arg + (uint64_t)frames * h.frame_size > len) { j->host_status[i] = CRI_ERR_HCA_DECODE; continue; }
        // format
        std::vector<uint32_t> key = {h.channels, h.version, h.frame_size, h.min_res, h.max_res, h.total_bands, h.base_bands, h.stereo_bands,
                                     h.bands_per_hfr_group, h.hfr_group_count, h.track_count, h.channel_config};
        std::vector<uint8_t> athv(h.ath, h.ath + 128);
        auto ai = ath_index.find(athv);
        uint32_t aidx;
        if (ai == ath_index.end()) { aidx = (uint32_t)(ath.size() / 128); ath.insert(ath.end(), athv.begin(), athv.end()); ath_index[athv] = aidx; } else aidx = ai->second;
        key.push_back(aidx);
        auto fi = fmt_index.find(key);
        uint32_t fidx;
        if (fi == fmt_index.end()) {
            HcaFormat F; memset(&F, 0, sizeof F);
            F.channels = h.channels; F.version = h.version; F.frame_size = h.frame_size; F.min_res = h.min_res; F.max_res = h.max_res;
            F.total_bands = h.total_bands; F.base_bands = h.base_bands; F.stereo_bands = h.stereo_bands;
            F.bands_per_hfr_group = h.bands_per_hfr_group; F.hfr_group_count = h.hfr_group_count; F.ath_index = aidx;
            F.record_bytes = hca_record_bytes(h.channels);
            for (uint32_t c = 0; c < 16; c++) { F.type[c] = h.type[c]; F.coded[c] = (uint8_t)h.coded[c]; }
            fidx = (uint32_t)formats.size(); formats.push_back(F); fmt_index[key] = fidx;
        } else fidx = fi->second;
        // cipher
        uint64_t mixed = hca_mix_key(keys ? keys[i] : 0, subkeys ? subkeys[i] : 0);
        uint32_t ctype = h.ciph_type;
        if (ctype == 56 && !mixed) ctype = 0;
        if (ctype != 56) mixed = 0;
        auto ck = std::make_pair(ctype, mixed);
        auto ci = cipher_index.find(ck);
        uint32_t cidx;
        if (ci == cipher_index.end()) {
            uint8_t t[256]; hca_cipher_table(ctype, mixed, t);
            cidx = (uint32_t)(cipher.size() / 256); cipher.insert(cipher.end(), t, t + 256); cipher_index[ck] = cidx;
        } else cidx = ci->second;
        // output WAV
        uint32_t ls = h.loop_start_frame * 1024 + h.loop_start_delay - h.delay;
        uint32_t le = h.loop_end_frame * 1024 + (1024 - h.loop_end_padding) - h.delay;
        out_pos = wav_item_start(out_pos, h.loop_flag ? 0x70 : 0x2C);
        j->out_offsets[i] = out_pos;
        Image im; im.dst = out_pos; im.bytes.assign(h.loop_flag ? 0x70 : 0x2C, 0);
        uint32_t wh = wav_write_header(im.bytes.data(), h.channels, h.rate, spc, h.loop_flag != 0, ls, le);
        j->images.push_back(std::move(im));
        HcaStream S; memset(&S, 0, sizeof S);
        S.src_offset = it.off(i) + hs_arg; S.dst_offset = out_pos + wh; S.format = fidx; S.cipher = cidx; S.frames = frames;
        S.delay = h.delay; S.samples = spc; S.item = i; S.float_offset = float_pos;
        float_pos += (uint64_t)frames * 1024 * h.channels;
        streams.push_back(S);
        out_pos = out_pos + wh + (uint64_t)spc * h.channels * 2;
        j->units += frames;
        j->alg_bytes += (uint64_t)frames * (h.frame_size + 2048ull * h.channels);
    }
    out_pos = align_up(out_pos, 64);
    j->out_offsets[n] = out_pos;
    j->float_offsets[n] = float_pos;
    // true (unaligned) end of each item for consumers: offsets[i+1] is the aligned start of the next item, so the
    // item length is carried by the WAV header itself (RIFF size + 8).
    j->out_bytes = out_pos;
    std::stable_sort(streams.begin(), streams.end(), [](const HcaStream& a, const HcaStream& b) { return a.format < b.format; });
    uint64_t scratch = 0;
    j->n_cipher = (uint32_t)(cipher.size() / 256);
    bool cipher_identity = j->n_cipher <= 1;                  // one table, and it maps every byte to itself
    for (size_t k = 0; k < cipher.size() && cipher_identity; k++) if (cipher[k] != (uint8_t)k) cipher_identity = false;
    for (size_t b = 0; b < streams.size();) {
        size_t e = b; uint32_t frames = 0, runs = 0;
        const HcaFormat& F = formats[streams[b].format];
        // A transform wave walks a RUN of consecutive frames of one stream; every run starts with a pass that only rebuilds the
        // overlap state (the halo), so long runs are cheaper -- where the group has waves enough to fill the chip many times over:
        // transform of 4.69 M frames 10.6 ms at 8 frames, 10.2 at 16, 10.05 at 32, 10.65 at 64; 1.88 M frames 4.06 / 3.89 / 3.97;
        // 469 k and fewer: no gain, and 30 k frames lose (0.085 / 0.100 / 0.142 ms)  (tools/debug/dec_runs.py).
        uint64_t group_frames = 0;
        for (size_t k = b; k < streams.size() && streams[k].format == streams[b].format; k++) group_frames += streams[k].frames;
        uint32_t run_frames = group_frames >= 3000000 ? 32u : (group_frames >= 1200000 ? 16u : 8u);
        if (knobs().hca_run) run_frames = (uint32_t)knobs().hca_run;
        while (e < streams.size() && streams[e].format == streams[b].format) {
            streams[e].first_frame = frames; streams[e].first_run = runs; streams[e].scratch_offset = scratch;
            frames += streams[e].frames; runs += (streams[e].frames + run_frames - 1) / run_frames;
            scratch += (uint64_t)streams[e].frames * F.record_bytes;
            e++;
        }
        HcaDecArgs a; memset(&a, 0, sizeof a);
        a.format = streams[b].format; a.stream_begin = (uint32_t)b; a.stream_end = (uint32_t)e; a.frames = frames; a.runs = runs; a.run_frames = run_frames;
        a.n_cipher = j->n_cipher; a.channels = F.channels;
        a.cipher_identity = cipher_identity ? 1 : 0; a.in_bytes = j->in_bytes;
        a.plain = (F.bands_per_hfr_group == 0 && F.stereo_bands == 0) ? 1 : 0;
        a.noise_fill = F.min_res == 0 ? 1 : 0;
        if (a.noise_fill) a.plain = 0;                                 // noise reconstruction lives in the general variant of the transform
        // int8 lines are read by k_hca_transform_plain<1>, <2>, <4> and (joint stereo / HFR formats) k_hca_transform<false, 1>,
        // <false, 2>; v3.0 noise fill and the wider layouts keep int16
        a.narrow = (!a.noise_fill && (a.channels <= 2 || a.plain)) ? 1 : 0;         // (plain formats of any channel count: k_hca_transform_plain in channel groups)
        a.pairs_even = 1;
        for (uint32_t c = 0; c < F.channels; c += 2) if (F.type[c] == CRI_CH_SECONDARY) a.pairs_even = 0;
        // in-lane instances: 1 / 2 / 4 channels with pairs on even channels (joint, or noise fill); every other joint or noise-fill layout
        // goes to the wide joint form -- groups of up to four consecutive channels, cut so that no pair is split
        a.inlane = (!a.plain && ((a.pairs_even && (a.channels == 1 || a.channels == 2 || a.channels == 4)) || a.channels >= 3)) ? 1 : 0;
        a.wide_waves = 0;
        for (uint32_t cb = 0; cb < F.channels; a.wide_waves++) {
            uint32_t n = F.channels - cb < 4 ? F.channels - cb : 4;
            if (n == 4 && cb + 4 < F.channels && F.type[cb + 3] == CRI_CH_PRIMARY && F.type[cb + 4] == CRI_CH_SECONDARY) n = 3;
            cb += n;
        }
        if (a.inlane && knobs().no_inlane) a.inlane = 0;                   // (parity tests: the general transform instead)
        if (a.channels > 8) { a.inlane = 0; a.narrow = 0; a.wide_waves = 0; }   // the wide forms are built for up to eight channels (two waves of
                                                                           // four): 9 .. 16 go to k_hca_transform_generic, which reads int16 lines
        if (a.inlane) a.narrow = 1;                                        // (every in-lane instance reads either form, noise fill included)
        if (a.channels == 4 && !a.inlane && !a.plain) a.narrow = 0;         // k_hca_transform<false, 4> reads int16 lines only
        j->hca_dec.push_back(a);
        j->hca_group_first_record.push_back(streams[b].scratch_offset);
        b = e;
    }
    // the quantised lines (tile-major) and the band code descriptions of every format group follow the frame records
    for (auto& a : j->hca_dec) {
        scratch = align_up(scratch, 4096);
        a.qc_offset = scratch;
        scratch += (uint64_t)((a.frames + 63) / 64) * HCA_QC_TILE(a.channels);
        a.resg_offset = scratch;
        scratch += (uint64_t)((a.frames + 63) / 64) * a.channels * 8 * 64 * 16;
    }
    j->scratch_bytes = scratch;
    int rc = 0;
    if (formats.empty()) { HcaFormat F; memset(&F, 0, sizeof F); formats.push_back(F); }
    if (streams.empty()) { HcaStream S; memset(&S, 0, sizeof S); streams.push_back(S); }
    if (cipher.empty()) cipher.assign(256, 0);
    if ((rc = j->d_formats.upload(formats)) || (rc = j->d_streams.upload(streams)) || (rc = j->d_cipher.upload(cipher)) ||
        (rc = j->d_ath.upload(ath)) || (rc = j->upload_images())) { delete j; return rc; }
    j->hca_streams_host = streams;
    if ((rc = j->meta.commit())) { delete j; return rc; }
    *out = j;
    return 0;
}

// (what the pipelined host path needs to plan a job that is not one format group again as parts: host_parts_ready)
static void hca_decode_keep_keys(cri_job* j, const uint64_t* keys, const uint16_t* subkeys) {
    if (keys) j->part_keys.assign(keys, keys + j->n);
    if (subkeys) j->part_subkeys.assign(subkeys, subkeys + j->n);
    j->partable = true;
}
extern "C" int cri_job_create_hca_decode(const uint8_t* blob, const uint64_t* offsets, uint32_t n, const uint64_t* keys,
                                         const uint16_t* subkeys, cri_job** job) {
    if (!blob || !offsets) return CRI_ERR_INVALID_ARG;
    const int rc = create_hca_decode(ItemSrc::from_blob(blob, offsets, n), keys, subkeys, nullptr, job);
    if (!rc) hca_decode_keep_keys(*job, keys, subkeys);
    return rc;
}
extern "C" int cri_job_create_hca_decode_items(const cri_items* items, const uint64_t* keys, const uint16_t* subkeys, cri_job** job) {
    ItemSrc it; int rc = ItemSrc::from_items(items, it);
    if (!rc) rc = create_hca_decode(it, keys, subkeys, nullptr, job);
    if (!rc) hca_decode_keep_keys(*job, keys, subkeys);
    return rc;
}


// Mapping choice: wave-per-file fills the chip with ~1 k files but spends most lanes idle in the serial
// section; lane-per-chain is cheaper per sample but one wave of 64 chains per SIMD needs 65 536 chains just to occupy the chip
// once.  Measured on 1 s stereo files (tools/debug/adx_chain_sweep.py, profiles/r02_adx_chain_sweep.jsonl): the lane-per-chain
// kernels take the same 7.5 ms (decode) / 17 ms (encode) for anything up to 32 768 chains and scale linearly beyond; the
// wave-per-file kernels saturate at 4.5 G (decode) / 3.1 G (encode) blocks/s from ~4 000 files.  Decode crosses over between
// 8 192 and 16 384 files (5.6 vs 7.5 ms, 10.9 vs 7.7 ms); encode between 32 768 and 100 000 (31.7 vs 35.0 ms; 3.1 vs 3.4 G blocks/s).
// CRICODECS_ADX_MAPPING = "chain" | "file" overrides the choice (tests use it to cover both kernels).
static bool adx_pick_wave_per_file(bool all_std, size_t n_streams, bool encode) {
    if (!all_std || n_streams == 0) return false;
    const int m = knobs().adx_mapping;
    if (m == ADX_MAP_CHAIN) return false;
    if (m == ADX_MAP_FILE) return true;
    return n_streams <= (encode ? 65536u : 12288u);
}

// Segmented chains (k_adx_seg_*, cri_adx.hip).  A decode that starts from a wrong history merges with the right one after a
// time that scales with 1 / (4096 - c0 - c1) (measured on synthetic material, oracle-side: mean 16 000 / g samples, 99.5 %
// within 5 x that); the warm-up is 3.5 x the mean -- a few per cent of the segments then need the repair pass, which costs a
// few more rows -- and a segment is at least three warm-ups long.  Segment counts are chosen for ~256 K lanes per job (four
// waves per SIMD).  Only the time depends on any of this.  Returns false when no file would get more than one segment (or the
// layout is not the standard one): the unsegmented kernels stay.
// CRICODECS_ADX_MAPPING = "seg" forces it where it applies, "chain" / "file" pick the unsegmented kernels.
static bool adx_plan_segments(std::vector<AdxStream>& streams, bool encode) {
    const int m = knobs().adx_mapping;
    if (m == ADX_MAP_CHAIN || m == ADX_MAP_FILE) return false;
    if (streams.empty()) return false;
    for (const AdxStream& S : streams) {
        if (!(S.blocksize == 18 && S.bitdepth == 4 && (S.mode == 2 || S.mode == 3 || (encode && S.mode == 4)))) return false;
        if (encode && S.channels > 2) return false;
    }
    const uint64_t warm_pct = knobs().adx_warm_pct;              // (100 outside the parity tests)
    if (encode) {
        // a WAVE per (file, segment): four waves per SIMD fill the chip (the wave-per-file encoder's rate stops growing there); the
        // encoder merges later than the decoder -- 120 rows on average for tonal material, 500 for sparse, at the standard coefficients
        const uint64_t w_target = std::max<uint64_t>(1, 4096 / streams.size());
        bool any = false;
        for (AdxStream& S : streams) {
            const int64_t g = S.mode == 2 ? 64 : 4096 - (int64_t)S.coef0 - (int64_t)S.coef1;
            S.seg_rows = S.frames; S.seg_count = S.frames ? 1 : 0; S.warm_rows = 0;
            if (g <= 0 || !S.frames) continue;
            const uint64_t warm = std::max<uint64_t>(4, (384ull * 39 * warm_pct / 100 / (uint64_t)g + 3) / 4 * 4);
            uint64_t rows = std::max<uint64_t>((S.frames + w_target - 1) / w_target, 3 * warm);
            rows = (rows + 3) / 4 * 4;                               // rounds of four rows: checkpoints sit on multiples of four
            if (rows >= S.frames) continue;
            S.seg_rows = (uint32_t)rows; S.seg_count = (uint32_t)((S.frames + rows - 1) / rows); S.warm_rows = (uint32_t)warm;
            any = true;
        }
        return any || m == ADX_MAP_SEG;
    }
    const uint64_t seg_mult = knobs().adx_seglen ? knobs().adx_seglen : 3;      // least segment length in warm-ups
    auto warm_of = [&](int64_t g) { return std::max<uint64_t>(1, (56000ull * warm_pct / 100 / (uint64_t)g + 31) / 32); };
    auto g_of = [](const AdxStream& S) { return S.mode == 2 ? (int64_t)64 : 4096 - (int64_t)S.coef0 - (int64_t)S.coef1; };     // (mode 2: the slowest of the four static filters)
    // The segment length: a launch takes the LONGER of (its longest lane: one row after the other, ~3 issue times a row at the two or
    // three waves per SIMD these launches have) and (all lanes' rows over all SIMDs).  A bank of 100 000 clips of 0.05-2 s has lanes
    // enough uncut, but then its 2 s clips are lanes of 3000 rows that the launch waits for (3.7 ms; the clips cut: see DESIGN section 2)
    // -- so the length is chosen per job over a few candidates (in warm-ups), never below seg_mult of them.
    // (candidate 0 is the rule of round 3 -- lanes enough for four waves per SIMD, segments of seg_mult warm-ups at least -- and stays
    //  unless a cap beats it by a fifth: between plans of similar cost the repairs decide, and they favour the shorter segments it makes)
    uint64_t chains = 0;
    for (const AdxStream& S : streams) chains += S.channels;
    const uint64_t p_target = std::max<uint64_t>(1, 262144 / chains);
    auto rows_of = [&](const AdxStream& S, uint64_t cand, uint64_t warm) {
        return cand ? std::max<uint64_t>(seg_mult, cand) * warm : std::max<uint64_t>((S.frames + p_target - 1) / p_target, seg_mult * warm);
    };
    uint64_t cap_mult = 0;                                           // a segment's most rows, in warm-ups (0: the rule above)
    {
        const uint32_t nsimd = device_simds();
        uint64_t best = ~0ull;
        for (uint64_t cand : {(uint64_t)0, (uint64_t)48, (uint64_t)24, (uint64_t)12, (uint64_t)8, (uint64_t)6, (uint64_t)4, (uint64_t)3}) {
            uint64_t longest = 0, total = 0;
            for (const AdxStream& S : streams) {
                const int64_t g = g_of(S);
                uint64_t rows = S.frames, segs = 1, warm = 0;
                if (g > 0 && S.frames) {
                    warm = warm_of(g);
                    rows = std::min<uint64_t>(S.frames, rows_of(S, cand, warm));
                    segs = (S.frames + rows - 1) / rows;
                }
                longest = std::max<uint64_t>(longest, rows + (segs > 1 ? warm : 0));
                total += (uint64_t)S.channels * (S.frames + (segs - 1) * warm);
            }
            const uint64_t cost = std::max<uint64_t>(3 * longest, total / 64 / nsimd);
            if (cand == 0) best = cost - cost / 5;
            else if (cost < best) { best = cost; cap_mult = cand; }
        }
    }
    bool any = false;
    for (AdxStream& S : streams) {
        const int64_t g = g_of(S);
        S.rows_avail = (uint32_t)std::min<uint64_t>(S.frames, (S.src_end - S.src_offset) / (18ull * S.channels));
        S.seg_rows = S.frames; S.seg_count = S.frames ? 1 : 0; S.warm_rows = 0;
        if (g <= 0 || !S.frames) continue;                           // no decay (high-pass 0): one segment, i.e. the plain serial decode
        const uint64_t warm = warm_of(g);
        const uint64_t rows = rows_of(S, cap_mult, warm);
        if (rows >= S.frames) continue;
        S.seg_rows = (uint32_t)rows; S.seg_count = (uint32_t)((S.frames + rows - 1) / rows); S.warm_rows = (uint32_t)warm;
        any = true;
    }
    return any || m == ADX_MAP_SEG;
}

// LDS plan of the lane-per-chain ADX kernels.  A wave stages, for each of its files, T rows of blocks and of PCM in LDS
// (regions padded to 4 bytes + 4), so what a wave needs is the sum over ITS files, not 64 x the largest item of the batch:
// one ADX item with bitdepth 1 and blocksize 255 (8 * 253 samples per block, 4.3 KB per chain and row) must not size -- or
// fail -- the launch of everybody else.  place() keeps a file inside one wave (as before) and also starts a new wave when
// the wave's rows would pass ADX_LDS_ROW_LIMIT; an item that does not fit a wave on its own is left to the caller
// (CRI_ERR_UNSUPPORTED for that item alone).  finish() picks T for ~56 KB per wave (occupancy) and the exact maxima.
static const uint32_t ADX_LDS_ROW_LIMIT = 150 * 1024;
// wave-per-file kernels: a workgroup is a file and runs as long as the file is -- the longest ones are started first
static std::vector<uint32_t> adx_longest_first(const std::vector<AdxStream>& streams) {
    std::vector<uint32_t> order(streams.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = (uint32_t)i;
    bool same = true;
    for (size_t i = 1; i < streams.size(); i++) if (streams[i].frames != streams[0].frames) same = false;
    if (same) return {};                                  // nothing to gain: the kernels take workgroup b = stream b
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return streams[x].frames > streams[y].frames; });
    return order;
}

struct AdxWavePlan {
    std::vector<uint32_t> in_row, out_row, files;        // per wave: sum of the files' row bytes (in / out), file count
    static bool fits(uint32_t channels, uint32_t in_row_bytes, uint32_t out_row_bytes) {
        return channels <= 64 && ((in_row_bytes + 3) & ~3u) + ((out_row_bytes + 3) & ~3u) + 8 <= ADX_LDS_ROW_LIMIT;
    }
    bool place(std::vector<uint32_t>& chain_stream, std::vector<int16_t>& history, uint32_t channels, uint32_t in_row_bytes, uint32_t out_row_bytes) {
        const uint32_t need = ((in_row_bytes + 3) & ~3u) + ((out_row_bytes + 3) & ~3u) + 8;
        if (!fits(channels, in_row_bytes, out_row_bytes)) return false;
        size_t w = chain_stream.size() / 64;
        if (in_row.size() <= w) { in_row.resize(w + 1, 0); out_row.resize(w + 1, 0); files.resize(w + 1, 0); }
        const uint32_t used = ((in_row[w] + 3) & ~3u) + ((out_row[w] + 3) & ~3u) + 8 * files[w] + 8 * 64;
        if ((chain_stream.size() % 64) + channels > 64 || (files[w] && used + need > ADX_LDS_ROW_LIMIT)) {
            while (chain_stream.size() % 64) { chain_stream.push_back(0xFFFFFFFFu); history.push_back(0); history.push_back(0); }
            w = chain_stream.size() / 64;
            in_row.resize(w + 1, 0); out_row.resize(w + 1, 0); files.resize(w + 1, 0);
        }
        in_row[w] += (in_row_bytes + 3) & ~3u; out_row[w] += (out_row_bytes + 3) & ~3u; files[w]++;
        return true;
    }
    void finish(AdxArgs& a) const {
        uint32_t worst = 1;
        for (size_t w = 0; w < in_row.size(); w++) worst = std::max(worst, in_row[w] + out_row[w]);
        uint32_t T = (56 * 1024) / worst;
        T = T < 1 ? 1 : (T > 16 ? 16 : T);
        uint32_t in_max = 0, out_max = 0;
        for (size_t w = 0; w < in_row.size(); w++) {      // region = T * row bytes rounded up to 4, + 4 (each row sum is already a multiple of 4 per file)
            in_max = std::max(in_max, T * in_row[w] + 8 * files[w]);
            out_max = std::max(out_max, T * out_row[w] + 8 * files[w]);
        }
        a.rows_per_round = T;
        a.lds_in_bytes = ((in_max + 15) & ~15u) + 64;
        a.lds_out_bytes = ((out_max + 15) & ~15u) + 64;
    }
};

// ------------------------------------------------------------------------------------------------ ADX decode
static int create_adx_decode(const ItemSrc& it, cri_job** out, const uint8_t* take = nullptr, uint8_t take_kind = 0) {
    const uint32_t n = it.n;
    if (!it.ok() || !out) return CRI_ERR_INVALID_ARG;
    if (!cri_device_available()) return CRI_ERR_HIP;
    cri_job* j = new_job(CRI_JOB_ADX_DECODE, it);
    if (!j) return CRI_ERR_HIP;
    j->dominant = "k_adx_decode";
    std::vector<AdxStream> streams, pend; std::vector<uint32_t> chain_stream; std::vector<int16_t> history, pend_hist;
    AdxWavePlan plan;
    bool all_std = true;
    uint64_t out_pos = 0;
    for (uint32_t i = 0; i < n; i++) {
        j->out_offsets[i] = out_pos;
        if (take && take[i] != take_kind) { j->host_status[i] = CRI_ITEM_SKIPPED; continue; }
        const uint8_t* d = it.ptr(i);
        size_t len = it.len(i);
        AdxHeader h;
        int rc = adx_parse_header(d, len, h);
        if (rc) { j->host_status[i] = rc; continue; }
        if ((uint64_t)h.sample_count * h.channels * 2 > 0x7FFFFF00ull) { j->host_status[i] = CRI_ERR_INVALID_ARG; continue; }
        out_pos = wav_item_start(out_pos, h.looping ? 0x70 : 0x2C);
        j->out_offsets[i] = out_pos;
        Image im; im.dst = out_pos; im.bytes.assign(h.looping ? 0x70 : 0x2C, 0);
        uint32_t wh = wav_write_header(im.bytes.data(), h.channels, h.rate, h.sample_count, h.looping, h.loop_start, h.loop_end);
        j->images.push_back(std::move(im));
        AdxStream S; memset(&S, 0, sizeof S);
        S.src_offset = it.off(i) + h.data_offset + 4; S.src_end = it.off(i) + it.len(i); S.dst_offset = out_pos + wh;
        if (S.src_offset > S.src_end) S.src_offset = S.src_end;
        S.frames = h.blocks; S.channels = h.channels; S.blocksize = h.blocksize; S.bitdepth = h.bitdepth; S.mode = h.mode;
        S.samples_per_block = h.samples_per_block; S.coef0 = h.coef[0]; S.coef1 = h.coef[1]; S.samples = h.sample_count;
        // (more than 64 channels, or one block row that does not fit a wave's LDS: bitdepth 1 with tens of channels)
        if (!AdxWavePlan::fits(h.channels, h.channels * h.blocksize, h.channels * h.samples_per_block * 2)) {
            j->host_status[i] = CRI_ERR_UNSUPPORTED; j->images.pop_back(); continue;
        }
        S.item = i;
        if (!(h.blocksize == 18 && h.bitdepth == 4 && h.channels <= 2)) all_std = false;
        pend.push_back(S);
        for (uint32_t c = 0; c < h.channels; c++) { pend_hist.push_back(h.history[2 * c]); pend_hist.push_back(h.history[2 * c + 1]); }
        out_pos = out_pos + wh + (uint64_t)h.sample_count * h.channels * 2;
        j->units += h.blocks; j->units2 += (uint64_t)h.blocks * h.channels;
        j->alg_bytes += (uint64_t)h.blocks * h.channels * (h.blocksize + 2ull * h.samples_per_block);
    }
    out_pos = align_up(out_pos, 64);
    j->out_offsets[n] = out_pos; j->out_bytes = out_pos;
    if (adx_plan_segments(pend, false)) {
        // segmented chains: every (segment, channel) of every file is a lane; no LDS staging, no padding besides even pair starts
        j->adx_seg = true; j->dominant = "k_adx_seg_decode";
        std::vector<uint32_t> seq(pend.size()), hist_at(pend.size()), seg_first;
        for (size_t k = 0, hpos = 0; k < pend.size(); k++) { seq[k] = (uint32_t)k; hist_at[k] = (uint32_t)hpos; hpos += 2 * pend[k].channels; }
        std::stable_sort(seq.begin(), seq.end(), [&](uint32_t x, uint32_t y) { return pend[x].seg_rows + pend[x].warm_rows > pend[y].seg_rows + pend[y].warm_rows; });
        uint32_t lanes = 0, chains = 0;
        for (uint32_t k : seq) {
            AdxStream S = pend[k];
            if (S.channels == 2 && (lanes & 1)) lanes++;
            S.first_seg = lanes; S.first_chain = chains; S.hist_offset = chains;
            seg_first.push_back(lanes);
            lanes += S.seg_count * S.channels; chains += S.channels;
            if (S.seg_count > j->adx.seg_max_count) j->adx.seg_max_count = S.seg_count;
            for (uint32_t c = 0; c < S.channels; c++) { history.push_back(pend_hist[hist_at[k] + 2 * c]); history.push_back(pend_hist[hist_at[k] + 2 * c + 1]); }
            streams.push_back(S);
        }
        seg_first.push_back(lanes);
        j->adx.chains = chains; j->adx.seg_lanes = lanes; j->adx.n_streams = (uint32_t)streams.size();
        j->adx_streams = (uint32_t)streams.size();
        j->adx_seg_flags_offset = align_up(16ull * lanes, 256);
        j->scratch_bytes = j->adx_seg_flags_offset + align_up(4ull * (chains + 1 + streams.size()), 256);      // flags, then the fallback's list (k_adx_seg_list)
        if (history.empty()) history.assign(2, 0);
        int rc = 0;
        if ((rc = j->d_adx_streams.upload(streams)) || (rc = j->d_seg_chain.upload(seg_first)) || (rc = j->d_history.upload(history)) || (rc = j->upload_images())) { delete j; return rc; }
        if ((rc = j->meta.commit())) { delete j; return rc; }
        *out = j;
        return 0;
    }
    j->adx_wave_per_file = adx_pick_wave_per_file(all_std, pend.size(), false);
    if (j->adx_wave_per_file) j->dominant = "k_adx_decode_wpf";
    {   // chains are laid out now.  Lane per chain: a wave lasts as long as the longest of its 64 chains, so the files go in by
        // length (outputs stay where the item order put them); wave per file: item order, and the kernel gets `wpf_order`
        std::vector<uint32_t> seq(pend.size()), hist_at(pend.size());
        for (size_t k = 0, hpos = 0; k < pend.size(); k++) { seq[k] = (uint32_t)k; hist_at[k] = (uint32_t)hpos; hpos += 2 * pend[k].channels; }
        if (!j->adx_wave_per_file) std::stable_sort(seq.begin(), seq.end(), [&](uint32_t x, uint32_t y) { return pend[x].frames > pend[y].frames; });
        for (uint32_t k : seq) {
            AdxStream S = pend[k];
            plan.place(chain_stream, history, S.channels, S.channels * S.blocksize, S.channels * S.samples_per_block * 2);
            S.first_chain = (uint32_t)chain_stream.size(); S.hist_offset = S.first_chain;
            for (uint32_t c = 0; c < S.channels; c++) {
                chain_stream.push_back((uint32_t)streams.size());
                history.push_back(pend_hist[hist_at[k] + 2 * c]); history.push_back(pend_hist[hist_at[k] + 2 * c + 1]);
            }
            streams.push_back(S);
        }
    }
    j->adx.chains = (uint32_t)chain_stream.size();
    plan.finish(j->adx);
    j->adx_streams = (uint32_t)streams.size();
    const std::vector<uint32_t> order = j->adx_wave_per_file ? adx_longest_first(streams) : std::vector<uint32_t>();
    if (streams.empty()) { AdxStream S; memset(&S, 0, sizeof S); streams.push_back(S); }
    if (chain_stream.empty()) { chain_stream.push_back(0xFFFFFFFFu); history.assign(2, 0); }
    int rc = 0;
    if ((rc = j->d_adx_streams.upload(streams)) || (rc = j->d_chain_stream.upload(chain_stream)) || (rc = j->d_history.upload(history)) ||
        (!order.empty() && (rc = j->d_adx_order.upload(order))) || (rc = j->upload_images())) { delete j; return rc; }
    if ((rc = j->meta.commit())) { delete j; return rc; }
    *out = j;
    return 0;
}

extern "C" int cri_job_create_adx_decode(const uint8_t* blob, const uint64_t* offsets, uint32_t n, cri_job** out) {
    if (!blob || !offsets) return CRI_ERR_INVALID_ARG;
    const int rc = create_adx_decode(ItemSrc::from_blob(blob, offsets, n), out);
    if (!rc) (*out)->partable = true;
    return rc;
}
extern "C" int cri_job_create_adx_decode_items(const cri_items* items, cri_job** job) {
    ItemSrc it; int rc = ItemSrc::from_items(items, it);
    if (!rc) rc = create_adx_decode(it, job);
    if (!rc) (*job)->partable = true;
    return rc;
}

// ------------------------------------------------------------------------------------------------ AWB (AFS2) front door
// Header and offset table as PyCriCodecs/awb.py:32-52 reads them ("<4sBBHIHH": magic, version, offset int size, id int
// size, file count, alignment, subkey; then ids, then count+1 offsets, each rounded up to the alignment); item ranges as
// getfiles() yields them (awb.py:83-88): item k = [offset k, offset k+1), the first one starting at the aligned header size.
extern "C" int cri_awb_index(const uint8_t* awb, size_t len, uint32_t* n_items, uint32_t* align_out, uint16_t* subkey,
                             uint32_t* header_size, uint64_t* offsets, uint8_t* kinds, uint32_t cap) {
    if (!awb || !n_items) return CRI_ERR_INVALID_ARG;
    if (len < 16 || memcmp(awb, "AFS2", 4) != 0) return CRI_ERR_AWB_HEADER;
    const uint32_t osz = awb[5], isz = le16(awb + 6), n = le32(awb + 8), align = le16(awb + 12);
    if ((osz != 1 && osz != 2 && osz != 4 && osz != 8) || (isz != 1 && isz != 2 && isz != 4 && isz != 8)) return CRI_ERR_AWB_INTSIZE;
    if (align == 0) return CRI_ERR_AWB_HEADER;               // the reference divides by it
    const uint64_t table = 16 + (uint64_t)isz * n, hs0 = table + (uint64_t)osz * ((uint64_t)n + 1);
    if (hs0 > len) return CRI_ERR_AWB_HEADER;
    *n_items = n;
    if (align_out) *align_out = align;
    if (subkey) *subkey = (uint16_t)le16(awb + 14);
    uint64_t hs = hs0 % align ? hs0 + (align - hs0 % align) : hs0;
    if (header_size) *header_size = (uint32_t)hs;
    if (!offsets) return 0;
    if (cap < n) return CRI_ERR_INVALID_ARG;
    uint64_t prev = 0;
    for (uint32_t k = 0; k <= n; k++) {
        const uint8_t* p = awb + table + (uint64_t)osz * k;
        uint64_t o = 0;
        for (uint32_t b = 0; b < osz; b++) o |= (uint64_t)p[b] << (8 * b);
        if (o % align) o += align - o % align;
        if (k == 0) o = hs;                                  // the reader starts at the aligned header size
        if (o > len) o = len;                                // a short read at the end of the file
        if (o < prev) return CRI_ERR_AWB_HEADER;
        offsets[k] = prev = o;
    }
    if (kinds) for (uint32_t k = 0; k < n; k++) {
        const uint8_t* d = awb + offsets[k];
        const uint64_t l = offsets[k + 1] - offsets[k];
        kinds[k] = CRI_AWB_OTHER;
        if (l >= 4 && (be32(d) == 0x48434100u || be32(d) == 0xC8C3C100u)) kinds[k] = CRI_AWB_HCA;   // HCAType.HCA / HCAType.EHCA (awb.py:60)
        else if (l >= 2 && d[0] == 0x80 && d[1] == 0x00) kinds[k] = CRI_AWB_ADX;
    }
    return 0;
}

extern "C" int cri_job_create_awb_decode(const uint8_t* awb, size_t len, uint64_t key, cri_job** hca_job, cri_job** adx_job) {
    if (!awb || !hca_job || !adx_job) return CRI_ERR_INVALID_ARG;
    *hca_job = *adx_job = nullptr;
    uint32_t n = 0; uint16_t subkey = 0;
    int rc = cri_awb_index(awb, len, &n, nullptr, &subkey, nullptr, nullptr, nullptr, 0);
    if (rc) return rc;
    std::vector<uint64_t> offsets((size_t)n + 1); std::vector<uint8_t> kinds(n ? n : 1);
    rc = cri_awb_index(awb, len, &n, nullptr, &subkey, nullptr, offsets.data(), kinds.data(), n);
    if (rc) return rc;
    std::vector<uint64_t> keys(n ? n : 1, key); std::vector<uint16_t> subkeys(n ? n : 1, subkey);   // awb.py:72: HCA(i, key=key, subkey=self.subkey)
    const ItemSrc it = ItemSrc::from_blob(awb, offsets.data(), n);
    rc = create_hca_decode(it, keys.data(), subkeys.data(), nullptr, hca_job, kinds.data(), CRI_AWB_HCA);
    if (rc) return rc;
    rc = create_adx_decode(it, adx_job, kinds.data(), CRI_AWB_ADX);
    if (rc) { cri_job_destroy(*hca_job); *hca_job = nullptr; }
    return rc;
}

// ------------------------------------------------------------------------------------------------ USM audio chunks
extern "C" int cri_usm_audio_mask(uint64_t key, uint8_t mask[32]) {
    if (!mask) return CRI_ERR_INVALID_ARG;
    // usm.py:60-117 (USM.init_key); key1 = low 32 bits, key2 = high 32 bits, both big-endian byte strings
    const uint8_t k1[4] = {(uint8_t)(key >> 24), (uint8_t)(key >> 16), (uint8_t)(key >> 8), (uint8_t)key};
    const uint8_t k2[4] = {(uint8_t)(key >> 56), (uint8_t)(key >> 48), (uint8_t)(key >> 40), (uint8_t)(key >> 32)};
    uint8_t t[32];
    t[0x00] = k1[3]; t[0x01] = k1[2]; t[0x02] = k1[1]; t[0x03] = (uint8_t)(k1[0] - 0x34);
    t[0x04] = (uint8_t)(k2[3] + 0xF9); t[0x05] = (uint8_t)(k2[2] ^ 0x13); t[0x06] = (uint8_t)(k2[1] + 0x61);
    t[0x07] = (uint8_t)(k1[3] ^ 0xFF); t[0x08] = (uint8_t)(k1[1] + k1[2]);
    t[0x09] = (uint8_t)(t[0x01] - t[0x07]); t[0x0A] = (uint8_t)(t[0x02] ^ 0xFF); t[0x0B] = (uint8_t)(t[0x01] ^ 0xFF);
    t[0x0C] = (uint8_t)(t[0x0B] + t[0x09]); t[0x0D] = (uint8_t)(t[0x08] - t[0x03]);
    t[0x0E] = (uint8_t)(t[0x0D] ^ 0xFF); t[0x0F] = (uint8_t)(t[0x0A] - t[0x0B]);
    t[0x10] = (uint8_t)(t[0x08] - t[0x0F]);
    t[0x11] = (uint8_t)(t[0x10] ^ t[0x07]); t[0x12] = (uint8_t)(t[0x0F] ^ 0xFF); t[0x13] = (uint8_t)(t[0x03] ^ 0x10);
    t[0x14] = (uint8_t)(t[0x04] - 0x32); t[0x15] = (uint8_t)(t[0x05] + 0xED); t[0x16] = (uint8_t)(t[0x06] ^ 0xF3);
    t[0x17] = (uint8_t)(t[0x13] - t[0x0F]); t[0x18] = (uint8_t)(t[0x15] + t[0x07]); t[0x19] = (uint8_t)(0x21 - t[0x13]);
    t[0x1A] = (uint8_t)(t[0x14] ^ t[0x17]); t[0x1B] = (uint8_t)(t[0x16] + t[0x16]);
    t[0x1C] = (uint8_t)(t[0x17] + 0x44); t[0x1D] = (uint8_t)(t[0x03] + t[0x04]); t[0x1E] = (uint8_t)(t[0x05] - t[0x16]);
    t[0x1F] = (uint8_t)(t[0x1D] ^ t[0x13]);
    static const char urug[4] = {'U', 'R', 'U', 'C'};
    for (int x = 0; x < 32; x++) mask[x] = (x & 1) ? (uint8_t)urug[(x >> 1) & 3] : (uint8_t)(t[x] ^ 0xFF);   // videomask2 = ~t
    return 0;
}

static bool usm_known_fourcc(const uint8_t* p) {             // chunk.py:14-24
    static const char* names[] = {"CRID", "SFSH", "@SFV", "@SFA", "@ALP", "@CUE", "@SBT", "@AHX", "@USR", "@PST"};
    for (const char* n : names) if (memcmp(p, n, 4) == 0) return true;
    return false;
}

extern "C" int cri_usm_index(const uint8_t* usm, size_t len, cri_usm_chunk* chunks, uint32_t cap, uint32_t* count) {
    if (!usm || !count) return CRI_ERR_INVALID_ARG;
    if (len < 4 || memcmp(usm, "CRID", 4) != 0) return CRI_ERR_USM_HEADER;
    uint32_t n = 0;
    uint64_t pos = 0;
    while (pos < len) {                                      // usm.py:138-190: USMChunkHeader = ">4sIBBHBBBBIIII"
        if (len - pos < 0x20) return CRI_ERR_USM_CHUNK;
        const uint8_t* h = usm + pos;
        if (!usm_known_fourcc(h)) return CRI_ERR_USM_CHUNK;
        const uint32_t size = be32(h + 4), offset = h[9], padding = ((uint32_t)h[10] << 8) | h[11];
        if (size < 0x18 || offset < 0x18) return CRI_ERR_USM_CHUNK;
        const uint64_t data = pos + 0x20, data_len = std::min<uint64_t>(size - 0x18, len - data);
        const uint64_t skip = std::min<uint64_t>(offset - 0x18, data_len);
        if (chunks) {
            if (n >= cap) return CRI_ERR_INVALID_ARG;
            cri_usm_chunk& c = chunks[n];
            memcpy(c.fourcc, h, 4); c.chno = h[12]; c.type = h[15]; c.padding = padding;
            c.payload_offset = data + skip; c.payload_len = (uint32_t)(data_len - skip);
            c.frame_time = be32(h + 16); c.frame_rate = be32(h + 20); c.pad = 0;
        }
        n++;
        pos = data + (uint64_t)(size - 0x18);
    }
    *count = n;
    return 0;
}

// One unsigned field of row 0 of an @UTF table, by column name (the reference's UTF class, utf.py:30-34, 53-75: 32-byte
// header, then per column a flag byte (storage << 4 | type) and a name offset; storage 3 carries the value inline, 5 in
// the row, 1 is the type's zero).  Enough to read AUDIO_HDRINFO.audio_codec (usm.py:165-168); encrypted tables are not read.
static bool utf_field_u32(const uint8_t* t, size_t len, const char* name, uint32_t* value) {
    static const uint8_t size_of[16] = {1, 1, 2, 2, 4, 4, 8, 8, 4, 8, 4, 8, 0, 0, 0, 0};   // "BbHhIiQqfdI" + bytes (offset, size)
    if (len < 32 || memcmp(t, "@UTF", 4) != 0) return false;
    const uint64_t rows = (uint64_t)be32(t + 8) + 8, strings = (uint64_t)be32(t + 12) + 8;
    const uint32_t ncol = be16(t + 24);
    uint64_t pos = 32, rowpos = rows;
    const size_t nlen = strlen(name);
    for (uint32_t c = 0; c < ncol; c++) {
        if (pos + 5 > len) return false;
        const uint32_t flag = t[pos], st = flag >> 4, sz = size_of[flag & 15];
        const uint64_t noff = strings + be32(t + pos + 1);
        pos += 5;
        const uint8_t* v = nullptr;
        if (st == 3) { v = t + pos; if (pos + sz > len) return false; pos += sz; }
        else if (st == 5) { v = t + rowpos; if (rowpos + sz > len) return false; rowpos += sz; }
        else if (st != 1) return false;
        if (sz == 0) return false;
        if (noff + nlen + 1 <= len && memcmp(t + noff, name, nlen + 1) == 0) {
            uint64_t x = 0;
            if (v) for (uint32_t k = 0; k < sz; k++) x = (x << 8) | v[k];
            *value = (uint32_t)x;
            return true;
        }
    }
    return false;
}

static int upload_segments(cri_job* j, const std::vector<Segment>& segs, const uint8_t mask[32]) {
    j->seg.n = (uint32_t)segs.size();
    memcpy(j->seg.mask, mask, 32);
    j->dominant = "k_usm_segments";
    if (segs.empty()) return 0;
    return j->d_segs.upload(segs);
}

extern "C" int cri_job_create_usm_audio_demux(const uint8_t* usm, size_t len, uint64_t key, uint32_t decrypt, cri_job** out) {
    if (!usm || !out) return CRI_ERR_INVALID_ARG;
    if (!cri_device_available()) return CRI_ERR_HIP;
    uint32_t nc = 0;
    int rc = cri_usm_index(usm, len, nullptr, 0, &nc);
    if (rc) return rc;
    std::vector<cri_usm_chunk> chunks(nc ? nc : 1);
    rc = cri_usm_index(usm, len, chunks.data(), nc, &nc);
    if (rc) return rc;
    // channels that carry @SFA data, ascending
    std::vector<int> item_of(256, -1);
    std::vector<uint32_t> chnos;
    for (uint32_t k = 0; k < nc; k++) if (memcmp(chunks[k].fourcc, "@SFA", 4) == 0 && chunks[k].type == 0) item_of[chunks[k].chno & 255] = 0;
    for (uint32_t c = 0; c < 256; c++) if (item_of[c] == 0) { item_of[c] = (int)chnos.size(); chnos.push_back(c); }
    const uint32_t n = (uint32_t)chnos.size();
    std::vector<uint64_t> in_off(n + 1, 0);
    for (uint32_t i = 0; i <= n; i++) in_off[i] = i == n ? (uint64_t)len : 0;          // every item reads the one container
    cri_job* j = new_job(CRI_JOB_USM_DEMUX, in_off.data(), n);
    if (!j) return CRI_ERR_HIP;
    j->in_bytes = len;
    std::vector<uint64_t> size(n, 0);
    std::vector<uint32_t> codec(n, 0);
    // The reference keeps ONE codec for the whole container: audio_codec of the most recent @SFA header chunk (type 1,
    // usm.py:165-168), and masks a payload when that is 2.  Without any header chunk the first payload decides (80 00 = ADX).
    std::vector<uint32_t> codec_at(nc, 0);
    uint32_t cur_codec = 0;
    for (uint32_t k = 0; k < nc; k++) {
        const cri_usm_chunk& c = chunks[k];
        if (memcmp(c.fourcc, "@SFA", 4) != 0) continue;
        if (c.type == 1) {
            uint32_t v = 0;
            if (utf_field_u32(usm + c.payload_offset, c.payload_len, "audio_codec", &v)) cur_codec = v;
            continue;
        }
        if (c.type != 0) continue;
        const int i = item_of[c.chno & 255];
        const uint32_t sniff = (c.payload_len >= 2 && usm[c.payload_offset] == 0x80 && usm[c.payload_offset + 1] == 0x00) ? CRI_USM_CODEC_ADX : CRI_USM_CODEC_HCA;
        codec_at[k] = cur_codec ? cur_codec : (codec[i] ? codec[i] : sniff);
        if (!codec[i]) codec[i] = cur_codec ? cur_codec : sniff;
        size[i] += c.payload_len > c.padding ? c.payload_len - c.padding : 0;
    }
    uint64_t pos = 0;
    std::vector<uint64_t> cur(n);
    for (uint32_t i = 0; i < n; i++) { j->out_offsets[i] = cur[i] = pos; pos = align_up(pos + size[i], 64); }
    j->out_offsets[n] = pos; j->out_bytes = pos;
    std::vector<Segment> segs;
    for (uint32_t k = 0; k < nc; k++) {
        const cri_usm_chunk& c = chunks[k];
        if (memcmp(c.fourcc, "@SFA", 4) != 0 || c.type != 0) continue;
        const int i = item_of[c.chno & 255];
        Segment g; memset(&g, 0, sizeof g);
        g.src = c.payload_offset; g.dst = cur[i]; g.len = c.payload_len > c.padding ? c.payload_len - c.padding : 0;
        g.mask_begin = g.mask_end = 0xFFFFFFFFu;
        if (decrypt && codec_at[k] == CRI_USM_CODEC_ADX && c.payload_len > 0x140) {      // usm.py:313-322: whole 8-byte words after 0x140, padding included
            g.mask_begin = 0x140; g.mask_end = 0x140 + ((c.payload_len - 0x140) / 8) * 8;
        }
        cur[i] += g.len;
        j->alg_bytes += 2ull * g.len;
        if (g.len) segs.push_back(g);
    }
    j->units = segs.size();
    for (uint32_t i = 0; i < n; i++) { j->item_tags.push_back(chnos[i] | (codec[i] << 16)); j->item_sizes.push_back(size[i]); }
    uint8_t mask[32];
    cri_usm_audio_mask(key, mask);
    rc = upload_segments(j, segs, mask);
    if (rc) { delete j; return rc; }
    if ((rc = j->meta.commit())) { delete j; return rc; }
    *out = j;
    return 0;
}

// USMChunkHeader.pack(...) of an @SFA chunk (chunk.py:5, usm.py:599-613)
static void sfa_chunk_header(std::vector<uint8_t>& h, uint32_t size, uint32_t padding, uint32_t chno, uint32_t type, uint32_t frame_time, uint32_t frame_rate) {
    h.assign(0x20, 0);
    memcpy(h.data(), "@SFA", 4);
    h[4] = (uint8_t)(size >> 24); h[5] = (uint8_t)(size >> 16); h[6] = (uint8_t)(size >> 8); h[7] = (uint8_t)size;
    h[9] = 0x18; h[10] = (uint8_t)(padding >> 8); h[11] = (uint8_t)padding; h[12] = (uint8_t)chno; h[15] = (uint8_t)type;
    h[16] = (uint8_t)(frame_time >> 24); h[17] = (uint8_t)(frame_time >> 16); h[18] = (uint8_t)(frame_time >> 8); h[19] = (uint8_t)frame_time;
    h[20] = (uint8_t)(frame_rate >> 24); h[21] = (uint8_t)(frame_rate >> 16); h[22] = (uint8_t)(frame_rate >> 8); h[23] = (uint8_t)frame_rate;
}

extern "C" int cri_job_create_sfa_pack(const uint8_t* blob, const uint64_t* offsets, uint32_t n, uint32_t codec, uint64_t key,
                                       uint32_t encrypt_audio, cri_job** out) {
    if (!blob || !offsets || !out || (codec != CRI_USM_CODEC_ADX && codec != CRI_USM_CODEC_HCA)) return CRI_ERR_INVALID_ARG;
    const ItemSrc it = ItemSrc::from_blob(blob, offsets, n);
    if (!cri_device_available()) return CRI_ERR_HIP;
    cri_job* j = new_job(CRI_JOB_SFA_PACK, it);
    if (!j) return CRI_ERR_HIP;
    std::vector<Segment> segs;
    uint64_t out_pos = 0;
    std::vector<uint8_t> hdr;
    // one chunk: header image + payload segment (+ zero padding: the output is cleared by the header images' gaps -- see below)
    auto emit = [&](uint32_t item, uint64_t src, uint32_t len, uint32_t frame_time, bool mask) {
        const uint32_t padding = len % 0x20 ? 0x20 - len % 0x20 : 0;
        sfa_chunk_header(hdr, len + 0x18 + padding, padding, item, 0, frame_time, 2997);
        Image im; im.dst = out_pos; im.bytes = hdr;
        im.bytes.resize(0x20);
        j->images.push_back(im);
        Segment g; memset(&g, 0, sizeof g);
        g.src = src; g.dst = out_pos + 0x20; g.len = len; g.mask_begin = g.mask_end = 0xFFFFFFFFu;
        if (mask && len > 0x140) { g.mask_begin = 0x140; g.mask_end = len; }             // usm.py:1290-1300: every byte from 0x140 on
        if (len) segs.push_back(g);
        if (padding) { Image z; z.dst = out_pos + 0x20 + len; z.bytes.assign(padding, 0); j->images.push_back(z); }
        out_pos += 0x20 + len + padding;
        j->alg_bytes += 2ull * len;
    };
    auto contents_end = [&](uint32_t item) {                     // usm.py:640-657: type 2, appended to the stream's last chunk
        sfa_chunk_header(hdr, 0x38, 0, item, 2, 0, 30);
        Image im; im.dst = out_pos; im.bytes = hdr;
        static const char tail[] = "#CONTENTS END   ===============";
        im.bytes.insert(im.bytes.end(), tail, tail + 32);        // 31 characters + NUL
        j->images.push_back(im);
        out_pos += 0x40;
    };
    for (uint32_t i = 0; i < n; i++) {
        j->out_offsets[i] = out_pos;
        const uint8_t* d = it.ptr(i);
        const size_t len = it.len(i);
        uint32_t chunks = 0;
        if (codec == CRI_USM_CODEC_HCA) {                        // usm.py:659-716
            HcaHeader h;
            const uint32_t hs = len >= 8 ? (((uint32_t)d[6] << 8) | d[7]) : 0;
            int rc = hca_parse_header(d, len, hs, h);
            if (rc) { j->host_status[i] = rc; j->item_tags.push_back(0); continue; }
            emit(i, it.off(i), h.header_size, 0, false); chunks++;
            uint32_t t = 0;
            for (uint32_t f = 0; f < h.frame_count; f++) {       // hca.py:297-301 get_frames: FrameSize bytes each
                const uint64_t fo = (uint64_t)h.header_size + (uint64_t)f * h.frame_size;
                if (fo + h.frame_size > len) break;
                emit(i, it.off(i) + fo, h.frame_size, t, false); chunks++;
                t += 64;                                         // base_interval_per_SFA_chunk, usm.py:1177
            }
            contents_end(i);
        } else {                                                 // usm.py:584-657 (for an ADX file: filetype "adx")
            AdxHeader h;
            int rc = adx_parse_header(d, len, h);
            if (rc) { j->host_status[i] = rc; j->item_tags.push_back(0); continue; }
            const uint32_t cs = (uint32_t)((double)(uint32_t)((double)h.rate / 29.97) / 32.0);   // int(rate // 29.97 // 32)
            const uint32_t chunk = cs * (h.blocksize * h.channels);
            const uint32_t first = h.data_offset + 4;
            if (chunk == 0 || len < (size_t)h.blocksize + first) { j->host_status[i] = CRI_ERR_UNSUPPORTED; j->item_tags.push_back(0); continue; }
            const uint64_t stream_size = len - h.blocksize;
            uint64_t tell = 0; uint32_t count = 0, interval = 0;
            while (tell < stream_size) {
                uint32_t take;
                if (tell == 0) take = first;
                else if (tell + chunk > stream_size) {           // Python's floor modulo (usm.py:598): the operand is negative for a stream shorter than one chunk
                    const int64_t m = ((int64_t)stream_size - (int64_t)first - (int64_t)chunk) % (int64_t)chunk;
                    take = (uint32_t)(m < 0 ? m + (int64_t)chunk : m);
                } else take = chunk;
                take = (uint32_t)std::min<uint64_t>(take, len - tell);
                if (take == 0) break;
                emit(i, it.off(i) + tell, take, interval, encrypt_audio != 0); chunks++;
                tell += take;
                interval = (uint32_t)(int64_t)((double)count * 99.9);                    // usm.py:619, 1168 (the builder is VP9 only)
                count++;
            }
            emit(i, it.off(i) + tell, (uint32_t)std::min<uint64_t>(h.blocksize, len - tell), interval, false); chunks++;
            contents_end(i);
        }
        j->item_tags.push_back(chunks);
        j->units += chunks;
    }
    j->out_offsets[n] = out_pos; j->out_bytes = out_pos;
    for (uint32_t i = 0; i < n; i++) j->item_sizes.push_back(j->out_offsets[i + 1] - j->out_offsets[i]);
    uint8_t mask[32];
    cri_usm_audio_mask(key, mask);
    int rc = j->upload_images();
    if (!rc) rc = upload_segments(j, segs, mask);
    if (rc) { delete j; return rc; }
    if ((rc = j->meta.commit())) { delete j; return rc; }
    *out = j;
    return 0;
}

extern "C" const uint64_t* cri_job_item_sizes(const cri_job* j) { return j && j->item_sizes.size() == j->n && j->n ? j->item_sizes.data() : nullptr; }
extern "C" const uint32_t* cri_job_item_tags(const cri_job* j) { return j && j->item_tags.size() == j->n && j->n ? j->item_tags.data() : nullptr; }

// ------------------------------------------------------------------------------------------------ ADX encode
static int create_adx_encode(const ItemSrc& it, const cri_adx_encode_params* p, cri_job** out) {
    const uint32_t n = it.n;
    if (!it.ok() || !out || !p) return CRI_ERR_INVALID_ARG;
    if (!cri_device_available()) return CRI_ERR_HIP;
    cri_job* j = new_job(CRI_JOB_ADX_ENCODE, it);
    if (!j) return CRI_ERR_HIP;
    j->dominant = "k_adx_encode";
    std::vector<AdxStream> streams; std::vector<uint32_t> chain_stream; std::vector<int16_t> history; std::vector<uint8_t> stale;
    AdxWavePlan plan;
    bool all_std = true;
    uint64_t out_pos = 0;
    for (uint32_t i = 0; i < n; i++) {
        j->out_offsets[i] = out_pos;
        const uint8_t* d = it.ptr(i);
        size_t len = it.len(i);
        WavInfo w;
        int rc = wav_parse(d, len, w);
        if (rc) { j->host_status[i] = rc; continue; }
        AdxEncodePlan pl;
        rc = adx_plan_encode(d, len, w, p->bitdepth, p->blocksize, p->encoding_mode, p->highpass_frequency, p->filter, p->adx_version,
                             p->force_no_looping != 0, pl);
        if (rc) { j->host_status[i] = rc; continue; }
        if ((rc = wav_convertible(w))) { j->host_status[i] = rc; continue; }
        uint32_t bs = p->blocksize, hs = pl.header_size;
        Image head; head.dst = out_pos; head.bytes.assign(pl.image.begin(), pl.image.begin() + std::min<size_t>(hs, pl.image.size()));
        j->images.push_back(std::move(head));
        Image tail; tail.dst = out_pos + hs + (uint64_t)pl.frames * pl.channels * bs; tail.bytes.assign(bs, 0);   // adx.cpp:499-502
        { uint32_t v = (bs - 4) & 0xFFFF; uint8_t t4[4] = {0x80, 0x01, (uint8_t)(v >> 8), (uint8_t)v};
          for (uint32_t k = 0; k < 4 && k < bs; k++) tail.bytes[k] = t4[k]; }
        j->images.push_back(std::move(tail));
        AdxStream S; memset(&S, 0, sizeof S);
        S.src_offset = it.off(i) + w.data_offset; S.src_end = it.off(i) + it.len(i); S.dst_offset = out_pos + hs;
        S.frames = pl.frames; S.channels = pl.channels; S.blocksize = bs; S.bitdepth = p->bitdepth; S.mode = p->encoding_mode;
        S.samples_per_block = pl.samples_per_block; S.coef0 = pl.coef[0]; S.coef1 = pl.coef[1]; S.samples = pl.samples_per_channel;
        if (!plan.place(chain_stream, history, pl.channels, pl.channels * pl.samples_per_block * 2, pl.channels * bs)) {
            j->host_status[i] = CRI_ERR_UNSUPPORTED; j->images.pop_back(); j->images.pop_back(); continue;
        }
        if (!wav_is_pcm16(w)) { S.src_offset = j->add_convert(it.off(i) + w.data_offset, w); S.src_end = S.src_offset + 2ull * w.column_size; S.src_in_scratch = 1; }
        if (!(bs == 18 && p->bitdepth == 4 && pl.channels <= 2 && pl.image.size() <= hs + 1)) all_std = false;
        S.filter_bits = p->filter << 13; S.item = i; S.first_chain = (uint32_t)chain_stream.size(); S.hist_offset = S.first_chain;
        if (pl.image.size() > hs) { S.stale_offset = (uint32_t)stale.size(); S.stale_len = (uint32_t)(pl.image.size() - hs);
                                    stale.insert(stale.end(), pl.image.begin() + hs, pl.image.end()); }
        for (uint32_t c = 0; c < pl.channels; c++) {
            chain_stream.push_back((uint32_t)streams.size());
            history.push_back(pl.history[2 * c]); history.push_back(pl.history[2 * c + 1]);
        }
        streams.push_back(S);
        out_pos = align_up(out_pos + pl.total_size, 64);
        j->units += pl.frames; j->units2 += (uint64_t)pl.frames * pl.channels;
        j->alg_bytes += (uint64_t)pl.frames * pl.channels * (bs + 2ull * pl.samples_per_block);
    }
    j->out_offsets[n] = out_pos; j->out_bytes = out_pos;
    j->adx.chains = (uint32_t)chain_stream.size();
    plan.finish(j->adx);
    j->adx_streams = (uint32_t)streams.size();
    j->adx_wave_per_file = adx_pick_wave_per_file(all_std, streams.size(), true);
    if (j->adx_wave_per_file) j->dominant = "k_adx_encode_wpf";
    std::vector<uint32_t> seg_first;
    // Segmented chains.  Many files: a LANE per (file, channel, segment) -- no warm-up, every segment is encoded twice at its start
    // (k_adx_lane_encode); few files: a WAVE per (file, segment) on the wave-per-file encoder (k_adx_seg_encode).
    // CRICODECS_ADX_MAPPING = "lane" / "wave" forces one of the two where segments apply.
    const int map_knob = knobs().adx_mapping;
    // measured (tools/debug/adx_seg_sweep.py): 10 s files -- wave 1.7 / 3.2 / 5.4 / 12.3 ms at 1 / 128 / 256 / 1000 files, lane 3.8 / 4.7 /
    // 4.8 / 5.2 ms; 1 s files -- wave 1.4 / 2.1 / 4.6 ms at 64 / 1000 / 4000 files, lane 3.4 / 3.7 / 3.9 ms (6.0 at 20 000): the lane form
    // has a floor of some 3.5 ms (one segment's rows plus the repair, a row at a time) and takes over from about 8 M blocks
    uint64_t blocks_total = 0;
    for (const AdxStream& S : streams) blocks_total += (uint64_t)S.frames * S.channels;
    const bool want_lane = all_std && !streams.empty() && !(map_knob == ADX_MAP_CHAIN || map_knob == ADX_MAP_FILE || map_knob == ADX_MAP_WAVE || map_knob == ADX_MAP_SEG) &&
                           (blocks_total >= 8000000ull || map_knob == ADX_MAP_LANE);
    if (want_lane) {
        uint64_t chains_total = 0;
        for (const AdxStream& S : streams) chains_total += S.channels;
        const uint64_t pct = knobs().adx_warm_pct;
        const uint64_t p_target = std::max<uint64_t>(1, 262144 / chains_total);
        const uint64_t seg_pct = knobs().adx_seglen ? knobs().adx_seglen : 50;      // a segment's least length, in percent of the warm-up
        // Two regimes.  Lanes enough at long segments (1024 rows: longer than the encoder's merge time; 1.5 waves per SIMD and more): no
        // warm-up, every segment is encoded from the raw samples before it and repaired where that was wrong -- the least work.  Fewer
        // lanes (1000 files of 10 s): the kernel is bound by the latency of one row after the other in a lane, ~3 us, and a lane's rows are
        // what counts -- a warm-up of 640 rows (the merge time) before segments of 320 gives three times the lanes, each with 960 rows
        // instead of 1024 + up to 600 of repair, at 2.4 times the instructions (4.45 -> 3.4 ms; 4000 files of 10 s would go 8.7 -> 9.2).
        bool short_segments = true;
        uint64_t lanes_long = 0, rows_long = 0;
        for (const AdxStream& S : streams) {
            const int64_t g = S.mode == 2 ? 64 : 4096 - (int64_t)S.coef0 - (int64_t)S.coef1;
            if (g <= 0 || !S.frames) { lanes_long += S.channels; rows_long = std::max<uint64_t>(rows_long, S.frames); continue; }
            const uint64_t rows = std::min<uint64_t>(S.frames, std::max<uint64_t>((S.frames + p_target - 1) / p_target, std::max<uint64_t>(4, 1024ull * 39 * pct / 100 / (uint64_t)g)));
            lanes_long += (uint64_t)S.channels * ((S.frames + rows - 1) / rows);
            rows_long = std::max<uint64_t>(rows_long, rows + (rows < S.frames ? 600 : 0));      // (a cut file without warm-up: the repair rounds re-encode up to the merge time)
        }
        auto warm_short = [&](int64_t g) { return std::max<uint64_t>(4, (640ull * 39 * pct / 100 / (uint64_t)g + 3) / 4 * 4); };
        // The least segment length.  With a warm-up the kernel's time is (waves per SIMD, rounded up) x (warm-up + segment): the planner
        // tries segment lengths from half a warm-up to the whole file and keeps the cheapest -- 1000 files of 10 s: 480 rows (1000 waves on
        // 1024 SIMDs, 2.2 ms) instead of 320 (1469 waves: every SIMD with two of them sets the pace, 3.4 ms); 12 500 files of 1 s: no cut at
        // all (391 waves of 1500 rows, 2.8 ms against 4.4).  (adx_seglen, tests only, fixes it instead.)
        uint64_t best_pct = seg_pct, best_cost = ~0ull;
        const uint32_t nsimd = device_simds();
        if (!knobs().adx_seglen) {
            for (uint64_t cand = 50; cand <= 6400; cand = cand * 9 / 8 + 1) {
                uint64_t lanes_c = 0, longest = 0;
                for (const AdxStream& S : streams) {
                    const int64_t g = S.mode == 2 ? 64 : 4096 - (int64_t)S.coef0 - (int64_t)S.coef1;
                    if (g <= 0 || !S.frames) { lanes_c += S.channels; longest = std::max<uint64_t>(longest, S.frames); continue; }
                    const uint64_t warm = warm_short(g), lmin = std::max<uint64_t>(4, (warm * cand / 100 + 3) / 4 * 4);
                    uint64_t rows = (std::max<uint64_t>((S.frames + p_target - 1) / p_target, lmin) + 3) / 4 * 4;
                    if (rows > S.frames) rows = S.frames;
                    lanes_c += (uint64_t)S.channels * ((S.frames + rows - 1) / rows);
                    // (a cut file pays the warm-up and, in the repair rounds, about as much again: some lane of nearly every wave has to
                    //  re-encode until it meets its checkpoints -- measured: 12 500 files of 1 s in two segments 4.7 ms, uncut 2.8)
                    longest = std::max<uint64_t>(longest, rows + (rows < S.frames ? 2 * warm : 0));
                }
                const uint64_t waves = (lanes_c + 63) / 64, cost = ((waves + nsimd - 1) / nsimd) * longest;
                if (cost < best_cost) { best_cost = cost; best_pct = cand; }
            }
        }
        // The long-segment regime is taken when the job has lanes enough for it AND the same cost model does not price the best short plan
        // (often: no cut at all) below it.  Round 4 switched on the lane count alone, and 25 000-32 000 files of 1 s -- two waves per SIMD
        // of 1024-row segments plus repairs, where 782-1000 uncut waves of 1500 rows fit one to a SIMD -- took 4.8-5.0 ms instead of 2.9
        // (the "cliff" between 24 000 and 26 000 files that DESIGN r4 put down to the memory system: it was this threshold).
        if (lanes_long >= 98304) {
            const uint64_t waves_long = (lanes_long + 63) / 64, cost_long = ((waves_long + nsimd - 1) / nsimd) * rows_long;
            short_segments = !knobs().adx_seglen && best_cost <= cost_long;
        }
        auto lane_warm = [&](int64_t g) { return short_segments ? warm_short(g) : (uint64_t)0; };
        auto lane_lmin = [&](int64_t g) { return short_segments ? std::max<uint64_t>(4, (lane_warm(g) * best_pct / 100 + 3) / 4 * 4)
                                                                 : std::max<uint64_t>(4, (1024ull * 39 * pct / 100 / (uint64_t)g + 3) / 4 * 4); };
        uint32_t lanes = 0, chains = 0; uint64_t rounds = 0;
        std::vector<int16_t> hist2;
        bool usable = true;
        // Lanes in order of decreasing length: the 64 lanes of a wave run as long as its longest segment, and a bank of ragged clips
        // in item order pairs 0.05 s clips with 2 s ones (the segmented decoder sorts the same way).  `first_chain` still names the
        // file's entries in `history`, which is re-packed below in the new order.
        {
            auto seg_len = [&](const AdxStream& S) {
                const int64_t g = S.mode == 2 ? 64 : 4096 - (int64_t)S.coef0 - (int64_t)S.coef1;
                if (g <= 0 || !S.frames) return (uint64_t)S.frames;
                const uint64_t warm = lane_warm(g), lmin = lane_lmin(g);
                const uint64_t rows = (std::max<uint64_t>((S.frames + p_target - 1) / p_target, lmin) + 3) / 4 * 4;
                return std::min<uint64_t>(rows, S.frames) + (rows < S.frames ? warm : 0);      // (a lane's rows: warm-up + segment)
            };
            bool wide = false;
            for (const AdxStream& S : streams) wide = wide || S.channels > 2;
            if (!wide) std::stable_sort(streams.begin(), streams.end(), [&](const AdxStream& x, const AdxStream& y) { return seg_len(x) > seg_len(y); });
        }
        for (AdxStream& S : streams) {
            const int64_t g = S.mode == 2 ? 64 : 4096 - (int64_t)S.coef0 - (int64_t)S.coef1;
            // the warm-up covers the encoder's merge time (600 rows at worst for tonal material at g = 39; sparse material takes up to 2200:
            // its segments are repaired by the rounds); a segment is half a warm-up long unless the job has lanes enough with longer ones
            uint64_t rows = S.frames, warm = 0;
            if (g > 0 && S.frames) {
                warm = lane_warm(g);
                const uint64_t lmin = lane_lmin(g);
                rows = std::max<uint64_t>((S.frames + p_target - 1) / p_target, lmin);
                rows = (rows + 3) / 4 * 4;
                if (rows > S.frames) rows = S.frames;
            }
            if (S.channels > 2) usable = false;
            S.seg_rows = (uint32_t)rows; S.seg_count = S.frames ? (uint32_t)((S.frames + rows - 1) / rows) : 0; S.warm_rows = (uint32_t)warm;
            if (S.channels == 2 && (lanes & 1)) lanes++;
            if (S.channels == 2 && (chains & 1)) { chains++; hist2.push_back(0); hist2.push_back(0); }
            S.first_seg = lanes; S.rows_avail = (uint32_t)rounds;
            seg_first.push_back(lanes);
            for (uint32_t c = 0; c < S.channels; c++) { hist2.push_back(history[2 * (S.first_chain + c)]); hist2.push_back(history[2 * (S.first_chain + c) + 1]); }
            S.first_chain = chains; S.hist_offset = chains;
            lanes += S.seg_count * S.channels; chains += S.channels; rounds += (S.frames + 3) / 4;
        }
        seg_first.push_back(lanes);
        if (usable) {
            j->adx_seg = true; j->adx_lane = true; j->adx_wave_per_file = false; j->dominant = "k_adx_lane_encode";
            history.swap(hist2);
            j->adx.chains = chains; j->adx.seg_lanes = lanes; j->adx.n_streams = (uint32_t)streams.size();
            j->adx_seg_state_offset = align_up(j->scratch_bytes, 256);
            j->adx_seg_ckpt_offset = j->adx_seg_state_offset + align_up(16ull * lanes, 256);
            j->adx_seg_flags_offset = j->adx_seg_ckpt_offset + align_up(8ull * rounds, 256);
            j->scratch_bytes = j->adx_seg_flags_offset + align_up(4ull * streams.size(), 256);
        } else seg_first.clear();
    }
    if (!j->adx_lane && all_std && !streams.empty() && adx_plan_segments(streams, true)) {
        // segmented chains on the wave-per-file kernel: a workgroup per (file, segment)
        j->adx_seg = true; j->adx_wave_per_file = false; j->dominant = "k_adx_seg_encode";
        uint32_t segs = 0; uint64_t rounds = 0;
        for (AdxStream& S : streams) {
            S.first_seg = segs; S.rows_avail = (uint32_t)rounds;      // (encode: the stream's first checkpoint)
            seg_first.push_back(segs);
            segs += S.seg_count; rounds += (S.frames + 3) / 4;
        }
        seg_first.push_back(segs);
        j->adx.seg_lanes = segs; j->adx.n_streams = (uint32_t)streams.size();
        j->adx_seg_state_offset = align_up(j->scratch_bytes, 256);
        j->adx_seg_ckpt_offset = j->adx_seg_state_offset + align_up(32ull * segs, 256);
        j->adx_seg_flags_offset = j->adx_seg_ckpt_offset + align_up(8ull * rounds, 256);
        j->scratch_bytes = j->adx_seg_flags_offset + align_up(4ull * streams.size(), 256);
    }
    const std::vector<uint32_t> order = j->adx_wave_per_file ? adx_longest_first(streams) : std::vector<uint32_t>();
    if (streams.empty()) { AdxStream S; memset(&S, 0, sizeof S); streams.push_back(S); }
    if (chain_stream.empty()) { chain_stream.push_back(0xFFFFFFFFu); history.assign(2, 0); }
    if (stale.empty()) stale.push_back(0);
    int rc = 0;
    if ((rc = j->d_adx_streams.upload(streams)) || (rc = j->d_chain_stream.upload(chain_stream)) || (rc = j->d_history.upload(history)) ||
        (!order.empty() && !j->adx_seg && (rc = j->d_adx_order.upload(order))) || (!seg_first.empty() && (rc = j->d_seg_chain.upload(seg_first))) ||
        (rc = j->d_stale.upload(stale)) || (rc = j->upload_images()) || (rc = j->upload_convert())) { delete j; return rc; }
    if ((rc = j->meta.commit())) { delete j; return rc; }
    *out = j;
    return 0;
}

extern "C" int cri_job_create_adx_encode(const uint8_t* blob, const uint64_t* offsets, uint32_t n, const cri_adx_encode_params* p, cri_job** out) {
    if (!blob || !offsets) return CRI_ERR_INVALID_ARG;
    return create_adx_encode(ItemSrc::from_blob(blob, offsets, n), p, out);
}
extern "C" int cri_job_create_adx_encode_items(const cri_items* items, const cri_adx_encode_params* p, cri_job** out) {
    ItemSrc it; int rc = ItemSrc::from_items(items, it);
    return rc ? rc : create_adx_encode(it, p, out);
}

// ------------------------------------------------------------------------------------------------ HCA crypt
static uint16_t crc_xpow_bytes(uint32_t nbytes);
static int create_hca_crypt(const ItemSrc& it, uint32_t encrypt, uint32_t type, const uint64_t* keys,
                            const uint16_t* subkeys, const uint32_t* header_sizes, cri_job** out) {
    const uint32_t n = it.n;
    if (!it.ok() || !out) return CRI_ERR_INVALID_ARG;
    if (!cri_device_available()) return CRI_ERR_HIP;
    cri_job* j = new_job(CRI_JOB_HCA_CRYPT, it);
    if (!j) return CRI_ERR_HIP;
    j->dominant = "k_hca_crypt";
    std::vector<HcaStream> streams; std::vector<uint32_t> frame_sizes, first_frame{0}; std::vector<uint8_t> cipher;
    std::map<std::tuple<uint32_t, uint64_t, uint32_t>, uint32_t> cipher_index;
    uint64_t out_pos = 0; uint32_t frames_total = 0;
    for (uint32_t i = 0; i < n; i++) {
        j->out_offsets[i] = out_pos;
        const uint8_t* d = it.ptr(i);
        size_t len = it.len(i);
        HcaHeader h;
        uint32_t hs = header_sizes ? header_sizes[i] : (len >= 8 ? be16(d + 6) : 0);
        int rc = hca_parse_header(d, len, hs, h);
        if (rc) { j->host_status[i] = CRI_ERR_HCA_HEADER; continue; }
        if ((uint64_t)hs + (uint64_t)h.frame_count * h.frame_size > len || hs < 8) { j->host_status[i] = CRI_ERR_HCA_HEADER; continue; }
        uint32_t ctype = encrypt == 1 ? type : h.ciph_type;
        uint64_t mixed = hca_mix_key(keys ? keys[i] : 0, subkeys ? subkeys[i] : 0);
        if (ctype == 56 && !mixed) ctype = 0;
        if (ctype != 56) mixed = 0;
        uint8_t t[256];
        if (hca_cipher_table(ctype, mixed, t)) { j->host_status[i] = CRI_ERR_HCA_HEADER; continue; }
        auto ck = std::make_tuple(ctype, mixed, encrypt ? 1u : 0u);
        auto ci = cipher_index.find(ck);
        uint32_t cidx;
        if (ci == cipher_index.end()) {
            if (encrypt) { uint8_t inv[256]; for (int k = 0; k < 256; k++) inv[t[k]] = (uint8_t)k; memcpy(t, inv, 256); }   // hca.cpp:3315-3320
            cidx = (uint32_t)(cipher.size() / 256); cipher.insert(cipher.end(), t, t + 256); cipher_index[ck] = cidx;
        } else cidx = ci->second;
        Image head; head.dst = out_pos; head.bytes.assign(d, d + hs);
        hca_crypt_header(head.bytes.data(), hs, encrypt, type);
        j->images.push_back(std::move(head));
        uint64_t body_end = (uint64_t)hs + (uint64_t)h.frame_count * h.frame_size;
        if (body_end < len) { Image tail; tail.dst = out_pos + body_end; tail.bytes.assign(d + body_end, d + len); j->images.push_back(std::move(tail)); }
        HcaStream S; memset(&S, 0, sizeof S);
        S.src_offset = it.off(i) + hs; S.dst_offset = out_pos + hs; S.cipher = cidx; S.frames = h.frame_count; S.item = i;
        streams.push_back(S); frame_sizes.push_back(h.frame_size);
        frames_total += h.frame_count; first_frame.push_back(frames_total);
        out_pos = align_up(out_pos + len, 64);
        j->units += h.frame_count; j->alg_bytes += 2ull * h.frame_count * h.frame_size;
    }
    j->out_offsets[n] = out_pos; j->out_bytes = out_pos;
    j->crypt.n_streams = (uint32_t)streams.size(); j->crypt.frames = frames_total;
    j->crypt.max_frame_size = 0;
    for (uint32_t fsz : frame_sizes) j->crypt.max_frame_size = std::max(j->crypt.max_frame_size, fsz);
    if (streams.empty()) { HcaStream S; memset(&S, 0, sizeof S); streams.push_back(S); frame_sizes.push_back(8); }
    if (cipher.empty()) cipher.assign(256, 0);
    // checksum tables of the wave-per-frame kernel, one per distinct frame size: a lane's chunk remainder times its place in the frame
    std::vector<uint16_t> crcpos; std::vector<uint32_t> crc_off; std::map<uint32_t, uint32_t> crc_by_size;
    for (uint32_t fsz : frame_sizes) {
        auto it2 = crc_by_size.find(fsz);
        if (it2 == crc_by_size.end()) {
            it2 = crc_by_size.emplace(fsz, (uint32_t)crcpos.size()).first;
            const uint32_t m = fsz >= 2 ? (fsz - 2 + 63) / 64 : 0;
            for (uint32_t l = 0; l < 64; l++) {
                uint32_t v = crc_xpow_bytes(m * (63 - l));
                for (uint32_t bit = 0; bit < 16; bit++) { crcpos.push_back((uint16_t)v); v = ((v << 1) ^ ((v & 0x8000) ? 0x8005u : 0u)) & 0xFFFF; }
            }
        }
        crc_off.push_back(it2->second);
    }
    int rc = 0;
    if ((rc = j->d_streams.upload(streams)) || (rc = j->d_frame_sizes.upload(frame_sizes)) || (rc = j->d_first_frame.upload(first_frame)) ||
        (rc = j->d_crcmul.upload(crcpos)) || (rc = j->d_crc_off.upload(crc_off)) ||
        (rc = j->d_cipher.upload(cipher)) || (rc = j->upload_images())) { delete j; return rc; }
    if ((rc = j->meta.commit())) { delete j; return rc; }
    *out = j;
    return 0;
}
extern "C" int cri_job_create_hca_crypt(const uint8_t* blob, const uint64_t* offsets, uint32_t n, uint32_t encrypt, uint32_t type,
                                        const uint64_t* keys, const uint16_t* subkeys, cri_job** job) {
    if (!blob || !offsets) return CRI_ERR_INVALID_ARG;
    return create_hca_crypt(ItemSrc::from_blob(blob, offsets, n), encrypt, type, keys, subkeys, nullptr, job);
}
extern "C" int cri_job_create_hca_crypt_items(const cri_items* items, uint32_t encrypt, uint32_t type, const uint64_t* keys, const uint16_t* subkeys, cri_job** job) {
    ItemSrc it; int rc = ItemSrc::from_items(items, it);
    return rc ? rc : create_hca_crypt(it, encrypt, type, keys, subkeys, nullptr, job);
}

// ------------------------------------------------------------------------------------------------ HCA encode
static uint16_t crc_xpow_bytes(uint32_t nbytes) {            // x^(8*nbytes) mod P (P = x^16 + x^15 + x^2 + 1)
    uint32_t v = 1;
    for (uint64_t i = 0; i < (uint64_t)nbytes * 8; i++) v = ((v << 1) ^ ((v & 0x8000) ? 0x8005u : 0u)) & 0xFFFF;
    return (uint16_t)v;
}

static int create_hca_encode(const ItemSrc& it, uint32_t force_no_looping, uint32_t quality, cri_job** out) {
    const uint32_t n = it.n;
    if (!it.ok() || !out) return CRI_ERR_INVALID_ARG;
    if (!cri_device_available()) return CRI_ERR_HIP;
    cri_job* j = new_job(CRI_JOB_HCA_ENCODE, it);
    if (!j) return CRI_ERR_HIP;
    j->dominant = "k_hca_encode";
    std::vector<HcaFormat> formats; std::vector<HcaStream> streams;
    std::map<std::vector<uint32_t>, uint32_t> fmt_index;
    uint64_t out_pos = 0;
    for (uint32_t i = 0; i < n; i++) {
        j->out_offsets[i] = out_pos;
        const uint8_t* d = it.ptr(i);
        size_t len = it.len(i);
        WavInfo w;
        int rc = wav_parse(d, len, w);
        if (rc) { j->host_status[i] = rc; continue; }
        const bool looping = w.looping && !force_no_looping;
        if (looping && w.num_loops == 0) { j->host_status[i] = CRI_ERR_UNSUPPORTED; continue; }   // the reference reads an empty loop array here
        if ((rc = wav_convertible(w))) { j->host_status[i] = rc; continue; }
        HcaEncSetup e;
        rc = hca_enc_setup(w.channels, w.rate, w.column_size / w.channels, quality, e);
        if (rc) { j->host_status[i] = rc; continue; }
        if (looping) hca_enc_setup_loop(e, w.loop_start[0], w.loop_end[0], w.column_size);
        std::vector<uint32_t> key = {e.channels, e.frame_size, e.total_bands, e.base_bands, e.stereo_bands, e.hfr_group_count,
                                     e.bands_per_hfr_group, e.hfr_band_count, e.channel_config};
        auto fi = fmt_index.find(key);
        uint32_t fidx;
        if (fi == fmt_index.end()) {
            HcaFormat F; memset(&F, 0, sizeof F);
            F.channels = e.channels; F.version = 0x0200; F.frame_size = e.frame_size; F.min_res = 1; F.max_res = 15;
            F.total_bands = e.total_bands; F.base_bands = e.base_bands; F.stereo_bands = e.stereo_bands;
            F.bands_per_hfr_group = e.bands_per_hfr_group; F.hfr_group_count = e.hfr_group_count; F.hfr_band_count = e.hfr_band_count;
            for (uint32_t c = 0; c < 16; c++) { F.type[c] = e.type[c]; F.coded[c] = (uint8_t)e.coded[c]; }
            fidx = (uint32_t)formats.size(); formats.push_back(F); fmt_index[key] = fidx;
        } else fidx = fi->second;
        Image head; head.dst = out_pos; head.bytes.assign(e.header_size, 0);
        hca_pack_header(e, head.bytes.data());
        j->images.push_back(std::move(head));
        HcaStream S; memset(&S, 0, sizeof S);
        S.src_offset = it.off(i) + w.data_offset; S.dst_offset = out_pos + e.header_size; S.format = fidx; S.frames = e.frame_count;
        S.samples = e.samples_per_channel; S.item = i;
        if (looping) {
            const uint32_t have = w.column_size / w.channels;
            const uint32_t seen_end = e.samples_per_channel == 0 ? 1024 : ((e.samples_per_channel - 1) / 1024 + 1) * 1024;   // chunks SaveLoopAudio saw
            S.enc_loop = 1; S.enc_pre = e.pre_samples; S.enc_pre_zero = e.pre_samples > 1024 ? ((e.pre_samples - 1) / 1024) * 1024 : 0;
            S.enc_post = e.post_samples; S.enc_loop_src = e.loop_start; S.enc_loop_src_end = std::min(seen_end, have); S.enc_have = have;
        }
        if (!wav_is_pcm16(w)) { S.src_offset = j->add_convert(it.off(i) + w.data_offset, w); S.src_in_scratch = 1; }
        streams.push_back(S);
        out_pos = align_up(out_pos + e.header_size + (uint64_t)e.frame_count * e.frame_size, 64);
        j->units += e.frame_count;
        j->alg_bytes += (uint64_t)e.frame_count * (e.frame_size + 2048ull * e.channels);
    }
    j->out_offsets[n] = out_pos; j->out_bytes = out_pos;
    std::stable_sort(streams.begin(), streams.end(), [](const HcaStream& a, const HcaStream& b) { return a.format < b.format; });
    std::vector<uint16_t> crcmul;
    std::vector<uint32_t> enc_hint;
    for (size_t b = 0; b < streams.size();) {
        size_t e = b; uint32_t frames = 0;
        const HcaFormat& F = formats[streams[b].format];
        while (e < streams.size() && streams[e].format == streams[b].format) { streams[e].first_frame = frames; frames += streams[e].frames; e++; }
        HcaEncArgs a; memset(&a, 0, sizeof a);
        a.format = streams[b].format; a.stream_begin = (uint32_t)b; a.stream_end = (uint32_t)e; a.frames = frames;
        a.channels = F.channels; a.frame_size = F.frame_size; a.crc_chunk = 4 * ((F.frame_size - 2 + 255) / 256);   // whole words per lane
        // frame -> stream without a binary search over the whole table (fourteen dependent loads at the head of every workgroup of a
        // 10 000-file job): the stream of every 16th frame of the launch; a workgroup walks on from there (a step or two)
        j->hca_enc_hint_off.push_back((uint32_t)enc_hint.size());
        for (uint32_t g = 0, sx = (uint32_t)b; g < frames; g += 16) {
            while (sx + 1 < e && streams[sx + 1].first_frame <= g) sx++;
            enc_hint.push_back(sx);
        }
        j->hca_enc_crc_off.push_back((uint32_t)crcmul.size());
        for (uint32_t l = 0; l < 64; l++) {
            uint32_t v = crc_xpow_bytes(a.crc_chunk * (63 - l) + 2);   // (+ 2 bytes: the checksum is the remainder of message * x^16)
            for (uint32_t bit = 0; bit < 16; bit++) { crcmul.push_back((uint16_t)v); v = ((v << 1) ^ ((v & 0x8000) ? 0x8005u : 0u)) & 0xFFFF; }
        }
        if (hca_encode_lds_bytes(F.channels, F.frame_size) > 160 * 1024) {
            for (size_t s = b; s < e; s++) j->host_status[streams[s].item] = CRI_ERR_UNSUPPORTED;
            a.frames = 0;
        }
        j->hca_enc.push_back(a);
        b = e;
    }
    if (formats.empty()) { HcaFormat F; memset(&F, 0, sizeof F); formats.push_back(F); }
    if (streams.empty()) { HcaStream S; memset(&S, 0, sizeof S); streams.push_back(S); }
    if (crcmul.empty()) crcmul.assign(1024, 0);
    if (enc_hint.empty()) enc_hint.push_back(0);
    static std::vector<uint8_t> enctab;                       // the same for every job: built once
    static std::once_flag enctab_once; static int enctab_rc = 0;
    std::call_once(enctab_once, [] { enctab_rc = hca_enc_build_tables(enctab); });
    int rc = enctab_rc;
    if (rc) { delete j; return rc; }
    if ((rc = j->d_enctab.upload(enctab)) || (rc = j->d_formats.upload(formats)) || (rc = j->d_streams.upload(streams)) || (rc = j->d_crcmul.upload(crcmul)) || (rc = j->d_enc_hint.upload(enc_hint)) || (rc = j->upload_images()) || (rc = j->upload_convert())) { delete j; return rc; }
    if ((rc = j->meta.commit())) { delete j; return rc; }
    *out = j;
    return 0;
}

extern "C" int cri_job_create_hca_encode(const uint8_t* blob, const uint64_t* offsets, uint32_t n, uint32_t force_no_looping, uint32_t quality, cri_job** out) {
    if (!blob || !offsets) return CRI_ERR_INVALID_ARG;
    return create_hca_encode(ItemSrc::from_blob(blob, offsets, n), force_no_looping, quality, out);
}
extern "C" int cri_job_create_hca_encode_items(const cri_items* items, uint32_t force_no_looping, uint32_t quality, cri_job** out) {
    ItemSrc it; int rc = ItemSrc::from_items(items, it);
    return rc ? rc : create_hca_encode(it, force_no_looping, quality, out);
}

// ------------------------------------------------------------------------------------------------ run
static int job_run(cri_job* j, const void* d_in, void* d_out, void* d_scratch, int32_t* d_status, float* d_floats, void* hip_stream) {
    if (!j || !d_in || (!d_out && j->out_bytes)) return CRI_ERR_INVALID_ARG;
    if (j->scratch_bytes && !d_scratch) return CRI_ERR_INVALID_ARG;
    DeviceGuard guard(j->device);                // the caller's buffers and stream must belong to the job's device
    if (!guard.ok()) return CRI_ERR_HIP;
    hipStream_t s = (hipStream_t)hip_stream;
    if (j->events_on) j->begin_run();
    if (d_status) launch_fill_i32(d_status, 0, j->n, s);
    if (j->n_images)
        launch_scatter_images((const uint8_t*)j->d_img.p, (const uint64_t*)j->d_img_off.p, (const uint64_t*)j->d_img_dst.p, j->n_images, (uint8_t*)d_out, s);
    if (j->convert_total) {
        ConvertArgs c; c.in = (const uint8_t*)d_in; c.scratch = (uint8_t*)d_scratch; c.items = (const ConvertItem*)j->d_convert.p;
        c.n_items = (uint32_t)j->convert.size(); c.total = j->convert_total;
        launch_pcm_convert(c, s);
    }
    switch (j->kind) {
        case CRI_JOB_HCA_DECODE:
            for (auto a : j->hca_dec) {
                a.in = (const uint8_t*)d_in; a.out = (uint8_t*)d_out; a.scratch = (uint8_t*)d_scratch; a.status = d_status;
                a.formats = (const HcaFormat*)j->d_formats.p; a.streams = (const HcaStream*)j->d_streams.p;
                a.cipher_tables = (const uint8_t*)j->d_cipher.p; a.ath_tables = (const uint8_t*)j->d_ath.p;
                a.float_out = d_floats;
                j->mark(0, true, s); launch_hca_parse(a, s); j->mark(0, false, s);
                j->mark(1, true, s); launch_hca_transform(a, s); j->mark(1, false, s);
            }
            break;
        case CRI_JOB_ADX_DECODE:
        case CRI_JOB_ADX_ENCODE: {
            AdxArgs a = j->adx;
            a.in = (const uint8_t*)d_in; a.out = (uint8_t*)d_out; a.status = d_status; a.scratch = (const uint8_t*)d_scratch;
            a.streams = (const AdxStream*)j->d_adx_streams.p; a.chain_stream = (const uint32_t*)j->d_chain_stream.p;
            a.history = (const int16_t*)j->d_history.p; a.stale = (const uint8_t*)j->d_stale.p; a.wpf_order = (const uint32_t*)j->d_adx_order.p;
            if (j->adx_seg && j->kind == CRI_JOB_ADX_ENCODE) {
                uint8_t* sc = (uint8_t*)d_scratch;
                a.seg_first = (const uint32_t*)j->d_seg_chain.p; a.seg_state = (uint32_t*)(sc + j->adx_seg_state_offset);
                a.seg_ckpt = (uint32_t*)(sc + j->adx_seg_ckpt_offset); a.seg_flags = (uint32_t*)(sc + j->adx_seg_flags_offset);
                j->mark(0, true, s);
                launch_fill_i32((int32_t*)a.seg_flags, 0, a.n_streams, s);
                if (j->adx_lane) launch_adx_encode_lane(a, s); else launch_adx_encode_seg(a, s);
                j->mark(0, false, s);
                break;
            }
            if (j->adx_seg) {
                a.seg_first = (const uint32_t*)j->d_seg_chain.p; a.seg_state = (uint32_t*)d_scratch;
                a.seg_flags = (uint32_t*)((uint8_t*)d_scratch + j->adx_seg_flags_offset);
                j->mark(0, true, s);
                launch_fill_i32((int32_t*)a.seg_flags, 0, a.chains + 1, s);
                launch_adx_decode_seg(a, s);
                j->mark(0, false, s);
                break;
            }
            j->mark(0, true, s);
            if (j->adx_wave_per_file) { if (j->kind == CRI_JOB_ADX_DECODE) launch_adx_decode_wpf(a, j->adx_streams, s); else launch_adx_encode_wpf(a, j->adx_streams, s); }
            else if (j->kind == CRI_JOB_ADX_DECODE) launch_adx_decode(a, s); else launch_adx_encode(a, s);
            j->mark(0, false, s);
            break;
        }
        case CRI_JOB_HCA_ENCODE:
            for (size_t k = 0; k < j->hca_enc.size(); k++) {
                HcaEncArgs a = j->hca_enc[k];
                a.in = (const uint8_t*)d_in; a.out = (uint8_t*)d_out; a.status = d_status; a.scratch = (const uint8_t*)d_scratch;
                a.formats = (const HcaFormat*)j->d_formats.p; a.streams = (const HcaStream*)j->d_streams.p;
                a.crc_mul = (const uint16_t*)j->d_crcmul.p + j->hca_enc_crc_off[k];
                a.stream_hint = (const uint32_t*)j->d_enc_hint.p + j->hca_enc_hint_off[k];
                a.tables = (const uint8_t*)j->d_enctab.p;
                j->mark(0, true, s); launch_hca_encode(a, s); j->mark(0, false, s);
            }
            break;
        case CRI_JOB_HCA_CRYPT: {
            CryptArgs a = j->crypt;
            a.in = (const uint8_t*)d_in; a.out = (uint8_t*)d_out; a.streams = (const HcaStream*)j->d_streams.p;
            a.frame_sizes = (const uint32_t*)j->d_frame_sizes.p; a.cipher_tables = (const uint8_t*)j->d_cipher.p;
            a.first_frame = (const uint32_t*)j->d_first_frame.p;
            a.crc_pos = (const uint16_t*)j->d_crcmul.p; a.crc_pos_off = (const uint32_t*)j->d_crc_off.p;
            j->mark(0, true, s); launch_hca_crypt(a, s); j->mark(0, false, s);
            break;
        }
        case CRI_JOB_USM_DEMUX:
        case CRI_JOB_SFA_PACK: {
            SegmentArgs a = j->seg;
            a.in = (const uint8_t*)d_in; a.out = (uint8_t*)d_out; a.segs = (const Segment*)j->d_segs.p;
            j->mark(0, true, s); launch_segments(a, s); j->mark(0, false, s);
            break;
        }
        default: return CRI_ERR_UNSUPPORTED;
    }
#ifdef CRI_TESTING
    if (knobs().bad_launch) launch_fill_i32_bad((int32_t*)d_out, s);
#endif
    const hipError_t launched = hipGetLastError();   // the launches' own verdict, read before anything else can clear it
    j->note_run(s);
    return launched == hipSuccess ? 0 : CRI_ERR_HIP;
}

extern "C" int cri_job_run(cri_job* j, const void* d_in, void* d_out, void* d_scratch, int32_t* d_status, void* hip_stream) {
    return job_run(j, d_in, d_out, d_scratch, d_status, nullptr, hip_stream);
}

// Validation run of an HCA decode job: the same kernels, in instances that also store wave[][] (the floats before the int16
// conversion, hca.cpp:1987-1992 -> 339-360) so that a test can hold the device to north_star's float tolerance.
extern "C" int cri_job_run_floats(cri_job* j, const void* d_in, void* d_out, void* d_scratch, int32_t* d_status, float* d_floats, void* hip_stream) {
    if (!j || j->kind != CRI_JOB_HCA_DECODE || !d_floats) return CRI_ERR_INVALID_ARG;
    return job_run(j, d_in, d_out, d_scratch, d_status, d_floats, hip_stream);
}
extern "C" uint64_t cri_job_float_count(const cri_job* j) { return j && !j->float_offsets.empty() ? j->float_offsets.back() : 0; }
extern "C" const uint64_t* cri_job_float_offsets(const cri_job* j) { return j && !j->float_offsets.empty() ? j->float_offsets.data() : nullptr; }

// Layout of an HCA decode job's frame records in scratch, per format group (= per launch set): lets a caller take a census of
// the record forms a run used (bench.py reports it next to the throughput) without knowing cri_types.h.
extern "C" int cri_job_hca_groups(const cri_job* j, cri_hca_group_info* out, int cap) {
    if (!j || j->kind != CRI_JOB_HCA_DECODE) return 0;
    int n = 0;
    for (const auto& a : j->hca_dec) {
        if (out && n < cap) {
            cri_hca_group_info g; memset(&g, 0, sizeof g);
            g.channels = a.channels; g.frames = a.frames; g.record_bytes = hca_record_bytes(a.channels);
            g.flags_offset = HCA_REC_TAIL(a.channels) + 8; g.narrow_flag = HCA_REC_NARROW; g.narrow_capable = a.narrow;
            g.plain = a.plain; g.transform_form = hca_transform_form(a); g.first_record_offset = j->hca_group_first_record[n];
            g.lines_offset = a.qc_offset; g.code_desc_offset = a.resg_offset;
            out[n] = g;
        }
        n++;
    }
    return n;
}

extern "C" int cri_job_enable_events(cri_job* j, int on) {
    if (!j) return CRI_ERR_INVALID_ARG;
    if (j->class_names.empty()) {
        if (j->kind == CRI_JOB_HCA_DECODE) j->class_names = {"k_hca_parse", "k_hca_transform"};
        else j->class_names = {j->dominant};
        j->class_events.resize(j->class_names.size());
        j->class_used.assign(j->class_names.size(), 0);
    }
    j->events_on = on != 0;
    return 0;
}

extern "C" int cri_job_event_ms(cri_job* j, float* ms, const char** names, int max_classes) {
    if (!j || !ms) return CRI_ERR_INVALID_ARG;
    DeviceGuard guard(j->device);
    if (!guard.ok()) return CRI_ERR_HIP;
    int n = (int)j->class_names.size();
    for (int c = 0; c < n && c < max_classes; c++) {
        float total = 0.f;
        for (size_t k = 0; k < j->class_used[c]; k++) {
            float t = 0.f;
            (void)hipEventSynchronize(j->class_events[c][k].second);
            if (hipEventElapsedTime(&t, j->class_events[c][k].first, j->class_events[c][k].second) == hipSuccess) total += t;
        }
        ms[c] = total;
        if (names) names[c] = j->class_names[c].c_str();
    }
    return n;
}

// ------------------------------------------------------------------------------------------------ host buffers in and out
// What a caller with host memory pays besides the kernels: device allocations, two PCIe crossings, a wait.  The host entry
// points keep, per device, an ARENA -- the four device buffers of the last call (grown on demand; released when a call leaves
// more than a quarter of the device's memory behind, or by cri_release_cache: allocating and freeing the 44 GB of a 10 000
// stream decode took anything between 10 ms and 0.9 s per call) and three private, non-blocking streams -- so a call allocates
// nothing in the steady state, waits for its own streams only (never hipDeviceSynchronize: the host's other streams are not
// this library's business) and copies its result straight into the buffer it returns.
// Large single-format HCA decode jobs are PIPELINED (the group cut into slices of whole parse tiles: slice k's input goes up
// on the upload stream while slice k-1 is parsed and transformed on the run stream and slice k-2's PCM comes down on the
// download stream, events ordering the three) -- see run_host_core for what that took.
namespace {
const size_t HOST_ARENA_KEEP_MIN = 512ull << 20;             // kept whatever the device: single-file calls
const uint64_t HOST_SLICE_BYTES = 128ull << 20;                // ... in slices of about this much traffic
const uint64_t HOST_STAGE_BYTES = 16ull << 20;                 // page-locked staging ring for pageable input: slots of this size
const uint32_t HOST_STAGE_SLOTS = 4;

// A caller's host buffer as the device sees it: page-locked memory (hipHostMalloc, hipHostRegister) is used in place, pageable
// memory is page-locked for the duration of the call (cheap next to the copies: 1-6 ms for 0.6-3 GB of resident memory on the
// host this was written on); `dev` stays null when neither works and the caller falls back to staged copies.
std::mutex g_locked_mu;
std::map<void*, std::pair<size_t, int>> g_locked;            // ranges this library has page-locked: base -> (bytes, calls using it)
struct PinnedView {
    void* host = nullptr; void* dev = nullptr; bool counted = false;
    bool open(void* p, size_t bytes) {
        host = p;
        if (!p || !bytes) return false;
        std::lock_guard<std::mutex> lk(g_locked_mu);
        auto it = g_locked.find(p);
        if (it != g_locked.end()) {                            // another call of ours holds it locked: share it if it covers this one
            if (it->second.first < bytes) return false;
            it->second.second++; counted = true;
        } else {
            hipPointerAttribute_t at;
            const bool known = hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeHost;
            (void)hipGetLastError();
            if (!known) {
                if (hipHostRegister(p, bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return false; }
                g_locked[p] = std::make_pair(bytes, 1); counted = true;
            }
        }
        if (hipHostGetDevicePointer(&dev, p, 0) != hipSuccess) { dev = nullptr; (void)hipGetLastError(); }
        return dev != nullptr;
    }
    ~PinnedView() {
        if (!counted) return;
        std::lock_guard<std::mutex> lk(g_locked_mu);
        auto it = g_locked.find(host);
        if (it != g_locked.end() && --it->second.second == 0) { (void)hipHostUnregister(host); g_locked.erase(it); }
    }
};

struct HostArena {
    std::mutex mu;
    hipStream_t s_up = nullptr, s_run = nullptr, s_down = nullptr;
    void* buf[4] = {nullptr, nullptr, nullptr, nullptr};       // in, out, scratch, status
    size_t cap[4] = {0, 0, 0, 0};
    std::vector<hipEvent_t> events;
    void* stage[HOST_STAGE_SLOTS] = {};                         // page-locked staging slots (pipelined path, pageable input)
    hipEvent_t stage_free[HOST_STAGE_SLOTS] = {};
    bool stage_busy[HOST_STAGE_SLOTS] = {};
    bool ensure_stage() {
        for (uint32_t k = 0; k < HOST_STAGE_SLOTS; k++) {
            if (!stage[k] && hipHostMalloc(&stage[k], HOST_STAGE_BYTES, hipHostMallocDefault) != hipSuccess) { stage[k] = nullptr; return false; }
            if (!stage_free[k] && hipEventCreateWithFlags(&stage_free[k], hipEventDisableTiming) != hipSuccess) { stage_free[k] = nullptr; return false; }
        }
        return true;
    }
    bool streams_ok() {
        if (s_run) return true;
        return hipStreamCreateWithFlags(&s_up, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&s_run, hipStreamNonBlocking) == hipSuccess &&
               hipStreamCreateWithFlags(&s_down, hipStreamNonBlocking) == hipSuccess;
    }
    bool ensure(int k, size_t need) {
        if (need < 256) need = 256;
        if (cap[k] >= need) return true;
        if (buf[k]) { (void)hipFree(buf[k]); buf[k] = nullptr; cap[k] = 0; }
        const size_t want = need < (64u << 20) ? ((need * 3 / 2 + 4095) & ~(size_t)4095) : need;    // small buffers grow with slack
        if (hipMalloc(&buf[k], want) != hipSuccess) { buf[k] = nullptr; return false; }
        cap[k] = want;
        return true;
    }
    hipEvent_t event(size_t i) {
        while (events.size() <= i) { hipEvent_t e; if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr; events.push_back(e); }
        return events[i];
    }
    void release_buffers() { for (int k = 0; k < 4; k++) { if (buf[k]) (void)hipFree(buf[k]); buf[k] = nullptr; cap[k] = 0; } }
    size_t keep = 0;                                            // bytes that may stay allocated between calls
    void trim() {
        if (!keep) {
            size_t fr = 0, total = 0;
            keep = hipMemGetInfo(&fr, &total) == hipSuccess && total / 4 > HOST_ARENA_KEEP_MIN ? total / 4 : HOST_ARENA_KEEP_MIN;
        }
        if (cap[0] + cap[1] + cap[2] + cap[3] > keep) release_buffers();
    }
    void destroy() {
        release_buffers();
        for (auto e : events) (void)hipEventDestroy(e);
        events.clear();
        for (uint32_t k = 0; k < HOST_STAGE_SLOTS; k++) {
            if (stage[k]) (void)hipHostFree(stage[k]);
            if (stage_free[k]) (void)hipEventDestroy(stage_free[k]);
            stage[k] = nullptr; stage_free[k] = nullptr; stage_busy[k] = false;
        }
        if (s_up) (void)hipStreamDestroy(s_up);
        if (s_run) (void)hipStreamDestroy(s_run);
        if (s_down) (void)hipStreamDestroy(s_down);
        s_up = s_run = s_down = nullptr;
    }
};
std::mutex g_arena_mu;
std::map<int, HostArena*> g_arenas;                            // per device; never destroyed at exit (the HIP runtime may be gone by then)
HostArena* arena_of(int device) {
    std::lock_guard<std::mutex> lk(g_arena_mu);
    auto it = g_arenas.find(device);
    if (it == g_arenas.end()) it = g_arenas.emplace(device, new HostArena()).first;
    return it->second;
}

// where the input bytes come from: one host blob laid out like the device input, or the items' own host buffers
struct HostSrc { const uint8_t* blob = nullptr; const cri_items* items = nullptr; };

// uploads device-input bytes [lo, hi) on stream s
int upload_range(const cri_job* j, const HostSrc& src, uint8_t* d_in, uint64_t lo, uint64_t hi, uint32_t& item_cursor, hipStream_t s) {
    if (hi <= lo) return 0;
    if (src.blob) return hipMemcpyAsync(d_in + lo, src.blob + lo, hi - lo, hipMemcpyHostToDevice, s) == hipSuccess ? 0 : CRI_ERR_HIP;
    // items: every item that starts inside the range goes up whole (ranges are cut at item starts by the callers)
    while (item_cursor < j->n && j->in_offsets[item_cursor] < hi) {
        const uint32_t i = item_cursor++;
        const uint64_t len = src.items->lens[i];
        if (len && hipMemcpyAsync(d_in + j->in_offsets[i], src.items->ptrs[i], len, hipMemcpyHostToDevice, s) != hipSuccess) return CRI_ERR_HIP;
    }
    return 0;
}

// A decode job the pipelined path can slice: one format group, on the run-per-wave transforms, streams in item order.
bool hca_decode_sliceable(const cri_job* j) {
    if (j->kind != CRI_JOB_HCA_DECODE || j->hca_dec.size() != 1 || j->convert_total) return false;
    const HcaDecArgs& a = j->hca_dec[0];
    if (a.noise_fill || !a.frames || !a.runs) return false;
    const bool in_regs = a.plain || a.inlane || a.channels == 1 || a.channels == 2 || a.channels == 4 || ((a.channels == 6 || a.channels == 8) && a.pairs_even);
    if (!in_regs) return false;
    const auto& S = j->hca_streams_host;
    if (S.size() != (size_t)(a.stream_end - a.stream_begin)) return false;
    for (size_t k = 1; k < S.size(); k++) if (S[k].item <= S[k - 1].item) return false;
    return true;
}

// The parts of a job the pipelined path cannot cut inside (an ADX decode job's lanes are laid out by length, not by item: a range of
// its lanes is not a range of its input; an HCA decode job of several format groups runs them one after the other, each over items
// from anywhere): the same items planned again as K jobs over consecutive item ranges of about equal traffic.
// A part addresses the parent's device input by the parent's own offsets and writes its output where the parent would -- item
// outputs are 64-byte aligned pieces back to back in both, which the builder checks, part by part.
bool host_parts_ready(cri_job* j, const HostSrc& src) {
    if (!j->partable || (j->kind != CRI_JOB_ADX_DECODE && j->kind != CRI_JOB_HCA_DECODE) || j->n < 8) return false;
    std::lock_guard<std::mutex> lk(j->parts_mu);
    if (j->parts_tried) return !j->host_parts.empty();
    // (latched below: on success, or when the layouts do not match -- a failure of the device (no memory for a part's metadata
    //  right now) leaves the question open for the next host run)
    bool transient = false;
    uint32_t K = (uint32_t)((j->in_bytes + j->out_bytes) / HOST_SLICE_BYTES);
    K = K < 4 ? 4 : (K > 64 ? 64 : K);
    if (K > j->n / 2) K = j->n / 2;
    const uint64_t total = j->in_bytes + j->out_bytes;
    std::vector<uint32_t> first{0};
    for (uint32_t i = 1; i < j->n; i++) {
        const uint64_t at = j->in_offsets[i] + j->out_offsets[i];
        if (at >= total / K * first.size() && first.size() < K) first.push_back(i);
    }
    first.push_back(j->n);
    bool good = true;
    for (size_t p = 0; p + 1 < first.size() && good; p++) {
        const uint32_t i0 = first[p], i1 = first[p + 1];
        ItemSrc ps;
        ps.blob = src.blob; ps.offsets = j->in_offsets.data() + i0; ps.n = i1 - i0;
        if (src.items) { ps.ptrs = src.items->ptrs + i0; ps.lens = src.items->lens + i0; }
        cri_job* part = nullptr;
        const int prc = j->kind == CRI_JOB_ADX_DECODE ? create_adx_decode(ps, &part)
                      : create_hca_decode(ps, j->part_keys.empty() ? nullptr : j->part_keys.data() + i0, j->part_subkeys.empty() ? nullptr : j->part_subkeys.data() + i0, nullptr, &part);
        if (prc || !part) { good = false; transient = prc == CRI_ERR_HIP; break; }
        j->host_parts.push_back(part);
        // (an item starts where its samples are line-aligned, wav_item_start: the part's layout is the parent's moved by whole lines)
        const uint64_t base = j->out_offsets[i0] - part->out_offsets[0];
        good = base % 128 == 0 && j->out_offsets[i0] >= part->out_offsets[0];
        for (uint32_t i = i0; i < i1 && good; i++) good = part->out_offsets[i - i0] + base == j->out_offsets[i] && part->host_status[i - i0] == j->host_status[i];
    }
    if (!good) { for (cri_job* part : j->host_parts) delete part; j->host_parts.clear(); j->parts_tried = !transient; return false; }
    j->host_part_first = first;
    j->parts_tried = true;
    return true;
}

// (out_from: the single-file calls copy one item out, from where it starts in the device output)
int run_host_core(cri_job* j, const HostSrc& src, uint8_t* out, uint64_t out_copy, int32_t* status, uint64_t out_from = 0) {
    DeviceGuard guard(j->device);
    if (!guard.ok()) return CRI_ERR_HIP;
    HostArena* shared = arena_of(j->device);
    HostArena own;                                             // a second thread on the same device does not wait for the first: it
    std::unique_lock<std::mutex> lk(shared->mu, std::try_to_lock);   // works with buffers and streams of its own for this call
    HostArena& A = lk.owns_lock() ? *shared : own;
    int rc = 0;
    auto ok = [&](hipError_t e) { if (e != hipSuccess && !rc) rc = CRI_ERR_HIP; return e == hipSuccess; };
    if (!A.streams_ok() || !A.ensure(0, j->in_bytes) || !A.ensure(1, j->out_bytes) || !A.ensure(2, j->scratch_bytes) || !A.ensure(3, (size_t)(j->n ? j->n : 1) * 4)) {
        if (&A == &own) own.destroy();
        return CRI_ERR_HIP;
    }
    uint8_t* d_in = (uint8_t*)A.buf[0]; uint8_t* d_out = (uint8_t*)A.buf[1]; uint8_t* d_scr = (uint8_t*)A.buf[2]; int32_t* d_st = (int32_t*)A.buf[3];
    if (out_from > j->out_bytes) out_from = j->out_bytes;
    if (out_copy > j->out_bytes - out_from) out_copy = j->out_bytes - out_from;
    const bool gaps = src.items && j->n && j->in_bytes;       // an items layout may leave bytes between the items: they are defined as zero
    uint32_t cursor = 0;
    // Large single-format HCA decode jobs are PIPELINED (CRICODECS_HOST_SLICE_MIN = the job size, bytes in + out, from which; 0 = always,
    // a huge value = never).  What it took on MI355X / ROCm 7.2 (2000 x 10 s streams: 0.64 GB in, 3.84 GB out; one stream in order:
    // 85 ms = upload 11 + kernels 4 + download 67 at the link's 57 GB/s):
    //  * with hipMemcpyAsync in both directions the sliced order took 130 ms: the runtime picks the DMA engine of a copy by what is idle
    //    when it is queued, a download queued while uploads are in flight lands on another engine than the uploads', and those run the
    //    host link at 30 GB/s (AMD_LOG_LEVEL=4 shows the engine per copy; from pageable memory the uploads are synchronous and nothing
    //    overlaps at all: 88 ms);
    //  * so the uploads do not use the DMA engines: a few workgroups pull the input across the link with plain loads from page-locked
    //    memory (k_pull_host) -- 8 workgroups, 19 GB/s: enough to stay ahead of a decoder whose output is 3.5-12x its input, and the
    //    downloads, alone on the engines, keep the 57 GB/s beside it (with 128 workgroups they drop to a fifth while a pull runs);
    //  * pageable input is made page-locked for the call (a blob: hipHostRegister, ~1 ms per 0.6 GB) or copied through a ring of
    //    page-locked staging slots by this thread (items: 26 GB/s on one core, beside the DMA); a pageable output buffer is locked for
    //    the call as well.
    // 72 ms for the same job (12.9 M frames/s; the download alone is 67).  The tests run both orders.
    const uint64_t slice_min = knobs().host_slice_min;
    const bool large = out_copy == j->out_bytes && !out_from && j->in_bytes + j->out_bytes >= slice_min && !j->events_on;
    const bool sliced = large && hca_decode_sliceable(j);
    const bool parted = large && !sliced && host_parts_ready(j, src);
    if (parted) for (const cri_job* part : j->host_parts) if (!A.ensure(2, part->scratch_bytes)) rc = CRI_ERR_HIP;
    d_scr = (uint8_t*)A.buf[2];
    if (rc) { if (&A == &own) own.destroy(); return rc; }
    if (!sliced && !parted) {
        if (gaps && !j->items_packed) ok(hipMemsetAsync(d_in, 0, j->in_bytes, A.s_run));
        if (!rc) rc = upload_range(j, src, d_in, 0, j->in_bytes, cursor, A.s_run);
        // bytes no kernel writes (alignment gaps, undecoded tails) are defined as zero
        if (j->out_bytes) ok(hipMemsetAsync(d_out, 0, j->out_bytes, A.s_run));
        if (!rc) rc = cri_job_run(j, d_in, d_out, d_scr, d_st, A.s_run);
        if (!rc && out_copy) ok(hipMemcpyAsync(out, d_out + out_from, out_copy, hipMemcpyDeviceToHost, A.s_run));
    } else {
        // ---- pipelined: the group cut into slices of whole parse tiles (or the job into its parts); three streams, two events per slice
        const HcaDecArgs base = sliced ? j->hca_dec[0] : HcaDecArgs{};
        const auto& S = j->hca_streams_host;
        const uint32_t tiles = (base.frames + 63) / 64, ns = (uint32_t)S.size();
        uint32_t K = (uint32_t)((j->in_bytes + j->out_bytes) / HOST_SLICE_BYTES);
        K = K < 4 ? 4 : (K > 64 ? 64 : K);
        if (K > tiles) K = tiles;
        const uint32_t TS = sliced ? (tiles + K - 1) / K : 1, nslices = sliced ? (tiles + TS - 1) / TS : (uint32_t)j->host_parts.size();
        for (uint32_t k = 0; k < 2 * nslices + 2; k++) if (!A.event(k)) rc = CRI_ERR_HIP;
        // Page-locked views of the caller's buffers (see the comment above): what is page-locked already is used as it is, a pageable
        // blob or output buffer is locked for the duration of the call, anything else goes through the staging ring.
        PinnedView vin, vout;
        const uint8_t* pull_src = nullptr;
        if (src.blob && vin.open((void*)src.blob, j->in_bytes)) pull_src = (const uint8_t*)vin.dev;
        vout.open(out, j->out_bytes);
        if (!pull_src && !A.ensure_stage()) rc = CRI_ERR_HIP;
        uint64_t piece_max = HOST_STAGE_BYTES;                    // (tests cut the pieces small: items then straddle them)
        { const uint64_t v = knobs().host_stage_piece; if (v >= 64 && v < piece_max) piece_max = v; }
        ok(hipMemsetAsync(d_out, 0, j->out_bytes, A.s_run));
        if (d_st) launch_fill_i32(d_st, 0, j->n, A.s_run);
        if (j->n_images)
            launch_scatter_images((const uint8_t*)j->d_img.p, (const uint64_t*)j->d_img_off.p, (const uint64_t*)j->d_img_dst.p, j->n_images, d_out, A.s_run);
        uint64_t in_pos = 0, out_pos = 0;
        uint32_t s_need = 0, s_done = 0;                         // streams whose input is up / whose PCM is down
        // (8 workgroups pull 19 GB/s: enough when the output is 6x the input and more -- HCA -- but not for ADX's 3.6x, where the upload
        //  of 1000 x 10 s became the longest stage: 44 ms; 16 workgroups: 37 ms, the downloads still at their rate; 32 begin to cost them)
        const uint32_t pull_wgs = knobs().host_pull_wgs ? (uint32_t)knobs().host_pull_wgs : (j->in_bytes * 5 > j->out_bytes ? 16u : 8u);
        uint32_t slot = 0;
        // device-input bytes [lo, hi) go up on the upload stream
        auto upload = [&](uint64_t lo, uint64_t hi) {
            if (pull_src) { launch_pull_host(d_in + lo, pull_src + lo, hi - lo, A.s_up, pull_wgs); return; }
            while (lo < hi && !rc) {
                const uint64_t n = hi - lo < piece_max ? hi - lo : piece_max;
                uint8_t* st = (uint8_t*)A.stage[slot];
                if (A.stage_busy[slot]) ok(hipEventSynchronize(A.stage_free[slot]));     // the pull that last read this slot
                if (src.blob) memcpy(st, src.blob + lo, n);
                else {                                            // the items that overlap [lo, lo + n), zeros between them
                    uint64_t at = lo;
                    while (cursor < j->n && j->in_offsets[cursor] < lo + n) {
                        const uint64_t b = j->in_offsets[cursor], e = b + src.items->lens[cursor];
                        const uint64_t cb = b > at ? b : at, ce = e < lo + n ? e : lo + n;
                        if (cb > at) memset(st + (at - lo), 0, cb - at);
                        if (ce > cb) { memcpy(st + (cb - lo), src.items->ptrs[cursor] + (cb - b), ce - cb); at = ce; } else if (cb > at) at = cb;
                        if (e > lo + n) break;                    // the rest of this item belongs to the next piece
                        cursor++;
                    }
                    if (at < lo + n) memset(st + (at - lo), 0, lo + n - at);
                }
                launch_pull_host(d_in + lo, st, n, A.s_up, pull_wgs);
                ok(hipEventRecord(A.stage_free[slot], A.s_up));
                A.stage_busy[slot] = true;
                slot = (slot + 1) % HOST_STAGE_SLOTS;
                lo += n;
            }
        };
        // A download is queued when the one before it has finished: the runtime picks the engine by what is idle at that moment
        // (see above), and only the first engine runs the link at full rate.  What the wait costs is the gap between two copies.
        hipEvent_t e_down = A.event(2 * nslices + 1);
        bool down_pending = false;
        auto download = [&](uint64_t lo, uint64_t hi) {
            if (down_pending) ok(hipEventSynchronize(e_down));
            ok(hipMemcpyAsync(out + lo, d_out + lo, hi - lo, hipMemcpyDeviceToHost, A.s_down));
            ok(hipEventRecord(e_down, A.s_down));
            down_pending = true;
        };
        // the input of slice u: every stream that has a frame below the slice's end
        auto slice_up = [&](uint32_t u) {
            const uint64_t t1 = (uint64_t)(u + 1) * TS < tiles ? (uint64_t)(u + 1) * TS : tiles;
            const uint64_t frames_end = t1 * 64 < base.frames ? t1 * 64 : base.frames;
            while (s_need < ns && S[s_need].first_frame < frames_end) s_need++;
            const uint64_t in_end = s_need == ns ? j->in_bytes : j->in_offsets[S[s_need].item];
            if (in_end > in_pos) { upload(in_pos, in_end); in_pos = in_end; }
            ok(hipEventRecord(A.event(2 * u), A.s_up));
        };
        if (parted) {
            // part k's items go up while part k - 1 decodes and part k - 2's output comes down
            auto part_up = [&](uint32_t u) {
                const uint64_t in_end = j->in_offsets[j->host_part_first[u + 1]];
                if (in_end > in_pos) { upload(in_pos, in_end); in_pos = in_end; }
                ok(hipEventRecord(A.event(2 * u), A.s_up));
            };
            if (!rc) part_up(0);
            for (uint32_t k = 0; k < nslices && !rc; k++) {
                const uint32_t i0 = j->host_part_first[k], i1 = j->host_part_first[k + 1];
                if (k + 1 < nslices) part_up(k + 1);
                ok(hipStreamWaitEvent(A.s_run, A.event(2 * k), 0));
                const int prc = cri_job_run(j->host_parts[k], d_in, d_out + (j->out_offsets[i0] - j->host_parts[k]->out_offsets[0]), d_scr, d_st ? d_st + i0 : nullptr, A.s_run);
                if (prc && !rc) rc = prc;
                ok(hipEventRecord(A.event(2 * k + 1), A.s_run)); ok(hipStreamWaitEvent(A.s_down, A.event(2 * k + 1), 0));
                if (j->out_offsets[i1] > out_pos) { download(out_pos, j->out_offsets[i1]); out_pos = j->out_offsets[i1]; }
            }
        }
        if (sliced && !rc) slice_up(0);
        for (uint32_t k = 0, t0 = 0; sliced && t0 < tiles && !rc; k++, t0 += TS) {
            const uint32_t t1 = t0 + TS < tiles ? t0 + TS : tiles;
            const uint64_t frames_end = (uint64_t)t1 * 64 < base.frames ? (uint64_t)t1 * 64 : base.frames;
            hipEvent_t e_up = A.event(2 * k), e_run = A.event(2 * k + 1);
            if (k + 1 < nslices) slice_up(k + 1);                 // one slice ahead: its upload has a whole download's time
            ok(hipStreamWaitEvent(A.s_run, e_up, 0));
            HcaDecArgs a = base;
            a.in = d_in; a.out = d_out; a.scratch = d_scr; a.status = d_st;
            a.formats = (const HcaFormat*)j->d_formats.p; a.streams = (const HcaStream*)j->d_streams.p;
            a.cipher_tables = (const uint8_t*)j->d_cipher.p; a.ath_tables = (const uint8_t*)j->d_ath.p; a.float_out = nullptr;
            a.tile_begin = t0; a.tile_count = t1 - t0;
            launch_hca_parse(a, A.s_run);
            // transform and download: the streams whose frames are all parsed now
            uint32_t s_to = s_done;
            while (s_to < ns && (uint64_t)S[s_to].first_frame + S[s_to].frames <= frames_end) s_to++;
            if (s_to > s_done) {
                const uint32_t r0 = S[s_done].first_run, r1 = s_to == ns ? base.runs : S[s_to].first_run;
                if (r1 > r0) { a.run_begin = r0; a.run_count = r1 - r0; launch_hca_transform(a, A.s_run); }
                ok(hipEventRecord(e_run, A.s_run)); ok(hipStreamWaitEvent(A.s_down, e_run, 0));
                const uint64_t out_end = s_to == ns ? j->out_bytes : j->out_offsets[S[s_to].item];
                if (out_end > out_pos) download(out_pos, out_end);
                out_pos = out_end > out_pos ? out_end : out_pos;
                s_done = s_to;
            }
        }
        if (!rc && hipGetLastError() != hipSuccess) rc = CRI_ERR_HIP;
        if (!rc && out_pos < j->out_bytes) {                       // (streams without frames at the end: their headers only)
            hipEvent_t e = A.event(2 * nslices);
            ok(hipEventRecord(e, A.s_run)); ok(hipStreamWaitEvent(A.s_down, e, 0));
            download(out_pos, j->out_bytes);
        }
        ok(hipStreamSynchronize(A.s_up));
        ok(hipStreamSynchronize(A.s_down));
        for (uint32_t k = 0; k < HOST_STAGE_SLOTS; k++) A.stage_busy[k] = false;
    }
    std::vector<int32_t> st(j->n ? j->n : 1, 0);
    if (status && j->n) ok(hipMemcpyAsync(st.data(), d_st, (size_t)j->n * sizeof(int32_t), hipMemcpyDeviceToHost, A.s_run));
    ok(hipStreamSynchronize(A.s_run));
    if (status) for (uint32_t i = 0; i < j->n; i++) status[i] = j->host_status[i] ? j->host_status[i] : st[i];
    if (&A == &own) own.destroy(); else A.trim();
    return rc;
}
}  // namespace

// Host buffers in, host buffers out.  `blob` is laid out like the device input (cri_job_input_offsets): for a job made from
// one blob, that blob; `out` must hold cri_job_output_bytes(job).
extern "C" int cri_job_run_host_into(cri_job* j, const uint8_t* blob, uint8_t* out, int32_t* status) {
    if (!j || !blob || (!out && j->out_bytes)) return CRI_ERR_INVALID_ARG;
    if (j->items_form && !j->items_packed) return CRI_ERR_INVALID_ARG;        // no blob describes that layout: cri_job_run_host_items
    HostSrc src; src.blob = blob;
    return run_host_core(j, src, out, j->out_bytes, status);
}

// The same for a job made from a cri_items list: every item goes up from its own host buffer to its place in the device input
// (`items` must describe the same n items, with the lengths the job was created with; its offsets are ignored).
extern "C" int cri_job_run_host_items(cri_job* j, const cri_items* items, uint8_t* out, int32_t* status) {
    if (!j || !items || items->n != j->n || (j->n && (!items->ptrs || !items->lens)) || (!out && j->out_bytes)) return CRI_ERR_INVALID_ARG;
    for (uint32_t i = 0; i < j->n; i++)
        if (items->lens[i] > j->in_offsets[i + 1] - j->in_offsets[i] || (items->lens[i] && !items->ptrs[i])) return CRI_ERR_INVALID_ARG;
    HostSrc src; src.items = items;
    return run_host_core(j, src, out, j->out_bytes, status);
}

extern "C" int cri_job_run_host(cri_job* j, const uint8_t* blob, uint8_t** out_blob, int32_t* status) {
    if (!j || !blob || !out_blob) return CRI_ERR_INVALID_ARG;
    uint8_t* host_out = (uint8_t*)malloc(j->out_bytes ? j->out_bytes : 1);
    if (!host_out) return CRI_ERR_NOMEM;
    const int rc = cri_job_run_host_into(j, blob, host_out, status);
    if (rc) { free(host_out); return rc; }
    *out_blob = host_out;
    return 0;
}

// Page-locked host memory for the buffers handed to cri_job_run_host*: with it the PCIe copies are asynchronous DMA.
extern "C" void* cri_pinned_alloc(size_t bytes) {
    void* p = nullptr;
    if (!cri_device_available() || hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
extern "C" void cri_pinned_free(void* p) { if (p) (void)hipHostFree(p); }

// Releases what the host entry points keep between calls on the calling thread's current device (buffers, streams, events).
extern "C" void cri_release_cache(void) {
    int d = -1;
    if (!cri_device_available() || hipGetDevice(&d) != hipSuccess) return;
    HostArena* A = arena_of(d);
    {
        std::lock_guard<std::mutex> lk(A->mu);
        A->destroy();
    }
    MetaCache* c = meta_cache_of(d);                           // recycled metadata allocations of destroyed jobs
    std::lock_guard<std::mutex> lk(c->mu);
    for (auto& e : c->free_list) (void)hipFree(e.first);
    c->free_list.clear();
}

// ------------------------------------------------------------------------------------------------ single-file entry points
// (one item: the result is copied from the device straight into the buffer the caller gets -- item_len bytes, not the aligned blob)
static int run_single(cri_job* j, const uint8_t* in, uint8_t** out, size_t* out_len, size_t item_len) {
    int32_t st = 0;
    int rc = j->host_status[0];
    uint8_t* res = nullptr;
    if (!rc) {
        res = (uint8_t*)malloc(item_len ? item_len : 1);
        if (!res) rc = CRI_ERR_NOMEM;
    }
    if (!rc) { HostSrc src; src.blob = in; rc = run_host_core(j, src, res, item_len, &st, j->out_offsets[0]); }
    if (!rc) rc = st;
    cri_job_destroy(j);
    if (rc) { free(res); return rc; }
    *out = res; *out_len = item_len;
    return 0;
}

extern "C" int cri_adx_decode(const uint8_t* adx, size_t len, uint8_t** out, size_t* out_len) {
    if (!adx || !out || !out_len) return CRI_ERR_INVALID_ARG;
    uint64_t offs[2] = {0, len};
    cri_job* j = nullptr;
    int rc = cri_job_create_adx_decode(adx, offs, 1, &j);
    if (rc) return rc;
    size_t item = 0;
    if (!j->host_status[0]) item = (size_t)(le32(j->images[0].bytes.data() + 4) + 8);   // RIFF size + 8
    return run_single(j, adx, out, out_len, item);
}

extern "C" int cri_adx_encode(const uint8_t* wav, size_t len, uint32_t bitdepth, uint32_t blocksize, uint32_t mode, uint32_t highpass,
                              uint32_t filter, uint32_t version, int force_no_looping, uint8_t** out, size_t* out_len) {
    if (!wav || !out || !out_len) return CRI_ERR_INVALID_ARG;
    uint64_t offs[2] = {0, len};
    cri_adx_encode_params p = {bitdepth, blocksize, mode, highpass, filter, version, (uint32_t)(force_no_looping != 0)};
    cri_job* j = nullptr;
    int rc = cri_job_create_adx_encode(wav, offs, 1, &p, &j);
    if (rc) return rc;
    size_t item = 0;
    if (!j->host_status[0]) item = (size_t)(j->images[1].dst + blocksize);             // tail block ends the file
    return run_single(j, wav, out, out_len, item);
}

extern "C" int cri_hca_decode(const uint8_t* hca, size_t len, uint32_t header_size, uint64_t key, uint16_t subkey, uint8_t** out, size_t* out_len) {
    if (!hca || !out || !out_len) return CRI_ERR_INVALID_ARG;
    uint64_t offs[2] = {0, len};
    cri_job* j = nullptr;
    int rc = create_hca_decode(ItemSrc::from_blob(hca, offs, 1), &key, &subkey, &header_size, &j);
    if (rc) return rc;
    size_t item = 0;
    if (!j->host_status[0]) item = (size_t)(le32(j->images[0].bytes.data() + 4) + 8);
    rc = run_single(j, hca, out, out_len, item);
    if (rc <= -211 && rc >= -216) rc = CRI_ERR_HCA_DECODE;                             // hca.cpp:3441-3444
    return rc;
}

extern "C" int cri_hca_crypt(uint8_t* hca, size_t len, uint32_t encrypt, uint32_t header_size, uint32_t type, uint64_t key, uint16_t subkey) {
    if (!hca) return CRI_ERR_INVALID_ARG;
    uint64_t offs[2] = {0, len};
    cri_job* j = nullptr;
    int rc = create_hca_crypt(ItemSrc::from_blob(hca, offs, 1), encrypt, type, &key, &subkey, &header_size, &j);
    if (rc) return rc;
    uint8_t* res = nullptr; size_t n = 0;
    rc = run_single(j, hca, &res, &n, len);
    if (rc) return rc;
    memcpy(hca, res, len);
    free(res);
    return 0;
}

extern "C" int cri_hca_encode(const uint8_t* wav, size_t len, uint32_t force_no_looping, uint32_t quality, uint8_t** out, size_t* out_len) {
    if (!wav || !out || !out_len) return CRI_ERR_INVALID_ARG;
    uint64_t offs[2] = {0, len};
    cri_job* j = nullptr;
    int rc = cri_job_create_hca_encode(wav, offs, 1, force_no_looping, quality, &j);
    if (rc) return rc;
    size_t item = 0;
    if (!j->host_status[0]) { const uint8_t* h = j->images[0].bytes.data(); item = (size_t)be16(h + 6) + (size_t)be32(h + 16) * be16(h + 28); }
    return run_single(j, wav, out, out_len, item);
}

#ifdef CRI_TESTING
// TEST BUILD ONLY: the encoder's table blob as the planner uploads it (HCA_ET_BYTES bytes)
extern "C" int cri_test_enc_tables(uint8_t* out, size_t cap) {
    std::vector<uint8_t> blob;
    const int rc = hca_enc_build_tables(blob);
    if (rc) return rc;
    if (!out || cap < blob.size()) return CRI_ERR_INVALID_ARG;
    memcpy(out, blob.data(), blob.size());
    return (int)blob.size();
}
// TEST BUILD ONLY (libcricodecs_hip_testing.so): sets a knob for the jobs created from here on.  Not thread-safe; not part of the C ABI.
extern "C" int cri_test_set(const char* key, long long value) {
    Knobs& k = knobs_mut();
    if (!key) { k = Knobs(); return 0; }                         // reset to the shipped defaults (the environment is not read again)
    if (!strcmp(key, "adx_mapping")) k.adx_mapping = (int)value;
    else if (!strcmp(key, "host_slice_min")) k.host_slice_min = (uint64_t)value;
    else if (!strcmp(key, "host_stage_piece")) k.host_stage_piece = (uint64_t)value;
    else if (!strcmp(key, "no_inlane")) k.no_inlane = (int)value;
    else if (!strcmp(key, "adx_warm_pct")) k.adx_warm_pct = (uint64_t)value;
    else if (!strcmp(key, "adx_seglen")) k.adx_seglen = (uint64_t)value;
    else if (!strcmp(key, "hca_run")) k.hca_run = (uint64_t)value;
    else if (!strcmp(key, "host_pull_wgs")) k.host_pull_wgs = (uint64_t)value;
    else if (!strcmp(key, "bad_launch")) k.bad_launch = (int)value;
    else return CRI_ERR_INVALID_ARG;
    return 0;
}
#endif
