/* cricodecs_hip.h -- C ABI of libcricodecs_hip.so: MI355X-native ADX / HCA encode + decode core.
 *
 * This is the drop-in boundary for the hot path of Youjose/PyCriCodecs: the five codec methods of the
 * reference's CPython extension `CriCodecs` (/root/reference/CriCodecs/CriCodecs.cpp:8-17) are replaced by the
 * five single-file entry points below; everything they compute per frame/block runs as hand-written HIP
 * kernels for gfx950.  Plain C types only (no torch / Python types); the library links libamdhip64 only.
 * The reference-side binding a maintainer would add is shown in INTEGRATION.md.
 *
 * There is NO CPU fallback: every entry point that has device work returns CRI_ERR_HIP when no gfx950 device
 * (or the HIP runtime) is available.
 *
 * Return codes (0 = success).  Negative codes keep the reference's own numbering per domain so that a binding
 * can raise the identical Python exception type and message:
 *   -1 .. -18    ADX domain        (adx.cpp:11-30; -3 maps to NotImplementedError, the rest to ValueError, adx.cpp:32-38)
 *   -101 .. -110 WAV/PCM domain    (pcm.cpp:22-33 numbered 1..10, offset by 100; ValueError, pcm.cpp:35-38)
 *   -201 .. -204 HCA domain        (py_codec_err(-1..-4), hca.cpp:3252-3268; ValueError)
 *   -211 .. -216 HCA per-frame detail (HCA_ERROR_PARAMS..BITREADER, hca.cpp:64-70, offset by 210); single-file
 *                entry points collapse them to -202 exactly as HcaDecode does (hca.cpp:3441-3444)
 *   -301 ..      library domain (argument, memory, HIP runtime, unsupported-on-device)
 */
#ifndef CRICODECS_HIP_H
#define CRICODECS_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CRI_OK 0
#define CRI_ERR_ADX(n) (-(n))
#define CRI_ERR_PCM(n) (-(100 + (n)))
#define CRI_ERR_HCA_HEADER (-201)
#define CRI_ERR_HCA_DECODE (-202)
#define CRI_ERR_HCA_CHANNEL_CONFIG (-203)
#define CRI_ERR_HCA_ENCODE (-204)
#define CRI_ERR_HCA_FRAME(n) (-(210 + (n))) /* n = 1 params, 2 header, 3 checksum, 4 sync, 5 unpack, 6 bitreader */
#define CRI_ERR_INVALID_ARG (-301)
#define CRI_ERR_NOMEM (-302)
#define CRI_ERR_HIP (-303)
#define CRI_ERR_UNSUPPORTED (-304)
#define CRI_ERR_AWB_HEADER (-401)   /* awb.py:37-38 "Invalid AWB header." */
#define CRI_ERR_AWB_INTSIZE (-402)  /* awb.py:95-106 "Unknown int size." */
#define CRI_ERR_USM_HEADER (-411)   /* usm.py:129-130 "Unsupported file type" (no CRID chunk first) */
#define CRI_ERR_USM_CHUNK (-412)    /* usm.py:189 "Unsupported chunk type", or a chunk header cut short by the end of the file */
#define CRI_ITEM_SKIPPED 1          /* per-item status of a job that was told to leave the item to another job */

/* ---------------------------------------------------------------------------------------------------------
 * Single-file entry points, host buffers in / host buffers out (malloc'ed; release with cri_free).
 * One call == one call of the reference extension method named in the comment.
 * ------------------------------------------------------------------------------------------------------- */

/* CriCodecs.AdxDecode(bytes) -> WAV bytes.  Replaces AdxDecode, adx.cpp:546-558 (ADX::Decode 380-415). */
int cri_adx_decode(const uint8_t* adx, size_t len, uint8_t** out, size_t* out_len);

/* CriCodecs.AdxEncode(wav, bitdepth, blocksize, encoding, highpass, filter, adx_version, force_no_looping).
 * Replaces AdxEncode, adx.cpp:517-544 (ADX::Encode 416-506); argument order as parsed at adx.cpp:527. */
int cri_adx_encode(const uint8_t* wav, size_t len, uint32_t bitdepth, uint32_t blocksize, uint32_t encoding_mode,
                   uint32_t highpass_frequency, uint32_t filter, uint32_t adx_version, int force_no_looping,
                   uint8_t** out, size_t* out_len);

/* CriCodecs.HcaDecode(data, header_size, key, subkey) -> WAV bytes.  Replaces HcaDecode, hca.cpp:3340-3457. */
int cri_hca_decode(const uint8_t* hca, size_t len, uint32_t header_size, uint64_t key, uint16_t subkey,
                   uint8_t** out, size_t* out_len);

/* CriCodecs.HcaEncode(wav, force_nolooping, quality) -> HCA bytes.  Replaces HcaEncode, hca.cpp:3459-3489.
 * quality: 0 Highest, 1 High, 2 Middle, 3 Low, 4 Lowest; any other value behaves as High (chunk.py:73 passes 5). */
int cri_hca_encode(const uint8_t* wav, size_t len, uint32_t force_no_looping, uint32_t quality,
                   uint8_t** out, size_t* out_len);

/* CriCodecs.HcaCrypt(buf, crypt, header_size, type, key, subkey): whole-file en/decrypt IN PLACE on `hca`
 * (the binding copies first and returns new bytes).  Replaces HcaCrypt, hca.cpp:3271-3337.
 * encrypt: 1 = encrypt with `type` (56 or 1), 0 = decrypt (type ignored). */
int cri_hca_crypt(uint8_t* hca, size_t len, uint32_t encrypt, uint32_t header_size, uint32_t type,
                  uint64_t key, uint16_t subkey);

void cri_free(void* p);

/* Message for a return code: the reference's own strings for the ADX / PCM / HCA domains
 * (adx.cpp:11-30, pcm.cpp:22-33, hca.cpp:3255-3264). */
const char* cri_strerror(int code);
/* Identity of the sources the loaded library was built from: 24 hex digits, a sha256 prefix over csrc/, this header and the compiler
 * flags (pycricodecs_amd/build.py, source_id()).  A binding that ships a prebuilt library can hold it to its source tree. */
const char* cri_build_id(void);

/* 1 when a gfx950 device is usable, 0 otherwise (thread-safe). */
int cri_device_available(void);

/* Device selection for hosts that drive more than one GPU from one process.  The current device is per THREAD (HIP's rule):
 * cri_set_device selects the GPU that jobs created by, and single-file calls made from, the calling thread use.  A job
 * stays bound to the device it was created on: cri_job_run / cri_job_run_host* / cri_job_destroy may be called from any
 * thread and switch to that device for the duration of the call (buffers and stream passed to cri_job_run must belong to
 * it).  One process per GPU (the multi-GPU layout of this path) needs none of these. */
int cri_device_count(void);
int cri_set_device(int device);   /* 0, CRI_ERR_INVALID_ARG (no such device) or CRI_ERR_HIP */
int cri_get_device(void);         /* the calling thread's current device, -1 without a device */

/* ---------------------------------------------------------------------------------------------------------
 * Batch jobs, device resident.  A job is built on the host from the items' headers only, then run any number
 * of times on device buffers (inputs already in HBM, outputs stay in HBM); nothing in cri_job_run allocates
 * or synchronises, so it can be captured in a hipGraph (tests/test_gpu_round4.py::test_job_run_captured_in_a_hip_graph
 * replays one capture per job kind).
 * Lifetime: a run's kernels read the job's metadata.  cri_job_destroy waits for the last run that was ENQUEUED through
 * cri_job_run on every stream the job was run on (an event per stream behind its last kernel) before that memory is reused -- a
 * job may be destroyed while its work is in flight.  A run captured into a graph leaves no such event: the job must outlive every launch of a graph that holds it.
 *
 * Items are described AFS2-style: one blob + offsets[n+1]; item i = blob[offsets[i], offsets[i+1]).
 * The device input handed to cri_job_run must be a byte-identical copy of that blob.
 * Output: one blob, item i at [out_offsets[i], out_offsets[i+1]) -- exactly the bytes the single-file call
 * would return (WAV incl. header for decodes, ADX / HCA files for encodes, the rewritten HCA for crypt).
 * Items whose header is rejected on the host get their reference error code in host_status and zero output.
 * ------------------------------------------------------------------------------------------------------- */
typedef struct cri_job cri_job;

enum { CRI_JOB_ADX_DECODE = 1, CRI_JOB_ADX_ENCODE = 2, CRI_JOB_HCA_DECODE = 3, CRI_JOB_HCA_ENCODE = 4, CRI_JOB_HCA_CRYPT = 5,
       CRI_JOB_USM_DEMUX = 6, CRI_JOB_SFA_PACK = 7 };

typedef struct cri_adx_encode_params {
    uint32_t bitdepth, blocksize, encoding_mode, highpass_frequency, filter, adx_version, force_no_looping;
} cri_adx_encode_params;

/* Items given one by one instead of as one blob (the *_items creators): n host pointers + lengths -- several items may point
 * at the same host bytes, so a batch that repeats files costs no host copy, and a binding can pass its list of buffers as it
 * is.  offsets[n+1] places the items in the DEVICE input handed to cri_job_run (item i at offsets[i], offsets[n] = its size;
 * offsets[i+1] - offsets[i] >= lens[i], e.g. the output offsets of the job that produced them); NULL =
 * packed back to back.  cri_job_input_offsets() returns the layout either way.  Host data is only read while the job is
 * created (headers) and by cri_job_run_host_items. */
typedef struct cri_items { const uint8_t* const* ptrs; const uint64_t* lens; const uint64_t* offsets; uint32_t n; } cri_items;

/* keys / subkeys: per-item arrays, or NULL for all-zero. */
int cri_job_create_hca_decode(const uint8_t* blob, const uint64_t* offsets, uint32_t n,
                              const uint64_t* keys, const uint16_t* subkeys, cri_job** job);
int cri_job_create_adx_decode(const uint8_t* blob, const uint64_t* offsets, uint32_t n, cri_job** job);
int cri_job_create_adx_encode(const uint8_t* blob, const uint64_t* offsets, uint32_t n,
                              const cri_adx_encode_params* params /* one for all items */, cri_job** job);

/* AFS2 / AWB wave bank as the batch descriptor (replaces the per-file loop of PyCriCodecs/awb.py:54-88, AWB.extract /
 * AWB.getfiles, and its header parse awb.py:32-52).  cri_awb_index is host-only: with offsets == NULL it returns the item
 * count; otherwise offsets[n+1] (already aligned and clipped to len) and kinds[n].  cri_job_create_awb_decode builds one
 * HCA decode job (key mixed with the bank's subkey, awb.py:72) and one ADX decode job over the SAME blob: upload the
 * bank once, run both jobs (items of the other kind carry status CRI_ITEM_SKIPPED and produce no output). */
#define CRI_AWB_OTHER 0
#define CRI_AWB_HCA 1
#define CRI_AWB_ADX 2
int cri_awb_index(const uint8_t* awb, size_t len, uint32_t* n_items, uint32_t* align, uint16_t* subkey, uint32_t* header_size,
                  uint64_t* offsets, uint8_t* kinds, uint32_t cap);
int cri_job_create_awb_decode(const uint8_t* awb, size_t len, uint64_t key, cri_job** hca_job, cri_job** adx_job);
int cri_job_create_hca_encode(const uint8_t* blob, const uint64_t* offsets, uint32_t n,
                              uint32_t force_no_looping, uint32_t quality, cri_job** job);

/* USM audio (@SFA) streams: the chunk layer either side of the ADX / HCA codecs (PyCriCodecs/usm.py).  The container's
 * tables (CRID / @UTF), video and subtitles stay with the caller.
 *  cri_usm_audio_mask     the 32-byte audio mask of a key (USM.init_key, usm.py:47-118 = USMBuilder.init_key 1179-1252).
 *  cri_usm_index          host-only walk over the chunk headers (USM.demux, usm.py:134-190): with chunks == NULL it returns
 *                         the count.  payload_offset/payload_len describe chunk data from its data offset on, padding included.
 *  cri_job_create_usm_audio_demux   one item per @SFA channel number, ascending: the channel's type-0 payloads concatenated,
 *                         padding stripped (USM.reader, usm.py:263-277); with decrypt != 0 ADX payloads get the extractor's
 *                         AudioMask (usm.py:313-322: bytes from 0x140 on, whole 8-byte words).  As in the reference the codec is
 *                         audio_codec of the most recent @SFA header chunk's @UTF table (usm.py:165-168); a container without
 *                         header chunks is judged by each channel's first payload (0x80 0x00 = ADX).  Input blob = the USM.
 *  cri_job_create_sfa_pack          one item per audio stream (ADX or HCA file bytes): its list of @SFA chunks, concatenated --
 *                         32-byte chunk headers, payloads padded to 0x20, "#CONTENTS END" last (USMBuilder.get_data,
 *                         usm.py:578-716); encrypt_audio masks ADX payloads (AudioMask, usm.py:1290-1300).
 *                         tags: cri_job_item_tags() = channel number | codec << 16 (demux), chunk count (pack). */
typedef struct cri_usm_chunk {
    char fourcc[4]; uint32_t chno, type, padding; uint64_t payload_offset; uint32_t payload_len, frame_time, frame_rate, pad;
} cri_usm_chunk;
#define CRI_USM_CODEC_ADX 2
#define CRI_USM_CODEC_HCA 4
int cri_usm_audio_mask(uint64_t key, uint8_t mask[32]);
int cri_usm_index(const uint8_t* usm, size_t len, cri_usm_chunk* chunks, uint32_t cap, uint32_t* count);
int cri_job_create_usm_audio_demux(const uint8_t* usm, size_t len, uint64_t key, uint32_t decrypt, cri_job** job);
int cri_job_create_sfa_pack(const uint8_t* blob, const uint64_t* offsets, uint32_t n, uint32_t codec, uint64_t key,
                            uint32_t encrypt_audio, cri_job** job);
const uint32_t* cri_job_item_tags(const cri_job* job);      /* n entries (jobs that define them), else NULL */
const uint64_t* cri_job_item_sizes(const cri_job* job);     /* byte length of every output item (these two job kinds), else NULL */
int cri_job_create_hca_crypt(const uint8_t* blob, const uint64_t* offsets, uint32_t n, uint32_t encrypt, uint32_t type,
                             const uint64_t* keys, const uint16_t* subkeys, cri_job** job);
int cri_job_create_hca_decode_items(const cri_items* items, const uint64_t* keys, const uint16_t* subkeys, cri_job** job);
int cri_job_create_adx_decode_items(const cri_items* items, cri_job** job);
int cri_job_create_adx_encode_items(const cri_items* items, const cri_adx_encode_params* params, cri_job** job);
int cri_job_create_hca_encode_items(const cri_items* items, uint32_t force_no_looping, uint32_t quality, cri_job** job);
int cri_job_create_hca_crypt_items(const cri_items* items, uint32_t encrypt, uint32_t type, const uint64_t* keys, const uint16_t* subkeys,
                                   cri_job** job);

int cri_job_device(const cri_job* job);                     /* device the job is bound to (see cri_set_device) */
uint32_t cri_job_kind(const cri_job* job);
uint32_t cri_job_items(const cri_job* job);
uint64_t cri_job_input_bytes(const cri_job* job);
uint64_t cri_job_output_bytes(const cri_job* job);
/* n+1 entries: item i starts at offsets[i] in d_out; offsets[n] = cri_job_output_bytes.  Bytes between items are zero after a run.
 * Encoded files start on 64-byte boundaries; a decoded WAV is placed so that the SAMPLES behind its header start a 128-byte line
 * (offsets[0] is 84 for a 44-byte header): the decoders' stores then write whole lines.  An item's length is its own
 * (RIFF size + 8, the ADX / HCA file's header fields; cri_job_item_sizes for jobs whose items carry none). */
const uint64_t* cri_job_output_offsets(const cri_job* job);
const uint64_t* cri_job_input_offsets(const cri_job* job);  /* n+1 entries: where cri_job_run expects item i in d_in */
const int32_t* cri_job_host_status(const cri_job* job);     /* n entries, header-stage result per item */
uint64_t cri_job_scratch_bytes(const cri_job* job);         /* device scratch the run needs (may be 0) */
/* Work units of the job in the metric's unit: HCA frames (1024 samples x all channels) or ADX frames
 * (one block per channel); cri_job_units2 = ADX blocks (0 for HCA jobs). */
uint64_t cri_job_units(const cri_job* job);
uint64_t cri_job_units2(const cri_job* job);
/* Algorithmic bytes one run moves (compressed bytes + PCM bytes of the units, headers excluded): the
 * numerator of the roofline figure. */
uint64_t cri_job_algorithmic_bytes(const cri_job* job);

/* Enqueue the job on `hip_stream` (a hipStream_t, NULL = default stream).  d_status: int32[n] on the device,
 * receives 0 or the first failing frame's code per item (may be NULL).  d_scratch: cri_job_scratch_bytes()
 * bytes (may be NULL when 0). */
int cri_job_run(cri_job* job, const void* d_in, void* d_out, void* d_scratch, int32_t* d_status, void* hip_stream);

/* Validation run of an HCA decode job (north_star states the HCA tolerance on the floats BEFORE the int16 clamp:
 * hca.cpp:1987-1992 wave[subframe][sample], read by clHCA_ReadSamples16 at 339-360).  Same as cri_job_run, and additionally
 * stores every decoded frame's pre-clamp samples: item i at d_floats + cri_job_float_offsets()[i], as
 * [frame][1024 samples][channels] floats, for the frames the decode visits (all of them unless the padding spans whole
 * frames).  d_floats holds cri_job_float_count() floats.  Not a product path: the instances that store floats are slower. */
uint64_t cri_job_float_count(const cri_job* job);
const uint64_t* cri_job_float_offsets(const cri_job* job);  /* n+1 entries, in floats; NULL for other job kinds */
int cri_job_run_floats(cri_job* job, const void* d_in, void* d_out, void* d_scratch, int32_t* d_status, float* d_floats, void* hip_stream);

/* Frame-record layout of an HCA decode job's scratch, one entry per format group (launch set), for diagnostics: the word at
 * first_record_offset + g * record_bytes + flags_offset of frame g (0 <= g < frames) has narrow_flag set when that frame's
 * quantised lines went through scratch in the narrow (int8) form.  Returns the number of groups (fills at most cap). */
typedef struct cri_hca_group_info {
    uint32_t channels, frames, record_bytes, flags_offset, narrow_flag, narrow_capable, plain;
    uint32_t transform_form;      /* which transform kernel takes the group: 0 k_hca_transform_generic, 1 k_hca_transform<.>, 2 / 3 / 4 the
                                     in-lane kernel k_hca_transform_plain (plain / joint stereo + HFR / + v3.0 noise fill), | 8 its wide form
                                     (a wave per four channels) */
    uint64_t first_record_offset;
    uint64_t lines_offset;        /* the group's quantised lines, tile-major (64 frames per tile) */
    uint64_t code_desc_offset;    /* the group's band code descriptions: [tile][channel][block 8][frame 64][band 16] bytes, low nibble =
                                     most bits a symbol of that band can take (0: the band carries no bits) */
} cri_hca_group_info;
int cri_job_hca_groups(const cri_job* job, cri_hca_group_info* out, int cap);

/* Name of the job's dominant kernel (for profiling cross-checks). */
const char* cri_job_dominant_kernel(const cri_job* job);

/* Per-kernel timing with HIP events recorded on the run's own stream.  After cri_job_enable_events(job, 1) every
 * cri_job_run brackets each kernel class with events; cri_job_event_ms waits for the last run and returns the
 * elapsed milliseconds per class (summed over the job's launches of that class) and the class names.
 * Classes: HCA decode {k_hca_parse, k_hca_transform}; other jobs have one class.  Returns the class count. */
int cri_job_enable_events(cri_job* job, int on);
int cri_job_event_ms(cri_job* job, float* ms, const char** names, int max_classes);

void cri_job_destroy(cri_job* job);

/* Host buffers in, host buffers out: upload, run, download -- what the five single-file entry points do for one item,
 * for a whole job.  Replaces the per-file host loops of the reference's callers (hca.py:250, adx.py, awb.py:54-88).
 *   cri_job_run_host        `blob` laid out like the device input (cri_job_input_offsets()): for a job created from one
 *                           blob, that blob.  *out_blob is malloc'ed (cri_free); status[n] is caller-provided.
 *   cri_job_run_host_into   the same into a caller-owned buffer of cri_job_output_bytes(job) bytes (reuse it across calls: a
 *                           fresh allocation of that size costs more in page faults than the copy itself).  A job created
 *                           from a cri_items list WITH caller offsets has no blob form: CRI_ERR_INVALID_ARG, use the next one.
 *   cri_job_run_host_items  for a job created from a cri_items list: every item is uploaded from its own host buffer
 *                           (`items`: the same n items with the same lengths; its offsets are ignored).
 * The library keeps, per device, the device buffers of the last host call (up to a quarter of the device's memory; more is
 * released when the call returns) and private streams: these calls allocate nothing in the steady state and wait for their
 * own streams only.  Large single-format HCA decode jobs (64 MB in + out and more; CRICODECS_HOST_SLICE_MIN = bytes moves the
 * switch, 0 = always) run PIPELINED -- uploads, kernels and downloads of successive slices overlap, the uploads pulled across
 * the link by a few workgroups so that the downloads have the DMA engines to themselves: 13 M frames/s against 11 M in one
 * piece on MI355X (PCIe-bound either way).  Any host memory works: page-locked buffers (cri_pinned_alloc, or memory the caller
 * registered with the HIP runtime) are used in place, a pageable blob or output buffer is page-locked for the duration of the
 * call, separate items are copied through a page-locked staging ring.  cri_release_cache drops what is kept for the calling
 * thread's current device. */
int cri_job_run_host(cri_job* job, const uint8_t* blob, uint8_t** out_blob, int32_t* status);
int cri_job_run_host_into(cri_job* job, const uint8_t* blob, uint8_t* out, int32_t* status);
int cri_job_run_host_items(cri_job* job, const cri_items* items, uint8_t* out, int32_t* status);
void* cri_pinned_alloc(size_t bytes);   /* NULL without a device / on failure */
void cri_pinned_free(void* p);
void cri_release_cache(void);

#ifdef __cplusplus
}
#endif
#endif /* CRICODECS_HIP_H */
