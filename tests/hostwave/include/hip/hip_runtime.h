// hostwave -- TEST INFRASTRUCTURE.  A stand-in for <hip/hip_runtime.h> that lets the product's kernel sources (pycricodecs_amd/csrc/*.hip,
// untouched apart from tests/hostwave/translate.py's rewriting of inline assembly and of LDS declarations) compile for x86-64 and
// run under a lockstep emulation of a gfx950 workgroup: one fiber per lane, wave64 cross-lane operations (DPP, ds_swizzle,
// ds_bpermute, readlane, ballot) as rendezvous between the fibers of a wave, LDS as a per-workgroup arena with a guard page behind the
// launch's dynamic size, "device memory" = host memory.  The parity tests of tests/test_gpu_*.py then run against the SAME sources
// without a GPU (tests/test_hostwave.py; tests/hostwave/README.md).  Nothing under pycricodecs_amd/ includes, links or loads this; it is no CPU fallback.
//
// What it checks: the kernels' data flow -- indices, bit twiddling, cross-lane patterns, LDS layouts and sizes, float arithmetic in
// IEEE binary32 with the same operation order (no contraction; explicit fma where the source says fma).  What it cannot check: the
// compiler's code for gfx950, timing, occupancy, or races that the hardware's in-order LDS pipeline hides.
#pragma once
#define HOSTWAVE 1
#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1
#endif
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#include <math.h>
// HIP's vector types, with 4-byte alignment: a global_load_dwordx4 / ds_read_b128 of the device only needs the alignment the
// kernels give it (dwords for global memory), an x86 movaps would fault on the same address
#define HW_VEC(T, N2, N4, M2, M4) \
    struct __attribute__((aligned(4))) N2 { T x, y; }; struct __attribute__((aligned(4))) N4 { T x, y, z, w; }; \
    static inline N2 M2(T x, T y) { return N2{x, y}; } static inline N4 M4(T x, T y, T z, T w) { return N4{x, y, z, w}; }
HW_VEC(uint32_t, uint2, uint4, make_uint2, make_uint4)
HW_VEC(int32_t, int2, int4, make_int2, make_int4)
HW_VEC(float, float2, float4, make_float2, make_float4)
HW_VEC(uint16_t, ushort2, ushort4, make_ushort2, make_ushort4)
HW_VEC(int16_t, short2, short4, make_short2, make_short4)
#undef HW_VEC

#undef __global__
#undef __device__
#undef __host__
#undef __shared__
#undef __constant__
#undef __forceinline__
#undef __noinline__
#undef __launch_bounds__
#define __global__
#define __device__
#define __host__
#define __constant__
#define __forceinline__ inline
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define HIP_SYMBOL(x) (&(x))
#define HIP_KERNEL_NAME(...) __VA_ARGS__

struct dim3 {
    uint32_t x, y, z;
    constexpr dim3(uint32_t x_ = 1, uint32_t y_ = 1, uint32_t z_ = 1) : x(x_), y(y_), z(z_) {}
};

// ---------------------------------------------------------------- the fake runtime API (hostwave.cpp)
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNoDevice = 100, hipErrorNotReady = 600 };
typedef struct hw_stream* hipStream_t;
typedef struct hw_event* hipEvent_t;
typedef struct hw_graph* hipGraph_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive = 1 };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2 };
struct hipPointerAttribute_t { hipMemoryType type; int device; void* devicePointer; void* hostPointer; };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipHostRegisterDefault = 0 };

extern "C" {
hipError_t hipMalloc(void** p, size_t n);
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags);
hipError_t hipHostFree(void* p);
hipError_t hipHostRegister(void* p, size_t n, unsigned flags);
hipError_t hipHostUnregister(void* p);
hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned flags);
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* at, const void* p);
hipError_t hipMemGetInfo(size_t* fr, size_t* total);
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st);
hipError_t hipMemset(void* d, int v, size_t n);
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st);
hipError_t hipMemcpyToSymbol(void* sym, const void* src, size_t n);
hipError_t hipMemcpyFromSymbol(void* dst, const void* sym, size_t n);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipStreamIsCapturing(hipStream_t s, hipStreamCaptureStatus* st);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventQuery(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipGetDevice(int* d);
hipError_t hipSetDevice(int d);
hipError_t hipGetDeviceCount(int* n);
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int dev);
hipError_t hipDeviceSynchronize(void);
hipError_t hipGetLastError(void);
hipError_t hipFuncSetAttribute(const void* f, hipFuncAttribute a, int v);
const char* hipGetErrorString(hipError_t e);
}

// ---------------------------------------------------------------- the workgroup emulation
#include <type_traits>
namespace hw {
template <class T> using val_t = std::remove_cv_t<std::remove_reference_t<T>>;
struct Wave;
struct Block;
struct uint3_ { uint32_t x, y, z; };
struct Lane {
    void* sp;                    // saved stack pointer while switched out
    uint3_ tid;                  // threadIdx
    uint32_t lane;               // lane of its wave
    uint32_t linear;             // thread of its block
    Wave* wave;
    Block* block;
    Lane *next, *prev;           // ring of the block's live lanes
    bool done;
    volatile bool released;      // set by whoever completes the rendezvous this lane waits in
    uint8_t* stack;
};
struct Wave {
    uint64_t live;               // lanes that have not returned
    // pending rendezvous: lanes of a wave normally arrive at the same site together; groups exist for divergent code
    struct Group { uint32_t site; uint64_t mask; uint32_t val[64]; };
    Group groups[4];
    int n_groups;
    uint32_t res[64];            // the completed rendezvous: every participant's value ...
    uint64_t res_mask;           // ... who took part ...
    uint64_t res_nz;             // ... and whose value was non-zero (ballots)
    uint32_t index;
};
struct Block {
    uint3_ bid, bdim, gdim;      // blockIdx, blockDim, gridDim
    uint32_t n_lanes, n_waves, live;
    uint32_t bar_count;          // lanes waiting in __syncthreads
    uint32_t stall;              // waits since anything last moved (deadlock = divergent rendezvous: see hostwave.cpp)
    uint8_t* lds_dynamic;        // ends at a guard page
    uint32_t lds_dynamic_bytes;
    uint8_t* lds_static;
    uint32_t lds_static_used;
    uint32_t lds_slot_site[32], lds_slot_off[32], n_lds_slots;
};
extern thread_local Lane* cur;

void launch_grid(dim3 grid, dim3 block, size_t lds, void (*fn)(void*), void* ctx, const char* name);
template <class F> static void launch_thunk(void* p) { (*(F*)p)(); }
template <class F> inline void launch(dim3 grid, dim3 block, size_t lds, hipStream_t, F f, const char* name) { launch_grid(grid, block, lds, &launch_thunk<F>, &f, name); }

void* dyn_lds();
void* static_lds(uint32_t bytes, uint32_t align, uint32_t site);
uint32_t rendezvous(uint32_t site, uint32_t v);      // deposit v, wait for the wave, then cur->wave->res / res_mask / res_nz hold the exchange
void syncthreads();
[[noreturn]] void fail(const char* what, uint32_t site);

// ---- cross-lane operations (all lanes of the wave that are in this code path take part)
inline uint32_t lane_id() { return cur->lane; }
inline uint32_t readlane(uint32_t v, uint32_t l, uint32_t site) {
    rendezvous(site, v);
    const Wave& w = *cur->wave;
    if (!((w.res_mask >> (l & 63)) & 1)) fail("readlane of a lane that is not in this code path", site);
    return w.res[l & 63];
}
inline uint32_t readfirstlane(uint32_t v, uint32_t site) { rendezvous(site, v); const Wave& w = *cur->wave; return w.res[__builtin_ctzll(w.res_mask)]; }
inline uint64_t ballot(bool p, uint32_t site) { rendezvous(site, p ? 1u : 0u); const Wave& w = *cur->wave; return w.res_nz & w.res_mask; }
inline bool any(bool p, uint32_t site) { return ballot(p, site) != 0; }
inline bool all(bool p, uint32_t site) { rendezvous(site, p ? 1u : 0u); const Wave& w = *cur->wave; return (w.res_nz & w.res_mask) == w.res_mask; }
inline void wave_barrier(uint32_t site) { rendezvous(site | 0x80000000u, 0); }      // (bit 31: among divergent groups, exchanges complete before barriers)
// value of lane `src` (valid = the pattern names a lane inside its row / group), 0 or `old` otherwise as the instruction says
inline uint32_t take(int src, bool valid, uint32_t old, bool zero_if_invalid, uint32_t site) {
    const Wave& w = *cur->wave;
    if (valid && ((w.res_mask >> src) & 1)) return w.res[src];
    return zero_if_invalid ? 0u : old;
}
uint32_t update_dpp(uint32_t old, uint32_t src, uint32_t ctrl, uint32_t row_mask, uint32_t bank_mask, bool bound_ctrl, uint32_t site);
uint32_t ds_swizzle(uint32_t v, uint32_t pattern, uint32_t site);
inline uint32_t ds_bpermute(uint32_t addr, uint32_t v, uint32_t site) { rendezvous(site, v); return take((addr >> 2) & 63, true, 0, true, site); }
inline uint32_t mbcnt_lo(uint32_t m, uint32_t v) { const uint32_t l = cur->lane; return v + __builtin_popcount(l >= 32 ? m : (m & ((1u << l) - 1u))); }
inline uint32_t mbcnt_hi(uint32_t m, uint32_t v) { const uint32_t l = cur->lane; return v + (l <= 32 ? 0 : __builtin_popcount(m & ((1u << (l - 32)) - 1u))); }

template <class T> inline T shfl_idx(T v, int index, uint32_t site) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "shfl of 4- or 8-byte values");
    uint32_t w[sizeof(T) / 4];
    memcpy(w, &v, sizeof(T));
    for (unsigned i = 0; i < sizeof(T) / 4; i++) w[i] = ds_bpermute((uint32_t)index << 2, w[i], site);
    memcpy(&v, w, sizeof(T));
    return v;
}
template <class T> inline T shfl(uint32_t site, T v, int src, int width = 64) { const int self = cur->lane; return shfl_idx(v, (src & (width - 1)) + (self & ~(width - 1)), site); }
template <class T> inline T shfl_xor(uint32_t site, T v, int m, int width = 64) { const int self = cur->lane; int i = self ^ m; if (i >= ((self + width) & ~(width - 1))) i = self; return shfl_idx(v, i, site); }
template <class T> inline T shfl_up(uint32_t site, T v, unsigned d, int width = 64) { const int self = cur->lane; int i = self - (int)d; if (i < (self & ~(width - 1))) i = self; return shfl_idx(v, i, site); }
template <class T> inline T shfl_down(uint32_t site, T v, unsigned d, int width = 64) { const int self = cur->lane; int i = self + (int)d; if ((self & (width - 1)) + (int)d >= width) i = self; return shfl_idx(v, i, site); }

// ---- scalar forms of single instructions
inline int32_t sext24(int32_t v) { return (int32_t)((uint32_t)v << 8) >> 8; }
inline int32_t mul24(int32_t a, int32_t b) { return (int32_t)((int64_t)sext24(a) * sext24(b)); }
inline int32_t v_mad_i32_i24(int32_t a, int32_t b, int32_t c) { return (int32_t)((uint32_t)mul24(a, b) + (uint32_t)c); }
inline uint32_t perm(uint32_t s0, uint32_t s1, uint32_t sel) {
    const uint64_t v = ((uint64_t)s0 << 32) | s1;
    uint32_t r = 0;
    for (int i = 0; i < 4; i++) {
        const uint32_t c = (sel >> (8 * i)) & 0xFF;
        uint32_t b;
        if (c <= 7) b = (uint32_t)(v >> (8 * c)) & 0xFF;
        else if (c <= 11) b = ((v >> (16 * (c - 8) + 15)) & 1) ? 0xFF : 0x00;
        else if (c == 12) b = 0x00;
        else b = 0xFF;
        r |= b << (8 * i);
    }
    return r;
}
inline uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (sh & 31)); }
inline uint32_t ubfe(uint32_t v, uint32_t off, uint32_t w) { off &= 31; w &= 31; return w ? (v >> off) & ((1u << w) - 1u) : 0u; }
inline int32_t sbfe(int32_t v, uint32_t off, uint32_t w) {
    off &= 31; w &= 31;
    if (!w) return 0;
    const uint32_t f = ((uint32_t)v >> off) & ((1u << w) - 1u);
    return (int32_t)(f << (32 - w)) >> (32 - w);
}
inline uint32_t sad_u8(uint32_t a, uint32_t b, uint32_t c) {
    for (int i = 0; i < 4; i++) { const int x = (a >> (8 * i)) & 0xFF, y = (b >> (8 * i)) & 0xFF; c += x > y ? x - y : y - x; }
    return c;
}
inline float fmed3f(float a, float b, float c) { const float lo = fminf(a, b), hi = fmaxf(a, b); return fmaxf(lo, fminf(hi, c)); }
typedef short s2v __attribute__((ext_vector_type(2)));
inline s2v cvt_pk_i16(int32_t a, int32_t b) {
    s2v r;
    r.x = (short)(a > 32767 ? 32767 : (a < -32768 ? -32768 : a));
    r.y = (short)(b > 32767 ? 32767 : (b < -32768 ? -32768 : b));
    return r;
}
typedef float f2v __attribute__((ext_vector_type(2)));
// v_pk_{mul,add}_f32 with op_sel / op_sel_hi / neg_lo / neg_hi on the two sources: bit 0 = source a, bit 1 = source b
template <char OP> inline f2v v_pk(f2v a, f2v b, int sel_lo, int sel_hi, int neg_lo, int neg_hi) {
    float al = (sel_lo & 1) ? a.y : a.x, bl = (sel_lo & 2) ? b.y : b.x;
    float ah = (sel_hi & 1) ? a.y : a.x, bh = (sel_hi & 2) ? b.y : b.x;
    if (neg_lo & 1) al = -al;
    if (neg_lo & 2) bl = -bl;
    if (neg_hi & 1) ah = -ah;
    if (neg_hi & 2) bh = -bh;
    f2v r;
    if (OP == '*') { r.x = al * bl; r.y = ah * bh; } else { r.x = al + bl; r.y = ah + bh; }
    return r;
}
inline void ds_add_u32(uint32_t lds_addr, uint32_t v) { *(uint32_t*)(uintptr_t)lds_addr += v; }
}  // namespace hw

// ---------------------------------------------------------------- names the kernel sources use
#define threadIdx (hw::cur->tid)
#define blockIdx (hw::cur->block->bid)
#define blockDim (hw::cur->block->bdim)
#define gridDim (hw::cur->block->gdim)
#define warpSize 64
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) hw::launch((grid), (block), (lds), (stream), [=]() { kernel(__VA_ARGS__); }, #kernel)

#define __syncthreads() hw::syncthreads()
#define __builtin_amdgcn_wave_barrier() hw::wave_barrier(__LINE__)
#define __builtin_amdgcn_s_barrier() hw::syncthreads()
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_readlane(v, l) ((hw::val_t<decltype(v)>)hw::readlane((uint32_t)(v), (uint32_t)(l), __LINE__))
#define __builtin_amdgcn_readfirstlane(v) ((hw::val_t<decltype(v)>)hw::readfirstlane((uint32_t)(v), __LINE__))
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) ((int)hw::update_dpp((uint32_t)(old), (uint32_t)(src), (ctrl), (rm), (bm), (bc), __LINE__))
#define __builtin_amdgcn_ds_swizzle(v, p) ((int)hw::ds_swizzle((uint32_t)(v), (p), __LINE__))
#define __builtin_amdgcn_ds_bpermute(a, v) ((int)hw::ds_bpermute((uint32_t)(a), (uint32_t)(v), __LINE__))
#define __builtin_amdgcn_ballot_w64(p) hw::ballot((p), __LINE__)
#define __builtin_amdgcn_mbcnt_lo(m, v) hw::mbcnt_lo((m), (v))
#define __builtin_amdgcn_mbcnt_hi(m, v) hw::mbcnt_hi((m), (v))
#define __builtin_amdgcn_perm(a, b, s) hw::perm((a), (b), (s))
#define __builtin_amdgcn_alignbit(a, b, s) hw::alignbit((a), (b), (s))
#define __builtin_amdgcn_ubfe(v, o, w) hw::ubfe((v), (o), (w))
#define __builtin_amdgcn_sbfe(v, o, w) hw::sbfe((v), (o), (w))
#define __builtin_amdgcn_sad_u8(a, b, c) hw::sad_u8((a), (b), (c))
#define __builtin_amdgcn_fmed3f(a, b, c) hw::fmed3f((a), (b), (c))
#define __builtin_amdgcn_cvt_pk_i16(a, b) hw::cvt_pk_i16((a), (b))
#define __ballot(p) hw::ballot((p), __LINE__)
#define __any(p) hw::any((p), __LINE__)
#define __all(p) hw::all((p), __LINE__)
#define __shfl(...) hw::shfl(__LINE__, __VA_ARGS__)
#define __shfl_xor(...) hw::shfl_xor(__LINE__, __VA_ARGS__)
#define __shfl_up(...) hw::shfl_up(__LINE__, __VA_ARGS__)
#define __shfl_down(...) hw::shfl_down(__LINE__, __VA_ARGS__)
#define __lane_id() hw::lane_id()

static inline int __mul24(int a, int b) { return hw::mul24(a, b); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
static inline double __longlong_as_double(long long v) { double d; memcpy(&d, &v, 8); return d; }
static inline long long __double_as_longlong(double d) { long long v; memcpy(&v, &d, 8); return v; }
template <class T> static inline T __ldg(const T* p) { return *p; }
#define HW_MINMAX(T) static inline T min(T a, T b) { return b < a ? b : a; } static inline T max(T a, T b) { return a < b ? b : a; }
HW_MINMAX(int) HW_MINMAX(unsigned) HW_MINMAX(long) HW_MINMAX(unsigned long) HW_MINMAX(long long) HW_MINMAX(unsigned long long) HW_MINMAX(float) HW_MINMAX(double)
#undef HW_MINMAX
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
template <class T> static inline hipError_t hipHostMalloc(T** p, size_t n, unsigned f) { return hipHostMalloc((void**)p, n, f); }

// atomics: workgroups run on several host threads, global counters are shared
template <class T> static inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicOr(T* p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicAnd(T* p, T v) { return __atomic_fetch_and(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicXor(T* p, T v) { return __atomic_fetch_xor(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicExch(T* p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicMin(T* p, T v) {
    T o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v < o && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return o;
}
template <class T> static inline T atomicMax(T* p, T v) {
    T o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v > o && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return o;
}
template <class T> static inline T atomicCAS(T* p, T cmp, T v) { __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED); return cmp; }

// streaming stores: the census build tells them apart (they do not allocate in the modelled L2: tools/l2_replay.py)
namespace hw { extern thread_local bool nt_store_now; }
#define __builtin_nontemporal_store(v, p) do { hw::nt_store_now = true; *(p) = (v); hw::nt_store_now = false; } while (0)

// sources guard gfx950-only helpers (inline assembly by name) with __HIPCC__; here those are translated and wanted
#ifndef __HIPCC__
#define __HIPCC__ 1
#endif
