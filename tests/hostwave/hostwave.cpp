// hostwave.cpp -- TEST INFRASTRUCTURE: the fake HIP runtime and the workgroup emulation behind tests/hostwave/include/hip/hip_runtime.h.
//
// A launch runs its workgroups on a small pool of host threads; a workgroup is a ring of fibers, one per lane, switched by hand
// (hw_switch below).  A lane runs until it needs the others: a cross-lane operation (rendezvous of its wave), __syncthreads (of
// the block) or its return.  Whoever arrives last completes the rendezvous and everybody reads the exchanged values out of the
// wave's `res` array afterwards.
//
// Divergence.  Lanes of a wave normally arrive at the same site (source line).  If every live lane of the block is waiting and a
// wave's lanes sit in different places -- a cross-lane operation inside `if (lane < 32)`, the others already at the barrier behind
// it -- the group with the smallest site number is completed with the lanes it has (the hardware runs the branch with the others
// masked off); sources that are not in the exchange read as the instruction defines (0 for ds_bpermute / bound_ctrl, `old` for DPP).
// HOSTWAVE_VERBOSE=1 reports every such completion.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <signal.h>
#include <execinfo.h>
#include <unistd.h>
#include <sys/mman.h>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>
#include <atomic>
#include <algorithm>
#include <string>

extern "C" void hw_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl hw_switch
.type hw_switch,@function
hw_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size hw_switch,.-hw_switch
)");

namespace hw {
thread_local Lane* cur = nullptr;
thread_local bool nt_store_now = false;

static const size_t STACK_BYTES = 1u << 20;        // per lane (virtual; touched pages only)
static const uint32_t MAX_LANES = 1024;
static const uint32_t LDS_STATIC_BYTES = 64 * 1024, LDS_DYNAMIC_MAX = 160 * 1024;
static int g_verbose = -1;
static int g_order = 0;                            // HOSTWAVE_ORDER: which lane of a workgroup runs first and which way the ring goes
static std::atomic<uint64_t> g_subset_completions{0};

struct Worker {                                    // one per host thread of the pool
    Block block;
    Lane lanes[MAX_LANES];
    Wave waves[MAX_LANES / 64];
    uint8_t* stacks = nullptr;
    uint8_t* lds = nullptr;                        // [static 64 KB][dynamic, ending at a PROT_NONE page], below 4 GB: LDS addresses are 32-bit in the kernels
    void* main_sp = nullptr;
    void (*fn)(void*) = nullptr;
    void* ctx = nullptr;
};
static thread_local Worker* tw = nullptr;

[[noreturn]] void fail(const char* what, uint32_t site) {
    Lane* l = cur;
    fprintf(stderr, "hostwave: %s (source line %u; block %u,%u thread %u)\n", what, site & 0x7FFFFFFFu, l ? l->block->bid.x : 0, l ? l->block->bid.y : 0, l ? l->linear : 0);
    if (l) {
        Worker* w = tw;
        for (uint32_t i = 0; i < w->block.n_waves; i++) {
            const Wave& wv = w->waves[i];
            fprintf(stderr, "  wave %u: live %016llx, %d pending group(s):", i, (unsigned long long)wv.live, wv.n_groups);
            for (int g = 0; g < wv.n_groups; g++) fprintf(stderr, " [line %u%s: %016llx]", wv.groups[g].site & 0x7FFFFFFFu, wv.groups[g].site >> 31 ? " (wave barrier)" : "", (unsigned long long)wv.groups[g].mask);
            fprintf(stderr, "\n");
        }
        fprintf(stderr, "  __syncthreads: %u of %u live lanes waiting\n", w->block.bar_count, w->block.live);
    }
    abort();
}

static inline void switch_to(Lane* to) {
    Lane* from = cur;
    cur = to;
    hw_switch(&from->sp, to->sp);
}

static void complete_group(Wave& w, int gi) {
    Wave::Group& g = w.groups[gi];
    uint64_t nz = 0;
    for (uint64_t m = g.mask; m; m &= m - 1) {
        const int l = __builtin_ctzll(m);
        w.res[l] = g.val[l];
        if (g.val[l]) nz |= 1ull << l;
    }
    w.res_mask = g.mask;
    w.res_nz = nz;
    Worker* k = tw;
    for (uint64_t m = g.mask; m; m &= m - 1) k->lanes[w.index * 64 + __builtin_ctzll(m)].released = true;
    w.groups[gi] = w.groups[--w.n_groups];
    k->block.stall = 0;
}

// every live lane of the block is waiting and nothing can complete by itself: finish the earliest partial rendezvous
static void resolve_stall() {
    Worker* k = tw;
    Block& b = k->block;
    Wave* best = nullptr;
    int best_g = -1;
    for (uint32_t i = 0; i < b.n_waves; i++) {
        Wave& w = k->waves[i];
        for (int g = 0; g < w.n_groups; g++)
            if (!best || w.groups[g].site < best->groups[best_g].site) { best = &w; best_g = g; }
    }
    if (!best) fail("deadlock: every lane waits in __syncthreads but not all of them arrive (a barrier in divergent code?)", 0);
    if (g_verbose) fprintf(stderr, "hostwave: divergent rendezvous at line %u completed with lanes %016llx of %016llx (block %u)\n", best->groups[best_g].site & 0x7FFFFFFFu,
                           (unsigned long long)best->groups[best_g].mask, (unsigned long long)best->live, b.bid.x);
    g_subset_completions++;
    complete_group(*best, best_g);
}

static inline void wait_released() {
    Lane* me = cur;
    Block& b = *me->block;
    while (!me->released) {
        if (++b.stall > 2 * b.live + 8) resolve_stall();
        if (me->released) break;
        switch_to(me->next);
    }
    me->released = false;
}

uint32_t rendezvous(uint32_t site, uint32_t v) {
    Lane* me = cur;
    Wave& w = *me->wave;
    me->block->stall = 0;
    int gi = -1;
    for (int g = 0; g < w.n_groups; g++) if (w.groups[g].site == site) { gi = g; break; }
    if (gi < 0) {
        if (w.n_groups == 4) fail("more than four divergent rendezvous pending in one wave", site);
        gi = w.n_groups++;
        w.groups[gi].site = site;
        w.groups[gi].mask = 0;
    }
    Wave::Group& g = w.groups[gi];
    if ((g.mask >> me->lane) & 1) fail("a lane arrived twice at a rendezvous that is not complete", site);
    g.mask |= 1ull << me->lane;
    g.val[me->lane] = v;
    if (g.mask == w.live) { complete_group(w, gi); me->released = false; return v; }
    wait_released();
    return v;
}

static void release_barrier(Block& b) {
    Worker* k = tw;
    for (Lane* l = k->lanes; l < k->lanes + b.n_lanes; l++) if (!l->done) l->released = true;
    b.bar_count = 0;
    b.stall = 0;
}

void syncthreads() {
    Lane* me = cur;
    Block& b = *me->block;
    b.stall = 0;
    if (++b.bar_count == b.live) { release_barrier(b); me->released = false; return; }
    wait_released();
}

static void lane_exit() {
    Lane* me = cur;
    Worker* k = tw;
    Block& b = k->block;
    Wave& w = *me->wave;
    me->done = true;
    b.live--;
    b.stall = 0;
    w.live &= ~(1ull << me->lane);
    // its leaving may complete what the others wait for
    for (int g = 0; g < w.n_groups; g++) if (w.live && w.groups[g].mask == w.live) { complete_group(w, g); break; }
    if (b.live && b.bar_count == b.live) release_barrier(b);
    if (!b.live) { cur = nullptr; void* dummy; hw_switch(&dummy, k->main_sp); __builtin_unreachable(); }
    me->prev->next = me->next;
    me->next->prev = me->prev;
    Lane* to = me->next;
    cur = to;
    void* dummy;
    hw_switch(&dummy, to->sp);
    __builtin_unreachable();
}

static void lane_entry() {
    Worker* k = tw;
    k->fn(k->ctx);
    lane_exit();
}

void* dyn_lds() { return cur->block->lds_dynamic; }

void* static_lds(uint32_t bytes, uint32_t align, uint32_t site) {
    Block& b = *cur->block;
    for (uint32_t i = 0; i < b.n_lds_slots; i++) if (b.lds_slot_site[i] == site) return b.lds_static + b.lds_slot_off[i];
    if (b.n_lds_slots == 32) fail("more than 32 __shared__ declarations in one kernel", site);
    if (align < 4) align = 4;
    const uint32_t off = (b.lds_static_used + align - 1) & ~(align - 1);
    if (off + bytes > LDS_STATIC_BYTES) fail("static LDS beyond 64 KB", site);
    b.lds_slot_site[b.n_lds_slots] = site;
    b.lds_slot_off[b.n_lds_slots++] = off;
    b.lds_static_used = off + bytes;
    return b.lds_static + off;
}

// A fault inside a kernel: say what was touched (the guard page behind the launch's LDS, a lane's stack, something else), by which
// lane, and where (return addresses for llvm-symbolizer; the build keeps line tables).
static void on_fault(int sig, siginfo_t* si, void*) {
    Worker* k = tw;
    Lane* l = cur;
    char buf[512];
    const uint8_t* a = (const uint8_t*)si->si_addr;
    int n = snprintf(buf, sizeof buf, "hostwave: signal %d at address %p", sig, (void*)a);
    if (k && k->lds && a >= k->lds && a < k->lds + LDS_STATIC_BYTES + LDS_DYNAMIC_MAX + 4096) {
        if (a >= k->block.lds_dynamic) n += snprintf(buf + n, sizeof buf - n, ": LDS byte %ld of a launch with %u bytes of dynamic LDS", (long)(a - k->block.lds_dynamic), k->block.lds_dynamic_bytes);
        else n += snprintf(buf + n, sizeof buf - n, ": LDS, %ld bytes BEFORE the dynamic region (static part: %u bytes used)", (long)(k->block.lds_dynamic - a), k->block.lds_static_used);
    } else if (k && k->stacks && a >= k->stacks - 4096 && a < k->stacks + STACK_BYTES * MAX_LANES) n += snprintf(buf + n, sizeof buf - n, ": a lane's stack (overflow?)");
    if (l) n += snprintf(buf + n, sizeof buf - n, "; block %u, thread %u (wave %u lane %u)", l->block->bid.x, l->linear, l->linear >> 6, l->lane);
    buf[n++] = '\n';
    (void)!write(2, buf, n);
    void* bt[48];
    const int d = backtrace(bt, 48);
    backtrace_symbols_fd(bt, d, 2);
    _exit(139);
}

static void install_fault_handler() {
    static uint8_t alt[1 << 16];
    static thread_local bool done = false;
    if (done) return;
    done = true;
    stack_t ss;
    ss.ss_sp = malloc(1 << 16); ss.ss_size = 1 << 16; ss.ss_flags = 0;
    (void)alt;
    sigaltstack(&ss, nullptr);
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_fault;
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
    sigaction(SIGSEGV, &sa, nullptr);
    sigaction(SIGBUS, &sa, nullptr);
}

static void worker_init(Worker* k) {
    install_fault_handler();
    k->stacks = (uint8_t*)mmap(nullptr, STACK_BYTES * MAX_LANES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    const size_t lds_map = LDS_STATIC_BYTES + LDS_DYNAMIC_MAX + 4096;
    k->lds = (uint8_t*)mmap(nullptr, lds_map, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_32BIT, -1, 0);
    if (k->stacks == MAP_FAILED || k->lds == MAP_FAILED) { perror("hostwave: mmap"); abort(); }
    mprotect(k->lds + LDS_STATIC_BYTES + LDS_DYNAMIC_MAX, 4096, PROT_NONE);
}

#ifdef HOSTWAVE_TRAFFIC
static void traffic_begin(const char* name, dim3 grid, dim3 block);
static void traffic_end();
static void traffic_block_end(Worker* k);
#endif

static void run_block(Worker* k, dim3 grid, dim3 block, uint32_t bx, uint32_t by, uint32_t bz, size_t lds) {
    Block& b = k->block;
    const uint32_t n = block.x * block.y * block.z;
    b.bid = {bx, by, bz};
    b.bdim = {block.x, block.y, block.z};
    b.gdim = {grid.x, grid.y, grid.z};
    b.n_lanes = n;
    b.n_waves = (n + 63) / 64;
    b.live = n;
    b.bar_count = 0;
    b.stall = 0;
    b.lds_static = k->lds;
    b.lds_static_used = 0;
    b.n_lds_slots = 0;
    b.lds_dynamic_bytes = (uint32_t)lds;
    // the dynamic region ENDS at the guard page (whole 16-byte units): a read or write past the launch's size faults
    b.lds_dynamic = k->lds + LDS_STATIC_BYTES + LDS_DYNAMIC_MAX - ((lds + 15) & ~(size_t)15);
    memset(b.lds_static, 0xA5, LDS_STATIC_BYTES);                      // LDS is not zero at a workgroup's start
    memset(b.lds_dynamic, 0xA5, (lds + 15) & ~(size_t)15);
    for (uint32_t w = 0; w < b.n_waves; w++) {
        Wave& wv = k->waves[w];
        const uint32_t cnt = n - w * 64 < 64 ? n - w * 64 : 64;
        wv.live = cnt == 64 ? ~0ull : ((1ull << cnt) - 1);
        wv.n_groups = 0;
        wv.res_mask = wv.res_nz = 0;
        wv.index = w;
    }
    for (uint32_t i = 0; i < n; i++) {
        Lane& l = k->lanes[i];
        l.linear = i;
        l.tid = {i % block.x, (i / block.x) % block.y, i / (block.x * block.y)};
        l.lane = i & 63;
        l.wave = &k->waves[i >> 6];
        l.block = &b;
        l.next = &k->lanes[(i + 1) % n];
        l.prev = &k->lanes[(i + n - 1) % n];
        if (g_order == 1) std::swap(l.next, l.prev);       // HOSTWAVE_ORDER=reverse: the ring runs the other way (waves and lanes in descending order)
        l.done = false;
        l.released = false;
        l.stack = k->stacks + (size_t)i * STACK_BYTES;
        void** top = (void**)(l.stack + STACK_BYTES);
        top[-1] = nullptr;                       // (a return address lane_entry never uses)
        top[-2] = (void*)&lane_entry;
        for (int r = 3; r <= 8; r++) top[-r] = nullptr;
        l.sp = &top[-8];
    }
    // the lane that runs first: 0, the last one (reverse), or the first lane of a wave drawn from the block's number (HOSTWAVE_ORDER=rotate)
    Lane* first = &k->lanes[g_order == 1 ? n - 1 : (g_order == 2 ? 64 * (((bx + 7 * by) * 2654435761u >> 7) % b.n_waves) : 0)];
    cur = first;
    hw_switch(&k->main_sp, first->sp);
    cur = nullptr;
#ifdef HOSTWAVE_TRAFFIC
    traffic_block_end(k);
#endif
}

// ---- the pool
struct Pool {
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::vector<std::thread> threads;
    uint64_t generation = 0;
    bool quit = false;
    // the current launch
    dim3 grid, block;
    size_t lds = 0;
    void (*fn)(void*) = nullptr;
    void* ctx = nullptr;
    std::atomic<uint64_t> next{0};
    uint64_t total = 0;
    uint32_t active = 0;

    void work(Worker* k) {
        for (;;) {
            const uint64_t i = next.fetch_add(1);
            if (i >= total) break;
            const uint32_t bx = (uint32_t)(i % grid.x), by = (uint32_t)((i / grid.x) % grid.y), bz = (uint32_t)(i / ((uint64_t)grid.x * grid.y));
            k->fn = fn;
            k->ctx = ctx;
            run_block(k, grid, block, bx, by, bz, lds);
        }
    }
    void thread_main() {
        Worker* k = new Worker;
        worker_init(k);
        tw = k;
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv_work.wait(lk, [&] { return quit || generation != seen; });
            if (quit) return;
            seen = generation;
            lk.unlock();
            work(k);
            lk.lock();
            if (--active == 0) cv_done.notify_all();
        }
    }
    void start() {
        int n = 0;
        if (const char* e = getenv("HOSTWAVE_THREADS")) n = atoi(e);
        if (n <= 0) { n = (int)std::thread::hardware_concurrency(); if (n > 16) n = 16; if (n < 1) n = 1; }
        for (int i = 0; i < n; i++) threads.emplace_back([this] { thread_main(); });
    }
};
static Pool* g_pool = nullptr;
static std::mutex g_launch_mu;

void launch_grid(dim3 grid, dim3 block, size_t lds, void (*fn)(void*), void* ctx, const char* name) {
    std::lock_guard<std::mutex> one(g_launch_mu);                      // launches are synchronous and one at a time
    if (g_verbose < 0) {
        const char* e = getenv("HOSTWAVE_VERBOSE"); g_verbose = e && *e == '1';
        const char* o = getenv("HOSTWAVE_ORDER"); g_order = o && !strcmp(o, "reverse") ? 1 : (o && !strcmp(o, "rotate") ? 2 : 0);
    }
    const uint64_t total = (uint64_t)grid.x * grid.y * grid.z;
    const uint32_t n = block.x * block.y * block.z;
    if (!total || !n) return;
    if (n > MAX_LANES || lds > LDS_DYNAMIC_MAX) { fprintf(stderr, "hostwave: launch of %u threads / %zu bytes of LDS\n", n, lds); abort(); }
    if (!g_pool) { g_pool = new Pool; g_pool->start(); }
    Pool& p = *g_pool;
#ifdef HOSTWAVE_TRAFFIC
    traffic_begin(name, grid, block);
#endif
    std::unique_lock<std::mutex> lk(p.mu);
    p.grid = grid; p.block = block; p.lds = lds; p.fn = fn; p.ctx = ctx;
    p.next = 0; p.total = total;
    p.active = (uint32_t)p.threads.size();
    p.generation++;
    p.cv_work.notify_all();
    p.cv_done.wait(lk, [&] { return p.active == 0; });
#ifdef HOSTWAVE_TRAFFIC
    traffic_end();
#endif
}

// ---- cross-lane instructions
uint32_t update_dpp(uint32_t old, uint32_t src, uint32_t ctrl, uint32_t row_mask, uint32_t bank_mask, bool bound_ctrl, uint32_t site) {
    rendezvous(site, src);
    const int l = cur->lane, row = l >> 4, in_row = l & 15, base = l & ~15;
    if (!((row_mask >> row) & 1) || !((bank_mask >> (in_row >> 2)) & 1)) return old;
    int s = -1;
    bool valid = true;
    if (ctrl <= 0xFF) s = (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);
    else if (ctrl >= 0x101 && ctrl <= 0x10F) { const int n = ctrl & 15; valid = in_row + n <= 15; s = l + n; }             // row_shl
    else if (ctrl >= 0x111 && ctrl <= 0x11F) { const int n = ctrl & 15; valid = in_row >= n; s = l - n; }                  // row_shr
    else if (ctrl >= 0x121 && ctrl <= 0x12F) { const int n = ctrl & 15; s = base + ((in_row - n) & 15); }                  // row_ror
    else if (ctrl == 0x130) { valid = l + 1 <= 63; s = l + 1; }                                                            // wave_shl:1
    else if (ctrl == 0x134) { s = (l + 1) & 63; }                                                                          // wave_rol:1
    else if (ctrl == 0x138) { valid = l >= 1; s = l - 1; }                                                                 // wave_shr:1
    else if (ctrl == 0x13C) { s = (l - 1) & 63; }                                                                          // wave_ror:1
    else if (ctrl == 0x140) s = base + 15 - in_row;                                                                        // row_mirror
    else if (ctrl == 0x141) s = (l & ~7) | (7 - (l & 7));                                                                  // row_half_mirror
    else if (ctrl == 0x142) { valid = row >= 1; s = base - 1; }                                                            // row_bcast:15
    else if (ctrl == 0x143) { valid = l >= 32; s = 31; }                                                                   // row_bcast:31
    else fail("DPP control not modelled", site);
    return take(s, valid, old, bound_ctrl, site);
}

uint32_t ds_swizzle(uint32_t v, uint32_t pattern, uint32_t site) {
    rendezvous(site, v);
    const int l = cur->lane;
    int s;
    if (pattern & 0x8000) s = (l & ~3) | ((pattern >> (2 * (l & 3))) & 3);
    else {
        const int a = pattern & 0x1F, o = (pattern >> 5) & 0x1F, x = (pattern >> 10) & 0x1F, j = l & 31;
        s = (l & 32) | (((j & a) | o) ^ x);
    }
    return take(s, true, 0, true, site);
}
}  // namespace hw


#ifdef HOSTWAVE_TRAFFIC
// ---------------------------------------------------------------- memory-traffic census (tools/traffic_census.py; never part of the test build)
// The kernel files are compiled with -fsanitize=thread and NOT linked with its runtime: the __tsan_* hooks below see every load and
// store of the device code.  An access to global ("device") memory is filed under its static instruction (the hook's return address),
// its wave and the how-many-th time that lane executes it -- which names one wave-level memory instruction -- and when the workgroup
// ends every such instruction yields: bytes the lanes asked for, distinct 128-byte lines, 64- and 32-byte sectors they lie in.  Summed
// per static instruction this says where a kernel's traffic comes from IF no line survived in a cache between two instructions
// (`line_bytes`), next to the launch's footprint (distinct lines: every line fetched once).  Measured HBM traffic lies in between.
#include <unordered_map>
#include <unordered_set>
#include <dlfcn.h>
namespace hw {
struct WaveInstr { uint32_t useful = 0; bool write = false; bool nt = false; uint32_t seq = 0; uint32_t wave = 0; std::vector<uint64_t> sectors; };      // 32-byte sector numbers touched; seq: order of first touch inside the workgroup
struct SiteTotal { uint64_t instrs = 0, useful = 0, line = 0, s64 = 0, s32 = 0; bool write = false; };
struct LaneOcc {
    std::unordered_map<uintptr_t, uint32_t> n;
    // the lane's previous access: a 16-byte vector load of the device is up to four dword accesses here (the types are 4-byte aligned
    // structs on the host) -- accesses that continue the previous one are folded into it, up to the 16 bytes of a dwordx4
    const uint8_t* last_end = nullptr; uint64_t last_key = 0; uint32_t last_bytes = 0; bool last_write = false;
};
struct Census {
    std::unordered_map<uint64_t, WaveInstr> live;       // key: hash(site, wave, occurrence) of the running workgroup
    std::unordered_map<uint64_t, uintptr_t> site_of;    // same key -> site
    LaneOcc occ[MAX_LANES];
    uint32_t next_seq = 0;
    std::unordered_map<uintptr_t, SiteTotal> totals;    // this worker's share of the launch
    std::unordered_set<uint64_t> lines_r, lines_w;      // footprint (line numbers)
    uint64_t lds_accesses = 0, lds_bytes = 0;
};
static thread_local Census* tc = nullptr;
static std::mutex g_census_mu;
static FILE* g_trace = nullptr;
static uint64_t g_launch_no = 0;
static std::unordered_map<uintptr_t, SiteTotal> g_totals;
static std::unordered_set<uint64_t> g_lines_r, g_lines_w;
static uint64_t g_lds_accesses, g_lds_bytes;
static std::string g_kernel;
static dim3 g_grid, g_block;

static void traffic_begin(const char* name, dim3 grid, dim3 block) {
    if (!g_trace) if (const char* t = getenv("HOSTWAVE_TRACE_OUT")) g_trace = fopen(t, "ab");
    g_launch_no++;
    g_totals.clear(); g_lines_r.clear(); g_lines_w.clear(); g_lds_accesses = g_lds_bytes = 0;
    g_kernel = name ? name : "?"; g_grid = grid; g_block = block;
}

static inline void census_access(const void* addr, uint32_t size, bool write, uintptr_t site) {
    Lane* l = cur;
    if (!l || !size) return;                                             // host code of the same translation unit
    Worker* k = tw;
    const uint8_t* a = (const uint8_t*)addr;
    if (a >= k->stacks && a < k->stacks + STACK_BYTES * MAX_LANES) return;      // a lane's own stack ("registers", spills)
    if ((a >= (const uint8_t*)k && a < (const uint8_t*)(k + 1)) || a == (const uint8_t*)&cur) return;      // the emulator's own state (threadIdx, a rendezvous' results)
    if (!tc) tc = new Census;
    Census& c = *tc;
    if (a >= k->lds && a < k->lds + LDS_STATIC_BYTES + LDS_DYNAMIC_MAX) { c.lds_accesses++; c.lds_bytes += size; return; }
    LaneOcc& lo = c.occ[l->linear];
    uint64_t key;
    if (a == lo.last_end && write == lo.last_write && lo.last_bytes + size <= 16 && c.live.count(lo.last_key)) {
        key = lo.last_key;
        lo.last_bytes += size;
    } else {
        const uint32_t occ = lo.n[site]++;
        key = ((uint64_t)site * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)l->wave->index << 40) ^ ((uint64_t)occ * 0xC2B2AE3D27D4EB4Full);
        lo.last_key = key; lo.last_bytes = size; lo.last_write = write;
    }
    lo.last_end = a + size;
    WaveInstr& w = c.live[key];
    if (w.sectors.empty()) { c.site_of[key] = site; w.seq = c.next_seq++; w.wave = l->wave->index; }
    w.useful += size;
    w.write = write;
    w.nt = w.nt || nt_store_now;
    for (uint64_t s = (uintptr_t)a >> 5; s <= ((uintptr_t)a + size - 1) >> 5; s++) w.sectors.push_back(s);
}

// HOSTWAVE_TRACE_OUT: every workgroup's memory instructions in program order (per wave: the order of first touch), each with the
// 128-byte lines it touches -- what tools/l2_replay.py interleaves over a model of one XCD's L2.  Binary records of uint64:
//   [0xB10C, launch number, block number, n] then n x [wave << 56 | (write | streaming << 1) << 48 | seq << 16 | n_lines, (line << 4 | sector mask)...]
static void trace_block(Worker* k, Census& c) {
    std::vector<const WaveInstr*> order;
    order.reserve(c.live.size());
    for (auto& kv : c.live) order.push_back(&kv.second);
    std::sort(order.begin(), order.end(), [](const WaveInstr* a, const WaveInstr* b) { return a->seq < b->seq; });
    std::vector<uint64_t> rec;
    const Block& b = k->block;
    rec.push_back(0xB10C); rec.push_back(g_launch_no); rec.push_back((uint64_t)b.bid.x + (uint64_t)b.gdim.x * (b.bid.y + (uint64_t)b.gdim.y * b.bid.z)); rec.push_back(order.size());
    for (const WaveInstr* w : order) {
        // (lines as line number << 4 | mask of the line's four 32-byte sectors the instruction touches)
        uint64_t prev = ~0ull; std::vector<uint64_t> lines;
        for (uint64_t sct : w->sectors) {
            if ((sct >> 2) != prev) { prev = sct >> 2; lines.push_back(prev << 4); }
            lines.back() |= 1ull << (sct & 3);
        }
        rec.push_back(((uint64_t)w->wave << 56) | ((uint64_t)(w->write ? (w->nt ? 3 : 1) : 0) << 48) | ((uint64_t)(w->seq & 0xFFFFFFFFu) << 16) | (lines.size() & 0xFFFF));
        rec.insert(rec.end(), lines.begin(), lines.end());
    }
    std::lock_guard<std::mutex> g(g_census_mu);
    fwrite(rec.data(), 8, rec.size(), g_trace);
}

static void traffic_block_end(Worker* k) {
    if (!tc) return;
    Census& c = *tc;
    for (auto& kv : c.live) { WaveInstr& w = kv.second; std::sort(w.sectors.begin(), w.sectors.end()); w.sectors.erase(std::unique(w.sectors.begin(), w.sectors.end()), w.sectors.end()); }
    if (g_trace) trace_block(k, c);
    c.next_seq = 0;
    for (auto& kv : c.live) {
        WaveInstr& w = kv.second;
        uint64_t n32 = w.sectors.size(), n64 = 0, n128 = 0, p64 = ~0ull, p128 = ~0ull;
        for (uint64_t s : w.sectors) {
            if ((s >> 1) != p64) { n64++; p64 = s >> 1; }
            if ((s >> 2) != p128) { n128++; p128 = s >> 2; (w.write ? c.lines_w : c.lines_r).insert(s >> 2); }
        }
        SiteTotal& t = c.totals[c.site_of[kv.first]];
        t.instrs++; t.useful += w.useful; t.line += 128 * n128; t.s64 += 64 * n64; t.s32 += 32 * n32; t.write = w.write;
    }
    c.live.clear(); c.site_of.clear();
    for (uint32_t i = 0; i < k->block.n_lanes; i++) { c.occ[i].n.clear(); c.occ[i].last_end = nullptr; }
    std::lock_guard<std::mutex> g(g_census_mu);
    for (auto& kv : c.totals) {
        SiteTotal& t = g_totals[kv.first];
        t.instrs += kv.second.instrs; t.useful += kv.second.useful; t.line += kv.second.line; t.s64 += kv.second.s64; t.s32 += kv.second.s32; t.write = kv.second.write;
    }
    c.totals.clear();
    g_lines_r.insert(c.lines_r.begin(), c.lines_r.end()); g_lines_w.insert(c.lines_w.begin(), c.lines_w.end());
    c.lines_r.clear(); c.lines_w.clear();
    g_lds_accesses += c.lds_accesses; g_lds_bytes += c.lds_bytes; c.lds_accesses = c.lds_bytes = 0;
}

static void traffic_end() {
    const char* path = getenv("HOSTWAVE_TRAFFIC_OUT");
    if (!path) return;
    FILE* f = fopen(path, "a");
    if (!f) return;
    Dl_info info;
    uintptr_t base = 0;
    const char* lib = "";
    if (dladdr((void*)&traffic_end, &info)) { base = (uintptr_t)info.dli_fbase; lib = info.dli_fname; }
    if (g_trace) fflush(g_trace);
    fprintf(f, "{\"kernel\": \"%s\", \"launch\": %llu, \"grid\": %u, \"block\": %u, \"lib\": \"%s\", \"footprint_read_bytes\": %llu, \"footprint_write_bytes\": %llu, \"lds_lane_accesses\": %llu, \"lds_bytes\": %llu, \"sites\": [",
            g_kernel.c_str(), (unsigned long long)g_launch_no, g_grid.x * g_grid.y * g_grid.z, g_block.x * g_block.y * g_block.z, lib, (unsigned long long)g_lines_r.size() * 128, (unsigned long long)g_lines_w.size() * 128,
            (unsigned long long)g_lds_accesses, (unsigned long long)g_lds_bytes);
    bool first = true;
    for (auto& kv : g_totals) {
        const SiteTotal& t = kv.second;
        fprintf(f, "%s{\"site\": %llu, \"rw\": \"%s\", \"instrs\": %llu, \"useful\": %llu, \"line\": %llu, \"s64\": %llu, \"s32\": %llu}", first ? "" : ", ", (unsigned long long)(kv.first - base),
                t.write ? "w" : "r", (unsigned long long)t.instrs, (unsigned long long)t.useful, (unsigned long long)t.line, (unsigned long long)t.s64, (unsigned long long)t.s32);
        first = false;
    }
    fprintf(f, "]}\n");
    fclose(f);
}
}  // namespace hw

#define HW_SITE ((uintptr_t)__builtin_return_address(0))
extern "C" {
void __tsan_init(void) {}
void __tsan_func_entry(void*) {}
void __tsan_func_exit(void) {}
void __tsan_vptr_update(void**, void*) {}
void __tsan_vptr_read(void**) {}
#define HW_RW(N) \
    void __tsan_read##N(void* a) { hw::census_access(a, N, false, HW_SITE); } \
    void __tsan_write##N(void* a) { hw::census_access(a, N, true, HW_SITE); } \
    void __tsan_unaligned_read##N(void* a) { hw::census_access(a, N, false, HW_SITE); } \
    void __tsan_unaligned_write##N(void* a) { hw::census_access(a, N, true, HW_SITE); }
HW_RW(1) HW_RW(2) HW_RW(4) HW_RW(8) HW_RW(16)
void __tsan_read_range(void* a, unsigned long n) { hw::census_access(a, (uint32_t)n, false, HW_SITE); }
void __tsan_write_range(void* a, unsigned long n) { hw::census_access(a, (uint32_t)n, true, HW_SITE); }
void* __tsan_memcpy(void* d, const void* s, size_t n) { hw::census_access(s, (uint32_t)n, false, HW_SITE); hw::census_access(d, (uint32_t)n, true, HW_SITE + 1); return memcpy(d, s, n); }
void* __tsan_memmove(void* d, const void* s, size_t n) { hw::census_access(s, (uint32_t)n, false, HW_SITE); hw::census_access(d, (uint32_t)n, true, HW_SITE + 1); return memmove(d, s, n); }
void* __tsan_memset(void* d, int v, size_t n) { hw::census_access(d, (uint32_t)n, true, HW_SITE); return memset(d, v, n); }
#define HW_ATOMIC(BITS, T) \
    T __tsan_atomic##BITS##_load(const volatile T* a, int) { hw::census_access((const void*)a, BITS / 8, false, HW_SITE); return __atomic_load_n(a, __ATOMIC_RELAXED); } \
    void __tsan_atomic##BITS##_store(volatile T* a, T v, int) { hw::census_access((const void*)a, BITS / 8, true, HW_SITE); __atomic_store_n(a, v, __ATOMIC_RELAXED); } \
    T __tsan_atomic##BITS##_exchange(volatile T* a, T v, int) { hw::census_access((const void*)a, BITS / 8, true, HW_SITE); return __atomic_exchange_n(a, v, __ATOMIC_RELAXED); } \
    T __tsan_atomic##BITS##_fetch_add(volatile T* a, T v, int) { hw::census_access((const void*)a, BITS / 8, true, HW_SITE); return __atomic_fetch_add(a, v, __ATOMIC_RELAXED); } \
    T __tsan_atomic##BITS##_fetch_sub(volatile T* a, T v, int) { hw::census_access((const void*)a, BITS / 8, true, HW_SITE); return __atomic_fetch_sub(a, v, __ATOMIC_RELAXED); } \
    T __tsan_atomic##BITS##_fetch_and(volatile T* a, T v, int) { hw::census_access((const void*)a, BITS / 8, true, HW_SITE); return __atomic_fetch_and(a, v, __ATOMIC_RELAXED); } \
    T __tsan_atomic##BITS##_fetch_or(volatile T* a, T v, int) { hw::census_access((const void*)a, BITS / 8, true, HW_SITE); return __atomic_fetch_or(a, v, __ATOMIC_RELAXED); } \
    T __tsan_atomic##BITS##_fetch_xor(volatile T* a, T v, int) { hw::census_access((const void*)a, BITS / 8, true, HW_SITE); return __atomic_fetch_xor(a, v, __ATOMIC_RELAXED); } \
    int __tsan_atomic##BITS##_compare_exchange_strong(volatile T* a, T* c, T v, int, int) { hw::census_access((const void*)a, BITS / 8, true, HW_SITE); return __atomic_compare_exchange_n(a, c, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED); } \
    T __tsan_atomic##BITS##_compare_exchange_val(volatile T* a, T c, T v, int, int) { hw::census_access((const void*)a, BITS / 8, true, HW_SITE); __atomic_compare_exchange_n(a, &c, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED); return c; } \
    int __tsan_atomic##BITS##_compare_exchange_weak(volatile T* a, T* c, T v, int, int) { hw::census_access((const void*)a, BITS / 8, true, HW_SITE); return __atomic_compare_exchange_n(a, c, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED); }
HW_ATOMIC(8, uint8_t) HW_ATOMIC(16, uint16_t) HW_ATOMIC(32, uint32_t) HW_ATOMIC(64, uint64_t)
void __tsan_atomic_thread_fence(int) {}
void __tsan_atomic_signal_fence(int) {}
}
#endif

// ---------------------------------------------------------------- the fake runtime: host memory, everything synchronous
struct hw_stream { int id; };
struct hw_event { std::chrono::steady_clock::time_point t; };

extern "C" {
hipError_t hipMalloc(void** p, size_t n) {
    const size_t sz = (n + 255 + 256) & ~(size_t)255;
    *p = aligned_alloc(256, sz);
    if (*p) memset(*p, 0xCD, sz);                  // device memory is not zero when it is handed out
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostRegister(void*, size_t, unsigned) { return hipSuccess; }
hipError_t hipHostUnregister(void*) { return hipSuccess; }
hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned) { *dev = host; return hipSuccess; }
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* at, const void* p) { at->type = hipMemoryTypeHost; at->device = 0; at->devicePointer = (void*)p; at->hostPointer = (void*)p; return hipSuccess; }
hipError_t hipMemGetInfo(size_t* fr, size_t* total) { *fr = (size_t)8 << 30; *total = (size_t)8 << 30; return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipMemcpyToSymbol(void* sym, const void* src, size_t n) { memcpy(sym, src, n); return hipSuccess; }
hipError_t hipMemcpyFromSymbol(void* dst, const void* sym, size_t n) { memcpy(dst, sym, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new hw_stream{1}; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* st) { *st = hipStreamCaptureStatusNone; return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new hw_event{std::chrono::steady_clock::now()}; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int) { if (a == hipDeviceAttributeMultiprocessorCount) { *v = 256; return hipSuccess; } return hipErrorInvalidValue; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hostwave error"; }
// for the tests: how many rendezvous were completed with part of a wave (divergent code) since the process began
unsigned long long hostwave_divergent_completions(void) { return hw::g_subset_completions.load(); }
}
