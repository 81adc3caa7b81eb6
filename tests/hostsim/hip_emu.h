// TEST INFRASTRUCTURE ONLY.  A tiny fiber-per-lane host emulator for the HIP kernels in
// gimm-vfi_amd/csrc, used by the CPU test-suite (pytest -m "not gpu") to check index math,
// LDS addressing, barrier placement and epilogues before spending GPU minutes.
//
// It is never part of the product: libgimmvfi_hip.so is always the hipcc/gfx950 build and the
// Python host code refuses to run without it.  This header is only seen when a source file is
// compiled with -DGVFI_HOSTSIM by tests/hostsim/build.py.
//
// Model (round 5: fibers instead of one OS thread per lane -- the thread-per-lane form spent most of its time in futex calls
// of 64- / 256-party std::barriers): every GPU thread of a workgroup is a FIBER (a private stack, switched in user space);
// the fibers of one workgroup run on ONE OS thread, round-robin, and change over only at `__syncthreads` / cross-lane
// operations (wave_sync), so a workgroup's execution order is deterministic.  Workgroups of a "coop" launch are handed to a
// small persistent pool of OS threads (the calling thread is one of them), `__shared__` is a thread_local static, i.e. one
// copy per worker, dirty from the previous workgroup like real LDS.  A launch whose workgroups combine through FLOAT atomics
// can ask for one worker (GVFI_EMU_SERIAL) so that its sums keep the block order; GVFI_EMU_THREADS=1 serialises everything.
// "simple" launches (no LDS / barrier / cross-lane) run lanes sequentially and workgroups in parallel.  MFMA builtins are
// emulated with the gfx950 register layouts of /opt/skills/guides/cdna_hip_programming.md section 3.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include <sys/mman.h>
#include <unistd.h>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint4 { uint32_t x, y, z, w; };
struct uint2 { uint32_t x, y; };
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return float2{a, b}; }

using std::isfinite;
#define __builtin_amdgcn_readfirstlane(x) (x)
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
typedef void* hipStream_t;
typedef int hipError_t;
static inline int hipGetLastError() { return 0; }

struct f32x16 {
    float v[16];
    float& operator[](int i) { return v[i]; }
    const float& operator[](int i) const { return v[i]; }
};

// ---- fibers ---------------------------------------------------------------------------------
// gvfi_emu_switch(save, load): push the callee-saved registers, store the stack pointer to *save, continue on `load`
// (x86-64 System V; one weak, hidden copy per translation unit).
#if !defined(__x86_64__)
#error "tests/hostsim/hip_emu.h: the fiber switch is written for x86-64"
#endif
extern "C" __attribute__((visibility("hidden"))) void gvfi_emu_switch(void** save_sp, void* load_sp);
asm(".pushsection .text\n"
    ".weak gvfi_emu_switch\n"
    ".hidden gvfi_emu_switch\n"
    ".type gvfi_emu_switch,@function\n"
    "gvfi_emu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n"
    "  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n"
    ".size gvfi_emu_switch, .-gvfi_emu_switch\n"
    ".popsection\n");

namespace emu {
constexpr size_t kStackBytes = 512u << 10;      // per lane; only touched pages are ever resident
constexpr int kMaxWaves = 32;

// ---- adversarial LDS-DMA timing (GVFI_EMU_DMA=1, the default; gvfi_emu_set_dma_mode() on the test library) -------------------
// On the GPU an LDS-DMA lands some time between its issue and the `s_waitcnt vmcnt(N)` that covers it; a kernel is correct only
// if it is correct for EVERY such time.  Mode 0 lands it at issue.  Mode 1 plays both extremes at once: the destination is
// poisoned (NaN patterns) at issue -- the old contents are gone as early as possible: a slot re-filled while a slower wave still
// reads it is caught -- and the data arrive only at the wait that covers them, as late as possible: a missing or too lenient
// counted wait, or a read in front of the barrier that publishes a chunk, reads NaNs.  Every lane keeps the FIFO of its
// outstanding operations (vmcnt retires in order); plain vector loads that a kernel includes in its counts are entered with
// GVFI_EMU_VMEM_OP().  Loads the model does not see only make it stricter than the hardware, never more lenient.
// Mode 2 (negative control of the tests): counted waits (N > 0) retire nothing.
struct PendingDma {
    const void* src;          // nullptr: a plain vector-memory operation (occupies a place in the count, moves nothing)
    unsigned char* dst;
};
inline int& dma_mode() {
    static int m = [] { const char* e = std::getenv("GVFI_EMU_DMA"); return e ? std::atoi(e) : 1; }();      // (the strict mode is the default)
    return m;
}
// one workgroup in flight on this OS thread
struct WorkGroup {
    int nt = 0, cur = -1, alive = 0;
    bool reverse = false, depth_first = false, random = false;      // the schedule of this launch (sched_mode())
    uint32_t rng = 1;
    void* main_sp = nullptr;
    std::vector<void*> sp;                  // saved stack pointer per fiber
    std::vector<unsigned char> done;
    std::vector<unsigned char*> stacks;     // mmap'ed once per worker thread, reused by every launch
    std::vector<unsigned> tid3;             // threadIdx (x, y, z) per fiber
    std::vector<const unsigned*> wait_on;   // per lane: the barrier generation it waits to move on from (nullptr: not waiting)
    std::vector<unsigned> wait_seen;
    int blk_arrived = 0;
    unsigned blk_gen = 0;
    int wv_arrived[kMaxWaves], wv_size[kMaxWaves];
    unsigned wv_gen[kMaxWaves];
    unsigned char wv_par[kMaxWaves];        // flips with every completed wave barrier: which MFMA operand buffer is free
    std::vector<uint32_t> xchg;             // per wave: 64 lanes x 16 dwords of exchange scratch + two MFMA operand buffers of the same size
    std::vector<std::deque<PendingDma>> pend;   // (GVFI_EMU_DMA) per lane: vector-memory operations issued and not yet awaited
    void (*call)(const void*) = nullptr;    // the kernel body of the running launch
    const void* ctx = nullptr;
};
// Everything a lane reads often lives in ONE thread_local object: in a dlopen'ed library every thread_local VARIABLE costs a
// __tls_get_addr call per use site, and the cross-lane operations sit in the kernels' inner loops.  (The address of a
// thread_local is fixed for an OS thread; fibers never migrate.)
struct LaneState {
    WorkGroup* wg = nullptr;
    int lane = 0, wave = 0;                  // of the running fiber
    dim3 tid, bid, bdim, gdim;               // threadIdx, blockIdx, blockDim, gridDim
    unsigned char* dyn_smem = nullptr;          // dynamic LDS of the running (coop) launch: one buffer per worker
    unsigned char* lds_base = nullptr;       // conv_mma.h: base of the object lds_address() was last asked about
    bool serial_hint = false;                // GVFI_EMU_SERIAL: the next coop launch of this thread runs on one worker
};
inline constinit thread_local LaneState tl;      // (constant-initialised: no guard / wrapper call on access)
inline uint32_t* wave_scratch() { return tl.wg->xchg.data() + (size_t)tl.wave * 64 * 48; }
// operand buffer of the next MFMA.  An MFMA publishes its operands, meets the wave ONCE and reads; it needs no second
// barrier because the next MFMA publishes into the OTHER buffer, and the one after that cannot publish before every lane
// has arrived at the barrier in between, i.e. has finished reading.  Other cross-lane operations use the first 1024 dwords
// with barriers of their own on both sides.
inline uint32_t* mfma_scratch() { return wave_scratch() + 1024 * (1 + tl.wg->wv_par[tl.wave]); }
inline void lane_vars(WorkGroup& g, int t);
// ---- schedules (GVFI_EMU_SCHED, or gvfi_emu_set_sched() on the test library) -------------------------------------------------
// A kernel with all its barriers computes the same thing under EVERY interleaving of its waves; one that lacks a barrier does
// not.  The default runs the lanes round-robin in thread order, which keeps the waves within one synchronisation step of
// each other and so hides such races.  bit 0: the waves take their turns in REVERSE order (lanes of a wave stay in order: the
// kernels mark intra-wave exchanges with wave-level syncs only where lock-step execution does not already order them for the
// round-robin).  bit 1: DEPTH first -- a lane waiting for its wave hands over to lanes of the SAME wave only, so a wave runs
// ahead alone until a workgroup barrier (or its end) stops it: the largest skew between waves the barriers allow.
// bit 2: RANDOM -- a waiting lane hands over to the first unfinished lane of a wave drawn at random (xorshift, seeded per
// workgroup from GVFI_EMU_SEED and the block index: reproducible): interleavings of three and more waves that the fixed
// orders never produce.  Every mode explores ONE interleaving per workgroup (3 is not a superset of 1 and 2): the suite runs its
// kernel cases under 1, 2, 3 and a random one.
inline uint32_t& sched_seed() {
    static uint32_t v = [] { const char* e = std::getenv("GVFI_EMU_SEED"); return e ? (uint32_t)std::atoi(e) : 1u; }();
    return v;
}
inline int& sched_mode() {
    static int m = [] { const char* e = std::getenv("GVFI_EMU_SCHED"); return e ? std::atoi(e) : 0; }();
    return m;
}
// leave fiber `from` (-1: the worker's own context) for fiber `to` (-1: back to the worker)
inline void switch_to(WorkGroup& g, int from, int to) {
    g.cur = to;
    if (to >= 0) lane_vars(g, to);
    gvfi_emu_switch(from >= 0 ? &g.sp[from] : &g.main_sp, to >= 0 ? g.sp[to] : g.main_sp);
}
[[noreturn]] inline void deadlock(const char* what) {
    std::fprintf(stderr, "hip_emu: deadlock in %s (a lane left the kernel, or took another path, while the others wait)\n", what);
    std::abort();
}
// a lane is runnable when it has not finished and is not waiting, or the barrier it waits for has completed meanwhile
inline bool runnable(const WorkGroup& g, int t) {
    return !g.done[t] && (g.wait_on[t] == nullptr || *(volatile const unsigned*)g.wait_on[t] != g.wait_seen[t]);
}
// the lane that runs next when lane `me` cannot go on: the first RUNNABLE one in the order of the launch's schedule
// (-1: nobody -- every unfinished lane waits for a barrier that cannot complete)
inline int pick_next(WorkGroup& g, int me, bool wave_level) {
    if (wave_level && g.depth_first) {          // own wave first
        const int w = me >> 6, n = g.wv_size[w];
        for (int i = 1, l = me & 63; i < n; ++i) {
            l = l + 1 == n ? 0 : l + 1;
            if (runnable(g, (w << 6) + l)) return (w << 6) + l;
        }
    }
    int t = me;
    if (g.random) {                              // start the search at the head of a wave drawn at random
        g.rng ^= g.rng << 13; g.rng ^= g.rng >> 17; g.rng ^= g.rng << 5;
        const int nw = (g.nt + 63) >> 6;
        t = ((int)(g.rng % (uint32_t)nw) << 6);
        if (t != me && runnable(g, t)) return t;
    }
    for (int i = 0; i < g.nt; ++i) {
        if (g.reverse) {                         // thread order with the waves reversed: next lane of the wave, then wave w - 1
            const int w = t >> 6, l = t & 63;
            if (l + 1 < g.wv_size[w]) t = t + 1;
            else t = (w == 0 ? (g.nt - 1) >> 6 : w - 1) << 6;
        } else t = t + 1 == g.nt ? 0 : t + 1;
        if (t != me && runnable(g, t)) return t;
    }
    return -1;
}
// wait until *gen moves on from `seen`, running the other lanes meanwhile
inline void wait_gen(WorkGroup& g, const unsigned* gen, unsigned seen, const char* what, bool wave_level) {
    const int me = g.cur;
    g.wait_on[me] = gen;
    g.wait_seen[me] = seen;
    while (*(volatile const unsigned*)gen == seen) {
        const int to = pick_next(g, me, wave_level);
        if (to < 0) deadlock(what);
        switch_to(g, me, to);
    }
    g.wait_on[me] = nullptr;
}
inline void wave_sync() {
    WorkGroup& g = *tl.wg;
    const int w = tl.wave;
    const unsigned seen = g.wv_gen[w];
    if (++g.wv_arrived[w] == g.wv_size[w]) { g.wv_arrived[w] = 0; g.wv_par[w] ^= 1; ++g.wv_gen[w]; return; }
    wait_gen(g, &g.wv_gen[w], seen, "a wave-level operation", true);
}
inline void block_sync() {
    WorkGroup& g = *tl.wg;
    const unsigned seen = g.blk_gen;
    if (++g.blk_arrived == g.nt) { g.blk_arrived = 0; ++g.blk_gen; return; }
    wait_gen(g, &g.blk_gen, seen, "__syncthreads", false);
}
inline void dma_retire(int keep) {
    if (tl.wg == nullptr || tl.wg->pend.empty()) return;
    std::deque<PendingDma>& q = tl.wg->pend[tl.wg->cur];
    if (keep > 0 && dma_mode() == 2) return;
    while ((int)q.size() > keep) {
        const PendingDma e = q.front();
        q.pop_front();
        if (e.src != nullptr) std::memcpy(e.dst, e.src, 16);
    }
}
inline void dma_issue(const void* src, unsigned char* dst) {
    if (dma_mode() == 0) { std::memcpy(dst, src, 16); return; }
    static const uint32_t poison[4] = {0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u};     // NaN as bf16, half and float
    std::memcpy(dst, poison, 16);
    tl.wg->pend[tl.wg->cur].push_back(PendingDma{src, dst});
}
inline void vmem_op() {
    if (dma_mode() != 0 && tl.wg != nullptr && !tl.wg->pend.empty()) tl.wg->pend[tl.wg->cur].push_back(PendingDma{nullptr, nullptr});
}
// first frame of every fiber: run the kernel body for this lane, then hand over for good
inline void fiber_entry() {
    WorkGroup& g = *tl.wg;
    g.call(g.ctx);
    dma_retire(0);     // (the end of the kernel: whatever is still in flight lands)
    const int me = g.cur;
    g.done[me] = 1;
    if (--g.alive == 0) switch_to(g, me, -1);
    else {
        const int to = pick_next(g, me, true);
        if (to < 0) deadlock("the end of a lane");
        switch_to(g, me, to);
    }
    std::abort();      // (a finished fiber is never resumed)
}
}  // namespace emu

#define threadIdx (emu::tl.tid)
#define blockIdx (emu::tl.bid)
#define blockDim (emu::tl.bdim)
#define gridDim (emu::tl.gdim)
inline void emu::lane_vars(WorkGroup& g, int t) {
    tl.lane = t & 63;
    tl.wave = t >> 6;
    threadIdx = dim3(g.tid3[3 * t], g.tid3[3 * t + 1], g.tid3[3 * t + 2]);
}

static inline void __syncthreads() { emu::block_sync(); }

template <typename V> static inline V emu_shfl_from(V v, int src_lane) {
    static_assert(sizeof(V) == 4, "32-bit shuffles only");
    uint32_t* s = emu::wave_scratch();
    uint32_t bits;
    std::memcpy(&bits, &v, 4);
    s[emu::tl.lane * 16] = bits;
    emu::wave_sync();
    uint32_t o = s[(src_lane & 63) * 16];
    emu::wave_sync();
    V r;
    std::memcpy(&r, &o, 4);
    return r;
}
template <typename V> static inline V __shfl_xor(V v, int mask) { return emu_shfl_from(v, emu::tl.lane ^ mask); }
template <typename V> static inline V __shfl_down(V v, int d) {
    int src = emu::tl.lane + d;
    return emu_shfl_from(v, src > 63 ? emu::tl.lane : src);
}

static inline float atomicAdd(float* p, float v) {
    uint32_t* u = reinterpret_cast<uint32_t*>(p);
    uint32_t old = __atomic_load_n(u, __ATOMIC_RELAXED), nw;
    float f;
    do {
        std::memcpy(&f, &old, 4);
        f += v;
        std::memcpy(&nw, &f, 4);
    } while (!__atomic_compare_exchange_n(u, &old, nw, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    std::memcpy(&f, &old, 4);
    return f;
}
static inline int atomicExch(int* p, int v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicMax(unsigned* p, unsigned v) {
    unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}

// ---- MFMA emulation (gfx950 layouts) ------------------------------------------------------
// v_mfma_f32_32x32x16_bf16: A lane l holds A[i=l&31][k=8*(l>>5)+0..7]; B lane l holds
// B[k=8*(l>>5)+0..7][j=l&31]; C/D: col=l&31, row=(r&3)+8*(r>>2)+4*(l>>5).
static inline float emu_bf2f(uint16_t v) {
    uint32_t u = ((uint32_t)v) << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
// (every lane converts ITS eight A and eight B values to float once and publishes them -- 16 dwords, the lane's scratch row;
// the products of two bf16 / half values are exact in float, the sums run in ascending k like before)
template <float (*CVT)(uint16_t)> static inline const float* emu_publish_ab(const uint4& a, const uint4& b) {
    float* base = reinterpret_cast<float*>(emu::mfma_scratch());
    float* s = base + emu::tl.lane * 16;
    uint16_t h[16];
    std::memcpy(h, &a, 16);
    std::memcpy(h + 8, &b, 16);
    for (int i = 0; i < 16; ++i) s[i] = CVT(h[i]);
    return base;
}
template <float (*CVT)(uint16_t)> static inline f32x16 emu_mfma_32x32x16(const uint4& a, const uint4& b, f32x16 c) {
    const float* s = emu_publish_ab<CVT>(a, b);
    emu::wave_sync();
    const int l = emu::tl.lane, j = l & 31, hi = l >> 5;
    const float* pb0 = s + j * 16 + 8;
    const float* pb1 = s + (j + 32) * 16 + 8;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        const float* pa0 = s + i * 16;
        const float* pa1 = s + (i + 32) * 16;
        float acc = c[r];
        for (int k = 0; k < 8; ++k) acc += pa0[k] * pb0[k];
        for (int k = 0; k < 8; ++k) acc += pa1[k] * pb1[k];
        c[r] = acc;
    }
    return c;
}
static inline f32x16 mfma_bf16_32x32x16(const uint4& a, const uint4& b, f32x16 c) { return emu_mfma_32x32x16<emu_bf2f>(a, b, c); }
// v_mfma_f32_16x16x32_bf16: A lane l holds A[m=l&15][k=8*(l>>4)+0..7]; B lane l holds B[k=8*(l>>4)+0..7][n=l&15];
// C/D: col n = l&15, row m = 4*(l>>4) + r.
struct f32x4 {
    float v[4];
    float& operator[](int i) { return v[i]; }
    const float& operator[](int i) const { return v[i]; }
};
static inline f32x4 mfma_bf16_16x16x32(const uint4& a, const uint4& b, f32x4 c) {
    const float* s = emu_publish_ab<emu_bf2f>(a, b);
    emu::wave_sync();
    const int l = emu::tl.lane, n = l & 15;
    for (int r = 0; r < 4; ++r) {
        const int m = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int kq = 0; kq < 4; ++kq) {
            const float* pa = s + (m + 16 * kq) * 16;
            const float* pb = s + (n + 16 * kq) * 16 + 8;
            for (int k = 0; k < 8; ++k) acc += pa[k] * pb[k];
        }
        c[r] = acc;
    }
    return c;
}
// v_mfma_f32_32x32x16_f16: the same layouts with IEEE half operands
static inline float emu_h2f(uint16_t v) {
    const uint32_t sg = (uint32_t)(v & 0x8000u) << 16, e = (v >> 10) & 31u, m = v & 0x3ffu;
    uint32_t u;
    float f;
    if (e == 0) {
        f = (float)m * (1.0f / 16777216.0f);
        std::memcpy(&u, &f, 4);
        u |= sg;
    } else {
        u = sg | (e == 31 ? 0x7f800000u | (m << 13) : ((e + 112u) << 23) | (m << 13));
    }
    std::memcpy(&f, &u, 4);
    return f;
}
static inline f32x16 mfma_f16_32x32x16(const uint4& a, const uint4& b, f32x16 c) { return emu_mfma_32x32x16<emu_h2f>(a, b, c); }
// v_mfma_f32_32x32x2_f32: A lane l holds A[i=l&31][k=l>>5]; B lane l holds B[k=l>>5][j=l&31].
static inline f32x16 mfma_f32_32x32x2(float a, float b, f32x16 c) {
    float* s = reinterpret_cast<float*>(emu::mfma_scratch());
    const int l = emu::tl.lane;
    s[l * 2] = a;
    s[l * 2 + 1] = b;
    emu::wave_sync();
    const int j = l & 31, hi = l >> 5;
    const float b0 = s[j * 2 + 1], b1 = s[(j + 32) * 2 + 1];
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        c[r] = std::fmaf(s[(i + 32) * 2], b1, std::fmaf(s[i * 2], b0, c[r]));
    }
    return c;
}

// ---- LDS-DMA emulation: global_load_lds_dwordx4 writes lane-linear at a WAVE-UNIFORM base -----------
#include <cassert>
static inline void emu_glds16(const void* gsrc, unsigned char* lds_wave_base) {
    uint32_t* s = emu::wave_scratch();
    const int l = emu::tl.lane;
    uint64_t b = reinterpret_cast<uint64_t>(lds_wave_base);
    std::memcpy(&s[l * 16], &b, 8);
    emu::wave_sync();
    uint64_t b0;
    std::memcpy(&b0, &s[0], 8);
    if (b0 != b) { std::fprintf(stderr, "emu_glds16: LDS base is not wave-uniform\n"); std::abort(); }
    emu::wave_sync();
    emu::dma_issue(gsrc, lds_wave_base + l * 16);
}

// ---- launches -----------------------------------------------------------------------------
namespace emu {
// persistent workers (leaked at exit on purpose: they sleep on a condition variable; rebuilt after a fork)
struct Pool {
    std::mutex m;
    std::condition_variable cv_job, cv_done;
    unsigned long long job = 0;
    int want = 0, pending = 0, nthreads = 0;
    void (*fn)(void*) = nullptr;
    void* arg = nullptr;
    pid_t pid = 0;
};
inline Pool*& pool_ptr() { static Pool* p = nullptr; return p; }
inline int max_workers() {
    static int n = 0;
    if (n == 0) {
        const char* e = std::getenv("GVFI_EMU_THREADS");
        n = e ? std::atoi(e) : (int)std::min(16u, std::thread::hardware_concurrency());
        if (n < 1) n = 1;
    }
    return n;
}
inline void pool_worker(Pool* p, int id) {
    unsigned long long seen = 0;
    std::unique_lock<std::mutex> lk(p->m);
    for (;;) {
        p->cv_job.wait(lk, [&] { return p->job != seen; });
        seen = p->job;
        if (id >= p->want) continue;
        void (*fn)(void*) = p->fn;
        void* arg = p->arg;
        lk.unlock();
        fn(arg);
        lk.lock();
        if (--p->pending == 0) p->cv_done.notify_all();
    }
}
// run fn(arg) on `n` threads at once (the caller is one of them)
inline void run_on(int n, void (*fn)(void*), void* arg) {
    if (n <= 1) { fn(arg); return; }
    static std::mutex one_job;                   // (launches from two host threads at once take turns: the pool holds one job)
    std::lock_guard<std::mutex> turn(one_job);
    Pool*& p = pool_ptr();
    if (p == nullptr || p->pid != getpid()) {      // (after a fork the parent's workers do not exist here)
        p = new Pool;
        p->pid = getpid();
        p->nthreads = max_workers() - 1;
        for (int i = 0; i < p->nthreads; ++i) std::thread(pool_worker, p, i).detach();
    }
    const int helpers = std::min(n - 1, p->nthreads);
    {
        std::lock_guard<std::mutex> lk(p->m);
        p->fn = fn;
        p->arg = arg;
        p->want = helpers;
        p->pending = helpers;
        ++p->job;
    }
    p->cv_job.notify_all();
    fn(arg);
    std::unique_lock<std::mutex> lk(p->m);
    p->cv_done.wait(lk, [&] { return p->pending == 0; });
}
inline unsigned char* new_stack() {
    const size_t page = 4096;
    void* m = mmap(nullptr, kStackBytes + page, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (m == MAP_FAILED) { std::perror("hip_emu: mmap of a lane stack"); std::abort(); }
    mprotect(m, page, PROT_NONE);      // guard page under the stack
    return (unsigned char*)m + page;
}
inline WorkGroup& my_wg() { static thread_local WorkGroup g; return g; }
inline std::vector<uint32_t>& my_dyn() { static thread_local std::vector<uint32_t> d; return d; }
template <typename F> struct CoopJob {
    dim3 grid, block;
    size_t shm;
    const F* f;
    std::atomic<long> next{0};
};
template <typename F> void coop_worker(void* jp) {
    CoopJob<F>& job = *(CoopJob<F>*)jp;
    WorkGroup& g = my_wg();                       // (per OS thread, NOT per kernel type: the lane stacks are reused by every launch)
    std::vector<uint32_t>& dyn = my_dyn();
    const dim3 grid = job.grid, block = job.block;
    const int nt = (int)(block.x * block.y * block.z);
    const int nw = (nt + 63) / 64;
    if (nw > kMaxWaves) { std::fprintf(stderr, "hip_emu: workgroup of %d lanes\n", nt); std::abort(); }
    g.nt = nt;
    g.sp.resize(nt);
    g.done.resize(nt);
    g.wait_on.assign(nt, nullptr);
    g.wait_seen.assign(nt, 0u);
    g.tid3.resize(3 * (size_t)nt);
    while ((int)g.stacks.size() < nt) g.stacks.push_back(new_stack());
    g.xchg.resize((size_t)nw * 64 * 48);
    for (int t = 0; t < nt; ++t) {
        g.tid3[3 * t] = t % block.x;
        g.tid3[3 * t + 1] = (t / block.x) % block.y;
        g.tid3[3 * t + 2] = t / (block.x * block.y);
    }
    for (int w = 0; w < nw; ++w) g.wv_size[w] = std::min(64, nt - 64 * w);
    g.call = [](const void* c) { (*(const F*)c)(); };
    g.reverse = (sched_mode() & 1) != 0;
    g.depth_first = (sched_mode() & 2) != 0;
    g.random = (sched_mode() & 4) != 0;
    g.ctx = job.f;
    if (job.shm) {
        dyn.assign(job.shm / 4 + 64, 0xdeadbeefu);
        tl.dyn_smem = reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(dyn.data()) + 63) & ~(uintptr_t)63);
    }
    WorkGroup* outer = tl.wg;
    tl.wg = &g;
    blockDim = block;
    gridDim = grid;
    const long nb = (long)grid.x * grid.y * grid.z;
    for (;;) {
        const long b = job.next.fetch_add(1);
        if (b >= nb) break;
        blockIdx = dim3((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((long)grid.x * grid.y)));
        g.alive = nt;
        g.blk_arrived = 0;
        g.rng = (sched_seed() * 2654435761u) ^ ((uint32_t)b * 40503u + 0x9e3779b9u);
        if (g.rng == 0) g.rng = 1;
        if (dma_mode() != 0) {
            g.pend.resize(nt);
            for (auto& q : g.pend) q.clear();
        } else g.pend.clear();
        for (int w = 0; w < nw; ++w) g.wv_arrived[w] = 0, g.wv_par[w] = 0;
        for (int t = 0; t < nt; ++t) {
            g.done[t] = 0;
            g.wait_on[t] = nullptr;
            // initial frame: six callee-saved registers, the entry as return address, a null return address above it
            // (after the switch's `ret` the stack pointer is 8 below a 16-byte boundary, as after a call)
            void** top = (void**)(g.stacks[t] + kStackBytes - 64);
            for (int i = 0; i < 6; ++i) top[i] = nullptr;
            top[6] = (void*)&fiber_entry;
            top[7] = nullptr;
            g.sp[t] = top;
        }
        switch_to(g, -1, g.reverse ? ((nt - 1) >> 6) << 6 : 0);        // returns when every lane of the workgroup has finished
    }
    tl.wg = outer;
    if (job.shm) {      // the words behind the dynamic LDS must still hold the fill pattern: a write past the end of the allocation
        const uint32_t* w = reinterpret_cast<const uint32_t*>((reinterpret_cast<uintptr_t>(tl.dyn_smem) + job.shm + 3) & ~(uintptr_t)3);
        for (const uint32_t* e = dyn.data() + dyn.size(); w < e; ++w)
            if (*w != 0xdeadbeefu) {
                std::fprintf(stderr, "hip_emu: a kernel wrote behind its %zu bytes of dynamic LDS\n", job.shm);
                std::abort();
            }
    }
    tl.dyn_smem = nullptr;
}
}  // namespace emu
template <typename F> static void emu_launch_coop(dim3 grid, dim3 block, size_t shm, F f) {
    emu::CoopJob<F> job;
    job.grid = grid;
    job.block = block;
    job.shm = shm;
    job.f = &f;
    const long nb = (long)grid.x * grid.y * grid.z;
    const bool serial = emu::tl.serial_hint;
    emu::tl.serial_hint = false;
    emu::run_on(serial ? 1 : (int)std::min<long>(nb, emu::max_workers()), &emu::coop_worker<F>, &job);
}
template <typename F> struct SimpleJob {
    dim3 grid, block;
    const F* f;
    std::atomic<long> next{0};
};
template <typename F> static void emu_simple_worker(void* jp) {
    SimpleJob<F>& job = *(SimpleJob<F>*)jp;
    const dim3 grid = job.grid, block = job.block;
    const long nb = (long)grid.x * grid.y * grid.z;
    blockDim = block;
    gridDim = grid;
    for (;;) {
        const long b = job.next.fetch_add(1);
        if (b >= nb) break;
        blockIdx = dim3((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((long)grid.x * grid.y)));
        for (unsigned tz = 0; tz < block.z; ++tz)
            for (unsigned ty = 0; ty < block.y; ++ty)
                for (unsigned tx = 0; tx < block.x; ++tx) {
                    threadIdx = dim3(tx, ty, tz);
                    (*job.f)();
                }
    }
}
template <typename F> static void emu_launch_simple(dim3 grid, dim3 block, F f) {
    SimpleJob<F> job;
    job.grid = grid;
    job.block = block;
    job.f = &f;
    const long nb = (long)grid.x * grid.y * grid.z;
    emu::run_on((int)std::min<long>(nb, emu::max_workers()), &emu_simple_worker<F>, &job);
}
#define GVFI_LAUNCH_COOP(kernel, grid, block, stream, ...) emu_launch_coop(grid, block, 0, [=] { kernel(__VA_ARGS__); })
#define GVFI_DYN_SMEM(name) unsigned char* name = emu::tl.dyn_smem
#define GVFI_LAUNCH_COOP_SHM(kernel, grid, block, shm, stream, ...) emu_launch_coop(grid, block, (size_t)(shm), [=] { kernel(__VA_ARGS__); })
// the next coop launch combines its workgroups through float atomics: keep the block order (one worker)
#define GVFI_EMU_SERIAL(cond) (emu::tl.serial_hint = (cond))
// a plain vector-memory operation that the kernel's counted waits include (see the adversarial LDS-DMA timing above)
#define GVFI_EMU_VMEM_OP() emu::vmem_op()
extern "C" __attribute__((weak)) void gvfi_emu_set_dma_mode(int m) { emu::dma_mode() = m; }
extern "C" __attribute__((weak)) void gvfi_emu_set_sched(int m) { emu::sched_mode() = m; }
extern "C" __attribute__((weak)) void gvfi_emu_set_seed(int v) { emu::sched_seed() = (uint32_t)v; }
#define GVFI_LAUNCH_SIMPLE(kernel, grid, block, stream, ...) emu_launch_simple(grid, block, [=] { kernel(__VA_ARGS__); })
