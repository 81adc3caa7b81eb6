// TEST INFRASTRUCTURE ONLY.  A tiny thread-per-lane host emulator for the HIP kernels in
// gimm-vfi_amd/csrc, used by the CPU test-suite (pytest -m "not gpu") to check index math,
// LDS addressing, barrier placement and epilogues before spending GPU minutes.
//
// It is never part of the product: libgimmvfi_hip.so is always the hipcc/gfx950 build and the
// Python host code refuses to run without it.  This header is only seen when a source file is
// compiled with -DGVFI_HOSTSIM by tests/hostsim/build.py.
//
// Model: every GPU thread of a workgroup is an OS thread; workgroups run one after another
// ("coop" launches) so `__shared__` can be a plain static; `__syncthreads` / cross-lane ops are
// std::barrier based.  "simple" launches (no LDS/barrier/cross-lane) run lanes sequentially and
// workgroups in parallel.  MFMA builtins are emulated with the gfx950 register layouts of
// /opt/skills/guides/cdna_hip_programming.md section 3.
#pragma once
#include <atomic>
#include <barrier>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
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

namespace emu {
struct BlockShared {
    int nthreads;
    std::barrier<> block_bar;
    std::vector<std::unique_ptr<std::barrier<>>> wave_bar;
    // per-wave exchange scratch: 64 lanes x 16 dwords
    std::vector<uint32_t> xchg;
    explicit BlockShared(int nt) : nthreads(nt), block_bar(nt) {
        int nw = (nt + 63) / 64;
        for (int w = 0; w < nw; ++w) {
            int cnt = std::min(64, nt - 64 * w);
            wave_bar.emplace_back(new std::barrier<>(cnt));
        }
        xchg.resize((size_t)nw * 64 * 16);
    }
};
inline thread_local BlockShared* tl_bs = nullptr;
inline unsigned char* dyn_smem = nullptr;   // dynamic LDS of the running (coop) launch
inline thread_local int tl_lane = 0, tl_wave = 0;
inline uint32_t* wave_scratch() { return tl_bs->xchg.data() + (size_t)tl_wave * 64 * 16; }
inline void wave_sync() { tl_bs->wave_bar[tl_wave]->arrive_and_wait(); }
}  // namespace emu

inline thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

static inline void __syncthreads() { emu::tl_bs->block_bar.arrive_and_wait(); }

template <typename V> static inline V emu_shfl_from(V v, int src_lane) {
    static_assert(sizeof(V) == 4, "32-bit shuffles only");
    uint32_t* s = emu::wave_scratch();
    uint32_t bits;
    std::memcpy(&bits, &v, 4);
    s[emu::tl_lane * 16] = bits;
    emu::wave_sync();
    uint32_t o = s[(src_lane & 63) * 16];
    emu::wave_sync();
    V r;
    std::memcpy(&r, &o, 4);
    return r;
}
template <typename V> static inline V __shfl_xor(V v, int mask) { return emu_shfl_from(v, emu::tl_lane ^ mask); }
template <typename V> static inline V __shfl_down(V v, int d) {
    int src = emu::tl_lane + d;
    return emu_shfl_from(v, src > 63 ? emu::tl_lane : src);
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
static inline f32x16 mfma_bf16_32x32x16(const uint4& a, const uint4& b, f32x16 c) {
    uint32_t* s = emu::wave_scratch();
    const int l = emu::tl_lane;
    std::memcpy(&s[l * 16], &a, 16);
    std::memcpy(&s[l * 16 + 4], &b, 16);
    emu::wave_sync();
    const int j = l & 31, hi = l >> 5;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int k = 0; k < 16; ++k) {
            const uint16_t* pa = reinterpret_cast<const uint16_t*>(&s[(i + 32 * (k >> 3)) * 16]);
            const uint16_t* pb = reinterpret_cast<const uint16_t*>(&s[(j + 32 * (k >> 3)) * 16 + 4]);
            acc += emu_bf2f(pa[k & 7]) * emu_bf2f(pb[k & 7]);
        }
        c[r] = acc;
    }
    emu::wave_sync();
    return c;
}
// v_mfma_f32_16x16x32_bf16: A lane l holds A[m=l&15][k=8*(l>>4)+0..7]; B lane l holds B[k=8*(l>>4)+0..7][n=l&15];
// C/D: col n = l&15, row m = 4*(l>>4) + r.
struct f32x4 {
    float v[4];
    float& operator[](int i) { return v[i]; }
    const float& operator[](int i) const { return v[i]; }
};
static inline f32x4 mfma_bf16_16x16x32(const uint4& a, const uint4& b, f32x4 c) {
    uint32_t* s = emu::wave_scratch();
    const int l = emu::tl_lane;
    std::memcpy(&s[l * 16], &a, 16);
    std::memcpy(&s[l * 16 + 4], &b, 16);
    emu::wave_sync();
    const int n = l & 15;
    for (int r = 0; r < 4; ++r) {
        const int m = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int k = 0; k < 32; ++k) {
            const uint16_t* pa = reinterpret_cast<const uint16_t*>(&s[(m + 16 * (k >> 3)) * 16]);
            const uint16_t* pb = reinterpret_cast<const uint16_t*>(&s[(n + 16 * (k >> 3)) * 16 + 4]);
            acc += emu_bf2f(pa[k & 7]) * emu_bf2f(pb[k & 7]);
        }
        c[r] = acc;
    }
    emu::wave_sync();
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
static inline f32x16 mfma_f16_32x32x16(const uint4& a, const uint4& b, f32x16 c) {
    uint32_t* s = emu::wave_scratch();
    const int l = emu::tl_lane;
    std::memcpy(&s[l * 16], &a, 16);
    std::memcpy(&s[l * 16 + 4], &b, 16);
    emu::wave_sync();
    const int j = l & 31, hi = l >> 5;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int k = 0; k < 16; ++k) {
            const uint16_t* pa = reinterpret_cast<const uint16_t*>(&s[(i + 32 * (k >> 3)) * 16]);
            const uint16_t* pb = reinterpret_cast<const uint16_t*>(&s[(j + 32 * (k >> 3)) * 16 + 4]);
            acc += emu_h2f(pa[k & 7]) * emu_h2f(pb[k & 7]);
        }
        c[r] = acc;
    }
    emu::wave_sync();
    return c;
}
// v_mfma_f32_32x32x2_f32: A lane l holds A[i=l&31][k=l>>5]; B lane l holds B[k=l>>5][j=l&31].
static inline f32x16 mfma_f32_32x32x2(float a, float b, f32x16 c) {
    uint32_t* s = emu::wave_scratch();
    const int l = emu::tl_lane;
    std::memcpy(&s[l * 16], &a, 4);
    std::memcpy(&s[l * 16 + 1], &b, 4);
    emu::wave_sync();
    const int j = l & 31, hi = l >> 5;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            float fa, fb;
            std::memcpy(&fa, &s[(i + 32 * k) * 16], 4);
            std::memcpy(&fb, &s[(j + 32 * k) * 16 + 1], 4);
            acc = std::fmaf(fa, fb, acc);
        }
        c[r] = acc;
    }
    emu::wave_sync();
    return c;
}

// ---- LDS-DMA emulation: global_load_lds_dwordx4 writes lane-linear at a WAVE-UNIFORM base -----------
#include <cassert>
static inline void emu_glds16(const void* gsrc, unsigned char* lds_wave_base) {
    uint32_t* s = emu::wave_scratch();
    const int l = emu::tl_lane;
    uint64_t b = reinterpret_cast<uint64_t>(lds_wave_base);
    std::memcpy(&s[l * 16], &b, 8);
    emu::wave_sync();
    uint64_t b0;
    std::memcpy(&b0, &s[0], 8);
    if (b0 != b) { std::fprintf(stderr, "emu_glds16: LDS base is not wave-uniform\n"); std::abort(); }
    emu::wave_sync();
    std::memcpy(lds_wave_base + l * 16, gsrc, 16);
}

// ---- launches -----------------------------------------------------------------------------
template <typename F> static void emu_launch_coop(dim3 grid, dim3 block, F f) {
    const int nt = (int)(block.x * block.y * block.z);
    emu::BlockShared bs(nt);
    std::vector<std::thread> th;
    th.reserve(nt);
    for (int t = 0; t < nt; ++t) {
        th.emplace_back([&, t] {
            emu::tl_bs = &bs;
            emu::tl_lane = t & 63;
            emu::tl_wave = t >> 6;
            blockDim = block;
            gridDim = grid;
            threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            for (unsigned bz = 0; bz < grid.z; ++bz)
                for (unsigned by = 0; by < grid.y; ++by)
                    for (unsigned bx = 0; bx < grid.x; ++bx) {
                        blockIdx = dim3(bx, by, bz);
                        f();
                        bs.block_bar.arrive_and_wait();
                    }
        });
    }
    for (auto& x : th) x.join();
}
template <typename F> static void emu_launch_simple(dim3 grid, dim3 block, F f) {
    const long nb = (long)grid.x * grid.y * grid.z;
    const int nw = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    std::atomic<long> next{0};
    std::vector<std::thread> th;
    for (int w = 0; w < nw; ++w) {
        th.emplace_back([&] {
            blockDim = block;
            gridDim = grid;
            for (;;) {
                long b = next.fetch_add(1);
                if (b >= nb) break;
                blockIdx = dim3((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((long)grid.x * grid.y)));
                for (unsigned tz = 0; tz < block.z; ++tz)
                    for (unsigned ty = 0; ty < block.y; ++ty)
                        for (unsigned tx = 0; tx < block.x; ++tx) {
                            threadIdx = dim3(tx, ty, tz);
                            f();
                        }
            }
        });
    }
    for (auto& x : th) x.join();
}
#define GVFI_LAUNCH_COOP(kernel, grid, block, stream, ...) emu_launch_coop(grid, block, [=] { kernel(__VA_ARGS__); })
#define GVFI_DYN_SMEM(name) unsigned char* name = emu::dyn_smem
#define GVFI_LAUNCH_COOP_SHM(kernel, grid, block, shm, stream, ...)                       \
    do {                                                                                  \
        std::vector<uint32_t> gvfi_dyn_((size_t)(shm) / 4 + 64, 0xdeadbeefu);             \
        emu::dyn_smem = reinterpret_cast<unsigned char*>(                                 \
            (reinterpret_cast<uintptr_t>(gvfi_dyn_.data()) + 63) & ~(uintptr_t)63);       \
        emu_launch_coop(grid, block, [=] { kernel(__VA_ARGS__); });                       \
        emu::dyn_smem = nullptr;                                                          \
    } while (0)
#define GVFI_LAUNCH_SIMPLE(kernel, grid, block, stream, ...) emu_launch_simple(grid, block, [=] { kernel(__VA_ARGS__); })
