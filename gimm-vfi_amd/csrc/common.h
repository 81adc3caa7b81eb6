// Shared device/host helpers for the GIMM-VFI HIP kernels (gfx950 / CDNA4 only).
//
// The same sources are also compiled for the host by tests/hostsim (a lane-level
// emulator used ONLY by the CPU test-suite to check index math before a GPU run);
// GVFI_HOSTSIM selects that build.  The product library is always the hipcc build.
#pragma once

#ifdef GVFI_HOSTSIM
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#endif

#include <stdint.h>
#include <string.h>
#include "../../include/gimmvfi_hip.h"

#ifndef GVFI_HOSTSIM
// (emulator only: a launch whose workgroups meet in float atomics keeps its block order; nothing on the device)
#define GVFI_EMU_SERIAL(cond) ((void)0)
// (emulator only: a plain vector load that a kernel's counted s_waitcnt includes -- tests/hostsim's adversarial LDS-DMA timing)
#define GVFI_EMU_VMEM_OP() ((void)0)
// simple = no LDS / barriers / cross-lane ops;  coop = anything else.  Identical on device.
#define GVFI_LAUNCH_SIMPLE(kernel, grid, block, stream, ...) \
    hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__)
#define GVFI_LAUNCH_COOP(kernel, grid, block, stream, ...) \
    hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__)
// dynamic LDS (up to the full 160 KiB of a CU needs the opt-in attribute)
#define GVFI_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#define GVFI_LAUNCH_COOP_SHM(kernel, grid, block, shm, stream, ...)                                         \
    do {                                                                                                    \
        static int gvfi_attr_set_[16] = {0}; /* per device: the attribute belongs to the device's code object */ \
        int gvfi_dev_ = 0;                                                                                  \
        (void)hipGetDevice(&gvfi_dev_);                                                                     \
        gvfi_dev_ &= 15;                                                                                    \
        if (gvfi_attr_set_[gvfi_dev_] < (int)(shm)) {                                                       \
            hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(shm)); \
            gvfi_attr_set_[gvfi_dev_] = (int)(shm);                                                         \
        }                                                                                                   \
        hipLaunchKernelGGL(kernel, grid, block, shm, stream, __VA_ARGS__);                                  \
    } while (0)
#endif

typedef uint16_t bf16_t;

__host__ __device__ __forceinline__ float bf2f(bf16_t v) {
    union { uint32_t u; float f; } c;
    c.u = ((uint32_t)v) << 16;
    return c.f;
}
__host__ __device__ __forceinline__ bf16_t f2bf(float f) {
    union { uint32_t u; float f; } c;
    c.f = f;
    uint32_t u = c.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);                                           // round to nearest even
    return (bf16_t)(u >> 16);
}

// IEEE half as a storage type of its own (bf16_t is a plain uint16_t): conversions saturate at +-65504
struct f16_t { uint16_t v; };
__host__ __device__ __forceinline__ float h2f(f16_t h) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (float)__builtin_bit_cast(_Float16, h.v);
#else
    const uint32_t s = (uint32_t)(h.v & 0x8000u) << 16, e = (h.v >> 10) & 31u, m = h.v & 0x3ffu;
    union { uint32_t u; float f; } c;
    if (e == 0) {            // zero / subnormal: m * 2^-24
        c.f = (float)m * (1.0f / 16777216.0f);
        c.u |= s;
        return c.f;
    }
    c.u = s | (e == 31 ? 0x7f800000u | (m << 13) : ((e + 112u) << 23) | (m << 13));
    return c.f;
#endif
}
__host__ __device__ __forceinline__ f16_t f2h(float f) {
    f16_t r;
#if defined(__HIP_DEVICE_COMPILE__)
    // saturate; v_med3_f32 returns min3 when an input is NaN (-65504 here), so NaN is put back explicitly: an overflow or
    // NaN inside the half-precision decoder must stay visible downstream, as it does in the bf16 / float paths
    const float sat = __builtin_amdgcn_fmed3f(f, -65504.f, 65504.f);
    r.v = __builtin_bit_cast(uint16_t, (_Float16)(f != f ? f : sat));   // v_cvt_f16_f32, round to nearest even
    return r;
#else
    if (f != f) { r.v = 0x7e00; return r; }
    f = f < -65504.f ? -65504.f : (f > 65504.f ? 65504.f : f);
    union { uint32_t u; float f; } c;
    c.f = f;
    const uint32_t s = (c.u >> 16) & 0x8000u;
    c.u &= 0x7fffffffu;
    const float a = c.f;
    if (a < 6.103515625e-05f) {                                   // subnormal half: round a * 2^24 to nearest even
        const float sc = a * 16777216.0f;
        uint32_t q = (uint32_t)sc;
        const float fr = sc - (float)q;
        if (fr > 0.5f || (fr == 0.5f && (q & 1u))) ++q;
        r.v = (uint16_t)(s | q);
        return r;
    }
    uint32_t u = c.u;
    u += 0xfffu + ((u >> 13) & 1u);                               // round the 13 dropped bits to nearest even
    const uint32_t e = (u >> 23) - 112u, m = (u >> 13) & 0x3ffu;
    r.v = (uint16_t)(s | (e << 10) | m);
    return r;
#endif
}

// element traits: T = float, bf16_t or f16_t
template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int VE = 4;  // elements per 16-byte vector
    __host__ __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __host__ __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int VE = 8;
    __host__ __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    __host__ __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

template <> struct Elem<f16_t> {
    static constexpr int VE = 8;
    __host__ __device__ static __forceinline__ float ld(const f16_t* p) { return h2f(*p); }
    __host__ __device__ static __forceinline__ void st(f16_t* p, float v) { *p = f2h(v); }
};

// load / store one element of a tensor whose dtype is a runtime flag (f32 = 1)
template <typename T> __host__ __device__ __forceinline__ float ld_any(const void* p, long i, int is_f32) {
    return is_f32 ? ((const float*)p)[i] : Elem<T>::ld(((const T*)p) + i);
}
template <typename T> __host__ __device__ __forceinline__ void st_any(void* p, long i, int is_f32, float v) {
    if (is_f32) ((float*)p)[i] = v; else Elem<T>::st(((T*)p) + i, v);
}

__device__ __forceinline__ float gvfi_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float apply_act(float v, int act, const float* slope, int c) {
    switch (act) {
        case GVFI_ACT_RELU: return v > 0.f ? v : 0.f;
        case GVFI_ACT_LRELU: return v > 0.f ? v : 0.1f * v;
        case GVFI_ACT_PRELU: return v > 0.f ? v : slope[c] * v;
        case GVFI_ACT_SIGMOID: return gvfi_sigmoid(v);
        case GVFI_ACT_TANH: return tanhf(v);
        case GVFI_ACT_SIN: return sinf(v);
        case GVFI_ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
        default: return v;
    }
}

__host__ __device__ __forceinline__ int reflect_idx(int i, int n) {
    // torch 'reflect' padding (no edge repeat); valid for -n < i < 2n-1
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

// torch upsample_bilinear2d source index, align_corners=False (ATen UpSample.h
// area_pixel_compute_source_index + guard_index_and_lambda)
struct Lerp { int i0, i1; float w0, w1; };
__device__ __forceinline__ Lerp src_index(int d, float rscale, int n) {
    float s = rscale * (d + 0.5f) - 0.5f;
    if (s < 0.f) s = 0.f;
    int i0 = (int)s;
    if (i0 > n - 1) i0 = n - 1;
    Lerp r;
    r.i0 = i0;
    r.i1 = i0 + (i0 < n - 1 ? 1 : 0);
    float l1 = s - (float)i0;
    l1 = l1 < 0.f ? 0.f : (l1 > 1.f ? 1.f : l1);
    r.w1 = l1;
    r.w0 = 1.f - l1;
    return r;
}

// InstanceNorm statistics (sum, sum of squares per image and channel) are accumulated as 64-bit FIXED-POINT integers: integer
// addition is associative, so the result does not depend on the order in which the workgroups' atomics arrive -- the float
// atomics they replace (round 6) were the one run-to-run freedom of the whole path (1e-2 px of flow after RAFT's 20 iterations).
// A workgroup's partial sums are float (fixed order inside the workgroup); 2^-24 / 2^-20 is below their own rounding.
// Layout: stats[(n * C + c) * 2 + {0, 1}] as long long (16 bytes per image and channel), zeroed by the caller.
#define GVFI_STATS_Q0 16777216.0f
#define GVFI_STATS_Q1 1048576.0f
__device__ __forceinline__ void gvfi_stats_add(float* stats, long long slot, float s0, float s1) {
    unsigned long long* q = (unsigned long long*)stats + slot * 2;
#ifndef GVFI_HOSTSIM
    atomicAdd(q, (unsigned long long)__float2ll_rn(s0 * GVFI_STATS_Q0));
    atomicAdd(q + 1, (unsigned long long)__float2ll_rn(s1 * GVFI_STATS_Q1));
#else
    __atomic_fetch_add(q, (unsigned long long)llrintf(s0 * GVFI_STATS_Q0), __ATOMIC_RELAXED);
    __atomic_fetch_add(q + 1, (unsigned long long)llrintf(s1 * GVFI_STATS_Q1), __ATOMIC_RELAXED);
#endif
}
__device__ __forceinline__ void gvfi_stats_get(const float* stats, long long slot, float& s0, float& s1) {
    const long long* q = (const long long*)stats + slot * 2;
    s0 = (float)((double)q[0] * (1.0 / 16777216.0));
    s1 = (float)((double)q[1] * (1.0 / 1048576.0));
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
// unsigned division by an invariant divisor d (1 <= d < 2^31) for dividends x < 2^31:
//   x / d == (umulhi(x, mul) + x) >> sh     (Granlund-Montgomery round-up method, 33-bit magic minus 2^32)
static inline void gvfi_magic_div(unsigned d, unsigned& mul, unsigned& sh) {
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    sh = l;
    mul = (unsigned)((((1ull << (32 + l)) / d) - (1ull << 32)) + 1);
}

#define GVFI_DISPATCH_T(dtype, ...)                         \
    do {                                                    \
        if ((dtype) == GVFI_F32) { typedef float T; __VA_ARGS__; } \
        else if ((dtype) == GVFI_F16) { typedef f16_t T; __VA_ARGS__; } \
        else { typedef bf16_t T; __VA_ARGS__; }             \
    } while (0)
