// Shared device helpers of the MFMA kernels (conv_igemm_glds.hip, inr_mlp.hip):
// MFMA wrappers, the inline-asm LDS-DMA, bf16 packing and the LDS-staged vector epilogue.
#pragma once
#include "common.h"

#ifndef GVFI_HOSTSIM
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma_bf16_32x32x16(const uint4& a, const uint4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
// v_mfma_f32_16x16x32_bf16: A lane l holds A[m = l&15][8 k-values of group l>>4], B lane l holds B[same k group][n = l&15];
// C/D: col n = l&15, rows m = 4*(l>>4) + r, r = 0..3
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma_bf16_16x16x32(const uint4& a, const uint4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma_f16_32x32x16(const uint4& a, const uint4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_f32_32x32x2(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
static __device__ __attribute__((aligned(16))) unsigned int gvfi_zero_page[16];   // zero-initialised, one per TU
// LDS-DMA issued through inline asm ON PURPOSE: with the builtin, hipcc treats the DMA as a pending LDS write
// and puts `s_waitcnt vmcnt(0)` in front of the next ds_read of the same __shared__ array, which serialises
// the prefetch of chunk k+1 behind the MFMAs of chunk k.  An asm statement is invisible to its waitcnt
// bookkeeping (cdna_hip_programming.md section 5.7); completion is awaited explicitly by glds_wait() +
// __syncthreads() at the top of the K loop.  M0 (LDS destination base) is saved/restored in the statement.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned char* lds_wave_base) {
    const unsigned lds_addr = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds_wave_base);
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_addr)
        : "memory");
}
// Buffer-resource form of the LDS-DMA: the 128-bit descriptor (wave-uniform base of this K chunk) lives in SGPRs,
// each lane only supplies a 32-bit byte offset, and offsets >= num_records return ZERO -- so image padding / ragged
// rows need no zero page and no 64-bit per-lane address arithmetic: an out-of-range lane just uses offset ~0u.
// M0 (LDS destination) is written in the same statement and not restored: nothing else in these kernels uses it.
typedef int gvfi_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ gvfi_i32x4 make_srd(const void* base) {
    const unsigned long long a = (unsigned long long)base;
    gvfi_i32x4 r;
    r.x = (int)(unsigned)(a & 0xffffffffull);
    r.y = (int)(unsigned)((a >> 32) & 0xffffull);   // stride 0
    r.z = 0x7fffff00;                               // num_records (bytes): offsets >= this read as zero
    r.w = 0x00020000;
    return r;
}
// LDS byte address of a __shared__ object (wave-uniform); add plain integers to it for sub-buffers
__device__ __forceinline__ unsigned lds_address(const void* p) {
    return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)p);
}
// soff: wave-uniform byte offset added to the descriptor base (the instruction's SGPR offset operand).  The range check
// of a raw buffer compares the per-lane offset with num_records - soff, so the out-of-range marker below still reads
// zeros.  With it the descriptors are CONSTANT for a kernel and a K chunk only costs one 32-bit offset per operand.
__device__ __forceinline__ void bufdma16(unsigned voff, gvfi_i32x4 srd, unsigned soff, unsigned lds_addr) {
    asm volatile(
        // hipcc pads nothing inside an asm string: an SGPR written by the SALU just before this statement needs 5 wait
        // states before a VMEM instruction reads it (and M0 needs 1): s_mov + s_nop 4 = 6 states.
        "s_mov_b32 m0, %3\n\t"
        "s_nop 4\n\t"
        "buffer_load_dwordx4 %0, %1, %2 offen lds\n\t"
        "s_nop 0"
        :
        : "v"(voff), "s"(srd), "s"(soff), "s"(lds_addr)
        : "memory");
}
#define GVFI_DMA_OOB 0xffffff00u
__device__ __forceinline__ void glds_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// wait until at most N of this wave's LDS-DMA / vector-memory operations are still outstanding (they retire in order)
template <int N> __device__ __forceinline__ void glds_wait_n() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }
#else
static unsigned int gvfi_zero_page[16];
static inline void glds16(const void* gsrc, unsigned char* lds_wave_base) { emu_glds16(gsrc, lds_wave_base); }
struct gvfi_i32x4 { const unsigned char* base; };
static inline gvfi_i32x4 make_srd(const void* base) { return gvfi_i32x4{(const unsigned char*)base}; }
#define GVFI_DMA_OOB 0xffffff00u
// host emulation: an "LDS address" is an offset from a per-thread-block base pointer registered by lds_address()
static inline unsigned lds_address(const void* p) { emu::tl.lds_base = (unsigned char*)p; return 0u; }
static inline void bufdma16(unsigned voff, gvfi_i32x4 srd, unsigned soff, unsigned lds_addr) {
    static const unsigned char zeros[16] = {0};
    emu_glds16(voff >= 0x7fffff00u - soff ? (const void*)zeros : (const void*)(srd.base + soff + voff),
               emu::tl.lds_base + lds_addr);
}
static inline void glds_wait() { emu::dma_retire(0); }
template <int N> static inline void glds_wait_n() { emu::dma_retire(N); }
#endif

// 16-byte buffer store with a per-lane byte offset relative to a wave-uniform base (offsets >= num_records are dropped: a
// masked lane uses GVFI_DMA_OOB and the INSTRUCTION is still issued, so counted s_waitcnt stay exact).  The scalar-offset
// operand stays 0 ON PURPOSE: hipcc (ROCm 7.2) assumes that a buffer store of more than 8 bytes WITH a register soffset may
// have its data registers overwritten by the next VALU instruction, and gfx950 says otherwise -- measured: the first dword of
// one store in ~500 replaced by the address the compiler computed into that register right behind it.  Without a register
// soffset the compiler pads the hazard itself.
#ifndef GVFI_HOSTSIM
typedef __amdgpu_buffer_rsrc_t gvfi_rsrc_t;
__device__ __forceinline__ gvfi_rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffff00, 0x00020000);
}
__device__ __forceinline__ void bufst16(const uint4& v, gvfi_rsrc_t r, unsigned voff) {
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{v.x, v.y, v.z, v.w}, r, (int)voff, 0, 0);
}
// ... and the matching load (out-of-range offsets read zeros)
__device__ __forceinline__ uint4 bufld16(gvfi_rsrc_t r, unsigned voff) {
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    const u32x4_t t = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0);
    uint4 o;
    o.x = t.x; o.y = t.y; o.z = t.z; o.w = t.w;
    return o;
}
// true in every lane iff x is true in every lane of the wave
__device__ __forceinline__ bool wave_all(bool x) { return __builtin_amdgcn_ballot_w64(x) == __builtin_amdgcn_ballot_w64(true); }
#else
struct gvfi_rsrc_t { unsigned char* base; };
static inline gvfi_rsrc_t make_rsrc(const void* base) { return gvfi_rsrc_t{(unsigned char*)base}; }
static inline void bufst16(const uint4& v, gvfi_rsrc_t r, unsigned voff) {
    GVFI_EMU_VMEM_OP();      // (takes its place in the lane's in-order queue of vector-memory operations)
    if (voff < 0x7fffff00u) memcpy(r.base + voff, &v, 16);
}
static inline uint4 bufld16(gvfi_rsrc_t r, unsigned voff) {
    GVFI_EMU_VMEM_OP();
    uint4 o = {0u, 0u, 0u, 0u};
    if (voff < 0x7fffff00u) memcpy(&o, r.base + voff, 16);
    return o;
}
static inline bool wave_all(bool x) {
    uint32_t* s = emu::wave_scratch();
    s[emu::tl.lane * 16] = x ? 1u : 0u;
    emu::wave_sync();
    bool r = true;
    for (int l = 0, n = emu::tl.wg->wv_size[emu::tl.wave]; l < n; ++l) r = r && s[l * 16] != 0u;
    emu::wave_sync();
    return r;
}
#endif

// instruction-scheduling fence: nothing is moved across it (keeps hand-written software pipelining in place)
#ifndef GVFI_HOSTSIM
#define GVFI_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#define GVFI_OPAQUE_V(x) asm volatile("" : "+v"(x))      /* the compiler may not derive anything about x across this point */
#define GVFI_OPAQUE_S(x) asm volatile("" : "+s"(x))      /* ... a wave-uniform x */
#else
#define GVFI_SCHED_BARRIER() ((void)0)
#define GVFI_OPAQUE_V(x) ((void)0)
#define GVFI_OPAQUE_S(x) ((void)0)
#endif

template <typename T> struct Mma2;
template <> struct Mma2<bf16_t> {
    static __device__ __forceinline__ void run(f32x16& acc, const uint4& a, const uint4& b) {
        acc = mfma_bf16_32x32x16(a, b, acc);
    }
};
template <> struct Mma2<f16_t> {
    static __device__ __forceinline__ void run(f32x16& acc, const uint4& a, const uint4& b) {
        acc = mfma_f16_32x32x16(a, b, acc);
    }
};
template <> struct Mma2<float> {
    static __device__ __forceinline__ void run(f32x16& acc, const uint4& a, const uint4& b) {
        acc = mfma_f32_32x32x2(__builtin_bit_cast(float, a.x), __builtin_bit_cast(float, b.x), acc);
        acc = mfma_f32_32x32x2(__builtin_bit_cast(float, a.y), __builtin_bit_cast(float, b.y), acc);
        acc = mfma_f32_32x32x2(__builtin_bit_cast(float, a.z), __builtin_bit_cast(float, b.z), acc);
        acc = mfma_f32_32x32x2(__builtin_bit_cast(float, a.w), __builtin_bit_cast(float, b.w), acc);
    }
};


// median of three (v_med3_f32)
__device__ __forceinline__ float med3f(float a, float b, float c) {
#ifndef GVFI_HOSTSIM
    return __builtin_amdgcn_fmed3f(a, b, c);
#else
    return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c));
#endif
}
// two floats -> packed bf16x2 (round to nearest even); the native cast lets hipcc emit v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
#ifndef GVFI_HOSTSIM
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    f32x2_t f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_t));
#else
    return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
#endif
}
// two floats -> packed IEEE half x2 (round to nearest even, saturating)
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {
    return (uint32_t)f2h(lo).v | ((uint32_t)f2h(hi).v << 16);
}
// 16-bit element types: pack two / unpack eight values of type T (bf16_t or f16_t)
template <typename T> __device__ __forceinline__ uint32_t pack16x2(float lo, float hi) {
    if constexpr (sizeof(T) == 2 && !__is_same(T, bf16_t)) return pack_f16x2(lo, hi);
    else return pack_bf16x2(lo, hi);
}
template <typename T> __device__ __forceinline__ float cvt16(uint32_t bits16) {
    if constexpr (sizeof(T) == 2 && !__is_same(T, bf16_t)) {
        f16_t h;
        h.v = (uint16_t)bits16;
        return h2f(h);
    } else return bf2f((bf16_t)bits16);
}
template <typename T> __device__ __forceinline__ void unpack16x8(const uint4& u, float (&o)[8]) {
    o[0] = cvt16<T>(u.x & 0xffff); o[1] = cvt16<T>(u.x >> 16);
    o[2] = cvt16<T>(u.y & 0xffff); o[3] = cvt16<T>(u.y >> 16);
    o[4] = cvt16<T>(u.z & 0xffff); o[5] = cvt16<T>(u.z >> 16);
    o[6] = cvt16<T>(u.w & 0xffff); o[7] = cvt16<T>(u.w >> 16);
}
template <typename T>
__device__ __forceinline__ void ld8(const void* base, long long idx, int is_f32, bool bf16_elems, float (&o)[8]) {
    if (is_f32 || !bf16_elems) {
        const float4 a = *(const float4*)((const float*)base + idx);
        const float4 b = *(const float4*)((const float*)base + idx + 4);
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    } else {
        const uint4 u = *(const uint4*)((const uint16_t*)base + idx);
        unpack16x8<T>(u, o);
    }
}
template <typename T>
__device__ __forceinline__ void st8(void* base, long long idx, int is_f32, bool bf16_elems, const float (&v)[8]) {
    if (is_f32 || !bf16_elems) {
        *(float4*)((float*)base + idx) = make_float4(v[0], v[1], v[2], v[3]);
        *(float4*)((float*)base + idx + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
        uint4 u;
        u.x = pack16x2<T>(v[0], v[1]);
        u.y = pack16x2<T>(v[2], v[3]);
        u.z = pack16x2<T>(v[4], v[5]);
        u.w = pack16x2<T>(v[6], v[7]);
        *(uint4*)((uint16_t*)base + idx) = u;
    }
}
// hardware-rate sigmoid / tanh for the bf16 path (v_exp_f32 + v_rcp_f32, ~1 ulp each: far inside bf16 rounding);
// the fp32 validation mode keeps expf / tanhf
#ifndef GVFI_HOSTSIM
__device__ __forceinline__ float fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float fast_tanh(float x) {
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177792681f * x));
}
// GELU(t) = t/2 (1 + erf(t/sqrt 2)) with Abramowitz-Stegun 7.1.26 for erf (|error| <= 1.5e-7, one v_exp + one v_rcp, ~14
// operations; erff costs ~35 and made the 229 k-row transformer MLPs of GIMM-VFI-F VALU-bound in their epilogue).  For negative
// arguments 1 + erf is evaluated as the complementary term directly, so there is no cancellation.
__device__ __forceinline__ float fast_gelu(float t) {
    const float z = fabsf(t) * 0.70710678118654752f;
    const float k = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    const float poly = k * (0.254829592f + k * (-0.284496736f + k * (1.421413741f + k * (-1.453152027f + k * 1.061405429f))));
    const float e = poly * __builtin_amdgcn_exp2f(-1.44269504088896341f * z * z);        // 1 - erf(z)
    return 0.5f * t * (t >= 0.f ? 2.0f - e : e);
}
#else
static inline float fast_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
static inline float fast_tanh(float x) { return tanhf(x); }
static inline float fast_gelu(float t) {
    const float z = fabsf(t) * 0.70710678118654752f;
    const float k = 1.0f / (1.0f + 0.3275911f * z);
    const float poly = k * (0.254829592f + k * (-0.284496736f + k * (1.421413741f + k * (-1.453152027f + k * 1.061405429f))));
    const float e = poly * expf(-z * z);
    return 0.5f * t * (t >= 0.f ? 2.0f - e : e);
}
#endif
__device__ __forceinline__ void unpack_bf16x8(const uint4& u, float (&o)[8]) {
    o[0] = bf2f((bf16_t)(u.x & 0xffff)); o[1] = bf2f((bf16_t)(u.x >> 16));
    o[2] = bf2f((bf16_t)(u.y & 0xffff)); o[3] = bf2f((bf16_t)(u.y >> 16));
    o[4] = bf2f((bf16_t)(u.z & 0xffff)); o[5] = bf2f((bf16_t)(u.z >> 16));
    o[6] = bf2f((bf16_t)(u.w & 0xffff)); o[7] = bf2f((bf16_t)(u.w >> 16));
}
__device__ __forceinline__ bool vec_ok(const void* ptr, int ld, int elem_bytes) {
    return ptr == nullptr || ((((uintptr_t)ptr) & 15) == 0 && ((ld * elem_bytes) & 15) == 0);
}
__device__ __forceinline__ void act8(float (&v)[8], int act, const float* slope, int cout0) {
    if (act == GVFI_ACT_NONE) return;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = apply_act(v[e], act, slope, cout0 + e);
}

// per-thread constants of its channel group (the group index is the same in every iteration of the
// epilogue loop because GROUPS_PER_ROW divides the thread count): bias and PReLU slopes are loaded once
struct GroupConst {
    float bias[8], s1[8], s2[8];
};
__device__ __forceinline__ void act8s(float (&v)[8], int act, const float (&s)[8]) {
    // one branch per 8 values, never per element
    switch (act) {
        case GVFI_ACT_NONE: break;
        case GVFI_ACT_RELU:
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            break;
        case GVFI_ACT_LRELU:
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.1f * v[e];
            break;
        case GVFI_ACT_PRELU:
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : s[e] * v[e];
            break;
        case GVFI_ACT_SIGMOID:
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = gvfi_sigmoid(v[e]);
            break;
        case GVFI_ACT_TANH:
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tanhf(v[e]);
            break;
        case GVFI_ACT_GELU:
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752f));
            break;
        default:
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = sinf(v[e]);
            break;
    }
}

// one group of 8 channels [cout0, cout0+8) of output pixel `pix`; `n_valid` channels are real
template <typename T>
__device__ __forceinline__ void epilogue_group(const gvfi_conv_params& p, const GroupConst& gc, float (&v)[8], int cout0,
                                               int n_valid, long long pix, bool vec) {
    constexpr bool BF = sizeof(T) == 2;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += gc.bias[e];
    if (p.epi_mode == GVFI_EPI_STD) {
        act8s(v, p.act1, gc.s1);
        if (p.res) {
            float r[8];
            if (vec) ld8<T>(p.res, pix * p.ldr + cout0, p.res_f32, BF, r);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) r[e] = (e < n_valid) ? ld_any<T>(p.res, pix * p.ldr + cout0 + e, p.res_f32) : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += r[e];
        }
        act8s(v, p.act2, gc.s2);
        if (p.out_scale != 1.0f) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= p.out_scale;
        }
        if (vec) st8<T>(p.y, pix * p.ldy + cout0, p.y_f32, BF, v);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (e < n_valid) st_any<T>(p.y, pix * p.ldy + cout0 + e, p.y_f32, v[e]);
        }
    } else if (p.epi_mode == GVFI_EPI_GRU_ZR) {
        const int half = p.Cout >> 1;   // groups never straddle the z / r halves (half % 8 == 0 checked on the host)
        if (p.res) {   // pre-activation term hoisted out of the recurrence (the context part of the gate convolution)
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (e < n_valid) v[e] += ld_any<T>(p.res, pix * p.ldr + cout0 + e, p.res_f32);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gvfi_sigmoid(v[e]);
        const int sf = p.state_f32;     // float recurrent state (z, h), see gvfi_conv_params.state_f32
        if (cout0 < half) {
            if (vec) st8<T>(p.y, pix * p.ldy + cout0, sf, BF, v);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (e < n_valid) st_any<T>(p.y, pix * p.ldy + cout0 + e, sf, v[e]);
            }
        } else {
            const int c0 = cout0 - half;
            float h[8];
            if (vec) ld8<T>(p.aux0, pix * p.lda0 + c0, sf, BF, h);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) h[e] = (e < n_valid) ? ld_any<T>(p.aux0, pix * p.lda0 + c0 + e, sf) : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= h[e];
            if (vec) st8<T>(p.y2, pix * p.ldy2 + c0, 0, BF, v);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (e < n_valid) st_any<T>(p.y2, pix * p.ldy2 + c0 + e, 0, v[e]);
            }
        }
    } else {  // GVFI_EPI_GRU_Q
        if (p.res) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (e < n_valid) v[e] += ld_any<T>(p.res, pix * p.ldr + cout0 + e, p.res_f32);
        }
        const int sf = p.state_f32;
        float h[8], z[8];
        if (vec) {
            ld8<T>(p.aux0, pix * p.lda0 + cout0, sf, BF, h);
            ld8<T>(p.aux1, pix * p.lda1 + cout0, sf, BF, z);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                h[e] = (e < n_valid) ? ld_any<T>(p.aux0, pix * p.lda0 + cout0 + e, sf) : 0.f;
                z[e] = (e < n_valid) ? ld_any<T>(p.aux1, pix * p.lda1 + cout0 + e, sf) : 0.f;
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (1.f - z[e]) * h[e] + z[e] * tanhf(v[e]);
        if (vec) st8<T>(p.y, pix * p.ldy + cout0, 0, BF, v);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (e < n_valid) st_any<T>(p.y, pix * p.ldy + cout0 + e, 0, v[e]);
        }
        if (sf && p.y2 != nullptr) {   // the float state beside its bf16 operand copy
            if (vec) st8<T>(p.y2, pix * p.ldy2 + cout0, 1, BF, v);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (e < n_valid) st_any<T>(p.y2, pix * p.ldy2 + cout0 + e, 1, v[e]);
            }
        }
    }
}

