// Implicit-GEMM convolution, LDS-DMA variant ("v2") for the FLOP-dominant layers.
//
// Same math, layouts and epilogue as conv_igemm.hip, restricted to channel counts that are multiples
// of one K chunk (BK = 128 bytes per tile row: 64 bf16 / 32 f32), i.e. every K chunk lies inside ONE
// filter tap and ONE of the two sources.  That makes the tap / source selection wave-uniform and lets
// both tiles be staged with `global_load_lds_dwordx4` (HBM/L2 -> LDS, no VGPR round trip, no ds_write):
//
//   * an LDS-DMA instruction writes lane-linear: base + lane*16 B.  A wave instruction therefore fills
//     8 tile rows x 8 slots of 16 B.  The bank-conflict swizzle (slot ^= (row>>1)&7) cannot be applied
//     to the destination, so it is applied to the per-lane SOURCE k-group and again on the ds_read_b128
//     fragment reads (same involution on both sides).
//   * zero padding / ragged edges: lanes whose source is outside the image or beyond Cout read from a
//     16-byte zero page in global memory instead (the DMA cannot write constants).
//   * two LDS stages (2 x 32 KiB for 128x128x64 bf16): the DMA of chunk k+1 is issued right after the
//     single barrier of chunk k and overlaps its 16 MFMA 32x32x16 per wave; `s_waitcnt vmcnt(0)` +
//     barrier at the top of the next iteration is the only synchronisation (2 workgroups per CU).
//   * XCD-aware workgroup order: each XCD walks a contiguous range of M tiles, N tiles of one M tile
//     back to back, so re-reads of the activation halo hit that XCD's L2.
#include "common.h"

#ifndef GVFI_HOSTSIM
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma_bf16_32x32x16(const uint4& a, const uint4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_f32_32x32x2(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __attribute__((aligned(16))) unsigned int gvfi_zero_page[16];   // zero-initialised
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
__device__ __forceinline__ void glds_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#else
static unsigned int gvfi_zero_page[16];
static inline void glds16(const void* gsrc, unsigned char* lds_wave_base) { emu_glds16(gsrc, lds_wave_base); }
static inline void glds_wait() {}
#endif

template <typename T> struct Mma2;
template <> struct Mma2<bf16_t> {
    static __device__ __forceinline__ void run(f32x16& acc, const uint4& a, const uint4& b) {
        acc = mfma_bf16_32x32x16(a, b, acc);
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

struct ConvArgs2 {
    gvfi_conv_params p;
    int chunks0;     // K chunks per tap that come from source 0  (c0 / BKE)
    int chunks_tap;  // K chunks per tap                             ((c0+c1) / BKE)
    int KT;          // total K chunks = KH*KW*chunks_tap
    int Mg;          // output pixels per weight group
    int MT, NT;      // tiles
    int per_xcd;     // ceil(MT*NT / 8)
    int dbg;         // ablation switches for profiling (algo >> 4): 1 = no A DMA, 2 = no B DMA, 4 = no MFMA
    long long Ktot;  // weight row length in elements
};

// ---------------------------------------------------------------- LDS-staged epilogue
// The accumulators (lane = cout, 16 pixels per lane) are first written to LDS as a row-major fp32
// [rows][BN] tile (the staging buffers are free after the K loop), then every thread takes groups of
// 8 consecutive output channels of one pixel: bias / activation / residual / GRU gate math on 8 values
// and ONE 16-byte store (bf16) or two (f32), fully coalesced along the channel axis.  This keeps the
// register footprint of the epilogue tiny (no spills with 128 accumulators) and replaces 2-byte stores.
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
__device__ __forceinline__ void ld8(const void* base, long long idx, int is_f32, bool bf16_elems, float (&o)[8]) {
    if (is_f32 || !bf16_elems) {
        const float4 a = *(const float4*)((const float*)base + idx);
        const float4 b = *(const float4*)((const float*)base + idx + 4);
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    } else {
        const uint4 u = *(const uint4*)((const bf16_t*)base + idx);
        o[0] = bf2f((bf16_t)(u.x & 0xffff)); o[1] = bf2f((bf16_t)(u.x >> 16));
        o[2] = bf2f((bf16_t)(u.y & 0xffff)); o[3] = bf2f((bf16_t)(u.y >> 16));
        o[4] = bf2f((bf16_t)(u.z & 0xffff)); o[5] = bf2f((bf16_t)(u.z >> 16));
        o[6] = bf2f((bf16_t)(u.w & 0xffff)); o[7] = bf2f((bf16_t)(u.w >> 16));
    }
}
__device__ __forceinline__ void st8(void* base, long long idx, int is_f32, bool bf16_elems, const float (&v)[8]) {
    if (is_f32 || !bf16_elems) {
        *(float4*)((float*)base + idx) = make_float4(v[0], v[1], v[2], v[3]);
        *(float4*)((float*)base + idx + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
        uint4 u;
        u.x = pack_bf16x2(v[0], v[1]);
        u.y = pack_bf16x2(v[2], v[3]);
        u.z = pack_bf16x2(v[4], v[5]);
        u.w = pack_bf16x2(v[6], v[7]);
        *(uint4*)((bf16_t*)base + idx) = u;
    }
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
            if (vec) ld8(p.res, pix * p.ldr + cout0, p.res_f32, BF, r);
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
        if (vec) st8(p.y, pix * p.ldy + cout0, p.y_f32, BF, v);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (e < n_valid) st_any<T>(p.y, pix * p.ldy + cout0 + e, p.y_f32, v[e]);
        }
    } else if (p.epi_mode == GVFI_EPI_GRU_ZR) {
        const int half = p.Cout >> 1;   // groups never straddle the z / r halves (half % 8 == 0 checked on the host)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gvfi_sigmoid(v[e]);
        if (cout0 < half) {
            if (vec) st8(p.y, pix * p.ldy + cout0, 0, BF, v);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (e < n_valid) st_any<T>(p.y, pix * p.ldy + cout0 + e, 0, v[e]);
            }
        } else {
            const int c0 = cout0 - half;
            float h[8];
            if (vec) ld8(p.aux0, pix * p.lda0 + c0, 0, BF, h);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) h[e] = (e < n_valid) ? ld_any<T>(p.aux0, pix * p.lda0 + c0 + e, 0) : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= h[e];
            if (vec) st8(p.y2, pix * p.ldy2 + c0, 0, BF, v);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (e < n_valid) st_any<T>(p.y2, pix * p.ldy2 + c0 + e, 0, v[e]);
            }
        }
    } else {  // GVFI_EPI_GRU_Q
        float h[8], z[8];
        if (vec) {
            ld8(p.aux0, pix * p.lda0 + cout0, 0, BF, h);
            ld8(p.aux1, pix * p.lda1 + cout0, 0, BF, z);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                h[e] = (e < n_valid) ? ld_any<T>(p.aux0, pix * p.lda0 + cout0 + e, 0) : 0.f;
                z[e] = (e < n_valid) ? ld_any<T>(p.aux1, pix * p.lda1 + cout0 + e, 0) : 0.f;
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (1.f - z[e]) * h[e] + z[e] * tanhf(v[e]);
        if (vec) st8(p.y, pix * p.ldy + cout0, 0, BF, v);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (e < n_valid) st_any<T>(p.y, pix * p.ldy + cout0 + e, 0, v[e]);
        }
    }
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N>
__global__ void __launch_bounds__(64 * WAVES_M * WAVES_N) conv_igemm_glds_kernel(ConvArgs2 a) {
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int VE = Elem<T>::VE;
    constexpr int BKE = 8 * VE;        // elements per K chunk (128 bytes)
    constexpr int RB = 128;            // LDS row bytes
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int A_INSTR = BM / (8 * NW);   // LDS-DMA instructions per wave for the A tile (8 rows each)
    constexpr int B_INSTR = BN / (8 * NW);
    constexpr int STAGE = (BM + BN) * RB;
    static_assert(MI >= 1 && NI >= 1 && A_INSTR >= 1 && B_INSTR >= 1, "tile");

    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];   // double buffered

    const gvfi_conv_params& p = a.p;
    // ---- XCD-aware tile order (blockIdx.x round-robins over the 8 XCDs)
    const int bid = blockIdx.x;
    const int v = (bid & 7) * a.per_xcd + (bid >> 3);
    if (v >= a.MT * a.NT) return;
    const int mt = v / a.NT, nt = v - mt * a.NT;
    const int g = blockIdx.z;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform -> SGPR (M0 bases)
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const long long m_tile0 = (long long)mt * BM;
    const int n0 = nt * BN;
    const int HoWo = p.Ho * p.Wo;
    const T* __restrict__ x0 = (const T*)p.x0;
    const T* __restrict__ x1 = (const T*)p.x1;
    const T* __restrict__ wg = (const T*)p.w + (long long)g * p.w_group_stride;
    const unsigned char* zero = (const unsigned char*)gvfi_zero_page;

    const int lrow = lane >> 3;   // row inside an 8-row DMA group
    const int lslot = lane & 7;   // destination slot

    // per-thread A rows (one per DMA instruction of this wave): pixel offset of tap (0,0) and the
    // top-left input coordinate; rows beyond the tile's valid range get an always-out-of-bounds y.
    int a_iy0[A_INSTR], a_ix0[A_INSTR], a_pix[A_INSTR], a_koff[A_INSTR];
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
        const int row = (wave * A_INSTR + i) * 8 + lrow;
        const long long m = m_tile0 + row;
        const bool ok = m < a.Mg;
        const long long mm = (long long)g * a.Mg + (ok ? m : 0);
        const int n = (int)(mm / HoWo);
        const int rem = (int)(mm - (long long)n * HoWo);
        const int oy = rem / p.Wo;
        const int ox = rem - oy * p.Wo;
        const int iy0 = oy * p.stride - p.pad_h, ix0 = ox * p.stride - p.pad_w;
        a_iy0[i] = ok ? iy0 : -(1 << 28);
        a_ix0[i] = ix0;
        a_pix[i] = (n * p.H + iy0) * p.W + ix0;
        a_koff[i] = (lslot ^ ((row >> 1) & 7)) * VE;   // source k-group (element offset inside the chunk)
    }
    const T* b_src[B_INSTR];
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) {
        const int row = (wave * B_INSTR + i) * 8 + lrow;
        const int n = n0 + row;
        if (p.w_layout == 1)   // chunk-major, pre-swizzled image: [K chunk][Cout][128 B] == the LDS image, linear
            b_src[i] = (n < p.Cout) ? wg + ((long long)n * 8 + lslot) * VE : (const T*)nullptr;
        else
            b_src[i] = (n < p.Cout) ? wg + (long long)n * a.Ktot + (lslot ^ ((row >> 1) & 7)) * VE : (const T*)nullptr;
    }

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31;
    const int fhalf = lane >> 5;

    int kh = 0, kw = 0, ck = 0;   // wave-uniform K walker of the NEXT chunk to stage
    // per-chunk uniform staging state (set by stage_begin, consumed by stage_piece)
    const T* st_xs = x0;
    int st_ld = 0, st_cbase = 0, st_tapoff = 0, st_kh = 0, st_kw = 0;
    long long st_kbase = 0;
    unsigned char* st_sa = smem;
    auto stage_begin = [&](int kt, int buf) {
        st_sa = smem + buf * STAGE;
        const bool from0 = ck < a.chunks0;
        st_xs = from0 ? x0 : x1;
        st_ld = from0 ? p.ld0 : p.ld1;
        st_cbase = (from0 ? ck : ck - a.chunks0) * BKE;
        st_tapoff = kh * p.W + kw;
        st_kh = kh;
        st_kw = kw;
        st_kbase = p.w_layout == 1 ? (long long)kt * p.Cout * BKE : (long long)kt * BKE;
        if (++ck == a.chunks_tap) {
            ck = 0;
            if (++kw == p.KW) { kw = 0; ++kh; }
        }
    };
    // one LDS-DMA instruction (1 KiB per wave): pieces [0, A_INSTR) are A rows, [A_INSTR, A_INSTR+B_INSTR) B rows
    auto stage_piece = [&](int pc) {
        if (pc < A_INSTR) {
            if (a.dbg & 1) return;
            const int i = pc;
            const int iy = a_iy0[i] + st_kh, ix = a_ix0[i] + st_kw;
            const bool ok = ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
            const T* sp = st_xs + (long long)(a_pix[i] + st_tapoff) * st_ld + (st_cbase + a_koff[i]);
            glds16(ok ? (const void*)sp : (const void*)zero, st_sa + (wave * A_INSTR + i) * 1024);
        } else if (pc < A_INSTR + B_INSTR) {
            if (a.dbg & 2) return;
            const int i = pc - A_INSTR;
            const void* sp = b_src[i] ? (const void*)(b_src[i] + st_kbase) : (const void*)zero;
            glds16(sp, st_sa + BM * RB + (wave * B_INSTR + i) * 1024);
        }
    };
    constexpr int NPIECE = A_INSTR + B_INSTR;
    constexpr int PIECES_PER_KK = (NPIECE + 3) / 4;

    stage_begin(0, 0);
#pragma unroll
    for (int pc = 0; pc < NPIECE; ++pc) stage_piece(pc);
    for (int kt = 0; kt < a.KT; ++kt) {
        const int buf = kt & 1;
        glds_wait();        // chunk kt has landed (issued during the previous iteration)
        __syncthreads();    // ... for every wave; and every wave is done reading buffer buf^1
        const bool more = kt + 1 < a.KT;
        if (more) stage_begin(kt + 1, buf ^ 1);
        const unsigned char* sa = smem + buf * STAGE;
        const unsigned char* sb = sa + BM * RB;
        // The DMA pieces of chunk kt+1 are issued BETWEEN the MFMA groups of chunk kt: right after the barrier
        // every wave of the workgroup is at the same point, and a burst of 8 DMA issues per wave (~150 cycles
        // each) would leave the matrix pipe idle; spread out, each issue hides under the previous MFMAs.
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int slot = 2 * kk + fhalf;
            uint4 fa[MI], fb[NI];
            if (!(a.dbg & 4)) {
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int row = wm * WM + i * 32 + frow;
                    fa[i] = *(const uint4*)(sa + row * RB + ((slot ^ ((row >> 1) & 7)) << 4));
                }
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int row = wn * WN + j * 32 + frow;
                    fb[j] = *(const uint4*)(sb + row * RB + ((slot ^ ((row >> 1) & 7)) << 4));
                }
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                if (!(a.dbg & 4)) {
#pragma unroll
                    for (int j = 0; j < NI; ++j) Mma2<T>::run(acc[i][j], fa[i], fb[j]);
                }
                // after each half of the MFMA group issue one DMA piece of the next chunk
                if (more && PIECES_PER_KK >= 1 && i == (MI - 1) / 2) {
                    const int pc = kk * PIECES_PER_KK;
                    if (pc < NPIECE) stage_piece(pc);
                }
                if (more && PIECES_PER_KK >= 2 && i == MI - 1) {
                    const int pc = kk * PIECES_PER_KK + 1;
                    if (pc < NPIECE) stage_piece(pc);
                }
            }
        }
    }

    // ---------------------------------------------------------------- epilogue through LDS (see above)
    constexpr int NT = 64 * NW;
    constexpr int PASS_ROWS_RAW = (2 * STAGE / 4) / BN;
    constexpr int PASS_ROWS = PASS_ROWS_RAW >= BM ? BM : (PASS_ROWS_RAW / 32) * 32;
    constexpr int NPASS = (BM + PASS_ROWS - 1) / PASS_ROWS;
    constexpr int GROUPS_PER_ROW = BN / 8;
    float* cs = (float*)smem;
    const int eY = p.y_f32 ? 4 : (int)sizeof(T);
    // vector path needs 16-byte aligned rows on every tensor the epilogue touches
    const bool vec_all = vec_ok(p.y, p.ldy, eY) && vec_ok(p.res, p.ldr, p.res_f32 ? 4 : (int)sizeof(T)) &&
                         vec_ok(p.y2, p.ldy2, (int)sizeof(T)) && vec_ok(p.aux0, p.lda0, (int)sizeof(T)) &&
                         vec_ok(p.aux1, p.lda1, (int)sizeof(T)) &&
                         (p.bias == nullptr || true);
    // this thread's channel group is fixed across the loop (NT % GROUPS_PER_ROW == 0): preload its constants
    static_assert(NT % GROUPS_PER_ROW == 0, "group index must be loop invariant");
    const int my_cg = tid % GROUPS_PER_ROW;
    const int my_cout0 = n0 + my_cg * 8;
    const int my_valid = (p.Cout - my_cout0) >= 8 ? 8 : (p.Cout - my_cout0 > 0 ? p.Cout - my_cout0 : 0);
    GroupConst gc;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const bool ok = e < my_valid;
        gc.bias[e] = (ok && p.bias) ? p.bias[my_cout0 + e] : 0.f;
        gc.s1[e] = (ok && p.act1 == GVFI_ACT_PRELU) ? p.slope1[my_cout0 + e] : 0.f;
        gc.s2[e] = (ok && p.act2 == GVFI_ACT_PRELU) ? p.slope2[my_cout0 + e] : 0.f;
    }
    __syncthreads();   // every wave is done reading the last staged chunk
#pragma unroll 1
    for (int ps = 0; ps < NPASS; ++ps) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int row0 = wm * WM + i * 32;
            if (row0 / PASS_ROWS != ps) continue;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int col = wn * WN + j * 32 + frow;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 - ps * PASS_ROWS + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
                    cs[row * BN + col] = acc[i][j][r];
                }
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int idx = tid; idx < PASS_ROWS * GROUPS_PER_ROW; idx += NT) {
            const int row = idx / GROUPS_PER_ROW;
            const int cg = my_cg;
            const long long m = m_tile0 + (long long)ps * PASS_ROWS + row;
            const int cout0 = my_cout0;
            if (m >= a.Mg || my_valid == 0) continue;
            const int n_valid = my_valid;
            float vv[8];
            const float4 c0 = *(const float4*)(cs + row * BN + cg * 8);
            const float4 c1 = *(const float4*)(cs + row * BN + cg * 8 + 4);
            vv[0] = c0.x; vv[1] = c0.y; vv[2] = c0.z; vv[3] = c0.w; vv[4] = c1.x; vv[5] = c1.y; vv[6] = c1.z; vv[7] = c1.w;
            epilogue_group<T>(p, gc, vv, cout0, n_valid, (long long)g * a.Mg + m, vec_all && n_valid == 8);
        }
        if (ps + 1 < NPASS) __syncthreads();
    }
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N>
static int launch_glds(const gvfi_conv_params& p, hipStream_t stream) {
    constexpr int BKE = 8 * Elem<T>::VE;
    ConvArgs2 a;
    a.p = p;
    a.chunks0 = p.c0 / BKE;
    a.chunks_tap = (p.c0 + p.c1) / BKE;
    a.KT = p.KH * p.KW * a.chunks_tap;
    a.Ktot = (long long)p.KH * p.KW * (p.c0 + p.c1);
    const int groups = p.groups > 0 ? p.groups : 1;
    a.Mg = (int)(((long long)p.N * p.Ho * p.Wo) / groups);
    a.MT = cdiv(a.Mg, BM);
    a.NT = cdiv(p.Cout, BN);
    a.per_xcd = cdiv((long long)a.MT * a.NT, 8);
    a.dbg = p.algo >> 4;
    dim3 grid(a.per_xcd * 8, 1, groups);
    GVFI_LAUNCH_COOP((conv_igemm_glds_kernel<T, BM, BN, WAVES_M, WAVES_N>), grid, dim3(64 * WAVES_M * WAVES_N), stream, a);
    return (int)hipGetLastError();
}

// returns 1 if the LDS-DMA kernel can run this convolution
extern "C" int gvfi_conv2d_glds_eligible(const gvfi_conv_params* pp) {
    const gvfi_conv_params& p = *pp;
    const int bke = p.dtype == GVFI_F32 ? 32 : 64;
    if (p.c0 <= 0 || (p.c0 % bke) || (p.c1 % bke)) return 0;
    if (p.pad_mode != GVFI_PAD_ZEROS) return 0;
    return 1;
}

extern "C" int gvfi_conv2d_glds(const gvfi_conv_params* pp, void* stream) {
    const gvfi_conv_params& p = *pp;
    if (!gvfi_conv2d_glds_eligible(pp)) return -2;
    if (((uintptr_t)p.x0 & 15) || ((uintptr_t)p.x1 & 15) || ((uintptr_t)p.w & 15)) return -3;
    if (p.groups > 1 && (p.N % p.groups)) return -4;
    // tile width: 256 (8 waves, 256x256) for Cout >= 192 on large images, 128 (4 waves) for Cout > 64, else 64
    const long long M = (long long)p.N * p.Ho * p.Wo / (p.groups > 0 ? p.groups : 1);
    int tile = p.tile_hint;
    if (tile == 0) tile = (p.Cout >= 192 && M >= 256 * 256) ? 256 : (p.Cout > 64 ? 128 : (p.Cout > 32 ? 64 : 32));
    hipStream_t st = (hipStream_t)stream;
    if (p.dtype == GVFI_F32) {
        if (tile >= 256) return launch_glds<float, 256, 256, 2, 4>(p, st);
        if (tile >= 128) return launch_glds<float, 128, 128, 2, 2>(p, st);
        return tile >= 64 ? launch_glds<float, 128, 64, 2, 2>(p, st) : launch_glds<float, 128, 32, 4, 1>(p, st);
    }
    if (tile >= 256) return launch_glds<bf16_t, 256, 256, 2, 4>(p, st);
    if (tile >= 128) return launch_glds<bf16_t, 128, 128, 2, 2>(p, st);
    return tile >= 64 ? launch_glds<bf16_t, 128, 64, 2, 2>(p, st) : launch_glds<bf16_t, 128, 32, 4, 1>(p, st);
}
