// Implicit-GEMM convolution for gfx950 (CDNA4) matrix cores.
//
//   M = output pixels (N*Ho*Wo),  N = Cout,  K = KH*KW*Cin  (Cin = c0 + c1, two NHWC sources)
//
// * A (im2col rows) is gathered straight from the NHWC activation(s): for a fixed filter tap a
//   pixel's channels are contiguous, so every global load is one aligned 16-byte vector.
//   K is walked in units of those 16-byte "k-groups" (tap-major, channel-minor) which makes the
//   same loop serve 1x1, 3x3, 5x5, 7x7, 1x5, 5x1, strided and 2..648-channel convolutions
//   with no wasted MFMA work beyond the last partial K chunk.
// * A and B (weights, [Cout][K] = "B^T") tiles are staged global -> VGPR -> LDS, double buffered,
//   one barrier per K chunk.  LDS rows are BKV*16 bytes; the 16-byte slot index is XOR-swizzled
//   with the row so that the ds_read_b128 fragment reads (16-lane groups, 64 banks) and the
//   ds_write_b128 staging writes are bank-conflict free.
// * 4 waves / workgroup, each wave owns a (BM/WAVES_M) x (BN/WAVES_N) sub-tile as MI x NI
//   32x32 MFMA accumulators: v_mfma_f32_32x32x16_bf16 (bf16) or 4 x v_mfma_f32_32x32x2_f32
//   (exact-f32 validation mode).  Any k-permutation that is identical for A and B is legal,
//   so both element types share the same LDS addressing (lane half = 16-byte slot parity).
// * Epilogue in registers: bias, activation, residual, second activation, scale, or the
//   ConvGRU gate math, then stores (lane = cout => 32 consecutive channels per pixel).
//
// Replaces: every nn.Conv2d on the GIMM-VFI-R path (see include/gimmvfi_hip.h).
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
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma_f16_32x32x16(const uint4& a, const uint4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
#endif

template <typename T> struct Mma;
template <> struct Mma<f16_t> {
    static __device__ __forceinline__ void run(f32x16& acc, const uint4& a, const uint4& b) {
        acc = mfma_f16_32x32x16(a, b, acc);
    }
};
template <> struct Mma<bf16_t> {
    static __device__ __forceinline__ void run(f32x16& acc, const uint4& a, const uint4& b) {
        acc = mfma_bf16_32x32x16(a, b, acc);
    }
};
template <> struct Mma<float> {
    static __device__ __forceinline__ void run(f32x16& acc, const uint4& a, const uint4& b) {
        acc = mfma_f32_32x32x2(__builtin_bit_cast(float, a.x), __builtin_bit_cast(float, b.x), acc);
        acc = mfma_f32_32x32x2(__builtin_bit_cast(float, a.y), __builtin_bit_cast(float, b.y), acc);
        acc = mfma_f32_32x32x2(__builtin_bit_cast(float, a.z), __builtin_bit_cast(float, b.z), acc);
        acc = mfma_f32_32x32x2(__builtin_bit_cast(float, a.w), __builtin_bit_cast(float, b.w), acc);
    }
};

struct ConvArgs {
    gvfi_conv_params p;
    int Cg;      // 16-byte k-groups per filter tap  = (c0+c1)/VE
    int KGtot;   // total k-groups = KH*KW*Cg
    int KT;      // K chunks = ceil(KGtot / BKV)
    int Mg;      // output pixels per weight group
    int c0g;     // k-groups that come from source 0
    long long Ktot;  // weight row length in elements
};

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int BKV>
__global__ void __launch_bounds__(256) conv_igemm_kernel(ConvArgs a) {
    constexpr int VE = Elem<T>::VE;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int RB = BKV * 16;   // LDS row bytes
    constexpr int RPB = 256 / RB;  // rows per 256-byte bank row
    constexpr int ROWS_PER_PASS = 256 / BKV;
    constexpr int A_VECS = BM / ROWS_PER_PASS;
    constexpr int B_VECS = (BN + ROWS_PER_PASS - 1) / ROWS_PER_PASS;
    constexpr int STAGE = (BM + BN) * RB;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves");
    static_assert(MI >= 1 && NI >= 1, "tile");

    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

    const gvfi_conv_params& p = a.p;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int g = blockIdx.z;
    const long long m_tile0 = (long long)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int HoWo = p.Ho * p.Wo;
    const T* __restrict__ x0 = (const T*)p.x0;
    const T* __restrict__ x1 = (const T*)p.x1;
    const T* __restrict__ wg = (const T*)p.w + (long long)g * p.w_group_stride;

    const int kv = tid % BKV;
    const int lrow = tid / BKV;

    int a_iy0[A_VECS], a_ix0[A_VECS], a_n[A_VECS];
    bool a_ok[A_VECS];
#pragma unroll
    for (int i = 0; i < A_VECS; ++i) {
        const int row = lrow + i * ROWS_PER_PASS;
        const long long m = m_tile0 + row;
        a_ok[i] = m < a.Mg;
        const long long mm = (long long)g * a.Mg + (a_ok[i] ? m : 0);
        const int n = (int)(mm / HoWo);
        const int rem = (int)(mm - (long long)n * HoWo);
        const int oy = rem / p.Wo;
        const int ox = rem - oy * p.Wo;
        a_n[i] = n;
        a_iy0[i] = oy * p.stride - p.pad_h;
        a_ix0[i] = ox * p.stride - p.pad_w;
    }
    // k-group walker for this thread's vector column
    int kg = kv;
    int cv = kg % a.Cg;
    int tap = kg / a.Cg;
    int kh = tap / p.KW;
    int kw = tap - kh * p.KW;

    uint4 ra[A_VECS], rb[B_VECS];

    auto load_tiles = [&]() {
        const bool kvalid = kg < a.KGtot;
#pragma unroll
        for (int i = 0; i < A_VECS; ++i) {
            uint4 v = {0u, 0u, 0u, 0u};
            if (kvalid && a_ok[i]) {
                int iy = a_iy0[i] + kh, ix = a_ix0[i] + kw;
                if (p.pad_mode == GVFI_PAD_REFLECT) {
                    iy = reflect_idx(iy, p.H);
                    ix = reflect_idx(ix, p.W);
                }
                if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
                    const long long pix = ((long long)a_n[i] * p.H + iy) * p.W + ix;
                    const T* src = (cv < a.c0g) ? (x0 + pix * p.ld0 + cv * VE) : (x1 + pix * p.ld1 + (cv - a.c0g) * VE);
                    v = *(const uint4*)src;
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_VECS; ++i) {
            uint4 v = {0u, 0u, 0u, 0u};
            const int row = lrow + i * ROWS_PER_PASS;
            const int n = n0 + row;
            if (kvalid && row < BN && n < p.Cout) v = *(const uint4*)(wg + (long long)n * a.Ktot + (long long)kg * VE);
            rb[i] = v;
        }
        // advance to the next K chunk
        kg += BKV;
        cv += BKV;
        while (cv >= a.Cg) {
            cv -= a.Cg;
            if (++kw == p.KW) { kw = 0; ++kh; }
        }
    };
    auto store_tiles = [&](int buf) {
        unsigned char* sa = smem + buf * STAGE;
        unsigned char* sb = sa + BM * RB;
#pragma unroll
        for (int i = 0; i < A_VECS; ++i) {
            const int row = lrow + i * ROWS_PER_PASS;
            *(uint4*)(sa + row * RB + ((kv ^ ((row / RPB) & (BKV - 1))) << 4)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_VECS; ++i) {
            const int row = lrow + i * ROWS_PER_PASS;
            if (row < BN) *(uint4*)(sb + row * RB + ((kv ^ ((row / RPB) & (BKV - 1))) << 4)) = rb[i];
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_tiles();
    store_tiles(0);
    __syncthreads();

    const int frow = lane & 31;
    const int fhalf = lane >> 5;
    for (int kt = 0; kt < a.KT; ++kt) {
        const int buf = kt & 1;
        const bool more = (kt + 1) < a.KT;
        if (more) load_tiles();
        const unsigned char* sa = smem + buf * STAGE;
        const unsigned char* sb = sa + BM * RB;
#pragma unroll
        for (int kk = 0; kk < BKV / 2; ++kk) {
            const int slot = 2 * kk + fhalf;
            uint4 fa[MI], fb[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int row = wm * WM + i * 32 + frow;
                fa[i] = *(const uint4*)(sa + row * RB + ((slot ^ ((row / RPB) & (BKV - 1))) << 4));
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int row = wn * WN + j * 32 + frow;
                fb[j] = *(const uint4*)(sb + row * RB + ((slot ^ ((row / RPB) & (BKV - 1))) << 4));
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) Mma<T>::run(acc[i][j], fa[i], fb[j]);
        }
        if (more) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // ---------------------------------------------------------------- epilogue
    const int half = p.Cout >> 1;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int cout = n0 + wn * WN + j * 32 + frow;
        const bool cok = cout < p.Cout;
        const float bias = (cok && p.bias) ? p.bias[cout] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
                const long long m = m_tile0 + row;
                if (!cok || m >= a.Mg) continue;
                const long long pix = (long long)g * a.Mg + m;
                float v = acc[i][j][r] + bias;
                if (p.epi_mode == GVFI_EPI_STD) {
                    v = apply_act(v, p.act1, p.slope1, cout);
                    if (p.res) v += ld_any<T>(p.res, pix * p.ldr + cout, p.res_f32);
                    v = apply_act(v, p.act2, p.slope2, cout);
                    v *= p.out_scale;
                    st_any<T>(p.y, pix * p.ldy + cout, p.y_f32, v);
                } else if (p.epi_mode == GVFI_EPI_GRU_ZR) {
                    if (p.res) v += ld_any<T>(p.res, pix * p.ldr + cout, p.res_f32);   // hoisted context term
                    const float s = gvfi_sigmoid(v);
                    if (cout < half) {
                        st_any<T>(p.y, pix * p.ldy + cout, p.state_f32, s);      // (state_f32: z stays float)
                    } else {
                        const int c = cout - half;
                        const float h = ld_any<T>(p.aux0, pix * p.lda0 + c, p.state_f32);
                        st_any<T>(p.y2, pix * p.ldy2 + c, 0, s * h);
                    }
                } else {  // GVFI_EPI_GRU_Q
                    if (p.res) v += ld_any<T>(p.res, pix * p.ldr + cout, p.res_f32);
                    const float q = tanhf(v);
                    const float h = ld_any<T>(p.aux0, pix * p.lda0 + cout, p.state_f32);
                    const float z = ld_any<T>(p.aux1, pix * p.lda1 + cout, p.state_f32);
                    const float hn = (1.f - z) * h + z * q;
                    st_any<T>(p.y, pix * p.ldy + cout, 0, hn);
                    if (p.state_f32 && p.y2 != nullptr) st_any<T>(p.y2, pix * p.ldy2 + cout, 1, hn);   // the float state
                }
            }
        }
    }
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int BKV>
static int launch_conv(const gvfi_conv_params& p, hipStream_t stream) {
    constexpr int VE = Elem<T>::VE;
    ConvArgs a;
    a.p = p;
    const int cin = p.c0 + p.c1;
    a.Cg = cin / VE;
    a.c0g = p.c0 / VE;
    a.KGtot = p.KH * p.KW * a.Cg;
    a.KT = (a.KGtot + BKV - 1) / BKV;
    a.Ktot = (long long)p.KH * p.KW * cin;
    const int groups = p.groups > 0 ? p.groups : 1;
    a.Mg = (int)(((long long)p.N * p.Ho * p.Wo) / groups);
    dim3 grid(cdiv(a.Mg, BM), cdiv(p.Cout, BN), groups);
    GVFI_LAUNCH_COOP((conv_igemm_kernel<T, BM, BN, WAVES_M, WAVES_N, BKV>), grid, dim3(256), stream, a);
    return (int)hipGetLastError();
}

static int generic_tile(const gvfi_conv_params& p) {
    int tile = p.tile_hint & 1023;
    if (tile == 0) tile = p.Cout > 64 ? 128 : (p.Cout > 32 ? 64 : 32);
    return tile >= 128 ? 128 : (tile >= 64 ? 64 : 32);
}

template <typename T> static int dispatch_conv(const gvfi_conv_params& p, hipStream_t stream) {
    switch (generic_tile(p)) {
        case 128: return launch_conv<T, 128, 128, 2, 2, 4>(p, stream);
        case 64: return launch_conv<T, 128, 64, 2, 2, 4>(p, stream);
        default: return launch_conv<T, 128, 32, 4, 1, 4>(p, stream);
    }
}

// Which kernel gvfi_conv2d would launch for *pp: plan[5] = {algo (1 generic, 2 LDS-DMA, 3 patch, 4 halo-staged 3x3, 5 its mid-channel sibling, 7 column kernel of the 7x7 few-channel layers), BM, BN, K-chunk bytes, LDS stages}.
// the patch kernel (conv_patch.hip) takes what the LDS-DMA kernel cannot: few channels / reflect padding at full resolution
static bool use_patch(const gvfi_conv_params& p) {
    if ((p.algo & 15) == 3) return true;
    return (p.algo & 15) == 0 && !gvfi_conv2d_glds_eligible(&p) && gvfi_conv2d_patch_eligible(&p) == 1;
}

static bool use_col7(const gvfi_conv_params& p) {
    if ((p.algo & 15) == 7) return true;
    return (p.algo & 15) == 0 && gvfi_conv2d_col7_eligible(&p) == 1;
}

static bool use_p3x3(const gvfi_conv_params& p) {
    if ((p.algo & 15) == 4) return true;
    return (p.algo & 15) == 0 && gvfi_conv2d_p3x3_eligible(&p) == 1;
}

static bool use_p3x3s(const gvfi_conv_params& p) {
    if ((p.algo & 15) == 5) return true;
    return (p.algo & 15) == 0 && gvfi_conv2d_p3x3s_eligible(&p) == 1;
}

extern "C" int gvfi_conv2d_plan(const gvfi_conv_params* pp, int* plan) {
    const gvfi_conv_params& p = *pp;
    if ((p.algo & 15) == 8) return -2;     // (algo 8 was the row-linear experiment, tools/experiments/csrc/conv_lin.hip: not in the library)
    if (use_p3x3s(p)) {
        plan[0] = 5; plan[1] = 256; plan[2] = p.Cout > 32 ? 64 : 32; plan[3] = 2 * p.c0; plan[4] = 2;
        return 0;
    }
    if (use_p3x3(p)) {
        plan[0] = 4; plan[1] = 256; plan[2] = 256; plan[3] = 128; plan[4] = 2;
        return 0;
    }
    if ((p.algo & 15) == 2 || ((p.algo & 15) == 0 && gvfi_conv2d_glds_eligible(pp))) return gvfi_conv2d_glds_plan(pp, plan);
    if (use_col7(p)) {
        plan[0] = 7; plan[1] = 1024; plan[2] = p.Cout > 16 ? 32 : 16; plan[3] = 16; plan[4] = 1;
        return 0;
    }
    if (use_patch(p)) {
        plan[0] = 3; plan[1] = 512; plan[2] = p.Cout > 32 ? 64 : 32; plan[3] = 16; plan[4] = 1;
        return 0;
    }
    if (p.w_layout != 0) return -5;
    plan[0] = 1; plan[1] = 128; plan[2] = generic_tile(p); plan[3] = 64; plan[4] = 2;
    return 0;
}

extern "C" int gvfi_conv2d(const gvfi_conv_params* pp, void* stream) {
    const gvfi_conv_params& p = *pp;
    if ((p.algo & 15) == 8) return -2;
    if (use_p3x3s(p)) return gvfi_conv2d_p3x3s(pp, stream);
    if (use_p3x3(p)) return gvfi_conv2d_p3x3(pp, stream);
    if ((p.algo & 15) == 2 || ((p.algo & 15) == 0 && gvfi_conv2d_glds_eligible(pp))) return gvfi_conv2d_glds(pp, stream);
    if (use_col7(p)) return gvfi_conv2d_col7(pp, stream);
    if (use_patch(p)) return gvfi_conv2d_patch(pp, stream);
    if (p.w_layout != 0) return -5;   // the chunked weight image is only understood by the LDS-DMA kernel
    if (p.stats != nullptr) return -6;   // fused statistics exist only in the LDS-DMA kernel (gvfi_conv2d_stats_ok)
    const int ve = p.dtype == GVFI_F32 ? 4 : 8;
    if (p.c0 <= 0 || (p.c0 % ve) || (p.c1 % ve) || (p.ld0 % ve) || (p.c1 > 0 && (p.ld1 % ve))) return -2;
    if (((uintptr_t)p.x0 & 15) || ((uintptr_t)p.x1 & 15) || ((uintptr_t)p.w & 15)) return -3;
    if (p.groups > 1 && (p.N % p.groups)) return -4;
    if (p.dtype == GVFI_F32) return dispatch_conv<float>(p, (hipStream_t)stream);
    if (p.dtype == GVFI_F16) return dispatch_conv<f16_t>(p, (hipStream_t)stream);
    return dispatch_conv<bf16_t>(p, (hipStream_t)stream);
}
