// 3x3 stride-1 zero-padded convolution for the FLOP-dominant layers (bf16, Cout a multiple of 256, channel counts
// multiples of 64, >= 65536 output pixels): the ResBlocks of NewMultiFlowDecoder (fi_components.py:97-154, 279-340) --
// 63-87 % of the path's FLOPs.  Same math, weight image (w_layout 1), tile (256 x 256, 8 waves, 128 x 64 per wave),
// chunk-pipelined K loop and epilogue arithmetic as conv_igemm_glds.hip; what differs is the A operand.
//
// conv_igemm_glds.hip walks K as (channel chunk, tap) and DMAs a fresh 256-pixel x 64-channel A tile for every tap: 9 x
// 32 KB per channel chunk, beside 9 x 32 KB of weights.  Round 2 measured that kernel bound on the L2 -> LDS delivery side
// (62 us of K loop per tile at any shader clock), so here the traffic itself goes down:
//   * the M tile is a 16 x 16 block of output pixels (not 256 consecutive ones), and the A operand of a channel chunk is
//     its 18 x 18 halo PATCH, staged ONCE (45.6 KB instead of 9 x 32 KB; -43 % of all L2 -> LDS bytes, -86 % of the
//     A-side DMA instructions).  Zero padding = out-of-range DMA offsets, as before;
//   * patch pixels are 144 bytes apart (128 + 16: an odd multiple of 16 B): the ds_read_b128 fragment reads of 16
//     neighbouring pixels tile the 64 banks without the XOR swizzle, so a tap is a uniform offset added to four per-lane
//     base addresses (a 16-lane read group is one pixel row of the tile: conflict-free).  A DMA instruction still writes 1 KiB lane-linear: lanes whose 16-byte
//     slot is the padding slot of a pixel use the out-of-range offset;
//   * two patch buffers (the next channel chunk's patch streams in one piece per tap, taps 0-5) + the two-stage weight
//     ring = 158 848 bytes of LDS.
#include "conv_mma.h"
#include <type_traits>

#define P3_TH 16
#define P3_TW 16
#define P3_PW (P3_TW + 2)
#define P3_PIX ((P3_TH + 2) * P3_PW)           // 324 patch pixels
#define P3_PITCH 144
#define P3_SLOTS (P3_PIX * 9)                  // 16-byte slots of a patch incl. the padding slot of every pixel
#define P3_PIECES ((P3_SLOTS + 63) / 64)       // 46 DMA instructions
#define P3_PATCH (P3_PIECES * 1024)            // 47 104 bytes: 324 x 144 rounded up to whole DMA instructions (the tail lanes write zeros)
#define P3_BSTAGE (256 * 128)

struct P3Args {
    gvfi_conv_params p;
    int chunks0, chunks;        // 64-channel chunks of source 0 / of both sources
    int tiles_x, tiles_y, mtiles, ntiles_n, per_xcd;
};

// WAVES_N = 4: 8 waves of 128 x 64 (2 per SIMD); WAVES_N = 2: 4 waves of 128 x 128 (1 per SIMD, 256 accumulator registers:
// a third fewer LDS fragment reads per MFMA)
template <int WAVES_N> __global__ void __launch_bounds__(128 * WAVES_N) conv_p3x3_kernel(P3Args a) {
    typedef bf16_t T;
    constexpr int NW = 2 * WAVES_N, NT = NW * 64, WM = 128, WN = 256 / WAVES_N, MI = 4, NI = WN / 32, BN = 256, BM = 256, RB = 128, KK = 4;
    constexpr int QP = (P3_PIECES + NW - 1) / NW;      // patch pieces per wave (6 / 12)
    constexpr int PPT = (QP + 5) / 6;                  // ... of the next channel chunk issued per tap (taps 0..5)
    constexpr int B_INSTR = 32 / NW;
    constexpr int RPI = NT / 32;                       // tile rows per store-loop iteration
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * P3_PATCH + 2 * P3_BSTAGE];
    const gvfi_conv_params& p = a.p;
    const int bid = blockIdx.x;
    const int v = (bid & 7) * a.per_xcd + (bid >> 3);          // XCD-aware order, as conv_igemm_glds.hip
    if (v >= a.mtiles * a.ntiles_n) return;
    const int mt = v / a.ntiles_n, nt = v - mt * a.ntiles_n;
    const int n0 = nt * BN;
    const int tiles_img = a.tiles_x * a.tiles_y;
    const int img = mt / tiles_img, trem = mt - img * tiles_img;
    const int tyi = trem / a.tiles_x, txi = trem - tyi * a.tiles_x;
    const int y0 = tyi * P3_TH, x0 = txi * P3_TW;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const T* __restrict__ xs0 = (const T*)p.x0;
    const T* __restrict__ xs1 = (const T*)p.x1;
    const T* __restrict__ wg = (const T*)p.w;

    // ---- patch DMA: per-lane byte offsets relative to the patch origin (image pixel (y0-1, x0-1), may lie in the padding)
    unsigned a_off0[QP], a_off1[QP];
#pragma unroll
    for (int q = 0; q < QP; ++q) {
        const int piece = q * NW + wave;
        const int s = piece * 64 + lane;
        const int pp = s / 9, col = s - pp * 9;
        const int py = pp / P3_PW, px = pp - py * P3_PW;
        const int gy = y0 - 1 + py, gx = x0 - 1 + px;
        const bool ok = piece < P3_PIECES && pp < P3_PIX && col < 8 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        a_off0[q] = ok ? (unsigned)(((py * p.W + px) * p.ld0 + col * 8) * 2) : GVFI_DMA_OOB;
        a_off1[q] = ok ? (unsigned)(((py * p.W + px) * p.ld1 + col * 8) * 2) : GVFI_DMA_OOB;
    }
    const long long pix_org = ((long long)img * p.H + (y0 - 1)) * p.W + (x0 - 1);
    const gvfi_i32x4 srd_a0 = make_srd(xs0 + pix_org * p.ld0);
    const gvfi_i32x4 srd_a1 = make_srd(xs1 != nullptr ? xs1 + pix_org * p.ld1 : xs0);
    const gvfi_i32x4 srd_b = make_srd(wg);
    // ---- weight tile DMA (chunk-major, pre-swizzled image: [K chunk][Cout][128 B] == the LDS image)
    const int lrow = lane >> 3, lslot = lane & 7;
    unsigned b_off[B_INSTR];
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) b_off[i] = (unsigned)(((n0 + (i * NW + wave) * 8 + lrow) * 8 + lslot) * 16);
    // ---- fragment read addresses
    auto swz = [](int row) { return (row >> 1) & 7; };
    unsigned abase[MI], b_rd[KK];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int row = wm * WM + i * 32 + (lane & 31);
        abase[i] = (unsigned)(((row >> 4) * P3_PW + (row & 15)) * P3_PITCH + (lane >> 5) * 16);
    }
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        const int rb = wn * WN + (lane & 31);
        const int slot = 2 * kk + (lane >> 5);
        b_rd[kk] = 2 * P3_PATCH + rb * RB + ((slot ^ swz(rb)) << 4);
    }
    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const unsigned smem_lds = lds_address(smem);
    auto issue_patch = [&](int c, int q) {
        const int piece = q * NW + wave;
        if (piece >= P3_PIECES) return;
        const bool from0 = c < a.chunks0;
        bufdma16(from0 ? a_off0[q] : a_off1[q], from0 ? srd_a0 : srd_a1, (unsigned)((from0 ? c : c - a.chunks0) * 128),
                 smem_lds + (c & 1) * P3_PATCH + piece * 1024);
    };
    auto issue_b = [&](int kt, int i) {
        bufdma16(b_off[i], srd_b, (unsigned)(kt * p.Cout * 128), smem_lds + 2 * P3_PATCH + (kt & 1) * P3_BSTAGE + (i * NW + wave) * 1024);
    };
    // ---- prologue: patch of chunk 0 + weights of (chunk 0, tap 0)
#pragma unroll
    for (int q = 0; q < QP; ++q) issue_patch(0, q);
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) issue_b(0, i);

    uint4 fa[2][MI], fb[2][NI];
    // c: channel chunk (runtime), tap / kk compile-time: the tap and k-step offsets are ds_read immediates
    unsigned pa[MI], pb[KK];        // this step's fragment base addresses
    auto set_bases = [&](int c, int tap) {
#pragma unroll
        for (int i = 0; i < MI; ++i) pa[i] = abase[i] + (c & 1) * P3_PATCH;
        const int bsel = ((c + tap) & 1) * P3_BSTAGE;      // (c * 9 + tap) & 1
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) pb[kk] = b_rd[kk] + bsel;
    };
    auto load_frags = [&](int toff, int kk, int buf) {
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[buf][i] = *(const uint4*)(smem + pa[i] + toff + kk * 32);
#pragma unroll
        for (int j = 0; j < NI; ++j) fb[buf][j] = *(const uint4*)(smem + pb[kk] + j * 32 * RB);
    };
    // one (channel chunk, tap) step: 4 k-steps of 8 MFMAs; the weights of the next step (and one patch piece of the next
    // channel chunk during taps 0..5) are issued right behind the first MFMA group; the barrier that publishes them
    // sits before the last k-step (see conv_igemm_glds.hip)
    auto step = [&](int c, auto tap_tag, auto last_chunk_tag) {
        constexpr int tap = decltype(tap_tag)::value;
        constexpr bool LAST_CHUNK = decltype(last_chunk_tag)::value;
        constexpr bool HAS_NEXT = !(LAST_CHUNK && tap == 8);
        constexpr int toff = ((tap / 3) * P3_PW + (tap % 3)) * P3_PITCH;
        constexpr int ntap = tap == 8 ? 0 : tap + 1;
        constexpr int ntoff = ((ntap / 3) * P3_PW + (ntap % 3)) * P3_PITCH;
        const int kt = c * 9 + tap;
#pragma unroll
        for (int kk = 0; kk + 1 < KK; ++kk) {
            load_frags(toff, kk + 1, (kk + 1) & 1);
            GVFI_SCHED_BARRIER();
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int j = 0; j < NI; ++j) Mma2<T>::run(acc[i][j], fa[kk & 1][i], fb[kk & 1][j]);
                if (HAS_NEXT && kk == 0 && i == 0) {
#pragma unroll
                    for (int q = 0; q < B_INSTR; ++q) issue_b(kt + 1, q);
                    if (!LAST_CHUNK && tap < 6) {
#pragma unroll
                        for (int q = tap * PPT; q < (tap + 1) * PPT && q < QP; ++q) issue_patch(c + 1, q);
                    }
                }
            }
            GVFI_SCHED_BARRIER();
        }
        if (HAS_NEXT) {
            glds_wait_n<0>();
            __syncthreads();
            set_bases(tap == 8 ? c + 1 : c, ntap);
            load_frags(ntoff, 0, KK & 1);
            GVFI_SCHED_BARRIER();
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) Mma2<T>::run(acc[i][j], fa[(KK - 1) & 1][i], fb[(KK - 1) & 1][j]);
        GVFI_SCHED_BARRIER();
    };
    auto chunk9 = [&](int c, auto last) {
        step(c, std::integral_constant<int, 0>{}, last);
        step(c, std::integral_constant<int, 1>{}, last);
        step(c, std::integral_constant<int, 2>{}, last);
        step(c, std::integral_constant<int, 3>{}, last);
        step(c, std::integral_constant<int, 4>{}, last);
        step(c, std::integral_constant<int, 5>{}, last);
        step(c, std::integral_constant<int, 6>{}, last);
        step(c, std::integral_constant<int, 7>{}, last);
        step(c, std::integral_constant<int, 8>{}, last);
    };
    glds_wait_n<0>();
    __syncthreads();
    set_bases(0, 0);
    load_frags(0, 0, 0);
    int c = 0;
    for (; c + 1 < a.chunks; ++c) chunk9(c, std::false_type{});
    chunk9(c, std::true_type{});

    // ---------------------------------------------------------------- epilogue (arithmetic of conv_igemm_glds.hip)
    // tile row r <-> output pixel (y0 + r/16, x0 + r%16)
    const int frow = lane & 31, fhalf = lane >> 5;
    const int my_cg = tid & 31;                         // 32 groups of 8 channels per row, RPI rows per iteration
    const int my_cout0 = n0 + my_cg * 8;
    const int row_a = tid >> 5;
    const long long img_pix = (long long)img * p.H * p.W;
    auto pix_of = [&](int r, bool& ok) {
        const int y = y0 + (r >> 4), x = x0 + (r & 15);
        ok = y < p.H && x < p.W;
        return img_pix + (long long)y * p.W + x;
    };
    const float f1 = p.act1 == GVFI_ACT_NONE ? 1.f : (p.act1 == GVFI_ACT_LRELU ? 0.1f : 0.f);
    const float f2 = p.act2 == GVFI_ACT_NONE ? 1.f : (p.act2 == GVFI_ACT_LRELU ? 0.1f : 0.f);
    const bool has_sc = p.out_scale != 1.0f;
    __syncthreads();   // every wave is done reading the last staged step
    if (p.res == nullptr && p.act2 == GVFI_ACT_NONE) {
        // plain activation: bias + activation in the accumulator layout (per-lane scalars), bf16 tile staged once
        bf16_t* cs16 = (bf16_t*)smem;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int col = wn * WN + j * 32 + frow;
            const float bj = p.bias ? p.bias[n0 + col] : 0.f;
            const float sj = p.act1 == GVFI_ACT_PRELU ? p.slope1[n0 + col] : f1;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
                    const float t = acc[i][j][r] + bj;
                    float vv = fmaxf(t, 0.f) + sj * fminf(t, 0.f);
                    if (has_sc) vv *= p.out_scale;
                    cs16[row * BN + col] = (bf16_t)(pack_bf16x2(vv, 0.f) & 0xffffu);
                }
            }
        }
        __syncthreads();
#pragma unroll 4
        for (int it = 0; it < BM / RPI; ++it) {
            const int row = row_a + it * RPI;
            bool ok;
            const long long pix = pix_of(row, ok);
            if (!ok) continue;
            *(uint4*)((bf16_t*)p.y + pix * p.ldy + my_cout0) = *(const uint4*)(cs16 + row * BN + my_cg * 8);
        }
        return;
    }
    // residual / second activation: two fp32 passes of 128 rows (each wave stages half of its accumulator blocks per pass)
    float gb[8], gs1[8], gs2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        gb[e] = p.bias ? p.bias[my_cout0 + e] : 0.f;
        gs1[e] = p.act1 == GVFI_ACT_PRELU ? p.slope1[my_cout0 + e] : f1;
        gs2[e] = p.act2 == GVFI_ACT_PRELU ? p.slope2[my_cout0 + e] : f2;
    }
#ifndef GVFI_HOSTSIM
#pragma unroll
    for (int e = 0; e < 8; ++e) asm volatile("" ::"v"(gb[e]), "v"(gs1[e]), "v"(gs2[e]));   // wait for them here (see glds)
#endif
    float* cs = (float*)smem;
    const bool has_res = p.res != nullptr, has_a2 = p.act2 != GVFI_ACT_NONE;
    constexpr int IPP = 2;     // accumulator blocks per wave and pass; staged row lr of pass ps <-> tile row tile_row(ps, lr)
    auto tile_row = [&](int ps, int lr) { return (lr / (IPP * 32)) * WM + ps * IPP * 32 + (lr % (IPP * 32)); };
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        // all residual lines of the pass are requested before the accumulators are staged (loads and stores share the
        // in-order vmcnt counter: nothing may be stored before the last load has been issued)
        uint4 rr[128 / RPI];
        long long pixs[128 / RPI];
        bool oks[128 / RPI];
#pragma unroll
        for (int it = 0; it < 128 / RPI; ++it) {
            pixs[it] = pix_of(tile_row(ps, row_a + it * RPI), oks[it]);
            if (has_res && oks[it]) rr[it] = *(const uint4*)((const bf16_t*)p.res + pixs[it] * p.ldr + my_cout0);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            if (i / IPP != ps) continue;
            const int lrow0 = wm * (IPP * 32) + (i - ps * IPP) * 32;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int col = wn * WN + j * 32 + frow;
#pragma unroll
                for (int r = 0; r < 16; ++r) cs[(lrow0 + (r & 3) + 8 * (r >> 2) + 4 * fhalf) * BN + col] = acc[i][j][r];
            }
        }
        __syncthreads();
        constexpr int ITERS = 128 / RPI;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            if (!oks[it]) continue;
            const float* cp = cs + (row_a + it * RPI) * BN + my_cg * 8;
            const float4 c0 = *(const float4*)cp, c1 = *(const float4*)(cp + 4);
            float vv[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = vv[e] + gb[e];
                vv[e] = fmaxf(t, 0.f) + gs1[e] * fminf(t, 0.f);
            }
            if (has_res) {
                float r[8];
                unpack_bf16x8(rr[it], r);
#pragma unroll
                for (int e = 0; e < 8; ++e) vv[e] += r[e];
            }
            if (has_a2) {
#pragma unroll
                for (int e = 0; e < 8; ++e) vv[e] = fmaxf(vv[e], 0.f) + gs2[e] * fminf(vv[e], 0.f);
            }
            if (has_sc) {
#pragma unroll
                for (int e = 0; e < 8; ++e) vv[e] *= p.out_scale;
            }
            uint4 u;
            u.x = pack_bf16x2(vv[0], vv[1]);
            u.y = pack_bf16x2(vv[2], vv[3]);
            u.z = pack_bf16x2(vv[4], vv[5]);
            u.w = pack_bf16x2(vv[6], vv[7]);
            *(uint4*)((bf16_t*)p.y + pixs[it] * p.ldy + my_cout0) = u;
        }
        if (ps == 0) __syncthreads();
    }
}

// 1 = gvfi_conv2d routes this problem here ahead of the LDS-DMA kernel; 2 = runnable on request (algo 4) but too few
// output pixels for the 16 x 16 tiles to pay; 0 = not this kernel's problem
extern "C" int gvfi_conv2d_p3x3_eligible(const gvfi_conv_params* pp) {
    const gvfi_conv_params& p = *pp;
    if (p.dtype != GVFI_BF16 || p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad_h != 1 || p.pad_w != 1) return 0;
    if (p.pad_mode != GVFI_PAD_ZEROS || p.groups > 1 || p.epi_mode != GVFI_EPI_STD || p.w_layout != 1 || p.stats != nullptr) return 0;
    if (p.c0 <= 0 || (p.c0 % 64) || (p.c1 % 64) || p.Cout <= 0 || (p.Cout % 256)) return 0;
    if (p.Ho != p.H || p.Wo != p.W) return 0;
    if (p.y_f32 || (p.res != nullptr && p.res_f32) || p.act1 > GVFI_ACT_PRELU || p.act2 > GVFI_ACT_PRELU) return 0;
    if ((((uintptr_t)p.y) & 15) || ((p.ldy * 2) & 15) || (p.res && ((((uintptr_t)p.res) & 15) || ((p.ldr * 2) & 15)))) return 0;
    if (((uintptr_t)p.x0 & 15) || ((uintptr_t)p.x1 & 15) || ((uintptr_t)p.w & 15) || (p.ld0 % 8) || (p.c1 > 0 && (p.ld1 % 8))) return 0;
    // per-lane DMA offsets are 32-bit and stay below the descriptor's range: 18 image rows of the widest source
    if ((long long)18 * p.W * (p.ld0 > p.ld1 ? p.ld0 : p.ld1) * 2 >= 0x7fffff00ll) return 0;
    return (long long)p.N * p.H * p.W >= 65536 ? 1 : 2;
}

extern "C" int gvfi_conv2d_p3x3(const gvfi_conv_params* pp, void* stream) {
    if (!gvfi_conv2d_p3x3_eligible(pp)) return -2;
    const gvfi_conv_params& p = *pp;
    P3Args a;
    a.p = p;
    a.chunks0 = p.c0 / 64;
    a.chunks = (p.c0 + p.c1) / 64;
    a.tiles_x = cdiv(p.W, P3_TW);
    a.tiles_y = cdiv(p.H, P3_TH);
    a.mtiles = a.tiles_x * a.tiles_y * p.N;
    a.ntiles_n = p.Cout / 256;
    a.per_xcd = cdiv((long long)a.mtiles * a.ntiles_n, 8);
    if (p.algo & 32) {     // A/B: 4 waves of 128 x 128
        GVFI_LAUNCH_COOP(conv_p3x3_kernel<2>, dim3(a.per_xcd * 8), dim3(256), (hipStream_t)stream, a);
    } else {
        GVFI_LAUNCH_COOP(conv_p3x3_kernel<4>, dim3(a.per_xcd * 8), dim3(512), (hipStream_t)stream, a);
    }
    return (int)hipGetLastError();
}
