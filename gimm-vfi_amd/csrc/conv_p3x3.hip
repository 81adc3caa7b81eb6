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

#ifndef GVFI_HOSTSIM
#define P3_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#else
#define P3_WAVE_SYNC() emu::wave_sync()
#endif
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

// 8 waves of 128 x 64, 2 per SIMD.  (Measured and dropped: 4 waves of 128 x 128 with 256 accumulator registers -- a third
// fewer LDS fragment reads per MFMA, but one wave per SIMD: 0.965 vs 0.861 ms on the 8 x 256 x 448 256->256 layer.)
// PROF (algo bit 15): wave 0 adds up shader-clock cycles per phase into aux1[block * 4 + {prologue, K loop, of which waiting
// for the DMA + barrier, epilogue}] (tools/p3x3_timeline.py)
template <bool PROF> __global__ void __launch_bounds__(512) conv_p3x3_kernel(P3Args a) {
    typedef bf16_t T;
    constexpr int WAVES_N = 4;
    constexpr int NW = 2 * WAVES_N, NT = NW * 64, WM = 128, WN = 256 / WAVES_N, MI = 4, NI = WN / 32, BN = 256, BM = 256, RB = 128, KK = 4;
    constexpr int QP = (P3_PIECES + NW - 1) / NW;      // patch pieces per wave (6 / 12)
    constexpr int PPT = (QP + 5) / 6;                  // ... of the next channel chunk issued per tap (taps 0..5)
    constexpr int B_INSTR = 32 / NW;
    constexpr int RPI = NT / 32;                       // tile rows per store-loop iteration
    constexpr int PTAB = 2 * P3_PATCH + 2 * P3_BSTAGE;      // per-channel epilogue parameters: bias, slope 1, slope 2 (3 x 256 floats)
    __shared__ __attribute__((aligned(16))) unsigned char smem[PTAB + 3 * 256 * 4];
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
    // ---- per-channel epilogue parameters -> LDS (published by the prologue barrier)
    if (tid < 256) {
        float* pt = (float*)(smem + PTAB);
        pt[tid] = p.bias ? p.bias[n0 + tid] : 0.f;
        pt[256 + tid] = p.act1 == GVFI_ACT_PRELU ? p.slope1[n0 + tid] : (p.act1 == GVFI_ACT_NONE ? 1.f : (p.act1 == GVFI_ACT_LRELU ? 0.1f : 0.f));
        pt[512 + tid] = p.act2 == GVFI_ACT_PRELU ? p.slope2[n0 + tid] : (p.act2 == GVFI_ACT_NONE ? 1.f : (p.act2 == GVFI_ACT_LRELU ? 0.1f : 0.f));
    }
    // ---- prologue: patch of chunk 0 + weights of (chunk 0, tap 0)
#pragma unroll
    for (int q = 0; q < QP; ++q) issue_patch(0, q);
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) issue_b(0, i);

    unsigned long long ph[4] = {0, 0, 0, 0}, tprev = 0;
    auto now = [&]() -> unsigned long long {
#ifndef GVFI_HOSTSIM
        return __builtin_readcyclecounter();
#else
        return 0;
#endif
    };
    if (PROF) tprev = now();
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
    // sits before the last k-step (see conv_igemm_glds.hip).
    // Measured alternatives for the issue point (same box, 8 x 256 x 448 256->256, tools/p3x3_timeline.py): right after the
    // barrier, before the fragment reads: K loop 92.6 instead of 85.0 kcycles; behind the first MFMA group of the LAST
    // k-step (weights two steps ahead): 91.6 kcycles, yet 0.849 against 0.833 ms -- the shader clock rises as utilisation
    // falls (1.63 -> 1.70 GHz): the kernel runs at the chip's power limit, not at a latency limit; waves 4-7 two / four /
    // six MFMA groups later than waves 0-3: 98.6 / 102.7 / 107.6 kcycles.
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
                for (int j = 0; j < NI; ++j) Mma2<T>::run(acc[i][j], fb[kk & 1][j], fa[kk & 1][i]);
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
            unsigned long long tw = 0;
            if (PROF) tw = now();
            glds_wait_n<0>();
            if (PROF) { const unsigned long long t = now(); ph[3] += t - tw; }     // (own DMA pieces landed)
            __syncthreads();
            if (PROF) ph[2] += now() - tw;
            set_bases(tap == 8 ? c + 1 : c, ntap);
            load_frags(ntoff, 0, KK & 1);
            GVFI_SCHED_BARRIER();
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int j = 0; j < NI; ++j) Mma2<T>::run(acc[i][j], fb[(KK - 1) & 1][j], fa[(KK - 1) & 1][i]);
        }
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
    if (PROF) { const unsigned long long t = now(); ph[0] = t - tprev; tprev = t; }
    set_bases(0, 0);
    load_frags(0, 0, 0);
    int c = 0;
    for (; c + 1 < a.chunks; ++c) chunk9(c, std::false_type{});
    chunk9(c, std::true_type{});
    if (PROF) { const unsigned long long t = now(); ph[1] = t - tprev; tprev = t; }
    auto prof_out = [&]() {
        if (PROF && tid == 0) {
            const unsigned long long own = ph[3];
            ph[3] = now() - tprev;
            ph[2] |= own << 32;      // low word: vmcnt + barrier wait, high word: the vmcnt part
            unsigned long long* o = (unsigned long long*)p.aux1 + (size_t)bid * 4;
            for (int k = 0; k < 4; ++k) o[k] = ph[k];
        }
    };

    // ---------------------------------------------------------------- epilogue
    // y = act2(act1(acc + bias) + res) * out_scale, the arithmetic of conv_igemm_glds.hip.  The MFMAs ran with the weights
    // as their row operand: a lane holds ONE pixel (tile row wm*128 + i*32 + lane%32) and, per accumulator block, 4 x 4
    // consecutive output channels (wn*64 + j*32 + 8*(r/4) + 4*(lane/32) + r%4).  Four channels are one 8-byte unit of the
    // NHWC line, so bias / activation run as packed pairs and a ds_write_b64 stages them (the column-per-lane layout
    // of the other kernels needs a 2-byte write and ~8 VALU operations per element: 11 of this kernel's 100 kcycles).
    // act(t) = max(t,0) + s*min(t,0) == med3(t, s*t, s <= 1 ? +inf : -inf) for every activation this kernel takes
    // (none: s = 1, ReLU: s = 0, leaky: s = 0.1, PReLU: per channel).
    // Staging tile: 256 rows x 512 bytes; the 16-byte unit u of row r sits at slot u ^ (r & 15), and rows with bit 4 set
    // swap the two 8-byte halves of a unit -- 32 lanes writing 8 bytes of 32 different rows cover the 64 banks once.
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
    const bool has_sc = p.out_scale != 1.0f, has_res = p.res != nullptr;
    const float* ptab = (const float*)(smem + PTAB);
    __syncthreads();   // every wave is done reading the last staged step
    if (has_res) {
        // residual tile -> staging area (same layout: the result overwrites it in place), 128 DMA instructions of 2 rows
        // (descriptor at the tile's first pixel, offsets relative to it: the residual tensor itself may exceed the 2 GB a
        // raw-buffer offset spans -- 7 timesteps of a 1024x544x256 decoder activation in one batch are 2.0 GB)
        const long long pix0 = img_pix + (long long)y0 * p.W + x0;
        const gvfi_i32x4 srd_r = make_srd((const bf16_t*)p.res + pix0 * p.ldr + n0);
#pragma unroll
        for (int q = 0; q < 128 / NW; ++q) {
            const int piece = q * NW + wave;
            const int row = piece * 2 + (lane >> 5);
            bool ok;
            const long long pix = pix_of(row, ok);
            const unsigned off = ok ? (unsigned)((pix - pix0) * p.ldr * 2 + (((lane & 31) ^ (row & 15)) << 4)) : GVFI_DMA_OOB;
            bufdma16(off, srd_r, 0u, smem_lds + piece * 1024);
        }
        glds_wait_n<0>();
        __syncthreads();
    }
    const float inf = __builtin_inff();
    auto stage = [&](auto res_tag, auto sc_tag) {
        constexpr bool RES = decltype(res_tag)::value, SC = decltype(sc_tag)::value;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c0 = wn * WN + j * 32 + 8 * g + 4 * fhalf;          // first of this lane's 4 channels (within the tile)
                const float4 b4 = *(const float4*)(ptab + c0), s4 = *(const float4*)(ptab + 256 + c0);
                const float bb[4] = {b4.x, b4.y, b4.z, b4.w}, s1[4] = {s4.x, s4.y, s4.z, s4.w};
                float s2[4] = {1.f, 1.f, 1.f, 1.f}, k1[4], k2[4];
                if (RES) {
                    const float4 z4 = *(const float4*)(ptab + 512 + c0);
                    s2[0] = z4.x; s2[1] = z4.y; s2[2] = z4.z; s2[3] = z4.w;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    k1[e] = s1[e] <= 1.f ? inf : -inf;
                    k2[e] = s2[e] <= 1.f ? inf : -inf;
                }
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int row = wm * WM + i * 32 + frow;
                    const int unit = row * 512 + (((c0 >> 3) ^ (row & 15)) << 4);
                    float vv[4];
#ifndef GVFI_HOSTSIM
                    {   // packed pairs: v_pk_add_f32 / v_pk_mul_f32
                        typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const f2 a2 = {acc[i][j][4 * g + 2 * h], acc[i][j][4 * g + 2 * h + 1]};
                            const f2 t2 = a2 + f2{bb[2 * h], bb[2 * h + 1]};
                            const f2 st = t2 * f2{s1[2 * h], s1[2 * h + 1]};
                            vv[2 * h] = med3f(t2.x, st.x, k1[2 * h]);
                            vv[2 * h + 1] = med3f(t2.y, st.y, k1[2 * h + 1]);
                        }
                    }
#else
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t = acc[i][j][4 * g + e] + bb[e];
                        vv[e] = med3f(t, s1[e] * t, k1[e]);
                    }
#endif
                    if (RES) {
                        const uint2 ru = *(const uint2*)(smem + unit + (((c0 >> 2) & 1) << 3));
                        vv[0] += __builtin_bit_cast(float, ru.x << 16);
                        vv[1] += __builtin_bit_cast(float, ru.x & 0xffff0000u);
                        vv[2] += __builtin_bit_cast(float, ru.y << 16);
                        vv[3] += __builtin_bit_cast(float, ru.y & 0xffff0000u);
#ifndef GVFI_HOSTSIM
                        {
                            typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const f2 t2 = {vv[2 * h], vv[2 * h + 1]};
                                const f2 st = t2 * f2{s2[2 * h], s2[2 * h + 1]};
                                vv[2 * h] = med3f(t2.x, st.x, k2[2 * h]);
                                vv[2 * h + 1] = med3f(t2.y, st.y, k2[2 * h + 1]);
                            }
                        }
#else
#pragma unroll
                        for (int e = 0; e < 4; ++e) vv[e] = med3f(vv[e], s2[e] * vv[e], k2[e]);
#endif
                        P3_WAVE_SYNC();   // rows with bit 4 set: lanes l and l + 32 write the halves the other one has just read
                    }
                    if (SC) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) vv[e] *= p.out_scale;
                    }
                    uint2 u;
                    u.x = pack_bf16x2(vv[0], vv[1]);
                    u.y = pack_bf16x2(vv[2], vv[3]);
                    *(uint2*)(smem + unit + ((((c0 >> 2) ^ (row >> 4)) & 1) << 3)) = u;
                }
            }
        }
    };
    if (has_res) {
        if (has_sc) stage(std::true_type{}, std::true_type{}); else stage(std::true_type{}, std::false_type{});
    } else {
        if (has_sc) stage(std::false_type{}, std::true_type{}); else stage(std::false_type{}, std::false_type{});
    }
    __syncthreads();
    {   // iteration it of a thread = tile row row_a + 16 * it = pixel (y0 + it, x0 + row_a): one pointer, a constant stride
        static_assert(RPI == P3_TW, "store loop geometry");
        const int x = x0 + row_a;
        bf16_t* yp = (bf16_t*)p.y + (img_pix + (long long)y0 * p.W + x) * p.ldy + my_cout0;
        const long long ystride = (long long)p.W * p.ldy;
        const unsigned char* sp = smem + row_a * 512 + ((my_cg ^ row_a) << 4);
        const int nrow = x < p.W ? (p.H - y0 < P3_TH ? p.H - y0 : P3_TH) : 0;
        uint4 u[P3_TH];                      // all LDS reads first (the accumulator registers are free by now)
#pragma unroll
        for (int it = 0; it < P3_TH; ++it) {
            const uint4 t = *(const uint4*)(sp + it * (RPI * 512));
            if (it & 1) {                       // rows with bit 4 set hold the two 8-byte halves swapped
                u[it].x = t.z; u[it].y = t.w; u[it].z = t.x; u[it].w = t.y;
            } else {
                u[it] = t;
            }
        }
#pragma unroll
        for (int it = 0; it < P3_TH; ++it)
            if (it < nrow) *(uint4*)(yp + it * ystride) = u[it];
    }
    prof_out();
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
    if (p.res && (long long)18 * p.W * p.ldr * 2 >= 0x7fffff00ll) return 0;               // ... and the residual tile's rows
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
    if (((p.algo >> 8) & 128) && p.aux1 != nullptr) {
        GVFI_LAUNCH_COOP(conv_p3x3_kernel<true>, dim3(a.per_xcd * 8), dim3(512), (hipStream_t)stream, a);
    } else {
        GVFI_LAUNCH_COOP(conv_p3x3_kernel<false>, dim3(a.per_xcd * 8), dim3(512), (hipStream_t)stream, a);
    }
    return (int)hipGetLastError();
}
