// 3x3 stride-1 zero-padded bf16 convolution of the MID-CHANNEL full-resolution layers: one source of exactly 32 or 64
// channels, at most 64 output channels -- the side branches of the decoder ResBlocks (conv2 / conv4, 64 -> 64,
// fi_components.py:107-133), the CNN encoder's 32 -> 32 residual layers (gimmvfi_r.py:84-109 via fi_components.py), the
// 32 -> 64 / 64 -> 32 transitions.  On the LDS-DMA kernel these layers stage a fresh A tile per tap for 8-16 MFMAs per wave
// and run at 280-450 TFLOP/s, 2.5-3x their HBM time.  This kernel is conv_p3x3.hip's scheme cut down to one channel chunk:
//   * 16 x 16-pixel output tile, its 18 x 18 halo patch staged ONCE (pixel pitch = channel bytes + 16: conflict-free
//     ds_read_b128 fragment reads, a tap is a constant address offset), so the K loop DMAs only weights: 9 steps of
//     BN x (2 CK) bytes from the plain [Cout][3][3][Cin] image, slot-swizzled on the fly;
//   * one patch buffer + a ring of small weight stages = 35-80 KB of LDS: 2-4 workgroups per CU overlap each other's prologue,
//     barriers and epilogue (the 256 x 256 tile of conv_p3x3.hip owns the CU alone);
//   * 4 waves, each 64 pixels x all BN output channels (an A fragment feeds BN / 32 MFMAs, a B fragment two);
//   * pixel-per-lane accumulators (weights are the MFMA row operand) and the packed epilogue of conv_p3x3.hip: bias /
//     activation on pairs, v_med3 activation, 8-byte staging writes, residual tile by LDS-DMA, 16-byte stores, fused
//     InstanceNorm statistics (gvfi_conv_params.stats) like the LDS-DMA kernel's.
// Results are bit-identical to the LDS-DMA kernel's (same K order: tap outer, channel inner; same epilogue arithmetic).
#include "conv_mma.h"
#include <type_traits>

struct P3sArgs {
    gvfi_conv_params p;
    int tiles_x, tiles_y, mtiles, per_xcd;
};

template <int BN, int CK, bool PROF = false> __global__ void __launch_bounds__(256) conv_p3x3s_kernel(P3sArgs a) {
    typedef bf16_t T;
    constexpr int NW = 4, NT = 256, WM = 64, MI = 2, NI = BN / 32, RBK = CK * 2, KK = CK / 16, BM = 256;
    constexpr int PW = 18, PIX = 18 * 18, PITCH = RBK + 16, SPP = RBK / 16 + 1;    // 16-byte slots per patch pixel incl. the pad slot
    constexpr int PIECES = (PIX * SPP + 63) / 64, PATCH = PIECES * 1024, QP = (PIECES + NW - 1) / NW;
    constexpr int BSTAGE = BN * RBK, BP = BSTAGE / 1024, B_INSTR = (BP + NW - 1) / NW;
    // weight ring: 4 stages for 64 input channels (the DMA of tap t+3 is issued during tap t: a tap is only 16-32 MFMAs per
    // wave, less than the ~1 us a DMA takes to land; the patch limits these variants to 2 workgroups per CU either way), 2
    // stages for 32 input channels (3-4 workgroups per CU hide the wait; a deeper ring would cost one of them).
    // Measured (tools/p3x3s_timeline.py, 64 -> 64 at 8 x 256 x 448): 25 kcycles per workgroup = prologue 10.5 k (the 512
    // resident workgroups request their 47 KB patches in the same microseconds: an HBM burst) + K loop 8.8 k + epilogue 5.7 k,
    // two workgroups per CU overlapping: 235 MB in 98 us = 2.4 TB/s, against 0.133 ms on the LDS-DMA kernel.
    constexpr int NSTB = (CK == 64 && BP % NW == 0) ? 4 : 2, AHEAD = NSTB - 1, CNT = BP / NW;
    constexpr int SP = BN * 2 + 16, STAGING = BM * SP;           // epilogue staging: 256 rows of BN bf16 + 16 bytes
    constexpr int RP = (STAGING + 1023) / 1024, RQ = (RP + NW - 1) / NW;
    constexpr int MAIN = (PATCH + NSTB * BSTAGE > RP * 1024) ? PATCH + NSTB * BSTAGE : RP * 1024;
    constexpr int U = BN / 8, RPI = NT / U;                      // 16-byte units per output row, rows per store iteration
    __shared__ __attribute__((aligned(16))) unsigned char smem[MAIN + 3 * BN * 4];
    const gvfi_conv_params& p = a.p;
    const int bid = blockIdx.x;
    const int mt = (bid & 7) * a.per_xcd + (bid >> 3);          // XCD-aware order, as conv_igemm_glds.hip
    if (mt >= a.mtiles) return;
    const int tiles_img = a.tiles_x * a.tiles_y;
    const int img = mt / tiles_img, trem = mt - img * tiles_img;
    const int tyi = trem / a.tiles_x, txi = trem - tyi * a.tiles_x;
    const int y0 = tyi * 16, x0 = txi * 16;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const T* __restrict__ xs0 = (const T*)p.x0;
    // PROF (algo bit 15): wave 0 writes shader-clock cycles of {prologue, K loop, epilogue, start time} to aux1[block * 4 ..]
    unsigned long long ph[4] = {0, 0, 0, 0}, tprev = 0;
    auto now = [&]() -> unsigned long long {
#ifndef GVFI_HOSTSIM
        return __builtin_readcyclecounter();
#else
        return 0;
#endif
    };
    if (PROF) { tprev = now(); ph[3] = tprev; }

    // ---- patch DMA: per-lane byte offsets relative to the patch origin (image pixel (y0-1, x0-1), may lie in the padding)
    unsigned a_off[QP];
#pragma unroll
    for (int q = 0; q < QP; ++q) {
        const int piece = q * NW + wave;
        const int s = piece * 64 + lane;
        const int pp = s / SPP, col = s - pp * SPP;
        const int py = pp / PW, px = pp - py * PW;
        const int gy = y0 - 1 + py, gx = x0 - 1 + px;
        const bool ok = piece < PIECES && pp < PIX && col < SPP - 1 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        a_off[q] = ok ? (unsigned)(((py * p.W + px) * p.ld0 + col * 8) * 2) : GVFI_DMA_OOB;
    }
    const long long pix_org = ((long long)img * p.H + (y0 - 1)) * p.W + (x0 - 1);
    const gvfi_i32x4 srd_a = make_srd(xs0 + pix_org * p.ld0);
    const gvfi_i32x4 srd_b = make_srd(p.w);
    // ---- weight stage DMA from the plain image [Cout][9][CK]: LDS row = output channel, slot s holds source slot s ^ swz(row)
    auto swz = [](int row) { return RBK == 128 ? (row >> 1) & 7 : (row >> 2) & 3; };
    unsigned b_off[B_INSTR];
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) {
        const int byte = (i * NW + wave) * 1024 + lane * 16;
        const int r = byte / RBK, s = (byte % RBK) >> 4;
        b_off[i] = (i * NW + wave < BP && r < p.Cout) ? (unsigned)(r * 9 * RBK + ((s ^ swz(r)) << 4)) : GVFI_DMA_OOB;
    }
    // ---- fragment read addresses
    unsigned abase[MI], b_rd[KK];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int row = wave * WM + i * 32 + (lane & 31);
        abase[i] = (unsigned)(((row >> 4) * PW + (row & 15)) * PITCH + (lane >> 5) * 16);
    }
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        const int rb = lane & 31;
        const int slot = 2 * kk + (lane >> 5);
        b_rd[kk] = PATCH + rb * RBK + ((slot ^ swz(rb)) << 4);
    }
    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const unsigned smem_lds = lds_address(smem);
    auto issue_b = [&](int tap) {
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i)
            if (i * NW + wave < BP) bufdma16(b_off[i], srd_b, (unsigned)(tap * RBK), smem_lds + PATCH + (tap % NSTB) * BSTAGE + (i * NW + wave) * 1024);
    };
    // ---- per-channel epilogue parameters -> LDS (published by the prologue barrier)
    if (tid < BN) {
        float* pt = (float*)(smem + MAIN);
        const bool in = tid < p.Cout;
        pt[tid] = (p.bias && in) ? p.bias[tid] : 0.f;
        pt[BN + tid] = (p.act1 == GVFI_ACT_PRELU && in) ? p.slope1[tid] : (p.act1 == GVFI_ACT_NONE ? 1.f : (p.act1 == GVFI_ACT_LRELU ? 0.1f : 0.f));
        pt[2 * BN + tid] = (p.act2 == GVFI_ACT_PRELU && in) ? p.slope2[tid] : (p.act2 == GVFI_ACT_NONE ? 1.f : (p.act2 == GVFI_ACT_LRELU ? 0.1f : 0.f));
    }
    // ---- prologue: the patch + the weights of tap 0
#pragma unroll
    for (int q = 0; q < QP; ++q)
        if (q * NW + wave < PIECES) bufdma16(a_off[q], srd_a, 0u, smem_lds + (q * NW + wave) * 1024);
#pragma unroll
    for (int t = 0; t < AHEAD; ++t) issue_b(t);

    uint4 fa[2][MI], fb[2][NI];
    auto load_frags = [&](int tap, int toff, int kk, int buf) {
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[buf][i] = *(const uint4*)(smem + abase[i] + toff + kk * 32);
#pragma unroll
        for (int j = 0; j < NI; ++j) fb[buf][j] = *(const uint4*)(smem + b_rd[kk] + (tap % NSTB) * BSTAGE + j * 32 * RBK);
    };
    // one tap: KK k-steps of MI x NI MFMAs; the weights of tap + AHEAD are issued behind the first MFMA group, the barrier
    // that publishes tap + 1 sits before the last k-step (conv_p3x3.hip / conv_igemm_glds.hip)
    auto step = [&](auto tap_tag) {
        constexpr int tap = decltype(tap_tag)::value;
        constexpr bool HAS_NEXT = tap < 8;
        constexpr int toff = ((tap / 3) * PW + (tap % 3)) * PITCH;
        constexpr int ntoff = (((tap + 1) / 3) * PW + ((tap + 1) % 3)) * PITCH;
#pragma unroll
        for (int kk = 0; kk + 1 < KK; ++kk) {
            load_frags(tap, toff, kk + 1, (kk + 1) & 1);
            GVFI_SCHED_BARRIER();
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int j = 0; j < NI; ++j) Mma2<T>::run(acc[i][j], fb[kk & 1][j], fa[kk & 1][i]);
                if (tap + AHEAD <= 8 && kk == 0 && i == 0) issue_b(tap + AHEAD);
            }
            GVFI_SCHED_BARRIER();
        }
        if (HAS_NEXT) {
            // the weights of tap + 1 have landed once only the younger stages (tap + 2 .. tap + AHEAD) are outstanding
            constexpr int later = (tap + AHEAD < 8 ? tap + AHEAD : 8) - (tap + 1);
            glds_wait_n<later * CNT>();
            __syncthreads();
            load_frags(tap + 1, ntoff, 0, KK & 1);
            GVFI_SCHED_BARRIER();
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) Mma2<T>::run(acc[i][j], fb[(KK - 1) & 1][j], fa[(KK - 1) & 1][i]);
        GVFI_SCHED_BARRIER();
    };
    glds_wait_n<(AHEAD - 1) * CNT>();
    __syncthreads();
    if (PROF) { const unsigned long long t = now(); ph[0] = t - tprev; tprev = t; }
    load_frags(0, 0, 0, 0);
    step(std::integral_constant<int, 0>{});
    step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{});
    step(std::integral_constant<int, 3>{});
    step(std::integral_constant<int, 4>{});
    step(std::integral_constant<int, 5>{});
    step(std::integral_constant<int, 6>{});
    step(std::integral_constant<int, 7>{});
    step(std::integral_constant<int, 8>{});
    if (PROF) { const unsigned long long t = now(); ph[1] = t - tprev; tprev = t; }

    // ---------------------------------------------------------------- epilogue (conv_p3x3.hip): y = act2(act1(acc + bias) + res) * out_scale
    // a lane holds one pixel (tile row wave*64 + i*32 + lane%32) and per accumulator block 4 x 4 consecutive output channels
    // (j*32 + 8*(r/4) + 4*(lane/32) + r%4); tile row r <-> output pixel (y0 + r/16, x0 + r%16)
    const int frow = lane & 31, fhalf = lane >> 5;
    const long long img_pix = (long long)img * p.H * p.W;
    auto pix_of = [&](int r, bool& ok) {
        const int y = y0 + (r >> 4), x = x0 + (r & 15);
        ok = y < p.H && x < p.W;
        return img_pix + (long long)y * p.W + x;
    };
    const bool has_sc = p.out_scale != 1.0f, has_res = p.res != nullptr;
    const float* ptab = (const float*)(smem + MAIN);
    __syncthreads();   // every wave is done reading the patch and the last weight stage
    if (has_res) {
        // residual tile -> staging area (row pitch SP; the result overwrites it in place)
        // (descriptor at the tile's first pixel, offsets relative to it: the tensor may exceed the 2 GB an offset spans)
        const long long pix0 = img_pix + (long long)y0 * p.W + x0;
        const gvfi_i32x4 srd_r = make_srd((const bf16_t*)p.res + pix0 * p.ldr);
#pragma unroll
        for (int q = 0; q < RQ; ++q) {
            const int piece = q * NW + wave;
            if (piece >= RP) continue;
            const int byte = piece * 1024 + lane * 16;
            const int row = byte / SP, unit = (byte % SP) >> 4;
            bool ok;
            const long long pix = pix_of(row < BM ? row : 0, ok);
            const unsigned off = (ok && row < BM && unit * 8 < p.Cout) ? (unsigned)((pix - pix0) * p.ldr * 2 + unit * 16) : GVFI_DMA_OOB;
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
                const int c0 = j * 32 + 8 * g + 4 * fhalf;          // first of this lane's 4 channels
                const float4 b4 = *(const float4*)(ptab + c0), s4 = *(const float4*)(ptab + BN + c0);
                const float bb[4] = {b4.x, b4.y, b4.z, b4.w}, s1[4] = {s4.x, s4.y, s4.z, s4.w};
                float s2[4] = {1.f, 1.f, 1.f, 1.f}, k1[4], k2[4];
                if (RES) {
                    const float4 z4 = *(const float4*)(ptab + 2 * BN + c0);
                    s2[0] = z4.x; s2[1] = z4.y; s2[2] = z4.z; s2[3] = z4.w;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    k1[e] = s1[e] <= 1.f ? inf : -inf;
                    k2[e] = s2[e] <= 1.f ? inf : -inf;
                }
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int row = wave * WM + i * 32 + frow;
                    unsigned char* sp = smem + row * SP + c0 * 2;
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
                        const uint2 ru = *(const uint2*)sp;
                        vv[0] += __builtin_bit_cast(float, ru.x << 16);
                        vv[1] += __builtin_bit_cast(float, ru.x & 0xffff0000u);
                        vv[2] += __builtin_bit_cast(float, ru.y << 16);
                        vv[3] += __builtin_bit_cast(float, ru.y & 0xffff0000u);
#pragma unroll
                        for (int e = 0; e < 4; ++e) vv[e] = med3f(vv[e], s2[e] * vv[e], k2[e]);
                    }
                    if (SC) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) vv[e] *= p.out_scale;
                    }
                    uint2 u;
                    u.x = pack_bf16x2(vv[0], vv[1]);
                    u.y = pack_bf16x2(vv[2], vv[3]);
                    *(uint2*)sp = u;
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
    {
        const int cg = tid % U, row_a = tid / U;
        const bool cok = cg * 8 < p.Cout;
        uint4 u[BM / RPI];
#pragma unroll
        for (int it = 0; it < BM / RPI; ++it) u[it] = *(const uint4*)(smem + (row_a + it * RPI) * SP + cg * 16);
        float st_sum[8], st_sq[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) st_sum[e] = st_sq[e] = 0.f;
        const bool do_stats = p.stats != nullptr;
#pragma unroll
        for (int it = 0; it < BM / RPI; ++it) {
            bool ok;
            const long long pix = pix_of(row_a + it * RPI, ok);
            if (ok && cok) {
                *(uint4*)((bf16_t*)p.y + pix * p.ldy + cg * 8) = u[it];
                if (do_stats) {   // fused InstanceNorm statistics of the values as stored (bf16-rounded), as conv_igemm_glds.hip
                    float sv[8];
                    unpack_bf16x8(u[it], sv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        st_sum[e] += sv[e];
                        st_sq[e] += sv[e] * sv[e];
                    }
                }
            }
        }
        if (do_stats) {   // workgroup reduction through the staging area: one atomic pair per (image, channel); a tile lies in one image
            __syncthreads();
            float* red = (float*)smem;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[tid * 16 + e] = st_sum[e];
                red[tid * 16 + 8 + e] = st_sq[e];
            }
            __syncthreads();
            if (tid < BN && tid < p.Cout) {
                const int cgi = tid >> 3, e = tid & 7;
                float s0 = 0.f, s1 = 0.f;
                for (int t = cgi; t < NT; t += U) {
                    s0 += red[t * 16 + e];
                    s1 += red[t * 16 + 8 + e];
                }
                gvfi_stats_add(p.stats, (long long)img * p.Cout + tid, s0, s1);
            }
        }
    }
    if (PROF && tid == 0) {
        ph[2] = now() - tprev;
        unsigned long long* o = (unsigned long long*)p.aux1 + (size_t)bid * 4;
        for (int k = 0; k < 4; ++k) o[k] = ph[k];
    }
}

// 1 = gvfi_conv2d routes this problem here ahead of the LDS-DMA kernel; 2 = runnable on request (algo 5) but too few output
// pixels for the 16 x 16 tiles to pay; 0 = not this kernel's problem
extern "C" int gvfi_conv2d_p3x3s_eligible(const gvfi_conv_params* pp) {
    const gvfi_conv_params& p = *pp;
    if (p.dtype != GVFI_BF16 || p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad_h != 1 || p.pad_w != 1) return 0;
    if (p.pad_mode != GVFI_PAD_ZEROS || p.groups > 1 || p.epi_mode != GVFI_EPI_STD || p.w_layout != 0) return 0;
    if ((p.c0 != 32 && p.c0 != 64) || p.c1 != 0 || p.Cout <= 0 || p.Cout > 64 || (p.Cout % 8)) return 0;
    if (p.Ho != p.H || p.Wo != p.W) return 0;
    if (p.y_f32 || (p.res != nullptr && p.res_f32) || p.act1 > GVFI_ACT_PRELU || p.act2 > GVFI_ACT_PRELU) return 0;
    if ((((uintptr_t)p.y) & 15) || ((p.ldy * 2) & 15) || (p.res && ((((uintptr_t)p.res) & 15) || ((p.ldr * 2) & 15)))) return 0;
    if (((uintptr_t)p.x0 & 15) || ((uintptr_t)p.w & 15) || (p.ld0 % 8)) return 0;
    // per-lane DMA offsets are 32-bit and stay below the descriptor's range
    if ((long long)18 * p.W * p.ld0 * 2 >= 0x7fffff00ll) return 0;
    if (p.res && (long long)18 * p.W * p.ldr * 2 >= 0x7fffff00ll) return 0;
    return (long long)p.N * p.H * p.W >= 65536 ? 1 : 2;
}

extern "C" int gvfi_conv2d_p3x3s(const gvfi_conv_params* pp, void* stream) {
    if (!gvfi_conv2d_p3x3s_eligible(pp)) return -2;
    const gvfi_conv_params& p = *pp;
    P3sArgs a;
    a.p = p;
    a.tiles_x = cdiv(p.W, 16);
    a.tiles_y = cdiv(p.H, 16);
    a.mtiles = a.tiles_x * a.tiles_y * p.N;
    a.per_xcd = cdiv((long long)a.mtiles, 8);
    const dim3 grid(a.per_xcd * 8), block(256);
    hipStream_t st = (hipStream_t)stream;
    GVFI_EMU_SERIAL(p.stats != nullptr);
    if (((p.algo >> 8) & 128) && p.aux1 != nullptr && p.c0 == 64 && p.Cout > 32) {
        GVFI_LAUNCH_COOP((conv_p3x3s_kernel<64, 64, true>), grid, block, st, a);
    } else if (p.c0 == 64) {
        if (p.Cout > 32) { GVFI_LAUNCH_COOP((conv_p3x3s_kernel<64, 64>), grid, block, st, a); }
        else { GVFI_LAUNCH_COOP((conv_p3x3s_kernel<32, 64>), grid, block, st, a); }
    } else {
        if (p.Cout > 32) { GVFI_LAUNCH_COOP((conv_p3x3s_kernel<64, 32>), grid, block, st, a); }
        else { GVFI_LAUNCH_COOP((conv_p3x3s_kernel<32, 32>), grid, block, st, a); }
    }
    return (int)hipGetLastError();
}
