// 7x7 stride-1 zero-padded convolution over FEW channels at full resolution (gfx950, bf16): the combination block of
// multi_flow_combine, 9 -> 18 (PReLU) -> 3 (+ mean image) (gimmvfi_r.py:60-64,305-308; AMT's comb_block) -- at 4K two
// launches over 8.9 Mpixel each, 13 of 91 ms of a step on the patch kernel (conv_patch.hip).
//
// What bounds the patch kernel there: with <= 32 output channels every 1 KiB fragment of input pixels read from LDS feeds
// ONE MFMA, and the LDS delivers ~75 B/clk to one wave per SIMD: 277 cycles per 4-MFMA step, twice the MFMA time, and
// Cout 18 / 3 is padded to 32.  Here the same fragment feeds SEVEN MFMAs:
//
//   * lanes run DOWN a column: an operand fragment = 16 vertically adjacent pixels x 32 K values, K = four 16-byte groups
//     (ky, 8 channels) of ONE input column xi.  The output column x and the horizontal tap kx only enter through
//     xi = x + kx, so the fragment of input column xi is the operand of (x, kx), (x+1, kx-1), ... : a wave that owns 8
//     adjacent output columns reads 14 input columns and issues 56 MFMAs per K step and block of 16 rows
//     (0.25 LDS reads per MFMA instead of 1.25);
//   * v_mfma_f32_16x16x32_bf16 with the WEIGHTS as the 16-row operand: output channels are padded to 16 (not 32), K steps
//     = ceil(7 * groups / 4) -- 9 -> 18: 4 steps x 2 channel blocks, 18 -> 3: 6 steps x 1 -- and the accumulator layout is
//     "lane = pixel, 4 registers = 4 consecutive channels", so the epilogue applies bias / PReLU in registers and
//     transposes through LDS with 8- / 16-byte writes;
//   * the seven weight fragments of a K step stay in registers for the step (each reloaded in place from LDS right after
//     its last use), the pixel fragments go through a ring of seven registers sets, column xi + 7 read into the slot of
//     column xi right after its MFMAs: every LDS read is issued 28 or more MFMAs ahead of its use;
//   * tile = 32 x 32 output pixels per 4-wave workgroup (patch 38 x 38, row pitch an odd multiple of 16 B: the 16 lanes of
//     a fragment read 16 different bank groups), persistent workgroups.  The WHOLE patch of the next tile is requested
//     into registers (12-18 x 16 bytes per thread) before the K loop of the current one and written to LDS after it: with
//     every CU in the same phase the 256 patches are an 18 MB burst that takes HBM ~6 k cycles -- behind 15-18 k cycles of
//     MFMAs it costs nothing, in front of them (LDS-DMA after the K loop, measured) it cost a third of the tile time;
//   * output rows leave LDS as whole 16-byte units of contiguous pixels (+ float residual) in a run-time loop of four
//     units per thread (unrolled, hipcc kept every unit's values live and spilled the prefetched patch -- and a scratch
//     reload waits for the HBM loads in flight, they share one counter: 1.0 instead of 0.5 ms).
// Summation order differs from the patch kernel (K is walked kx-major per input column): results agree to fp32 rounding
// of the accumulation, not bit for bit.
#include "conv_mma.h"

#define C7_T 32          // tile edge (output pixels)
#define C7_P 38          // patch edge
#define C7_CW 8          // output columns per wave

struct Col7Args {
    gvfi_conv_params p;
    int tiles_x, tiles_y, ntiles;
    int off_w, off_s;        // byte offsets of the weight fragments / the output staging area in LDS
    int sb, srow;            // staged bytes per output pixel (16-byte rounded channel extent), staging row pitch
    int upp;                 // 16-byte units per staged pixel
    unsigned upp_rcp, upr_rcp;   // k / upp == (k * upp_rcp) >> 16 for k < 128; u / (32 upp) == (u * upr_rcp) >> 20 for u < 8192
};

template <int GPT> struct Col7Geom {
    static constexpr int PITCH = GPT * 16;                               // LDS bytes per patch pixel
    static constexpr int RU = C7_P * GPT + 1;                            // 16-byte units per patch row: 38 pixels + one pad unit
    static constexpr int ROW = RU * 16;                                  // patch row pitch: an odd multiple of 16 bytes
    static constexpr int G = 7 * GPT;                                    // K groups of 16 bytes: (ky, channel group)
    static constexpr int NS = (G + 3) / 4;                               // MFMA steps of four groups
    static constexpr int UNITS = C7_P * RU;                              // the patch is ONE contiguous run of 16-byte units
};

template <int MB, int GPT>
__global__ void __launch_bounds__(256) conv_col7_kernel(Col7Args a) {
    typedef Col7Geom<GPT> GE;
    constexpr int PITCH = GE::PITCH, ROW = GE::ROW, G = GE::G, NS = GE::NS, RU = GE::RU;
    GVFI_DYN_SMEM(smem);
    const gvfi_conv_params& p = a.p;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, q = lane >> 4;
    unsigned char* wl = smem + a.off_w;
    unsigned char* stg = smem + a.off_s;
    const bf16_t* __restrict__ wsrc = (const bf16_t*)p.w;
    constexpr int CIN_PAD = GPT * 8;

    // ---- once per workgroup: weights in fragment order [step][kx][channel block][lane][16 B]: lane l holds the 8 values
    // w[n = 16 mb + (l & 15)][ky][kx][8 cg ..] of K group g = 4 step + (l >> 4) = (ky, cg); zeros beyond G / Cout
    for (int idx = tid; idx < NS * 7 * MB * 64; idx += 256) {
        const int l = idx & 63, f = idx >> 6;
        const int mb = f % MB, kx = (f / MB) % 7, s = f / (7 * MB);
        const int g = 4 * s + (l >> 4), n = mb * 16 + (l & 15);
        uint4 v;
        v.x = v.y = v.z = v.w = 0u;
        if (g < G && n < p.Cout) {
            const int ky = g / GPT, cg = g - ky * GPT;
            v = *(const uint4*)(wsrc + ((long long)n * 49 + ky * 7 + kx) * CIN_PAD + cg * 8);
        }
        *(uint4*)(wl + (long long)idx * 16) = v;
    }
    // epilogue constants of this lane: channels 16 mb + 4 q + r
    float cbias[MB][4], cs1[MB][4];
    {
        const float f1 = p.act1 == GVFI_ACT_NONE ? 1.f : (p.act1 == GVFI_ACT_LRELU ? 0.1f : 0.f);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ch = mb * 16 + 4 * q + r;
                const bool in = ch < p.Cout;
                cbias[mb][r] = (p.bias && in) ? p.bias[ch] : 0.f;
                cs1[mb][r] = (p.act1 == GVFI_ACT_PRELU) ? (in ? p.slope1[ch] : 0.f) : f1;
            }
    }

    // ---- input patch -> registers -> LDS.  The patch (38 rows of 38 pixels + one pad unit) is one contiguous run of
    // 16-byte units; thread t moves units t, t + 256, ...  = (row ty, pixel tx, channel group cg).  For a tile whose patch
    // lies inside the image the byte offset relative to the patch origin is tile-independent (rel[], computed once) and
    // the origin is a scalar; border tiles mask the units outside the image (zero padding = zeros in LDS).
    constexpr int NI = (GE::UNITS + 255) / 256;
    unsigned rel[NI];
    unsigned vmask0 = 0u;      // bit j: unit j of this thread is a pixel unit (not the row pad, inside the patch)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int unit = tid + 256 * j;
        const int ty = unit / RU, u = unit - ty * RU;
        const int tx = u / GPT, cg = u - tx * GPT;
        const bool ok = ty < C7_P && u < C7_P * GPT;
        rel[j] = ok ? (unsigned)(((ty * p.W + tx) * p.ld0 + cg * 8) * 2) : 0u;
        vmask0 |= ok ? (1u << j) : 0u;
    }
    const int tiles_img = a.tiles_x * a.tiles_y;
    uint4 pv[NI];
    auto issue_patch = [&](int tile_) {
        const int n_img = tile_ / tiles_img;
        const int trem = tile_ - n_img * tiles_img;
        const int tyi = trem / a.tiles_x, txi = trem - tyi * a.tiles_x;
        const int iy0 = tyi * C7_T - 3, ix0 = txi * C7_T - 3;
        // (border tiles only re-derive WHICH units are inside the image; the loads themselves are the same)
        unsigned m = vmask0;
        if (!(iy0 >= 0 && ix0 >= 0 && iy0 + C7_P <= p.H && ix0 + C7_P <= p.W)) {
            int tidv = tid;
            GVFI_OPAQUE_V(tidv);     // (re-derived per border tile: hoisted out of the tile loop these 2 x NI values cost registers)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int unit = tidv + 256 * j;
                const int ty = unit / RU, u = unit - ty * RU;
                const int tx = u / GPT;
                if (!((unsigned)(iy0 + ty) < (unsigned)p.H && (unsigned)(ix0 + tx) < (unsigned)p.W)) m &= ~(1u << j);
            }
        }
        const unsigned char* base = (const unsigned char*)p.x0 + (((long long)n_img * p.H + iy0) * p.W + ix0) * p.ld0 * 2;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            pv[j].x = pv[j].y = pv[j].z = pv[j].w = 0u;
            if ((m >> j) & 1u) pv[j] = *(const uint4*)(base + rel[j]);
        }
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int j = 0; j < NI; ++j)
            if (tid + 256 * j < GE::UNITS) *(uint4*)(smem + (tid + 256 * j) * 16) = pv[j];
    };

    if ((int)blockIdx.x < a.ntiles) issue_patch(blockIdx.x);
    const bool prof = ((p.algo >> 8) & 128) != 0 && tid == 0;
    unsigned ph[4] = {0, 0, 0, 0}, tprev = 0;      // (32-bit: a workgroup's phase totals stay far below 2^32 cycles)
    auto stamp = [&](int k) {
#ifndef GVFI_HOSTSIM
        if (prof) {
            const unsigned t = (unsigned)__builtin_readcyclecounter();
            if (k >= 0) ph[k] += t - tprev;
            tprev = t;
        }
#endif
    };
    stamp(-1);

    // this lane's patch offset for K step s: group g = 4 s + q = (ky, cg); groups beyond G carry zero weights and may
    // read any finite data (group 0)
    auto step_off = [&](int s) {
        int g = 4 * s + q;
        if (g >= G) g = 0;
        const int ky = g / GPT, cg = g - ky * GPT;
        return (li + ky) * ROW + cg * 16 + wave * C7_CW * PITCH;
    };

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int n_img = tile / tiles_img;
        const int trem = tile - n_img * tiles_img;
        const int tyi = trem / a.tiles_x, txi = trem - tyi * a.tiles_x;
        const int oy0 = tyi * C7_T, ox0 = txi * C7_T;
        store_patch();     // (requested before the previous K loop; all waves left that loop before the staging barrier)
        __syncthreads();   // patch complete (first tile: the weights too); previous tile's staging area fully read
        stamp(0);
        if (tile + (int)gridDim.x < a.ntiles) issue_patch(tile + gridDim.x);     // in flight behind the K loop

        f32x4 acc[C7_CW][2][MB];
#pragma unroll
        for (int c = 0; c < C7_CW; ++c)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[c][rb][mb][r] = 0.f;

        // ---- K loop over phases (step s, row block rb) of 14 input columns.  Registers hold the 7 x MB weight fragments of
        // the step and a ring of 7 pixel fragments: the fragment of input column xi + 7 (of the next phase when that runs
        // over) is read into the slot of column xi right after its MFMAs, the weights of tap kx for the next step right
        // after their last use -- every LDS read is issued 28 x MB or more MFMAs ahead of its use.
        constexpr int NX = C7_CW + 6, RING = 7;
        uint4 wf[7][MB], fa[RING];
        {
            const unsigned char* ap = smem + step_off(0);
#pragma unroll
            for (int kx = 0; kx < 7; ++kx)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) wf[kx][mb] = *(const uint4*)(wl + ((long long)(kx * MB + mb) * 64 + lane) * 16);
#pragma unroll
            for (int xi = 0; xi < RING; ++xi) fa[xi] = *(const uint4*)(ap + xi * PITCH);
        }
#pragma unroll 1
        for (int s = 0; s < NS; ++s) {
            const int sn = s + 1 < NS ? s + 1 : s;          // (last step: the reloads fetch valid, unused data)
            const unsigned char* ap0 = smem + step_off(s);                // row block 0 / 1 of this step
            const unsigned char* ap1 = ap0 + 16 * ROW;
            const unsigned char* an0 = smem + step_off(sn);               // row block 0 of the next step
            const unsigned char* wp = wl + ((long long)sn * 7 * MB * 64 + lane) * 16;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
                for (int xi = 0; xi < NX; ++xi) {
#pragma unroll
                    for (int kx = 0; kx < 7; ++kx) {
                        const int c = xi - kx;
                        if (c < 0 || c >= C7_CW) continue;
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb) acc[c][rb][mb] = mfma_bf16_16x16x32(wf[kx][mb], fa[xi % RING], acc[c][rb][mb]);
                    }
                    GVFI_SCHED_BARRIER();
                    {   // slot xi % 7 <- column xi + 7 of this phase, or column xi - 7 of the next one
                        const int xn = xi + RING;
                        const unsigned char* src = xn < NX ? (rb == 0 ? ap0 : ap1) + xn * PITCH : (rb == 0 ? ap1 : an0) + (xn - NX) * PITCH;
                        fa[xi % RING] = *(const uint4*)src;
                    }
                    if (rb == 1 && xi >= C7_CW - 1) {       // weights of tap kx = xi - 7 are not used again in this step
                        const int kx = xi - (C7_CW - 1);
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb) wf[kx][mb] = *(const uint4*)(wp + (long long)(kx * MB + mb) * 1024);
                    }
                    GVFI_SCHED_BARRIER();
                }
            }
        }
        stamp(1);

        // ---- epilogue 1: activation in registers (lane = pixel (rb*16 + li, wave*8 + c), channels 16 mb + 4 q + r),
        // converted to the output type, one 8- / 16-byte LDS write per (pixel, channel quad)
        const int cround = a.sb / (p.y_f32 ? 4 : 2);      // staged channels
        const bool scaled = p.out_scale != 1.0f;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int ch0 = mb * 16 + 4 * q;
            if (ch0 >= cround) continue;
#pragma unroll
            for (int c = 0; c < C7_CW; ++c)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float t = acc[c][rb][mb][r] + cbias[mb][r];
                        v[r] = fmaxf(t, 0.f) + cs1[mb][r] * fminf(t, 0.f);
                    }
                    if (scaled) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] *= p.out_scale;
                    }
                    unsigned char* d = stg + (rb * 16 + li) * a.srow + (wave * C7_CW + c) * a.sb;
                    if (p.y_f32) *(float4*)(d + ch0 * 4) = make_float4(v[0], v[1], v[2], v[3]);
                    else {
                        uint2 u;
                        u.x = pack_bf16x2(v[0], v[1]);
                        u.y = pack_bf16x2(v[2], v[3]);
                        *(uint2*)(d + ch0 * 2) = u;
                    }
                    GVFI_SCHED_BARRIER();     // (else hipcc copies all accumulators to VGPRs first: 128 live values + the prefetched patch)
                }
        }
        __syncthreads();   // staging complete; every wave is past its K loop (the patch may be overwritten)
        stamp(2);
        // ---- epilogue 2: whole 16-byte units, contiguous along image rows (+ float residual of the same geometry).  A
        // run-time loop over groups of four units per thread (unrolled, hipcc keeps every unit's values live at once and
        // spills the prefetched patch: a scratch reload then waits for the HBM loads in flight -- they share one counter)
        {
            const bool inner = oy0 + C7_T <= p.Ho && ox0 + C7_T <= p.Wo;
            const long long t0 = ((long long)n_img * p.Ho + oy0) * p.Wo + ox0;
            const unsigned ysz = (unsigned)(p.ldy * (p.y_f32 ? 4 : 2)), rsz = (unsigned)(p.ldr * 4);
            unsigned char* yb = (unsigned char*)p.y + t0 * ysz;
            const unsigned char* rb_ = (const unsigned char*)p.res + t0 * rsz;
            const int nunits = C7_T * C7_T * a.upp, upr = C7_T * a.upp;
            // algo bit 6: the 3-channel float result leaves as imgt_pred = clamp((y + 1) / 2, 0, 1) in PLANAR (N, 3, Ho, Wo)
            // float at y2 (gimmvfi_r.py:308, fi_components.py:92) -- the arithmetic of gvfi_finalize_image on the value that
            // would have been stored to y; the NHWC y itself is not written (one pass less over the full-resolution frame)
            const bool planar = (p.algo & 64) != 0;
            float* pl = (float*)p.y2 + (long long)n_img * 3 * p.Ho * p.Wo + (long long)oy0 * p.Wo + ox0;
            const long long plane = (long long)p.Ho * p.Wo;
            constexpr int PF = 4;
#pragma unroll 1
            for (int u0 = tid; u0 < nunits; u0 += 256 * PF) {
                uint4 val[PF];
                float4 rv[PF];
                unsigned pixo[PF], jj[PF];
                bool on[PF];
#pragma unroll
                for (int k = 0; k < PF; ++k) {
                    const int u = u0 + k * 256;
                    const int row = (int)(((unsigned)u * a.upr_rcp) >> 20), kk = u - row * upr;
                    const int px = (int)(((unsigned)kk * a.upp_rcp) >> 16), j = kk - px * a.upp;
                    on[k] = u < nunits && (inner || (oy0 + row < p.Ho && ox0 + px < p.Wo));
                    pixo[k] = (unsigned)(row * p.Wo + px);
                    jj[k] = (unsigned)j;
                    if (on[k]) {
                        val[k] = *(const uint4*)(stg + row * a.srow + kk * 16);
                        if (p.res) rv[k] = *(const float4*)(rb_ + (pixo[k] * rsz + jj[k] * 16u));
                    }
                }
#pragma unroll
                for (int k = 0; k < PF; ++k) {
                    if (!on[k]) continue;
                    if (p.res) {
                        float4 f = __builtin_bit_cast(float4, val[k]);
                        const int cl = p.Cout - 4 * (int)jj[k];    // live channels of this piece; the residual's pad channels may hold anything
                        f.x += rv[k].x;
                        f.y = cl > 1 ? f.y + rv[k].y : 0.f;
                        f.z = cl > 2 ? f.z + rv[k].z : 0.f;
                        f.w = cl > 3 ? f.w + rv[k].w : 0.f;
                        val[k] = __builtin_bit_cast(uint4, f);
                    }
                    if (planar) {
                        const float4 f = __builtin_bit_cast(float4, val[k]);
                        pl[pixo[k]] = fminf(fmaxf((f.x + 1.0f) / 2.0f, 0.f), 1.f);
                        pl[plane + pixo[k]] = fminf(fmaxf((f.y + 1.0f) / 2.0f, 0.f), 1.f);
                        pl[2 * plane + pixo[k]] = fminf(fmaxf((f.z + 1.0f) / 2.0f, 0.f), 1.f);
                        continue;
                    }
                    *(uint4*)(yb + (pixo[k] * ysz + jj[k] * 16u)) = val[k];
                }
            }
        }
        stamp(3);
    }
#ifndef GVFI_HOSTSIM
    if (prof)
        for (int k = 0; k < 4; ++k) ((unsigned long long*)p.aux1)[(long long)blockIdx.x * 4 + k] = (unsigned long long)ph[k];
#endif
}

// LDS plan; returns total bytes (0 = not eligible)
static int col7_plan(const gvfi_conv_params& p, Col7Args& a) {
    if (p.dtype != GVFI_BF16 || p.KH != 7 || p.KW != 7 || p.stride != 1 || p.pad_h != 3 || p.pad_w != 3) return 0;
    if (p.pad_mode != GVFI_PAD_ZEROS || p.c1 != 0 || p.x1 != nullptr || p.groups > 1 || p.epi_mode != GVFI_EPI_STD) return 0;
    if (p.w_layout != 0 || p.stats != nullptr || p.Cout <= 0 || p.Cout > 32 || p.c0 <= 0 || p.c0 > 24 || (p.c0 % 8) || (p.ld0 % 8)) return 0;
    if (p.act1 > GVFI_ACT_PRELU || p.act2 != GVFI_ACT_NONE || !(p.algo & 16)) return 0;     // pad16: whole 16-byte units of y
    const int ey = p.y_f32 ? 4 : 2, unit = p.y_f32 ? 4 : 8;
    if ((p.ldy * ey) % 16 || ((uintptr_t)p.y & 15) || ((uintptr_t)p.x0 & 15) || ((uintptr_t)p.w & 15)) return 0;
    if (p.res != nullptr && (!p.res_f32 || !p.y_f32 || ((p.ldr * 4) % 16) || ((uintptr_t)p.res & 15))) return 0;
    if (p.res != nullptr && p.out_scale != 1.0f) return 0;     // (the scale is applied before the residual here, after it in the other kernels)
    if ((p.algo & 64) && (p.Cout != 3 || !p.y_f32 || p.y2 == nullptr || ((uintptr_t)p.y2 & 3))) return 0;   // planar finalize: 3 float channels
    a.p = p;
    const int cround = (p.Cout + unit - 1) / unit * unit;
    a.sb = cround * ey;
    a.upp = a.sb / 16;
    a.upp_rcp = 65536u / (unsigned)a.upp + 1u;
    a.upr_rcp = (1u << 20) / (unsigned)(C7_T * a.upp) + 1u;
    a.srow = C7_T * a.sb + 16;
    const int gpt = p.c0 / 8, mb = p.Cout > 16 ? 2 : 1;
    const int units = C7_P * (C7_P * gpt + 1);
    const int ns = (7 * gpt + 3) / 4;
    const int patch = (units * 16 + 1023) & ~1023;          // (16-byte units, rounded up to 1 KiB)
    if ((long long)C7_P * p.W * p.ld0 * 2 >= 0x7fff0000ll || (long long)C7_T * p.Wo * (p.ldy * ey > p.ldr * 4 ? p.ldy * ey : p.ldr * 4) >= 0x7fff0000ll) return 0;   // 32-bit offsets inside a tile
    a.off_w = patch;
    a.off_s = a.off_w + ns * 7 * mb * 1024;
    const int total = a.off_s + C7_T * a.srow;
    a.tiles_x = cdiv(p.Wo, C7_T);
    a.tiles_y = cdiv(p.Ho, C7_T);
    a.ntiles = a.tiles_x * a.tiles_y * p.N;
    if (total > 160 * 1024 || (gpt == 3 && mb == 2)) return 0;
    return total;
}

// 1 = gvfi_conv2d routes this problem here (ahead of the patch kernel)
extern "C" int gvfi_conv2d_col7_eligible(const gvfi_conv_params* pp) {
    Col7Args a;
    if (col7_plan(*pp, a) == 0) return 0;
    if (pp->Wo < 32 || pp->Ho < 32 || (long long)pp->N * pp->Ho * pp->Wo < 65536) return 2;    // runnable (algo 7), not routed
    return 1;
}

extern "C" int gvfi_conv2d_col7(const gvfi_conv_params* pp, void* stream) {
    const gvfi_conv_params& p = *pp;
    Col7Args a;
    const int shm = col7_plan(p, a);
    if (shm == 0) return -2;
    hipStream_t st = (hipStream_t)stream;
    int grid = a.ntiles < 256 ? a.ntiles : 256;      // persistent: one workgroup per CU (one wave per SIMD holds a whole K step)
    if (grid < 1) grid = 1;
    const int gpt = p.c0 / 8, mb = p.Cout > 16 ? 2 : 1;
#define C7_LAUNCH(MM, GG) GVFI_LAUNCH_COOP_SHM((conv_col7_kernel<MM, GG>), dim3(grid), dim3(256), shm, st, a)
    // (the combinations that fit the 160 KiB of LDS: patch + weight fragments + output staging)
    if (gpt == 1 && mb == 1) { C7_LAUNCH(1, 1); }
    else if (gpt == 1) { C7_LAUNCH(2, 1); }
    else if (gpt == 2 && mb == 1) { C7_LAUNCH(1, 2); }
    else if (gpt == 2) { C7_LAUNCH(2, 2); }
    else if (gpt == 3 && mb == 1) { C7_LAUNCH(1, 3); }
    else return -2;
#undef C7_LAUNCH
    return (int)hipGetLastError();
}
