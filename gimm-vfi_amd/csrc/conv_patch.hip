// "Patch" convolution for the few-channel layers of the path (gfx950): 7x7 / 5x5 / 3x3 filters over 2..64 input and
// <= 64 output channels at FULL resolution -- the combination block of multi_flow_combine (9 -> 18 -> 3, 7x7,
// gimmvfi_r.py:60-64,305-308), the encoder stems (3 -> 64, 7x7 stride 2, raft/extractor.py:137), the reflect-padded
// 3x3 layers of the motion encoder / latent refiner (gimmvfi_r.py:84-109) and the decoder front ends (2 -> 16,
// 8 -> 32 5x5, fi_components.py:284-295).  These shapes cannot feed the LDS-DMA implicit GEMM (channel counts are no
// multiple of a 128-byte K chunk, or the padding is `reflect`) and ran on the register-staged kernel at 20-90 TFLOP/s:
// it fetches every input pixel once per filter tap -- 49 times for a 7x7 -- through VGPRs.  At 4K the two 7x7 layers
// alone were 26 ms of a 143 ms step.
//
// Here one workgroup owns an 8 x 64 block of output pixels and reads its input ONCE:
//   * the (8-1)*S+KH x (64-1)*S+KW halo patch of the input goes to LDS (zero or reflect padding resolved while loading),
//     pixel pitch = channel bytes rounded up to an odd multiple of 16 B so that the fragment reads of 16 neighbouring
//     pixels fall into different banks;
//   * the whole weight tensor goes to LDS once per workgroup, already in MFMA B-fragment order ([step][lane][16 B]);
//     workgroups are persistent (grid = a multiple of the CU count, tiles strided), so this happens once per CU;
//   * K = taps x channel groups of 16 bytes, two groups per v_mfma_f32_32x32x16_bf16 (lanes 0-31 / 32-63 take one
//     group each: a group is 8 consecutive channels of ONE tap, i.e. one aligned 16-byte LDS read at pixel + tap offset);
//     every wave owns 4 row-blocks of 32 pixels x NB column blocks of 32 output channels;
//   * epilogue through LDS with the shared 8-channel group routine of the other convolution kernels (bias, activation,
//     residual, second activation, f32 / bf16 stores, ragged channel counts).
// fp32 validation mode uses the same code with 4-channel groups and v_mfma_f32_32x32x2_f32.
#include "conv_mma.h"

struct PatchArgs {
    gvfi_conv_params p;
    int gpt;          // 16-byte channel groups per tap (c0 / VE)
    int G;            // K groups = KH*KW*gpt
    int G2;           // MFMA steps = ceil(G / 2) rounded up to a multiple of 4 (LDS holds 3 more, all zero)
    int pitch;        // bytes between patch pixels in LDS
    int tph, twp;     // patch height / width in pixels
    int tiles_x, tiles_y, ntiles;
    int off_w, off_t; // byte offsets of the weight image / the step table inside the dynamic LDS
    unsigned gpt_mul, gpt_sh, twp_mul, twp_sh;   // x / d == (umulhi(x, mul) + x) >> sh  (gvfi_magic_div)
};

#define GVFI_PATCH_TH 8
#define GVFI_PATCH_TW 64

template <typename T, int S, int NB>
__global__ void __launch_bounds__(256) conv_patch_kernel(PatchArgs a) {
    constexpr int VE = Elem<T>::VE;
    constexpr int TH = GVFI_PATCH_TH, TW = GVFI_PATCH_TW;
    GVFI_DYN_SMEM(smem);
    const gvfi_conv_params& p = a.p;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, li = lane & 31;
    unsigned char* wl = smem + a.off_w;
    int* tab = (int*)(smem + a.off_t);
    const T* __restrict__ x0 = (const T*)p.x0;
    const T* __restrict__ wsrc = (const T*)p.w;
    const int taps = p.KH * p.KW;
    const int cin_pad = a.gpt * VE;

    // ---- once per workgroup: weights in B-fragment order + the per-group patch offsets
    // MFMA step j, lane l: group g = 2j + (l >> 5), output channel n = nb*32 + (l & 31): the 16 bytes
    // w[n][tap(g)][cg(g)*VE ..] of the plain [Cout][KH][KW][cin_pad] image (zeros beyond G / Cout)
    for (int idx = tid; idx < (a.G2 + 3) * NB * 64; idx += 256) {
        const int l = idx & 63, jn = idx >> 6;
        const int nb = jn % NB, j = jn / NB;
        const int g = 2 * j + (l >> 5), n = nb * 32 + (l & 31);
        uint4 v;
        v.x = v.y = v.z = v.w = 0u;
        if (g < a.G && n < p.Cout) {
            const int tap = g / a.gpt, cg = g - tap * a.gpt;
            v = *(const uint4*)(wsrc + ((long long)n * taps + tap) * cin_pad + cg * VE);
        }
        *(uint4*)(wl + (long long)idx * 16) = v;
    }
    for (int g = tid; g < 2 * (a.G2 + 11); g += 256) {
        int off = 0;
        if (g < a.G) {
            const int tap = g / a.gpt, cg = g - tap * a.gpt;
            const int ky = tap / p.KW, kx = tap - ky * p.KW;
            off = (ky * a.twp + kx) * a.pitch + cg * 16;
        }
        tab[g] = off;
    }
    // fragment base offsets of this wave's four 32-pixel blocks: rows 2*wave, 2*wave+1 x column halves
    int abase[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int row = 2 * wave + (b >> 1), col = (b & 1) * 32 + li;
        abase[b] = ((row * S) * a.twp + col * S) * a.pitch;
    }

    // ---- input patch -> LDS, one 16-byte group per item.  UNR loads are issued back to back before the first LDS store
    // (with one wave per SIMD a load -> store loop runs one HBM/L2 round trip per iteration), and the first UNR items of
    // the NEXT tile are requested before the K loop of the current one and stored after its epilogue, so their round trip
    // hides behind the MFMAs.  Divisions by the uniform patch width / groups per tap are host magic numbers.
    constexpr int UNR = 8;
    const int items = a.tph * a.twp * a.gpt;
    const int tiles_img = a.tiles_x * a.tiles_y;
    auto issue_batch = [&](int tile_, int base, uint4 (&v)[UNR], int (&dst)[UNR]) {
        const int n_img = tile_ / tiles_img;
        const int trem = tile_ - n_img * tiles_img;
        const int tyi = trem / a.tiles_x, txi = trem - tyi * a.tiles_x;
        const int iy0 = tyi * TH * S - p.pad_h, ix0 = txi * TW * S - p.pad_w;
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int it = base + u * 256 + tid;
            v[u].x = v[u].y = v[u].z = v[u].w = 0u;
            dst[u] = -1;
            if (it < items) {
                const unsigned px = (__umulhi((unsigned)it, a.gpt_mul) + (unsigned)it) >> a.gpt_sh;     // it / gpt
                const int cg = it - (int)px * a.gpt;
                const unsigned ty = (__umulhi(px, a.twp_mul) + px) >> a.twp_sh;                           // px / twp
                const int tx = (int)px - (int)ty * a.twp;
                int gy = iy0 + (int)ty, gx = ix0 + tx;
                bool ok = true;
                if (p.pad_mode == GVFI_PAD_REFLECT) {
                    // rows / columns of the patch that only serve output pixels beyond the image may map anywhere valid
                    gy = reflect_idx(gy < -(p.H - 1) ? 0 : (gy > 2 * p.H - 2 ? p.H - 1 : gy), p.H);
                    gx = reflect_idx(gx < -(p.W - 1) ? 0 : (gx > 2 * p.W - 2 ? p.W - 1 : gx), p.W);
                } else {
                    ok = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
                }
                dst[u] = (int)px * a.pitch + cg * 16;
                if (ok) v[u] = *(const uint4*)(x0 + ((long long)(n_img * p.H + gy) * p.W + gx) * p.ld0 + cg * VE);
            }
        }
    };
    auto store_batch = [&](const uint4 (&v)[UNR], const int (&dst)[UNR]) {
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            if (dst[u] >= 0) *(uint4*)(smem + dst[u]) = v[u];
    };
    // ---- epilogue constants of this thread, per 32-channel block: (pixel, 8-channel group) items -- with 1 / 2 live
    // groups in a block all threads still take pixels -- and the group's bias / slopes, loaded once per workgroup (inside
    // the tile loop their round trip was a sixth of a tile).  Slim path for the activation family none / ReLU / LeakyReLU
    // / PReLU written as max(v,0) + s*min(v,0) (exact for every member: s = 1, 0, 0.1, slope[c]); anything else takes the
    // shared generic routine.
    const bool slim = p.act1 <= GVFI_ACT_PRELU && p.act2 <= GVFI_ACT_PRELU;
    int ep_gshift[NB], ep_nvalid[NB];
    GroupConst ep_gc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int live = p.Cout - nb * 32;
        ep_gshift[nb] = live <= 8 ? 0 : (live <= 16 ? 1 : 2);
        const int cout0 = nb * 32 + (tid & ((1 << ep_gshift[nb]) - 1)) * 8;
        const int n_valid = p.Cout - cout0 >= 8 ? 8 : (p.Cout - cout0 > 0 ? p.Cout - cout0 : 0);
        ep_nvalid[nb] = n_valid;
        const float f1 = p.act1 == GVFI_ACT_NONE ? 1.f : (p.act1 == GVFI_ACT_LRELU ? 0.1f : 0.f);
        const float f2 = p.act2 == GVFI_ACT_NONE ? 1.f : (p.act2 == GVFI_ACT_LRELU ? 0.1f : 0.f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool in = e < n_valid;
            ep_gc[nb].bias[e] = (p.bias && in) ? p.bias[cout0 + e] : 0.f;
            ep_gc[nb].s1[e] = (p.act1 == GVFI_ACT_PRELU && in) ? p.slope1[cout0 + e] : (slim ? f1 : 0.f);
            ep_gc[nb].s2[e] = (p.act2 == GVFI_ACT_PRELU && in) ? p.slope2[cout0 + e] : (slim ? f2 : 0.f);
        }
    }

    uint4 pv[UNR];
    int pd[UNR];
    if ((int)blockIdx.x < a.ntiles) issue_batch(blockIdx.x, 0, pv, pd);
    // profiling only (algo bit 15): cycles per phase summed over this workgroup's tiles -> aux1[block*4 + phase]
    const bool prof = ((p.algo >> 8) & 128) != 0 && tid == 0;
    unsigned long long ph[4] = {0, 0, 0, 0}, tprev = 0;
    auto stamp = [&](int k) {
#ifndef GVFI_HOSTSIM
        if (prof) {
            const unsigned long long t = __builtin_readcyclecounter();
            if (k >= 0) ph[k] += t - tprev;
            tprev = t;
        }
#endif
    };
    stamp(-1);

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int n_img = tile / tiles_img;
        const int trem = tile - n_img * tiles_img;
        const int tyi = trem / a.tiles_x, txi = trem - tyi * a.tiles_x;
        const int oy0 = tyi * TH, ox0 = txi * TW;
        __syncthreads();   // previous tile's epilogue reads of the staging area are complete
        store_batch(pv, pd);
        for (int base = 256 * UNR; base < items; base += 256 * UNR) {
            uint4 v[UNR];
            int dst[UNR];
            issue_batch(tile, base, v, dst);
            store_batch(v, dst);
        }
        __syncthreads();
        stamp(0);   // patch staged
        if (tile + (int)gridDim.x < a.ntiles) issue_batch(tile + gridDim.x, 0, pv, pd);

        f32x16 acc[4][NB];
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[b][nb][r] = 0.f;

        // ---- K loop: the fragments of step j+3 are requested before the MFMAs of step j (ring of 4 register sets).  With
        // one wave per SIMD nothing else covers the LDS round trip, and one step of 4 MFMAs (128 cycles) is shorter than
        // it: at a distance of one step the loop ran at 290 cycles per step.  The step count is padded to a multiple
        // of 4 plus look-ahead steps of zero weights, so the body has no branches (with conditionals around the MFMA
        // groups hipcc shuttled all accumulators between AGPRs and VGPRs in every iteration); patch offsets come from
        // the table one iteration (4 steps) before their fragments.
        uint4 fa[4][4], fb[4][NB];
        auto load_step = [&](int j, int off, int buf) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) fb[buf][nb] = *(const uint4*)(wl + ((long long)(j * NB + nb) * 64 + lane) * 16);
#pragma unroll
            for (int b = 0; b < 4; ++b) fa[buf][b] = *(const uint4*)(smem + abase[b] + off);
        };
        const int* tb = tab + half;
        int offs[4];
#pragma unroll
        for (int u = 0; u < 3; ++u) load_step(u, tb[2 * u], u);
#pragma unroll
        for (int u = 0; u < 4; ++u) offs[u] = tb[2 * (u + 3)];
        for (int j = 0; j < a.G2; j += 4) {
            int nxt[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) nxt[u] = tb[2 * (j + u + 7)];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                load_step(j + u + 3, offs[u], (u + 3) & 3);
                GVFI_SCHED_BARRIER();
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) Mma2<T>::run(acc[b][nb], fa[u][b], fb[u][nb]);
                GVFI_SCHED_BARRIER();
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) offs[u] = nxt[u];
        }

        stamp(1);   // prefetch issue + K loop
        // ---- epilogue: one 32-channel block at a time through LDS ([512 pixels][32] floats over the patch area)
        float* cs = (float*)smem;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            __syncthreads();   // patch (or the previous block's staging) no longer read
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int row = 2 * wave + (b >> 1), col0 = (b & 1) * 32;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
                    cs[(row * TW + col0 + m) * 32 + li] = acc[b][nb][r];
                }
            }
            __syncthreads();
            if (nb == 0) stamp(2);   // accumulators staged
            const int gshift = ep_gshift[nb], my_cg = tid & ((1 << gshift) - 1);
            const int cout0 = nb * 32 + my_cg * 8;
            const int n_valid = ep_nvalid[nb];
            if (n_valid > 0) {
                const GroupConst& gc = ep_gc[nb];
                const int eY = p.y_f32 ? 4 : (int)sizeof(T);
                // algo bit 4 ("pad16", set by a host that owns the channel padding of y and res): a ragged last group may
                // be accessed in whole 16-byte units -- the pad channels of y receive zeros.  Without it the 18-channel
                // output of the combination block is written as 36 of every 48 bytes with 2-byte stores for the tail:
                // partially written memory lines, measured at a tenth of the store rate of whole lines.
                const bool pad16 = (p.algo & 16) != 0 && n_valid < 8;
                const int nst = pad16 ? (p.y_f32 ? ((n_valid + 3) & ~3) : 8) : n_valid;      // channels stored by this item
                const bool vec = (n_valid == 8 || pad16) && vec_ok(p.y, p.ldy, eY) && (((uintptr_t)p.y + (size_t)cout0 * eY) & 15) == 0 &&
                                 vec_ok(p.res, p.ldr, p.res_f32 ? 4 : (int)sizeof(T)) &&
                                 (p.res == nullptr || (((uintptr_t)p.res + (size_t)cout0 * (p.res_f32 ? 4 : sizeof(T))) & 15) == 0);
                const bool has_res = p.res != nullptr, has_a2 = p.act2 != GVFI_ACT_NONE, has_sc = p.out_scale != 1.0f;
                constexpr bool BF = sizeof(T) == 2;
                if (!slim) {
                    // (kept out of the unrolled loop below: the generic routine is large, and a store loop that does not
                    // fit the instruction cache runs at a fraction of its speed)
#pragma unroll 1
                    for (int lp = tid >> gshift; lp < TH * TW; lp += 256 >> gshift) {
                        const int oy = oy0 + lp / TW, ox = ox0 + lp % TW;
                        if (oy >= p.Ho || ox >= p.Wo) continue;
                        const float4 c0 = *(const float4*)(cs + lp * 32 + my_cg * 8);
                        const float4 c1 = *(const float4*)(cs + lp * 32 + my_cg * 8 + 4);
                        float vv[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
                        epilogue_group<T>(p, gc, vv, cout0, n_valid, ((long long)n_img * p.Ho + oy) * p.Wo + ox, vec);
                    }
                } else {
                    // every residual value of this thread's (at most 8) items is requested before the first store: loads
                    // and stores retire through one in-order counter, so a load issued behind a store waits for the
                    // store's acknowledgement (~1 us each)
                    const int iters = 2 << gshift, lstep = 256 >> gshift, lp0 = tid >> gshift;
                    float rr[8][8];
                    if (has_res) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) rr[k][e] = 0.f;
                            const int lp = lp0 + k * lstep;
                            const int oy = oy0 + lp / TW, ox = ox0 + lp % TW;
                            if (k < iters && oy < p.Ho && ox < p.Wo) {
                                const long long pix = ((long long)n_img * p.Ho + oy) * p.Wo + ox;
                                if (vec && !pad16) ld8<T>(p.res, pix * p.ldr + cout0, p.res_f32, BF, rr[k]);
                                else if (vec && (p.res_f32 || !BF)) {     // whole float4 units up to the padded end
                                    const float* rp = (const float*)p.res + pix * p.ldr + cout0;
                                    const float4 q0 = *(const float4*)rp;
                                    rr[k][0] = q0.x; rr[k][1] = q0.y; rr[k][2] = q0.z; rr[k][3] = q0.w;
                                    if (n_valid > 4) {
                                        const float4 q1 = *(const float4*)(rp + 4);
                                        rr[k][4] = q1.x; rr[k][5] = q1.y; rr[k][6] = q1.z; rr[k][7] = q1.w;
                                    }
                                } else if (vec) ld8<T>(p.res, pix * p.ldr + cout0, 0, BF, rr[k]);
                                else {
#pragma unroll
                                    for (int e = 0; e < 8; ++e)
                                        if (e < n_valid) rr[k][e] = ld_any<T>(p.res, pix * p.ldr + cout0 + e, p.res_f32);
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int lp = lp0 + k * lstep;
                        const int oy = oy0 + lp / TW, ox = ox0 + lp % TW;
                        if (k >= iters || oy >= p.Ho || ox >= p.Wo) continue;
                        const float4 c0 = *(const float4*)(cs + lp * 32 + my_cg * 8);
                        const float4 c1 = *(const float4*)(cs + lp * 32 + my_cg * 8 + 4);
                        float vv[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
                        const long long pix = ((long long)n_img * p.Ho + oy) * p.Wo + ox;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float t = vv[e] + gc.bias[e];
                            vv[e] = fmaxf(t, 0.f) + gc.s1[e] * fminf(t, 0.f);
                        }
                        if (has_res) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) vv[e] += rr[k][e];
                        }
                        if (has_a2) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) vv[e] = fmaxf(vv[e], 0.f) + gc.s2[e] * fminf(vv[e], 0.f);
                        }
                        if (has_sc) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) vv[e] *= p.out_scale;
                        }
                        if (pad16) {
#pragma unroll
                            for (int e = 0; e < 8; ++e)
                                if (e >= n_valid) vv[e] = 0.f;      // pad channels are written as zeros
                        }
                        if (vec && nst == 8) st8<T>(p.y, pix * p.ldy + cout0, p.y_f32, BF, vv);
                        else if (vec) {    // f32 output, one float4 (nst == 4)
                            *(float4*)((float*)p.y + pix * p.ldy + cout0) = make_float4(vv[0], vv[1], vv[2], vv[3]);
                        } else {
#pragma unroll
                            for (int e = 0; e < 8; ++e)
                                if (e < n_valid) st_any<T>(p.y, pix * p.ldy + cout0 + e, p.y_f32, vv[e]);
                        }
                    }
                }
            }
        }
        stamp(3);   // stores issued
    }
#ifndef GVFI_HOSTSIM
    if (prof)
        for (int k = 0; k < 4; ++k) ((unsigned long long*)p.aux1)[(long long)blockIdx.x * 4 + k] = ph[k];
#endif
}

// LDS plan of a problem; returns total bytes (0 = not eligible)
static int patch_plan(const gvfi_conv_params& p, PatchArgs& a) {
    const int ve = p.dtype == GVFI_F32 ? 4 : 8, esz = p.dtype == GVFI_F32 ? 4 : 2;
    if (p.c1 != 0 || p.x1 != nullptr || p.c0 <= 0 || (p.c0 % ve) || p.groups > 1 || p.epi_mode != GVFI_EPI_STD) return 0;
    if (p.w_layout != 0 || p.stats != nullptr || p.Cout > 64 || p.Cout <= 0 || p.dtype == GVFI_F16) return 0;
    if (p.stride != 1 && p.stride != 2) return 0;
    if (p.KH > 7 || p.KW > 7 || p.KH * p.KW < 1) return 0;
    if (p.pad_mode == GVFI_PAD_REFLECT && (p.pad_h >= p.H || p.pad_w >= p.W)) return 0;
    if (p.ld0 % ve) return 0;
    a.p = p;
    a.gpt = p.c0 / ve;
    a.G = p.KH * p.KW * a.gpt;
    a.G2 = ((a.G + 1) / 2 + 3) & ~3;
    const int cbytes = p.c0 * esz;
    a.pitch = ((cbytes / 16) % 2 == 1) ? cbytes : cbytes + 16;      // odd multiple of 16 bytes
    a.tph = (GVFI_PATCH_TH - 1) * p.stride + p.KH;
    a.twp = (GVFI_PATCH_TW - 1) * p.stride + p.KW;
    const int nb = p.Cout > 32 ? 2 : 1;
    // a fragment read may run past the patch by (S-1) pixels of the last row for the unused tail lanes: pad one row
    int patch = (a.tph * a.twp + a.twp) * a.pitch;
    const int staging = GVFI_PATCH_TH * GVFI_PATCH_TW * 32 * 4;
    if (patch < staging) patch = staging;
    patch = (patch + 255) & ~255;
    a.off_w = patch;
    a.off_t = a.off_w + (a.G2 + 3) * nb * 1024;
    const int total = a.off_t + 2 * (a.G2 + 11) * 4;
    gvfi_magic_div((unsigned)a.gpt, a.gpt_mul, a.gpt_sh);
    gvfi_magic_div((unsigned)a.twp, a.twp_mul, a.twp_sh);
    a.tiles_x = cdiv(p.Wo, GVFI_PATCH_TW);
    a.tiles_y = cdiv(p.Ho, GVFI_PATCH_TH);
    a.ntiles = a.tiles_x * a.tiles_y * p.N;
    if (total > 160 * 1024) return 0;
    return total;
}

// 1 = gvfi_conv2d routes this problem to the patch kernel (when the LDS-DMA kernel is not eligible)
extern "C" int gvfi_conv2d_patch_eligible(const gvfi_conv_params* pp) {
    PatchArgs a;
    if (patch_plan(*pp, a) == 0) return 0;
    // (the launcher needs 16-byte aligned bases: an unaligned problem is NOT eligible, so that auto-routing falls
    // through to the generic kernel instead of failing with -3)
    if (((uintptr_t)pp->x0 & 15) || ((uintptr_t)pp->w & 15)) return 0;
    // worth it when the input is re-read per tap by the generic kernel and the output is wide enough for 64-pixel rows
    if (pp->KH * pp->KW < 9 || pp->Wo < 32 || (long long)pp->N * pp->Ho * pp->Wo < 16384) return 0;
    return 1;
}

extern "C" int gvfi_conv2d_patch(const gvfi_conv_params* pp, void* stream) {
    const gvfi_conv_params& p = *pp;
    PatchArgs a;
    const int shm = patch_plan(p, a);
    if (shm == 0) return -2;
    if (((uintptr_t)p.x0 & 15) || ((uintptr_t)p.w & 15)) return -3;
    hipStream_t st = (hipStream_t)stream;
    // persistent workgroups: one per CU and round (every workgroup stages the weights once)
    const int per_cu = shm <= 78 * 1024 ? 2 : 1;        // what the 160 KiB of a CU admit
    int grid = a.ntiles < 256 * per_cu ? a.ntiles : 256 * per_cu;
    if (grid < 1) grid = 1;
#define PATCH_LAUNCH(TT, SS, NN) \
    GVFI_LAUNCH_COOP_SHM((conv_patch_kernel<TT, SS, NN>), dim3(grid), dim3(256), shm, st, a)
#define PATCH_DISPATCH(TT)                                         \
    if (p.stride == 1) {                                           \
        if (p.Cout > 32) { PATCH_LAUNCH(TT, 1, 2); } else { PATCH_LAUNCH(TT, 1, 1); } \
    } else {                                                       \
        if (p.Cout > 32) { PATCH_LAUNCH(TT, 2, 2); } else { PATCH_LAUNCH(TT, 2, 1); } \
    }
    if (p.dtype == GVFI_F32) { PATCH_DISPATCH(float) } else { PATCH_DISPATCH(bf16_t) }
#undef PATCH_DISPATCH
#undef PATCH_LAUNCH
    return (int)hipGetLastError();
}
