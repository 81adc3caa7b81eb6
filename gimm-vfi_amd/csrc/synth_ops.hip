// Frame-synthesis glue kernels (flow time-scaling, bidirectional lookup coordinates, multi-flow
// warp/blend, decoder head fix-up, output finalisation) and library identity.
#include "common.h"

#define GVFI_BLOCK 256
static inline dim3 grid1d(long long n) { return dim3((unsigned)((n + GVFI_BLOCK - 1) / GVFI_BLOCK)); }

// F(t->0) = -t * flow_t,  F(t->1) = (1-t) * flow_t     gimmvfi_r.py:239-240
__global__ void flow_split_t_kernel(const float* __restrict__ flow_t, const float* __restrict__ t,
                                    float* __restrict__ ft0, float* __restrict__ ft1, long long total, long long HW2) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const float tt = t[idx / HW2];
    const float f = flow_t[idx];
    ft0[idx] = f * (-tt);
    ft1[idx] = f * (1.0f - tt);
}
extern "C" int gvfi_flow_split_t(const float* flow_t, const float* t, float* ft0, float* ft1, int B, int HW,
                                 void* stream) {
    const long long total = 2LL * B * HW;
    GVFI_LAUNCH_SIMPLE(flow_split_t_kernel, grid1d(total), dim3(GVFI_BLOCK), (hipStream_t)stream, flow_t, t, ft0, ft1,
                       total, 2LL * HW);
    return (int)hipGetLastError();
}

// c0 = grid + fl1 * 1/(1-t),  c1 = grid + fl0 * 1/t     gimmvfi_r.py:494-507
__global__ void lookup_coords_kernel(const float* __restrict__ fl0, const float* __restrict__ fl1,
                                     const float* __restrict__ t, float* __restrict__ c0, float* __restrict__ c1,
                                     long long total, int h, int w) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over pixels
    if (idx >= total) return;
    const float tt = t[idx / ((long long)h * w)];
    const float s0 = 1.0f / tt, s1 = 1.0f / (1.0f - tt);
    const float gx = (float)(idx % w), gy = (float)((idx / w) % h);
    c0[idx * 2 + 0] = gx + fl1[idx * 2 + 0] * s1;
    c0[idx * 2 + 1] = gy + fl1[idx * 2 + 1] * s1;
    c1[idx * 2 + 0] = gx + fl0[idx * 2 + 0] * s0;
    c1[idx * 2 + 1] = gy + fl0[idx * 2 + 1] * s0;
}
extern "C" int gvfi_lookup_coords(const float* fl0, const float* fl1, const float* t, float* c0, float* c1, int B,
                                  int h, int w, void* stream) {
    const long long total = (long long)B * h * w;
    GVFI_LAUNCH_SIMPLE(lookup_coords_kernel, grid1d(total), dim3(GVFI_BLOCK), (hipStream_t)stream, fl0, fl1, t, c0, c1,
                       total, h, w);
    return (int)hipGetLastError();
}

// bilinear sample of a 4-float-per-pixel image, border padding, align_corners=True (fi_utils.py:19-49)
__device__ __forceinline__ void sample_border_rgb(const float* __restrict__ img, int H, int W, float fx, float fy,
                                                  float o[3]) {
    fx = fminf(fmaxf(fx, 0.f), (float)(W - 1));
    fy = fminf(fmaxf(fy, 0.f), (float)(H - 1));
    const float x0f = floorf(fx), y0f = floorf(fy);
    const int x0 = (int)x0f, y0 = (int)y0f;
    const float ax = fx - x0f, ay = fy - y0f;
    const bool x1ok = x0 + 1 < W, y1ok = y0 + 1 < H;
    const float* p = img + ((long long)y0 * W + x0) * 4;
    // the four taps as UNCONDITIONAL 16-byte loads (an absent neighbour re-reads the pixel itself): loads inside the `if`s
    // below cannot be hoisted by the compiler, and six samples per pixel then are 24 dependent memory round trips instead
    // of one batch of 24 loads in flight.  The arithmetic (which taps are added, in which order) is unchanged.
    const long long dx = x1ok ? 4 : 0, dy = y1ok ? 4LL * W : 0;
    const float4 t00 = *(const float4*)p, t01 = *(const float4*)(p + dx);
    const float4 t10 = *(const float4*)(p + dy), t11 = *(const float4*)(p + dy + dx);
    float w = (1.f - ax) * (1.f - ay);
    o[0] = w * t00.x; o[1] = w * t00.y; o[2] = w * t00.z;
    if (x1ok) { w = ax * (1.f - ay); o[0] += w * t01.x; o[1] += w * t01.y; o[2] += w * t01.z; }
    if (y1ok) {
        w = (1.f - ax) * ay; o[0] += w * t10.x; o[1] += w * t10.y; o[2] += w * t10.z;
        if (x1ok) { w = ax * ay; o[0] += w * t11.x; o[1] += w * t11.y; o[2] += w * t11.z; }
    }
}

// warp_w_mask + (x+1)/2 + clamp     gimmvfi_r.py:213-220, 259-261   (mask given pre-sigmoid)
__global__ void warp_blend_kernel(const float* __restrict__ i0, const float* __restrict__ i1,
                                  const float* __restrict__ f0, const float* __restrict__ f1,
                                  const float* __restrict__ mask, float* __restrict__ out, long long total, int H,
                                  int W, int src_B) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const long long HW = (long long)H * W;
    const long long pix = idx % HW, b = idx / HW;
    const int x = (int)(pix % W), y = (int)(pix / W);
    float a[3], c[3];
    const long long bs = src_B > 0 ? b % src_B : b;
    sample_border_rgb(i0 + bs * HW * 4, H, W, (float)x + f0[idx * 2], (float)y + f0[idx * 2 + 1], a);
    sample_border_rgb(i1 + bs * HW * 4, H, W, (float)x + f1[idx * 2], (float)y + f1[idx * 2 + 1], c);
    const float m = gvfi_sigmoid(mask[idx]);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float v = (m * a[k] + (1.f - m) * c[k] + 1.0f) / 2.0f;
        out[(b * 3 + k) * HW + pix] = fminf(fmaxf(v, 0.f), 1.f);
    }
}
extern "C" int gvfi_warp_blend(const float* img4_0, const float* img4_1, const float* f0, const float* f1,
                               const float* mask, float* out_nchw, int B, int src_B, int H, int W, void* stream) {
    if (src_B < 0 || src_B > B) return -2;
    const long long total = (long long)B * H * W;
    GVFI_LAUNCH_SIMPLE(warp_blend_kernel, grid1d(total), dim3(GVFI_BLOCK), (hipStream_t)stream, img4_0, img4_1, f0, f1,
                       mask, out_nchw, total, H, W, src_B);
    return (int)hipGetLastError();
}

// multi_flow_combine front half     modules/fi_components.py:57-88
template <typename T>
__global__ void combine_warps_kernel(const float* __restrict__ i0, const float* __restrict__ i1,
                                     const float* __restrict__ dec, int ldd, T* __restrict__ act, int lda, int pad,
                                     float* __restrict__ mean4, long long total, int H, int W) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const long long HW = (long long)H * W;
    const long long pix = idx % HW, b = idx / HW;
    const int x = (int)(pix % W), y = (int)(pix / W);
    const float* d = dec + idx * ldd;
    float mean[3] = {0.f, 0.f, 0.f};
    T* a = act + idx * lda;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float w0[3], w1[3];
        sample_border_rgb(i0 + b * HW * 4, H, W, (float)x + d[2 * k], (float)y + d[2 * k + 1], w0);
        sample_border_rgb(i1 + b * HW * 4, H, W, (float)x + d[6 + 2 * k], (float)y + d[6 + 2 * k + 1], w1);
        const float m = d[12 + k];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = m * w0[c] + (1.f - m) * w1[c] + d[15 + 3 * k + c];
            mean[c] += v;
            Elem<T>::st(a + 3 * k + c, v);
        }
    }
    for (int c = 9; c < pad; ++c) Elem<T>::st(a + c, 0.f);
    float* mo = mean4 + idx * 4;
    mo[0] = mean[0] / 3.0f; mo[1] = mean[1] / 3.0f; mo[2] = mean[2] / 3.0f; mo[3] = 0.f;
}
extern "C" int gvfi_combine_warps(const float* img4_0, const float* img4_1, const float* dec, int ldd, void* act,
                                  int lda, int pad, float* mean4, int B, int H, int W, int dtype, void* stream) {
    const long long total = (long long)B * H * W;
    GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_SIMPLE((combine_warps_kernel<T>), grid1d(total), dim3(GVFI_BLOCK),
                                              (hipStream_t)stream, img4_0, img4_1, dec, ldd, (T*)act, lda, pad, mean4,
                                              total, H, W));
    return (int)hipGetLastError();
}

// The same front half reading the decoder output at the WORKING resolution (DS_SCALE < 1, gimmvfi_r.py:294-303): the
// bilinear up-sampling of the 24 decoder channels (flows additionally x Hf/H), the six warps, the blends, and the
// planar (B,3,2,Hf,Wf) copies of the up-sampled flows that the reference returns as flowt0_pred / flowt1_pred -- one
// pass, nothing materialised at full resolution except what leaves the function.  At 4K the separate passes (two
// resizes writing + combine_warps re-reading an 855 MB tensor, two NHWC -> NCHW transposes re-reading it again) were
// ~4.6 ms of the 19 ms per timestep.  Per-channel arithmetic is that of resize_nhwc_kernel (mul * (ly.w0*(lx.w0*v00 +
// lx.w1*v01) + ly.w1*(lx.w0*v10 + lx.w1*v11))) followed by that of combine_warps_kernel, so the results are bit-equal to
// the separate passes; with H == Hf (rscale = inv = 1) the interpolation weights are exactly (1, 0).
// STAGE (up-sampling by >= 2, Wf a multiple of 64, Hf of 4): a block is a 64 x 4 tile of full-resolution pixels, one row per
// wave.  The <= 34 x 4 decoder pixels its bilinear taps touch are copied to LDS once (the taps become LDS reads), and -- what
// matters: the 24 image taps per pixel bind this kernel, profiles/r4_combine_bound_probe.txt -- the four rows of a tile share
// their source rows in the CU's L1 (5 source rows per sample instead of 4 x 2 when every block was one 256-pixel row segment).
// The values read are the same, so the results are bit-identical to the direct form.
#define CWU_SX 34
#define CWU_SY 4
template <typename T, bool STAGE>
__global__ void __launch_bounds__(256) combine_warps_up_kernel(const float* __restrict__ i0, const float* __restrict__ i1,
                                        const float* __restrict__ dec, int ldd, int H, int W, float rscale, float inv,
                                        T* __restrict__ act, int lda, int pad, float* __restrict__ mean4,
                                        float* __restrict__ f0p, float* __restrict__ f1p, long long total, int Hf, int Wf,
                                        int src_B) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long HWf = (long long)Hf * Wf;
    __shared__ __attribute__((aligned(16))) float sdec[STAGE ? CWU_SY * CWU_SX * 24 : 4];
    int sx0 = 0, sy0 = 0;
    if (STAGE) {
        // block -> tile (b0, rows 4 tyi .. +3, columns 64 txi .. +63); thread -> pixel (row = wave, column = lane)
        const int bx = Wf / 64, by = Hf / 4;
        const long long blk = blockIdx.x;
        const int txi = (int)(blk % bx), tyi = (int)((blk / bx) % by);
        const long long b0 = blk / ((long long)bx * by);
        const int xb = 64 * txi, yb = 4 * tyi;
        idx = (b0 * Hf + yb + (threadIdx.x >> 6)) * (long long)Wf + xb + (threadIdx.x & 63);
        sx0 = src_index(xb, rscale, W).i0;
        sy0 = src_index(yb, rscale, H).i0;
        const int nsx = src_index(xb + 63, rscale, W).i1 - sx0 + 1;        // <= CWU_SX (host: inv >= 2)
        const int nsy = src_index(yb + 3, rscale, H).i1 - sy0 + 1;         // <= CWU_SY
        for (int u = threadIdx.x; u < nsy * nsx * 6; u += 256) {
            const int r = u / (nsx * 6), v = u - r * (nsx * 6);
            const int px = v / 6, q = v - px * 6;
            const float* src = dec + ((b0 * H + sy0 + r) * (long long)W + sx0 + px) * ldd + 4 * q;
            *(float4*)(sdec + (r * CWU_SX + px) * 24 + 4 * q) = *(const float4*)src;
        }
        __syncthreads();
    }
    if (idx >= total) return;
    const long long pix = idx % HWf, b = idx / HWf;
    const int x = (int)(pix % Wf), y = (int)(pix / Wf);
    const Lerp ly = src_index(y, rscale, H), lx = src_index(x, rscale, W);
    const float* p00 = STAGE ? sdec + ((ly.i0 - sy0) * CWU_SX + lx.i0 - sx0) * 24 : dec + ((b * H + ly.i0) * (long long)W + lx.i0) * ldd;
    const float* p01 = STAGE ? sdec + ((ly.i0 - sy0) * CWU_SX + lx.i1 - sx0) * 24 : dec + ((b * H + ly.i0) * (long long)W + lx.i1) * ldd;
    const float* p10 = STAGE ? sdec + ((ly.i1 - sy0) * CWU_SX + lx.i0 - sx0) * 24 : dec + ((b * H + ly.i1) * (long long)W + lx.i0) * ldd;
    const float* p11 = STAGE ? sdec + ((ly.i1 - sy0) * CWU_SX + lx.i1 - sx0) * 24 : dec + ((b * H + ly.i1) * (long long)W + lx.i1) * ldd;
    float d[24];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const float4 a00 = *(const float4*)(p00 + 4 * q), a01 = *(const float4*)(p01 + 4 * q);
        const float4 a10 = *(const float4*)(p10 + 4 * q), a11 = *(const float4*)(p11 + 4 * q);
        const float m = q < 3 ? inv : 1.0f;
        d[4 * q + 0] = m * (ly.w0 * (lx.w0 * a00.x + lx.w1 * a01.x) + ly.w1 * (lx.w0 * a10.x + lx.w1 * a11.x));
        d[4 * q + 1] = m * (ly.w0 * (lx.w0 * a00.y + lx.w1 * a01.y) + ly.w1 * (lx.w0 * a10.y + lx.w1 * a11.y));
        d[4 * q + 2] = m * (ly.w0 * (lx.w0 * a00.z + lx.w1 * a01.z) + ly.w1 * (lx.w0 * a10.z + lx.w1 * a11.z));
        d[4 * q + 3] = m * (ly.w0 * (lx.w0 * a00.w + lx.w1 * a01.w) + ly.w1 * (lx.w0 * a10.w + lx.w1 * a11.w));
    }
    float mean[3] = {0.f, 0.f, 0.f};
    float v9[9];
    const long long bs = src_B > 0 ? b % src_B : b;       // image of the source batch (timestep-batched decoder output)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float w0[3], w1[3];
        sample_border_rgb(i0 + bs * HWf * 4, Hf, Wf, (float)x + d[2 * k], (float)y + d[2 * k + 1], w0);
        sample_border_rgb(i1 + bs * HWf * 4, Hf, Wf, (float)x + d[6 + 2 * k], (float)y + d[6 + 2 * k + 1], w1);
        const float m = d[12 + k];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = m * w0[c] + (1.f - m) * w1[c] + d[15 + 3 * k + c];
            mean[c] += v;
            v9[3 * k + c] = v;
        }
    }
    T* a = act + idx * lda;
    if (sizeof(T) == 2 && pad == 16 && ((((uintptr_t)act) | (uintptr_t)(lda * 2)) & 15) == 0) {   // two 16-byte stores
        uint32_t w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float lo = 2 * i < 9 ? v9[2 * i < 9 ? 2 * i : 0] : 0.f, hi = 2 * i + 1 < 9 ? v9[2 * i + 1 < 9 ? 2 * i + 1 : 0] : 0.f;
            w[i] = (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
        }
        uint4 u0, u1;
        u0.x = w[0]; u0.y = w[1]; u0.z = w[2]; u0.w = w[3];
        u1.x = w[4]; u1.y = w[5]; u1.z = w[6]; u1.w = w[7];
        ((uint4*)a)[0] = u0;
        ((uint4*)a)[1] = u1;
    } else {
#pragma unroll
        for (int c = 0; c < 9; ++c) Elem<T>::st(a + c, v9[c]);
        for (int c = 9; c < pad; ++c) Elem<T>::st(a + c, 0.f);
    }
    *(float4*)(mean4 + idx * 4) = make_float4(mean[0] / 3.0f, mean[1] / 3.0f, mean[2] / 3.0f, 0.f);
    // planar flows: channel k*2 + c of (B, 3, 2, Hf, Wf)
    if (f0p != nullptr) {
#pragma unroll
        for (int c = 0; c < 6; ++c) f0p[(b * 6 + c) * HWf + pix] = d[c];
    }
    if (f1p != nullptr) {
#pragma unroll
        for (int c = 0; c < 6; ++c) f1p[(b * 6 + c) * HWf + pix] = d[6 + c];
    }
}
extern "C" int gvfi_combine_warps_up(const float* img4_0, const float* img4_1, const float* dec, int ldd, int H, int W,
                                     void* act, int lda, int pad, float* mean4, float* flow0_planar, float* flow1_planar,
                                     int B, int src_B, int Hf, int Wf, int dtype, void* stream) {
    if (src_B < 0 || src_B > B) return -2;
    if (H <= 0 || W <= 0 || Hf <= 0 || Wf <= 0 || (ldd & 3) || (((uintptr_t)dec | (uintptr_t)mean4) & 15)) return -2;
    if ((long long)Hf * W != (long long)H * Wf) return -3;       // one isotropic scale
    const float inv = (float)((double)Hf / (double)H);           // torch: scale_factor = 1 / ds_factor, flows x the same
    const float rscale = (float)(1.0 / ((double)Hf / (double)H));
    const long long total = (long long)B * Hf * Wf;
    // staged form: 64 x 4 pixel tiles, up-sampling by 2 or more (<= 34 x 4 decoder pixels per tile), ldd == 24
    const bool stage = GVFI_BLOCK == 256 && (Wf % 64) == 0 && (Hf % 4) == 0 && Hf >= 2 * H && ldd == 24;
    if (stage) {
        GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_COOP((combine_warps_up_kernel<T, true>), grid1d(total), dim3(256),
                                                (hipStream_t)stream, img4_0, img4_1, dec, ldd, H, W, rscale, inv, (T*)act,
                                                lda, pad, mean4, flow0_planar, flow1_planar, total, Hf, Wf, src_B));
    } else {
        GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_SIMPLE((combine_warps_up_kernel<T, false>), grid1d(total), dim3(GVFI_BLOCK),
                                                  (hipStream_t)stream, img4_0, img4_1, dec, ldd, H, W, rscale, inv, (T*)act,
                                                  lda, pad, mean4, flow0_planar, flow1_planar, total, Hf, Wf, src_B));
    }
    return (int)hipGetLastError();
}

// decoder head     modules/fi_components.py:331-340
__global__ void decoder_head_kernel(float* __restrict__ dec, int ldd, const float* __restrict__ flow0,
                                    const float* __restrict__ flow1, const float* __restrict__ mask, long long total) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over (pix, 15)
    if (idx >= total) return;
    const int c = (int)(idx % 15);
    const long long pix = idx / 15;
    float* d = dec + pix * ldd + c;
    if (c < 6) *d += flow0[pix * 2 + (c & 1)];
    else if (c < 12) *d += flow1[pix * 2 + (c & 1)];
    else *d = gvfi_sigmoid(*d + mask[pix]);
}
extern "C" int gvfi_decoder_head(float* dec, int ldd, const float* flow0, const float* flow1, const float* mask,
                                 long long npix, void* stream) {
    const long long total = npix * 15;
    GVFI_LAUNCH_SIMPLE(decoder_head_kernel, grid1d(total), dim3(GVFI_BLOCK), (hipStream_t)stream, dec, ldd, flow0,
                       flow1, mask, total);
    return (int)hipGetLastError();
}

// imgt_pred = clamp((x + 1)/2, 0, 1)     modules/fi_components.py:92, gimmvfi_r.py:308
__global__ void finalize_image_kernel(const float* __restrict__ x, int ld, float* __restrict__ out, long long total,
                                      long long HW) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over (b, c, pix)
    if (idx >= total) return;
    const long long pix = idx % HW;
    const int c = (int)((idx / HW) % 3);
    const long long b = idx / (3 * HW);
    const float v = (x[(b * HW + pix) * ld + c] + 1.0f) / 2.0f;
    out[idx] = fminf(fmaxf(v, 0.f), 1.f);
}
extern "C" int gvfi_finalize_image(const float* x, int ld, float* out_nchw, int B, int H, int W, void* stream) {
    const long long HW = (long long)H * W, total = 3 * HW * B;
    GVFI_LAUNCH_SIMPLE(finalize_image_kernel, grid1d(total), dim3(GVFI_BLOCK), (hipStream_t)stream, x, ld, out_nchw,
                       total, HW);
    return (int)hipGetLastError();
}

// float NCHW [0,1] frame -> uint8 HWC (truncating x*255 like reference src/video_Nx.py:192-196 astype(np.uint8))
__global__ void frames_to_u8_kernel(const float* __restrict__ src, unsigned char* __restrict__ dst, long long total,
                                    long long HW) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over (b, pix, c)
    if (idx >= total) return;
    const int c = (int)(idx % 3);
    const long long bp = idx / 3;
    const long long b = bp / HW, pix = bp % HW;
    float v = src[(b * 3 + c) * HW + pix] * 255.0f;
    v = fminf(fmaxf(v, 0.f), 255.f);
    dst[idx] = (unsigned char)v;
}
extern "C" int gvfi_frames_to_u8(const float* src_nchw, unsigned char* dst_nhwc, int B, int H, int W, void* stream) {
    const long long HW = (long long)H * W, total = 3 * HW * B;
    GVFI_LAUNCH_SIMPLE(frames_to_u8_kernel, grid1d(total), dim3(GVFI_BLOCK), (hipStream_t)stream, src_nchw, dst_nhwc,
                       total, HW);
    return (int)hipGetLastError();
}

// Side-by-side frames of the CLI's output.mp4 (reference src/video_Nx.py:139-151, 198-216: cv2.hconcat([ori_image[-1],
// images[-1]]) per written frame), composed on the device from what is already resident: the padded float input frames of a
// block of consecutive pairs and its interpolated uint8 frames -- one D2H per block, no second PNG decode and no hconcat on
// the host.  Frame f of the block (after an optional leading [orig 0 | orig 0], the video's first frame): pair jj = g / N, slot
// i = g % N; i < N-1: [orig jj | interpolated frame i of pair jj], i = N-1: [orig jj+1 | orig jj+1].  Originals are converted
// like the reference converts them, (float * 255.0f) truncated to uint8; output BGR (cv2 / VideoSink order).
__global__ void compose_sbs_kernel(const float* __restrict__ frames, int Hp, int Wp, int pad_top, int pad_left,
                                   const unsigned char* __restrict__ pred, unsigned char* __restrict__ out, long long total,
                                   int lead, int N, int H0, int W0) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // over (f, y, x of 2 * W0)
    if (idx >= total) return;
    const int W2 = 2 * W0;
    const int x = (int)(idx % W2), y = (int)((idx / W2) % H0);
    const int f = (int)(idx / ((long long)W2 * H0));
    const bool first = lead && f == 0;
    const int g = f - lead;
    const int jj = first ? 0 : g / N, i = first ? N - 1 : g - (g / N) * N;
    const int k = first ? 0 : (i == N - 1 ? jj + 1 : jj);        // the original frame shown in this video frame
    const bool right = x >= W0;
    const int xx = right ? x - W0 : x;
    unsigned char* o = out + idx * 3;
    if (right && !first && i < N - 1) {
        const unsigned char* p = pred + ((((long long)jj * (N - 1) + i) * H0 + y) * W0 + xx) * 3;     // RGB
        o[0] = p[2]; o[1] = p[1]; o[2] = p[0];
        return;
    }
    const long long plane = (long long)Hp * Wp;
    const float* src = frames + (long long)k * 3 * plane + (long long)(y + pad_top) * Wp + (xx + pad_left);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = src[(2 - c) * plane] * 255.0f;          // BGR: channel c of the output = plane 2 - c
        o[c] = (unsigned char)v;                                 // astype(np.uint8) of a value in [0, 255]: truncation
    }
}
extern "C" int gvfi_compose_sbs_u8(const float* frames, int n_frames, int Hp, int Wp, int pad_top, int pad_left,
                                   const unsigned char* pred_u8, int pairs, int N, int lead, unsigned char* out, int H0, int W0,
                                   void* stream) {
    if (pairs <= 0 || n_frames != pairs + 1 || N < 2 || H0 <= 0 || W0 <= 0 || pad_top < 0 || pad_left < 0 ||
        pad_top + H0 > Hp || pad_left + W0 > Wp || (lead != 0 && lead != 1))
        return -2;
    const long long total = ((long long)pairs * N + lead) * H0 * 2 * W0;
    GVFI_LAUNCH_SIMPLE(compose_sbs_kernel, grid1d(total), dim3(GVFI_BLOCK), (hipStream_t)stream, frames, Hp, Wp, pad_top,
                       pad_left, pred_u8, out, total, lead, N, H0, W0);
    return (int)hipGetLastError();
}

// Flow colour coding of the CLI's flow.mp4 side output (reference src/utils/flow_viz.py:20-136 `flow_to_image`, Middlebury
// wheel): per image rad_max = max |flow|, (u, v) / (rad_max + 1e-5), hue from atan2 on the 55-entry wheel, saturation from the
// radius.  numpy evaluates the angle chain in float32 and the colour blend in float64 (float32 - int32 promotes): the same
// types are used here, without FMA contraction, so that a picture differs from numpy's only where atan2f differs by an ulp.
// Host post-processing of the 2K CLI was 250 ms per pair for these pictures; the model needs 77 ms.
__global__ void flow_radmax_kernel(const float* __restrict__ flow, long long img_stride, long long HW, unsigned* __restrict__ radmax) {
#pragma clang fp contract(off)
    const long long img = blockIdx.y;
    const float* u = flow + img * img_stride;
    const float* v = u + HW;
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (long long)gridDim.x * blockDim.x) {
        const float a = u[i], b = v[i];
        const float aa = a * a, bb = b * b;
        m = fmaxf(m, sqrtf(aa + bb));
    }
    // non-negative floats order like their bit patterns
    atomicMax(radmax + img, __builtin_bit_cast(unsigned, m));
}
__global__ void flow_to_image_kernel(const float* __restrict__ flow, long long img_stride, long long HW, const unsigned* __restrict__ radmax,
                                     const float* __restrict__ wheel /*[55][3]*/, unsigned char* __restrict__ out, int bgr) {
#pragma clang fp contract(off)
    const long long img = blockIdx.y;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HW) return;
    const float den = __builtin_bit_cast(float, radmax[img]) + 1e-5f;
    const float u = flow[img * img_stride + i] / den, v = flow[img * img_stride + HW + i] / den;
    const float uu = u * u, vv = v * v;
    const float rad = sqrtf(uu + vv);
    const float pi = 3.14159265358979323846f;
    float fk = atan2f(-v, -u) / pi;
    fk = fk + 1.0f;
    fk = fk / 2.0f;
    fk = fk * 54.0f;
    const int k0 = (int)floorf(fk);
    const int k1 = (k0 + 1) % 55;
    const double f = (double)fk - (double)k0;
    unsigned char* o = out + (img * HW + i) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double col = (1.0 - f) * (double)wheel[k0 * 3 + c] / 255.0 + f * (double)wheel[k1 * 3 + c] / 255.0;
        col = rad <= 1.0f ? 1.0 - (double)rad * (1.0 - col) : col * 0.75;
        o[bgr ? 2 - c : c] = (unsigned char)floor(255.0 * col);
    }
}
extern "C" int gvfi_flow_to_image(const float* flow, long long img_stride, int n_img, int h, int w, const float* wheel,
                                  unsigned* radmax_zeroed, unsigned char* out, int bgr, void* stream) {
    const long long HW = (long long)h * w;
    GVFI_LAUNCH_SIMPLE(flow_radmax_kernel, dim3(64, n_img), dim3(GVFI_BLOCK), (hipStream_t)stream, flow, img_stride, HW, radmax_zeroed);
    GVFI_LAUNCH_SIMPLE(flow_to_image_kernel, dim3((unsigned)((HW + GVFI_BLOCK - 1) / GVFI_BLOCK), n_img), dim3(GVFI_BLOCK), (hipStream_t)stream,
                       flow, img_stride, HW, radmax_zeroed, wheel, out, bgr);
    return (int)hipGetLastError();
}

extern "C" const char* gvfi_version(void) {
#ifdef GVFI_HOSTSIM
    return "gimmvfi-hostsim (test emulator, not a product build)";
#else
    return "gimmvfi-hip gfx950 r1";
#endif
}
extern "C" int gvfi_device_ok(void) {
#ifdef GVFI_HOSTSIM
    return 0;
#else
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 0;
    return strstr(prop.gcnArchName, "gfx950") != nullptr ? 1 : 0;
#endif
}
