// HBM-bound NHWC glue kernels: input preparation, InstanceNorm, bilinear resize, backward warp,
// pixel shuffle, channel copies and layout conversion.  One thread per output element with the
// channel index fastest, so a wave touches consecutive addresses (coalesced) on both sides.
#include "common.h"

#define GVFI_BLOCK 256
static inline dim3 grid1d(long long n) { return dim3((unsigned)((n + GVFI_BLOCK - 1) / GVFI_BLOCK)); }

// ------------------------------------------------------------------ resize of float planes
__global__ void resize_planes_kernel(const float* __restrict__ src, float* __restrict__ dst, long long total, int H,
                                     int W, int Ho, int Wo, float rscale) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int ox = (int)(idx % Wo);
    const int oy = (int)((idx / Wo) % Ho);
    const long long pl = idx / ((long long)Wo * Ho);
    const Lerp ly = src_index(oy, rscale, H), lx = src_index(ox, rscale, W);
    const float* s = src + pl * (long long)H * W;
    dst[idx] = ly.w0 * (lx.w0 * s[ly.i0 * W + lx.i0] + lx.w1 * s[ly.i0 * W + lx.i1]) +
               ly.w1 * (lx.w0 * s[ly.i1 * W + lx.i0] + lx.w1 * s[ly.i1 * W + lx.i1]);
}
extern "C" int gvfi_resize_planes_f32(const float* src, float* dst, int planes, int H, int W, int Ho, int Wo,
                                      float rscale, void* stream) {
    const long long total = (long long)planes * Ho * Wo;
    GVFI_LAUNCH_SIMPLE(resize_planes_kernel, grid1d(total), dim3(GVFI_BLOCK), (hipStream_t)stream, src, dst, total, H,
                       W, Ho, Wo, rscale);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------ image preparation
template <typename T>
__global__ void prep_images_kernel(const float* __restrict__ img, T* __restrict__ act, float* __restrict__ img4, int B,
                                   int H, int W) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long HW = (long long)H * W;
    if (idx >= 2 * B * HW) return;
    const long long pix = idx % HW;
    const int n = (int)(idx / HW);
    const int f = n / B, b = n % B;
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = 2.f * img[(((long long)b * 3 + c) * 2 + f) * HW + pix] - 1.0f;
    T* a = act + idx * 8;
#pragma unroll
    for (int c = 0; c < 8; ++c) Elem<T>::st(a + c, c < 3 ? v[c] : 0.f);
    float* o = img4 + idx * 4;
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = 0.f;
}
extern "C" int gvfi_prep_images(const float* img_xs, void* act, float* img4, int B, int H, int W, int dtype,
                                void* stream) {
    const long long total = 2LL * B * H * W;
    GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_SIMPLE((prep_images_kernel<T>), grid1d(total), dim3(GVFI_BLOCK),
                                              (hipStream_t)stream, img_xs, (T*)act, img4, B, H, W));
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------ InstanceNorm2d   raft/extractor.py:17-25,168-220
// (nn.InstanceNorm2d defaults: biased variance over H*W per (n, c), eps = 1e-5, no affine)
// Both passes are pure HBM streams: 16-byte vectors (8 bf16 / 4 f32 channels of one pixel per lane), the
// statistics pass reduces the pixel lanes of a workgroup through LDS and issues one (fixed-point, order-independent: gvfi_stats_add)
// atomic pair per channel.
#define IN_CHUNK 2048
template <typename T> struct alignas(16) Vec16 { T e[Elem<T>::VE]; };

template <typename T>
__global__ void instnorm_stats_kernel(const T* __restrict__ x, int ld, int C, int HW, float* __restrict__ stats) {
    constexpr int VE = Elem<T>::VE;
    __shared__ float red[2][GVFI_BLOCK * VE];
    const int n = blockIdx.y;
    const int G = C / VE;                      // channel groups per pixel
    const int lanes = blockDim.x / G;          // pixel lanes (>= 1 because C <= blockDim.x)
    const int cg = threadIdx.x % G;
    const int pl = threadIdx.x / G;
    const long long p0 = (long long)blockIdx.x * IN_CHUNK;
    long long p1 = p0 + IN_CHUNK;
    if (p1 > HW) p1 = HW;
    float s[VE], ss[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) s[e] = ss[e] = 0.f;
    if (pl < lanes) {
        for (long long p = p0 + pl; p < p1; p += lanes) {
            const Vec16<T> v = *(const Vec16<T>*)(x + ((long long)n * HW + p) * ld + cg * VE);
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                const float f = Elem<T>::ld(&v.e[e]);
                s[e] += f;
                ss[e] += f * f;
            }
        }
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            red[0][pl * C + cg * VE + e] = s[e];
            red[1][pl * C + cg * VE + e] = ss[e];
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < C) {
        float a0 = 0.f, a1 = 0.f;
        for (int l = 0; l < lanes; ++l) {
            a0 += red[0][l * C + threadIdx.x];
            a1 += red[1][l * C + threadIdx.x];
        }
        gvfi_stats_add(stats, (long long)n * C + threadIdx.x, a0, a1);
    }
}
extern "C" int gvfi_instnorm_stats(const void* x, int ld, int C, int N, int HW, float* stats, int dtype, void* stream) {
    const int ve = dtype == GVFI_F32 ? 4 : 8;
    if (C > GVFI_BLOCK || (C % ve) || (ld % ve) || ((uintptr_t)x & 15)) return -2;
    dim3 grid((unsigned)((HW + IN_CHUNK - 1) / IN_CHUNK), (unsigned)N);
    GVFI_EMU_SERIAL(true);
    GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_COOP((instnorm_stats_kernel<T>), grid, dim3(GVFI_BLOCK), (hipStream_t)stream,
                                            (const T*)x, ld, C, HW, stats));
    return (int)hipGetLastError();
}
template <typename T>
__global__ void instnorm_apply_kernel(const T* __restrict__ x, int ld, int C, long long total, int HW,
                                      const float* __restrict__ stats, int relu, const T* __restrict__ res, int ldr,
                                      T* __restrict__ out, int ldo) {
    constexpr int VE = Elem<T>::VE;
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over (pixel, channel group)
    if (idx >= total) return;
    const int G = C / VE;
    const int c0 = (int)(idx % G) * VE;
    const long long pix = idx / G;
    const int n = (int)(pix / HW);
    const float inv = 1.0f / (float)HW;
    const Vec16<T> v = *(const Vec16<T>*)(x + pix * ld + c0);
    Vec16<T> r, o;
    if (res) r = *(const Vec16<T>*)(res + pix * ldr + c0);
#pragma unroll
    for (int e = 0; e < VE; ++e) {
        float s0, s1;
        gvfi_stats_get(stats, (long long)n * C + c0 + e, s0, s1);
        const float mean = s0 * inv;
        float var = s1 * inv - mean * mean;
        if (var < 0.f) var = 0.f;
        float f = (Elem<T>::ld(&v.e[e]) - mean) / sqrtf(var + 1e-5f);
        if (relu && f < 0.f) f = 0.f;
        if (res) {
            f += Elem<T>::ld(&r.e[e]);
            if (f < 0.f) f = 0.f;
        }
        Elem<T>::st(&o.e[e], f);
    }
    *(Vec16<T>*)(out + pix * ldo + c0) = o;
}
extern "C" int gvfi_instnorm_apply(const void* x, int ld, int C, int N, int HW, const float* stats, int relu,
                                   const void* res, int ldr, void* out, int ldo, int dtype, void* stream) {
    const int ve = dtype == GVFI_F32 ? 4 : 8;
    if ((C % ve) || (ld % ve) || (ldo % ve) || (res && (ldr % ve)) || ((uintptr_t)x & 15) || ((uintptr_t)out & 15) ||
        ((uintptr_t)res & 15))
        return -2;
    const long long total = (long long)N * HW * (C / ve);
    GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_SIMPLE((instnorm_apply_kernel<T>), grid1d(total), dim3(GVFI_BLOCK),
                                              (hipStream_t)stream, (const T*)x, ld, C, total, HW, stats, relu,
                                              (const T*)res, ldr, (T*)out, ldo));
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------ im2col for tiny channel counts
// A 7x7 convolution over 2..4 channels (raft/update.py:100,107; modules/fi_components.py:177) has K = 98..196 real
// products per output; run as a padded-channel implicit GEMM it spends 4x that.  The patch matrix of such a layer
// is small (K <= 256 per pixel at 1/8 or 1/4 resolution), so it is materialised once, K ordered (kh, kw, c) and
// zero-padded to a whole K chunk, and the layer becomes a 1x1 convolution on the LDS-DMA kernel.
template <typename T>
__global__ void im2col_kernel(const T* __restrict__ x, int ld, int c, int H, int W, int KH, int KW, int pad_h,
                              int pad_w, int Ho, int Wo, T* __restrict__ out, int ldo, long long total) {
    constexpr int VE = Elem<T>::VE;
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over (pixel, group of VE patch entries)
    if (idx >= total) return;
    const int G = ldo / VE;
    const int g = (int)(idx % G);
    const long long pix = idx / G;
    const int ox = (int)(pix % Wo);
    const int oy = (int)((pix / Wo) % Ho);
    const long long n = pix / ((long long)Wo * Ho);
    const int K = KH * KW * c;
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < VE; ++e) {
        const int k = g * VE + e;
        float v = 0.f;
        if (k < K) {
            const int tap = k / c, ch = k - tap * c;
            const int kh = tap / KW, kw = tap - kh * KW;
            const int iy = oy - pad_h + kh, ix = ox - pad_w + kw;
            if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                v = Elem<T>::ld(x + ((n * H + iy) * W + ix) * ld + ch);
        }
        Elem<T>::st(&o.e[e], v);
    }
    *(Vec16<T>*)(out + pix * ldo + g * VE) = o;
}
extern "C" int gvfi_im2col(const void* x, int ld, int c, int N, int H, int W, int KH, int KW, int pad_h, int pad_w,
                           void* out, int ldo, int dtype, void* stream) {
    const int ve = dtype == GVFI_F32 ? 4 : 8;
    const int Ho = H + 2 * pad_h - KH + 1, Wo = W + 2 * pad_w - KW + 1;
    if (c <= 0 || Ho <= 0 || Wo <= 0 || (ldo % ve) || ldo < KH * KW * c || ((uintptr_t)out & 15)) return -2;
    const long long total = (long long)N * Ho * Wo * (ldo / ve);
    GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_SIMPLE((im2col_kernel<T>), grid1d(total), dim3(GVFI_BLOCK), (hipStream_t)stream,
                                              (const T*)x, ld, c, H, W, KH, KW, pad_h, pad_w, Ho, Wo, (T*)out, ldo,
                                              total));
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------ bilinear resize, NHWC
template <typename T>
__global__ void resize_nhwc_kernel(const void* __restrict__ src, int lds, int src_f32, void* __restrict__ dst, int ldd,
                                   int dst_f32, int C, long long total, int H, int W, int Ho, int Wo, float rscale,
                                   float mul) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    long long pix = idx / C;
    const int ox = (int)(pix % Wo);
    const int oy = (int)((pix / Wo) % Ho);
    const long long n = pix / ((long long)Wo * Ho);
    const Lerp ly = src_index(oy, rscale, H), lx = src_index(ox, rscale, W);
    const long long b = n * (long long)H * W;
    const float v00 = ld_any<T>(src, (b + (long long)ly.i0 * W + lx.i0) * lds + c, src_f32);
    const float v01 = ld_any<T>(src, (b + (long long)ly.i0 * W + lx.i1) * lds + c, src_f32);
    const float v10 = ld_any<T>(src, (b + (long long)ly.i1 * W + lx.i0) * lds + c, src_f32);
    const float v11 = ld_any<T>(src, (b + (long long)ly.i1 * W + lx.i1) * lds + c, src_f32);
    const float v = ly.w0 * (lx.w0 * v00 + lx.w1 * v01) + ly.w1 * (lx.w0 * v10 + lx.w1 * v11);
    st_any<T>(dst, pix * ldd + c, dst_f32, mul * v);
}
// same arithmetic, one 16-byte channel vector (8 bf16 / 4 f32) per thread: used whenever both tensors have the
// runtime element type and vector-aligned pitches (every feature-map resize of the path)
template <typename T>
__global__ void resize_nhwc_vec_kernel(const T* __restrict__ src, int lds, T* __restrict__ dst, int ldd, int C,
                                       long long total, int H, int W, int Ho, int Wo, float rscale, float mul) {
    constexpr int VE = Elem<T>::VE;
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int G = C / VE;
    const int c = (int)(idx % G) * VE;
    long long pix = idx / G;
    const int ox = (int)(pix % Wo);
    const int oy = (int)((pix / Wo) % Ho);
    const long long n = pix / ((long long)Wo * Ho);
    const Lerp ly = src_index(oy, rscale, H), lx = src_index(ox, rscale, W);
    const long long b = n * (long long)H * W;
    const Vec16<T> a00 = *(const Vec16<T>*)(src + (b + (long long)ly.i0 * W + lx.i0) * lds + c);
    const Vec16<T> a01 = *(const Vec16<T>*)(src + (b + (long long)ly.i0 * W + lx.i1) * lds + c);
    const Vec16<T> a10 = *(const Vec16<T>*)(src + (b + (long long)ly.i1 * W + lx.i0) * lds + c);
    const Vec16<T> a11 = *(const Vec16<T>*)(src + (b + (long long)ly.i1 * W + lx.i1) * lds + c);
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < VE; ++e) {
        const float v00 = Elem<T>::ld(&a00.e[e]), v01 = Elem<T>::ld(&a01.e[e]);
        const float v10 = Elem<T>::ld(&a10.e[e]), v11 = Elem<T>::ld(&a11.e[e]);
        const float v = ly.w0 * (lx.w0 * v00 + lx.w1 * v01) + ly.w1 * (lx.w0 * v10 + lx.w1 * v11);
        Elem<T>::st(&o.e[e], mul * v);
    }
    *(Vec16<T>*)(dst + pix * ldd + c) = o;
}
template <typename T> static bool vec_tensors_ok(const void* a, int lda, int a_f32, const void* b, int ldb, int b_f32, int C) {
    constexpr int VE = Elem<T>::VE;
    const bool a_t = sizeof(T) == 4 ? true : !a_f32, b_t = sizeof(T) == 4 ? true : !b_f32;   // tensor holds T elements
    return a_t && b_t && C % VE == 0 && lda % VE == 0 && ldb % VE == 0 && (((uintptr_t)a | (uintptr_t)b) & 15) == 0;
}
extern "C" int gvfi_resize_nhwc(const void* src, int lds, int src_f32, void* dst, int ldd, int dst_f32, int C, int N,
                                int H, int W, int Ho, int Wo, float rscale, float mul, int dtype, void* stream) {
    {
        bool vec = false;
        GVFI_DISPATCH_T(dtype, vec = (vec_tensors_ok<T>(src, lds, src_f32, dst, ldd, dst_f32, C)));
        if (vec) {
            const long long totv = (long long)N * Ho * Wo * (C / (dtype == GVFI_F32 ? 4 : 8));
            GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_SIMPLE((resize_nhwc_vec_kernel<T>), grid1d(totv), dim3(GVFI_BLOCK),
                                                      (hipStream_t)stream, (const T*)src, lds, (T*)dst, ldd, C, totv, H,
                                                      W, Ho, Wo, rscale, mul));
            return (int)hipGetLastError();
        }
    }
    const long long total = (long long)N * Ho * Wo * C;
    GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_SIMPLE((resize_nhwc_kernel<T>), grid1d(total), dim3(GVFI_BLOCK),
                                              (hipStream_t)stream, src, lds, src_f32, dst, ldd, dst_f32, C, total, H, W,
                                              Ho, Wo, rscale, mul));
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------ backward warp (border, align_corners=True)
// grid_sample semantics of modules/fi_utils.py:19-49 when input and flow have the same size:
// sample position = pixel + flow, clipped to [0, size-1]; taps outside the image are skipped.
template <typename T>
__global__ void warp_nhwc_kernel(const void* __restrict__ src, int lds, int src_f32, const float* __restrict__ flow,
                                 int ldf, float fmul, void* __restrict__ dst, int ldd, int dst_f32, int C,
                                 long long total, int H, int W, int src_N) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const long long pix = idx / C;
    const int x = (int)(pix % W);
    const int y = (int)((pix / W) % H);
    const long long n = pix / ((long long)W * H);
    float fx = (float)x + fmul * flow[pix * ldf + 0];
    float fy = (float)y + fmul * flow[pix * ldf + 1];
    fx = fminf(fmaxf(fx, 0.f), (float)(W - 1));
    fy = fminf(fmaxf(fy, 0.f), (float)(H - 1));
    const float x0f = floorf(fx), y0f = floorf(fy);
    const int x0 = (int)x0f, y0 = (int)y0f;
    const float ax = fx - x0f, ay = fy - y0f;
    const long long b = (src_N > 0 ? n % src_N : n) * (long long)H * W;
    // (the four taps as unconditional loads -- an absent neighbour re-reads the pixel itself -- so that they are in flight
    // together; which taps are added, and in which order, is unchanged)
    const bool xin = x0 + 1 < W, yin = y0 + 1 < H;
    const long long p00 = b + (long long)y0 * W + x0, ox = xin ? 1 : 0, oy = yin ? W : 0;
    const float t00 = ld_any<T>(src, p00 * lds + c, src_f32), t01 = ld_any<T>(src, (p00 + ox) * lds + c, src_f32);
    const float t10 = ld_any<T>(src, (p00 + oy) * lds + c, src_f32), t11 = ld_any<T>(src, (p00 + oy + ox) * lds + c, src_f32);
    float v = 0.f;
    v += (1.f - ax) * (1.f - ay) * t00;
    if (xin) v += ax * (1.f - ay) * t01;
    if (yin) v += (1.f - ax) * ay * t10;
    if (xin && yin) v += ax * ay * t11;
    st_any<T>(dst, pix * ldd + c, dst_f32, v);
}
template <typename T>
__global__ void warp_nhwc_vec_kernel(const T* __restrict__ src, int lds, const float* __restrict__ flow, int ldf,
                                     float fmul, T* __restrict__ dst, int ldd, int C, long long total, int H, int W,
                                     int src_N) {
    constexpr int VE = Elem<T>::VE;
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int G = C / VE;
    const int c = (int)(idx % G) * VE;
    const long long pix = idx / G;
    const int x = (int)(pix % W);
    const int y = (int)((pix / W) % H);
    const long long n = pix / ((long long)W * H);
    float fx = (float)x + fmul * flow[pix * ldf + 0];
    float fy = (float)y + fmul * flow[pix * ldf + 1];
    fx = fminf(fmaxf(fx, 0.f), (float)(W - 1));
    fy = fminf(fmaxf(fy, 0.f), (float)(H - 1));
    const float x0f = floorf(fx), y0f = floorf(fy);
    const int x0 = (int)x0f, y0 = (int)y0f;
    const float ax = fx - x0f, ay = fy - y0f;
    const long long b = (src_N > 0 ? n % src_N : n) * (long long)H * W;
    const bool xin = x0 + 1 < W, yin = y0 + 1 < H;
    const T* p00 = src + (b + (long long)y0 * W + x0) * lds + c;
    // four unconditional vector loads in flight together (an absent neighbour re-reads the pixel itself; its weight is never used)
    const long long ox = xin ? lds : 0, oy = yin ? (long long)W * lds : 0;
    const Vec16<T> a00 = *(const Vec16<T>*)p00;
    const Vec16<T> a01 = *(const Vec16<T>*)(p00 + ox);
    const Vec16<T> a10 = *(const Vec16<T>*)(p00 + oy);
    const Vec16<T> a11 = *(const Vec16<T>*)(p00 + oy + ox);
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < VE; ++e) {
        float v = 0.f;
        v += (1.f - ax) * (1.f - ay) * Elem<T>::ld(&a00.e[e]);
        if (xin) v += ax * (1.f - ay) * Elem<T>::ld(&a01.e[e]);
        if (yin) v += (1.f - ax) * ay * Elem<T>::ld(&a10.e[e]);
        if (xin && yin) v += ax * ay * Elem<T>::ld(&a11.e[e]);
        Elem<T>::st(&o.e[e], v);
    }
    *(Vec16<T>*)(dst + pix * ldd + c) = o;
}
extern "C" int gvfi_warp_nhwc(const void* src, int lds, int src_f32, const float* flow, int ldf, float fmul, void* dst,
                              int ldd, int dst_f32, int C, int N, int src_N, int H, int W, int dtype, void* stream) {
    if (src_N < 0 || src_N > N) return -2;
    {
        bool vec = false;
        GVFI_DISPATCH_T(dtype, vec = (vec_tensors_ok<T>(src, lds, src_f32, dst, ldd, dst_f32, C)));
        if (vec) {
            const long long totv = (long long)N * H * W * (C / (dtype == GVFI_F32 ? 4 : 8));
            GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_SIMPLE((warp_nhwc_vec_kernel<T>), grid1d(totv), dim3(GVFI_BLOCK),
                                                      (hipStream_t)stream, (const T*)src, lds, flow, ldf, fmul, (T*)dst,
                                                      ldd, C, totv, H, W, src_N));
            return (int)hipGetLastError();
        }
    }
    const long long total = (long long)N * H * W * C;
    GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_SIMPLE((warp_nhwc_kernel<T>), grid1d(total), dim3(GVFI_BLOCK),
                                              (hipStream_t)stream, src, lds, src_f32, flow, ldf, fmul, dst, ldd,
                                              dst_f32, C, total, H, W, src_N));
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------ PixelShuffle(2)
template <typename T>
__global__ void pixel_shuffle2_kernel(const T* __restrict__ src, int lds, T* __restrict__ dst, int ldd, int Cout,
                                      long long total, int H, int W) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % Cout);
    const long long pix = idx / Cout;  // output pixel
    const int W2 = 2 * W, H2 = 2 * H;
    const int ox = (int)(pix % W2);
    const int oy = (int)((pix / W2) % H2);
    const long long n = pix / ((long long)W2 * H2);
    const long long ip = (n * H + (oy >> 1)) * (long long)W + (ox >> 1);
    dst[pix * ldd + c] = src[ip * lds + c * 4 + (oy & 1) * 2 + (ox & 1)];
}
extern "C" int gvfi_pixel_shuffle2(const void* src, int lds, void* dst, int ldd, int Cout, int N, int H, int W,
                                   int dtype, void* stream) {
    const long long total = (long long)N * H * W * 4 * Cout;
    GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_SIMPLE((pixel_shuffle2_kernel<T>), grid1d(total), dim3(GVFI_BLOCK),
                                              (hipStream_t)stream, (const T*)src, lds, (T*)dst, ldd, Cout, total, H, W));
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------ channel copy / axpy with dtype conversion
template <typename T>
__global__ void copy_channels_kernel(const void* __restrict__ src, int lds, int src_f32, const void* __restrict__ add,
                                     int lda, int add_f32, void* __restrict__ dst, int ldd, int dst_f32, int C,
                                     float mul, long long total, long long src_npix) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const long long pix = idx / C;
    float v = mul * ld_any<T>(src, (src_npix > 0 ? pix % src_npix : pix) * lds + c, src_f32);
    if (add) v += ld_any<T>(add, pix * lda + c, add_f32);
    st_any<T>(dst, pix * ldd + c, dst_f32, v);
}
extern "C" int gvfi_copy_channels(const void* src, int lds, int src_f32, const void* add, int lda, int add_f32,
                                  void* dst, int ldd, int dst_f32, int C, float mul, long long npix, long long src_npix,
                                  int dtype, void* stream) {
    if (src_npix < 0 || src_npix > npix) return -2;
    const long long total = npix * C;
    GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_SIMPLE((copy_channels_kernel<T>), grid1d(total), dim3(GVFI_BLOCK),
                                              (hipStream_t)stream, src, lds, src_f32, add, lda, add_f32, dst, ldd,
                                              dst_f32, C, mul, total, src_npix));
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------ NHWC float -> NCHW float
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ src, int ld, float* __restrict__ dst, int C,
                                    long long total, long long HW) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over (n, c, pix), pix fastest
    if (idx >= total) return;
    const long long pix = idx % HW;
    const int c = (int)((idx / HW) % C);
    const long long n = idx / (HW * C);
    dst[idx] = src[(n * HW + pix) * ld + c];
}
extern "C" int gvfi_nhwc_to_nchw_f32(const float* src, int ld, float* dst, int C, int N, int H, int W, void* stream) {
    const long long HW = (long long)H * W, total = HW * C * N;
    GVFI_LAUNCH_SIMPLE(nhwc_to_nchw_kernel, grid1d(total), dim3(GVFI_BLOCK), (hipStream_t)stream, src, ld, dst, C,
                       total, HW);
    return (int)hipGetLastError();
}
