// HBM-bound NHWC glue kernels: input preparation, InstanceNorm, bilinear resize, backward warp,
// pixel shuffle, channel copies and layout conversion.  One thread per output element with the
// channel index fastest, so a wave touches consecutive addresses (coalesced) on both sides.
#include "common.h"

#define GVFI_BLOCK 256
static inline dim3 grid1d(long long n) { return dim3((unsigned)((n + GVFI_BLOCK - 1) / GVFI_BLOCK)); }

// torch upsample_bilinear2d source index, align_corners=False (ATen UpSample.h
// area_pixel_compute_source_index + guard_index_and_lambda)
struct Lerp { int i0, i1; float w0, w1; };
__device__ __forceinline__ Lerp src_index(int d, float rscale, int n) {
    float s = rscale * (d + 0.5f) - 0.5f;
    if (s < 0.f) s = 0.f;
    int i0 = (int)s;
    if (i0 > n - 1) i0 = n - 1;
    Lerp r;
    r.i0 = i0;
    r.i1 = i0 + (i0 < n - 1 ? 1 : 0);
    float l1 = s - (float)i0;
    l1 = l1 < 0.f ? 0.f : (l1 > 1.f ? 1.f : l1);
    r.w1 = l1;
    r.w0 = 1.f - l1;
    return r;
}

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

// ------------------------------------------------------------------ InstanceNorm2d
#define IN_CHUNK 1024
template <typename T>
__global__ void instnorm_stats_kernel(const T* __restrict__ x, int ld, int C, int HW, float* __restrict__ stats) {
    // block = one chunk of IN_CHUNK pixels of one image; thread = (pixel lane, channel)
    const int n = blockIdx.y;
    const int lanes = blockDim.x / C;          // pixel lanes (>= 1 because C <= blockDim.x)
    const int c = threadIdx.x % C;
    const int pl = threadIdx.x / C;
    if (pl >= lanes) return;
    const long long p0 = (long long)blockIdx.x * IN_CHUNK;
    long long p1 = p0 + IN_CHUNK;
    if (p1 > HW) p1 = HW;
    float s = 0.f, ss = 0.f;
    for (long long p = p0 + pl; p < p1; p += lanes) {
        const float v = Elem<T>::ld(x + ((long long)n * HW + p) * ld + c);
        s += v;
        ss += v * v;
    }
    atomicAdd(&stats[((long long)n * C + c) * 2 + 0], s);
    atomicAdd(&stats[((long long)n * C + c) * 2 + 1], ss);
}
extern "C" int gvfi_instnorm_stats(const void* x, int ld, int C, int N, int HW, float* stats, int dtype, void* stream) {
    if (C > GVFI_BLOCK) return -2;
    dim3 grid((unsigned)((HW + IN_CHUNK - 1) / IN_CHUNK), (unsigned)N);
    GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_SIMPLE((instnorm_stats_kernel<T>), grid, dim3(GVFI_BLOCK), (hipStream_t)stream,
                                              (const T*)x, ld, C, HW, stats));
    return (int)hipGetLastError();
}
template <typename T>
__global__ void instnorm_apply_kernel(const T* __restrict__ x, int ld, int C, long long total, int HW,
                                      const float* __restrict__ stats, int relu, const T* __restrict__ res, int ldr,
                                      T* __restrict__ out, int ldo) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const long long pix = idx / C;
    const int n = (int)(pix / HW);
    const float inv = 1.0f / (float)HW;
    const float mean = stats[((long long)n * C + c) * 2] * inv;
    float var = stats[((long long)n * C + c) * 2 + 1] * inv - mean * mean;
    if (var < 0.f) var = 0.f;
    float v = (Elem<T>::ld(x + pix * ld + c) - mean) / sqrtf(var + 1e-5f);
    if (relu && v < 0.f) v = 0.f;
    if (res) {
        v += Elem<T>::ld(res + pix * ldr + c);
        if (v < 0.f) v = 0.f;
    }
    Elem<T>::st(out + pix * ldo + c, v);
}
extern "C" int gvfi_instnorm_apply(const void* x, int ld, int C, int N, int HW, const float* stats, int relu,
                                   const void* res, int ldr, void* out, int ldo, int dtype, void* stream) {
    const long long total = (long long)N * HW * C;
    GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_SIMPLE((instnorm_apply_kernel<T>), grid1d(total), dim3(GVFI_BLOCK),
                                              (hipStream_t)stream, (const T*)x, ld, C, total, HW, stats, relu,
                                              (const T*)res, ldr, (T*)out, ldo));
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
extern "C" int gvfi_resize_nhwc(const void* src, int lds, int src_f32, void* dst, int ldd, int dst_f32, int C, int N,
                                int H, int W, int Ho, int Wo, float rscale, float mul, int dtype, void* stream) {
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
                                 long long total, int H, int W) {
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
    const long long b = n * (long long)H * W;
    float v = 0.f;
    v += (1.f - ax) * (1.f - ay) * ld_any<T>(src, (b + (long long)y0 * W + x0) * lds + c, src_f32);
    if (x0 + 1 < W) v += ax * (1.f - ay) * ld_any<T>(src, (b + (long long)y0 * W + x0 + 1) * lds + c, src_f32);
    if (y0 + 1 < H) v += (1.f - ax) * ay * ld_any<T>(src, (b + (long long)(y0 + 1) * W + x0) * lds + c, src_f32);
    if (x0 + 1 < W && y0 + 1 < H)
        v += ax * ay * ld_any<T>(src, (b + (long long)(y0 + 1) * W + x0 + 1) * lds + c, src_f32);
    st_any<T>(dst, pix * ldd + c, dst_f32, v);
}
extern "C" int gvfi_warp_nhwc(const void* src, int lds, int src_f32, const float* flow, int ldf, float fmul, void* dst,
                              int ldd, int dst_f32, int C, int N, int H, int W, int dtype, void* stream) {
    const long long total = (long long)N * H * W * C;
    GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_SIMPLE((warp_nhwc_kernel<T>), grid1d(total), dim3(GVFI_BLOCK),
                                              (hipStream_t)stream, src, lds, src_f32, flow, ldf, fmul, dst, ldd,
                                              dst_f32, C, total, H, W));
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
                                     float mul, long long total) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const long long pix = idx / C;
    float v = mul * ld_any<T>(src, pix * lds + c, src_f32);
    if (add) v += ld_any<T>(add, pix * lda + c, add_f32);
    st_any<T>(dst, pix * ldd + c, dst_f32, v);
}
extern "C" int gvfi_copy_channels(const void* src, int lds, int src_f32, const void* add, int lda, int add_f32,
                                  void* dst, int ldd, int dst_f32, int C, float mul, long long npix, int dtype,
                                  void* stream) {
    const long long total = npix * C;
    GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_SIMPLE((copy_channels_kernel<T>), grid1d(total), dim3(GVFI_BLOCK),
                                              (hipStream_t)stream, src, lds, src_f32, add, lda, add_f32, dst, ldd,
                                              dst_f32, C, mul, total));
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
