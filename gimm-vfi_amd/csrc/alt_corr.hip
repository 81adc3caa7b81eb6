// Volume-free correlation lookup: the MI355X statement of the reference's one C++/CUDA component,
// flowformer/alt_cuda_corr (correlation.cpp:19-54 `forward`, correlation_kernel.cu:18-119 `corr_forward_kernel`,
// :255-286 host side), used by raft/corr.py:96-124 (AlternateCorrBlock) for resolutions where the all-pairs volume
// (P8^2 floats per direction and level) is not wanted.
//
//   corr[b][n][a + rd*c][h][w] = sum over the four integer neighbours of the bilinear sample of
//        s(y, x) = < fmap1[b,h,w,:], fmap2[b,y,x,:] >      (0 outside fmap2)
//   at (y, x) = (coords[b,n,h,w,1] - r + a, coords[b,n,h,w,0] - r + c),   rd = 2r+1, a = dy index (fast), c = dx index
//   -- i.e. the window axes are ordered like CorrBlock's (first axis moves x).  No 1/sqrt(C): the caller divides.
//
// The CUDA original tiles 4x8 queries per block, stages 32-channel slices of both maps through shared memory and
// atomically accumulates the four bilinear contributions of every integer tap into global memory.  Here one WAVE owns
// one query: its 64 lanes stride the channel axis (coalesced 256-byte reads of a fmap2 pixel, fmap1 row in registers),
// the (rd+1)^2 integer-tap dot products are reduced with cross-lane shuffles into LDS, and the rd^2 outputs are
// written once -- no atomics, no zero-initialised output needed, deterministic.
#include "common.h"

#define GVFI_BLOCK 256
static inline dim3 grid1d(long long n) { return dim3((unsigned)((n + GVFI_BLOCK - 1) / GVFI_BLOCK)); }

#ifndef GVFI_HOSTSIM
#define GVFI_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#else
#define GVFI_WAVE_SYNC() emu::wave_sync()
#endif
#define ALT_WAVES 4
#define ALT_MAX_RD1 12       // (2r+1)+1 <= 12  -> r <= 5
#define ALT_MAX_CREG 8       // channels <= 64 * 8

template <typename T>
__global__ void __launch_bounds__(64 * ALT_WAVES) alt_corr_kernel(const T* __restrict__ f1, const T* __restrict__ f2,
                                                                  const float* __restrict__ coords,
                                                                  float* __restrict__ corr, long long nquery, int N,
                                                                  int H1, int W1, int H2, int W2, int C, int r) {
    __shared__ float s_tap[ALT_WAVES][ALT_MAX_RD1 * ALT_MAX_RD1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long q = (long long)blockIdx.x * ALT_WAVES + wave;   // over (b, n, h, w)
    if (q >= nquery) return;   // (no block-wide barrier below: waves are independent)
    const int w = (int)(q % W1);
    const int h = (int)((q / W1) % H1);
    const int n = (int)((q / ((long long)W1 * H1)) % N);
    const int b = (int)(q / ((long long)W1 * H1 * N));
    const int rd = 2 * r + 1, rd1 = rd + 1;
    const float x2 = coords[q * 2 + 0], y2 = coords[q * 2 + 1];
    const float fx = floorf(x2), fy = floorf(y2);
    const float dx = x2 - fx, dy = y2 - fy;
    const int ix0 = (int)fx - r, iy0 = (int)fy - r;
    float a1[ALT_MAX_CREG];
    const T* p1 = f1 + (((long long)b * H1 + h) * W1 + w) * C;
#pragma unroll
    for (int k = 0; k < ALT_MAX_CREG; ++k) a1[k] = (lane + 64 * k < C) ? Elem<T>::ld(p1 + lane + 64 * k) : 0.f;
    for (int t = 0; t < rd1 * rd1; ++t) {
        const int iy = t / rd1, ix = t - iy * rd1;
        const int yy = iy0 + iy, xx = ix0 + ix;
        float s = 0.f;
        if ((unsigned)yy < (unsigned)H2 && (unsigned)xx < (unsigned)W2) {   // wave-uniform
            const T* p2 = f2 + (((long long)b * H2 + yy) * W2 + xx) * C;
#pragma unroll
            for (int k = 0; k < ALT_MAX_CREG; ++k)
                if (lane + 64 * k < C) s += a1[k] * Elem<T>::ld(p2 + lane + 64 * k);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        }
        if (lane == 0) s_tap[wave][iy * ALT_MAX_RD1 + ix] = s;
    }
    GVFI_WAVE_SYNC();   // lane 0's LDS writes are visible to the wave (DS operations of one wave execute in order)
    float* out = corr + (((long long)b * N + n) * rd * rd) * H1 * W1 + (long long)h * W1 + w;
    for (int t = lane; t < rd * rd; t += 64) {
        const int a = t % rd, c = t / rd;   // a: dy index (fast), c: dx index  (correlation_kernel.cu:97-100)
        const float* st = &s_tap[wave][a * ALT_MAX_RD1 + c];
        const float v = (1.f - dy) * (1.f - dx) * st[0] + (1.f - dy) * dx * st[1] + dy * (1.f - dx) * st[ALT_MAX_RD1] +
                        dy * dx * st[ALT_MAX_RD1 + 1];
        out[(long long)t * H1 * W1] = v;
    }
}

// fmap1 [B,H1,W1,C], fmap2 [B,H2,W2,C] (dtype: GVFI_F32 like the reference op, or GVFI_BF16), coords [B,N,H1,W1,2]
// float (x, y) in fmap2 pixels, corr [B,N,(2r+1)^2,H1,W1] float -- every element is written.
extern "C" int gvfi_alt_corr_forward(const void* fmap1, const void* fmap2, const float* coords, float* corr, int B,
                                     int N, int H1, int W1, int H2, int W2, int C, int radius, int dtype,
                                     void* stream) {
    if (radius < 0 || 2 * radius + 2 > ALT_MAX_RD1 || C <= 0 || C > 64 * ALT_MAX_CREG) return -2;
    const long long nquery = (long long)B * N * H1 * W1;
    if (nquery == 0) return 0;
    dim3 grid((unsigned)((nquery + ALT_WAVES - 1) / ALT_WAVES));
    GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_COOP((alt_corr_kernel<T>), grid, dim3(64 * ALT_WAVES), (hipStream_t)stream,
                                            (const T*)fmap1, (const T*)fmap2, coords, corr, nquery, N, H1, W1, H2, W2,
                                            C, radius));
    return (int)hipGetLastError();
}

// 2x2 average pooling of an NHWC map (the fmap2 pyramid of raft/corr.py:101-105; floor semantics of F.avg_pool2d)
template <typename T>
__global__ void avgpool2_nhwc_kernel(const T* __restrict__ src, T* __restrict__ dst, long long total, int H, int W, int C) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int Ho = H / 2, Wo = W / 2;
    const int c = (int)(idx % C);
    const long long pix = idx / C;
    const int x = (int)(pix % Wo);
    const int y = (int)((pix / Wo) % Ho);
    const long long n = pix / ((long long)Wo * Ho);
    const T* p = src + ((n * H + 2 * y) * W + 2 * x) * (long long)C + c;
    const float v = (Elem<T>::ld(p) + Elem<T>::ld(p + C) + Elem<T>::ld(p + (long long)W * C) +
                     Elem<T>::ld(p + (long long)W * C + C)) * 0.25f;
    Elem<T>::st(dst + idx, v);
}
extern "C" int gvfi_avgpool2_nhwc(const void* src, void* dst, int N, int H, int W, int C, int dtype, void* stream) {
    const long long total = (long long)N * (H / 2) * (W / 2) * C;
    if (total == 0) return 0;
    GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_SIMPLE((avgpool2_nhwc_kernel<T>), grid1d(total), dim3(GVFI_BLOCK),
                                              (hipStream_t)stream, (const T*)src, (T*)dst, total, H, W, C));
    return (int)hipGetLastError();
}
