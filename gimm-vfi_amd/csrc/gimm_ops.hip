// GIMM (motion INR) side kernels: flow (un)normalisation, splatting metric, softmax-splat forward
// warp (the reference's one native CUDA kernel on the path) and INR input packing.
#include "common.h"

#define GVFI_BLOCK 256
static inline dim3 grid1d(long long n) { return dim3((unsigned)((n + GVFI_BLOCK - 1) / GVFI_BLOCK)); }

// ------------------------------------------------------------------ per-sample abs-max   modules/fi_utils.py:52-57
// non-negative floats order like their bit patterns -> atomicMax on the raw bits
__global__ void flow_absmax_kernel(const float* __restrict__ f01, const float* __restrict__ f10,
                                   unsigned* __restrict__ scaler, long long per_b) {
    const int b = blockIdx.y;
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < per_b; i += (long long)gridDim.x * blockDim.x) {
        m = fmaxf(m, fabsf(f01[b * per_b + i]));
        m = fmaxf(m, fabsf(f10[b * per_b + i]));
    }
    // wave-level max first (64 lanes), then one atomic per wave instead of one per thread
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    union { float f; unsigned u; } c;
    c.f = m;
    if ((threadIdx.x & 63) == 0) atomicMax(scaler + b, c.u);
}
extern "C" int gvfi_flow_absmax(const float* f01, const float* f10, float* scaler, int B, int HW, void* stream) {
    const long long per_b = 2LL * HW;
    dim3 grid((unsigned)(per_b < 64 * GVFI_BLOCK ? (per_b + GVFI_BLOCK - 1) / GVFI_BLOCK : 64), (unsigned)B);
    GVFI_LAUNCH_COOP(flow_absmax_kernel, grid, dim3(GVFI_BLOCK), (hipStream_t)stream, f01, f10, (unsigned*)scaler,
                     per_b);
    return (int)hipGetLastError();
}

// n = (f/s + 1)/2 for [f01, -f10]   modules/fi_utils.py:52-60, gimmvfi_r.py:142-145
template <typename T>
__global__ void flow_normalize_kernel(const float* __restrict__ f01, const float* __restrict__ f10,
                                      const float* __restrict__ scaler, T* __restrict__ act, int ld, int pad,
                                      float* __restrict__ nflow, int B, long long HW) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over (d, b, pix)
    if (idx >= 2 * B * HW) return;
    const long long pix = idx % HW;
    const int n = (int)(idx / HW);
    const int d = n / B, b = n % B;
    const float s = scaler[b];
    float v[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float f = d == 0 ? f01[((long long)b * HW + pix) * 2 + c] : -f10[((long long)b * HW + pix) * 2 + c];
        v[c] = (f / s + 1.0f) / 2.0f;
        if (nflow) nflow[(((long long)b * 2 + c) * 2 + d) * HW + pix] = v[c];
    }
    T* a = act + idx * ld;
    for (int c = 0; c < pad; ++c) Elem<T>::st(a + c, c < 2 ? v[c] : 0.f);
}
extern "C" int gvfi_flow_normalize(const float* f01, const float* f10, const float* scaler, void* act, int ld, int pad,
                                   float* nflow_out, int B, int H, int W, int dtype, void* stream) {
    const long long HW = (long long)H * W, total = 2 * B * HW;
    GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_SIMPLE((flow_normalize_kernel<T>), grid1d(total), dim3(GVFI_BLOCK),
                                              (hipStream_t)stream, f01, f10, scaler, (T*)act, ld, pad, nflow_out, B,
                                              HW));
    return (int)hipGetLastError();
}

// flow_t = (n*2 - 1) * s   modules/fi_utils.py:63-64
__global__ void flow_unnormalize_kernel(const float* __restrict__ ninr, const float* __restrict__ scaler,
                                        float* __restrict__ flow_t, float* __restrict__ ninr_nchw, long long total,
                                        long long HW) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over (b, pix, c)
    if (idx >= total) return;
    const int c = (int)(idx & 1);
    const long long bp = idx >> 1;
    const long long b = bp / HW, pix = bp % HW;
    const float v = ninr[idx];
    flow_t[idx] = (v * 2.0f - 1.0f) * scaler[b];
    if (ninr_nchw) ninr_nchw[(b * 2 + c) * HW + pix] = v;
}
extern "C" int gvfi_flow_unnormalize(const float* ninr, const float* scaler, float* flow_t, float* ninr_nchw, int B,
                                     int HW, void* stream) {
    const long long total = 2LL * B * HW;
    GVFI_LAUNCH_SIMPLE(flow_unnormalize_kernel, grid1d(total), dim3(GVFI_BLOCK), (hipStream_t)stream, ninr, scaler,
                       flow_t, ninr_nchw, total, (long long)HW);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------ splatting metric   gimmvfi_r.py:444-492
// Z = 1/(1 + err*alpha_fe) + 1/(1 + std*alpha_v);  std = mean_c sqrt(clamp(G*f^2 - (G*f)^2, 1e-9)) with a
// reflect-padded 3x3 filter;  err = mean_c | -warp(f_rev, f) - f |  (border warp, align_corners=True).
__device__ __forceinline__ void sample_border2(const float* __restrict__ img, int H, int W, float fx, float fy,
                                               float& o0, float& o1) {
    fx = fminf(fmaxf(fx, 0.f), (float)(W - 1));
    fy = fminf(fmaxf(fy, 0.f), (float)(H - 1));
    const float x0f = floorf(fx), y0f = floorf(fy);
    const int x0 = (int)x0f, y0 = (int)y0f;
    const float ax = fx - x0f, ay = fy - y0f;
    // the four taps as UNCONDITIONAL loads: a neighbour beyond the border is read at the clamped index and enters with the
    // weight the clamped coordinate gives it -- exactly 0 (ax == 0 on the last column, ay == 0 on the last row), so the sum
    // is the border sample.  (Round 6: with the taps behind `if (x1ok)` / `if (y1ok)` the compiler emitted a load / wait /
    // packed-fp32 sequence per tap, and this kernel returned different values in lanes 48..63 of a few waves whenever an
    // LDS-DMA kernel was resident on the same CUs; see profiles/r6_concurrency_repro.txt.  The library is built without
    // packed fp32 instructions since -- build.py -- and this form stays because it is also the faster one: four loads in flight.)
    const int x1 = x0 + 1 < W ? x0 + 1 : x0, y1 = y0 + 1 < H ? y0 + 1 : y0;
    const float2 p00 = *(const float2*)(img + ((long long)y0 * W + x0) * 2), p10 = *(const float2*)(img + ((long long)y0 * W + x1) * 2);
    const float2 p01 = *(const float2*)(img + ((long long)y1 * W + x0) * 2), p11 = *(const float2*)(img + ((long long)y1 * W + x1) * 2);
    float w = (1.f - ax) * (1.f - ay);
    o0 = w * p00.x;
    o1 = w * p00.y;
    w = ax * (1.f - ay); o0 += w * p10.x; o1 += w * p10.y;
    w = (1.f - ax) * ay; o0 += w * p01.x; o1 += w * p01.y;
    w = ax * ay; o0 += w * p11.x; o1 += w * p11.y;
}
__global__ void splat_weights_kernel(const float* __restrict__ f01, const float* __restrict__ f10,
                                     const float* __restrict__ g9, float alpha_v, float alpha_fe,
                                     float* __restrict__ z0, float* __restrict__ z1, int B, int H, int W) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over (d, b, y, x)
    const long long HW = (long long)H * W;
    if (idx >= 2 * B * HW) return;
    const long long pix = idx % HW;
    const int x = (int)(pix % W), y = (int)(pix / W);
    const int n = (int)(idx / HW);
    const int d = n / B, b = n % B;
    const float* f = (d == 0 ? f01 : f10) + (long long)b * HW * 2;
    const float* fr = (d == 0 ? f10 : f01) + (long long)b * HW * 2;
    float sq[2] = {0.f, 0.f}, mn[2] = {0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int yy = reflect_idx(y + ky - 1, H), xx = reflect_idx(x + kx - 1, W);
            const float g = g9[ky * 3 + kx];
            const float a = f[((long long)yy * W + xx) * 2 + 0], c = f[((long long)yy * W + xx) * 2 + 1];
            sq[0] += g * (a * a);
            sq[1] += g * (c * c);
            mn[0] += g * a;
            mn[1] += g * c;
        }
    const float v0 = sqrtf(fmaxf(sq[0] - mn[0] * mn[0], 1e-9f));
    const float v1 = sqrtf(fmaxf(sq[1] - mn[1] * mn[1], 1e-9f));
    const float var = (v0 + v1) / 2.0f;
    const float fx = f[pix * 2 + 0], fy = f[pix * 2 + 1];
    float w0, w1;
    sample_border2(fr, H, W, (float)x + fx, (float)y + fy, w0, w1);
    const float err = (fabsf(-w0 - fx) + fabsf(-w1 - fy)) / 2.0f;
    const float z = 1.0f / (1.0f + err * alpha_fe) + 1.0f / (1.0f + var * alpha_v);
    (d == 0 ? z0 : z1)[(long long)b * HW + pix] = z;
}
extern "C" int gvfi_splat_weights(const float* f01, const float* f10, const float* gfilt9, float alpha_v,
                                  float alpha_fe, float* z0, float* z1, int B, int H, int W, void* stream) {
    if (H < 2 || W < 2) return -2;
    const long long total = 2LL * B * H * W;
    GVFI_LAUNCH_SIMPLE(splat_weights_kernel, grid1d(total), dim3(GVFI_BLOCK), (hipStream_t)stream, f01, f10, gfilt9,
                       alpha_v, alpha_fe, z0, z1, B, H, W);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------ softmax-splat forward   modules/softsplat.py:371-421
// One thread per (source pixel, channel) of [lat*Z, Z] (channel fastest: the C+1 atomics of one source
// pixel land in consecutive floats of the NHWC accumulator).  Targets outside the image are dropped,
// non-finite targets are skipped (softsplat.py:384-385, 406-420).
template <typename T>
__global__ void softsplat_accum_kernel(const T* __restrict__ lat, int ldl, int C, const float* __restrict__ flow,
                                       const float* __restrict__ z, const float* __restrict__ t, int one_minus_t,
                                       float* __restrict__ acc, long long total, int H, int W) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int C1 = C + 1;
    const int c = (int)(idx % C1);
    const long long pix = idx / C1;
    const int x = (int)(pix % W);
    const int y = (int)((pix / W) % H);
    const long long b = pix / ((long long)W * H);
    const float ts = one_minus_t ? (1.0f - t[b]) : t[b];
    const float fx = (float)x + flow[pix * 2 + 0] * ts;
    const float fy = (float)y + flow[pix * 2 + 1] * ts;
    if (!isfinite(fx) || !isfinite(fy)) return;
    const float zz = z[pix];
    const float v = c < C ? Elem<T>::ld(lat + pix * ldl + c) * zz : zz;
    const float x0f = floorf(fx), y0f = floorf(fy);
    const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    const float wnw = ((float)x1 - fx) * ((float)y1 - fy);
    const float wne = (fx - (float)x0) * ((float)y1 - fy);
    const float wsw = ((float)x1 - fx) * (fy - (float)y0);
    const float wse = (fx - (float)x0) * (fy - (float)y0);
    float* o = acc + b * (long long)H * W * C1 + c;
    const bool x0in = x0 >= 0 && x0 < W, x1in = x1 >= 0 && x1 < W;
    const bool y0in = y0 >= 0 && y0 < H, y1in = y1 >= 0 && y1 < H;
    if (x0in && y0in) atomicAdd(o + ((long long)y0 * W + x0) * C1, v * wnw);
    if (x1in && y0in) atomicAdd(o + ((long long)y0 * W + x1) * C1, v * wne);
    if (x0in && y1in) atomicAdd(o + ((long long)y1 * W + x0) * C1, v * wsw);
    if (x1in && y1in) atomicAdd(o + ((long long)y1 * W + x1) * C1, v * wse);
}
extern "C" int gvfi_softsplat_accum(const void* lat, int ldl, int C, const float* flow, const float* z, const float* t,
                                    int one_minus_t, float* acc, int B, int H, int W, int dtype, void* stream) {
    const long long total = (long long)B * H * W * (C + 1);
    GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_SIMPLE((softsplat_accum_kernel<T>), grid1d(total), dim3(GVFI_BLOCK),
                                              (hipStream_t)stream, (const T*)lat, ldl, C, flow, z, t, one_minus_t, acc,
                                              total, H, W));
    return (int)hipGetLastError();
}
// "linear-zeroeps": out = splat(x*Z) / splat(Z) with exact-zero denominators replaced by 1  softsplat.py:325-344
template <typename T>
__global__ void softsplat_normalize_kernel(const float* __restrict__ acc, int C, T* __restrict__ dst, int ldd,
                                           long long total) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const long long pix = idx / C;
    float nrm = acc[pix * (C + 1) + C];
    if (nrm == 0.0f) nrm = 1.0f;
    Elem<T>::st(dst + pix * ldd + c, acc[pix * (C + 1) + c] / nrm);
}
extern "C" int gvfi_softsplat_normalize(const float* acc, int C, void* dst, int ldd, long long npix, int dtype,
                                        void* stream) {
    const long long total = npix * C;
    GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_SIMPLE((softsplat_normalize_kernel<T>), grid1d(total), dim3(GVFI_BLOCK),
                                              (hipStream_t)stream, acc, C, (T*)dst, ldd, total));
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------ the same splat as a deterministic gather
// The scatter above issues (C+1) x 4 device-scope float atomics per source pixel (62 M for 8 x 256 x 448: 1.8 TB/s of fabric
// traffic for 132 MB of algorithmic bytes, and a sum whose rounding depends on arrival order).  Here a source pixel is
// entered ONCE into the list of the cell (floor(fx), floor(fy)) of its target (one integer exchange), and every TARGET
// pixel then walks the lists of the four cells whose sources reach it -- (tx-1..tx) x (ty-1..ty) -- and adds their
// contributions in ascending source index: one writer per output, no float atomics, a result that does not depend on
// scheduling, and the "linear-zeroeps" normalisation (softsplat.py:325-344) applied in the same pass.  Every product is
// the reference's expression (v = lat * Z; v * w with the four corner weights of softsplat.py:394-404); only the ORDER of
// the additions is fixed where the reference leaves it to the hardware.
// Cells: x0 in [-1, W-1], y0 in [-1, H-1] -> (H+1) x (W+1) list heads per image and direction, -1 = empty (set by the
// caller); next[] has one entry per source pixel.  Both directions run in one launch: image n = d * B + b with d = 0:
// flow 0->1 scaled by t, d = 1: flow 1->0 scaled by 1 - t (gimmvfi_r.py:171-186).
__global__ void softsplat_lists_kernel(const float* __restrict__ f01, const float* __restrict__ f10, const float* __restrict__ t,
                                       int* __restrict__ head, int* __restrict__ next, long long total, int B, int H, int W) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // over (d, b, y, x)
    if (idx >= total) return;
    const long long HW = (long long)H * W;
    const long long n = idx / HW, pix = idx - n * HW;
    const int d = (int)(n / B), b = (int)(n - (long long)d * B);
    const int x = (int)(pix % W), y = (int)(pix / W);
    const float* fl = (d ? f10 : f01) + ((long long)b * HW + pix) * 2;
    const float ts = d ? (1.0f - t[b]) : t[b];
    const float fx = (float)x + fl[0] * ts, fy = (float)y + fl[1] * ts;
    if (!isfinite(fx) || !isfinite(fy)) return;
    const float x0f = floorf(fx), y0f = floorf(fy);
    if (!(x0f >= -1.0f && x0f <= (float)(W - 1) && y0f >= -1.0f && y0f <= (float)(H - 1))) return;   // no corner inside the image
    const long long cell = (n * (H + 1) + ((int)y0f + 1)) * (long long)(W + 1) + ((int)x0f + 1);
    next[idx] = atomicExch(head + cell, (int)pix);
}
template <typename T>
__global__ void softsplat_gather_kernel(const T* __restrict__ lat, int ldl, const float* __restrict__ f01,
                                        const float* __restrict__ f10, const float* __restrict__ z0,
                                        const float* __restrict__ z1, const float* __restrict__ t,
                                        const int* __restrict__ head, const int* __restrict__ next, T* __restrict__ dst, int ldd,
                                        long long total, int B, int H, int W) {
    constexpr int C = 16, CAP = 8, NONE = 0x7fffffff;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // over (d, b, ty, tx)
    if (idx >= total) return;
    const long long HW = (long long)H * W;
    const long long n = idx / HW, pix = idx - n * HW;
    const int d = (int)(n / B), b = (int)(n - (long long)d * B);
    const int tx = (int)(pix % W), ty = (int)(pix / W);
    const float* fl = (d ? f10 : f01) + (long long)b * HW * 2;
    const float* zz = (d ? z1 : z0) + (long long)b * HW;
    const T* la = lat + (long long)b * HW * ldl + 16 * d;
    const float ts = d ? (1.0f - t[b]) : t[b];
    const int* nx = next + n * HW;
    // the four cells (cx, cy) = (tx - 1 + i, ty - 1 + j) in list coordinates (+1): always inside the (H+1) x (W+1) grid
    const int* hd = head + (n * (H + 1) + ty) * (long long)(W + 1) + tx;
    float acc[C + 1];
#pragma unroll
    for (int c = 0; c <= C; ++c) acc[c] = 0.f;
    constexpr int NV = C * (int)sizeof(T) / 16;          // 16-byte vectors of a source's latent
    struct Src { float fx, fy, zv; uint4 q[NV]; };
    auto fetch = [&](int s, Src& r) {                     // loads only: several sources are requested before any is used
        r.fx = fl[(long long)s * 2];
        r.fy = fl[(long long)s * 2 + 1];
        r.zv = zz[s];
#pragma unroll
        for (int k = 0; k < NV; ++k) r.q[k] = ((const uint4*)(la + (long long)s * ldl))[k];
    };
    auto accumulate = [&](int s, const Src& r) {
        const int sx = s % W, sy = s / W;
        const float fx = (float)sx + r.fx * ts, fy = (float)sy + r.fy * ts;
        const float x0f = floorf(fx), y0f = floorf(fy);
        const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
        // the corner of the source's 2 x 2 footprint this target is: the reference's four weight expressions
        const float wx = tx == x0 ? ((float)x1 - fx) : (fx - (float)x0);
        const float wy = ty == y0 ? ((float)y1 - fy) : (fy - (float)y0);
        const float w = wx * wy;
        const T* e = (const T*)r.q;
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] += (Elem<T>::ld(e + c) * r.zv) * w;
        acc[C] += r.zv * w;
    };
    // sources of the four lists in ascending index: up to CAP of them through a sorted register array ...
    int a[CAP];
#pragma unroll
    for (int k = 0; k < CAP; ++k) a[k] = NONE;
    int cnt = 0;
    int hs[4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) hs[2 * j + i] = hd[(long long)j * (W + 1) + i];      // the four heads in flight together
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
        int s = hs[c4];
        while (s >= 0) {
            ++cnt;
            int e = s;
#pragma unroll
            for (int k = 0; k < CAP; ++k) {
                const int lo = e < a[k] ? e : a[k];
                e = e < a[k] ? a[k] : e;
                a[k] = lo;
            }
            s = nx[s];
        }
    }
    if (cnt <= CAP) {
        // groups of four sources: their flows, weights and latents are requested together, then added in ascending index
        for (int k0 = 0; k0 < cnt; k0 += 4) {
            Src r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) fetch(a[u] != NONE ? a[u] : a[0], r[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (k0 + u < cnt) accumulate(a[u], r[u]);
#pragma unroll
            for (int q = 0; q + 4 < CAP; ++q) a[q] = a[q + 4];
#pragma unroll
            for (int q = CAP - 4; q < CAP; ++q) a[q] = NONE;
        }
    } else {
        // ... longer lists (many sources converging on one cell): every walk over the four lists selects the next CAP larger
        // indices (the same insertion network, restricted to indices above the last one consumed), so a target with cnt
        // sources costs cnt / CAP walks -- O(cnt^2 / CAP) dependent loads instead of one walk per source (a
        // degenerate flow field that collapses thousands of sources into one cell was a latency cliff).  Same ascending order
        // of additions as the short path: bit-identical sums.
        int last = -1;
        for (int done = 0; done < cnt; done += CAP) {
#pragma unroll
            for (int k = 0; k < CAP; ++k) a[k] = NONE;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                int s = hs[c4];
                while (s >= 0) {
                    if (s > last) {
                        int e = s;
#pragma unroll
                        for (int k = 0; k < CAP; ++k) {
                            const int lo = e < a[k] ? e : a[k];
                            e = e < a[k] ? a[k] : e;
                            a[k] = lo;
                        }
                    }
                    s = nx[s];
                }
            }
#pragma unroll
            for (int u0 = 0; u0 < CAP; u0 += 4) {
                if (done + u0 >= cnt) break;
                Src r[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) fetch(a[u0 + u] != NONE ? a[u0 + u] : a[0], r[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (done + u0 + u < cnt) {
                        accumulate(a[u0 + u], r[u]);
                        last = a[u0 + u];
                    }
            }
        }
    }
    float nrm = acc[C];
    if (nrm == 0.0f) nrm = 1.0f;
    __attribute__((aligned(16))) T o[C];
#pragma unroll
    for (int c = 0; c < C; ++c) Elem<T>::st(o + c, acc[c] / nrm);
    T* dp = dst + ((long long)b * HW + pix) * ldd + 16 * d;
    if (sizeof(T) == 2) {
        *(uint4*)dp = *(const uint4*)o;
        *(uint4*)(dp + 8) = *(const uint4*)(o + 8);
    } else {
#pragma unroll
        for (int c = 0; c < C; ++c) dp[c] = o[c];
    }
}
extern "C" int gvfi_softsplat_lists(const float* f01, const float* f10, const float* t, int* head, int* next, int B, int H,
                                    int W, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0 || (long long)H * W > 0x7fffffffll) return -2;
    const long long total = 2LL * B * H * W;
    GVFI_LAUNCH_SIMPLE(softsplat_lists_kernel, grid1d(total), dim3(GVFI_BLOCK), (hipStream_t)stream, f01, f10, t, head, next,
                       total, B, H, W);
    return (int)hipGetLastError();
}
extern "C" int gvfi_softsplat_gather(const void* lat, int ldl, const float* f01, const float* f10, const float* z0,
                                     const float* z1, const float* t, const int* head, const int* next, void* dst, int ldd,
                                     int B, int H, int W, int dtype, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0) return -2;
    // lat: [B,H,W,ldl] with the direction-d latent in channels [16 d, 16 d + 16); dst likewise (16-byte aligned rows)
    if (dtype != GVFI_F32 && ((((uintptr_t)lat | (uintptr_t)dst) & 15) || (ldl & 7) || (ldd & 7))) return -2;
    if (dtype == GVFI_F32 && ((((uintptr_t)lat) & 15) || (ldl & 3))) return -2;
    const long long total = 2LL * B * H * W;
    GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_SIMPLE((softsplat_gather_kernel<T>), grid1d(total), dim3(GVFI_BLOCK),
                                              (hipStream_t)stream, (const T*)lat, ldl, f01, f10, z0, z1, t, head, next,
                                              (T*)dst, ldd, total, B, H, W));
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------ the reference's native op, same contract
// softsplat_func.forward / kernel `softsplat_out` (modules/softsplat.py:358-446): tenIn (N,C,H,W) f32, tenFlow (N,2,H,W)
// f32, tenOut (N,C,H,W) f32 ZERO-INITIALISED BY THE CALLER, accumulated with float atomics.  One thread per (n, y, x);
// the channel loop keeps the lanes of a wave on consecutive x of one plane (coalesced atomics).
__global__ void softsplat_out_nchw_kernel(const float* __restrict__ tenIn, const float* __restrict__ tenFlow,
                                          float* __restrict__ tenOut, long long total, int C, int H, int W) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const long long HW = (long long)H * W;
    const int x = (int)(idx % W), y = (int)((idx / W) % H);
    const long long n = idx / HW, pix = idx % HW;
    const float fx = (float)x + tenFlow[(n * 2 + 0) * HW + pix];
    const float fy = (float)y + tenFlow[(n * 2 + 1) * HW + pix];
    if (!isfinite(fx) || !isfinite(fy)) return;
    const float x0f = floorf(fx), y0f = floorf(fy);
    if (!(x0f >= -1.0f && x0f < (float)W && y0f >= -1.0f && y0f < (float)H)) return;   // no corner inside the image
    const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    const float wnw = ((float)x1 - fx) * ((float)y1 - fy);
    const float wne = (fx - (float)x0) * ((float)y1 - fy);
    const float wsw = ((float)x1 - fx) * (fy - (float)y0);
    const float wse = (fx - (float)x0) * (fy - (float)y0);
    const bool x0in = x0 >= 0 && x0 < W, x1in = x1 >= 0 && x1 < W;
    const bool y0in = y0 >= 0 && y0 < H, y1in = y1 >= 0 && y1 < H;
    for (int c = 0; c < C; ++c) {
        const float v = tenIn[(n * C + c) * HW + pix];
        float* o = tenOut + (n * C + c) * HW;
        if (x0in && y0in) atomicAdd(o + (long long)y0 * W + x0, v * wnw);
        if (x1in && y0in) atomicAdd(o + (long long)y0 * W + x1, v * wne);
        if (x0in && y1in) atomicAdd(o + (long long)y1 * W + x0, v * wsw);
        if (x1in && y1in) atomicAdd(o + (long long)y1 * W + x1, v * wse);
    }
}
extern "C" int gvfi_softsplat_out_nchw(const float* tenIn, const float* tenFlow, float* tenOut, int N, int C, int H,
                                       int W, void* stream) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return -2;
    const long long total = (long long)N * H * W;
    GVFI_LAUNCH_SIMPLE(softsplat_out_nchw_kernel, grid1d(total), dim3(GVFI_BLOCK), (hipStream_t)stream, tenIn, tenFlow,
                       tenOut, total, C, H, W);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------ INR input   modules/hyponet.py:91-95
template <typename T>
__global__ void inr_pack_kernel(const T* __restrict__ lat, int ldl, int C, const float* __restrict__ coord,
                                T* __restrict__ dst, int ldd, int pad, long long total) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % pad);
    const long long pix = idx / pad;
    float v = 0.f;
    if (c < C) v = Elem<T>::ld(lat + pix * ldl + c);
    else if (c < C + 3) v = coord[pix * 3 + (c - C)];
    Elem<T>::st(dst + pix * ldd + c, v);
}
extern "C" int gvfi_inr_pack(const void* lat, int ldl, int C, const float* coord, void* dst, int ldd, int pad,
                             long long npix, int dtype, void* stream) {
    const long long total = npix * pad;
    GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_SIMPLE((inr_pack_kernel<T>), grid1d(total), dim3(GVFI_BLOCK),
                                              (hipStream_t)stream, (const T*)lat, ldl, C, coord, (T*)dst, ldd, pad,
                                              total));
    return (int)hipGetLastError();
}
