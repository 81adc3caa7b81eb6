// FlowFormer glue kernels of the GIMM-VFI-F flow estimator (reference src/models/generalizable_INR/flowformer/core/
// FlowFormer/LatentCostFormer/*): everything that is not a dense contraction.  The linear layers, patch / sub-sampling
// convolutions, the cost volume, QK^T and attention*V of the global motion aggregation run on gvfi_conv2d (MFMA).
//
// Token tensors are row matrices [rows][ld] in the activation type (a row = one token, channel-contiguous like an
// NHWC pixel); all reductions (LayerNorm statistics, soft-max, dot products) are evaluated in float.
//
// First correct version: one thread per output row / (query, head); the heavy parts of the path are the
// contractions, these kernels stream a few hundred bytes per token.
#include "common.h"
#include <stdlib.h>

#define GVFI_BLOCK 256
static inline dim3 grid1d(long long n) { return dim3((unsigned)((n + GVFI_BLOCK - 1) / GVFI_BLOCK)); }

// ------------------------------------------------------------------ nn.LayerNorm over the channel axis
// twins.py:1169 (eps 1e-6 in the Twins blocks), torch default 1e-5 elsewhere (twins.py:1146, encoder.py:65,...)
// Input: the residual stream (float when x_f32, else the activation type); output: activation type (the operand
// of the next contraction).  A row is shared by C/VE lanes (one 16-byte output vector each, 64/(C/VE) rows per
// wave); mean and variance are two shuffle reductions over registers.
// Every lane group normalises R rows; the R loads are issued back to back before the first reduction (one row per
// lane group left ~1 KB in flight per wave: 0.5 TB/s on the 229 k-row launches of the cost encoder).
template <typename T, int R>
__global__ void layernorm_vec_kernel(const void* __restrict__ x, int ldx, int x_f32, const float* __restrict__ gamma,
                                     const float* __restrict__ beta, float eps, T* __restrict__ y, int ldy,
                                     long long rows, int C, int lpr) {
    constexpr int VE = Elem<T>::VE;
    const int lane = threadIdx.x & 63;
    const int sub = lane % lpr;
    const int rpw = 64 / lpr;                                   // rows per wave and pass
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long r0 = wave * (rpw * R) + lane / lpr;
    float v[R][VE];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const long long r = r0 + (long long)i * rpw;
        if (r < rows) {
            if (x_f32) {
                const float* xr = (const float*)x + r * ldx + sub * VE;
#pragma unroll
                for (int e = 0; e < VE; e += 4) {
                    const float4 f = *(const float4*)(xr + e);
                    v[i][e] = f.x; v[i][e + 1] = f.y; v[i][e + 2] = f.z; v[i][e + 3] = f.w;
                }
            } else {
                struct alignas(16) Vec { T e[VE]; };
                const Vec q = *(const Vec*)((const T*)x + r * ldx + sub * VE);
#pragma unroll
                for (int e = 0; e < VE; ++e) v[i][e] = Elem<T>::ld(&q.e[e]);
            }
        } else {
#pragma unroll
            for (int e = 0; e < VE; ++e) v[i][e] = 0.f;
        }
    }
    float g[VE], bt[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) {
        g[e] = gamma[sub * VE + e];
        bt[e] = beta[sub * VE + e];
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const long long r = r0 + (long long)i * rpw;
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < VE; ++e) s += v[i][e];
        for (int off = lpr >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off);
        const float mean = s / (float)C;
        float q2 = 0.f;
#pragma unroll
        for (int e = 0; e < VE; ++e) q2 += (v[i][e] - mean) * (v[i][e] - mean);
        for (int off = lpr >> 1; off > 0; off >>= 1) q2 += __shfl_xor(q2, off);
        if (r >= rows) continue;     // (after the shuffles: every lane of the wave takes part in them)
        const float rstd = 1.0f / sqrtf(q2 / (float)C + eps);
        struct alignas(16) VecO { T e[VE]; } o;
#pragma unroll
        for (int e = 0; e < VE; ++e) Elem<T>::st(&o.e[e], (v[i][e] - mean) * rstd * g[e] + bt[e]);
        *(VecO*)(y + r * ldy + sub * VE) = o;
    }
}
template <typename T>
__global__ void layernorm_kernel(const void* __restrict__ x, int ldx, int x_f32, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, float eps, T* __restrict__ y, int ldy, long long rows,
                                 int C) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += ld_any<T>(x, r * ldx + c, x_f32);
    const float mean = s / (float)C;
    float v = 0.f;
    for (int c = 0; c < C; ++c) {
        const float d = ld_any<T>(x, r * ldx + c, x_f32) - mean;
        v += d * d;
    }
    const float rstd = 1.0f / sqrtf(v / (float)C + eps);
    T* yr = y + r * ldy;
    for (int c = 0; c < C; ++c) Elem<T>::st(yr + c, (ld_any<T>(x, r * ldx + c, x_f32) - mean) * rstd * gamma[c] + beta[c]);
}
extern "C" int gvfi_layernorm(const void* x, int ldx, int x_f32, const float* gamma, const float* beta, float eps, void* y,
                              int ldy, long long rows, int C, int dtype, void* stream) {
    if (C <= 0 || ldx < C || ldy < C) return -2;
    const int ve = dtype == GVFI_F32 ? 4 : 8;
    const int lpr = C / ve;
    const int xbytes = (x_f32 || dtype == GVFI_F32) ? 4 : 2;
    const bool vec = (C % ve) == 0 && lpr >= 1 && lpr <= 64 && (lpr & (lpr - 1)) == 0 && (ldy % ve) == 0 &&
                     (((uintptr_t)y) & 15) == 0 && (((uintptr_t)x) & 15) == 0 && ((ldx * xbytes) & 15) == 0;
#ifdef GVFI_HOSTSIM
    // the emulator runs cooperative launches lane by lane: keep the CPU suite fast, the vector kernel
    // is exercised by the unit tests at small row counts
    const bool vec_ok_rows = rows <= 256;
    const long long r4_min = 65;       // ... and let those unit tests reach the four-row variant too
#else
    const bool vec_ok_rows = true;
    const long long r4_min = 4096;
#endif
    if (vec && vec_ok_rows) {
        if (rows >= r4_min) {      // four rows per lane group
            const long long waves = (rows + 4 * (64 / lpr) - 1) / (4 * (64 / lpr));
            GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_COOP((layernorm_vec_kernel<T, 4>), grid1d(waves * 64), dim3(GVFI_BLOCK),
                                                    (hipStream_t)stream, x, ldx, x_f32, gamma, beta, eps, (T*)y, ldy, rows,
                                                    C, lpr));
        } else {
            const long long waves = (rows + (64 / lpr) - 1) / (64 / lpr);
            GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_COOP((layernorm_vec_kernel<T, 1>), grid1d(waves * 64), dim3(GVFI_BLOCK),
                                                    (hipStream_t)stream, x, ldx, x_f32, gamma, beta, eps, (T*)y, ldy, rows,
                                                    C, lpr));
        }
    } else {
        GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_SIMPLE((layernorm_kernel<T>), grid1d(rows), dim3(GVFI_BLOCK),
                                                  (hipStream_t)stream, x, ldx, x_f32, gamma, beta, eps, (T*)y, ldy, rows, C));
    }
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------ PEG: y = x + depthwise3x3(x) + bias   twins.py:1100-1119
// io_f32: x and y are the float residual stream, else the activation type
template <typename T>
__global__ void dwconv3x3_res_kernel(const void* __restrict__ x, int ldx, const float* __restrict__ w /*[9][C]*/,
                                     const float* __restrict__ bias, void* __restrict__ y, int ldy, int io_f32,
                                     long long total, int H, int W, int C) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over (pixel, channel)
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const long long pix = idx / C;
    const int px = (int)(pix % W), py = (int)((pix / W) % H);
    float acc = 0.f;
    for (int ky = 0; ky < 3; ++ky) {
        const int yy = py + ky - 1;
        if ((unsigned)yy >= (unsigned)H) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int xx = px + kx - 1;
            if ((unsigned)xx >= (unsigned)W) continue;
            acc += w[(ky * 3 + kx) * C + c] * ld_any<T>(x, (pix + (long long)(ky - 1) * W + (kx - 1)) * ldx + c, io_f32);
        }
    }
    st_any<T>(y, pix * ldy + c, io_f32, acc + bias[c] + ld_any<T>(x, pix * ldx + c, io_f32));
}
extern "C" int gvfi_dwconv3x3_res(const void* x, int ldx, const float* w, const float* bias, void* y, int ldy, int io_f32,
                                  int N, int H, int W, int C, int dtype, void* stream) {
    const long long total = (long long)N * H * W * C;
    GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_SIMPLE((dwconv3x3_res_kernel<T>), grid1d(total), dim3(GVFI_BLOCK),
                                              (hipStream_t)stream, x, ldx, w, bias, y, ldy, io_f32, total, H, W, C));
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------ LinearPositionEmbeddingSine   attention.py:170-182
// enc(x, y)[k] for dim channels: [sin(3.14 x f/200) | cos(3.14 x f/200) | sin(3.14 y f/200) | cos(..)], f = 0..dim/4-1
__device__ __forceinline__ float pos_enc_channel(float px, float py, int c, int dim) {
    const int q = dim >> 2;
    const int part = c / q;
    const float f = (float)(c - part * q);
    const float a = 3.14f * (part < 2 ? px : py) * f * (1.0f / 200.0f);
    return (part & 1) ? cosf(a) : sinf(a);
}
// out[row, 0:dim] (+)= enc(scale * coords[row % period] + offset)
template <typename T>
__global__ void pos_embed_kernel(const float* __restrict__ coords, long long period, float scale, float offset, int dim,
                                 T* __restrict__ out, int ldo, long long total, int accumulate) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over (row, channel)
    if (idx >= total) return;
    const int c = (int)(idx % dim);
    const long long r = idx / dim;
    const long long cr = r % period;
    const float e = pos_enc_channel(coords[cr * 2] * scale + offset, coords[cr * 2 + 1] * scale + offset, c, dim);
    T* o = out + r * ldo + c;
    Elem<T>::st(o, accumulate ? Elem<T>::ld(o) + e : e);
}
extern "C" int gvfi_pos_embed(const float* coords, long long period, float scale, float offset, int dim, void* out, int ldo,
                              long long rows, int accumulate, int dtype, void* stream) {
    if (dim <= 0 || (dim & 3) || period <= 0) return -2;
    const long long total = rows * dim;
    GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_SIMPLE((pos_embed_kernel<T>), grid1d(total), dim3(GVFI_BLOCK), (hipStream_t)stream,
                                              coords, period, scale, offset, dim, (T*)out, ldo, total, accumulate));
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------ first cost-map convolution   encoder.py:39-41
// Conv2d(1, 16, 6, stride 2, padding 2) + ReLU over the cost maps [maps][H][W] (float, the all-pairs volume itself),
// zero-extended to a multiple of the patch size on the right / bottom (encoder.py:70-75): out [maps][Ho][Wo][16].
// (A space-to-depth output layout that turns the two following 6x6 stride-2 convolutions into 3x3 stride-1 ones on
// the LDS-DMA kernel was measured in round 2: 130.6 frames/s either way, removed.)
template <typename T>
__global__ void cost_embed1_kernel(const float* __restrict__ vol, const float* __restrict__ w /*[36][16]*/,
                                   const float* __restrict__ bias, T* __restrict__ out, int ldo, long long total, int H,
                                   int W, int Ho, int Wo) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over output pixels
    if (idx >= total) return;
    const int ox = (int)(idx % Wo), oy = (int)((idx / Wo) % Ho);
    const long long m = idx / ((long long)Wo * Ho);
    const float* src = vol + m * (long long)H * W;
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = bias[c];
    // (measured in round 4: the 36 taps as unconditional, fully unrolled loads ran 4.3x SLOWER -- 3.09 instead of 0.72 ms: the
    // taps of neighbouring output pixels overlap, so the loads are L1 hits either way, and the unrolled form spills)
    for (int ky = 0; ky < 6; ++ky) {
        const int yy = oy * 2 - 2 + ky;
        if ((unsigned)yy >= (unsigned)H) continue;
        for (int kx = 0; kx < 6; ++kx) {
            const int xx = ox * 2 - 2 + kx;
            if ((unsigned)xx >= (unsigned)W) continue;
            const float v = src[(long long)yy * W + xx];
            const float* wk = w + (ky * 6 + kx) * 16;
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[c] += v * wk[c];
        }
    }
    T* o = out + idx * ldo;
#pragma unroll
    for (int c = 0; c < 16; ++c) Elem<T>::st(o + c, acc[c] > 0.f ? acc[c] : 0.f);
}
// The same convolution with ONE WORKGROUP PER COST MAP (round 4): the map (H x W floats, 7 KB at 448x256, 35 KB at the 2K / 4K
// working grid) is copied to LDS once with coalesced 16-byte loads -- the volume crosses the memory system exactly once -- and the
// 36 taps of every output pixel are LDS reads.  The thread-per-pixel form above re-reads every cost value nine times through the
// L1 path in 36 dependent round trips per thread (0.72 ms for the 205 MB volume of 8 pairs at 448x256: 0.29 TB/s).  Same
// accumulation order (bias, then ky, kx ascending): bit-identical results.
template <typename T>
__global__ void __launch_bounds__(256) cost_embed1_lds_kernel(const float* __restrict__ vol, const float* __restrict__ w,
                                                              const float* __restrict__ bias, T* __restrict__ out, int ldo,
                                                              int H, int W, int Ho, int Wo) {
    GVFI_DYN_SMEM(smem);
    float* mp = (float*)smem;                       // [H][W]
    const long long m = blockIdx.x;
    const float* src = vol + m * (long long)H * W;
    const int n = H * W;
    if (((((uintptr_t)src) & 15) == 0) && (n & 3) == 0) {
        for (int i = threadIdx.x * 4; i < n; i += 256 * 4) *(float4*)(mp + i) = *(const float4*)(src + i);
    } else {
        for (int i = threadIdx.x; i < n; i += 256) mp[i] = src[i];
    }
    __syncthreads();
    for (int o = threadIdx.x; o < Ho * Wo; o += 256) {
        const int ox = o % Wo, oy = o / Wo;
        float acc[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = bias[c];
        for (int ky = 0; ky < 6; ++ky) {
            const int yy = oy * 2 - 2 + ky;
            if ((unsigned)yy >= (unsigned)H) continue;
            for (int kx = 0; kx < 6; ++kx) {
                const int xx = ox * 2 - 2 + kx;
                if ((unsigned)xx >= (unsigned)W) continue;
                const float v = mp[yy * W + xx];
                const float* wk = w + (ky * 6 + kx) * 16;
#pragma unroll
                for (int c = 0; c < 16; ++c) acc[c] += v * wk[c];
            }
        }
        T* op = out + (m * (long long)Ho * Wo + o) * ldo;
#pragma unroll
        for (int c = 0; c < 16; ++c) Elem<T>::st(op + c, acc[c] > 0.f ? acc[c] : 0.f);
    }
}
extern "C" int gvfi_cost_embed1(const float* vol, const float* w, const float* bias, void* out, int ldo, long long maps,
                                int H, int W, int Ho, int Wo, int dtype, void* stream) {
    if (ldo < 16) return -2;
    const long long shm = (long long)H * W * 4;
    if (shm <= 64 * 1024 && maps <= 0x7fffffffll && getenv("GVFI_COST_EMBED_LDS0") == nullptr) {
        GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_COOP_SHM((cost_embed1_lds_kernel<T>), dim3((unsigned)maps), dim3(256), (int)shm,
                                                    (hipStream_t)stream, vol, w, bias, (T*)out, ldo, H, W, Ho, Wo));
        return (int)hipGetLastError();
    }
    const long long total = maps * Ho * Wo;
    GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_SIMPLE((cost_embed1_kernel<T>), grid1d(total), dim3(GVFI_BLOCK), (hipStream_t)stream,
                                              vol, w, bias, (T*)out, ldo, total, H, W, Ho, Wo));
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------ 81-tap cost lookup   decoder.py:237-255
// out[q, i*win + j] = bilinear(cost_map[q]; x + (i - r), y + (j - r)), zeros outside, same float expression as
// bilinear_sampler + grid_sample(align_corners=True) (utils/utils.py:83-97); the window axes are transposed exactly
// as in RAFT's lookup (delta = stack(meshgrid(dy, dx)) is added to (x, y)).
template <typename T>
__global__ void cost_lookup_kernel(const float* __restrict__ maps, const float* __restrict__ coords, T* __restrict__ out,
                                   int ldo, long long total, int h, int w, int radius) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over (query, tap)
    if (idx >= total) return;
    const int win = 2 * radius + 1;
    const int tap = (int)(idx % (win * win));
    const long long q = idx / (win * win);
    const int i = tap / win, j = tap % win;
    const float* base = maps + q * (long long)h * w;
    const float cx = coords[q * 2 + 0] + (float)(i - radius);
    const float cy = coords[q * 2 + 1] + (float)(j - radius);
    const float xn = 2.f * cx / (float)(w - 1) - 1.f;
    const float yn = 2.f * cy / (float)(h - 1) - 1.f;
    const float ix = ((xn + 1.f) * 0.5f) * (float)(w - 1);
    const float iy = ((yn + 1.f) * 0.5f) * (float)(h - 1);
    const float x0f = floorf(ix), y0f = floorf(iy);
    // (cells far outside the map: every tap is absent; clamping first keeps the int conversion defined)
    const int x0 = (int)fminf(fmaxf(x0f, -4.f), (float)w + 4.f), y0 = (int)fminf(fmaxf(y0f, -4.f), (float)h + 4.f);
    const float ax = ix - x0f, ay = iy - y0f;
    const bool xin0 = x0 >= 0 && x0 < w, xin1 = x0 + 1 >= 0 && x0 + 1 < w;
    const bool yin0 = y0 >= 0 && y0 < h, yin1 = y0 + 1 >= 0 && y0 + 1 < h;
    // the four map reads as unconditional loads at clamped cells (in flight together); the conditions select what is added
    const int xa = x0 < 0 ? 0 : (x0 > w - 1 ? w - 1 : x0), xb = x0 + 1 < 0 ? 0 : (x0 + 1 > w - 1 ? w - 1 : x0 + 1);
    const int ya = y0 < 0 ? 0 : (y0 > h - 1 ? h - 1 : y0), yb = y0 + 1 < 0 ? 0 : (y0 + 1 > h - 1 ? h - 1 : y0 + 1);
    const float t00 = base[(long long)ya * w + xa], t01 = base[(long long)ya * w + xb];
    const float t10 = base[(long long)yb * w + xa], t11 = base[(long long)yb * w + xb];
    float v = 0.f;
    if (xin0 && yin0) v += (1.f - ax) * (1.f - ay) * t00;
    if (xin1 && yin0) v += ax * (1.f - ay) * t01;
    if (xin0 && yin1) v += (1.f - ax) * ay * t10;
    if (xin1 && yin1) v += ax * ay * t11;
    if (ax != ax || ay != ay) v = ax + ay;          // NaN coordinates stay visible
    Elem<T>::st(out + q * ldo + tap, v);
}
extern "C" int gvfi_cost_lookup(const float* maps, const float* coords, void* out, int ldo, long long Q, int h, int w,
                                int radius, int dtype, void* stream) {
    const int win = 2 * radius + 1;
    if (h < 2 || w < 2 || ldo < win * win) return -2;
    const long long total = Q * win * win;
    GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_SIMPLE((cost_lookup_kernel<T>), grid1d(total), dim3(GVFI_BLOCK), (hipStream_t)stream,
                                              maps, coords, (T*)out, ldo, total, h, w, radius));
    return (int)hipGetLastError();
}

// A/B switch of the MFMA attention kernels (attn_mfma.hip): GVFI_ATTN_MFMA=0 keeps the scalar kernels below
static bool attn_mfma_enabled() {
#ifdef GVFI_HOSTSIM
    // emulator: only on request (the engine-level emulations with a torch statement of the convolutions stay at seconds); read at
    // every call so that a kernel test can switch it on inside a long-lived test process
    const char* eh = getenv("GVFI_ATTN_MFMA");
    return eh != nullptr && eh[0] == '1';
#endif
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("GVFI_ATTN_MFMA");
        on = (e != nullptr && e[0] == '0') ? 0 : 1;
    }
    return on == 1;
}

// ------------------------------------------------------------------ attention over small key sets
// One thread = one (query row, head): soft-max(q.k_j * scale) over the keys in one pass (running maximum), V accumulated
// in registers.  HD = head dimension (8, 16, 32).
template <typename T, int HD> struct AttnAcc {
    float q[HD], o[HD], m, l;
    __device__ __forceinline__ void init(const T* qp) {
#pragma unroll
        for (int d = 0; d < HD; ++d) { q[d] = Elem<T>::ld(qp + d); o[d] = 0.f; }
        m = -INFINITY;
        l = 0.f;
    }
    template <typename KT> __device__ __forceinline__ void key(const KT* kp, const KT* vp, float scale) {
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) s += q[d] * Elem<KT>::ld(kp + d);
        s *= scale;
        const float mn = fmaxf(m, s);
        const float a = expf(m - mn), p = expf(s - mn);   // expf(-inf) = 0 on the first key
        l = l * a + p;
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] = o[d] * a + p * Elem<KT>::ld(vp + d);
        m = mn;
    }
    __device__ __forceinline__ void store(T* op) {
        const float inv = 1.0f / l;
#pragma unroll
        for (int d = 0; d < HD; ++d) Elem<T>::st(op + d, o[d] * inv);
    }
};

// Locally-grouped attention (twins.py:814-867, 331-427): tokens on an H x W grid per image, a query attends to the ws x ws
// window that contains it; window positions beyond the grid (the reference zero-pads the token grid AFTER the norm,
// so those tokens carry the projection of zero [+ positional code]) use kpad / vpad [ws*ws][C] (float).
template <typename T, int HD>
__global__ void attn_window_kernel(const T* __restrict__ q, int ldq, const T* __restrict__ k, int ldk,
                                   const T* __restrict__ v, int ldv, const float* __restrict__ kpad,
                                   const float* __restrict__ vpad, T* __restrict__ out, int ldo, long long total, int H,
                                   int W, int ws, int heads, float scale) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over (token row, head)
    if (idx >= total) return;
    const int hd = (int)(idx % heads);
    const long long row = idx / heads;
    const int px = (int)(row % W), py = (int)((row / W) % H);
    const long long img0 = row - (long long)py * W - px;
    const int wy = py / ws * ws, wx = px / ws * ws;
    const int C = heads * HD;
    AttnAcc<T, HD> acc;
    acc.init(q + row * ldq + hd * HD);
    for (int dy = 0; dy < ws; ++dy)
        for (int dx = 0; dx < ws; ++dx) {
            const int yy = wy + dy, xx = wx + dx;
            if (yy < H && xx < W) {
                const long long kr = img0 + (long long)yy * W + xx;
                acc.key(k + kr * ldk + hd * HD, v + kr * ldv + hd * HD, scale);
            } else {
                const int pos = dy * ws + dx;
                acc.key(kpad + pos * C + hd * HD, vpad + pos * C + hd * HD, scale);
            }
        }
    acc.store(out + row * ldo + hd * HD);
}
// (LDS-staged variants of this kernel and of attn_global_kernel -- a workgroup staging its window's / group's key and
// value rows once -- were measured in round 2: 130.3 vs 130.6 frames/s, i.e. the per-key global loads are served by
// L1/L2 as fast as LDS serves them; removed.)
extern "C" int gvfi_attn_window(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const float* kpad,
                                const float* vpad, void* out, int ldo, int n_img, int H, int W, int ws, int heads,
                                int head_dim, float scale, int dtype, void* stream) {
    // bf16 / IEEE half, head dimension 16 / 32, 7x7 windows: one wave per (window, head) on the matrix pipe (attn_mfma.hip)
    if (attn_mfma_enabled() && gvfi_attn_mfma_ok(1, ws * ws, ws * ws, head_dim, ws, dtype)) {
        const int rc = (dtype == GVFI_F16 ? gvfi_attn_window_mfma_f16 : gvfi_attn_window_mfma)(q, ldq, k, ldk, v, ldv, kpad, vpad, out, ldo, n_img,
                                                                                               H, W, ws, heads, head_dim, scale, stream);
        if (rc != -3) return rc;      // (-3: a pointer / pitch the vector loads cannot take -> scalar kernel)
    }
    const long long total = (long long)n_img * H * W * heads;
#define GVFI_AW(HD_)                                                                                                 \
    GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_SIMPLE((attn_window_kernel<T, HD_>), grid1d(total), dim3(GVFI_BLOCK),          \
                                              (hipStream_t)stream, (const T*)q, ldq, (const T*)k, ldk, (const T*)v, ldv, \
                                              kpad, vpad, (T*)out, ldo, total, H, W, ws, heads, scale))
    if (head_dim == 8) GVFI_AW(8);
    else if (head_dim == 16) GVFI_AW(16);
    else if (head_dim == 32) GVFI_AW(32);
    else return -2;
#undef GVFI_AW
    return (int)hipGetLastError();
}

// Attention of NQ queries per group against M keys per group (twins.py:870-925, 430-546; attention.py:10-66;
// encoder.py:214-346; decoder.py:35-120).  Groups g = (g1, g0), g0 < G0; rows are addressed as
//   query  : g1*qb1 + g0*qb0 + i*qs      key/value : g1*kb1 + g0*kb0 + j*ks      output : g1*ob1 + g0*ob0 + i*os
// which covers batched global attention (G0 = 1), one shared query set (qb1 = qb0 = 0), the self-attention over the K
// latent tokens of a cost map and the decoder's one-query cross-attention in the image-major latent layout
// [(b, k)][p] (key stride P).
template <typename T, int HD>
__global__ void attn_global_kernel(const T* __restrict__ q, int ldq, long long qb1, long long qb0, long long qs,
                                   const T* __restrict__ k, int ldk, const T* __restrict__ v, int ldv, long long kb1,
                                   long long kb0, long long ks, T* __restrict__ out, int ldo, long long ob1, long long ob0,
                                   long long os, long long total, int G0, int NQ, int M, int heads, float scale) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over (g1, g0, i, head)
    if (idx >= total) return;
    const int hd = (int)(idx % heads);
    long long r = idx / heads;
    const int i = (int)(r % NQ);
    r /= NQ;
    const long long g0 = r % G0, g1 = r / G0;
    AttnAcc<T, HD> acc;
    acc.init(q + (g1 * qb1 + g0 * qb0 + i * qs) * ldq + hd * HD);
    const long long kbase = g1 * kb1 + g0 * kb0;
    for (int j = 0; j < M; ++j) {
        const long long kr = kbase + j * ks;
        acc.key(k + kr * ldk + hd * HD, v + kr * ldv + hd * HD, scale);
    }
    acc.store(out + (g1 * ob1 + g0 * ob0 + i * os) * ldo + hd * HD);
}
extern "C" int gvfi_attn_global(const void* q, int ldq, long long qb1, long long qb0, long long qs, const void* k, int ldk,
                                const void* v, int ldv, long long kb1, long long kb0, long long ks, void* out, int ldo,
                                long long ob1, long long ob0, long long os, long long G1, int G0, int NQ, int M, int heads,
                                int head_dim, float scale, int dtype, void* stream) {
    if (G0 <= 0 || NQ <= 0 || M <= 0) return -2;
    if (attn_mfma_enabled() && gvfi_attn_mfma_ok(0, M, NQ, head_dim, 0, dtype)) {
        const int rc = (dtype == GVFI_F16 ? gvfi_attn_global_mfma_f16 : gvfi_attn_global_mfma)(q, ldq, qb1, qb0, qs, k, ldk, v, ldv, kb1, kb0, ks, out,
                                                                                               ldo, ob1, ob0, os, G1, G0, NQ, M, heads, head_dim, scale,
                                                                                               stream);
        if (rc != -3) return rc;
    }
    const long long total = G1 * G0 * NQ * heads;
#define GVFI_AG(HD_)                                                                                                   \
    GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_SIMPLE((attn_global_kernel<T, HD_>), grid1d(total), dim3(GVFI_BLOCK),            \
                                              (hipStream_t)stream, (const T*)q, ldq, qb1, qb0, qs, (const T*)k, ldk,     \
                                              (const T*)v, ldv, kb1, kb0, ks, (T*)out, ldo, ob1, ob0, os, total, G0, NQ, \
                                              M, heads, scale))
    if (head_dim == 8) GVFI_AG(8);
    else if (head_dim == 16) GVFI_AG(16);
    else if (head_dim == 32) GVFI_AG(32);
    else return -2;
#undef GVFI_AG
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------ [x | context] (+ positional code)   twins.py:366-395, 465-493
// Row r = (im, p) of n_img images with P = H*W tokens each: out[r, 0:Cx] = x[r], out[r, Cx:Cx+Cc] = ctx[cimg(im)][p].
// The reference tiles the context batch (`context.repeat(B // nb, 1, 1, 1)`, twins.py:366) over the (batch, latent
// token) axis, so image (b, k) of a direction with nb pairs reads context (b*K + k) % nb -- reproduced here:
// cimg = d*nb + (im % (nb*K)) % nb with d = im / (nb*K).
// enc_mode 1: += enc(x % ws, y % ws) (window position), 2: += enc(x, y); over all Cx + Cc channels.
template <typename T>
__global__ void xqk_kernel(const T* __restrict__ x, int ldx, int Cx, const T* __restrict__ ctx, int ldc, int Cc,
                           T* __restrict__ out, int ldo, long long total, int P, int W, int K, int nb, int enc_mode,
                           int ws) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over (row, channel)
    if (idx >= total) return;
    const int Ct = Cx + Cc;
    const int c = (int)(idx % Ct);
    const long long r = idx / Ct;
    const int p = (int)(r % P);
    const long long im = r / P;
    float val;
    if (c < Cx) {
        val = Elem<T>::ld(x + r * ldx + c);
    } else {
        const long long per = (long long)nb * K;
        const long long cimg = (im / per) * nb + (im % per) % nb;
        val = Elem<T>::ld(ctx + (cimg * P + p) * ldc + (c - Cx));
    }
    if (enc_mode) {
        int px = p % W, py = p / W;
        if (enc_mode == 1) { px %= ws; py %= ws; }
        val += pos_enc_channel((float)px, (float)py, c, Ct);
    }
    Elem<T>::st(out + r * ldo + c, val);
}
// table[pos][c] = enc(x, y)[c] for the positions xqk adds (enc_mode 1: the ws*ws window positions, 2: the H*W grid
// positions), Ct channels -- the same per-channel expression as the inline evaluation above, computed once per forward
// instead of once per (row, channel): the sin / cos of 229 k rows x 192 channels were 1.6 ms of a GIMM-VFI-F step
__global__ void pos_table_kernel(float* __restrict__ table, int npos, int Ct, int W, int enc_mode, int ws) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npos * Ct) return;
    const int c = idx % Ct, pos = idx / Ct;
    const int px = enc_mode == 1 ? pos % ws : pos % W, py = enc_mode == 1 ? pos / ws : pos / W;
    table[idx] = pos_enc_channel((float)px, (float)py, c, Ct);
}
extern "C" int gvfi_ff_pos_table(float* table, int H, int W, int Ct, int enc_mode, int ws, void* stream) {
    if ((enc_mode != 1 && enc_mode != 2) || (Ct & 3) || (enc_mode == 1 && ws <= 0)) return -2;
    const int npos = enc_mode == 1 ? ws * ws : H * W;
    GVFI_LAUNCH_SIMPLE(pos_table_kernel, grid1d((long long)npos * Ct), dim3(GVFI_BLOCK), (hipStream_t)stream, table, npos, Ct,
                       W, enc_mode, ws);
    return (int)hipGetLastError();
}
// vector form: one 16-byte group of channels per thread, positional code from the table
template <typename T>
__global__ void xqk_vec_kernel(const T* __restrict__ x, int ldx, int Cx, const T* __restrict__ ctx, int ldc, int Cc,
                               T* __restrict__ out, int ldo, long long total, int P, int W, int K, int nb, int enc_mode,
                               int ws, const float* __restrict__ table) {
    constexpr int VE = Elem<T>::VE;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over (row, channel group)
    if (idx >= total) return;
    const int Ct = Cx + Cc, G = Ct / VE;
    const int c = (int)(idx % G) * VE;
    const long long r = idx / G;
    const int p = (int)(r % P);
    const long long im = r / P;
    struct alignas(16) Vec { T e[VE]; };
    Vec v;
    if (c < Cx) {
        v = *(const Vec*)(x + r * ldx + c);
    } else {
        const long long per = (long long)nb * K;
        const long long cimg = (im / per) * nb + (im % per) % nb;
        v = *(const Vec*)(ctx + (cimg * P + p) * ldc + (c - Cx));
    }
    if (enc_mode) {
        const int px = p % W, py = p / W;
        const int pos = enc_mode == 1 ? (py % ws) * ws + (px % ws) : p;
        const float* tr = table + (long long)pos * Ct + c;
#pragma unroll
        for (int e = 0; e < VE; ++e) Elem<T>::st(&v.e[e], Elem<T>::ld(&v.e[e]) + tr[e]);
    }
    *(Vec*)(out + r * ldo + c) = v;
}
extern "C" int gvfi_ff_xqk(const void* x, int ldx, int Cx, const void* ctx, int ldc, int Cc, void* out, int ldo, int n_img,
                           int H, int W, int K, int nb, int enc_mode, int ws, const float* enc_table, int dtype,
                           void* stream) {
    if (((Cx + Cc) & 3) || nb <= 0 || K <= 0 || (enc_mode == 1 && ws <= 0)) return -2;
    const int ve = dtype == GVFI_F32 ? 4 : 8;
    const bool vec = (enc_mode == 0 || enc_table != nullptr) && Cx % ve == 0 && Cc % ve == 0 && ldx % ve == 0 &&
                     ldc % ve == 0 && ldo % ve == 0 && ((((uintptr_t)x) | ((uintptr_t)ctx) | ((uintptr_t)out)) & 15) == 0;
    if (vec) {
        const long long totv = (long long)n_img * H * W * ((Cx + Cc) / ve);
        GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_SIMPLE((xqk_vec_kernel<T>), grid1d(totv), dim3(GVFI_BLOCK), (hipStream_t)stream,
                                                  (const T*)x, ldx, Cx, (const T*)ctx, ldc, Cc, (T*)out, ldo, totv, H * W,
                                                  W, K, nb, enc_mode, ws, enc_table));
        return (int)hipGetLastError();
    }
    const long long total = (long long)n_img * H * W * (Cx + Cc);
    GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_SIMPLE((xqk_kernel<T>), grid1d(total), dim3(GVFI_BLOCK), (hipStream_t)stream,
                                              (const T*)x, ldx, Cx, (const T*)ctx, ldc, Cc, (T*)out, ldo, total, H * W, W,
                                              K, nb, enc_mode, ws));
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------ out[row] = table[(row / P) % K]   encoder.py:420 (latent tokens)
template <typename T>
__global__ void tile_rows_kernel(const float* __restrict__ table, void* __restrict__ out, int ldo, int out_f32,
                                 long long total, int P, int K, int C) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const long long r = idx / C;
    st_any<T>(out, r * ldo + c, out_f32, table[((r / P) % K) * C + c]);
}
extern "C" int gvfi_tile_rows(const float* table, void* out, int ldo, int out_f32, long long rows, int P, int K, int C,
                              int dtype, void* stream) {
    const long long total = rows * C;
    GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_SIMPLE((tile_rows_kernel<T>), grid1d(total), dim3(GVFI_BLOCK), (hipStream_t)stream,
                                              table, out, ldo, out_f32, total, P, K, C));
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------ row soft-max of the GMA similarity   gma.py:70-74
// x float [rows][n] -> y [rows][ldy] in the activation type (pad columns n..ldy zeroed: y is the A operand of the
// attention*V contraction).  One 64-lane wave per row.
template <typename T>
__global__ void softmax_rows_kernel(const float* __restrict__ x, int n, T* __restrict__ y, int ldy, long long rows) {
    const long long r = (long long)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const int lane = threadIdx.x & 63;
    const bool live = r < rows;
    const float* xr = x + (live ? r : 0) * (long long)n;
    float m = -INFINITY;
    if (live)
        for (int c = lane; c < n; c += 64) m = fmaxf(m, xr[c]);
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    float s = 0.f;
    if (live)
        for (int c = lane; c < n; c += 64) s += expf(xr[c] - m);
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (!live) return;
    const float inv = 1.0f / s;
    T* yr = y + r * ldy;
    for (int c = lane; c < ldy; c += 64) Elem<T>::st(yr + c, c < n ? expf(xr[c] - m) * inv : 0.f);
}
extern "C" int gvfi_softmax_rows(const float* x, int n, void* y, int ldy, long long rows, int dtype, void* stream) {
    if (n <= 0 || ldy < n) return -2;
    const int wpb = GVFI_BLOCK / 64;
    dim3 grid((unsigned)((rows + wpb - 1) / wpb));
    GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_COOP((softmax_rows_kernel<T>), grid, dim3(GVFI_BLOCK), (hipStream_t)stream, x, n,
                                            (T*)y, ldy, rows));
    return (int)hipGetLastError();
}
