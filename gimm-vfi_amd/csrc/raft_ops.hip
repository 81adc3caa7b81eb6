// RAFT-side HBM-bound kernels: correlation pyramid pooling, 4-level 9x9 bilinear correlation
// lookup, coordinate grid, flow packing and the convex 8x flow upsampler.
#include "common.h"

#define GVFI_BLOCK 256
static inline dim3 grid1d(long long n) { return dim3((unsigned)((n + GVFI_BLOCK - 1) / GVFI_BLOCK)); }

// ------------------------------------------------------------------ F.avg_pool2d(.., 2, stride=2)   raft/corr.py:139-142
__global__ void avgpool2_kernel(const float* __restrict__ src, float* __restrict__ dst, long long total, int h, int w,
                                int ho, int wo) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int x = (int)(idx % wo);
    const int y = (int)((idx / wo) % ho);
    const long long m = idx / ((long long)wo * ho);
    const float* s = src + m * (long long)h * w + (long long)(2 * y) * w + 2 * x;
    dst[idx] = 0.25f * (s[0] + s[1] + s[w] + s[w + 1]);
}
extern "C" int gvfi_avgpool2_f32(const float* src, float* dst, long long maps, int h, int w, void* stream) {
    const int ho = h / 2, wo = w / 2;
    const long long total = maps * ho * wo;
    GVFI_LAUNCH_SIMPLE(avgpool2_kernel, grid1d(total), dim3(GVFI_BLOCK), (hipStream_t)stream, src, dst, total, h, w, ho,
                       wo);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------ correlation lookup   raft/corr.py:144-165
// Output channel  l*(2r+1)^2 + i*(2r+1) + j  samples level l of the query's correlation map at
// (x = cx/2^l + (i-r),  y = cy/2^l + (j-r)):  the FIRST window index moves x (the reference adds
// meshgrid(dy,dx) to (x,y) coordinates).  grid_sample(align_corners=True, zeros padding) semantics
// including the normalise/un-normalise round trip of raft/utils/utils.py:66-80.
// One thread = one (query, level, dx) column of the window: its 2r+1 outputs (dy = -r..r) share the two source
// columns x0, x0+1, so the 2r+2 rows are loaded once (20 loads for 9 outputs instead of 36) and the outputs are 9
// consecutive channels.  Every output is computed with exactly the reference's per-tap float expression; when the
// normalise/un-normalise rounding moves a tap's row off the cached run, that tap falls back to direct loads.
#define CORR_MAX_WIN 9
template <typename T>
__global__ void corr_lookup_kernel(const float* __restrict__ l0, const float* __restrict__ l1,
                                   const float* __restrict__ l2, const float* __restrict__ l3,
                                   const float* __restrict__ coords, T* __restrict__ out, int ldo, long long total,
                                   int h2, int w2, int radius, long long src_nq) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int win = 2 * radius + 1;
    const int i = (int)(idx % win);
    const int l = (int)((idx / win) & 3);
    const long long q = idx / (4 * win);  // global query index (n*h*w + y*w + x)
    const int hl = h2 >> l, wl = w2 >> l;
    // src_nq > 0: the volume holds src_nq maps and query q reads map q % src_nq (timestep-batched look-ups of one pyramid)
    const float* base = (l == 0 ? l0 : (l == 1 ? l1 : (l == 2 ? l2 : l3))) + (src_nq > 0 ? q % src_nq : q) * (long long)hl * wl;
    const float sc = 1.0f / (float)(1 << l);
    const float qx = coords[q * 2 + 0] * sc, qy = coords[q * 2 + 1] * sc;
    // bilinear_sampler: normalise to [-1,1] then grid_sample un-normalises (align_corners=True)
    const float cx = qx + (float)(i - radius);
    const float xn = 2.f * cx / (float)(wl - 1) - 1.f;
    const float ix = ((xn + 1.f) * 0.5f) * (float)(wl - 1);
    const float x0f = floorf(ix);
    const int x0 = (int)x0f;
    const float ax = ix - x0f;
    const bool xin0 = x0 >= 0 && x0 < wl, xin1 = x0 + 1 >= 0 && x0 + 1 < wl;
    auto row_of = [&](int j, float& ay) {
        const float cy = qy + (float)(j - radius);
        const float yn = 2.f * cy / (float)(hl - 1) - 1.f;
        const float iy = ((yn + 1.f) * 0.5f) * (float)(hl - 1);
        const float y0f = floorf(iy);
        ay = iy - y0f;
        return (int)y0f;
    };
    auto fetch = [&](int y, float& a, float& b) {
        const bool yin = y >= 0 && y < hl;
        a = (yin && xin0) ? base[(long long)y * wl + x0] : 0.f;
        b = (yin && xin1) ? base[(long long)y * wl + x0 + 1] : 0.f;
    };
    float ay0;
    const int ybase = row_of(0, ay0);
    float ca[CORR_MAX_WIN + 1], cb[CORR_MAX_WIN + 1];
#pragma unroll
    for (int k = 0; k <= CORR_MAX_WIN; ++k)
        if (k <= win) fetch(ybase + k, ca[k], cb[k]);
    T* o = out + q * ldo + l * win * win + i * win;
#pragma unroll
    for (int j = 0; j < CORR_MAX_WIN; ++j) {
        if (j >= win) break;
        float ay;
        const int y0 = row_of(j, ay);
        float v00, v01, v10, v11;
        if (y0 == ybase + j) {
            v00 = ca[j]; v01 = cb[j]; v10 = ca[j + 1]; v11 = cb[j + 1];
        } else {
            fetch(y0, v00, v01);
            fetch(y0 + 1, v10, v11);
        }
        // same accumulation order as the per-tap statement: (x0,y0), (x1,y0), (x0,y1), (x1,y1); absent taps add 0
        float v = 0.f;
        const bool yin0 = y0 >= 0 && y0 < hl, yin1 = y0 + 1 >= 0 && y0 + 1 < hl;
        if (xin0 && yin0) v += (1.f - ax) * (1.f - ay) * v00;
        if (xin1 && yin0) v += ax * (1.f - ay) * v01;
        if (xin0 && yin1) v += (1.f - ax) * ay * v10;
        if (xin1 && yin1) v += ax * ay * v11;
        Elem<T>::st(o + j, v);
    }
}
extern "C" int gvfi_corr_lookup(const float* l0, const float* l1, const float* l2, const float* l3,
                                const float* coords, void* out, int ldo, int dtype, int N, int src_N, int h, int w, int h2,
                                int w2, int radius, void* stream) {
    if (src_N < 0 || src_N > N) return -2;
    const int win = 2 * radius + 1;
    const long long total = (long long)N * h * w * 4 * win;
    if ((h2 >> 3) < 2 || (w2 >> 3) < 2) return -2;  // coarsest level must be >= 2x2 (reference divides by W-1)
    if (win > CORR_MAX_WIN) return -2;
    GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_SIMPLE((corr_lookup_kernel<T>), grid1d(total), dim3(GVFI_BLOCK),
                                              (hipStream_t)stream, l0, l1, l2, l3, coords, (T*)out, ldo, total, h2, w2,
                                              radius, (long long)src_N * h * w));
    return (int)hipGetLastError();
}

// LDS-staged variant (BASELINE.json north_star: "correlation volume staged through LDS for coalesced HBM reads"; A/B'd
// against the kernel above in round 3, profiles/r3_lookup_ab.txt).  A workgroup owns LQ queries.  Phase 1: the (2r+4)^2
// window of every level of every query -- rows of 12 consecutive floats, zero outside the map -- is copied to LDS by
// consecutive lanes (48-byte row segments; the one-column-per-thread kernel reads every float of the window twice).  Phase
// 2: the outputs, consecutive lanes = consecutive channels of a query (coalesced 2-byte stores), each with exactly the
// per-tap float expression of the kernel above, its four taps read from LDS.  Note: a query's window lies in its OWN
// correlation map (row q of the volume), so there is no reuse across queries for LDS to exploit -- the bytes that cross
// the fabric are the same 128-byte lines either way; what staging changes is the number of load instructions.
#define LQ 8
#define LWIN 12     // staged rows / columns per level: 2r+2 taps + one spare on either side (rounding of the round trip)
template <typename T>
__global__ void __launch_bounds__(256) corr_lookup_lds_kernel(const float* __restrict__ l0, const float* __restrict__ l1,
                                                              const float* __restrict__ l2, const float* __restrict__ l3,
                                                              const float* __restrict__ coords, T* __restrict__ out, int ldo,
                                                              long long nq, int h2, int w2, int radius, long long src_nq) {
    __shared__ float win[LQ][4][LWIN][LWIN];
    __shared__ int org[LQ][4][2];          // map coordinates of the staged window's corner (x, y)
    const int tid = threadIdx.x;
    const long long q0 = (long long)blockIdx.x * LQ;
    const int nwin = 2 * radius + 1;
    auto level_geom = [&](int l, long long q, int& hl, int& wl, float& qx, float& qy) {
        hl = h2 >> l;
        wl = w2 >> l;
        const float sc = 1.0f / (float)(1 << l);
        qx = coords[q * 2 + 0] * sc;
        qy = coords[q * 2 + 1] * sc;
    };
    // the window corner: the cell of tap (0, 0) minus the spare row / column
    auto tap_cell = [&](float c0, int n) {
        const float xn = 2.f * c0 / (float)(n - 1) - 1.f;
        const float ix = ((xn + 1.f) * 0.5f) * (float)(n - 1);
        return floorf(ix);
    };
    if (tid < LQ * 4) {
        const int ql = tid >> 2, l = tid & 3;
        const long long q = q0 + ql;
        if (q < nq) {
            int hl, wl;
            float qx, qy;
            level_geom(l, q, hl, wl, qx, qy);
            // (clamped far outside the map: the whole window is zeros there and int conversion must not overflow)
            const float fx = fminf(fmaxf(tap_cell(qx - (float)radius, wl), -1.0e6f), 1.0e6f);
            const float fy = fminf(fmaxf(tap_cell(qy - (float)radius, hl), -1.0e6f), 1.0e6f);
            org[ql][l][0] = (int)fx - 1;
            org[ql][l][1] = (int)fy - 1;
        }
    }
    __syncthreads();
    for (int idx = tid; idx < LQ * 4 * LWIN * LWIN; idx += 256) {
        const int c = idx % LWIN, r = (idx / LWIN) % LWIN, l = (idx / (LWIN * LWIN)) & 3, ql = idx / (4 * LWIN * LWIN);
        const long long q = q0 + ql;
        float v = 0.f;
        if (q < nq) {
            const int hl = h2 >> l, wl = w2 >> l;
            const int x = org[ql][l][0] + c, y = org[ql][l][1] + r;
            if (x >= 0 && x < wl && y >= 0 && y < hl) {
                const float* base = (l == 0 ? l0 : (l == 1 ? l1 : (l == 2 ? l2 : l3))) + (src_nq > 0 ? q % src_nq : q) * (long long)hl * wl;
                v = base[(long long)y * wl + x];
            }
        }
        win[ql][l][r][c] = v;
    }
    __syncthreads();
    const int per_q = 4 * nwin * nwin;
    for (int o = tid; o < LQ * per_q; o += 256) {
        const int ql = o / per_q, ch = o - ql * per_q;
        const long long q = q0 + ql;
        if (q >= nq) continue;
        const int l = ch / (nwin * nwin), i = (ch / nwin) % nwin, j = ch % nwin;
        int hl, wl;
        float qx, qy;
        level_geom(l, q, hl, wl, qx, qy);
        const float cx = qx + (float)(i - radius);
        const float xn = 2.f * cx / (float)(wl - 1) - 1.f;
        const float ix = ((xn + 1.f) * 0.5f) * (float)(wl - 1);
        const float x0f = floorf(ix);
        const float ax = ix - x0f;
        const float cy = qy + (float)(j - radius);
        const float yn = 2.f * cy / (float)(hl - 1) - 1.f;
        const float iy = ((yn + 1.f) * 0.5f) * (float)(hl - 1);
        const float y0f = floorf(iy);
        const float ay = iy - y0f;
        // cells relative to the staged corner; far outside the map everything is zero
        const float rxf = x0f - (float)org[ql][l][0], ryf = y0f - (float)org[ql][l][1];
        float v = 0.f;
        if (rxf >= 0.f && rxf <= (float)(LWIN - 2) && ryf >= 0.f && ryf <= (float)(LWIN - 2)) {
            const int rx = (int)rxf, ry = (int)ryf;
            const int x0 = org[ql][l][0] + rx, y0 = org[ql][l][1] + ry;
            const bool xin0 = x0 >= 0 && x0 < wl, xin1 = x0 + 1 >= 0 && x0 + 1 < wl;
            const bool yin0 = y0 >= 0 && y0 < hl, yin1 = y0 + 1 >= 0 && y0 + 1 < hl;
            const float v00 = win[ql][l][ry][rx], v01 = win[ql][l][ry][rx + 1];
            const float v10 = win[ql][l][ry + 1][rx], v11 = win[ql][l][ry + 1][rx + 1];
            // same accumulation order as the per-tap statement of corr_lookup_kernel; absent taps add nothing
            if (xin0 && yin0) v += (1.f - ax) * (1.f - ay) * v00;
            if (xin1 && yin0) v += ax * (1.f - ay) * v01;
            if (xin0 && yin1) v += (1.f - ax) * ay * v10;
            if (xin1 && yin1) v += ax * ay * v11;
        } else if (x0f > -2.f && x0f < (float)wl && y0f > -2.f && y0f < (float)hl) {
            // (a tap whose cell left the staged window through rounding: direct reads, same expression)
            const int x0 = (int)x0f, y0 = (int)y0f;
            const float* base = (l == 0 ? l0 : (l == 1 ? l1 : (l == 2 ? l2 : l3))) + (src_nq > 0 ? q % src_nq : q) * (long long)hl * wl;
            const bool xin0 = x0 >= 0 && x0 < wl, xin1 = x0 + 1 >= 0 && x0 + 1 < wl;
            const bool yin0 = y0 >= 0 && y0 < hl, yin1 = y0 + 1 >= 0 && y0 + 1 < hl;
            if (xin0 && yin0) v += (1.f - ax) * (1.f - ay) * base[(long long)y0 * wl + x0];
            if (xin1 && yin0) v += ax * (1.f - ay) * base[(long long)y0 * wl + x0 + 1];
            if (xin0 && yin1) v += (1.f - ax) * ay * base[(long long)(y0 + 1) * wl + x0];
            if (xin1 && yin1) v += ax * ay * base[(long long)(y0 + 1) * wl + x0 + 1];
        }
        Elem<T>::st(out + q * ldo + ch, v);
    }
}
extern "C" int gvfi_corr_lookup_lds(const float* l0, const float* l1, const float* l2, const float* l3,
                                    const float* coords, void* out, int ldo, int dtype, int N, int src_N, int h, int w, int h2,
                                    int w2, int radius, void* stream) {
    const long long nq = (long long)N * h * w;
    if (src_N < 0 || src_N > N) return -2;
    if ((h2 >> 3) < 2 || (w2 >> 3) < 2 || radius != 4) return -2;
    const unsigned grid = (unsigned)((nq + LQ - 1) / LQ);
    GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_COOP((corr_lookup_lds_kernel<T>), dim3(grid), dim3(256), (hipStream_t)stream, l0,
                                            l1, l2, l3, coords, (T*)out, ldo, nq, h2, w2, radius, (long long)src_N * h * w));
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------ coords grid   raft/utils/utils.py:83-88
__global__ void coords_init_kernel(float* __restrict__ coords, long long total, int h, int w) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    coords[idx * 2 + 0] = (float)(idx % w);
    coords[idx * 2 + 1] = (float)((idx / w) % h);
}
extern "C" int gvfi_coords_init(float* coords, int N, int h, int w, void* stream) {
    const long long total = (long long)N * h * w;
    GVFI_LAUNCH_SIMPLE(coords_init_kernel, grid1d(total), dim3(GVFI_BLOCK), (hipStream_t)stream, coords, total, h, w);
    return (int)hipGetLastError();
}

// flow = coords1 - coords0   raft/raft.py:148 ; written where the motion encoder / GRU read it
template <typename T>
__global__ void flow_pack_kernel(const float* __restrict__ coords1, T* __restrict__ dst0, int ld0, int pad0,
                                 T* __restrict__ dst1, int ld1, long long total, int h, int w) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const float fx = coords1[idx * 2 + 0] - (float)(idx % w);
    const float fy = coords1[idx * 2 + 1] - (float)((idx / w) % h);
    T* d0 = dst0 + idx * ld0;
    Elem<T>::st(d0 + 0, fx);
    Elem<T>::st(d0 + 1, fy);
    for (int c = 2; c < pad0; ++c) Elem<T>::st(d0 + c, 0.f);
    if (dst1) {
        Elem<T>::st(dst1 + idx * ld1 + 0, fx);
        Elem<T>::st(dst1 + idx * ld1 + 1, fy);
    }
}
extern "C" int gvfi_flow_pack(const float* coords1, void* dst0, int ld0, int pad0, void* dst1, int ld1, int N, int h,
                              int w, int dtype, void* stream) {
    const long long total = (long long)N * h * w;
    GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_SIMPLE((flow_pack_kernel<T>), grid1d(total), dim3(GVFI_BLOCK),
                                              (hipStream_t)stream, coords1, (T*)dst0, ld0, pad0, (T*)dst1, ld1, total,
                                              h, w));
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------ tap sum   raft/update.py:6-14 (FlowHead.conv2)
// A KHxKW convolution with 2-4 output channels over hundreds of input channels is a latency chain on the GEMM
// engine (K = 2304 per 128-pixel tile for two useful columns).  It is evaluated as a 1x1 convolution producing the
// KH*KW*C per-tap partial sums P[pixel][tap*C + c] (one pass over the input, all MFMA columns useful) followed by this
// gather:  out[n,y,x,c] = res + bias[c] + sum_tap P[n, y+dy, x+dx, tap*C + c]   (zero outside the image).
__global__ void tap_sum_kernel(const float* __restrict__ P, int ldp, int C, int KH, int KW, const float* __restrict__ bias,
                               const float* __restrict__ res, int ldr, float* __restrict__ out, int ldo, long long total,
                               int H, int W) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over (pixel, channel)
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const long long pix = idx / C;
    const int x = (int)(pix % W);
    const int y = (int)((pix / W) % H);
    float v = bias ? bias[c] : 0.f;
    int tap = 0;
    for (int kh = 0; kh < KH; ++kh) {
        const int yy = y + kh - KH / 2;
        for (int kw = 0; kw < KW; ++kw, ++tap) {
            const int xx = x + kw - KW / 2;
            if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
                v += P[(pix + (long long)(kh - KH / 2) * W + (kw - KW / 2)) * ldp + tap * C + c];
        }
    }
    if (res) v += res[pix * ldr + c];
    out[pix * ldo + c] = v;
}
extern "C" int gvfi_tap_sum(const float* P, int ldp, int C, int KH, int KW, const float* bias, const float* res, int ldr,
                            float* out, int ldo, int N, int H, int W, void* stream) {
    if (C <= 0 || ldp < KH * KW * C) return -2;
    const long long total = (long long)N * H * W * C;
    GVFI_LAUNCH_SIMPLE(tap_sum_kernel, grid1d(total), dim3(GVFI_BLOCK), (hipStream_t)stream, P, ldp, C, KH, KW, bias, res,
                       ldr, out, ldo, total, H, W);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------ fused iteration seam of the recurrence (round 3)
// Between two update iterations three small launches ran back to back on every image: gvfi_tap_sum (coords1 += flow-head
// output of iteration i, raft/update.py:6-14 + raft/raft.py:158-159), gvfi_flow_pack (flow = coords1 - coords0 as an
// activation, raft/raft.py:150) and gvfi_im2col (the 7x7 patch of that flow for convf1, raft/update.py:100): 19 us of
// ~200 us per iteration, all latency.  One workgroup per 8x8 pixel tile: the updated coordinates of the tile and its
// 3-pixel halo are (re)computed from the per-tap partial sums P (9 x 2 floats per pixel; halo pixels redundantly -- same
// values, the data is L2 resident), the flow tile goes to LDS in the activation type, the centre pixels write coords1 /
// flow / the flow slot of the GRU input, and the 98 (+ zero padded) patch entries of every pixel are gathered from LDS.
// Arithmetic and rounding are those of the three kernels in sequence (bit-identical: flow_step_case).
// P == nullptr: no pending update (first iteration): coords1 is read as it is.  With an update the new coordinates go to a
// SECOND tensor (coords_out != coords_in): neighbouring workgroups read each other's centre pixels as halo, an in-place
// update would race with them.
#define FS_T 8
#define FS_HALO 3
#define FS_W (FS_T + 2 * FS_HALO)
template <typename T>
__global__ void __launch_bounds__(256) flow_step_kernel(const float* __restrict__ P, int ldp, const float* __restrict__ bias,
                                                        const float* __restrict__ coords, float* __restrict__ coords_out,
                                                        T* __restrict__ fl, int ldf, int padf,
                                                        T* __restrict__ xb, int ldx, T* __restrict__ col, int ldc, int N, int H,
                                                        int W, int tiles_x, int tiles_y) {
    constexpr int VE = Elem<T>::VE;
    __shared__ T flow[FS_W][FS_W][2];
    __shared__ float cnew[FS_T][FS_T][2];
    const int tid = threadIdx.x;
    int b = blockIdx.x;
    const int tx = b % tiles_x;
    b /= tiles_x;
    const int ty = b % tiles_y;
    const int n = b / tiles_y;
    const int x0 = tx * FS_T, y0 = ty * FS_T;
    const long long img = (long long)n * H * W;
    // phase 1: updated coordinates (-> flow) of the tile + halo; zero flow outside the image (im2col's zero padding)
    for (int i = tid; i < FS_W * FS_W * 2; i += 256) {
        const int c = i & 1, px = (i >> 1) % FS_W, py = (i >> 1) / FS_W;
        const int x = x0 + px - FS_HALO, y = y0 + py - FS_HALO;
        float f = 0.f;
        if ((unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H) {
            const long long pix = img + (long long)y * W + x;
            float v = coords[pix * 2 + c];
            if (P != nullptr) {
                // gvfi_tap_sum: v = bias + sum of the in-image taps (row-major tap order) + res
                float acc = bias ? bias[c] : 0.f;
                int tap = 0;
                for (int kh = 0; kh < 3; ++kh) {
                    const int yy = y + kh - 1;
                    for (int kw = 0; kw < 3; ++kw, ++tap) {
                        const int xx = x + kw - 1;
                        if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
                            acc += P[(pix + (long long)(kh - 1) * W + (kw - 1)) * ldp + tap * 2 + c];
                    }
                }
                v = acc + v;
            }
            const int cx = px - FS_HALO, cy = py - FS_HALO;
            if ((unsigned)cx < (unsigned)FS_T && (unsigned)cy < (unsigned)FS_T) cnew[cy][cx][c] = v;
            f = v - (float)(c == 0 ? x : y);       // gvfi_flow_pack
        }
        Elem<T>::st(&flow[py][px][c], f);
    }
    __syncthreads();
    // phase 2: centre pixels -> coords1 (float), flow activation (2 channels + zero pad), flow slot of the GRU input
    if (tid < FS_T * FS_T) {
        const int cx = tid % FS_T, cy = tid / FS_T;
        const int x = x0 + cx, y = y0 + cy;
        if (x < W && y < H) {
            const long long pix = img + (long long)y * W + x;
            if (P != nullptr) {
                coords_out[pix * 2 + 0] = cnew[cy][cx][0];
                coords_out[pix * 2 + 1] = cnew[cy][cx][1];
            }
            T* d0 = fl + pix * ldf;
            d0[0] = flow[cy + FS_HALO][cx + FS_HALO][0];
            d0[1] = flow[cy + FS_HALO][cx + FS_HALO][1];
            for (int c = 2; c < padf; ++c) Elem<T>::st(d0 + c, 0.f);
            if (xb != nullptr) {
                xb[pix * ldx + 0] = d0[0];
                xb[pix * ldx + 1] = d0[1];
            }
        }
    }
    // phase 3: 7x7x2 patch of every centre pixel, K order (kh, kw, c), zero padded to ldc: one 16-byte vector per step
    const int G = ldc / VE;
    for (int i = tid; i < FS_T * FS_T * G; i += 256) {
        const int g = i % G, cp = i / G;
        const int cx = cp % FS_T, cy = cp / FS_T;
        const int x = x0 + cx, y = y0 + cy;
        if (x >= W || y >= H) continue;
        struct alignas(16) V16 { T e[VE]; } o;
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            const int k = g * VE + e;
            T v;
            Elem<T>::st(&v, 0.f);
            if (k < 98) {
                const int tap = k >> 1, kh = tap / 7, kw = tap - kh * 7;
                v = flow[cy + kh][cx + kw][k & 1];
            }
            o.e[e] = v;
        }
        *(V16*)(col + (img + (long long)y * W + x) * ldc + g * VE) = o;
    }
}
extern "C" int gvfi_flow_step(const float* P, int ldp, const float* bias, const float* coords, float* coords_out, void* fl,
                              int ldf, int padf, void* xb, int ldx, void* col, int ldc, int N, int h, int w, int dtype,
                              void* stream) {
    const int ve = dtype == GVFI_F32 ? 4 : 8;
    if ((ldc % ve) || ldc < 98 || ((uintptr_t)col & 15) || (P != nullptr && ldp < 18) || padf < 2 || padf > ldf) return -2;
    if (P != nullptr && (coords_out == nullptr || coords_out == coords)) return -3;     // the update is not in place
    const int tiles_x = (w + FS_T - 1) / FS_T, tiles_y = (h + FS_T - 1) / FS_T;
    GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_COOP((flow_step_kernel<T>), dim3((unsigned)(N * tiles_x * tiles_y)), dim3(256),
                                            (hipStream_t)stream, P, ldp, bias, coords, coords_out, (T*)fl, ldf, padf, (T*)xb, ldx, (T*)col,
                                            ldc, N, h, w, tiles_x, tiles_y));
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------ convex upsampling   raft/raft.py:86-97
// out[n, 8y+i, 8x+j, c] = sum_k softmax_k(mask[n,y,x, k*64+i*8+j]) * 8*flow[n, y+k/3-1, x+k%3-1, c]
template <typename T>
__global__ void convex_upsample_kernel(const float* __restrict__ coords1, const void* __restrict__ mask, int ldm,
                                       int mask_f32, float* __restrict__ out, long long total, int h, int w) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int W = 8 * w, H = 8 * h;
    const int X = (int)(idx % W);
    const int Y = (int)((idx / W) % H);
    const long long n = idx / ((long long)W * H);
    const int x = X >> 3, j = X & 7, y = Y >> 3, i = Y & 7;
    const long long q = (n * h + y) * (long long)w + x;
    float m[9];
    float mx = -3.0e38f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        m[k] = ld_any<T>(mask, q * ldm + k * 64 + i * 8 + j, mask_f32);
        mx = fmaxf(mx, m[k]);
    }
    float den = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        m[k] = expf(m[k] - mx);
        den += m[k];
    }
    float ox = 0.f, oy = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
        if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;  // unfold zero padding
        const long long qq = (n * h + yy) * (long long)w + xx;
        const float fx = 8.f * (coords1[qq * 2 + 0] - (float)xx);
        const float fy = 8.f * (coords1[qq * 2 + 1] - (float)yy);
        const float wk = m[k] / den;
        ox += wk * fx;
        oy += wk * fy;
    }
    out[idx * 2 + 0] = ox;
    out[idx * 2 + 1] = oy;
}
extern "C" int gvfi_convex_upsample(const float* coords1, const void* mask, int ldm, int mask_f32, float* flow_up,
                                    int N, int h, int w, int dtype, void* stream) {
    const long long total = (long long)N * 64 * h * w;
    GVFI_DISPATCH_T(dtype, GVFI_LAUNCH_SIMPLE((convex_upsample_kernel<T>), grid1d(total), dim3(GVFI_BLOCK),
                                              (hipStream_t)stream, coords1, mask, ldm, mask_f32, flow_up, total, h, w));
    return (int)hipGetLastError();
}
