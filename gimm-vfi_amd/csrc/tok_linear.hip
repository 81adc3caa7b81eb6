// Linear layer over a row matrix of tokens with a SHORT reduction (K = 64 or 128 bf16 input channels, one or two sources):
// y[r][n] = act(sum_k [x0 | x1][r][k] W[n][k] + b[n]) (+ res[r][n]).  FlowFormer's decoder runs ~380 of these per forward on
// 14 336 rows (flow-token encoder, cross-attention q / proj / ffn: decoder.py:84-120,237-255): on the LDS-DMA convolution
// kernel each costs ~10 us for 0.1 GFLOP -- prologue DMA, two K chunks behind barriers, an LDS-staged epilogue.
// Here a wave owns 32 rows and never touches LDS: the 32 x K input block is loaded straight into MFMA operand registers
// (16 bytes per lane and k-step), the weights stream from L2 (rows of W are the MFMA row operand, the next block of 32 output
// channels is prefetched while the current one is multiplied), and the pixel-per-lane accumulator layout gives every lane
// 4 consecutive output channels of its row per register group: bias, activation (none / ReLU / GELU), residual (bf16 or
// float stream) and an 8- or 16-byte store per group.  Same products and the same summation order as the convolution kernel.
#include "conv_mma.h"

struct TokLinArgs {
    const bf16_t *x0, *x1, *w;
    const float* bias;
    const void* res;
    void* y;
    long long rows;
    int ld0, k0, ld1, ldw, N, act, res_f32, ldr, y_f32, ldy;
};

template <int KSTEPS> __global__ void __launch_bounds__(256) tok_linear_kernel(TokLinArgs a) {
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const long long row = (((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 32 + c;
    if (row - c >= a.rows) return;                       // (whole wave)
    const bool rok = row < a.rows;
    uint4 xf[KSTEPS];
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) {
        uint4 z;
        z.x = z.y = z.z = z.w = 0u;
        const int k = kk * 16;
        const bf16_t* src = k < a.k0 ? a.x0 + row * a.ld0 + k + 8 * h : a.x1 + row * a.ld1 + (k - a.k0) + 8 * h;
        xf[kk] = rok ? *(const uint4*)src : z;
    }
    const int nblocks = (a.N + 31) >> 5;
    auto load_w = [&](int nb, uint4 (&wf)[KSTEPS]) {
        const int n = nb * 32 + c;
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            uint4 z;
            z.x = z.y = z.z = z.w = 0u;
            wf[kk] = n < a.N ? *(const uint4*)(a.w + (long long)n * a.ldw + kk * 16 + 8 * h) : z;
        }
    };
    uint4 wcur[KSTEPS], wnext[KSTEPS];
    load_w(0, wcur);
    for (int nb = 0; nb < nblocks; ++nb) {
        if (nb + 1 < nblocks) load_w(nb + 1, wnext);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) acc = mfma_bf16_32x32x16(wcur[kk], xf[kk], acc);
        // lane (row, h): registers 4g..4g+3 = output channels nb*32 + 8g + 4h + (0..3)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n0 = nb * 32 + 8 * g + 4 * h;
            if (n0 < a.N && rok) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = acc[4 * g + e] + (a.bias ? a.bias[n0 + e] : 0.f);
                    v[e] = a.act == GVFI_ACT_GELU ? fast_gelu(t) : (a.act == GVFI_ACT_RELU ? fmaxf(t, 0.f) : t);
                }
                if (a.res != nullptr) {
                    if (a.res_f32) {
                        const float4 r4 = *(const float4*)((const float*)a.res + row * a.ldr + n0);
                        v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
                    } else {
                        const uint2 ru = *(const uint2*)((const bf16_t*)a.res + row * a.ldr + n0);
                        v[0] += __builtin_bit_cast(float, ru.x << 16);
                        v[1] += __builtin_bit_cast(float, ru.x & 0xffff0000u);
                        v[2] += __builtin_bit_cast(float, ru.y << 16);
                        v[3] += __builtin_bit_cast(float, ru.y & 0xffff0000u);
                    }
                }
                if (a.y_f32) {
                    float4 o;
                    o.x = v[0]; o.y = v[1]; o.z = v[2]; o.w = v[3];
                    *(float4*)((float*)a.y + row * a.ldy + n0) = o;
                } else {
                    uint2 u;
                    u.x = pack_bf16x2(v[0], v[1]);
                    u.y = pack_bf16x2(v[2], v[3]);
                    *(uint2*)((bf16_t*)a.y + row * a.ldy + n0) = u;
                }
            }
        }
        if (nb + 1 < nblocks) {
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) wcur[kk] = wnext[kk];
        }
    }
}

// 1 when gvfi_tok_linear takes these arguments: K = k0 + k1 in {64, 128}, sources / weights 16-byte aligned, N % 8 == 0
extern "C" int gvfi_tok_linear_ok(const void* x0, int ld0, int k0, const void* x1, int ld1, int k1, const void* w, int ldw, int N,
                                  int act, const void* res, int res_f32, int ldr, const void* y, int y_f32, int ldy) {
    const int K = k0 + k1;
    if ((K != 64 && K != 128) || k0 <= 0 || (k0 % 16) || (k1 % 16) || N <= 0 || (N % 8) || ldw < K) return 0;
    if (act != GVFI_ACT_NONE && act != GVFI_ACT_RELU && act != GVFI_ACT_GELU) return 0;
    if ((((uintptr_t)x0 | (uintptr_t)w) & 15) || (ld0 % 8) || (ldw % 8) || (k1 > 0 && ((((uintptr_t)x1) & 15) || (ld1 % 8)))) return 0;
    if (((uintptr_t)y & (y_f32 ? 15 : 7)) || (ldy % 4)) return 0;
    if (res != nullptr && (((uintptr_t)res & (res_f32 ? 15 : 7)) || (ldr % 4))) return 0;
    return 1;
}

extern "C" int gvfi_tok_linear(const void* x0, int ld0, int k0, const void* x1, int ld1, int k1, const void* w, int ldw,
                               const float* bias, int N, int act, const void* res, int res_f32, int ldr, void* y, int y_f32,
                               int ldy, long long rows, void* stream) {
    if (!gvfi_tok_linear_ok(x0, ld0, k0, x1, ld1, k1, w, ldw, N, act, res, res_f32, ldr, y, y_f32, ldy)) return -2;
    if (rows <= 0) return 0;
    TokLinArgs a;
    a.x0 = (const bf16_t*)x0; a.x1 = (const bf16_t*)(k1 > 0 ? x1 : x0); a.w = (const bf16_t*)w; a.bias = bias; a.res = res; a.y = y;
    a.rows = rows; a.ld0 = ld0; a.k0 = k0; a.ld1 = k1 > 0 ? ld1 : ld0; a.ldw = ldw; a.N = N; a.act = act; a.res_f32 = res_f32;
    a.ldr = ldr; a.y_f32 = y_f32; a.ldy = ldy;
    const long long waves = (rows + 31) / 32;
    const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    if (k0 + k1 == 64) { GVFI_LAUNCH_COOP(tok_linear_kernel<4>, grid, block, (hipStream_t)stream, a); }
    else { GVFI_LAUNCH_COOP(tok_linear_kernel<8>, grid, block, (hipStream_t)stream, a); }
    return (int)hipGetLastError();
}
