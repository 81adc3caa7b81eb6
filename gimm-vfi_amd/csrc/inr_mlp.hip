// Fused SIREN hypo-network of GIMM (modules/hyponet.py:71-146, config n_layer = 5, hidden 128):
//     [latent(32) | coord(t, y, x)] -> 128 -> 128 -> 128 -> 128 -> 2,  sin between layers, output bias folded.
// Run layer by layer through the convolution engine this path moves 4 x [pixels x 128] activations through HBM
// (~1.6 ms per frame batch at 448x256x8).  Here one wave keeps 64 pixels in registers through all five layers:
//
//   * the GEMMs are computed TRANSPOSED, D[channel][pixel] = W[channel][k] * X^T[k][pixel], so the MFMA result layout
//     (lane = pixel column, 16 registers = 16 channel rows) IS the B-operand layout of the next layer: a lane's
//     registers 8s..8s+7 of the 32-channel block `ob` are its 8 k-values of k-step 2*ob+s.  The k order inside a
//     step is therefore permuted (slot j of lane half h <-> channel 16t + 8(j>>2) + 4h + (j&3)); the weights are
//     packed host-side (gvfi_inr_mlp_pack) with the same permutation, so no activation ever touches LDS or HBM.
//   * the packed weights of all layers (116 KiB of MFMA A-fragments, lane-linear) sit in LDS for the lifetime of a
//     persistent workgroup; each ds_read_b128 fragment feeds two MFMAs (two 32-pixel column blocks per wave).
//   * the INR input concatenation (hyponet.py:91-95) is fused: latent read as bf16 vectors, coordinates as floats.
//
// bf16 activations / weights with fp32 accumulation (the path's bf16 mode); the fp32 validation mode keeps the
// layer-by-layer path (gvfi_conv2d with GVFI_ACT_SIN).
#include "conv_mma.h"

#define INR_IN 35
#define INR_LAT 32
#define INR_HID 128
#define INR_OUT 2
#define INR_FR0 12                 // layer 0: 4 channel blocks x 3 k-steps (K = 35 -> 48)
#define INR_FRH 32                 // hidden layers: 4 channel blocks x 8 k-steps
#define INR_FR4 8                  // last layer: 1 channel block (2 of 32 rows used) x 8 k-steps
#define INR_NFRAG (INR_FR0 + 3 * INR_FRH + INR_FR4)
#define INR_NBIAS (4 * INR_HID + 32)
#define INR_WAVES 8
#define INR_TILE (INR_WAVES * 64)  // pixels per workgroup iteration

#ifndef GVFI_HOSTSIM
__device__ __forceinline__ float inr_sin(float v) { return __sinf(v); }   // v_sin_f32: ample for a bf16 result
#else
static inline float inr_sin(float v) { return sinf(v); }
#endif

// one layer for the wave's two 32-pixel column blocks: acc[cb][ob] = bias + W_frag * bin[cb]
template <int NOB, int NSTEP>
__device__ __forceinline__ void inr_layer(const uint4* __restrict__ wf, const float* __restrict__ bb, int lane,
                                          const uint4 (&bin)[2][8], f32x16 (&acc)[2][4]) {
    const int h = lane >> 5;
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 b4 = *(const float4*)(bb + 32 * ob + 8 * q + 4 * h);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                acc[cb][ob][4 * q + 0] = b4.x;
                acc[cb][ob][4 * q + 1] = b4.y;
                acc[cb][ob][4 * q + 2] = b4.z;
                acc[cb][ob][4 * q + 3] = b4.w;
            }
        }
#pragma unroll
        for (int t = 0; t < NSTEP; ++t) {
            const uint4 a = wf[(ob * NSTEP + t) * 64 + lane];
            Mma2<bf16_t>::run(acc[0][ob], a, bin[0][t]);
            Mma2<bf16_t>::run(acc[1][ob], a, bin[1][t]);
        }
    }
}

// sin + bf16 rounding of the 128-channel result -> the next layer's B operands (register-to-register)
__device__ __forceinline__ void inr_activate(const f32x16 (&acc)[2][4], uint4 (&bin)[2][8]) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int ob = 0; ob < 4; ++ob)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = inr_sin(acc[cb][ob][8 * s + j]);
                uint4 u;
                u.x = pack_bf16x2(v[0], v[1]);
                u.y = pack_bf16x2(v[2], v[3]);
                u.z = pack_bf16x2(v[4], v[5]);
                u.w = pack_bf16x2(v[6], v[7]);
                bin[cb][2 * ob + s] = u;
            }
}

__global__ void __launch_bounds__(INR_TILE) inr_mlp_kernel(const bf16_t* __restrict__ lat, int ldl,
                                                            const float* __restrict__ coord,
                                                            const uint4* __restrict__ wfrag,
                                                            const float* __restrict__ bias, float* __restrict__ out,
                                                            long long npix, long long tiles) {
    __shared__ __attribute__((aligned(16))) uint4 wl[INR_NFRAG * 64];
    __shared__ __attribute__((aligned(16))) float bl[INR_NBIAS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < INR_NFRAG * 64; i += INR_TILE) wl[i] = wfrag[i];
    for (int i = tid; i < INR_NBIAS; i += INR_TILE) bl[i] = bias[i];
    __syncthreads();
    const int col = lane & 31, h = lane >> 5;
    const uint4 zero4 = {0u, 0u, 0u, 0u};
    for (long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const long long p0 = tile * INR_TILE + wave * 64;
        if (p0 >= npix) continue;          // wave-uniform
        uint4 bin[2][8];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const long long pix = p0 + cb * 32 + col;
            const bool ok = pix < npix;
            // layer 0 uses the natural k order: slot j of lane half h <-> input channel 16 s + 8 h + j
            bin[cb][0] = ok ? *(const uint4*)(lat + pix * ldl + 8 * h) : zero4;
            bin[cb][1] = ok ? *(const uint4*)(lat + pix * ldl + 16 + 8 * h) : zero4;
            uint4 c = zero4;
            if (ok && h == 0) {
                c.x = pack_bf16x2(coord[pix * 3 + 0], coord[pix * 3 + 1]);
                c.y = pack_bf16x2(coord[pix * 3 + 2], 0.f);
            }
            bin[cb][2] = c;
        }
        f32x16 acc[2][4];
        inr_layer<4, 3>(wl, bl, lane, bin, acc);
        inr_activate(acc, bin);
#pragma unroll 1
        for (int l = 0; l < 3; ++l) {
            inr_layer<4, 8>(wl + (INR_FR0 + l * INR_FRH) * 64, bl + (l + 1) * INR_HID, lane, bin, acc);
            inr_activate(acc, bin);
        }
        inr_layer<1, 8>(wl + (INR_FR0 + 3 * INR_FRH) * 64, bl + 4 * INR_HID, lane, bin, acc);
        if (h == 0) {   // rows 0, 1 of the last block = registers 0, 1 of lane half 0
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const long long pix = p0 + cb * 32 + col;
                if (pix < npix) *(float2*)(out + pix * 2) = make_float2(acc[cb][0][0], acc[cb][0][1]);
            }
        }
    }
}

// ---- host side: weight image.  w[l] row-major [out][in] (in = 35, 128, 128, 128, 128; out = 128 x4, 2),
// b[l] [out] (the caller folds hyponet's output_bias into b[4]).  wfrag: INR_NFRAG*64*8 bf16, bias: INR_NBIAS floats.
extern "C" int gvfi_inr_mlp_pack(const float* const* w, const float* const* b, void* wfrag_bf16, float* bias) {
    bf16_t* dst = (bf16_t*)wfrag_bf16;
    auto frag = [&](int f, int lane, int j) -> bf16_t& { return dst[((long long)f * 64 + lane) * 8 + j]; };
    for (int ob = 0; ob < 4; ++ob)
        for (int s = 0; s < 3; ++s)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int row = lane & 31, h = lane >> 5, k = 16 * s + 8 * h + j;
                    frag(ob * 3 + s, lane, j) = f2bf(k < INR_IN ? w[0][(32 * ob + row) * INR_IN + k] : 0.f);
                }
    for (int l = 1; l <= 4; ++l) {
        const int nob = l < 4 ? 4 : 1, base = INR_FR0 + (l - 1) * INR_FRH, nout = l < 4 ? INR_HID : INR_OUT;
        for (int ob = 0; ob < nob; ++ob)
            for (int t = 0; t < 8; ++t)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int row = 32 * ob + (lane & 31), h = lane >> 5;
                        const int c = 16 * t + 8 * (j >> 2) + 4 * h + (j & 3);
                        frag(base + ob * 8 + t, lane, j) = f2bf(row < nout ? w[l][row * INR_HID + c] : 0.f);
                    }
    }
    for (int l = 0; l < 4; ++l)
        for (int c = 0; c < INR_HID; ++c) bias[l * INR_HID + c] = b[l][c];
    for (int c = 0; c < 32; ++c) bias[4 * INR_HID + c] = c < INR_OUT ? b[4][c] : 0.f;
    return 0;
}
extern "C" int gvfi_inr_mlp_pack_sizes(int* wfrag_bytes, int* bias_floats) {
    *wfrag_bytes = INR_NFRAG * 64 * 16;
    *bias_floats = INR_NBIAS;
    return 0;
}

extern "C" int gvfi_inr_mlp(const void* lat, int ldl, const float* coord, const void* wfrag, const float* bias,
                            float* out, long long npix, int dtype, void* stream) {
    if (dtype != GVFI_BF16) return -2;   // fp32 validation mode: layer-by-layer gvfi_conv2d
    if ((ldl % 8) || ((uintptr_t)lat & 15) || ((uintptr_t)wfrag & 15) || ((uintptr_t)out & 7)) return -3;
    if (npix <= 0) return 0;
    const long long tiles = (npix + INR_TILE - 1) / INR_TILE;
    const unsigned grid = (unsigned)(tiles < 256 ? tiles : 256);   // one persistent workgroup per CU
    GVFI_LAUNCH_COOP(inr_mlp_kernel, dim3(grid), dim3(INR_TILE), (hipStream_t)stream, (const bf16_t*)lat, ldl, coord,
                     (const uint4*)wfrag, bias, out, npix, tiles);
    return (int)hipGetLastError();
}
