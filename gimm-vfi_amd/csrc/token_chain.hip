// Per-token chains of small linear layers of the FlowFormer decoder (gfx950, 16-bit operand types): three linears of
// <= 128 -> 64 -> 64 -> 64 features with a LayerNorm, a position code, GELUs and residuals in between -- the two halves of
// the flow-token path of MemoryDecoder's iteration (decoder.py:237-255 encode_flow_token + flow_token_encoder,
// decoder.py:84-120 CrossAttentionLayer around its attention).  Unfused they are 5 + 4 launches of 5-10 us with 14 336 rows
// each, 32 times per forward and lane: launch latency, nothing else (skipping the whole token path is worth +8 % of a
// GIMM-VFI-F step, profiles/r3_f_s2d_ab.txt).
//
// One WAVE owns 32 tokens from the first load to the last store; nothing is shared between waves, so there is no barrier.
//   * every linear runs with the WEIGHTS as the row operand of v_mfma_f32_32x32x16: D[feature][token], i.e. a lane holds
//     ONE token (lane & 31) and per 32-feature block 16 features (r & 3) + 8 (r >> 2) + 4 (lane >> 5) -- groups of four
//     consecutive features: bias / activation / residual / LayerNorm run in registers, the result goes to a wave-private
//     LDS tile [32 tokens][64 features] with 8-byte writes, and the next linear reads its operand (8 consecutive features
//     of the lane's token = 16 bytes) straight from there;
//   * weight fragments come from global memory in MFMA order (one coalesced 1 KiB load per operand, L2-resident: 32 KB
//     per chain), the first layer's tokens from global memory as 16-byte loads of their rows;
//   * LayerNorm: a token's 64 features sit in two lanes (l, l ^ 32): 32 register values + one cross-lane add each for
//     the mean and the centred second moment (the two-pass form of gvfi_layernorm);
//   * rounding to the activation type happens exactly where the unfused sequence stores a tensor, so the two paths agree
//     to the accumulation order of the MFMAs.
#include "conv_mma.h"

#ifndef GVFI_HOSTSIM
#define TC_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#else
#define TC_WAVE_SYNC() emu::wave_sync()
#endif

#define TC_ROWB 144      // LDS bytes per token row: 64 features x 2 B + 16 (an odd multiple of 16 bytes)

__device__ __forceinline__ float tc_pos_enc(float px, float py, int c) {
    // LinearPositionEmbeddingSine, dim 64 (attention.py:170-182; the expression of flowformer_ops.hip:pos_enc_channel)
    const int part = c >> 4;
    const float f = (float)(c & 15);
    const float a = 3.14f * (part < 2 ? px : py) * f * (1.0f / 200.0f);
    return (part & 1) ? cosf(a) : sinf(a);
}

template <typename T>
__global__ void __launch_bounds__(256) token_chain_kernel(gvfi_token_chain_params p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * 32 * TC_ROWB];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long wave = (long long)blockIdx.x * 4 + wv;
    const long long t0 = wave * 32;
    if (t0 >= p.rows) return;                       // (whole waves only; no block-level synchronisation below)
    const int n = lane & 31, h = lane >> 5;
    const long long row = t0 + n < p.rows ? t0 + n : p.rows - 1;      // tail lanes recompute the last token, never store
    const bool live = t0 + n < p.rows;
    unsigned char* act = lds + wv * (32 * TC_ROWB);
    const uint4* wf = (const uint4*)p.wfrag;        // [L0: 2 x 8][L1: 2 x 4][L2: 2 x 4] fragments of 64 lanes x 16 bytes
    const int fbase = 4 * h;                        // feature of register r in block mb: 32 mb + (r & 3) + 8 (r >> 2) + fbase

    auto feat = [&](int mb, int r) { return 32 * mb + (r & 3) + 8 * (r >> 2) + fbase; };
    auto round_t = [&](float v) {                  // what a store to the activation type + reload gives
        T t;
        Elem<T>::st(&t, v);
        return Elem<T>::ld(&t);
    };

    f32x16 acc[2];
    float x0[2][16];                                // layer 0's output (after its residual), kept for the final residual
    // ---- layer 0: K = 128 from one or two global sources
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
    {
        uint4 tb[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int k = 16 * kk + 8 * h;
            const T* src = k < p.k0a ? (const T*)p.in0 + row * p.ld0 + k : (const T*)p.in1 + row * p.ld1 + (k - p.k0a);
            tb[kk] = *(const uint4*)src;
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) Mma2<T>::run(acc[mb], wf[(mb * 8 + kk) * 64 + lane], tb[kk]);
    }
    auto ln_posenc = [&](float (&v)[2][16]) {
        // LayerNorm over the token's 64 features (this lane's 32 + lane ^ 32's), then the optional position code;
        // rounded to T after each, as gvfi_layernorm / gvfi_pos_embed store them
        float s = 0.f;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += v[mb][r];
        s += __shfl_xor(s, 32);
        const float mean = s / 64.0f;
        float q2 = 0.f;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) q2 += (v[mb][r] - mean) * (v[mb][r] - mean);
        q2 += __shfl_xor(q2, 32);
        const float rstd = 1.0f / sqrtf(q2 / 64.0f + p.eps);
        float px = 0.f, py = 0.f;
        if (p.coords) {
            const long long cr = row % p.period;
            px = p.coords[cr * 2];
            py = p.coords[cr * 2 + 1];
        }
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int f0 = feat(mb, 4 * g);
                const float4 gm = *(const float4*)(p.ln_g + f0), bt = *(const float4*)(p.ln_b + f0);
                const float gg[4] = {gm.x, gm.y, gm.z, gm.w}, bb[4] = {bt.x, bt.y, bt.z, bt.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float y = round_t((v[mb][4 * g + e] - mean) * rstd * gg[e] + bb[e]);
                    if (p.coords) y = round_t(y + tc_pos_enc(px, py, f0 + e));
                    v[mb][4 * g + e] = y;
                }
            }
    };
    auto to_lds = [&](const float (&v)[2][16]) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 u;
                u.x = pack16x2<T>(v[mb][4 * g], v[mb][4 * g + 1]);
                u.y = pack16x2<T>(v[mb][4 * g + 2], v[mb][4 * g + 3]);
                *(uint2*)(act + n * TC_ROWB + feat(mb, 4 * g) * 2) = u;
            }
        TC_WAVE_SYNC();
    };
    auto to_global = [&](const float (&v)[2][16], void* dst, int ld) {
        if (!live) return;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 u;
                u.x = pack16x2<T>(v[mb][4 * g], v[mb][4 * g + 1]);
                u.y = pack16x2<T>(v[mb][4 * g + 2], v[mb][4 * g + 3]);
                *(uint2*)((T*)dst + row * ld + feat(mb, 4 * g)) = u;
            }
    };
    auto linear64 = [&](int frag0) {               // acc = W (fragments frag0 ..) x the LDS tile
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
        uint4 tb[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) tb[kk] = *(const uint4*)(act + n * TC_ROWB + (16 * kk + 8 * h) * 2);
        TC_WAVE_SYNC();                             // (every lane has read its operands before the tile is rewritten)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) Mma2<T>::run(acc[mb], wf[frag0 + (mb * 4 + kk) * 64 + lane], tb[kk]);
    };
    auto epilogue = [&](int layer, int act_kind, float (&v)[2][16]) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b4 = *(const float4*)(p.bias + layer * 64 + feat(mb, 4 * g));
                const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = acc[mb][4 * g + e] + bb[e];
                    v[mb][4 * g + e] = act_kind == GVFI_ACT_GELU ? fast_gelu(t) : t;
                }
            }
    };

    float v[2][16];
    epilogue(0, p.act0, v);
    if (p.res0) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const uint2 u = *(const uint2*)((const T*)p.res0 + row * p.ldr0 + feat(mb, 4 * g));
                v[mb][4 * g] += cvt16<T>(u.x & 0xffffu);
                v[mb][4 * g + 1] += cvt16<T>(u.x >> 16);
                v[mb][4 * g + 2] += cvt16<T>(u.y & 0xffffu);
                v[mb][4 * g + 3] += cvt16<T>(u.y >> 16);
            }
    }
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) x0[mb][r] = v[mb][r] = round_t(v[mb][r]);
    if (p.ln_after == 0) ln_posenc(v);
    to_lds(v);
    // ---- layer 1
    linear64(2 * 8 * 64);
    epilogue(1, p.act1, v);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[mb][r] = round_t(v[mb][r]);
    if (p.out1) to_global(v, p.out1, p.ldo1);
    if (p.ln_after == 1) ln_posenc(v);
    to_lds(v);
    // ---- layer 2
    linear64(2 * 8 * 64 + 2 * 4 * 64);
    epilogue(2, GVFI_ACT_NONE, v);
    if (p.res2_from0) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) v[mb][r] += x0[mb][r];
    }
    to_global(v, p.out2, p.ldo2);
}

extern "C" int gvfi_token_chain(const gvfi_token_chain_params* pp, void* stream) {
    const gvfi_token_chain_params& p = *pp;
    if (p.dtype != GVFI_BF16 && p.dtype != GVFI_F16) return -2;
    if (p.rows <= 0 || p.in0 == nullptr || p.wfrag == nullptr || p.bias == nullptr || p.out2 == nullptr) return -2;
    if (p.k0a <= 0 || p.k0a > 128 || (p.k0a & 7) || (p.k0a < 128 && p.in1 == nullptr)) return -2;
    if ((p.ln_after != 0 && p.ln_after != 1) || p.ln_g == nullptr || p.ln_b == nullptr) return -2;
    if (p.coords != nullptr && p.period <= 0) return -2;
    auto al = [](const void* q, int ld) { return q == nullptr || ((((uintptr_t)q) & 15) == 0 && ((ld * 2) & 15) == 0); };
    auto al8 = [](const void* q, int ld) { return q == nullptr || ((((uintptr_t)q) & 7) == 0 && ((ld * 2) & 7) == 0); };
    if (!al(p.in0, p.ld0) || !al(p.in1, p.ld1) || (((uintptr_t)p.wfrag) & 15)) return -3;
    if (!al8(p.res0, p.ldr0) || !al8(p.out1, p.ldo1) || !al8(p.out2, p.ldo2)) return -3;
    if ((((uintptr_t)p.bias) | ((uintptr_t)p.ln_g) | ((uintptr_t)p.ln_b)) & 15) return -3;
    const long long waves = (p.rows + 31) / 32;
    const int grid = (int)((waves + 3) / 4);
    if (p.dtype == GVFI_F16) {
        GVFI_LAUNCH_COOP((token_chain_kernel<f16_t>), dim3(grid), dim3(256), (hipStream_t)stream, p);
    } else {
        GVFI_LAUNCH_COOP((token_chain_kernel<bf16_t>), dim3(grid), dim3(256), (hipStream_t)stream, p);
    }
    return (int)hipGetLastError();
}
