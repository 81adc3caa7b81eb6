// The whole flow-token path of one MemoryDecoder iteration as ONE launch (gfx950, 16-bit operand types):
//   81-tap cost look-up (decoder.py:237-255)  ->  flow_token_encoder.0 GELU, .2 (= query), norm1 + position code, q
//   ->  the query's cross-attention over its cost map's 8 latent tokens (decoder.py:35-120, 8 heads of 8)
//   ->  proj([attention | query]) + query, norm2, ffn.0 GELU, ffn.3 + x  ->  cost_global.
// Until round 4 these were four launches per iteration and lane (gvfi_cost_lookup 10.5 us, gvfi_token_chain 20.9,
// gvfi_attn_global 8.8, gvfi_token_chain 20.9: 256 launches of pure latency per forward on 14 336 rows each).  Every step is
// per token, so one WAVE owns 32 tokens from the look-up to the last store, exactly as in token_chain.hip (same MFMA
// operand roles, same fragment-ordered weights, same wave-private LDS tiles, no barrier); the tensors that used to travel
// through HBM between the launches (look-up taps as the first linear's operand, q, the attention output, query) stay in
// LDS / registers.  Rounding to the activation type happens exactly where the separate launches store a tensor, and every
// float expression is theirs (cost_lookup_kernel, token_chain_kernel, AttnAcc of flowformer_ops.hip): bit-identical results.
#include "conv_mma.h"
#include "gimmvfi_experiments.h"

#ifndef GVFI_HOSTSIM
#define TP_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#else
#define TP_WAVE_SYNC() emu::wave_sync()
#endif

#define TP_ROWB 144      // LDS bytes per token row of a 64-feature tile: 128 + 16 (an odd multiple of 16 bytes)
#define TP_ROWB0 272     // ... of the 128-feature operand tile of the first linear: 256 + 16
#define TP_WAVE_LDS (32 * TP_ROWB0 + 32 * TP_ROWB)      // per wave: [operand tile | query tile]; the 64-feature work tile aliases the operand tile

__device__ __forceinline__ float tp_pos_enc(float px, float py, int c) {
    // LinearPositionEmbeddingSine, dim 64 (attention.py:170-182) -- the expression of token_chain.hip:tc_pos_enc
    const int part = c >> 4;
    const float f = (float)(c & 15);
    const float a = 3.14f * (part < 2 ? px : py) * f * (1.0f / 200.0f);
    return (part & 1) ? cosf(a) : sinf(a);
}

template <typename T>
__global__ void __launch_bounds__(256) token_path_kernel(gvfi_token_path_params p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * TP_WAVE_LDS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long wave = (long long)blockIdx.x * 4 + wv;
    const long long t0 = wave * 32;
    if (t0 >= p.rows) return;                       // (whole waves only; no block-level synchronisation below)
    const int n = lane & 31, h = lane >> 5;
    const long long row = t0 + n < p.rows ? t0 + n : p.rows - 1;      // tail lanes recompute the last token, never store
    const bool live = t0 + n < p.rows;
    unsigned char* big = lds + wv * TP_WAVE_LDS;    // [32][128] operand of the first linear, later the [32][64] work tile
    unsigned char* act = big;
    unsigned char* qt = big + 32 * TP_ROWB0;        // [32][64] query (second operand half of the third chain's first linear)
    const int fbase = 4 * h;
    auto feat = [&](int mb, int r) { return 32 * mb + (r & 3) + 8 * (r >> 2) + fbase; };
    auto round_t = [&](float v) {
        T t;
        Elem<T>::st(&t, v);
        return Elem<T>::ld(&t);
    };

    // ---------------------------------------------------------------- 1. look-up: 81 taps of each token's own cost map
    {
        // lanes 0-31: even taps, lanes 32-63: odd taps, of token n; TU taps per lane in flight at once -- the four map reads of
        // a tap are UNCONDITIONAL loads at clamped positions (a wave owns its tokens from here to the last store, so nothing
        // else hides a dependent load -> use chain), the reference's conditions only select what is added
        const int win = 2 * p.radius + 1, ntap = win * win;
        const float* base = p.maps + row * (long long)p.h * p.w;
        const float qx = p.coords[row * 2 + 0], qy = p.coords[row * 2 + 1];
        constexpr int TU = 14;
#pragma unroll 1
        for (int it0 = 0; it0 * 2 < ntap + 1; it0 += TU) {
            float ax[TU], ay[TU], t00[TU], t01[TU], t10[TU], t11[TU];
            unsigned ok[TU];
#pragma unroll
            for (int u = 0; u < TU; ++u) {
                const int tap = 2 * (it0 + u) + h;
                const int tc = tap < ntap ? tap : 0;
                const int i = tc / win, j = tc - i * win;
                const float cx = qx + (float)(i - p.radius);
                const float cy = qy + (float)(j - p.radius);
                const float xn = 2.f * cx / (float)(p.w - 1) - 1.f;
                const float yn = 2.f * cy / (float)(p.h - 1) - 1.f;
                const float ix = ((xn + 1.f) * 0.5f) * (float)(p.w - 1);
                const float iy = ((yn + 1.f) * 0.5f) * (float)(p.h - 1);
                // (far outside the map every tap is absent; clamping first keeps the int conversion defined)
                const float x0f = floorf(fminf(fmaxf(ix, -4.f), (float)p.w + 4.f)), y0f = floorf(fminf(fmaxf(iy, -4.f), (float)p.h + 4.f));
                const float x0e = floorf(ix), y0e = floorf(iy);
                const int x0 = (int)x0f, y0 = (int)y0f;
                ax[u] = ix - x0e;
                ay[u] = iy - y0e;
                const bool xin0 = x0 >= 0 && x0 < p.w, xin1 = x0 + 1 >= 0 && x0 + 1 < p.w;
                const bool yin0 = y0 >= 0 && y0 < p.h, yin1 = y0 + 1 >= 0 && y0 + 1 < p.h;
                ok[u] = (xin0 && yin0 ? 1u : 0u) | (xin1 && yin0 ? 2u : 0u) | (xin0 && yin1 ? 4u : 0u) | (xin1 && yin1 ? 8u : 0u);
                const int xa = x0 < 0 ? 0 : (x0 > p.w - 1 ? p.w - 1 : x0), xb = x0 + 1 < 0 ? 0 : (x0 + 1 > p.w - 1 ? p.w - 1 : x0 + 1);
                const int ya = y0 < 0 ? 0 : (y0 > p.h - 1 ? p.h - 1 : y0), yb = y0 + 1 < 0 ? 0 : (y0 + 1 > p.h - 1 ? p.h - 1 : y0 + 1);
                t00[u] = base[(long long)ya * p.w + xa];
                t01[u] = base[(long long)ya * p.w + xb];
                t10[u] = base[(long long)yb * p.w + xa];
                t11[u] = base[(long long)yb * p.w + xb];
            }
#pragma unroll
            for (int u = 0; u < TU; ++u) {
                const int tap = 2 * (it0 + u) + h;
                float v = 0.f;
                if (ok[u] & 1u) v += (1.f - ax[u]) * (1.f - ay[u]) * t00[u];
                if (ok[u] & 2u) v += ax[u] * (1.f - ay[u]) * t01[u];
                if (ok[u] & 4u) v += (1.f - ax[u]) * ay[u] * t10[u];
                if (ok[u] & 8u) v += ax[u] * ay[u] * t11[u];
                if (ax[u] != ax[u] || ay[u] != ay[u]) v = ax[u] + ay[u];     // NaN coordinates stay visible (the clamped cell hid them)
                if (tap < ntap) {
                    T tv;
                    Elem<T>::st(&tv, v);
                    *(T*)(big + n * TP_ROWB0 + tap * 2) = tv;
                    if (live) ((T*)p.taps_out)[row * p.ldt + tap] = tv;      // cost_forward: an input of the update block
                }
            }
        }
        // features ntap .. 127 of the first linear's operand are zeros (the channel padding of the cost tensor)
        for (int f = ntap + h; f < 128; f += 2) {
            T z;
            Elem<T>::st(&z, 0.f);
            *(T*)(big + n * TP_ROWB0 + f * 2) = z;
        }
        TP_WAVE_SYNC();
    }

    // ---------------------------------------------------------------- the chain machinery of token_chain.hip
    f32x16 acc[2];
    const gvfi_token_chain_params* cp = &p.a;
    const uint4* wf = (const uint4*)cp->wfrag;
    auto zero_acc = [&]() {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
    };
    auto ln_posenc = [&](float (&v)[2][16], bool with_pos) {
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
        const float rstd = 1.0f / sqrtf(q2 / 64.0f + cp->eps);
        float px = 0.f, py = 0.f;
        if (with_pos) {
            px = p.coords[row * 2];
            py = p.coords[row * 2 + 1];
        }
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int f0 = feat(mb, 4 * g);
                const float4 gm = *(const float4*)(cp->ln_g + f0), bt = *(const float4*)(cp->ln_b + f0);
                const float gg[4] = {gm.x, gm.y, gm.z, gm.w}, bb[4] = {bt.x, bt.y, bt.z, bt.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float y = round_t((v[mb][4 * g + e] - mean) * rstd * gg[e] + bb[e]);
                    if (with_pos) y = round_t(y + tp_pos_enc(px, py, f0 + e));
                    v[mb][4 * g + e] = y;
                }
            }
    };
    auto to_tile = [&](const float (&v)[2][16], unsigned char* tile) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 u;
                u.x = pack16x2<T>(v[mb][4 * g], v[mb][4 * g + 1]);
                u.y = pack16x2<T>(v[mb][4 * g + 2], v[mb][4 * g + 3]);
                *(uint2*)(tile + n * TP_ROWB + feat(mb, 4 * g) * 2) = u;
            }
        TP_WAVE_SYNC();
    };
    auto linear64 = [&](int frag0) {               // acc = W (fragments frag0 ..) x the work tile
        zero_acc();
        uint4 tb[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) tb[kk] = *(const uint4*)(act + n * TP_ROWB + (16 * kk + 8 * h) * 2);
        TP_WAVE_SYNC();                             // (every lane has read its operands before the tile is rewritten)
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
                const float4 b4 = *(const float4*)(cp->bias + layer * 64 + feat(mb, 4 * g));
                const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = acc[mb][4 * g + e] + bb[e];
                    v[mb][4 * g + e] = act_kind == GVFI_ACT_GELU ? fast_gelu(t) : t;
                }
            }
    };
    auto round_all = [&](float (&v)[2][16]) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) v[mb][r] = round_t(v[mb][r]);
    };

    // ---------------------------------------------------------------- 2. first chain: taps -> query, q
    float v[2][16], query[2][16];
    {
        zero_acc();
        uint4 tb[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) tb[kk] = *(const uint4*)(big + n * TP_ROWB0 + (16 * kk + 8 * h) * 2);
        TP_WAVE_SYNC();                             // the operand tile is free: the work tile aliases it from here on
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) Mma2<T>::run(acc[mb], wf[(mb * 8 + kk) * 64 + lane], tb[kk]);
    }
    epilogue(0, cp->act0, v);
    round_all(v);
    to_tile(v, act);
    linear64(2 * 8 * 64);
    epilogue(1, cp->act1, v);
    round_all(v);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) query[mb][r] = v[mb][r];          // = out1 of the separate launch (already rounded)
    to_tile(v, qt);
    ln_posenc(v, true);
    to_tile(v, act);
    linear64(2 * 8 * 64 + 2 * 4 * 64);
    epilogue(2, GVFI_ACT_NONE, v);
    to_tile(v, act);                                // q, rounded as its store to the activation type would

    // ---------------------------------------------------------------- 3. cross-attention over the map's latent tokens
    {
        // lane (n, h) owns heads 4h .. 4h+3 of token n: the same one-pass soft-max as AttnAcc<T, 8> (flowformer_ops.hip)
        const long long img = row / p.P, pp = row - img * p.P;
        const T* kvb = (const T*)p.kv + (img * p.K * p.P + pp) * (long long)p.ldkv;
#pragma unroll 1
        for (int hh = 0; hh < 4; ++hh) {
            const int hd = 4 * h + hh;
            float q8[8], o8[8], m = -INFINITY, l = 0.f;
            const uint4 qv = *(const uint4*)(act + n * TP_ROWB + hd * 16);
            const T* qe = (const T*)&qv;
#pragma unroll
            for (int d = 0; d < 8; ++d) { q8[d] = Elem<T>::ld(qe + d); o8[d] = 0.f; }
            // all key / value rows of the head in flight at once (a wave has nothing else to hide 8 dependent round trips behind)
            constexpr int KB = 8;
            for (int j0 = 0; j0 < p.K; j0 += KB) {
                uint4 kq[KB], vq[KB];
#pragma unroll
                for (int u = 0; u < KB; ++u) {
                    const int j = j0 + u < p.K ? j0 + u : p.K - 1;
                    const T* kr = kvb + (long long)j * p.P * p.ldkv;
                    kq[u] = *(const uint4*)(kr + hd * 8);
                    vq[u] = *(const uint4*)(kr + 64 + hd * 8);
                }
#pragma unroll
                for (int u = 0; u < KB; ++u) {
                    if (j0 + u >= p.K) break;
                    const T* ke = (const T*)&kq[u];
                    const T* ve = (const T*)&vq[u];
                    float s = 0.f;
#pragma unroll
                    for (int d = 0; d < 8; ++d) s += q8[d] * Elem<T>::ld(ke + d);
                    s *= p.scale;
                    const float mn = fmaxf(m, s);
                    const float a = expf(m - mn), pe = expf(s - mn);
                    l = l * a + pe;
#pragma unroll
                    for (int d = 0; d < 8; ++d) o8[d] = o8[d] * a + pe * Elem<T>::ld(ve + d);
                    m = mn;
                }
            }
            const float inv = 1.0f / l;
            __attribute__((aligned(16))) T ov[8];
#pragma unroll
            for (int d = 0; d < 8; ++d) Elem<T>::st(ov + d, o8[d] * inv);
            *(uint4*)(act + n * TP_ROWB + hd * 16) = *(const uint4*)ov;       // in place: only this lane reads this head's q
        }
        TP_WAVE_SYNC();
    }

    // ---------------------------------------------------------------- 4. second chain: [attention | query] -> cost_global
    cp = &p.c;
    wf = (const uint4*)cp->wfrag;
    float x0[2][16];
    {
        zero_acc();
        uint4 tb[8];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) tb[kk] = *(const uint4*)(act + n * TP_ROWB + (16 * kk + 8 * h) * 2);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) tb[4 + kk] = *(const uint4*)(qt + n * TP_ROWB + (16 * kk + 8 * h) * 2);
        TP_WAVE_SYNC();
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) Mma2<T>::run(acc[mb], wf[(mb * 8 + kk) * 64 + lane], tb[kk]);
    }
    epilogue(0, cp->act0, v);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) x0[mb][r] = v[mb][r] = round_t(v[mb][r] + query[mb][r]);      // res0 = query
    ln_posenc(v, false);
    to_tile(v, act);
    linear64(2 * 8 * 64);
    epilogue(1, cp->act1, v);
    round_all(v);
    to_tile(v, act);
    linear64(2 * 8 * 64 + 2 * 4 * 64);
    epilogue(2, GVFI_ACT_NONE, v);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[mb][r] += x0[mb][r];
    if (live) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 u;
                u.x = pack16x2<T>(v[mb][4 * g], v[mb][4 * g + 1]);
                u.y = pack16x2<T>(v[mb][4 * g + 2], v[mb][4 * g + 3]);
                *(uint2*)((T*)p.out + row * p.ldo + feat(mb, 4 * g)) = u;
            }
    }
}

extern "C" int gvfi_token_path(const gvfi_token_path_params* pp, void* stream) {
    const gvfi_token_path_params& p = *pp;
    if (p.dtype != GVFI_BF16 && p.dtype != GVFI_F16) return -2;
    if (p.rows <= 0 || p.maps == nullptr || p.coords == nullptr || p.taps_out == nullptr || p.kv == nullptr || p.out == nullptr) return -2;
    if (p.h < 2 || p.w < 2 || p.radius != 4 || p.K <= 0 || p.P <= 0 || (p.rows % p.P) != 0) return -2;     // 81 taps <= 128 features
    if (p.ldt < 81) return -2;
    for (const gvfi_token_chain_params* c : {&p.a, &p.c}) {
        if (c->wfrag == nullptr || c->bias == nullptr || c->ln_g == nullptr || c->ln_b == nullptr) return -2;
        if ((((uintptr_t)c->wfrag) | ((uintptr_t)c->bias) | ((uintptr_t)c->ln_g) | ((uintptr_t)c->ln_b)) & 15) return -3;
    }
    // the arrangement the decoder uses (engine_f.py): chain A = [GELU linear, linear (= query), LayerNorm + position code, linear],
    // chain C = [linear + query, LayerNorm, GELU linear, linear + x]
    if (p.a.ln_after != 1 || p.c.ln_after != 0 || p.c.res2_from0 != 1 || p.a.res2_from0 != 0) return -2;
    if ((((uintptr_t)p.kv) & 15) || ((p.ldkv * 2) & 15) || p.ldkv < 128 || (((uintptr_t)p.out) & 7) || ((p.ldo * 2) & 7)) return -3;
    const long long waves = (p.rows + 31) / 32;
    const int grid = (int)((waves + 3) / 4);
    if (p.dtype == GVFI_F16) {
        GVFI_LAUNCH_COOP((token_path_kernel<f16_t>), dim3(grid), dim3(256), (hipStream_t)stream, p);
    } else {
        GVFI_LAUNCH_COOP((token_path_kernel<bf16_t>), dim3(grid), dim3(256), (hipStream_t)stream, p);
    }
    return (int)hipGetLastError();
}
