// MFMA attention for the small-key-set attentions of FlowFormer's Twins / cost encoders (bf16 or -- round 5, the "enc:f16"
// stage of GIMM-VFI-F's precision policy -- IEEE half operands: same layouts, the f16 MFMA; head dimension 16 or 32):
//   * gvfi_attn_global_mfma: NQ queries against M <= 128 keys per group (globally sub-sampled attention twins.py:870-925,
//     430-546: M = 112 at 448x256; any row layout gvfi_attn_global addresses);
//   * gvfi_attn_window_mfma: the 7x7 locally grouped attention (twins.py:814-867, 331-427): 49 queries x 49 keys per window,
//     out-of-grid window positions from the pad tables.
// The scalar kernels of flowformer_ops.hip (one thread per (query, head), a serial loop over the keys) spend 6.5 of the
// 50 ms of a GIMM-VFI-F forward here.  One WAVE owns a (group, head): the key fragments K[key][d] (MFMA row operand) and the
// transposed value fragments V^T[d][key] stay in registers while the wave walks its blocks of 32 queries:
//   S^T[key][query] = K Q^T      (32x32x16 MFMAs; a lane holds ONE query and, per key block, 16 of its 32 keys)
//   soft-max over the keys        (registers + one exchange with lane ^ 32, exp2 with the scale folded in)
//   O^T[d][query]   = V^T P^T    (P^T packed to bf16 straight from the S^T registers: the key order of a k-step is a fixed
//                                  permutation, applied once when V^T is gathered)
// No LDS, no barriers.  The float validation mode keeps the scalar kernels.
#include "conv_mma.h"

struct AttnGArgs {
    const bf16_t *q, *k, *v;
    bf16_t* o;
    int ldq, ldk, ldv, ldo;
    long long qb1, qb0, qs, kb1, kb0, ks, ob1, ob0, os, G1;
    int G0, NQ, M, heads, qb_per_task, nchunk;
    float scale_log2e;
};
struct AttnWArgs {
    const bf16_t *q, *k, *v;
    const float *kpad, *vpad;
    bf16_t* o;
    int ldq, ldk, ldv, ldo, n_img, H, W, heads, nwx, nwy;
    float scale_log2e;
};

__device__ __forceinline__ float at_exp2(float x) {
#ifndef GVFI_HOSTSIM
    return __builtin_amdgcn_exp2f(x);
#else
    return exp2f(x);
#endif
}
__device__ __forceinline__ uint4 at_zero4() {
    uint4 z;
    z.x = z.y = z.z = z.w = 0u;
    return z;
}
template <typename T> __device__ __forceinline__ uint4 at_pack8(const float* f) {
    uint4 u;
    u.x = pack16x2<T>(f[0], f[1]);
    u.y = pack16x2<T>(f[2], f[3]);
    u.z = pack16x2<T>(f[4], f[5]);
    u.w = pack16x2<T>(f[6], f[7]);
    return u;
}
// key (within a block of 32) of accumulator register r in lane half h / of slot e of k-step t (16 keys) in lane half h
__device__ __forceinline__ int at_key_of_reg(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// soft-max over the keys of one query block + P V; NKB key blocks of 32 (compile-time bound, nkb of them live)
template <typename T, int HD, int NKB>
__device__ __forceinline__ void attn_block(const uint4 (&kf)[NKB][HD / 16], const uint4 (&vf)[2 * NKB], const uint4 (&qf)[HD / 16],
                                           int nkb, int M, int h, float scale_log2e, f32x16& o, float& inv_sum) {
    f32x16 s[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
        if (kb < nkb) {
#pragma unroll
            for (int ks = 0; ks < HD / 16; ++ks) Mma2<T>::run(s[kb], kf[kb][ks], qf[ks]);
        }
    }
    float m = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kb * 32 + at_key_of_reg(r, h);
            s[kb][r] = key < M ? s[kb][r] * scale_log2e : -INFINITY;
            m = fmaxf(m, s[kb][r]);
        }
    m = fmaxf(m, __shfl_xor(m, 32));
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[kb][r] = at_exp2(s[kb][r] - m);
            sum += s[kb][r];
        }
    sum += __shfl_xor(sum, 32);
    inv_sum = 1.0f / sum;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        if (kb < nkb) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float pv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) pv[e] = s[kb][8 * t + e];
                Mma2<T>::run(o, vf[2 * kb + t], at_pack8<T>(pv));
            }
        }
    }
}
// O^T registers of a lane (one query; head channels (r&3) + 8*(r>>2) + 4h) -> HD/8 stores of 4 channels
template <typename T, int HD> __device__ __forceinline__ void attn_store(const f32x16& o, float inv_sum, bf16_t* op, int h) {
#pragma unroll
    for (int g = 0; g < HD / 8; ++g) {
        uint2 u;
        u.x = pack16x2<T>(o[4 * g + 0] * inv_sum, o[4 * g + 1] * inv_sum);
        u.y = pack16x2<T>(o[4 * g + 2] * inv_sum, o[4 * g + 3] * inv_sum);
        *(uint2*)(op + 8 * g + 4 * h) = u;
    }
}

// (T: the 16-bit operand type -- bf16_t or f16_t; the argument blocks carry raw 16-bit pointers either way)
template <typename T, int HD> __global__ void __launch_bounds__(256) attn_global_mfma_kernel(AttnGArgs a) {
    constexpr int NKB = 4, KS = HD / 16;
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const long long task = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int chunk = (int)(task % a.nchunk);
    const int head = (int)((task / a.nchunk) % a.heads);
    const long long g = task / ((long long)a.nchunk * a.heads);
    if (g >= a.G1 * a.G0) return;                       // (whole wave)
    const long long g0 = g % a.G0, g1 = g / a.G0;
    const int nkb = (a.M + 31) >> 5;
    const long long kbase = g1 * a.kb1 + g0 * a.kb0;
    uint4 kf[NKB][KS], vf[2 * NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        const int key = kb * 32 + c;
        const bf16_t* kp = a.k + (kbase + (long long)key * a.ks) * a.ldk + head * HD + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kf[kb][ks] = key < a.M ? *(const uint4*)(kp + ks * 16) : at_zero4();
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            unsigned short e8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int vk = kb * 32 + 16 * t + at_key_of_reg(e, h);
                e8[e] = (vk < a.M && c < HD) ? a.v[(kbase + (long long)vk * a.ks) * a.ldv + head * HD + c] : (unsigned short)0;
            }
            uint4 u;
            u.x = e8[0] | ((unsigned)e8[1] << 16); u.y = e8[2] | ((unsigned)e8[3] << 16);
            u.z = e8[4] | ((unsigned)e8[5] << 16); u.w = e8[6] | ((unsigned)e8[7] << 16);
            vf[2 * kb + t] = u;
        }
    }
    const int nqb = (a.NQ + 31) >> 5;
    const int qb_end = (chunk + 1) * a.qb_per_task < nqb ? (chunk + 1) * a.qb_per_task : nqb;
    for (int qb = chunk * a.qb_per_task; qb < qb_end; ++qb) {
        const int i = qb * 32 + c;
        const bool valid = i < a.NQ;
        const bf16_t* qp = a.q + (g1 * a.qb1 + g0 * a.qb0 + (long long)i * a.qs) * a.ldq + head * HD + 8 * h;
        uint4 qf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = valid ? *(const uint4*)(qp + ks * 16) : at_zero4();
        f32x16 o;
        float inv;
        attn_block<T, HD, NKB>(kf, vf, qf, nkb, a.M, h, a.scale_log2e, o, inv);
        if (valid) attn_store<T, HD>(o, inv, a.o + (g1 * a.ob1 + g0 * a.ob0 + (long long)i * a.os) * a.ldo + head * HD, h);
    }
}

template <typename T, int HD> __global__ void __launch_bounds__(256) attn_window_mfma_kernel(AttnWArgs a) {
    constexpr int NKB = 2, KS = HD / 16, WS = 7, NK = WS * WS;
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const long long task = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int head = (int)(task % a.heads);
    long long r = task / a.heads;
    const int wx = (int)(r % a.nwx);
    r /= a.nwx;
    const int wy = (int)(r % a.nwy);
    const long long img = r / a.nwy;
    if (img >= a.n_img) return;                         // (whole wave)
    const int C = a.heads * HD;
    const long long img0 = img * a.H * a.W;
    // token row of window position pos (or -1: outside the grid -> pad tables)
    auto row_of = [&](int pos) -> long long {
        const int yy = wy * WS + pos / WS, xx = wx * WS + pos % WS;
        return (yy < a.H && xx < a.W) ? img0 + (long long)yy * a.W + xx : -1;
    };
    uint4 kf[NKB][KS], vf[2 * NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        const int pos = kb * 32 + c;
        const long long kr = pos < NK ? row_of(pos) : -1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (pos >= NK) {
                kf[kb][ks] = at_zero4();
            } else if (kr >= 0) {
                kf[kb][ks] = *(const uint4*)(a.k + kr * a.ldk + head * HD + ks * 16 + 8 * h);
            } else {
                float f[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = a.kpad[pos * C + head * HD + ks * 16 + 8 * h + e];
                kf[kb][ks] = at_pack8<T>(f);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int vp = kb * 32 + 16 * t + at_key_of_reg(e, h);
                f[e] = 0.f;
                if (vp < NK && c < HD) {
                    const long long vr = row_of(vp);
                    f[e] = vr >= 0 ? cvt16<T>(a.v[vr * a.ldv + head * HD + c]) : a.vpad[vp * C + head * HD + c];
                }
            }
            vf[2 * kb + t] = at_pack8<T>(f);
        }
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int pos = qb * 32 + c;
        const long long qr = pos < NK ? row_of(pos) : -1;
        uint4 qf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = qr >= 0 ? *(const uint4*)(a.q + qr * a.ldq + head * HD + ks * 16 + 8 * h) : at_zero4();
        f32x16 o;
        float inv;
        attn_block<T, HD, NKB>(kf, vf, qf, NKB, NK, h, a.scale_log2e, o, inv);
        if (qr >= 0) attn_store<T, HD>(o, inv, a.o + qr * a.ldo + head * HD, h);
    }
}

// 1 when gvfi_attn_global / gvfi_attn_window take the MFMA kernels for these arguments (bf16 / IEEE half)
extern "C" int gvfi_attn_mfma_ok(int window, int M, int NQ, int head_dim, int ws, int dtype) {
    if ((dtype != GVFI_BF16 && dtype != GVFI_F16) || (head_dim != 16 && head_dim != 32)) return 0;
    if (window) return ws == 7;
    return M >= 9 && M <= 128 && NQ >= 16;
}

static int attn_global_mfma_t(int dtype, const void* q, int ldq, long long qb1, long long qb0, long long qs, const void* k, int ldk,
                              const void* v, int ldv, long long kb1, long long kb0, long long ks, void* out, int ldo,
                              long long ob1, long long ob0, long long os, long long G1, int G0, int NQ, int M, int heads,
                              int head_dim, float scale, void* stream) {
    if (!gvfi_attn_mfma_ok(0, M, NQ, head_dim, 0, dtype) || G0 <= 0 || G1 <= 0) return -2;
    if ((((uintptr_t)q | (uintptr_t)k) & 15) || (((uintptr_t)out) & 7) || (ldq % 8) || (ldk % 8) || (ldo % 4)) return -3;
    AttnGArgs a;
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (bf16_t*)out;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
    a.qb1 = qb1; a.qb0 = qb0; a.qs = qs; a.kb1 = kb1; a.kb0 = kb0; a.ks = ks; a.ob1 = ob1; a.ob0 = ob0; a.os = os;
    a.G1 = G1; a.G0 = G0; a.NQ = NQ; a.M = M; a.heads = heads;
    a.scale_log2e = scale * 1.44269504088896341f;
    const int nqb = (NQ + 31) / 32;
    // enough wave tasks to fill the chip (>= ~8 per SIMD), but as many query blocks per task as that allows: the key /
    // value fragments are gathered once per task
    const long long groups = G1 * G0 * heads;
    long long per = (groups * nqb) / 8192;
    per = per < 1 ? 1 : (per > 16 ? 16 : per);
    a.qb_per_task = (int)per;
    a.nchunk = (nqb + a.qb_per_task - 1) / a.qb_per_task;
    const long long tasks = groups * a.nchunk;
    const dim3 grid((unsigned)((tasks + 3) / 4)), block(256);
    if (dtype == GVFI_F16) {
        if (head_dim == 16) { GVFI_LAUNCH_COOP((attn_global_mfma_kernel<f16_t, 16>), grid, block, (hipStream_t)stream, a); }
        else { GVFI_LAUNCH_COOP((attn_global_mfma_kernel<f16_t, 32>), grid, block, (hipStream_t)stream, a); }
    } else {
        if (head_dim == 16) { GVFI_LAUNCH_COOP((attn_global_mfma_kernel<bf16_t, 16>), grid, block, (hipStream_t)stream, a); }
        else { GVFI_LAUNCH_COOP((attn_global_mfma_kernel<bf16_t, 32>), grid, block, (hipStream_t)stream, a); }
    }
    return (int)hipGetLastError();
}
extern "C" int gvfi_attn_global_mfma(const void* q, int ldq, long long qb1, long long qb0, long long qs, const void* k, int ldk,
                                     const void* v, int ldv, long long kb1, long long kb0, long long ks, void* out, int ldo,
                                     long long ob1, long long ob0, long long os, long long G1, int G0, int NQ, int M, int heads,
                                     int head_dim, float scale, void* stream) {
    return attn_global_mfma_t(GVFI_BF16, q, ldq, qb1, qb0, qs, k, ldk, v, ldv, kb1, kb0, ks, out, ldo, ob1, ob0, os, G1, G0, NQ, M, heads,
                              head_dim, scale, stream);
}
extern "C" int gvfi_attn_global_mfma_f16(const void* q, int ldq, long long qb1, long long qb0, long long qs, const void* k, int ldk,
                                         const void* v, int ldv, long long kb1, long long kb0, long long ks, void* out, int ldo,
                                         long long ob1, long long ob0, long long os, long long G1, int G0, int NQ, int M, int heads,
                                         int head_dim, float scale, void* stream) {
    return attn_global_mfma_t(GVFI_F16, q, ldq, qb1, qb0, qs, k, ldk, v, ldv, kb1, kb0, ks, out, ldo, ob1, ob0, os, G1, G0, NQ, M, heads,
                              head_dim, scale, stream);
}

static int attn_window_mfma_t(int dtype, const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const float* kpad,
                              const float* vpad, void* out, int ldo, int n_img, int H, int W, int ws, int heads,
                              int head_dim, float scale, void* stream) {
    if (!gvfi_attn_mfma_ok(1, 49, 49, head_dim, ws, dtype)) return -2;
    if ((((uintptr_t)q | (uintptr_t)k) & 15) || (((uintptr_t)out) & 7) || (ldq % 8) || (ldk % 8) || (ldo % 4)) return -3;
    AttnWArgs a;
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.kpad = kpad; a.vpad = vpad; a.o = (bf16_t*)out;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.n_img = n_img; a.H = H; a.W = W; a.heads = heads;
    a.nwx = (W + 6) / 7; a.nwy = (H + 6) / 7;
    a.scale_log2e = scale * 1.44269504088896341f;
    const long long tasks = (long long)n_img * a.nwx * a.nwy * heads;
    const dim3 grid((unsigned)((tasks + 3) / 4)), block(256);
    if (dtype == GVFI_F16) {
        if (head_dim == 16) { GVFI_LAUNCH_COOP((attn_window_mfma_kernel<f16_t, 16>), grid, block, (hipStream_t)stream, a); }
        else { GVFI_LAUNCH_COOP((attn_window_mfma_kernel<f16_t, 32>), grid, block, (hipStream_t)stream, a); }
    } else {
        if (head_dim == 16) { GVFI_LAUNCH_COOP((attn_window_mfma_kernel<bf16_t, 16>), grid, block, (hipStream_t)stream, a); }
        else { GVFI_LAUNCH_COOP((attn_window_mfma_kernel<bf16_t, 32>), grid, block, (hipStream_t)stream, a); }
    }
    return (int)hipGetLastError();
}
extern "C" int gvfi_attn_window_mfma(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const float* kpad,
                                     const float* vpad, void* out, int ldo, int n_img, int H, int W, int ws, int heads,
                                     int head_dim, float scale, void* stream) {
    return attn_window_mfma_t(GVFI_BF16, q, ldq, k, ldk, v, ldv, kpad, vpad, out, ldo, n_img, H, W, ws, heads, head_dim, scale, stream);
}
extern "C" int gvfi_attn_window_mfma_f16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const float* kpad,
                                         const float* vpad, void* out, int ldo, int n_img, int H, int W, int ws, int heads,
                                         int head_dim, float scale, void* stream) {
    return attn_window_mfma_t(GVFI_F16, q, ldq, k, ldk, v, ldv, kpad, vpad, out, ldo, n_img, H, W, ws, heads, head_dim, scale, stream);
}
