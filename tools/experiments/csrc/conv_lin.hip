// Linear layers on very many rows with a small K (gfx950, 16-bit operand types): the transformer linears of GIMM-VFI-F's
// Twins encoders and latent cost encoder (twins.py:331-546, encoder.py:214-346) -- 115-800 k token rows, K = 128 ... 512,
// N = 128 ... 512.  On the LDS-DMA tiles these launches are fixed cost per workgroup (a 128 x 128 tile lives for a
// prologue, ONE or two K steps and an epilogue: 120-280 TFLOP/s, half of what their HBM traffic allows; 5 of the 42 ms of a
// GIMM-VFI-F step at 448x256).  They are streaming problems: the weights are a few tens of KB, the rows cross the chip once.
//
//   * WEIGHTS STAY IN REGISTERS.  A workgroup is 4 waves; wave w owns the output features [32 NBW w, 32 NBW (w + 1)) for all
//     rows and loads its NBW x K/16 MFMA fragments ONCE from the fragment-ordered image (w_layout 2 of conv_igemm_glds.hip:
//     one coalesced 1 KiB load per fragment) -- 32 ... 128 VGPRs that are reused for every row tile of the persistent loop.
//   * The weights are the ROW operand of v_mfma_f32_32x32x16 (D[feature][token], as in token_chain.hip): a lane holds one
//     token and groups of four consecutive features, so a token's operand is 16 bytes straight from its row in global
//     memory (no LDS at all: the four waves of a workgroup read the same rows, the L1 serves three of them), bias /
//     activation / residual run in registers on four features at a time, and the result leaves as 8-byte (16-bit outputs) or
//     16-byte (float outputs: the residual streams of the transformer blocks) pieces of the token's row.
//   * The rows stream through a ring of operand registers AR k-steps deep (the loads of tile t + 1 are behind the MFMAs of
//     tile t in program order; hipcc places the counted waits).
// Same arithmetic as the LDS-DMA kernel up to the order in which an MFMA adds its 16 products (results agree to fp32
// rounding of the accumulation, as between any two tile shapes of that kernel).
#include "conv_mma.h"
#include "gimmvfi_experiments.h"

struct LinArgs {
    gvfi_conv_params p;
    long long rows;
    int nb_tot;         // 32-feature blocks of the weight image (ceil(Cout / 32))
};

// NBW: feature blocks per wave; KS: k-steps of 16 (K = 16 KS = c0 + c1); G: token groups of 32 per tile.
// STAGE: the rows of a tile travel global -> registers -> LDS as COALESCED 16-byte chunks (consecutive lanes = consecutive
// chunks of a row), double-buffered across the persistent loop, and the MFMA operands are ds_read_b128 of a padded tile (row
// pitch an odd multiple of 16 bytes: the 32 rows of a fragment read fall into distinct bank groups).  Without it (first
// version, measured: profiles/r4_lin_kernel_ab.txt) every lane loads 16 bytes of ITS row straight from global memory -- 64
// cache lines per load instruction -- and the kernel is slower than the LDS-DMA tiles it was meant to beat.
template <typename T, int NBW, int KS, int G, bool STAGE>
__global__ void __launch_bounds__(256) conv_lin_kernel(LinArgs a) {
    const gvfi_conv_params& p = a.p;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, h = lane >> 5;
    constexpr int TT = 32 * G;                      // tokens per tile
    constexpr int K = 16 * KS;
    constexpr int PITCH = K * 2 + 16;               // LDS bytes per staged row
    constexpr int CPR = K / 8;                      // 16-byte chunks per row
    constexpr int NCH = TT * CPR / 256;             // chunks per thread and tile
    static_assert((TT * CPR) % 256 == 0, "tile chunks must divide over the workgroup");
    constexpr int AR = KS < 16 ? KS : 16;           // operand ring depth of the un-staged form
    __shared__ __attribute__((aligned(16))) unsigned char lds[STAGE ? 2 * TT * PITCH : 16];
    // ---- this wave's weight fragments: (block nb, k-step ks) = fragment ((ks / 4) * nb_tot + nb) * 4 + ks % 4 of the image
    uint4 wreg[NBW][KS];
    const uint4* wf = (const uint4*)p.w;
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
        const int nb = wave * NBW + j;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (nb < a.nb_tot) wreg[j][ks] = wf[(((long long)(ks >> 2) * a.nb_tot + nb) * 4 + (ks & 3)) * 64 + lane];
            else { wreg[j][ks].x = 0u; wreg[j][ks].y = 0u; wreg[j][ks].z = 0u; wreg[j][ks].w = 0u; }
        }
    }
    const T* __restrict__ x0 = (const T*)p.x0;
    const T* __restrict__ x1 = (const T*)p.x1;
    const long long ntiles = (a.rows + TT - 1) / TT;
    // staged form: chunk c = tid + 256 i of a tile = (row c / CPR, 16-byte column c % CPR)
    auto chunk_src = [&](long long t, int i) -> const uint4* {
        const int c = tid + 256 * i;
        const int r = c / CPR, k = (c - r * CPR) * 8;
        long long row = t * TT + r;
        row = row < a.rows ? row : a.rows - 1;
        return (const uint4*)(k < p.c0 ? x0 + row * p.ld0 + k : x1 + row * p.ld1 + (k - p.c0));
    };
    auto chunk_dst = [&](int buf, int i) -> uint4* {
        const int c = tid + 256 * i;
        const int r = c / CPR, col = c - r * CPR;
        return (uint4*)(lds + buf * (TT * PITCH) + r * PITCH + col * 16);
    };
    uint4 nxt[STAGE ? NCH : 1];
    int cur = 0;
    if (STAGE) {
        if ((long long)blockIdx.x < ntiles) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) nxt[i] = *chunk_src(blockIdx.x, i);
#pragma unroll
            for (int i = 0; i < NCH; ++i) *chunk_dst(0, i) = nxt[i];
        }
        __syncthreads();
    }
    for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        long long row[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const long long r = t * TT + 32 * g + n;
            row[g] = r < a.rows ? r : a.rows - 1;          // tail lanes recompute the last row, never store
        }
        const long long tn = t + gridDim.x;
        if (STAGE && tn < ntiles) {                         // the next tile's rows: requested now, written to LDS after the MFMAs
#pragma unroll
            for (int i = 0; i < NCH; ++i) nxt[i] = *chunk_src(tn, i);
        }
        auto a_src = [&](int g, int ks) -> const uint4* {
            const int k = 16 * ks + 8 * h;
            return (const uint4*)(k < p.c0 ? x0 + row[g] * p.ld0 + k : x1 + row[g] * p.ld1 + (k - p.c0));
        };
        f32x16 acc[NBW][G];
#pragma unroll
        for (int j = 0; j < NBW; ++j)
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][g][r] = 0.f;
        if (STAGE) {
            const unsigned char* tb = lds + cur * (TT * PITCH);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const uint4 av = *(const uint4*)(tb + (32 * g + n) * PITCH + (16 * ks + 8 * h) * 2);
#pragma unroll
                    for (int j = 0; j < NBW; ++j) Mma2<T>::run(acc[j][g], wreg[j][ks], av);
                }
            }
        } else {
            uint4 areg[G][AR];
#pragma unroll
            for (int ks = 0; ks < AR; ++ks)
#pragma unroll
                for (int g = 0; g < G; ++g) areg[g][ks] = *a_src(g, ks);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
#pragma unroll
                    for (int j = 0; j < NBW; ++j) Mma2<T>::run(acc[j][g], wreg[j][ks], areg[g][ks % AR]);
                    if (ks + AR < KS) areg[g][ks % AR] = *a_src(g, ks + AR);
                }
            }
        }
        if (STAGE) {
            // (the other buffer was last read one tile ago: every wave has passed the barrier that ended that tile)
            if (tn < ntiles) {
#pragma unroll
                for (int i = 0; i < NCH; ++i) *chunk_dst(cur ^ 1, i) = nxt[i];
            }
        }
        // ---- epilogue: y = act1(acc + bias) (+ res); register r of block nb = feature 32 nb + (r & 3) + 8 (r >> 2) + 4 h
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const long long r_ = t * TT + 32 * g + n;
            if (r_ >= a.rows) continue;
#pragma unroll
            for (int j = 0; j < NBW; ++j) {
                const int nb = wave * NBW + j;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int f0 = 32 * nb + 8 * q + 4 * h;
                    if (f0 >= p.Cout) continue;
                    float v[4];
                    float bb[4] = {0.f, 0.f, 0.f, 0.f};
                    if (p.bias) {
                        const float4 b4 = *(const float4*)(p.bias + f0);
                        bb[0] = b4.x; bb[1] = b4.y; bb[2] = b4.z; bb[3] = b4.w;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float s = acc[j][g][4 * q + e] + bb[e];
                        v[e] = p.act1 == GVFI_ACT_GELU ? fast_gelu(s) : (p.act1 == GVFI_ACT_RELU ? fmaxf(s, 0.f) : s);
                    }
                    if (p.res) {
                        if (p.res_f32) {
                            const float4 rv = *(const float4*)((const float*)p.res + r_ * p.ldr + f0);
                            v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                        } else {
                            const uint2 u = *(const uint2*)((const T*)p.res + r_ * p.ldr + f0);
                            v[0] += cvt16<T>(u.x & 0xffffu); v[1] += cvt16<T>(u.x >> 16);
                            v[2] += cvt16<T>(u.y & 0xffffu); v[3] += cvt16<T>(u.y >> 16);
                        }
                    }
                    if (p.y_f32) {
                        float4 o4;
                        o4.x = v[0]; o4.y = v[1]; o4.z = v[2]; o4.w = v[3];
                        *(float4*)((float*)p.y + r_ * p.ldy + f0) = o4;
                    } else {
                        uint2 u;
                        u.x = pack16x2<T>(v[0], v[1]);
                        u.y = pack16x2<T>(v[2], v[3]);
                        *(uint2*)((T*)p.y + r_ * p.ldy + f0) = u;
                    }
                }
            }
        }
        if (STAGE) {
            __syncthreads();        // the next tile is in LDS, and nobody reads this tile's buffer any more
            cur ^= 1;
        }
    }
}

// which instantiation serves (K, N): 0 = none.  Encoded NBW * 1000 + KS * 10 + G
static int lin_variant(int K, int N) {
    const int nbw = (N + 127) / 128;          // feature blocks per wave: 4 waves x NBW x 32 >= N
    if (K == 128) return nbw == 1 ? 1082 : nbw == 2 ? 2082 : nbw == 3 ? 3081 : nbw == 4 ? 4081 : 0;
    if (K == 192) return nbw == 1 ? 1122 : nbw == 2 ? 2121 : 0;
    if (K == 256) return nbw == 1 ? 1162 : nbw == 2 ? 2161 : 0;
    if (K == 512) return nbw == 1 ? 1321 : 0;
    return 0;
}

// 1 = a many-row problem this kernel was built for, 2 = runnable but latency-sized.  gvfi_conv2d never routes here by itself:
// the kernel runs on an explicit algo 8 only (the Python host asks for it when GVFI_LIN=1 -- off by default: measured slower
// than the LDS-DMA tiles, profiles/r4_lin_kernel_ab.txt)
extern "C" int gvfi_conv2d_lin_eligible(const gvfi_conv_params* pp) {
    const gvfi_conv_params& p = *pp;
    if (p.dtype != GVFI_BF16 && p.dtype != GVFI_F16) return 0;
    if (p.KH != 1 || p.KW != 1 || p.stride != 1 || p.pad_h != 0 || p.pad_w != 0 || p.groups > 1 || p.w_layout != 2) return 0;
    if (p.epi_mode != GVFI_EPI_STD || p.stats != nullptr || p.act2 != GVFI_ACT_NONE || p.out_scale != 1.0f) return 0;
    if (p.act1 != GVFI_ACT_NONE && p.act1 != GVFI_ACT_GELU && p.act1 != GVFI_ACT_RELU) return 0;
    if ((p.c0 % 16) || (p.c1 % 16) || (p.c0 % 64 && p.c1 > 0) || (p.Cout % 4) || p.Cout <= 0) return 0;
    if (lin_variant(p.c0 + p.c1, p.Cout) == 0) return 0;
    if (((uintptr_t)p.x0 & 15) || ((p.ld0 * 2) & 15) || (p.c1 > 0 && (((uintptr_t)p.x1 & 15) || ((p.ld1 * 2) & 15))) || ((uintptr_t)p.w & 15)) return 0;
    const int ey = p.y_f32 ? 4 : 2;
    if (((uintptr_t)p.y & (p.y_f32 ? 15 : 7)) || ((p.ldy * ey) & (p.y_f32 ? 15 : 7)) || (p.bias && ((uintptr_t)p.bias & 15))) return 0;
    if (p.res) {
        const int er = p.res_f32 ? 4 : 2;
        if (((uintptr_t)p.res & (p.res_f32 ? 15 : 7)) || ((p.ldr * er) & (p.res_f32 ? 15 : 7))) return 0;
    }
    const long long rows = (long long)p.N * p.Ho * p.Wo;
    return rows >= 65536 ? 1 : 2;       // below that the launch is latency, whatever the kernel
}

extern "C" int gvfi_conv2d_lin(const gvfi_conv_params* pp, void* stream) {
    if (!gvfi_conv2d_lin_eligible(pp)) return -2;
    const gvfi_conv_params& p = *pp;
    LinArgs a;
    a.p = p;
    a.rows = (long long)p.N * p.Ho * p.Wo;
    a.nb_tot = (p.Cout + 31) / 32;
    const int var = lin_variant(p.c0 + p.c1, p.Cout);
    const int g = var % 10;
    const long long ntiles = (a.rows + 32 * g - 1) / (32 * g);
    // persistent: two workgroups per CU (8 waves: one hides the other's row loads), fewer when there is less work
    const int grid = (int)(ntiles < 512 ? ntiles : 512);
    hipStream_t st = (hipStream_t)stream;
#define GVFI_LIN(NBW_, KS_, G_)                                                                                             \
    do {                                                                                                                    \
        if (p.algo & 32) {      /* A/B: the un-staged first version */                                                      \
            if (p.dtype == GVFI_F16) { GVFI_LAUNCH_COOP((conv_lin_kernel<f16_t, NBW_, KS_, G_, false>), dim3(grid), dim3(256), st, a); }  \
            else { GVFI_LAUNCH_COOP((conv_lin_kernel<bf16_t, NBW_, KS_, G_, false>), dim3(grid), dim3(256), st, a); }                 \
        } else {                                                                                                            \
            if (p.dtype == GVFI_F16) { GVFI_LAUNCH_COOP((conv_lin_kernel<f16_t, NBW_, KS_, G_, true>), dim3(grid), dim3(256), st, a); }   \
            else { GVFI_LAUNCH_COOP((conv_lin_kernel<bf16_t, NBW_, KS_, G_, true>), dim3(grid), dim3(256), st, a); }                  \
        }                                                                                                                   \
    } while (0)
    switch (var) {
        case 1082: GVFI_LIN(1, 8, 2); break;
        case 2082: GVFI_LIN(2, 8, 2); break;
        case 3081: GVFI_LIN(3, 8, 1); break;
        case 4081: GVFI_LIN(4, 8, 1); break;
        case 1122: GVFI_LIN(1, 12, 2); break;
        case 2121: GVFI_LIN(2, 12, 1); break;
        case 1162: GVFI_LIN(1, 16, 2); break;
        case 2161: GVFI_LIN(2, 16, 1); break;
        case 1321: GVFI_LIN(1, 32, 1); break;
        default: return -2;
    }
#undef GVFI_LIN
    return (int)hipGetLastError();
}
