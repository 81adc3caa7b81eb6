// 3x3 stride-1 zero-padded convolution for the FLOP-dominant layers (bf16, Cout a multiple of 256, channel counts
// multiples of 64, >= 65536 output pixels): the ResBlocks of NewMultiFlowDecoder (fi_components.py:97-154, 279-340) --
// 63-87 % of the path's FLOPs.  Same math, weight image (w_layout 1), tile (256 x 256, 8 waves, 128 x 64 per wave),
// chunk-pipelined K loop and epilogue arithmetic as conv_igemm_glds.hip; what differs is the A operand.
//
// conv_igemm_glds.hip walks K as (channel chunk, tap) and DMAs a fresh 256-pixel x 64-channel A tile for every tap: 9 x
// 32 KB per channel chunk, beside 9 x 32 KB of weights.  Round 2 measured that kernel bound on the L2 -> LDS delivery side
// (62 us of K loop per tile at any shader clock), so here the traffic itself goes down:
//   * the M tile is a 16 x 16 block of output pixels (not 256 consecutive ones), and the A operand of a channel chunk is
//     its 18 x 18 halo PATCH, staged ONCE (45.6 KB instead of 9 x 32 KB; -43 % of all L2 -> LDS bytes, -86 % of the
//     A-side DMA instructions).  Zero padding = out-of-range DMA offsets, as before;
//   * patch pixels are 144 bytes apart (128 + 16: an odd multiple of 16 B): the ds_read_b128 fragment reads of 16
//     neighbouring pixels tile the 64 banks without the XOR swizzle, so a tap is a uniform offset added to four per-lane
//     base addresses (a 16-lane read group is one pixel row of the tile: conflict-free).  A DMA instruction still writes 1 KiB lane-linear: lanes whose 16-byte
//     slot is the padding slot of a pixel use the out-of-range offset;
//   * two patch buffers (the next channel chunk's patch streams in one piece per tap, taps 0-5) + the two-stage weight
//     ring = 158 848 bytes of LDS.
#include "conv_mma.h"
#include <type_traits>

#ifndef GVFI_HOSTSIM
#define P3_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#else
#define P3_WAVE_SYNC() emu::wave_sync()
#endif
// (a wave-uniform value the compiler keeps in a VGPR -- e.g. derived from an integer division -- for an SGPR asm operand)
#define P3_UNIFORM(x) ((unsigned)__builtin_amdgcn_readfirstlane((int)(x)))
#define P3_TH 16
#define P3_TW 16
#define P3_PW (P3_TW + 2)
#define P3_PIX ((P3_TH + 2) * P3_PW)           // 324 patch pixels
#define P3_PITCH 144
#define P3_SLOTS (P3_PIX * 9)                  // 16-byte slots of a patch incl. the padding slot of every pixel
#define P3_PIECES ((P3_SLOTS + 63) / 64)       // 46 DMA instructions
#define P3_PATCH (P3_PIECES * 1024)            // 47 104 bytes: 324 x 144 rounded up to whole DMA instructions (the tail lanes write zeros)
#define P3_BSTAGE (256 * 128)

struct P3Args {
    gvfi_conv_params p;
    int chunks0, chunks;        // 64-channel chunks of source 0 / of both sources
    int tiles_x, tiles_y, mtiles, ntiles_n, per_xcd;
    int slots;                  // (stream kernel) workgroups per XCD: the grid is 8 x slots
    int dbg;                    // (stream kernel, profiling build only: tile_hint bits 10, 11) 1: no patch offsets, 2: no stores
};

// ---- wave-private epilogue ----------------------------------------------------------------------------------------------
// y = act2(act1(acc + bias) + res) * out_scale for ONE wave's 128 pixels x 64 channels, in four passes of 32 pixels (one
// accumulator row block i each) through two 4 KB staging regions of LDS that belong to this wave alone -- no workgroup barrier
// anywhere: a pass writes its 8-byte pieces (lane = pixel, 4 consecutive channels), reads them back as 16-byte units
// (8 lanes = the 128 bytes one pixel has in this wave's channel range) and stores four times 8 pixels x 128 bytes, whole
// cache lines.  Row `px` of a region keeps its 16-byte unit u at slot u ^ ((px >> 1) & 7): the 64 lanes of a ds_write_b64 cover
// the 64 banks twice, the 16 lanes of a ds_read_b128 group (two pixels) once.  The residual arrives in the same layout
// (pre-swizzled on the source side) and the result overwrites it in place: passes 0 and 1 by LDS-DMA into the two regions,
// passes 2 and 3 into REGISTERS first and from there into the region its predecessor has just left.  A wave's vector-memory
// operations retire in order, so every residual load has to be in the queue before the first store: a load behind a store would
// wait for that store to reach memory -- microseconds when every CU of the chip stores its tile at the same time.
//   queue: DMA0 DMA1 | ld2 ld3 st0 st1 | st2 st3      (4 instructions each; without a residual only the stores)
// FAST: every slope of the launch is <= 1 (none = 1, ReLU = 0, leaky = 0.1, the usual PReLU): act(t) = max(t, s t) -- one
// operation instead of the select of +-inf and the median (v_med3(t, s t, s <= 1 ? +inf : -inf), same value for finite t).
// Same arithmetic and order as the workgroup-wide form below: bit-identical.
// reg0: byte offset of the wave's 2 x 4 KB inside smem; pix0: pixel index of the tile's first pixel; the tile's row r is pixel
// (y0 + r / 16, x0 + r % 16).
// what the epilogue needs of the launch (the stream kernel keeps it in LDS between tiles instead of in 12 scalar registers)
struct P3Epi {
    const void* y;
    const void* res;
    int ldy, ldr, H, W;
    float out_scale;
    int nostore;      // (profiling experiments)
};
template <bool RES, bool SC, bool FAST>
__device__ __forceinline__ void p3_wave_epilogue(const P3Epi& p, f32x16 (&acc)[4][2], unsigned char* smem, unsigned smem_lds,
                                                 unsigned reg0, const float* ptab, int lane, int wm, int wn, long long pix0, int y0,
                                                 int x0, int n0) {
    const int px = lane & 31, fhalf = lane >> 5, fsw = (px >> 1) & 7;
    const int q0 = lane >> 3, un = lane & 7;
    const float inf = __builtin_inff();
    const gvfi_rsrc_t rs_y = make_rsrc((const bf16_t*)p.y + pix0 * p.ldy + n0 + wn * 64);
    const bf16_t* rbase = RES ? (const bf16_t*)p.res + pix0 * p.ldr + n0 + wn * 64 : (const bf16_t*)p.y;
    const gvfi_i32x4 srd_r = make_srd(rbase);
    const gvfi_rsrc_t rs_r = make_rsrc(rbase);
    // per-lane byte offsets of its pixel column x = q0 / q0 + 8 (k even / odd); the pixel row enters as a scalar offset
    unsigned yv[2], rv[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int xx = q0 + 8 * h;
        const bool ok = x0 + xx < p.W;
        yv[h] = ok ? (unsigned)((xx * p.ldy + un * 8) * 2) : GVFI_DMA_OOB;
        rv[h] = ok ? (unsigned)((xx * p.ldr + ((un ^ (q0 >> 1) ^ (4 * h)) * 8)) * 2) : GVFI_DMA_OOB;
    }
    auto res_dma = [&](int i) {      // residual of pass i -> its staging region
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int yy = wm * 8 + i * 2 + (k >> 1);
            const bool oky = y0 + yy < p.H;
            bufdma16(oky ? rv[k & 1] : GVFI_DMA_OOB, srd_r, (unsigned)(yy * p.W * p.ldr * 2), smem_lds + reg0 + (unsigned)((i & 1) * 4096 + k * 1024));
        }
    };
    constexpr int GF = RES ? 2 : 4;
    uint4 rreg[2][4];
    if (RES) {
        res_dma(0);
        res_dma(1);
    }
    // two passes at a time (i = 2 pr, 2 pr + 1: the two staging regions): the per-channel parameters of four channel groups are
    // fetched once and serve both -- a parameter fetch in front of every 4 x 4 values made the epilogue a chain of 32 LDS
    // round trips per wave (6.9 of the stream kernel's 8.3 kcycles per tile)
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        if (RES) {
            if (pr == 0) {
                glds_wait_n<0>();      // DMA0, DMA1
            } else {
#pragma unroll
                for (int i = 2; i < 4; ++i)
#pragma unroll
                    for (int k = 0; k < 4; ++k) *(uint4*)(smem + reg0 + (i & 1) * 4096 + k * 1024 + lane * 16) = rreg[i - 2][k];
            }
            P3_WAVE_SYNC();
        }
#pragma unroll
        for (int jg = 0; jg < 8; jg += GF) {      // GF channel groups per parameter fetch (their registers: 8 / 12 per group)
            const int j = jg >> 2;
            float4 b4[GF], s4[GF], z4[GF];
            int cj = wn * 64 + jg * 8 + 4 * fhalf;
            GVFI_OPAQUE_V(cj);      // (fetched per pass pair: kept across the pairs the parameters would cost 96 registers)
#pragma unroll
            for (int gg = 0; gg < GF; ++gg) {
                const int c0 = cj + 8 * gg;          // first of this lane's 4 channels (within the tile)
                b4[gg] = *(const float4*)(ptab + c0);
                s4[gg] = *(const float4*)(ptab + 256 + c0);
                if (RES) z4[gg] = *(const float4*)(ptab + 512 + c0);
            }
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = 2 * pr + ii;
                const unsigned reg = reg0 + (unsigned)(ii * 4096);
#pragma unroll
                for (int gg = 0; gg < GF; ++gg) {
                    const int g = (jg & 3) + gg;
                    const float bb[4] = {b4[gg].x, b4[gg].y, b4[gg].z, b4[gg].w}, s1[4] = {s4[gg].x, s4[gg].y, s4[gg].z, s4[gg].w};
                    float s2[4] = {1.f, 1.f, 1.f, 1.f};
                    if (RES) { s2[0] = z4[gg].x; s2[1] = z4[gg].y; s2[2] = z4[gg].z; s2[3] = z4[gg].w; }
                    unsigned char* sp = smem + reg + px * 128 + (((j * 4 + g) ^ fsw) << 4) + fhalf * 8;
                    float vv[4];
#ifndef GVFI_HOSTSIM
                    {   // packed pairs: v_pk_add_f32 / v_pk_mul_f32
                        typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const f2 a2 = {acc[i][j][4 * g + 2 * h], acc[i][j][4 * g + 2 * h + 1]};
                            const f2 t2 = a2 + f2{bb[2 * h], bb[2 * h + 1]};
                            const f2 st = t2 * f2{s1[2 * h], s1[2 * h + 1]};
                            if (FAST) {
                                vv[2 * h] = __builtin_fmaxf(t2.x, st.x);
                                vv[2 * h + 1] = __builtin_fmaxf(t2.y, st.y);
                            } else {
                                vv[2 * h] = med3f(t2.x, st.x, s1[2 * h] <= 1.f ? inf : -inf);
                                vv[2 * h + 1] = med3f(t2.y, st.y, s1[2 * h + 1] <= 1.f ? inf : -inf);
                            }
                        }
                    }
#else
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t = acc[i][j][4 * g + e] + bb[e];
                        vv[e] = FAST ? fmaxf(t, s1[e] * t) : med3f(t, s1[e] * t, s1[e] <= 1.f ? inf : -inf);
                    }
#endif
                    if (RES) {
                        const uint2 ru = *(const uint2*)sp;
                        vv[0] += __builtin_bit_cast(float, ru.x << 16);
                        vv[1] += __builtin_bit_cast(float, ru.x & 0xffff0000u);
                        vv[2] += __builtin_bit_cast(float, ru.y << 16);
                        vv[3] += __builtin_bit_cast(float, ru.y & 0xffff0000u);
#ifndef GVFI_HOSTSIM
                        {
                            typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const f2 t2 = {vv[2 * h], vv[2 * h + 1]};
                                const f2 st = t2 * f2{s2[2 * h], s2[2 * h + 1]};
                                if (FAST) {
                                    vv[2 * h] = __builtin_fmaxf(t2.x, st.x);
                                    vv[2 * h + 1] = __builtin_fmaxf(t2.y, st.y);
                                } else {
                                    vv[2 * h] = med3f(t2.x, st.x, s2[2 * h] <= 1.f ? inf : -inf);
                                    vv[2 * h + 1] = med3f(t2.y, st.y, s2[2 * h + 1] <= 1.f ? inf : -inf);
                                }
                            }
                        }
#else
#pragma unroll
                        for (int e = 0; e < 4; ++e) vv[e] = FAST ? fmaxf(vv[e], s2[e] * vv[e]) : med3f(vv[e], s2[e] * vv[e], s2[e] <= 1.f ? inf : -inf);
#endif
                    }
                    if (SC) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) vv[e] *= p.out_scale;
                    }
                    uint2 u;
                    u.x = pack_bf16x2(vv[0], vv[1]);
                    u.y = pack_bf16x2(vv[2], vv[3]);
                    *(uint2*)sp = u;
                }
            }
        }
        if (RES && pr == 0) {
            // passes 2 and 3: requested here -- behind the first pair's arithmetic (their 32 registers do not fit beside it), in
            // front of its stores
#pragma unroll
            for (int i = 2; i < 4; ++i) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int yy = wm * 8 + i * 2 + (k >> 1);
                    const bool oky = y0 + yy < p.H && rv[k & 1] != GVFI_DMA_OOB;
                    rreg[i - 2][k] = bufld16(rs_r, oky ? rv[k & 1] + (unsigned)(yy * p.W * p.ldr * 2) : GVFI_DMA_OOB);
                }
            }
        }
        P3_WAVE_SYNC();
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = 2 * pr + ii;
            const unsigned reg = reg0 + (unsigned)(ii * 4096);
            uint4 t[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int q = k * 8 + q0;
                t[k] = *(const uint4*)(smem + reg + q * 128 + ((un ^ ((q >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int yy = wm * 8 + i * 2 + (k >> 1);
                const bool oky = y0 + yy < p.H;
                if (!p.nostore) bufst16(t[k], rs_y, oky && yv[k & 1] != GVFI_DMA_OOB ? yv[k & 1] + (unsigned)(yy * p.W * p.ldy * 2) : GVFI_DMA_OOB);
            }
        }
        P3_WAVE_SYNC();
    }
}

// 8 waves of 128 x 64, 2 per SIMD.  (Measured and dropped: 4 waves of 128 x 128 with 256 accumulator registers -- a third
// fewer LDS fragment reads per MFMA, but one wave per SIMD: 0.965 vs 0.861 ms on the 8 x 256 x 448 256->256 layer.)
// PROF (algo bit 15): wave 0 adds up shader-clock cycles per phase into aux1[block * 4 + {prologue, K loop, of which waiting
// for the DMA + barrier, epilogue}] (tools/p3x3_timeline.py)
// VAR bit 0: wave-private epilogue (p3_wave_epilogue) instead of the workgroup-wide staging tile
template <bool PROF, int VAR> __global__ void __launch_bounds__(512) conv_p3x3_kernel(P3Args a) {
    typedef bf16_t T;
    constexpr int WAVES_N = 4;
    constexpr int NW = 2 * WAVES_N, NT = NW * 64, WM = 128, WN = 256 / WAVES_N, MI = 4, NI = WN / 32, BN = 256, BM = 256, RB = 128, KK = 4;
    constexpr int QP = (P3_PIECES + NW - 1) / NW;      // patch pieces per wave (6 / 12)
    constexpr int PPT = (QP + 5) / 6;                  // ... of the next channel chunk issued per tap (taps 0..5)
    constexpr int B_INSTR = 32 / NW;
    constexpr int RPI = NT / 32;                       // tile rows per store-loop iteration
    constexpr int PTAB = 2 * P3_PATCH + 2 * P3_BSTAGE;      // per-channel epilogue parameters: bias, slope 1, slope 2 (3 x 256 floats)
    __shared__ __attribute__((aligned(16))) unsigned char smem[PTAB + 3 * 256 * 4];
    const gvfi_conv_params& p = a.p;
    const int bid = blockIdx.x;
    const int v = (bid & 7) * a.per_xcd + (bid >> 3);          // XCD-aware order, as conv_igemm_glds.hip
    if (v >= a.mtiles * a.ntiles_n) return;
    const int mt = v / a.ntiles_n, nt = v - mt * a.ntiles_n;
    const int n0 = nt * BN;
    const int tiles_img = a.tiles_x * a.tiles_y;
    const int img = mt / tiles_img, trem = mt - img * tiles_img;
    const int tyi = trem / a.tiles_x, txi = trem - tyi * a.tiles_x;
    const int y0 = tyi * P3_TH, x0 = txi * P3_TW;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const T* __restrict__ xs0 = (const T*)p.x0;
    const T* __restrict__ xs1 = (const T*)p.x1;
    const T* __restrict__ wg = (const T*)p.w;

    // ---- patch DMA: per-lane byte offsets relative to the patch origin (image pixel (y0-1, x0-1), may lie in the padding)
    unsigned a_off0[QP], a_off1[QP];
#pragma unroll
    for (int q = 0; q < QP; ++q) {
        const int piece = q * NW + wave;
        const int s = piece * 64 + lane;
        const int pp = s / 9, col = s - pp * 9;
        const int py = pp / P3_PW, px = pp - py * P3_PW;
        const int gy = y0 - 1 + py, gx = x0 - 1 + px;
        const bool ok = piece < P3_PIECES && pp < P3_PIX && col < 8 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        a_off0[q] = ok ? (unsigned)(((py * p.W + px) * p.ld0 + col * 8) * 2) : GVFI_DMA_OOB;
        a_off1[q] = ok ? (unsigned)(((py * p.W + px) * p.ld1 + col * 8) * 2) : GVFI_DMA_OOB;
    }
    const long long pix_org = ((long long)img * p.H + (y0 - 1)) * p.W + (x0 - 1);
    const gvfi_i32x4 srd_a0 = make_srd(xs0 + pix_org * p.ld0);
    const gvfi_i32x4 srd_a1 = make_srd(xs1 != nullptr ? xs1 + pix_org * p.ld1 : xs0);
    const gvfi_i32x4 srd_b = make_srd(wg);
    // ---- weight tile DMA (chunk-major, pre-swizzled image: [K chunk][Cout][128 B] == the LDS image)
    const int lrow = lane >> 3, lslot = lane & 7;
    unsigned b_off[B_INSTR];
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) b_off[i] = (unsigned)(((n0 + (i * NW + wave) * 8 + lrow) * 8 + lslot) * 16);
    // ---- fragment read addresses
    auto swz = [](int row) { return (row >> 1) & 7; };
    unsigned abase[MI], b_rd[KK];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int row = wm * WM + i * 32 + (lane & 31);
        abase[i] = (unsigned)(((row >> 4) * P3_PW + (row & 15)) * P3_PITCH + (lane >> 5) * 16);
    }
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        const int rb = wn * WN + (lane & 31);
        const int slot = 2 * kk + (lane >> 5);
        b_rd[kk] = 2 * P3_PATCH + rb * RB + ((slot ^ swz(rb)) << 4);
    }
    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const unsigned smem_lds = lds_address(smem);
    auto issue_patch = [&](int c, int q) {
        const int piece = q * NW + wave;
        if (piece >= P3_PIECES) return;
        const bool from0 = c < a.chunks0;
        bufdma16(from0 ? a_off0[q] : a_off1[q], from0 ? srd_a0 : srd_a1, (unsigned)((from0 ? c : c - a.chunks0) * 128),
                 smem_lds + (c & 1) * P3_PATCH + piece * 1024);
    };
    auto issue_b = [&](int kt, int i) {
        bufdma16(b_off[i], srd_b, (unsigned)(kt * p.Cout * 128), smem_lds + 2 * P3_PATCH + (kt & 1) * P3_BSTAGE + (i * NW + wave) * 1024);
    };
    // ---- per-channel epilogue parameters -> LDS (published by the prologue barrier)
    if (tid < 256) {
        float* pt = (float*)(smem + PTAB);
        pt[tid] = p.bias ? p.bias[n0 + tid] : 0.f;
        pt[256 + tid] = p.act1 == GVFI_ACT_PRELU ? p.slope1[n0 + tid] : (p.act1 == GVFI_ACT_NONE ? 1.f : (p.act1 == GVFI_ACT_LRELU ? 0.1f : 0.f));
        pt[512 + tid] = p.act2 == GVFI_ACT_PRELU ? p.slope2[n0 + tid] : (p.act2 == GVFI_ACT_NONE ? 1.f : (p.act2 == GVFI_ACT_LRELU ? 0.1f : 0.f));
    }
    // ---- prologue: patch of chunk 0 + weights of (chunk 0, tap 0)
#pragma unroll
    for (int q = 0; q < QP; ++q) issue_patch(0, q);
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) issue_b(0, i);

    unsigned long long ph[4] = {0, 0, 0, 0}, tprev = 0;
    auto now = [&]() -> unsigned long long {
#ifndef GVFI_HOSTSIM
        return __builtin_readcyclecounter();
#else
        return 0;
#endif
    };
    if (PROF) tprev = now();
    uint4 fa[2][MI], fb[2][NI];
    // c: channel chunk (runtime), tap / kk compile-time: the tap and k-step offsets are ds_read immediates
    unsigned pa[MI], pb[KK];        // this step's fragment base addresses
    auto set_bases = [&](int c, int tap) {
#pragma unroll
        for (int i = 0; i < MI; ++i) pa[i] = abase[i] + (c & 1) * P3_PATCH;
        const int bsel = ((c + tap) & 1) * P3_BSTAGE;      // (c * 9 + tap) & 1
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) pb[kk] = b_rd[kk] + bsel;
    };
    auto load_frags = [&](int toff, int kk, int buf) {
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[buf][i] = *(const uint4*)(smem + pa[i] + toff + kk * 32);
#pragma unroll
        for (int j = 0; j < NI; ++j) fb[buf][j] = *(const uint4*)(smem + pb[kk] + j * 32 * RB);
    };
    // one (channel chunk, tap) step: 4 k-steps of 8 MFMAs; the weights of the next step (and one patch piece of the next
    // channel chunk during taps 0..5) are issued right behind the first MFMA group; the barrier that publishes them
    // sits before the last k-step (see conv_igemm_glds.hip).
    // Measured alternatives for the issue point (same box, 8 x 256 x 448 256->256, tools/p3x3_timeline.py): right after the
    // barrier, before the fragment reads: K loop 92.6 instead of 85.0 kcycles; behind the first MFMA group of the LAST
    // k-step (weights two steps ahead): 91.6 kcycles, yet 0.849 against 0.833 ms -- the shader clock rises as utilisation
    // falls (1.63 -> 1.70 GHz): the kernel runs at the chip's power limit, not at a latency limit; waves 4-7 two / four /
    // six MFMA groups later than waves 0-3: 98.6 / 102.7 / 107.6 kcycles.
    auto step = [&](int c, auto tap_tag, auto last_chunk_tag) {
        constexpr int tap = decltype(tap_tag)::value;
        constexpr bool LAST_CHUNK = decltype(last_chunk_tag)::value;
        constexpr bool HAS_NEXT = !(LAST_CHUNK && tap == 8);
        constexpr int toff = ((tap / 3) * P3_PW + (tap % 3)) * P3_PITCH;
        constexpr int ntap = tap == 8 ? 0 : tap + 1;
        constexpr int ntoff = ((ntap / 3) * P3_PW + (ntap % 3)) * P3_PITCH;
        const int kt = c * 9 + tap;
#pragma unroll
        for (int kk = 0; kk + 1 < KK; ++kk) {
            load_frags(toff, kk + 1, (kk + 1) & 1);
            GVFI_SCHED_BARRIER();
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int j = 0; j < NI; ++j) Mma2<T>::run(acc[i][j], fb[kk & 1][j], fa[kk & 1][i]);
                if (HAS_NEXT && kk == 0 && i == 0) {
#pragma unroll
                    for (int q = 0; q < B_INSTR; ++q) issue_b(kt + 1, q);
                    if (!LAST_CHUNK && tap < 6) {
#pragma unroll
                        for (int q = tap * PPT; q < (tap + 1) * PPT && q < QP; ++q) issue_patch(c + 1, q);
                    }
                }
            }
            GVFI_SCHED_BARRIER();
        }
        if (HAS_NEXT) {
            unsigned long long tw = 0;
            if (PROF) tw = now();
            glds_wait_n<0>();
            if (PROF) { const unsigned long long t = now(); ph[3] += t - tw; }     // (own DMA pieces landed)
            __syncthreads();
            if (PROF) ph[2] += now() - tw;
            set_bases(tap == 8 ? c + 1 : c, ntap);
            load_frags(ntoff, 0, KK & 1);
            GVFI_SCHED_BARRIER();
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int j = 0; j < NI; ++j) Mma2<T>::run(acc[i][j], fb[(KK - 1) & 1][j], fa[(KK - 1) & 1][i]);
        }
        GVFI_SCHED_BARRIER();
    };
    auto chunk9 = [&](int c, auto last) {
        step(c, std::integral_constant<int, 0>{}, last);
        step(c, std::integral_constant<int, 1>{}, last);
        step(c, std::integral_constant<int, 2>{}, last);
        step(c, std::integral_constant<int, 3>{}, last);
        step(c, std::integral_constant<int, 4>{}, last);
        step(c, std::integral_constant<int, 5>{}, last);
        step(c, std::integral_constant<int, 6>{}, last);
        step(c, std::integral_constant<int, 7>{}, last);
        step(c, std::integral_constant<int, 8>{}, last);
    };
    glds_wait_n<0>();
    __syncthreads();
    if (PROF) { const unsigned long long t = now(); ph[0] = t - tprev; tprev = t; }
    set_bases(0, 0);
    load_frags(0, 0, 0);
    int c = 0;
    for (; c + 1 < a.chunks; ++c) chunk9(c, std::false_type{});
    chunk9(c, std::true_type{});
    if (PROF) { const unsigned long long t = now(); ph[1] = t - tprev; tprev = t; }
    auto prof_out = [&]() {
        if (PROF && tid == 0) {
            const unsigned long long own = ph[3];
            ph[3] = now() - tprev;
            ph[2] |= own << 32;      // low word: vmcnt + barrier wait, high word: the vmcnt part
            unsigned long long* o = (unsigned long long*)p.aux1 + (size_t)bid * 4;
            for (int k = 0; k < 4; ++k) o[k] = ph[k];
        }
    };

    // ---------------------------------------------------------------- epilogue
    // y = act2(act1(acc + bias) + res) * out_scale, the arithmetic of conv_igemm_glds.hip.  The MFMAs ran with the weights
    // as their row operand: a lane holds ONE pixel (tile row wm*128 + i*32 + lane%32) and, per accumulator block, 4 x 4
    // consecutive output channels (wn*64 + j*32 + 8*(r/4) + 4*(lane/32) + r%4).  Four channels are one 8-byte unit of the
    // NHWC line, so bias / activation run as packed pairs and a ds_write_b64 stages them (the column-per-lane layout
    // of the other kernels needs a 2-byte write and ~8 VALU operations per element: 11 of this kernel's 100 kcycles).
    // act(t) = max(t,0) + s*min(t,0) == med3(t, s*t, s <= 1 ? +inf : -inf) for every activation this kernel takes
    // (none: s = 1, ReLU: s = 0, leaky: s = 0.1, PReLU: per channel).
    // Staging tile: 256 rows x 512 bytes; the 16-byte unit u of row r sits at slot u ^ (r & 15), and rows with bit 4 set
    // swap the two 8-byte halves of a unit -- 32 lanes writing 8 bytes of 32 different rows cover the 64 banks once.
    // tile row r <-> output pixel (y0 + r/16, x0 + r%16)
    const int frow = lane & 31, fhalf = lane >> 5;
    const int my_cg = tid & 31;                         // 32 groups of 8 channels per row, RPI rows per iteration
    const int my_cout0 = n0 + my_cg * 8;
    const int row_a = tid >> 5;
    const long long img_pix = (long long)img * p.H * p.W;
    auto pix_of = [&](int r, bool& ok) {
        const int y = y0 + (r >> 4), x = x0 + (r & 15);
        ok = y < p.H && x < p.W;
        return img_pix + (long long)y * p.W + x;
    };
    const bool has_sc = p.out_scale != 1.0f, has_res = p.res != nullptr;
    const float* ptab = (const float*)(smem + PTAB);
    if constexpr ((VAR & 1) != 0) {
        // wave-private epilogue (p3_wave_epilogue): one barrier (every wave has read its last fragments), then each wave stages,
        // reads back and stores its own 128 x 64 part of the tile with no further workgroup synchronisation
        __syncthreads();
        const long long pix0 = img_pix + (long long)y0 * p.W + x0;
        const unsigned reg0 = (unsigned)wave * 8192u;
        const P3Epi ep = {p.y, p.res, p.ldy, p.ldr, p.H, p.W, p.out_scale, 0};
        if (has_res) {
            if (has_sc) p3_wave_epilogue<true, true, false>(ep, acc, smem, smem_lds, reg0, ptab, lane, wm, wn, pix0, y0, x0, n0);
            else p3_wave_epilogue<true, false, false>(ep, acc, smem, smem_lds, reg0, ptab, lane, wm, wn, pix0, y0, x0, n0);
        } else {
            if (has_sc) p3_wave_epilogue<false, true, false>(ep, acc, smem, smem_lds, reg0, ptab, lane, wm, wn, pix0, y0, x0, n0);
            else p3_wave_epilogue<false, false, false>(ep, acc, smem, smem_lds, reg0, ptab, lane, wm, wn, pix0, y0, x0, n0);
        }
        prof_out();
        return;
    }
    __syncthreads();   // every wave is done reading the last staged step
    if (has_res) {
        // residual tile -> staging area (same layout: the result overwrites it in place), 128 DMA instructions of 2 rows
        // (descriptor at the tile's first pixel, offsets relative to it: the residual tensor itself may exceed the 2 GB a
        // raw-buffer offset spans -- 7 timesteps of a 1024x544x256 decoder activation in one batch are 2.0 GB)
        const long long pix0 = img_pix + (long long)y0 * p.W + x0;
        const gvfi_i32x4 srd_r = make_srd((const bf16_t*)p.res + pix0 * p.ldr + n0);
#pragma unroll
        for (int q = 0; q < 128 / NW; ++q) {
            const int piece = q * NW + wave;
            const int row = piece * 2 + (lane >> 5);
            bool ok;
            const long long pix = pix_of(row, ok);
            const unsigned off = ok ? (unsigned)((pix - pix0) * p.ldr * 2 + (((lane & 31) ^ (row & 15)) << 4)) : GVFI_DMA_OOB;
            bufdma16(off, srd_r, 0u, smem_lds + piece * 1024);
        }
        glds_wait_n<0>();
        __syncthreads();
    }
    const float inf = __builtin_inff();
    auto stage = [&](auto res_tag, auto sc_tag) {
        constexpr bool RES = decltype(res_tag)::value, SC = decltype(sc_tag)::value;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c0 = wn * WN + j * 32 + 8 * g + 4 * fhalf;          // first of this lane's 4 channels (within the tile)
                const float4 b4 = *(const float4*)(ptab + c0), s4 = *(const float4*)(ptab + 256 + c0);
                const float bb[4] = {b4.x, b4.y, b4.z, b4.w}, s1[4] = {s4.x, s4.y, s4.z, s4.w};
                float s2[4] = {1.f, 1.f, 1.f, 1.f}, k1[4], k2[4];
                if (RES) {
                    const float4 z4 = *(const float4*)(ptab + 512 + c0);
                    s2[0] = z4.x; s2[1] = z4.y; s2[2] = z4.z; s2[3] = z4.w;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    k1[e] = s1[e] <= 1.f ? inf : -inf;
                    k2[e] = s2[e] <= 1.f ? inf : -inf;
                }
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int row = wm * WM + i * 32 + frow;
                    const int unit = row * 512 + (((c0 >> 3) ^ (row & 15)) << 4);
                    float vv[4];
#ifndef GVFI_HOSTSIM
                    {   // packed pairs: v_pk_add_f32 / v_pk_mul_f32
                        typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const f2 a2 = {acc[i][j][4 * g + 2 * h], acc[i][j][4 * g + 2 * h + 1]};
                            const f2 t2 = a2 + f2{bb[2 * h], bb[2 * h + 1]};
                            const f2 st = t2 * f2{s1[2 * h], s1[2 * h + 1]};
                            vv[2 * h] = med3f(t2.x, st.x, k1[2 * h]);
                            vv[2 * h + 1] = med3f(t2.y, st.y, k1[2 * h + 1]);
                        }
                    }
#else
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t = acc[i][j][4 * g + e] + bb[e];
                        vv[e] = med3f(t, s1[e] * t, k1[e]);
                    }
#endif
                    if (RES) {
                        const uint2 ru = *(const uint2*)(smem + unit + (((c0 >> 2) & 1) << 3));
                        vv[0] += __builtin_bit_cast(float, ru.x << 16);
                        vv[1] += __builtin_bit_cast(float, ru.x & 0xffff0000u);
                        vv[2] += __builtin_bit_cast(float, ru.y << 16);
                        vv[3] += __builtin_bit_cast(float, ru.y & 0xffff0000u);
#ifndef GVFI_HOSTSIM
                        {
                            typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const f2 t2 = {vv[2 * h], vv[2 * h + 1]};
                                const f2 st = t2 * f2{s2[2 * h], s2[2 * h + 1]};
                                vv[2 * h] = med3f(t2.x, st.x, k2[2 * h]);
                                vv[2 * h + 1] = med3f(t2.y, st.y, k2[2 * h + 1]);
                            }
                        }
#else
#pragma unroll
                        for (int e = 0; e < 4; ++e) vv[e] = med3f(vv[e], s2[e] * vv[e], k2[e]);
#endif
                        P3_WAVE_SYNC();   // rows with bit 4 set: lanes l and l + 32 write the halves the other one has just read
                    }
                    if (SC) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) vv[e] *= p.out_scale;
                    }
                    uint2 u;
                    u.x = pack_bf16x2(vv[0], vv[1]);
                    u.y = pack_bf16x2(vv[2], vv[3]);
                    *(uint2*)(smem + unit + ((((c0 >> 2) ^ (row >> 4)) & 1) << 3)) = u;
                }
            }
        }
    };
    if (has_res) {
        if (has_sc) stage(std::true_type{}, std::true_type{}); else stage(std::true_type{}, std::false_type{});
    } else {
        if (has_sc) stage(std::false_type{}, std::true_type{}); else stage(std::false_type{}, std::false_type{});
    }
    __syncthreads();
    {   // iteration it of a thread = tile row row_a + 16 * it = pixel (y0 + it, x0 + row_a): one pointer, a constant stride
        static_assert(RPI == P3_TW, "store loop geometry");
        const int x = x0 + row_a;
        bf16_t* yp = (bf16_t*)p.y + (img_pix + (long long)y0 * p.W + x) * p.ldy + my_cout0;
        const long long ystride = (long long)p.W * p.ldy;
        const unsigned char* sp = smem + row_a * 512 + ((my_cg ^ row_a) << 4);
        const int nrow = x < p.W ? (p.H - y0 < P3_TH ? p.H - y0 : P3_TH) : 0;
        uint4 u[P3_TH];                      // all LDS reads first (the accumulator registers are free by now)
#pragma unroll
        for (int it = 0; it < P3_TH; ++it) {
            const uint4 t = *(const uint4*)(sp + it * (RPI * 512));
            if (it & 1) {                       // rows with bit 4 set hold the two 8-byte halves swapped
                u[it].x = t.z; u[it].y = t.w; u[it].z = t.x; u[it].w = t.y;
            } else {
                u[it] = t;
            }
        }
#pragma unroll
        for (int it = 0; it < P3_TH; ++it)
            if (it < nrow) *(uint4*)(yp + it * ystride) = u[it];
    }
    prof_out();
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same convolution as ONE STREAM of channel chunks per compute unit (algo bit 14; Cout == 256).  The kernel above pays, per
// 256 x 256 tile, a prologue (2-3 k cycles: the first patch + weight stage arrive before any MFMA), an epilogue behind two
// workgroup barriers (9.5 k, 21 k with a residual) and a workgroup turn-over (the LDS admits one workgroup per CU: the next one
// starts when the last store of this one has retired) -- 12-25 % of a tile's time with the matrix pipe idle.  Here a workgroup
// is PERSISTENT (one per CU, tiles t_begin + k * slots of its XCD's contiguous range) and the LDS-DMA ring never drains: during the
// last channel chunk of tile k the patch of chunk 0 of tile k + 1 streams into the other patch buffer (taps 0-5) and its first
// weight stage behind tap 8, exactly like any other next chunk.  The epilogue is wave-private (p3_wave_epilogue): after the
// barrier of the tile's last step the patch buffer and the weight stage that step read are dead -- waves 0-3 stage in the former,
// waves 4-7 in the latter (8 KB each) --, the stores are fire-and-forget (they retire under the next tile's K loop), one barrier
// ends the epilogue (the staging areas are the next step's DMA targets) and the accumulators are re-initialised by the first MFMA
// of the next tile taking a zero C operand.  K order, roundings and epilogue arithmetic are those of the kernel above: bit-identical.
template <bool PROF> __global__ void __launch_bounds__(512) conv_p3x3_stream_kernel(P3Args a) {
    typedef bf16_t T;
    constexpr int NW = 8, WM = 128, WN = 64, MI = 4, NI = 2, RB = 128, KK = 4;
    constexpr int QP = (P3_PIECES + NW - 1) / NW;      // patch pieces per wave (6): one per tap during taps 0..5
    constexpr int B_INSTR = 32 / NW;
    constexpr int PTAB = 2 * P3_PATCH + 2 * P3_BSTAGE;
    __shared__ __attribute__((aligned(16))) unsigned char smem[PTAB + 3 * 256 * 4 + 32];
    const gvfi_conv_params& p = a.p;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int t_last = (xcd + 1) * a.per_xcd < a.mtiles ? (xcd + 1) * a.per_xcd : a.mtiles;      // end of this XCD's range of tiles
    int tile = xcd * a.per_xcd + slot;
    if (tile >= t_last) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const T* __restrict__ xs0 = (const T*)p.x0;
    const T* __restrict__ xs1 = (const T*)p.x1;
    const int tiles_img = a.tiles_x * a.tiles_y;
    auto decode = [&](int v, int& img, int& y0, int& x0) {
        img = v / tiles_img;
        const int trem = v - img * tiles_img;
        const int tyi = trem / a.tiles_x;
        y0 = tyi * P3_TH;
        x0 = (trem - tyi * a.tiles_x) * P3_TW;
    };
    // ---- patch DMA: the lane's slot of each of its pieces in one register: pixel offset py * W + px (17 bits), 16-byte group
    // (4 bits), py, px (5 bits each), "no slot" in the sign bit.  The byte offset of a chunk is two operations away (pitch and
    // group), the border mask of a tile four more; nothing tile-dependent is kept in registers.
    unsigned geo[QP];
#pragma unroll
    for (int q = 0; q < QP; ++q) {
        const int piece = q * NW + wave;
        const int s = piece * 64 + lane;
        const int pp = s / 9, col = s - pp * 9;
        const int py = pp / P3_PW, px = pp - py * P3_PW;
        geo[q] = (piece < P3_PIECES && pp < P3_PIX && col < 8) ? (unsigned)((py * p.W + px) | (col << 17) | (py << 21) | (px << 26)) : 0x80000000u;
    }
    const unsigned smem_lds = lds_address(smem);
    const gvfi_i32x4 srd_b = make_srd(p.w);
    const int lrow = lane >> 3, lslot = lane & 7;
    const unsigned b_off = (unsigned)(((wave * 8 + lrow) * 8 + lslot) * 16);      // + i * 8 KB (scalar): instruction i of a stage
    auto swz = [](int row) { return (row >> 1) & 7; };
    // fragment read addresses: pa = patch buffer of the chunk being read, pbe / pbo = the weight stages of its even / odd taps
    // ((gc + tap) & 1); they change only at a chunk boundary: pa moves to the other buffer, pbe and pbo swap (9 taps per chunk)
    unsigned pa[MI], pbe[KK], pbo[KK];
    // (re-derived from the lane index at the start of every tile: kept across the epilogue they are 12 registers it cannot spare)
    auto init_bases = [&](unsigned gcpar) {
        int l = lane;
        GVFI_OPAQUE_V(l);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int row = wm * WM + i * 32 + (l & 31);
            pa[i] = (unsigned)(((row >> 4) * P3_PW + (row & 15)) * P3_PITCH + (l >> 5) * 16) + gcpar * (unsigned)P3_PATCH;
        }
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const int rb = wn * WN + (l & 31);
            const int slot16 = 2 * kk + (l >> 5);
            const unsigned b = 2 * P3_PATCH + rb * RB + ((slot16 ^ swz(rb)) << 4);
            pbe[kk] = b + gcpar * (unsigned)P3_BSTAGE;
            pbo[kk] = b + (gcpar ^ 1u) * (unsigned)P3_BSTAGE;
        }
    };
    init_bases(0u);
    // piece q of the patch of a chunk: descriptor at the patch origin of its tile (y0p - 1, x0p - 1), channel chunk offset soff.
    // The per-lane offset (a dozen VALU operations) is computed one step ahead, in front of the step barrier where the wave is
    // about to wait anyway: VALU work inside the MFMA stream costs far more than its issue slots.
    auto patch_off = [&](int q, int ld, int y0p, int x0p) -> unsigned {
        if (PROF && (a.dbg & 1)) return GVFI_DMA_OOB;
        unsigned g = geo[q];
        GVFI_OPAQUE_V(g);      // (decoded here, every time: hoisted out of the K loop the decoded fields cost 18 registers)
        const unsigned off = (g & 0x1ffffu) * (unsigned)(ld * 2) + ((g >> 17) & 15u) * 16u;
        bool ok = (int)g >= 0;
        // a tile whose halo lies inside the image (most of them) needs no border test
        if (!(y0p >= 1 && y0p + P3_TH < p.H && x0p >= 1 && x0p + P3_TW < p.W)) {
            const int py = (g >> 21) & 31, px = (g >> 26) & 31;
            ok = ok && (unsigned)(y0p - 1 + py) < (unsigned)p.H && (unsigned)(x0p - 1 + px) < (unsigned)p.W;
        }
        return ok ? off : GVFI_DMA_OOB;
    };
    auto issue_patch = [&](int q, unsigned off, unsigned par, gvfi_i32x4 srd, unsigned soff) {
        const int piece = q * NW + wave;
        if (piece >= P3_PIECES) return;
        bufdma16(off, srd, P3_UNIFORM(soff), P3_UNIFORM(smem_lds + par * P3_PATCH + piece * 1024));
    };
    auto issue_b = [&](unsigned soff, unsigned par, int i) {
        bufdma16(b_off, srd_b, P3_UNIFORM(soff + (unsigned)(i * NW * 1024)), P3_UNIFORM(smem_lds + 2 * P3_PATCH + par * P3_BSTAGE + (i * NW + wave) * 1024));
    };
    auto patch_srd = [&](int cn, int img, int y0, int x0, gvfi_i32x4& srd, unsigned& soff, int& ld) {
        const bool from0 = cn < a.chunks0;
        const long long pix_org = ((long long)img * p.H + (y0 - 1)) * p.W + (x0 - 1);
        ld = from0 ? p.ld0 : p.ld1;
        srd = make_srd((from0 ? xs0 : xs1) + pix_org * ld);
        soff = (unsigned)((from0 ? cn : cn - a.chunks0) * 128);
    };
    // ---- per-channel epilogue parameters -> LDS (published by the prologue barrier; Cout == 256: one table per launch)
    if (tid < 256) {
        float* pt = (float*)(smem + PTAB);
        pt[tid] = p.bias ? p.bias[tid] : 0.f;
        pt[256 + tid] = p.act1 == GVFI_ACT_PRELU ? p.slope1[tid] : (p.act1 == GVFI_ACT_NONE ? 1.f : (p.act1 == GVFI_ACT_LRELU ? 0.1f : 0.f));
        pt[512 + tid] = p.act2 == GVFI_ACT_PRELU ? p.slope2[tid] : (p.act2 == GVFI_ACT_NONE ? 1.f : (p.act2 == GVFI_ACT_LRELU ? 0.1f : 0.f));
    }
    if (tid == 0) {      // the epilogue's launch parameters, read back per tile (see P3Epi)
        unsigned* e = (unsigned*)(smem + PTAB + 3072);
        const unsigned long long yp = (unsigned long long)p.y, rp = (unsigned long long)p.res;
        e[0] = (unsigned)yp; e[1] = (unsigned)(yp >> 32); e[2] = (unsigned)rp; e[3] = (unsigned)(rp >> 32);
        e[4] = (unsigned)p.ldy; e[5] = (unsigned)p.ldr; e[6] = (unsigned)p.H; e[7] = (unsigned)p.W;
    }
    const float* ptab = (const float*)(smem + PTAB);
    const unsigned wstride = (unsigned)(p.Cout * 128);        // bytes of one (chunk, tap) stage of the weight image

    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
    auto now = [&]() -> unsigned long long {
#ifndef GVFI_HOSTSIM
        return __builtin_readcyclecounter();
#else
        return 0;
#endif
    };
    if (PROF) tprev = now();
    int img, y0, x0;
    decode(tile, img, y0, x0);
    {   // prologue of the stream: patch of (first tile, chunk 0) + weights of its (chunk 0, tap 0)
        gvfi_i32x4 srd;
        unsigned soff;
        int ld;
        patch_srd(0, img, y0, x0, srd, soff, ld);
#pragma unroll
        for (int q = 0; q < QP; ++q) issue_patch(q, patch_off(q, ld, y0, x0), 0u, srd, soff);
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i) issue_b(0u, 0u, i);
    }
    f32x16 acc[MI][NI];
    uint4 fa[2][MI], fb[2][NI];
    auto load_frags = [&](int toff, int kk, int buf, const unsigned (&pb)[KK]) {
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[buf][i] = *(const uint4*)(smem + pa[i] + toff + kk * 32);
#pragma unroll
        for (int j = 0; j < NI; ++j) fb[buf][j] = *(const uint4*)(smem + pb[kk] + j * 32 * RB);
    };
    // chunk boundary: the other patch buffer, the weight stages of even and odd taps change places
    auto next_chunk_bases = [&](unsigned gc_new) {
        const unsigned da = (gc_new & 1) ? (unsigned)P3_PATCH : 0u - (unsigned)P3_PATCH;
#pragma unroll
        for (int i = 0; i < MI; ++i) pa[i] += da;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const unsigned t = pbe[kk];
            pbe[kk] = pbo[kk];
            pbo[kk] = t;
        }
    };
    glds_wait_n<0>();
    __syncthreads();
    if (PROF) { const unsigned long long t = now(); ph[0] = t - tprev; tprev = t; }
    // every slope of both activations <= 1?  (each wave looks at all 2 x 256 table entries: 4 per lane and table)
    bool fast_act;
    {
        const float4 sa = *(const float4*)(ptab + 256 + lane * 4), sb = *(const float4*)(ptab + 512 + lane * 4);
        fast_act = wave_all(sa.x <= 1.f && sa.y <= 1.f && sa.z <= 1.f && sa.w <= 1.f && sb.x <= 1.f && sb.y <= 1.f && sb.z <= 1.f && sb.w <= 1.f);
    }
    load_frags(0, 0, 0, pbe);
    unsigned gc = 0;      // chunks this workgroup has been through: patch buffer gc & 1, weight stage of tap t (gc + t) & 1
    unsigned ntiles_done = 0;
    for (;;) {
        const int tnext = tile + a.slots;
        const bool has_next = tnext < t_last;
        int imgn = img, y0n = y0, x0n = x0;
        if (has_next) decode(tnext, imgn, y0n, x0n);
        for (int c = 0; c < a.chunks; ++c, ++gc) {
            const bool last = c + 1 == a.chunks;
            // what streams in during this chunk: the patch of this tile's next chunk, or of the next tile's chunk 0
            const bool do_next = !last || has_next;
            gvfi_i32x4 srd_n;
            unsigned soff_n;
            int ld_n;
            patch_srd(last ? 0 : c + 1, last ? imgn : img, last ? y0n : y0, last ? x0n : x0, srd_n, soff_n, ld_n);
            const int y0p = last ? y0n : y0, x0p = last ? x0n : x0;
            // first chunk of a tile (its first MFMAs take C = 0).  Read through an opaque copy of the chunk index: tied to the
            // induction variable hipcc peels the first chunk into a second copy of the whole K loop (and spills in it)
            int c_o = c;
            GVFI_OPAQUE_S(c_o);
            const bool first = c_o == 0;
            unsigned noff = patch_off(0, ld_n, y0p, x0p);      // (tap 0's piece: the one offset computed inside the MFMA stream)
            auto step = [&](auto tap_tag) {
                constexpr int tap = decltype(tap_tag)::value;
                constexpr int toff = ((tap / 3) * P3_PW + (tap % 3)) * P3_PITCH;
                constexpr int ntap = tap == 8 ? 0 : tap + 1;
                constexpr int ntoff = ((ntap / 3) * P3_PW + (ntap % 3)) * P3_PITCH;
                const unsigned (&pb)[KK] = (tap & 1) ? pbo : pbe;
                const unsigned wsoff_n = tap < 8 ? (unsigned)(c * 9 + tap + 1) * wstride : (last ? 0u : (unsigned)((c + 1) * 9) * wstride);
                const bool tile_end = tap == 8 && last;
#pragma unroll
                for (int kk = 0; kk + 1 < KK; ++kk) {
                    load_frags(toff, kk + 1, (kk + 1) & 1, pb);
                    GVFI_SCHED_BARRIER();
                    if (tap == 0 && kk == 0 && first) {
                        // first MFMA of a tile into every accumulator block: C = 0
#pragma unroll
                        for (int i = 0; i < MI; ++i) {
#pragma unroll
                            for (int j = 0; j < NI; ++j) {
#pragma unroll
                                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
                                Mma2<T>::run(acc[i][j], fb[0][j], fa[0][i]);
                            }
                            if (i == 0) {
#pragma unroll
                                for (int q = 0; q < B_INSTR; ++q) issue_b(wsoff_n, (gc + tap + 1) & 1, q);
                                issue_patch(tap, noff, (gc + 1) & 1, srd_n, soff_n);
                            }
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < MI; ++i) {
#pragma unroll
                            for (int j = 0; j < NI; ++j) Mma2<T>::run(acc[i][j], fb[kk & 1][j], fa[kk & 1][i]);
                            if (kk == 0 && i == 0 && (tap < 8 || do_next)) {
#pragma unroll
                                for (int q = 0; q < B_INSTR; ++q) issue_b(wsoff_n, (gc + tap + 1) & 1, q);
                                if (tap < 6 && tap < QP && do_next) issue_patch(tap, noff, (gc + 1) & 1, srd_n, soff_n);
                            }
                        }
                    }
                    GVFI_SCHED_BARRIER();
                }
                // next step's piece: in front of the barrier (same-binary A/B against a place behind the step's last MFMAs: equal)
                if (tap + 1 < 6 && tap + 1 < QP) noff = patch_off(tap + 1, ld_n, y0p, x0p);
                {
                    unsigned long long tw = 0;
                    if (PROF) tw = now();
                    glds_wait_n<0>();
                    __syncthreads();
                    if (PROF) ph[2] += now() - tw;
                }
                if (!tile_end) {
                    if (tap == 8) {
                        next_chunk_bases(gc + 1);
                        load_frags(ntoff, 0, KK & 1, pbe);
                    } else {
                        load_frags(ntoff, 0, KK & 1, (tap & 1) ? pbe : pbo);
                    }
                    GVFI_SCHED_BARRIER();
                }
#pragma unroll
                for (int i = 0; i < MI; ++i) {
#pragma unroll
                    for (int j = 0; j < NI; ++j) Mma2<T>::run(acc[i][j], fb[(KK - 1) & 1][j], fa[(KK - 1) & 1][i]);
                }
                GVFI_SCHED_BARRIER();
            };
            step(std::integral_constant<int, 0>{});
            step(std::integral_constant<int, 1>{});
            step(std::integral_constant<int, 2>{});
            step(std::integral_constant<int, 3>{});
            step(std::integral_constant<int, 4>{});
            step(std::integral_constant<int, 5>{});
            step(std::integral_constant<int, 6>{});
            step(std::integral_constant<int, 7>{});
            step(std::integral_constant<int, 8>{});
        }
        // ---- the tile's epilogue: staging in what its last step read (patch buffer / weight stage (gc - 1) & 1)
        if (PROF) { const unsigned long long t = now(); ph[1] += t - tprev; tprev = t; }
        {
            const unsigned par = (gc - 1) & 1;
            const unsigned reg0 = wave < 4 ? par * P3_PATCH + (unsigned)wave * 8192u : 2 * P3_PATCH + par * P3_BSTAGE + (unsigned)(wave - 4) * 8192u;
            // (the epilogue's per-lane addresses are tile-independent: behind an opaque copy of the lane index hipcc re-derives
            // them here -- ~30 VALU operations per tile -- instead of keeping two dozen registers alive across the K loop)
            int lane_e = lane;
            GVFI_OPAQUE_V(lane_e);
            P3Epi ep;
            {
                const uint4 e0 = *(const uint4*)(smem + PTAB + 3072), e1 = *(const uint4*)(smem + PTAB + 3072 + 16);
                ep.y = (const void*)((unsigned long long)P3_UNIFORM(e0.x) | ((unsigned long long)P3_UNIFORM(e0.y) << 32));
                ep.res = (const void*)((unsigned long long)P3_UNIFORM(e0.z) | ((unsigned long long)P3_UNIFORM(e0.w) << 32));
                ep.ldy = (int)P3_UNIFORM(e1.x);
                ep.ldr = (int)P3_UNIFORM(e1.y);
                ep.H = (int)P3_UNIFORM(e1.z);
                ep.W = (int)P3_UNIFORM(e1.w);
                ep.out_scale = 1.0f;
                ep.nostore = PROF ? (a.dbg & 2) : 0;
            }
            const long long pix0 = ((long long)img * ep.H + y0) * ep.W + x0;
            const bool has_res = ep.res != nullptr;
            if (has_res) {
                if (fast_act) p3_wave_epilogue<true, false, true>(ep, acc, smem, smem_lds, reg0, ptab, lane_e, wm, wn, pix0, y0, x0, 0);
                else p3_wave_epilogue<true, false, false>(ep, acc, smem, smem_lds, reg0, ptab, lane_e, wm, wn, pix0, y0, x0, 0);
            } else {
                if (fast_act) p3_wave_epilogue<false, false, true>(ep, acc, smem, smem_lds, reg0, ptab, lane_e, wm, wn, pix0, y0, x0, 0);
                else p3_wave_epilogue<false, false, false>(ep, acc, smem, smem_lds, reg0, ptab, lane_e, wm, wn, pix0, y0, x0, 0);
            }
        }
        if (PROF) ph[4] += now() - tprev;      // (the wave's own epilogue, without the barrier behind it)
        ++ntiles_done;
        if (!has_next) {
            if (PROF) { const unsigned long long t = now(); ph[3] += t - tprev; tprev = t; }
            break;
        }
        // the next tile's first fragments: published since the last step's barrier, read while the other waves finish
        init_bases(gc & 1);
        load_frags(0, 0, 0, pbe);
        __syncthreads();      // the staging areas are the DMA targets of the next tile's first step
        if (PROF) { const unsigned long long t = now(); ph[3] += t - tprev; tprev = t; }
        tile = tnext;
        img = imgn;
        y0 = y0n;
        x0 = x0n;
    }
    if (PROF && tid == 0) {
        // per workgroup (8 words): cycles summed over its tiles; word 0 carries the tile count in its upper half
        unsigned long long* o = (unsigned long long*)p.aux1 + (size_t)bid * 8;
        o[0] = ph[0] | ((unsigned long long)ntiles_done << 32);
        o[1] = ph[1];
        o[2] = ph[2];
        o[3] = ph[3];
        o[4] = ph[4];
        o[5] = o[6] = o[7] = 0;
    }
}

// workgroups of the stream kernel per XCD: one per compute unit (the LDS admits one)
static int p3_slots_per_xcd() {
#ifndef GVFI_HOSTSIM
    static int cached[16] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 15;
    if (cached[dev] == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
        cached[dev] = cus / 8;
    }
    return cached[dev];
#else
    return 1;        // (emulator: 8 persistent workgroups -- small test images still give every workgroup several tiles)
#endif
}

// 1 = gvfi_conv2d routes this problem here ahead of the LDS-DMA kernel; 2 = runnable on request (algo 4) but too few
// output pixels for the 16 x 16 tiles to pay; 0 = not this kernel's problem
extern "C" int gvfi_conv2d_p3x3_eligible(const gvfi_conv_params* pp) {
    const gvfi_conv_params& p = *pp;
    if (p.dtype != GVFI_BF16 || p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad_h != 1 || p.pad_w != 1) return 0;
    if (p.pad_mode != GVFI_PAD_ZEROS || p.groups > 1 || p.epi_mode != GVFI_EPI_STD || p.w_layout != 1 || p.stats != nullptr) return 0;
    if (p.c0 <= 0 || (p.c0 % 64) || (p.c1 % 64) || p.Cout <= 0 || (p.Cout % 256)) return 0;
    if (p.Ho != p.H || p.Wo != p.W) return 0;
    if (p.y_f32 || (p.res != nullptr && p.res_f32) || p.act1 > GVFI_ACT_PRELU || p.act2 > GVFI_ACT_PRELU) return 0;
    if ((((uintptr_t)p.y) & 15) || ((p.ldy * 2) & 15) || (p.res && ((((uintptr_t)p.res) & 15) || ((p.ldr * 2) & 15)))) return 0;
    if (((uintptr_t)p.x0 & 15) || ((uintptr_t)p.x1 & 15) || ((uintptr_t)p.w & 15) || (p.ld0 % 8) || (p.c1 > 0 && (p.ld1 % 8))) return 0;
    // per-lane DMA offsets are 32-bit and stay below the descriptor's range: 18 image rows of the widest source
    if ((long long)18 * p.W * (p.ld0 > p.ld1 ? p.ld0 : p.ld1) * 2 >= 0x7fffff00ll) return 0;
    if (p.res && (long long)18 * p.W * p.ldr * 2 >= 0x7fffff00ll) return 0;               // ... and the residual tile's rows
    if ((long long)18 * p.W * p.ldy * 2 >= 0x7fffff00ll) return 0;                        // ... and the output tile's (register-side epilogue)
    return (long long)p.N * p.H * p.W >= 65536 ? 1 : 2;
}

// the launch geometry and form of a problem (shared by the launcher and gvfi_conv2d_p3x3_form)
static int p3_plan(const gvfi_conv_params& p, P3Args& a) {
    a.p = p;
    a.chunks0 = p.c0 / 64;
    a.chunks = (p.c0 + p.c1) / 64;
    a.tiles_x = cdiv(p.W, P3_TW);
    a.tiles_y = cdiv(p.H, P3_TH);
    a.mtiles = a.tiles_x * a.tiles_y * p.N;
    a.ntiles_n = p.Cout / 256;
    a.per_xcd = cdiv((long long)a.mtiles * a.ntiles_n, 8);
    // algo bits 13, 14: 0 = auto (the stream kernel where it applies and every CU gets at least two tiles, else the tile-per-
    // workgroup kernel with the wave-private epilogue); A/B switches: 1 = tile per workgroup, workgroup-wide epilogue (the
    // round-2 kernel), 2 = tile per workgroup, wave-private epilogue, 3 = stream kernel whenever it applies
    const int var = (p.algo >> 13) & 3;
    const bool stream_ok = a.ntiles_n == 1 && p.out_scale == 1.0f && 18 * p.W + 18 < (1 << 17);
    a.slots = p3_slots_per_xcd();
    if (a.slots > a.per_xcd) a.slots = a.per_xcd;
    a.dbg = 0;
    if (stream_ok && (var == 3 || (var == 0 && a.per_xcd >= 2 * a.slots))) return 3;
    return var == 1 ? 1 : 2;
}

extern "C" int gvfi_conv2d_p3x3_form(const gvfi_conv_params* pp) {
    if (!gvfi_conv2d_p3x3_eligible(pp)) return 0;
    P3Args a;
    return p3_plan(*pp, a);
}

extern "C" int gvfi_conv2d_p3x3(const gvfi_conv_params* pp, void* stream) {
    if (!gvfi_conv2d_p3x3_eligible(pp)) return -2;
    const gvfi_conv_params& p = *pp;
    P3Args a;
    const int form = p3_plan(p, a);
    const bool prof = ((p.algo >> 8) & 128) && p.aux1 != nullptr;
    if (form == 3) {
        a.dbg = prof ? (p.tile_hint >> 10) & 3 : 0;
        if (prof) GVFI_LAUNCH_COOP(conv_p3x3_stream_kernel<true>, dim3(a.slots * 8), dim3(512), (hipStream_t)stream, a);
        else GVFI_LAUNCH_COOP(conv_p3x3_stream_kernel<false>, dim3(a.slots * 8), dim3(512), (hipStream_t)stream, a);
        return (int)hipGetLastError();
    }
#define P3_LAUNCH(PR, V) GVFI_LAUNCH_COOP((conv_p3x3_kernel<PR, V>), dim3(a.per_xcd * 8), dim3(512), (hipStream_t)stream, a)
    if (prof) {
        if (form == 2) P3_LAUNCH(true, 1); else P3_LAUNCH(true, 0);
    } else {
        if (form == 2) P3_LAUNCH(false, 1); else P3_LAUNCH(false, 0);
    }
#undef P3_LAUNCH
    return (int)hipGetLastError();
}
