// One half of the SepConvGRU of the flow estimators' update block as ONE launch (round 5; raft/update.py:58-73,
// FlowFormer gru.py:130-160):
//     z = sigmoid(conv_z([h | x]))   r = sigmoid(conv_r([h | x]))   q = tanh(conv_q([r * h | x]))   h' = (1 - z) * h + z * q
// with 1 x 5 (horizontal half) or 5 x 1 (vertical half) filters.  As two launches of the weights-direct LDS-DMA kernel
// (conv_igemm_glds.hip: z | r with the GRU_ZR epilogue, q with GRU_Q) every iteration pays two prologues, two epilogues that
// move z and r * h through HBM, a launch gap, and a K loop that re-stages the same pixels once per filter tap behind a
// barrier.  A separable filter has no halo ACROSS lines, so a workgroup that owns whole lines of the image needs nobody else's
// r * h:
//   * tile = 64 output pixels = ONE image row (horizontal half, W <= 64) or TWO image columns (vertical half, H <= 32);
//   * the tile's input -- [h | x] of its pixels plus the two zero-padded positions beyond either end of each line -- is
//     staged ONCE by LDS-DMA as 64-channel planes [line position][128 B] (the slot swizzle of conv_igemm_glds.hip); a filter
//     tap is a row offset into a plane, so the three contractions (z, r, q) walk K with NO barrier and NO further pixel
//     traffic: only the weight fragments stream, from the fragment-ordered image (w_layout 2) through a 4-deep register ring;
//   * z stays in registers, r * h goes to LDS planes of the same layout (the q contraction reads it instead of h), h' leaves
//     through a staging tile as whole 16-byte units.
// K order, operand rounding (z and r * h rounded to the 16-bit activation type, as the two-launch path stores them) and the
// gate arithmetic (conv_mma.h: fast_sigmoid / fast_tanh, (acc + context term) + bias) are those of the two launches: the
// result is bit-identical to them (tests/kernel_cases.py:gru_fused_case, emulator + GPU).
#include "conv_mma.h"
#include "gimmvfi_experiments.h"
#include <type_traits>

struct GruArgs {
    gvfi_gru_params p;
    int tiles_per_img;      // H (horizontal) or ceil(W / 2) (vertical)
    int ntiles;
};

#define GRU_ROWS 80                    // LDS rows of a plane: 2 segments of 36 (or one of 68) + slack, a multiple of 8
#define GRU_PLANE (GRU_ROWS * 128)

// (two workgroups per CU: 242 registers with the budget stated, 282 without it -- hipcc then parks values in AGPRs and halves
// the occupancy)
#ifndef GVFI_HOSTSIM
#define GRU_KERNEL_ATTRS __attribute__((amdgpu_waves_per_eu(2, 2)))
#else
#define GRU_KERNEL_ATTRS
#endif
template <typename T, int CX> __global__ void __launch_bounds__(256) GRU_KERNEL_ATTRS gru_half_kernel(GruArgs a) {
    constexpr int NCH = 2 + CX;                 // 64-channel chunks of [h | x]
    constexpr int KT = NCH * 5;                 // K chunks: channel chunk outer, filter tap inner (the weight image's order)
    constexpr int KK = 4;                       // MFMA k-steps per chunk
    constexpr int NPL = NCH + 2;                // planes: [h0 h1 x0 .. | rh0 rh1]
    __shared__ __attribute__((aligned(16))) unsigned char smem[NPL * GRU_PLANE];
    const gvfi_gru_params& p = a.p;
    const int tile = blockIdx.x;
    if (tile >= a.ntiles) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, yy = lane & 31;
    const int vert = p.vertical;
    const int img = tile / a.tiles_per_img, tl = tile - img * a.tiles_per_img;
    const int L = vert ? p.H : p.W;             // line length
    const int SEGR = vert ? 36 : 68;            // LDS rows per line segment (line positions -2 .. L + 1)
    const int SEGI = vert ? 36 : 32;            // LDS row distance of the two 32-row MFMA blocks
    const long long img_pix = (long long)img * p.H * p.W;
    // pixel (relative to the image) of line position c of segment s; validity of the segment itself
    auto seg_ok = [&](int s) { return vert ? (2 * tl + s) < p.W : s == 0; };
    auto pix_of = [&](int s, int c) { return vert ? (long long)c * p.W + (2 * tl + s) : (long long)tl * p.W + c; };
    auto swz = [](int row) { return (row >> 1) & 7; };

    // ---- stage [h | x] of the tile: NCH planes x GRU_ROWS rows x 8 slots, one LDS-DMA instruction = 8 rows of a plane
    const unsigned smem_lds = lds_address(smem);
    {
        const gvfi_i32x4 srd_h = make_srd((const T*)p.h + img_pix * p.ldh);
        const gvfi_i32x4 srd_x = make_srd((const T*)p.x + img_pix * p.ldx);
        constexpr int PIECES = NCH * (GRU_ROWS / 8);
        for (int pc = wave; pc < PIECES; pc += 4) {
            const int pl = pc / (GRU_ROWS / 8), r8 = pc - pl * (GRU_ROWS / 8);
            const int row = r8 * 8 + (lane >> 3), slot = lane & 7;
            const int s = row / SEGR, c = row - s * SEGR - 2;
            const bool ok = s < (vert ? 2 : 1) && seg_ok(s) && c >= 0 && c < L;
            const bool from_h = pl < 2;
            const int ld = from_h ? p.ldh : p.ldx;
            const unsigned off = ok ? (unsigned)((pix_of(s, c) * ld + (from_h ? pl : pl - 2) * 64 + ((slot ^ swz(row)) << 3)) * 2) : GVFI_DMA_OOB;
            bufdma16(off, from_h ? srd_h : srd_x, 0u, smem_lds + pl * GRU_PLANE + r8 * 1024);
        }
        // the r * h planes start as zeros (line ends, idle rows)
        uint4 z4;
        z4.x = z4.y = z4.z = z4.w = 0u;
        for (int i = tid; i < 2 * GRU_PLANE / 16; i += 256) *(uint4*)(smem + NCH * GRU_PLANE + i * 16) = z4;
    }
    // ---- per-lane constants
    const int col = wave * 32 + yy;                      // this lane's output channel within a 128-column pass
    // LDS row of the lane's A-fragment pixel (tap 0) in MFMA block i, and of accumulator element (i, r)'s pixel (centre tap)
    int lrow0[2];
    lrow0[0] = yy;
    lrow0[1] = SEGI + yy;
    auto acc_row = [&](int r) { return 8 * (r >> 2) + (r & 3) + 4 * half; };     // row of accumulator register r inside a block
    // validity + pixel index of accumulator element (i, r)
    auto elem_pix = [&](int i, int r, bool& ok) -> long long {
        const int rr = acc_row(r);
        if (vert) {
            ok = seg_ok(i) && rr < p.H;
            return img_pix + pix_of(i, rr);
        }
        const int c = i * 32 + rr;
        ok = c < p.W;
        return img_pix + pix_of(0, c);
    };
    // 2-byte element (channel ch of 128) of plane pair `base` at LDS row `row`
    auto elem_addr = [&](int base, int row, int ch) {
        return (base + (ch >> 6)) * GRU_PLANE + row * 128 + ((((ch & 63) >> 3) ^ swz(row)) << 4) + (ch & 7) * 2;
    };

    const size_t wf_chunk_zr = (size_t)8 * KK * 1024, wf_chunk_q = (size_t)4 * KK * 1024;     // bytes per K chunk (8 / 4 column blocks)
    glds_wait_n<0>();
    __syncthreads();

    float zreg[2][16];
    f32x16 acc[2];
    // fragment addresses inside a plane: tap t of MFMA block i reads LDS row lrow0[i] + t; its swizzle term is per (tap, block)
    unsigned fbase[5][2], fswz[5][2];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            fbase[t][i] = (unsigned)((lrow0[i] + t) * 128);
            fswz[t][i] = (unsigned)swz(lrow0[i] + t);
        }
    // per accumulator element (i, r): pixel validity and the element's pixel index, clamped to the image's first pixel when
    // invalid (loads stay inside the tensors; LDS rows of invalid pixels exist and hold zeros, so nothing needs a branch)
    // three 128-column contractions over the same staged pixels: pass 0 = z, 1 = r (weights wzr, column blocks 0-3 / 4-7), 2 = q
    // (weights wq; the r * h planes in place of h).  ONE copy of the code (a run-time pass loop, a run-time loop over the channel
    // chunks, the five taps of a chunk unrolled with the ring slot = tap): a wave carries the ring of weight fragments (5 chunks
    // = 80 registers, refilled four chunks ahead), the accumulators and z.
#pragma unroll 1
    for (int pass = 0; pass < 3; ++pass) {
        const unsigned char* wbase = (const unsigned char*)(pass == 2 ? p.wq : p.wzr) + ((size_t)((pass == 1 ? 4 : 0) + wave) * KK) * 1024 + lane * 16;
        const size_t wf_chunk = pass == 2 ? wf_chunk_q : wf_chunk_zr;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        uint4 bq[5][KK];
        auto load_b = [&](int kt, int slot) {
            const unsigned char* src = wbase + (size_t)kt * wf_chunk;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) bq[slot][kk] = *(const uint4*)(src + kk * 1024);
        };
#pragma unroll
        for (int t = 0; t < 5; ++t) load_b(t, t);
        // (the refill of a slot is UNCONDITIONAL inside the loop and the last channel chunk is a copy of the body without it:
        // behind a run-time `if` hipcc's counted waits assume the path on which nothing was issued and drain the whole ring at
        // the end of every channel chunk -- vmcnt(0) in front of the chunk's last MFMAs)
        auto group = [&](int ck, auto more_tag) {
            constexpr bool MORE = decltype(more_tag)::value;
            const int pl = (pass == 2 && ck < 2) ? NCH + ck : ck;
            const unsigned char* pb = smem + pl * GRU_PLANE;
#pragma unroll
            for (int t = 0; t < 5; ++t) {
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) {
                    const uint4 f0 = *(const uint4*)(pb + fbase[t][0] + (((unsigned)(2 * kk + half) ^ fswz[t][0]) << 4));
                    const uint4 f1 = *(const uint4*)(pb + fbase[t][1] + (((unsigned)(2 * kk + half) ^ fswz[t][1]) << 4));
                    Mma2<T>::run(acc[0], f0, bq[t][kk]);
                    Mma2<T>::run(acc[1], f1, bq[t][kk]);
                }
                GVFI_SCHED_BARRIER();
                if constexpr (MORE) load_b((ck + 1) * 5 + t, t);          // this slot's next occupant: the same tap of the next channel chunk
                GVFI_SCHED_BARRIER();
            }
        };
#pragma unroll 1
        for (int ck = 0; ck + 1 < NCH; ++ck) group(ck, std::true_type{});
        group(NCH - 1, std::false_type{});
        // (the epilogue's per-element pixel indices and LDS addresses do not depend on the pass: left alone hipcc computes all of
        // them ONCE in front of the pass loop and carries ~130 registers through the three K loops -- occupancy 1.  They are
        // made to depend on values the compiler cannot see through.)
        int half_e = half, col_e = col;
        GVFI_OPAQUE_V(half_e);
        GVFI_OPAQUE_V(col_e);
        auto acc_row_e = [&](int r) { return 8 * (r >> 2) + (r & 3) + 4 * half_e; };
        const float* ctx = pass == 2 ? p.ctx_q : p.ctx_zr;
        const int ldc = pass == 2 ? p.ld_cq : p.ld_czr;
        const int ccol = (pass == 1 ? 128 : 0) + col_e;
        const float* bias = pass == 2 ? p.bq : p.bzr;
        const float bb = bias ? bias[ccol] : 0.f;
        if (pass == 2) __syncthreads();        // every wave is done reading the r * h planes: they become the staging tile [64 rows][128 channels]
        uint16_t* stg = (uint16_t*)(smem + NCH * GRU_PLANE);
        // the context terms of all 32 elements are requested TOGETHER (one memory round trip; the ring's 80 registers are free
        // by now) -- fetched four at a time between scheduling fences they cost eight dependent round trips per pass, and the
        // first GPU A/B of this kernel lost 4 % to the two launches it replaces
        float cv[2][16];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = acc_row_e(r);
                bool ok;
                long long pix;
                if (vert) {
                    ok = seg_ok(i) && rr < p.H;
                    pix = img_pix + pix_of(i, rr);
                } else {
                    ok = i * 32 + rr < p.W;
                    pix = img_pix + pix_of(0, i * 32 + rr);
                }
                cv[i][r] = ctx ? ctx[(ok ? pix : img_pix) * ldc + ccol] : 0.f;
            }
        GVFI_SCHED_BARRIER();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = acc_row_e(r);
                float v = acc[i][r];
                if (ctx) v += cv[i][r];
                const float t = v + bb;
                const int row = i * SEGI + rr + 2;
                if (pass == 0) {
                    // z: stays in registers, rounded as the two-launch path stores (and re-reads) it
                    zreg[i][r] = cvt16<T>(pack16x2<T>(fast_sigmoid(t), 0.f) & 0xffffu);
                } else {
                    const float hh = cvt16<T>(*(const uint16_t*)(smem + elem_addr(0, row, col_e)));
                    if (pass == 1) {
                        // r * h -> LDS (rows of invalid pixels hold h = 0: they write zeros)
                        *(uint16_t*)(smem + elem_addr(NCH, row, col_e)) = (uint16_t)(pack16x2<T>(fast_sigmoid(t) * hh, 0.f) & 0xffffu);
                    } else {
                        const float zz = zreg[i][r];
                        const float hn = (1.f - zz) * hh + zz * fast_tanh(t);
                        stg[(i * 32 + rr) * 128 + col_e] = (uint16_t)(pack16x2<T>(hn, 0.f) & 0xffffu);
                    }
                }
            }
        if (pass == 1) __syncthreads();        // r * h of every column (all four waves) visible
    }
    __syncthreads();
    // ---- h' rows -> global, 16-byte units: 64 rows x 16 units / 256 threads
    for (int u = tid; u < 64 * 16; u += 256) {
        const int row = u >> 4, sl = u & 15;
        const int i = row >> 5, rr = row & 31;
        bool ok;
        long long pix;
        if (vert) {
            ok = seg_ok(i) && rr < p.H;
            pix = img_pix + pix_of(i, rr);
        } else {
            ok = row < p.W;
            pix = img_pix + pix_of(0, row);
        }
        if (ok) *(uint4*)((T*)p.out + pix * p.ldo + sl * 8) = *(const uint4*)(smem + NCH * GRU_PLANE + row * 256 + sl * 16);
    }
}

// 1 = gvfi_gru_half takes this problem
extern "C" int gvfi_gru_half_ok(const gvfi_gru_params* pp) {
    const gvfi_gru_params& p = *pp;
    if (p.dtype != GVFI_BF16 && p.dtype != GVFI_F16) return 0;
    if (p.cx != 128 && p.cx != 256) return 0;
    if (p.N <= 0 || p.H <= 0 || p.W <= 0) return 0;
    if (p.vertical ? p.H > 32 : p.W > 64) return 0;
    if ((((uintptr_t)p.h | (uintptr_t)p.x | (uintptr_t)p.out | (uintptr_t)p.wzr | (uintptr_t)p.wq) & 15) || (p.ldh % 8) || (p.ldx % 8) || (p.ldo % 8)) return 0;
    if (p.ldh < 128 || p.ldx < p.cx || p.ldo < 128) return 0;
    if ((p.ctx_zr && p.ld_czr < 256) || (p.ctx_q && p.ld_cq < 128)) return 0;
    // 32-bit DMA offsets inside one image
    if ((long long)p.H * p.W * (p.ldh > p.ldx ? p.ldh : p.ldx) * 2 >= 0x7fffff00ll) return 0;
    return 1;
}

extern "C" int gvfi_gru_half(const gvfi_gru_params* pp, void* stream) {
    if (!gvfi_gru_half_ok(pp)) return -2;
    GruArgs a;
    a.p = *pp;
    a.tiles_per_img = pp->vertical ? (pp->W + 1) / 2 : pp->H;
    a.ntiles = a.tiles_per_img * pp->N;
    const dim3 grid((unsigned)a.ntiles), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (pp->dtype == GVFI_F16) {
        if (pp->cx == 256) { GVFI_LAUNCH_COOP((gru_half_kernel<f16_t, 4>), grid, block, st, a); }
        else { GVFI_LAUNCH_COOP((gru_half_kernel<f16_t, 2>), grid, block, st, a); }
    } else {
        if (pp->cx == 256) { GVFI_LAUNCH_COOP((gru_half_kernel<bf16_t, 4>), grid, block, st, a); }
        else { GVFI_LAUNCH_COOP((gru_half_kernel<bf16_t, 2>), grid, block, st, a); }
    }
    return (int)hipGetLastError();
}
