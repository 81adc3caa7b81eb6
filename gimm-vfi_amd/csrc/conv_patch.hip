// Direct ("patch") convolution on the matrix cores for stride-1 'same' convolutions.
//
// The implicit-GEMM kernel of conv_igemm_glds.hip re-stages the activation tile for every filter tap
// (KH*KW times, from L2), and its LDS-DMA bandwidth per CU (~70 GB/s measured) is what bounds it.  Here a
// workgroup owns a 2-D output tile of TH x 32 pixels and stages the (TH+KH-1) x (32+KW-1) input PATCH of one
// channel chunk ONCE; the KH*KW taps then read their A fragments from that patch at shifted row offsets
// (one MFMA row block = one image row of 32 pixels, so a shift is a constant LDS row offset).  Only the
// weight tile is streamed per (tap, chunk) step.  L2->LDS bytes per MFMA drop ~1.75x for 3x3 (256-wide tile)
// and ~5x for small-channel full-resolution layers, which were DMA / overhead bound.
//
//   K order : channel chunk outer, tap inner   (K-step s = chunk*taps + tap)
//   LDS     : [patch buffers (1 or 2)] [weight tile x2]   (dynamic), rows of KB bytes (128: 64 bf16 / 64: 32 bf16),
//             16-byte slots XOR-swizzled per row exactly as in conv_igemm_glds.hip (source-side for the DMA)
//   pipeline: weight tile of step s+1 and a slice of the NEXT chunk's patch are DMA'd between the MFMA groups of
//             step s; one s_waitcnt vmcnt(0) + barrier per step
//   epilogue: LDS-staged, 16-byte coalesced stores (conv_mma.h)
#include "conv_mma.h"

#define PATCH_MAX_AI 10   // max LDS-DMA instructions per wave for one patch

struct PatchArgs {
    gvfi_conv_params p;
    int chunks0, chunks;   // channel chunks from source 0 / total
    int taps;              // KH*KW
    int S;                 // K steps = chunks * taps
    int PH, PW;            // patch height / width in pixels
    int n_ai;              // patch DMA instructions per wave
    int pp;                // patch DMA instructions issued per step while prefetching the next chunk
    int npb;               // patch buffers (1: reload between chunks, 2: prefetch during the previous chunk)
    int patch_bytes;       // bytes of one patch buffer
    int lds_bytes;         // total dynamic LDS
    int tiles_x, tiles_y;  // output tiles per image
    int MT, NT, per_xcd;
    long long Ktot;        // weight row length in elements (w_layout 0)
};

template <typename T, int TH, int BN, int WAVES_M, int WAVES_N, int KB>
__global__ void __launch_bounds__(64 * WAVES_M * WAVES_N) conv_patch_kernel(PatchArgs a) {
    constexpr int VE = Elem<T>::VE;
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int NTHR = 64 * NW;
    constexpr int BM = TH * 32;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int SL = KB / 16;              // 16-byte slots per LDS row
    constexpr int RPI = 64 / SL;             // rows written by one LDS-DMA instruction
    constexpr int BKE = KB / (int)sizeof(T); // channels per chunk
    constexpr int KK = KB / 32;              // MFMA k-steps per chunk
    constexpr int B_TOTAL = BN / RPI;        // DMA instructions for the weight tile
    constexpr int BI = (B_TOTAL + NW - 1) / NW;
    constexpr int NSLOT = KK * MI;
    static_assert(MI >= 1 && NI >= 1 && BN % RPI == 0, "tile");
    auto swz = [](int row) { return KB == 128 ? ((row >> 1) & 7) : ((row >> 2) & 3); };

    GVFI_DYN_SMEM(smem);
    const gvfi_conv_params& p = a.p;
    const int bid = blockIdx.x;
    const int v = (bid & 7) * a.per_xcd + (bid >> 3);   // XCD-aware tile order
    if (v >= a.MT * a.NT) return;
    const int mt = v / a.NT, nt = v - mt * a.NT;
    const int tiles_img = a.tiles_x * a.tiles_y;
    const int img = mt / tiles_img;
    const int trem = mt - img * tiles_img;
    const int tyi = trem / a.tiles_x, txi = trem - tyi * a.tiles_x;
    const int y0 = tyi * TH, x0 = txi * 32;
    const int n0 = nt * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int lrow = lane / SL, lslot = lane % SL;
    const int frow = lane & 31, fhalf = lane >> 5;
    const T* __restrict__ x0p = (const T*)p.x0;
    const T* __restrict__ x1p = (const T*)p.x1;
    const T* __restrict__ wg = (const T*)p.w;
    const unsigned char* zero = (const unsigned char*)gvfi_zero_page;

    unsigned char* patch0 = smem;
    unsigned char* wbuf0 = smem + a.npb * a.patch_bytes;

    // ---- per-thread patch rows: pixel index (or -1) and source k-group of each DMA instruction of this wave
    int a_pix[PATCH_MAX_AI], a_koff[PATCH_MAX_AI], a_step[PATCH_MAX_AI];
    const int nrows = a.PH * a.PW;
#pragma unroll
    for (int i = 0; i < PATCH_MAX_AI; ++i) {
        const int q = (i * NW + wave) * RPI + lrow;
        int pix = -1;
        if (i < a.n_ai && q < nrows) {
            const int py = q / a.PW, px = q - py * a.PW;
            const int iy = y0 - p.pad_h + py, ix = x0 - p.pad_w + px;
            if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) pix = (img * p.H + iy) * p.W + ix;
        }
        a_pix[i] = pix;
        a_koff[i] = (lslot ^ swz(q)) * VE;
        a_step[i] = i / a.pp;   // tap index (within the previous chunk) at which this piece is prefetched
    }
    // ---- per-thread weight rows
    long long b_base[BI];
    bool b_ok[BI];
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int row = (i * NW + wave) * RPI + lrow;
        const int n = n0 + row;
        b_ok[i] = (i * NW + wave) < B_TOTAL && n < p.Cout;
        if (p.w_layout == 1) b_base[i] = ((long long)n * SL + lslot) * VE;            // chunk-major image
        else b_base[i] = (long long)n * a.Ktot + (lslot ^ swz(row)) * VE;             // [Cout][K]
    }

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // chunk-uniform source selection
    auto chunk_src = [&](int c, const T*& xs, int& ld, int& cbase) {
        const bool from0 = c < a.chunks0;
        xs = from0 ? x0p : x1p;
        ld = from0 ? p.ld0 : p.ld1;
        cbase = (from0 ? c : c - a.chunks0) * BKE;
    };
    auto patch_piece = [&](int i, int c) {   // i compile-time after unrolling
        const T* xs; int ld, cbase;
        chunk_src(c, xs, ld, cbase);
        const T* sp = xs + (long long)a_pix[i] * ld + (cbase + a_koff[i]);
        glds16(a_pix[i] >= 0 ? (const void*)sp : (const void*)zero,
               patch0 + (c & (a.npb - 1)) * a.patch_bytes + ((i * NW + wave) * RPI) * KB);
    };
    auto weight_piece = [&](int i, int s) {
        if ((i * NW + wave) >= B_TOTAL) return;
        const int c = s / a.taps, t = s - c * a.taps;
        const long long kt = (long long)t * a.chunks + c;   // chunk index in the (tap-major) weight image
        const long long off = p.w_layout == 1 ? kt * p.Cout * BKE : kt * BKE;
        glds16(b_ok[i] ? (const void*)(wg + b_base[i] + off) : (const void*)zero,
               wbuf0 + (s & 1) * (BN * KB) + ((i * NW + wave) * RPI) * KB);
    };

    // ---- prologue: patch of chunk 0 and weights of step 0
#pragma unroll
    for (int i = 0; i < PATCH_MAX_AI; ++i)
        if (i < a.n_ai) patch_piece(i, 0);
#pragma unroll
    for (int i = 0; i < BI; ++i) weight_piece(i, 0);

    int c = 0, t = 0, kh = 0, kw = 0;
    for (int s = 0; s < a.S; ++s) {
        glds_wait();
        __syncthreads();
        if (a.npb == 1 && t == 0 && c > 0) {
            // single patch buffer: the patch of this chunk can only be fetched now (all waves left chunk c-1)
#pragma unroll
            for (int i = 0; i < PATCH_MAX_AI; ++i)
                if (i < a.n_ai) patch_piece(i, c);
            glds_wait();
            __syncthreads();
        }
        const bool more = s + 1 < a.S;
        const bool prefetch_patch = a.npb == 2 && (c + 1 < a.chunks);
        const unsigned char* pa = patch0 + (c & (a.npb - 1)) * a.patch_bytes;
        const unsigned char* pb = wbuf0 + (s & 1) * (BN * KB);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const int slot = 2 * kk + fhalf;
            uint4 fa[MI], fb[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int row = (wm * MI + i + kh) * a.PW + kw + frow;   // patch pixel of this lane's output pixel
                fa[i] = *(const uint4*)(pa + row * KB + ((slot ^ swz(row)) << 4));
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int row = wn * WN + j * 32 + frow;
                fb[j] = *(const uint4*)(pb + row * KB + ((slot ^ swz(row)) << 4));
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int j = 0; j < NI; ++j) Mma2<T>::run(acc[i][j], fa[i], fb[j]);
                // DMA pieces of the next step / next chunk, spread behind the MFMA groups
                const int slot_id = kk * MI + i;
#pragma unroll
                for (int b = 0; b < BI; ++b)
                    if (b % NSLOT == slot_id && more) weight_piece(b, s + 1);
#pragma unroll
                for (int q = 0; q < PATCH_MAX_AI; ++q)
                    if ((BI + q) % NSLOT == slot_id && prefetch_patch && q < a.n_ai && a_step[q] == t) patch_piece(q, c + 1);
            }
        }
        if (++kw == p.KW) { kw = 0; ++kh; }
        if (++t == a.taps) { t = 0; kh = 0; kw = 0; ++c; }
    }

    // ---------------------------------------------------------------- epilogue through LDS
    constexpr int GROUPS_PER_ROW = BN / 8;
    static_assert(NTHR % GROUPS_PER_ROW == 0, "group index must be loop invariant");
    float* cs = (float*)smem;
    int pass_rows = (a.lds_bytes / 4 / BN / 32) * 32;
    if (pass_rows > BM) pass_rows = BM;
    const int npass = (BM + pass_rows - 1) / pass_rows;
    const int eY = p.y_f32 ? 4 : (int)sizeof(T);
    const bool vec_all = vec_ok(p.y, p.ldy, eY) && vec_ok(p.res, p.ldr, p.res_f32 ? 4 : (int)sizeof(T)) &&
                         vec_ok(p.y2, p.ldy2, (int)sizeof(T)) && vec_ok(p.aux0, p.lda0, (int)sizeof(T)) &&
                         vec_ok(p.aux1, p.lda1, (int)sizeof(T));
    const int my_cg = tid % GROUPS_PER_ROW;
    const int my_cout0 = n0 + my_cg * 8;
    const int my_valid = (p.Cout - my_cout0) >= 8 ? 8 : (p.Cout - my_cout0 > 0 ? p.Cout - my_cout0 : 0);
    GroupConst gc;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const bool ok = e < my_valid;
        gc.bias[e] = (ok && p.bias) ? p.bias[my_cout0 + e] : 0.f;
        gc.s1[e] = (ok && p.act1 == GVFI_ACT_PRELU) ? p.slope1[my_cout0 + e] : 0.f;
        gc.s2[e] = (ok && p.act2 == GVFI_ACT_PRELU) ? p.slope2[my_cout0 + e] : 0.f;
    }
    __syncthreads();   // all waves are done with the patch / weight buffers
#pragma unroll 1
    for (int ps = 0; ps < npass; ++ps) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int row0 = wm * WM + i * 32;
            if (row0 / pass_rows != ps) continue;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int col = wn * WN + j * 32 + frow;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 - ps * pass_rows + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
                    cs[row * BN + col] = acc[i][j][r];
                }
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int idx = tid; idx < pass_rows * GROUPS_PER_ROW; idx += NTHR) {
            const int row = ps * pass_rows + idx / GROUPS_PER_ROW;   // pixel index inside the tile
            const int y = y0 + (row >> 5), x = x0 + (row & 31);
            if (row >= BM || y >= p.H || x >= p.W || my_valid == 0) continue;
            float vv[8];
            const float* src = cs + (row - ps * pass_rows) * BN + my_cg * 8;
            const float4 c0 = *(const float4*)(src);
            const float4 c1 = *(const float4*)(src + 4);
            vv[0] = c0.x; vv[1] = c0.y; vv[2] = c0.z; vv[3] = c0.w; vv[4] = c1.x; vv[5] = c1.y; vv[6] = c1.z; vv[7] = c1.w;
            const long long pix = ((long long)img * p.H + y) * p.W + x;
            epilogue_group<T>(p, gc, vv, my_cout0, my_valid, pix, vec_all && my_valid == 8);
        }
        if (ps + 1 < npass) __syncthreads();
    }
}

template <typename T, int TH, int BN, int WAVES_M, int WAVES_N, int KB>
static int launch_patch(const gvfi_conv_params& p, hipStream_t stream) {
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int SL = KB / 16, RPI = 64 / SL;
    constexpr int BKE = KB / (int)sizeof(T);
    PatchArgs a;
    a.p = p;
    a.chunks0 = p.c0 / BKE;
    a.chunks = (p.c0 + p.c1) / BKE;
    a.taps = p.KH * p.KW;
    a.S = a.chunks * a.taps;
    a.PH = TH + p.KH - 1;
    a.PW = 32 + p.KW - 1;
    const int nrows = a.PH * a.PW;
    a.n_ai = (nrows + RPI * NW - 1) / (RPI * NW);
    if (a.n_ai > PATCH_MAX_AI) return -6;
    a.patch_bytes = a.n_ai * NW * RPI * KB;
    const int wbytes = 2 * BN * KB;
    const int lds_max = 160 * 1024;
    a.npb = (a.chunks > 1 && 2 * a.patch_bytes + wbytes <= lds_max) ? 2 : 1;
    a.lds_bytes = a.npb * a.patch_bytes + wbytes;
    const int epi_min = 32 * BN * 4;   // the epilogue stages at least 32 rows of fp32
    if (a.lds_bytes < epi_min) a.lds_bytes = epi_min;
    if (a.lds_bytes > lds_max) return -6;
    a.pp = (a.n_ai + a.taps - 1) / a.taps;
    a.tiles_x = (p.W + 31) / 32;
    a.tiles_y = (p.H + TH - 1) / TH;
    a.MT = p.N * a.tiles_x * a.tiles_y;
    a.NT = cdiv(p.Cout, BN);
    a.per_xcd = cdiv((long long)a.MT * a.NT, 8);
    a.Ktot = (long long)p.KH * p.KW * (p.c0 + p.c1);
    dim3 grid(a.per_xcd * 8, 1, 1);
    GVFI_LAUNCH_COOP_SHM((conv_patch_kernel<T, TH, BN, WAVES_M, WAVES_N, KB>), grid, dim3(64 * NW), a.lds_bytes, stream, a);
    return (int)hipGetLastError();
}

// 0 = not eligible, else the K-chunk row size in bytes (128 or 64) the patch kernel would use
extern "C" int gvfi_conv2d_patch_eligible(const gvfi_conv_params* pp) {
    const gvfi_conv_params& p = *pp;
    if (p.stride != 1 || p.pad_mode != GVFI_PAD_ZEROS || p.groups > 1) return 0;
    if (p.KH * p.KW <= 1 || p.pad_h != p.KH / 2 || p.pad_w != p.KW / 2 || p.Ho != p.H || p.Wo != p.W) return 0;
    if (p.KH > 7 || p.KW > 7 || p.c0 <= 0) return 0;
    const int e128 = p.dtype == GVFI_F32 ? 32 : 64;
    if (p.c0 % e128 == 0 && p.c1 % e128 == 0) return 128;
    if (p.w_layout != 0) return 0;   // the 64-byte chunking needs the plain [Cout][K] weight image
    const int e64 = e128 / 2;
    if (p.c0 % e64 == 0 && p.c1 % e64 == 0) return 64;
    return 0;
}

extern "C" int gvfi_conv2d_patch(const gvfi_conv_params* pp, void* stream) {
    const gvfi_conv_params& p = *pp;
    const int kb = gvfi_conv2d_patch_eligible(pp);
    if (!kb) return -2;
    if (((uintptr_t)p.x0 & 15) || ((uintptr_t)p.x1 & 15) || ((uintptr_t)p.w & 15)) return -3;
    hipStream_t st = (hipStream_t)stream;
    const long long M = (long long)p.N * p.H * p.W;
    int tile = p.tile_hint & 1023;
    if (tile == 0) tile = (p.Cout >= 192 && M >= 256 * 256 && kb == 128) ? 256 : (p.Cout > 64 ? 128 : (p.Cout > 32 ? 64 : 32));
#define PATCH_DISPATCH(TT)                                                                              \
    if (kb == 128) {                                                                                    \
        if (tile >= 256) return launch_patch<TT, 8, 256, 2, 4, 128>(p, st);                             \
        if (tile >= 128) return launch_patch<TT, 4, 128, 2, 2, 128>(p, st);                             \
        if (tile >= 64) return launch_patch<TT, 4, 64, 2, 2, 128>(p, st);                               \
        return launch_patch<TT, 4, 32, 4, 1, 128>(p, st);                                               \
    } else {                                                                                            \
        if (tile >= 128) return launch_patch<TT, 4, 128, 2, 2, 64>(p, st);                              \
        if (tile >= 64) return launch_patch<TT, 4, 64, 2, 2, 64>(p, st);                                \
        return launch_patch<TT, 4, 32, 4, 1, 64>(p, st);                                                \
    }
    if (p.dtype == GVFI_F32) { PATCH_DISPATCH(float) }
    PATCH_DISPATCH(bf16_t)
#undef PATCH_DISPATCH
}
