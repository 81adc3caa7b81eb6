// Implicit-GEMM convolution, LDS-DMA variant ("v2") for the FLOP-dominant layers.
//
// Same math, layouts and epilogue as conv_igemm.hip, restricted to channel counts that are multiples
// of one K chunk (BK = 128 bytes per tile row: 64 bf16 / 32 f32), i.e. every K chunk lies inside ONE
// filter tap and ONE of the two sources.  That makes the tap / source selection wave-uniform and lets
// both tiles be staged with `global_load_lds_dwordx4` (HBM/L2 -> LDS, no VGPR round trip, no ds_write):
//
//   * an LDS-DMA instruction writes lane-linear: base + lane*16 B.  A wave instruction therefore fills
//     8 tile rows x 8 slots of 16 B.  The bank-conflict swizzle (slot ^= (row>>1)&7) cannot be applied
//     to the destination, so it is applied to the per-lane SOURCE k-group and again on the ds_read_b128
//     fragment reads (same involution on both sides).
//   * the DMA uses the buffer-resource form (`buffer_load_dwordx4 ... offen lds`): a wave-uniform 128-bit descriptor
//     per K chunk + a 32-bit per-lane byte offset that is CONSTANT for the whole kernel; zero padding / ragged
//     edges are an out-of-range offset (the hardware returns zeros), selected by a per-row bit mask over the taps.
//     This keeps the per-chunk instruction overhead of the gather at ~4 instructions per 1 KiB piece -- the
//     kernel is instruction-issue bound, not bandwidth bound (profiles/).
//   * two LDS stages (2 x 32 KiB for 128x128x64 bf16): the DMA of chunk k+1 is issued right after the
//     single barrier of chunk k and overlaps its 16 MFMA 32x32x16 per wave; `s_waitcnt vmcnt(0)` +
//     barrier at the top of the next iteration is the only synchronisation (2 workgroups per CU).
//   * XCD-aware workgroup order: each XCD walks a contiguous range of M tiles, N tiles of one M tile
//     back to back, so re-reads of the activation halo hit that XCD's L2.
#include "conv_mma.h"
#include <type_traits>

// Measured and not kept (round 3, profiles/r3_occupancy_ab.txt): budgeting the 64 x 128 weights-direct tile for three
// workgroups per CU (amdgpu_waves_per_eu(3,3): 167 registers, bias / slopes fetched after the K loop, 36 spilled registers
// in the tail sequences) so that the other launch sequence's next kernel could start under this one's K loop: 328.5 against
// 332.8 frames/s (R) and 172.8 against 178.3 (F).  -DGVFI_WDIR_OCC3=1 rebuilds that variant.
#ifndef GVFI_WDIR_OCC3
#define GVFI_WDIR_OCC3 0
#endif

struct ConvArgs2 {
    gvfi_conv_params p;
    int chunks0;     // K chunks per tap that come from source 0  (c0 / BKE)
    int chunks_tap;  // K chunks per tap                             ((c0+c1) / BKE)
    int KT;          // total K chunks = KH*KW*chunks_tap
    int Mg;          // output pixels per weight group
    int MT, NT;      // tiles
    int per_xcd;     // ceil(MT*NT / 8)
    int dbg;         // profiling switches (algo >> 8): 8 = no epilogue, 16 = no K loop, 128 = s_memtime stamps
    long long Ktot;  // weight row length in elements
    unsigned howo_mul, howo_sh, wo_mul, wo_sh;   // magic numbers: x / d == (umulhi(x, mul) + x) >> sh  for x < 2^31
};

// ---------------------------------------------------------------- LDS-staged epilogue
// The accumulators (lane = cout, 16 pixels per lane) are first written to LDS as a row-major fp32
// [rows][BN] tile (the staging buffers are free after the K loop), then every thread takes groups of
// 8 consecutive output channels of one pixel: bias / activation / residual / GRU gate math on 8 values
// and ONE 16-byte store (bf16) or two (f32), fully coalesced along the channel axis.  This keeps the
// register footprint of the epilogue tiny (no spills with 128 accumulators) and replaces 2-byte stores.
//
// BDIR ("weights direct", round 3; w_layout = 2): the recurrence convolutions (M = 14-28 k pixels, 224-448 workgroups, all
// resident at once) were bound by the LATENCY of their own staging chain -- a 2-deep ring holds one 24 KB chunk in flight
// per workgroup and a chunk takes ~1 us to land, against 256 cycles of MFMAs per wave (tools/ring_bench.py: 900-1350
// cycles per K step whatever the tile).  Here only the A operand (pixels) goes through LDS; every wave owns a 32 / 64
// column slice of the output for ALL rows of the tile and loads its weight fragments straight from a fragment-ordered
// image ([K chunk][32-column block][k-step][lane][8]: one fully coalesced 1 KiB global load = one MFMA operand) into a
// REGISTER ring NSTAGE chunks deep -- the register file, 512 KB per CU, is the prefetch buffer of the weight stream; the
// LDS ring of the same depth costs 8 KB per stage.  The loads are ordinary loads (the compiler places their counted
// s_waitcnt); the counted waits of the A pieces include them (every wave issues A_INSTR + NI*KK operations per chunk, in
// that order, fenced by the asm statements around them).
// The kernel body: workgroup `bid` (of the problem's own grid) and weight group `g` of problem `a`.  Two entry points call it:
// conv_igemm_glds_kernel (one problem per launch) and conv_igemm_glds_pair_kernel (two independent problems of the same
// variant in ONE launch, see below).
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int KB, int NSTAGE, bool PIPE, int PPS = 1, bool BDIR = false>
__device__ __forceinline__ void conv_igemm_glds_body(const ConvArgs2& a, const int bid, const int g) {
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int VE = Elem<T>::VE;
    constexpr int RB = KB;                         // LDS row bytes = one K chunk of one tile row
    constexpr int SL = KB / 16;                    // 16-byte slots per row (8 | 4)
    constexpr int RPI = 64 / SL;                   // rows per LDS-DMA instruction (8 | 16)
    constexpr int BKE = KB / (int)sizeof(T);       // elements per K chunk
    constexpr int KK = KB / 32;                    // MFMA k-steps per chunk
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int A_TOTAL = BM / RPI, B_TOTAL = BDIR ? 1 : BN / RPI;
    constexpr int A_INSTR = (A_TOTAL + NW - 1) / NW;   // LDS-DMA instructions per wave for the A / B tile
    constexpr int B_INSTR = BDIR ? 0 : (B_TOTAL + NW - 1) / NW;
    constexpr int NPIECE = A_INSTR + B_INSTR;
    constexpr int STAGE = (BM + (BDIR ? 0 : BN)) * RB;
    static_assert(!BDIR || (WAVES_M == 1 && KB == 128 && sizeof(T) == 2 && A_TOTAL % NW == 0), "weights-direct variant");
    constexpr int AHEAD = NSTAGE - 1;              // chunks in flight beyond the one being consumed
    static_assert(MI >= 1 && NI >= 1 && NSTAGE >= 2 && NSTAGE <= 4, "tile");
    static_assert((NSTAGE - 2) * NPIECE <= 63, "vmcnt range");
    // counted waits assume every wave issues exactly NPIECE DMAs per chunk
    // (a wave without its own B rows re-stages another wave's rows -- same bytes to the same LDS address -- so that
    // every wave issues exactly NPIECE DMAs per chunk)
    static_assert(NSTAGE == 2 || A_TOTAL % NW == 0, "ring needs whole A pieces per wave");
    auto swz = [](int row) { return KB == 128 ? ((row >> 1) & 7) : ((row >> 2) & 3); };

    __shared__ __attribute__((aligned(16))) unsigned char smem[NSTAGE * STAGE];   // ring of K chunks

    const gvfi_conv_params& p = a.p;
#ifndef GVFI_HOSTSIM
    // Every kernel argument the prologue and the K loop read, fetched in ONE batch of scalar loads.  Left alone hipcc
    // loads each field right before its first use: ~10 dependent s_load -> s_waitcnt round trips (the scalar cache is
    // cold at kernel start) spread over the prologue -- 3 us per workgroup in s_memtime stamps, more than the K loop of
    // the small recurrence layers (tools/ring_bench.py --stamps).  The empty asm statements only make the values live here.
    asm volatile("" ::"s"(p.x0), "s"(p.x1), "s"(p.w), "s"(p.w_group_stride), "s"(p.ld0), "s"(p.ld1), "s"(p.c0), "s"(p.N), "s"(p.H),
                 "s"(p.W), "s"(p.Cout), "s"(p.KH), "s"(p.KW), "s"(p.stride), "s"(p.pad_h), "s"(p.pad_w), "s"(p.Ho), "s"(p.Wo),
                 "s"(p.groups), "s"(p.w_layout));
    asm volatile("" ::"s"(a.chunks0), "s"(a.chunks_tap), "s"(a.KT), "s"(a.Mg), "s"(a.MT), "s"(a.NT), "s"(a.per_xcd), "s"(a.dbg),
                 "s"(a.Ktot), "s"(a.howo_mul), "s"(a.howo_sh), "s"(a.wo_mul), "s"(a.wo_sh), "s"(p.bias), "s"(p.act1), "s"(p.act2),
                 "s"(p.slope1), "s"(p.slope2), "s"(p.epi_mode));
#endif
    // ---- XCD-aware tile order (the workgroup index round-robins over the 8 XCDs)
    const int v = (bid & 7) * a.per_xcd + (bid >> 3);
    if (v >= a.MT * a.NT) return;
    const int mt = v / a.NT, nt = v - mt * a.NT;

    const int tid = threadIdx.x;
    // profiling only (algo bit 8+7): wave 0 of every workgroup stamps s_memtime at the phase boundaries into
    // aux1[bid*16 + k] (k: 0 start, 7 rows decoded, 8 offsets + accumulators ready, 9 ring prologue issued, 1 prologue done,
    // 2 first chunk landed, 3 K loop done, 4 staged, 5 stores issued, 6 stores acknowledged)
    auto stamp = [&](int k) {
#ifndef GVFI_HOSTSIM
        if ((a.dbg & 128) && tid == 0) ((unsigned long long*)p.aux1)[(long long)bid * 16 + k] = __builtin_readcyclecounter();
#endif
    };
    stamp(0);
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform -> SGPR (M0 bases)
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const long long m_tile0 = (long long)mt * BM;
    const int n0 = nt * BN;
    const int HoWo = p.Ho * p.Wo;
    const T* __restrict__ x0 = (const T*)p.x0;
    const T* __restrict__ x1 = (const T*)p.x1;
    const T* __restrict__ wg = (const T*)p.w + (long long)g * p.w_group_stride;

    const int lrow = lane / SL;   // row inside a DMA group
    const int lslot = lane % SL;  // destination slot

    // per-thread A rows (one per DMA instruction of this wave).  The source address of a K chunk is
    //   [uniform SRD base = xs + ((pix_ref + tap offset) * ld + chunk offset)]  +  [per-lane 32-bit byte offset],
    // the per-lane part is constant for the whole kernel (one value per source because their pitches differ);
    // padding / ragged rows are a per-row bit mask over the filter taps -> offset GVFI_DMA_OOB -> zeros.
    const int esz = (int)sizeof(T);
    // m -> (image, oy, ox) with host-provided magic-number divisions (a 64-bit hardware division per row made the
    // prologue cost ~9 us per workgroup, a fifth of a small RAFT convolution)
    const int imgs_per_group = p.N / (p.groups > 0 ? p.groups : 1);
    auto decode = [&](unsigned m, int& n, int& oy, int& ox) {
        const unsigned q = (__umulhi(m, a.howo_mul) + m) >> a.howo_sh;     // m / (Ho*Wo)
        const unsigned rem = m - q * (unsigned)HoWo;
        const unsigned y = (__umulhi(rem, a.wo_mul) + rem) >> a.wo_sh;     // rem / Wo
        n = g * imgs_per_group + (int)q;
        oy = (int)y;
        ox = (int)(rem - y * (unsigned)p.Wo);
    };
    int pix_ref;   // input pixel index of tile row 0, tap (0,0) (wave-uniform, may lie in the padding)
    {
        int n, oy, ox;
        decode((unsigned)(m_tile0 < a.Mg ? m_tile0 : 0), n, oy, ox);
        pix_ref = (n * p.H + oy * p.stride - p.pad_h) * p.W + ox * p.stride - p.pad_w;
    }
    unsigned a_off0[A_INSTR], a_off1[A_INSTR], a_mask[A_INSTR];
    unsigned rowsum = 0u;      // wave-uniform: bit t*KW set for every filter row t
    for (int th = 0; th < p.KH; ++th) rowsum |= 1u << (th * p.KW);
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
        const int row = (i * NW + wave) * RPI + lrow;
        const long long m = m_tile0 + row;
        const bool ok = (A_TOTAL % NW == 0 || (i * NW + wave) < A_TOTAL) && m < a.Mg;
        int n, oy, ox;
        decode((unsigned)(ok ? m : 0), n, oy, ox);
        const int iy0 = oy * p.stride - p.pad_h, ix0 = ox * p.stride - p.pad_w;
        const int dpix = (n * p.H + iy0) * p.W + ix0 - pix_ref;
        const int koff = (lslot ^ swz(row)) * VE;
        a_off0[i] = (unsigned)(dpix * p.ld0 + koff) * esz;
        a_off1[i] = (unsigned)(dpix * p.ld1 + koff) * esz;
        // bit t of the mask = filter tap t reads inside the image for this output pixel.  Closed form (the prologue is
        // instruction-count bound: ~6 cycles per instruction for a wave alone on its SIMD, tools/microbench/icache.hip):
        // the valid taps of a row are a contiguous range [lo, hi) in x and in y, so
        //   xbits = 2^hi_x - 2^lo_x,   mask = xbits * (rowsum & (2^(hi_y KW) - 2^(lo_y KW))),   rowsum = sum_t 2^(t KW)
        // (xbits < 2^KW: the product has no carries between rows; KH*KW <= 32 is checked on the host)
        const int lox = ix0 < 0 ? -ix0 : 0, hix = p.W - ix0 < p.KW ? p.W - ix0 : p.KW;
        const int loy = iy0 < 0 ? -iy0 : 0, hiy = p.H - iy0 < p.KH ? p.H - iy0 : p.KH;
        unsigned mask = 0u;
        if (lox < hix && loy < hiy) {
            const unsigned xbits = (unsigned)((1ull << hix) - (1ull << lox));
            const unsigned yr = (unsigned)((1ull << (hiy * p.KW)) - (1ull << (loy * p.KW)));
            mask = xbits * (rowsum & yr);
        }
        a_mask[i] = ok ? mask : 0u;
    }
    stamp(7);   // (profiling) per-row decode + tap masks done
    unsigned b_off[B_INSTR > 0 ? B_INSTR : 1];
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) {
        const int row = ((i * NW + wave) % B_TOTAL) * RPI + lrow;
        const int n = n0 + row;
        const bool ok = n < p.Cout;
        if (p.w_layout == 1)   // chunk-major, pre-swizzled image: [K chunk][Cout][128 B] == the LDS image (KB = 128)
            b_off[i] = ok ? (unsigned)((n * SL + lslot) * VE * esz) : GVFI_DMA_OOB;
        else
            b_off[i] = ok ? (unsigned)(((long long)n * a.Ktot + (lslot ^ swz(row)) * VE) * esz) : GVFI_DMA_OOB;
    }
    // fragment read offsets inside a stage: one per MFMA k-step; the 32-row blocks of a wave are immediate offsets
    unsigned a_rd[KK], b_rd[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        const int ra = wm * WM + (lane & 31), rb = wn * WN + (lane & 31);
        const int slot = 2 * kk + (lane >> 5);
        a_rd[kk] = ra * RB + ((slot ^ swz(ra)) << 4);
        b_rd[kk] = BM * RB + rb * RB + ((slot ^ swz(rb)) << 4);
    }

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31;
    const int fhalf = lane >> 5;
    stamp(8);

    int kh = 0, kw = 0, ck = 0, tap = 0;   // wave-uniform K walker of the NEXT chunk to stage
    // The three buffer descriptors are constant for the kernel (source 0 / source 1 at the tile's reference pixel, the
    // weights of this group); a K chunk is selected by a 32-bit byte offset in the instruction's SGPR-offset operand.
    // (Re-building two 128-bit descriptors per chunk with 64-bit address arithmetic had been ~50 scalar instructions at
    // the head of every chunk, in front of the first fragment read.)
    const gvfi_i32x4 srd_a0 = make_srd(x0 + (long long)pix_ref * p.ld0);
    const gvfi_i32x4 srd_a1 = make_srd(x1 != nullptr ? x1 + (long long)pix_ref * p.ld1 : x0);
    const gvfi_i32x4 srd_b = make_srd(wg);
    // per-chunk uniform staging state (set by stage_begin, consumed by stage_piece)
    bool st_from0 = true;
    unsigned st_tapbit = 1u, st_soff_a = 0u, st_soff_b = 0u;
    const unsigned smem_lds = lds_address(smem);
    unsigned st_sa = smem_lds;
    // K order: channel chunk OUTER, filter tap INNER.  While a chunk's taps are walked, the workgroups of an XCD only
    // touch that chunk's 128-byte slice of their input pixels (1/chunks_tap of the footprint), so the 9 shifted
    // re-reads of a 3x3 filter hit the XCD's L2; with the tap outside, every re-read of the 256->256 layer had gone
    // back to the fabric (FETCH_SIZE 5x the input, profiles/).  The chunk-major weight image is packed in this order.
    auto stage_begin = [&](int kt) {
        st_sa = smem_lds + (kt % NSTAGE) * STAGE;
        st_from0 = ck < a.chunks0;
        const int ld = st_from0 ? p.ld0 : p.ld1;
        const int cbase = (st_from0 ? ck : ck - a.chunks0) * BKE;
        st_soff_a = (unsigned)(((kh * p.W + kw) * ld + cbase) * esz);
        st_soff_b = (unsigned)((p.w_layout == 1 ? kt * p.Cout : tap * a.chunks_tap + ck) * (BKE * esz));
        st_tapbit = 1u << tap;
        ++tap;
        if (++kw == p.KW) {
            kw = 0;
            if (++kh == p.KH) { kh = 0; tap = 0; ++ck; }
        }
    };
    // one LDS-DMA instruction (1 KiB per wave): pieces [0, A_INSTR) are A rows, [A_INSTR, NPIECE) B rows
    auto stage_piece = [&](int pc) {
        if (pc < A_INSTR) {
            const int i = pc;
            if (A_TOTAL % NW != 0 && (i * NW + wave) >= A_TOTAL) return;
            const unsigned off = st_from0 ? a_off0[i] : a_off1[i];
            bufdma16((a_mask[i] & st_tapbit) ? off : GVFI_DMA_OOB, st_from0 ? srd_a0 : srd_a1, st_soff_a,
                     st_sa + ((i * NW + wave) * RPI) * RB);
        } else if (pc < NPIECE) {
            const int i = pc - A_INSTR;
            bufdma16(b_off[i], srd_b, st_soff_b, st_sa + BM * RB + (((i * NW + wave) % B_TOTAL) * RPI) * RB);
        }
    };

    // this thread's epilogue channel group is fixed (NT % GROUPS_PER_ROW == 0): its bias / PReLU slopes are loaded once.
    // 4-wave tiles have registers to spare, so they fetch them HERE, before the K loop (a dependent ~1.5 us global
    // load chain otherwise sits between the last MFMA and the first store of every small RAFT convolution)
    constexpr int NT = 64 * NW;
    constexpr int GROUPS_PER_ROW = BN / 8;
    static_assert(NT % GROUPS_PER_ROW == 0, "group index must be loop invariant");
    // (GVFI_WDIR_OCC3 variant of the weights-direct tile: fetched after the K loop -- 24 registers less across it; the
    // loads are issued ahead of the accumulator staging and awaited behind its barrier)
    constexpr bool GC_LATE_WDIR = BDIR && (GVFI_WDIR_OCC3 || BM == 128);
    constexpr bool GC_EARLY = NW <= 4 && !GC_LATE_WDIR;
    const int my_cg = tid % GROUPS_PER_ROW;
    const int my_cout0 = n0 + my_cg * 8;
    const int my_valid = (p.Cout - my_cout0) >= 8 ? 8 : (p.Cout - my_cout0 > 0 ? p.Cout - my_cout0 : 0);
    GroupConst gc;
    auto load_gc = [&]() {
        // negative-side slope of the none / ReLU / LeakyReLU / PReLU family: act(v) = max(v,0) + s * min(v,0)
        // (exact for every member: s = 1, 0, 0.1, slope[c]); the generic path reads it for PReLU only
        const float f1 = p.act1 == GVFI_ACT_NONE ? 1.f : (p.act1 == GVFI_ACT_LRELU ? 0.1f : 0.f);
        const float f2 = p.act2 == GVFI_ACT_NONE ? 1.f : (p.act2 == GVFI_ACT_LRELU ? 0.1f : 0.f);
        auto ld8f = [&](const float* src, float fill, float (&dst)[8]) {
            if (src != nullptr && my_valid == 8) {   // two 16-byte loads (cout0 is a multiple of 8 floats)
                const float4 lo = *(const float4*)(src + my_cout0), hi = *(const float4*)(src + my_cout0 + 4);
                dst[0] = lo.x; dst[1] = lo.y; dst[2] = lo.z; dst[3] = lo.w;
                dst[4] = hi.x; dst[5] = hi.y; dst[6] = hi.z; dst[7] = hi.w;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) dst[e] = (src != nullptr && e < my_valid) ? src[my_cout0 + e] : fill;
            }
        };
        ld8f(p.bias, 0.f, gc.bias);
        ld8f(p.act1 == GVFI_ACT_PRELU ? p.slope1 : nullptr, f1, gc.s1);
        ld8f(p.act2 == GVFI_ACT_PRELU ? p.slope2 : nullptr, f2, gc.s2);
    };
    // (BDIR) register ring of weight fragments: slot = chunk % NSTAGE, [column block][k-step]
    constexpr int NBL = NI * KK;                       // fragment loads per chunk and wave
    uint4 bq[BDIR ? NSTAGE : 1][NI][KK];
    // (the image holds ceil(Cout / 32) column blocks; blocks of a ragged last tile beyond it re-read the last one --
    // their columns are never stored)
    const int nb_tot = (p.Cout + 31) >> 5;
    const size_t wf_chunk = (size_t)nb_tot * KK * 1024;     // bytes per K chunk of the fragment image
    const unsigned char* wfrag[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        int nb = n0 / 32 + wn * NI + j;
        nb = nb < nb_tot ? nb : nb_tot - 1;
        wfrag[j] = (const unsigned char*)wg + ((size_t)nb * KK) * 1024 + lane * 16;
    }
    auto load_b = [&](int kt_, int slot) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const unsigned char* src = wfrag[j] + (size_t)kt_ * wf_chunk;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                bq[slot][j][kk] = *(const uint4*)(src + kk * 1024);
                GVFI_EMU_VMEM_OP();
            }
        }
    };
    // ---- prologue: fill the ring with chunks 0 .. AHEAD-1
    if constexpr (BDIR) {
        // unconditional issue (hipcc's counted waits need a fixed number of loads in flight on every path into the K
        // loop): with fewer than AHEAD chunks the missing ones are phantoms -- all-zero A pieces (every row out of range)
        // and a re-load of the last chunk's fragments into a slot nobody reads
#pragma unroll
        for (int q = 0; q < AHEAD; ++q) {
            if (q < a.KT) stage_begin(q);
            else {
                st_sa = smem_lds + q * STAGE;
                st_tapbit = 0u;
            }
#pragma unroll
            for (int pc = 0; pc < NPIECE; ++pc) stage_piece(pc);
            load_b(q < a.KT ? q : a.KT - 1, q);
            GVFI_SCHED_BARRIER();
        }
    } else {
#pragma unroll
    for (int q = 0; q < AHEAD; ++q) {
        if (q < a.KT) {
            stage_begin(q);
#pragma unroll
            for (int pc = 0; pc < NPIECE; ++pc) stage_piece(pc);
        }
    }
    }
    stamp(9);
    if (GC_EARLY) load_gc();   // behind the first chunk's DMA, in its shadow
    // MFMAs of chunk kt; when DMA is true the pieces of chunk kt+AHEAD are issued behind the MFMA groups (every wave
    // of the workgroup is at the same point after the barrier; a burst of issues would idle the matrix pipe)
    auto compute = [&](int kt, auto dma_tag) {
        constexpr bool DMA = decltype(dma_tag)::value;
        const unsigned char* sa = smem + (kt % NSTAGE) * STAGE;
        // fragments are double-buffered by hand: the ds_reads of k-step kk+1 are issued BEFORE the MFMAs of k-step kk
        // and the scheduler may not move them back (left alone, hipcc re-uses one fragment set and every MFMA group
        // waits a full LDS round trip, ~45% of the K loop).  rows i*32 further down share the swizzle term
        // (32 rows = a multiple of its period): immediate offsets.
        uint4 fa[2][MI], fb[2][NI];
        auto load_frags = [&](int kk, int buf) {
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[buf][i] = *(const uint4*)(sa + a_rd[kk] + i * 32 * RB);
#pragma unroll
            for (int j = 0; j < NI; ++j) fb[buf][j] = *(const uint4*)(sa + b_rd[kk] + j * 32 * RB);
        };
        load_frags(0, 0);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            if (kk + 1 < KK) load_frags(kk + 1, (kk + 1) & 1);
            GVFI_SCHED_BARRIER();
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int j = 0; j < NI; ++j) Mma2<T>::run(acc[i][j], fa[kk & 1][i], fb[kk & 1][j]);
                if (DMA) {
                    constexpr int NSLOT = KK * MI;
                    const int slot_id = kk * MI + i;
#pragma unroll
                    for (int pc = 0; pc < NPIECE; ++pc)
                        if (pc % NSLOT == slot_id) stage_piece(pc);
                }
            }
            GVFI_SCHED_BARRIER();
        }
    };
    // Steady state: chunk kt must have landed while the AHEAD-1 younger chunks stay in flight (LDS-DMA issue->landed
    // is ~1 us, longer than the MFMAs of one chunk).  vmcnt retires in order and every wave issues exactly NPIECE DMAs
    // per chunk, so (AHEAD-1)*NPIECE is the count to wait for.  The last AHEAD chunks are drained without prefetch.
    stamp(1);
    int kt = (a.dbg & 16) ? a.KT : 0;   // profiling only: skip the K loop
    if constexpr (BDIR) {
        stamp(2);      // (the first chunk's landing is part of the first K step in this variant's timeline)
        constexpr int OPS = A_INSTR + NBL;             // vector-memory operations per chunk and wave
        static_assert((AHEAD - 1) * OPS <= 63, "vmcnt range");
        // Unrolled by the ring depth: the register slot of a chunk is a compile-time index (U = c % NSTAGE), and so is
        // whether a chunk issues the chunk AHEAD of it (ISSUE) and how many younger chunks are in flight behind it
        // (YOUNGER): with the issue inside a run-time `if`, hipcc's own counted waits for the weight fragments fall back to
        // the path on which nothing was issued and drain the ring (vmcnt(7) instead of 12+: one chunk of look-ahead left).
        auto chunk = [&](int c, auto u_tag, auto issue_tag, auto younger_tag) {
            constexpr int U = decltype(u_tag)::value;
            constexpr bool ISSUE = decltype(issue_tag)::value;
            constexpr int YOUNGER = decltype(younger_tag)::value;
            glds_wait_n<YOUNGER * OPS>();     // chunk c (A pieces of this wave + its weight fragments) has landed
            __syncthreads();      // every wave's A pieces of chunk c visible; every wave is done with chunk c-1's slot
            // the first fragment reads go out BEFORE the scalar work of the next issue: their LDS round trip (~150
            // cycles, once per chunk, otherwise exposed in front of the first MFMA) runs under it
            const unsigned char* sa = smem + U * STAGE;
            uint4 fa[2][MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[0][i] = *(const uint4*)(sa + a_rd[0] + i * 32 * RB);
            GVFI_SCHED_BARRIER();
            if constexpr (ISSUE) {
                stage_begin(c + AHEAD);
                st_sa = smem_lds + ((U + AHEAD) % NSTAGE) * STAGE;     // (compile-time slot: no run-time modulo)
#pragma unroll
                for (int pc = 0; pc < NPIECE; ++pc) stage_piece(pc);
                load_b(c + AHEAD, (U + AHEAD) % NSTAGE);
            }
            GVFI_SCHED_BARRIER();
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                if (kk + 1 < KK) {
#pragma unroll
                    for (int i = 0; i < MI; ++i) fa[(kk + 1) & 1][i] = *(const uint4*)(sa + a_rd[kk + 1] + i * 32 * RB);
                }
                GVFI_SCHED_BARRIER();
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) Mma2<T>::run(acc[i][j], fa[kk & 1][i], bq[U][j][kk]);
                GVFI_SCHED_BARRIER();
            }
        };
        static_assert(NSTAGE == 4 && AHEAD == 3, "the chunk sequences below are written out for a ring of 4");
        using std::integral_constant;
        typedef std::true_type Y;
        typedef std::false_type N;
#define GVFI_CH(off, issue, younger) chunk(kt + (off), integral_constant<int, (off) % 4>{}, issue{}, integral_constant<int, younger>{})
        // steady state: every chunk of the group issues
        for (; kt + NSTAGE + AHEAD <= a.KT; kt += NSTAGE) {
            GVFI_CH(0, Y, 2); GVFI_CH(1, Y, 2); GVFI_CH(2, Y, 2); GVFI_CH(3, Y, 2);
        }
        // tail: R = KT - kt chunks left (kt % 4 == 0); the first R - 3 of them still issue, the last three drain
        switch (a.KT - kt) {
            case 1: GVFI_CH(0, N, 2); break;                       // (KT < 3: phantom chunks are the younger ones)
            case 2: GVFI_CH(0, N, 2); GVFI_CH(1, N, 1); break;
            case 3: GVFI_CH(0, N, 2); GVFI_CH(1, N, 1); GVFI_CH(2, N, 0); break;
            case 4: GVFI_CH(0, Y, 2); GVFI_CH(1, N, 2); GVFI_CH(2, N, 1); GVFI_CH(3, N, 0); break;
            case 5: GVFI_CH(0, Y, 2); GVFI_CH(1, Y, 2); GVFI_CH(2, N, 2); GVFI_CH(3, N, 1); GVFI_CH(4, N, 0); break;
            case 6: GVFI_CH(0, Y, 2); GVFI_CH(1, Y, 2); GVFI_CH(2, Y, 2); GVFI_CH(3, N, 2); GVFI_CH(4, N, 1); GVFI_CH(5, N, 0); break;
            default: break;
        }
#undef GVFI_CH
        glds_wait_n<0>();      // (phantom chunks of a K loop shorter than the ring: nothing may land after this point)
        kt = a.KT;
    } else if constexpr (NSTAGE == 2 && KK >= 2 && PIPE) {
        // ---- (8-wave tile only: with 2-3 resident 4-wave workgroups another workgroup fills the bubble and the shorter
        // DMA slack of this schedule costs 5-10 %, re-measured in round 2 with the peeled loop: 4-wave tiles stay on the
        // plain loop below)  two-buffer ring, software-pipelined across chunks: the barrier that publishes chunk kt+1 sits INSIDE the
        // MFMA stream of chunk kt -- after it every wave issues the first fragment reads of chunk kt+1 and then still
        // has the last k-step of chunk kt to feed the matrix pipe, so the LDS round trip that used to follow every
        // barrier (all waves idle, ~10 % of a K step) is covered.
        //   hazards: the DMA of chunk kt+1 (into the buffer of chunk kt-1) is issued after the barrier of the previous
        //   iteration, which every wave passes only with all its reads of that buffer complete (__syncthreads waits
        //   lgkmcnt(0)); chunk kt+1 is read only after vmcnt(0) + that same barrier.
        // The last chunk is a separate copy of the body (HAS_NEXT = false): inside the steady-state loop there is no
        // control-flow merge between "barrier + prefetch" and "no next chunk", so hipcc's waitcnt pass does not make the
        // last k-step wait for the fragment reads of the NEXT chunk that were issued a few instructions earlier (it did:
        // s_waitcnt lgkmcnt(1) in front of the first MFMA after the barrier -- the round trip was back), and the DMA
        // pieces carry no branches.
        uint4 fa[2][MI], fb[2][NI];
        auto load_frags = [&](const unsigned char* sa, int kk, int buf) {
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[buf][i] = *(const uint4*)(sa + a_rd[kk] + i * 32 * RB);
#pragma unroll
            for (int j = 0; j < NI; ++j) fb[buf][j] = *(const uint4*)(sa + b_rd[kk] + j * 32 * RB);
        };
        auto mma_step = [&](int kk) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) Mma2<T>::run(acc[i][j], fa[kk & 1][i], fb[kk & 1][j]);
        };
        auto chunk = [&](int kt_, auto next_tag) {
            constexpr bool HAS_NEXT = decltype(next_tag)::value;
            const unsigned char* sa = smem + (kt_ & 1) * STAGE;
            if (HAS_NEXT) stage_begin(kt_ + 1);
#pragma unroll
            for (int kk = 0; kk + 1 < KK; ++kk) {
                load_frags(sa, kk + 1, (kk + 1) & 1);
                GVFI_SCHED_BARRIER();
#pragma unroll
                for (int i = 0; i < MI; ++i) {
#pragma unroll
                    for (int j = 0; j < NI; ++j) Mma2<T>::run(acc[i][j], fa[kk & 1][i], fb[kk & 1][j]);
                    if (HAS_NEXT) {   // all DMA pieces of chunk kt+1 go out before the barrier below, PPS per MFMA group
                        // (PPS > 1 front-loads them: the issue -> landed time of a piece, ~1 us under load, is about one
                        // chunk of MFMAs, so every cycle a piece leaves earlier comes off the wait at the barrier:
                        // 4 per group measured -5 % on the hot layer against 1 per group, same box.  A 4-deep ring of
                        // 64-byte chunks with chunk-major weights -- DMA issued two chunks ahead -- measured between the
                        // two and was removed: with the wait gone the loop still takes ~62 us per tile at any shader
                        // clock from 1.5 to 1.9 GHz, i.e. it is bound on the L2 -> LDS delivery side)
                        constexpr int NSLOT = (KK - 1) * MI;
                        const int slot_id = kk * MI + i;
#pragma unroll
                        for (int pc = 0; pc < NPIECE; ++pc)
                            if ((PPS > 1 ? pc / PPS : pc % NSLOT) == slot_id) stage_piece(pc);
                    }
                }
                GVFI_SCHED_BARRIER();
            }
            if (HAS_NEXT) {
                glds_wait_n<0>();
                __syncthreads();      // chunk kt+1 visible; every wave's reads of chunk kt are complete
                load_frags(smem + ((kt_ + 1) & 1) * STAGE, 0, KK & 1);
                GVFI_SCHED_BARRIER();
            }
            mma_step(KK - 1);
            GVFI_SCHED_BARRIER();
        };
        if (kt < a.KT) {
            glds_wait_n<0>();
            __syncthreads();          // chunk 0 visible
            stamp(2);
            load_frags(smem, 0, 0);
            for (; kt + 1 < a.KT; ++kt) chunk(kt, std::true_type{});
            chunk(kt, std::false_type{});
            ++kt;
        }
    } else {
    for (; kt + AHEAD < a.KT; ++kt) {
        glds_wait_n<(AHEAD - 1) * NPIECE>();
        __syncthreads();    // chunk kt visible to every wave; every wave is done reading chunk kt-1's buffer
        if (kt == 0) stamp(2);
        stage_begin(kt + AHEAD);   // goes into the buffer chunk kt-1 occupied
        compute(kt, std::true_type{});
    }
    for (; kt < a.KT; ++kt) {
        glds_wait_n<0>();
        __syncthreads();
        compute(kt, std::false_type{});
    }
    }

    stamp(3);
    if (a.dbg & 8) return;   // profiling only: skip the epilogue
    // ---------------------------------------------------------------- epilogue through LDS (see above)
    // the fp32 C tile is staged in NPASS passes of PASS_ROWS rows; in every pass EACH wave stages 1/NPASS of its
    // accumulator blocks, so no wave carries all its accumulators through a whole store loop (register pressure of
    // the 8-wave tile); staged row lr of pass ps <-> tile row tile_row(ps, lr)
    constexpr int PASS_ROWS_RAW = (NSTAGE * STAGE / 4) / BN;
    constexpr int NPASS = PASS_ROWS_RAW >= BM ? 1 : 2;
    constexpr int PASS_ROWS = BM / NPASS;
    constexpr int IPP = MI / NPASS;                // 32-row accumulator blocks per wave and pass
    static_assert(MI % NPASS == 0 && PASS_ROWS <= PASS_ROWS_RAW, "epilogue staging does not fit");
    auto tile_row = [&](int ps, int lr) { return (lr / (IPP * 32)) * WM + ps * IPP * 32 + (lr % (IPP * 32)); };
    float* cs = (float*)smem;
    const int eY = p.y_f32 ? 4 : (int)sizeof(T);
    // vector path needs 16-byte aligned rows on every tensor the epilogue touches
    const bool vec_all = vec_ok(p.y, p.ldy, eY) && vec_ok(p.res, p.ldr, p.res_f32 ? 4 : (int)sizeof(T)) &&
                         vec_ok(p.y2, p.ldy2, (int)sizeof(T)) && vec_ok(p.aux0, p.lda0, (int)sizeof(T)) &&
                         vec_ok(p.aux1, p.lda1, (int)sizeof(T)) &&
                         (p.bias == nullptr || true);
    // (8-wave tile) uniform: plain activation epilogue that can be applied in the accumulator layout, see below
    const bool direct16 = NW >= 8 && sizeof(T) == 2 && BM * BN * 2 <= NSTAGE * STAGE && p.epi_mode == GVFI_EPI_STD &&
                          vec_all && !p.y_f32 && p.res == nullptr && (p.act1 <= GVFI_ACT_PRELU || p.act1 == GVFI_ACT_GELU) &&
                          p.act2 == GVFI_ACT_NONE && n0 + BN <= p.Cout;
    // GELU (the transformer MLPs of GIMM-VFI-F: 229 k-row linears) takes the slim store loops too: on the generic loop those
    // launches spent more in the epilogue than in the contraction
    const bool gelu1 = p.act1 == GVFI_ACT_GELU;
    if (!GC_EARLY && !direct16) load_gc();
    auto gc_wait = [&]() {
#ifndef GVFI_HOSTSIM
#pragma unroll
        for (int e = 0; e < 8; ++e) asm volatile("" ::"v"(gc.bias[e]), "v"(gc.s1[e]), "v"(gc.s2[e]));
#endif
    };
#ifndef GVFI_HOSTSIM
    if (!GC_LATE_WDIR && (GC_EARLY || !direct16))
    // Make the compiler wait for the bias / slope loads HERE.  Their first real use is inside the store loop; the
    // s_waitcnt vmcnt(0) it would put there also waits, in every iteration, for the previous iteration's global
    // store to be acknowledged (stores share the counter): ~1500 cycles x 8-16 iterations per tile, 15 % of the
    // hot convolution and a quarter of a small RAFT one (tools/conv_timeline.py).
#pragma unroll
    for (int e = 0; e < 8; ++e) asm volatile("" ::"v"(gc.bias[e]), "v"(gc.s1[e]), "v"(gc.s2[e]));
#endif
    // (a ragged last channel group with an even number of channels -- RAFT's 126-channel motion features -- takes the slim
    // loop too, with dword stores: on the generic loop its lanes kept every wave of the tile in a 3x longer epilogue,
    // 14.7 k instead of 4.8 k cycles per workgroup.  Its 16-byte residual read stays inside the row pitch.)
    const bool ragged_ok = my_valid >= 2 && !(my_valid & 1) && (p.res == nullptr || my_cout0 + 8 <= p.ldr) && p.stats == nullptr;
    // (float outputs without a residual -- the decoder head, the flow head's per-tap sums -- take it as well: float4 stores)
    // and a FLOAT residual (the float residual streams of GIMM-VFI-F's transformer blocks: every attention / MLP output
    // linear) on the 4-wave tiles, which have the registers for two vectors per row.
    constexpr bool F32RES = NT <= 256;
    const bool fast = sizeof(T) == 2 && p.epi_mode == GVFI_EPI_STD && vec_all && (my_valid == 8 || ragged_ok) &&
                      (p.res == nullptr || !p.res_f32 || F32RES) &&
                      (p.act1 <= GVFI_ACT_PRELU || p.act1 == GVFI_ACT_GELU) && p.act2 <= GVFI_ACT_PRELU;
    // (the 8-wave tile is never used for the GRU convolutions and has no registers for their operands)
    const bool fast_gru = sizeof(T) == 2 && NT <= 256 && p.epi_mode != GVFI_EPI_STD && vec_all && my_valid == 8 &&
                          (p.res == nullptr || p.res_f32);
    // fused InstanceNorm statistics (p.stats): this thread's 8 channels over its rows.  4-wave tiles only (the layers
    // that are normalised have <= 128 channels; the 8-wave tile has no registers to spare)
    constexpr bool STATS = NW <= 4;
    float st_sum[STATS ? 8 : 1], st_sq[STATS ? 8 : 1];
#pragma unroll
    for (int e = 0; e < (STATS ? 8 : 1); ++e) st_sum[e] = st_sq[e] = 0.f;
    __syncthreads();   // every wave is done reading the last staged chunk
    bool done16 = false;
    if constexpr (NW >= 8 && sizeof(T) == 2 && BM * BN * 2 <= NSTAGE * STAGE) {
        // ---- 8-wave tile, plain activation epilogue (4 of the 5 convolutions of a ResBlock): bias + activation are
        // applied in the accumulator layout (a lane owns ONE output channel per 32-column block, so they are per-lane
        // scalars), the bf16 result is staged -- the whole 256x256 tile fits the ring once, instead of two fp32
        // passes -- and the store loop only moves 16-byte vectors.  Same arithmetic and rounding as the loop below.
        if (direct16) {
            bf16_t* cs16 = (bf16_t*)smem;
            const float f1 = p.act1 == GVFI_ACT_NONE ? 1.f : (p.act1 == GVFI_ACT_LRELU ? 0.1f : 0.f);
            const bool has_sc = p.out_scale != 1.0f;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int col = wn * WN + j * 32 + frow;
                const float bj = p.bias ? p.bias[n0 + col] : 0.f;
                const float sj = p.act1 == GVFI_ACT_PRELU ? p.slope1[n0 + col] : f1;
#pragma unroll
                for (int i = 0; i < MI; ++i) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
                        const float t = acc[i][j][r] + bj;
                        float v = gelu1 ? fast_gelu(t) : fmaxf(t, 0.f) + sj * fminf(t, 0.f);
                        if (has_sc) v *= p.out_scale;
                        cs16[row * BN + col] = (bf16_t)(pack16x2<T>(v, 0.f) & 0xffffu);
                    }
                }
            }
            __syncthreads();
            constexpr int ITERS16 = (BM * GROUPS_PER_ROW) / NT, ROWS_PER_IT = NT / GROUPS_PER_ROW;
            const int row_a = tid / GROUPS_PER_ROW;
            bf16_t* yp = (bf16_t*)p.y + ((long long)g * a.Mg + m_tile0) * p.ldy + my_cout0;
#pragma unroll 4
            for (int it = 0; it < ITERS16; ++it) {
                const int row = row_a + it * ROWS_PER_IT;
                if (m_tile0 + row >= a.Mg) continue;
                *(uint4*)(yp + (long long)row * p.ldy) = *(const uint4*)(cs16 + row * BN + my_cg * 8);
            }
            done16 = true;
        }
    }
    if (!done16)
#pragma unroll   // at most 2 passes; unrolled so that a pass's staged accumulators are dead registers afterwards
    for (int ps = 0; ps < NPASS; ++ps) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            if (i / IPP != ps) continue;
            const int lrow0 = wm * (IPP * 32) + (i - ps * IPP) * 32;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int col = wn * WN + j * 32 + frow;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = lrow0 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
                    cs[row * BN + col] = acc[i][j][r];
                }
            }
        }
        __syncthreads();
        if (ps == 0) stamp(4);
        if (GC_LATE_WDIR && ps == 0) gc_wait();     // (see GC_EARLY)
        if (fast) {
            // ---- slim store loop of the common case (bf16 in/out, bias, none/ReLU/LeakyReLU/PReLU, optional bf16
            // residual + second activation, scale): ~40 VALU instructions per 8 channels instead of the ~150 of the
            // generic body below, which had made the epilogue VALU-issue bound (10 k cycles per 128x128 tile)
            constexpr int ITERS = (PASS_ROWS * GROUPS_PER_ROW) / NT;
            static_assert((PASS_ROWS * GROUPS_PER_ROW) % NT == 0, "whole iterations");
            constexpr int ROWS_PER_IT = NT / GROUPS_PER_ROW;
            const int row_a = tid / GROUPS_PER_ROW;       // staged row of iteration 0; iteration it adds it * ROWS_PER_IT
            const long long pix0 = (long long)g * a.Mg + m_tile0;
            bf16_t* yp = (bf16_t*)p.y + pix0 * p.ldy + my_cout0;
            const bf16_t* rp = (const bf16_t*)p.res + pix0 * p.ldr + my_cout0;
            const float* rpf = (const float*)p.res + pix0 * p.ldr + my_cout0;
            const bool res32 = F32RES && p.res_f32 != 0;
            const float* cp = cs + row_a * BN + my_cg * 8;
            const bool has_res = p.res != nullptr, has_a2 = p.act2 != GVFI_ACT_NONE, has_sc = p.out_scale != 1.0f;
            const bool do_stats = STATS && p.stats != nullptr;
            // residual vectors are fetched PF iterations at a time, all in flight together (the 8-wave tile, still
            // holding the other pass's accumulators, only has registers for 4)
            constexpr int PF = NT > 256 ? (ITERS < 2 ? ITERS : 2) : ITERS;
            static_assert(ITERS % PF == 0, "prefetch chunks");
#pragma unroll
            for (int c = 0; c < ITERS / PF; ++c) {
                uint4 rpre[PF], rpre2[F32RES ? PF : 1];
                if (has_res) {
#pragma unroll
                    for (int q = 0; q < PF; ++q) {
                        const int it = c * PF + q;
                        const int tr = tile_row(ps, row_a + it * ROWS_PER_IT);
                        if (m_tile0 + tr >= a.Mg) continue;
                        if (res32) {
                            if constexpr (F32RES) {
                                const uint4* r4 = (const uint4*)(rpf + (long long)tr * p.ldr);
                                rpre[q] = r4[0];
                                rpre2[q] = r4[1];
                            }
                        } else rpre[q] = *(const uint4*)(rp + (long long)tr * p.ldr);
                    }
                }
#pragma unroll
                for (int q = 0; q < PF; ++q) {
                    const int it = c * PF + q;
                    const int tr = tile_row(ps, row_a + it * ROWS_PER_IT);
                    if (m_tile0 + tr >= a.Mg) continue;
                    const float4 c0 = *(const float4*)(cp + it * ROWS_PER_IT * BN);
                    const float4 c1 = *(const float4*)(cp + it * ROWS_PER_IT * BN + 4);
                    float vv[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
                    if (gelu1) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float t = vv[e] + gc.bias[e];
                            vv[e] = fast_gelu(t);
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float t = vv[e] + gc.bias[e];
                            vv[e] = fmaxf(t, 0.f) + gc.s1[e] * fminf(t, 0.f);
                        }
                    }
                    if (has_res) {
                        float r[8];
                        if (res32) {
                            if constexpr (F32RES) {
                                r[0] = __builtin_bit_cast(float, rpre[q].x); r[1] = __builtin_bit_cast(float, rpre[q].y);
                                r[2] = __builtin_bit_cast(float, rpre[q].z); r[3] = __builtin_bit_cast(float, rpre[q].w);
                                r[4] = __builtin_bit_cast(float, rpre2[q].x); r[5] = __builtin_bit_cast(float, rpre2[q].y);
                                r[6] = __builtin_bit_cast(float, rpre2[q].z); r[7] = __builtin_bit_cast(float, rpre2[q].w);
                            }
                        } else unpack16x8<T>(rpre[q], r);
#pragma unroll
                        for (int e = 0; e < 8; ++e) vv[e] += r[e];
                    }
                    if (has_a2) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) vv[e] = fmaxf(vv[e], 0.f) + gc.s2[e] * fminf(vv[e], 0.f);
                    }
                    if (has_sc) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) vv[e] *= p.out_scale;
                    }
                    uint4 u;
                    u.x = pack16x2<T>(vv[0], vv[1]);
                    u.y = pack16x2<T>(vv[2], vv[3]);
                    u.z = pack16x2<T>(vv[4], vv[5]);
                    u.w = pack16x2<T>(vv[6], vv[7]);
                    if (p.y_f32) {
                        float* yf = (float*)p.y + (pix0 + tr) * p.ldy + my_cout0;
                        if (my_valid == 8) {
                            *(float4*)yf = make_float4(vv[0], vv[1], vv[2], vv[3]);
                            *(float4*)(yf + 4) = make_float4(vv[4], vv[5], vv[6], vv[7]);
                        } else {
#pragma unroll
                            for (int e = 0; e < 6; e += 2)
                                if (e < my_valid) *(float2*)(yf + e) = make_float2(vv[e], vv[e + 1]);
                        }
                    } else if (my_valid == 8) *(uint4*)(yp + (long long)tr * p.ldy) = u;
                    else {
                        uint32_t* yd = (uint32_t*)(yp + (long long)tr * p.ldy);
                        yd[0] = u.x;
                        if (my_valid > 2) yd[1] = u.y;
                        if (my_valid > 4) yd[2] = u.z;
                    }
                    if constexpr (STATS) {
                        if (do_stats) {   // statistics of the values as stored (bf16-rounded), like gvfi_instnorm_stats
                            float sv[8];
                            unpack16x8<T>(u, sv);
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                st_sum[e] += sv[e];
                                st_sq[e] += sv[e] * sv[e];
                            }
                        }
                    }
                }
            }
            if (ps + 1 < NPASS) __syncthreads();
            continue;
        }
        if (fast_gru) {
            // ---- slim ConvGRU store loops (raft/update.py:58-73): z / r gates and the h update, state operands of
            // the whole pass prefetched, hardware exp / rcp
            constexpr int ITERS = (PASS_ROWS * GROUPS_PER_ROW) / NT;
            constexpr int ROWS_PER_IT = NT / GROUPS_PER_ROW;
            const int row_a = tid / GROUPS_PER_ROW;
            const long long pix0 = (long long)g * a.Mg + m_tile0;
            const bool is_q = p.epi_mode == GVFI_EPI_GRU_Q;
            const bool sf = p.state_f32 != 0;    // float recurrent state: h (aux0), z (y of ZR / aux1 of Q) float, + y2 of Q
            const int half = p.Cout >> 1;
            const bool zhalf = !is_q && my_cout0 < half;
            const int c0 = (is_q || zhalf) ? my_cout0 : my_cout0 - half;
            bf16_t* yp = (zhalf || is_q ? (bf16_t*)p.y + pix0 * p.ldy : (bf16_t*)p.y2 + pix0 * p.ldy2) + c0;
            const int ldo = (zhalf || is_q) ? p.ldy : p.ldy2;
            float* yzf = (float*)p.y + pix0 * p.ldy + c0;               // (sf) z as float
            float* yhf = (float*)p.y2 + pix0 * p.ldy2 + c0;             // (sf, Q) new state as float
            const bf16_t* hp = (const bf16_t*)p.aux0 + pix0 * p.lda0 + c0;
            const bf16_t* zp = (const bf16_t*)p.aux1 + pix0 * p.lda1 + c0;
            const float* hpf = (const float*)p.aux0 + pix0 * p.lda0 + c0;
            const float* zpf = (const float*)p.aux1 + pix0 * p.lda1 + c0;
            const float* cp = cs + row_a * BN + my_cg * 8;
            // pre-activation context term (f32 [pixel][Cout]): the part of the gate convolution that reads RAFT's
            // constant context features is evaluated once per forward, not once per iteration
            const float* rp = (const float*)p.res + pix0 * p.ldr + my_cout0;
            const bool has_ctx = p.res != nullptr;
            auto unpack_f32x8 = [](const uint4 (&u)[2], float (&o)[8]) {
                o[0] = __builtin_bit_cast(float, u[0].x); o[1] = __builtin_bit_cast(float, u[0].y);
                o[2] = __builtin_bit_cast(float, u[0].z); o[3] = __builtin_bit_cast(float, u[0].w);
                o[4] = __builtin_bit_cast(float, u[1].x); o[5] = __builtin_bit_cast(float, u[1].y);
                o[6] = __builtin_bit_cast(float, u[1].z); o[7] = __builtin_bit_cast(float, u[1].w);
            };
            auto store_f32x8 = [](float* dst, const float (&v)[8]) {
                *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
                *(float4*)(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
            };
            // rows are processed PF iterations at a time, the state operands of a group all in flight together: the whole
            // pass with bf16 state, two iterations with float state (twice the registers per row)
            auto gru_rows = [&](auto sf_tag) {
                constexpr bool SF = decltype(sf_tag)::value;
                constexpr int PF = (SF || GC_LATE_WDIR) ? (ITERS < 2 ? ITERS : 2) : ITERS;
                static_assert(ITERS % PF == 0, "prefetch groups");
#pragma unroll
                for (int c = 0; c < ITERS / PF; ++c) {
                    uint4 cpre[PF][2], hpre[PF][SF ? 2 : 1], zpre[PF][SF ? 2 : 1];
#pragma unroll
                    for (int q = 0; q < PF; ++q) {
                        const int it = c * PF + q;
                        const int tr = tile_row(ps, row_a + it * ROWS_PER_IT);
                        if (m_tile0 + tr >= a.Mg) continue;
                        if (has_ctx) {
                            const uint4* q4 = (const uint4*)(rp + (long long)tr * p.ldr);
                            cpre[q][0] = q4[0];
                            cpre[q][1] = q4[1];
                        }
                        if (!zhalf) {
                            if constexpr (SF) {
                                const uint4* h4 = (const uint4*)(hpf + (long long)tr * p.lda0);
                                hpre[q][0] = h4[0];
                                hpre[q][1] = h4[1];
                                if (is_q) {
                                    const uint4* z4 = (const uint4*)(zpf + (long long)tr * p.lda1);
                                    zpre[q][0] = z4[0];
                                    zpre[q][1] = z4[1];
                                }
                            } else {
                                hpre[q][0] = *(const uint4*)(hp + (long long)tr * p.lda0);
                                if (is_q) zpre[q][0] = *(const uint4*)(zp + (long long)tr * p.lda1);
                            }
                        }
                    }
#pragma unroll
                    for (int q = 0; q < PF; ++q) {
                        const int it = c * PF + q;
                        const int tr = tile_row(ps, row_a + it * ROWS_PER_IT);
                        if (m_tile0 + tr >= a.Mg) continue;
                        const float4 c0v = *(const float4*)(cp + it * ROWS_PER_IT * BN);
                        const float4 c1v = *(const float4*)(cp + it * ROWS_PER_IT * BN + 4);
                        float vv[8] = {c0v.x, c0v.y, c0v.z, c0v.w, c1v.x, c1v.y, c1v.z, c1v.w};
                        float hh[8], zz[8];
                        if (!zhalf) {
                            if constexpr (SF) unpack_f32x8(hpre[q], hh); else unpack16x8<T>(hpre[q][0], hh);
                        }
                        if (is_q) {
                            if constexpr (SF) unpack_f32x8(zpre[q], zz); else unpack16x8<T>(zpre[q][0], zz);
                        }
                        if (has_ctx) {
                            float cc[8];
                            unpack_f32x8(cpre[q], cc);
#pragma unroll
                            for (int e = 0; e < 8; ++e) vv[e] += cc[e];
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float t = vv[e] + gc.bias[e];
                            if (is_q) vv[e] = (1.f - zz[e]) * hh[e] + zz[e] * fast_tanh(t);
                            else vv[e] = zhalf ? fast_sigmoid(t) : fast_sigmoid(t) * hh[e];
                        }
                        if (SF && zhalf) {
                            store_f32x8(yzf + (long long)tr * p.ldy, vv);       // z stays float
                            continue;
                        }
                        uint4 u;
                        u.x = pack16x2<T>(vv[0], vv[1]);
                        u.y = pack16x2<T>(vv[2], vv[3]);
                        u.z = pack16x2<T>(vv[4], vv[5]);
                        u.w = pack16x2<T>(vv[6], vv[7]);
                        *(uint4*)(yp + (long long)tr * ldo) = u;
                        if (SF && is_q && p.y2 != nullptr) store_f32x8(yhf + (long long)tr * p.ldy2, vv);   // the state itself
                    }
                }
            };
            if (sf) gru_rows(std::true_type{}); else gru_rows(std::false_type{});
            if (ps + 1 < NPASS) __syncthreads();
            continue;
        }
#pragma unroll 1
        for (int idx = tid; idx < PASS_ROWS * GROUPS_PER_ROW; idx += NT) {
            const int row = idx / GROUPS_PER_ROW;
            const int cg = my_cg;
            const long long m = m_tile0 + tile_row(ps, row);
            const int cout0 = my_cout0;
            if (m >= a.Mg || my_valid == 0) continue;
            const int n_valid = my_valid;
            float vv[8];
            const float4 c0 = *(const float4*)(cs + row * BN + cg * 8);
            const float4 c1 = *(const float4*)(cs + row * BN + cg * 8 + 4);
            vv[0] = c0.x; vv[1] = c0.y; vv[2] = c0.z; vv[3] = c0.w; vv[4] = c1.x; vv[5] = c1.y; vv[6] = c1.z; vv[7] = c1.w;
            epilogue_group<T>(p, gc, vv, cout0, n_valid, (long long)g * a.Mg + m, vec_all && n_valid == 8);
        }
        if (ps + 1 < NPASS) __syncthreads();
    }
    if constexpr (STATS)
    if (p.stats != nullptr) {   // (uniform: threads of channel groups beyond Cout carry zeros but must reach the barriers)
        // workgroup reduction through the (now free) staging area: [thread][16] partials -> one atomic pair per channel.
        // The tile lies inside one image (Ho*Wo % BM == 0, checked on the host).
        __syncthreads();
        float* red = cs;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[tid * 16 + e] = st_sum[e];
            red[tid * 16 + 8 + e] = st_sq[e];
        }
        __syncthreads();
        if (tid < BN) {   // channel n0 + tid: partials of the NT / GROUPS_PER_ROW threads that own its group
            const int cgi = tid >> 3, e = tid & 7;
            float s0 = 0.f, s1 = 0.f;
            for (int t = cgi; t < NT; t += GROUPS_PER_ROW) {
                s0 += red[t * 16 + e];
                s1 += red[t * 16 + 8 + e];
            }
            const int c = n0 + tid;
            if (c < p.Cout) {
                const long long img = ((long long)g * a.Mg + m_tile0) / HoWo;
                gvfi_stats_add(p.stats, img * p.Cout + c, s0, s1);
            }
        }
    }
    stamp(5);
#ifndef GVFI_HOSTSIM
    if (a.dbg & 128) {   // profiling only: time until this wave's stores are acknowledged
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(6);
    }
#endif
}

#ifndef GVFI_HOSTSIM
// the 64 x 128 weights-direct tile is budgeted for three workgroups per CU (<= 168 registers per wave) in the GVFI_WDIR_OCC3
// build: the recurrence runs two independent launch sequences, and a free slot lets the other sequence's next kernel start
// its prologue under this one's K loop
// The 128 x 128 weights-direct tile (round 5) is budgeted for TWO workgroups per CU: unconstrained hipcc takes 317 registers
// (occupancy 1: the variant lost in round 3); inside 256 -- with the bias / slope fetch moved behind the K loop -- the steady-state
// K loop is spill-free and 24 scratch accesses remain in the once-per-workgroup tail sequences.  Twice the rows per tile = half
// the weight stream per pixel, the bound of the recurrences.
#define GVFI_GLDS_KERNEL_ATTRS                                                                                        \
    __attribute__((amdgpu_waves_per_eu((BDIR && BM == 128) ? 2 : ((GVFI_WDIR_OCC3 && BDIR && BN == 128 && BM == 64) ? 3 : 1),     \
                                       (BDIR && BM == 128) ? 2 : ((GVFI_WDIR_OCC3 && BDIR && BN == 128 && BM == 64) ? 3 : 8))))
#else
#define GVFI_GLDS_KERNEL_ATTRS
#endif
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int KB, int NSTAGE, bool PIPE, int PPS = 1, bool BDIR = false>
__global__ void __launch_bounds__(64 * WAVES_M * WAVES_N) GVFI_GLDS_KERNEL_ATTRS conv_igemm_glds_kernel(ConvArgs2 a) {
    conv_igemm_glds_body<T, BM, BN, WAVES_M, WAVES_N, KB, NSTAGE, PIPE, PPS, BDIR>(a, (int)blockIdx.x, (int)blockIdx.z);
}
// Two INDEPENDENT convolutions in one launch (round 5): workgroups [0, grid_a) run problem a, the rest problem b.  For the
// branches of the flow estimators' motion encoder (raft/update.py:94-112: convc1 || convf1, convc2 || convf2): the small
// flow-branch layers (224 workgroups, a third of a CU slot each) ride in the shadow of the correlation-branch layers instead of
// costing a launch of their own on the iteration's critical path.  Same body, same arithmetic: bit-identical to two launches.
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int KB, int NSTAGE, bool PIPE, int PPS = 1, bool BDIR = false>
__global__ void __launch_bounds__(64 * WAVES_M * WAVES_N) GVFI_GLDS_KERNEL_ATTRS conv_igemm_glds_pair_kernel(ConvArgs2 a, ConvArgs2 b) {
    const int ga = a.per_xcd * 8;           // (a multiple of 8: the XCD round-robin phase of problem b's workgroups is kept)
    const int bx = (int)blockIdx.x;
    if (bx < ga) conv_igemm_glds_body<T, BM, BN, WAVES_M, WAVES_N, KB, NSTAGE, PIPE, PPS, BDIR>(a, bx, 0);
    else conv_igemm_glds_body<T, BM, BN, WAVES_M, WAVES_N, KB, NSTAGE, PIPE, PPS, BDIR>(b, bx - ga, 0);
}

template <int BM, int BN>
static void fill_args(const gvfi_conv_params& p, ConvArgs2& a, int bke) {
    a.p = p;
    a.chunks0 = p.c0 / bke;
    a.chunks_tap = (p.c0 + p.c1) / bke;
    a.KT = p.KH * p.KW * a.chunks_tap;
    a.Ktot = (long long)p.KH * p.KW * (p.c0 + p.c1);
    const int groups = p.groups > 0 ? p.groups : 1;
    a.Mg = (int)(((long long)p.N * p.Ho * p.Wo) / groups);
    a.MT = cdiv(a.Mg, BM);
    a.NT = cdiv(p.Cout, BN);
    a.per_xcd = cdiv((long long)a.MT * a.NT, 8);
    gvfi_magic_div((unsigned)(p.Ho * p.Wo), a.howo_mul, a.howo_sh);
    gvfi_magic_div((unsigned)p.Wo, a.wo_mul, a.wo_sh);
    a.dbg = (p.algo >> 8) & 0xff;   // profiling switches: algo bits 8.. (8 = no epilogue, 16 = no K loop)
}

template <typename T>
static int launch_glds_pair(const gvfi_conv_params& pa, const gvfi_conv_params& pb, hipStream_t stream) {
    ConvArgs2 a, b;
    fill_args<64, 128>(pa, a, 128 / (int)sizeof(T));
    fill_args<64, 128>(pb, b, 128 / (int)sizeof(T));
    dim3 grid((a.per_xcd + b.per_xcd) * 8, 1, 1);
    GVFI_LAUNCH_COOP((conv_igemm_glds_pair_kernel<T, 64, 128, 1, 4, 128, 4, false, 1, true>), grid, dim3(256), stream, a, b);
    return (int)hipGetLastError();
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int KB, int NSTAGE, bool PIPE = false, int PPS = 1, bool BDIR = false>
static int launch_glds(const gvfi_conv_params& p, hipStream_t stream) {
    constexpr int BKE = KB / (int)sizeof(T);
    ConvArgs2 a;
    a.p = p;
    a.chunks0 = p.c0 / BKE;
    a.chunks_tap = (p.c0 + p.c1) / BKE;
    a.KT = p.KH * p.KW * a.chunks_tap;
    a.Ktot = (long long)p.KH * p.KW * (p.c0 + p.c1);
    const int groups = p.groups > 0 ? p.groups : 1;
    a.Mg = (int)(((long long)p.N * p.Ho * p.Wo) / groups);
    a.MT = cdiv(a.Mg, BM);
    a.NT = cdiv(p.Cout, BN);
    a.per_xcd = cdiv((long long)a.MT * a.NT, 8);
    gvfi_magic_div((unsigned)(p.Ho * p.Wo), a.howo_mul, a.howo_sh);
    gvfi_magic_div((unsigned)p.Wo, a.wo_mul, a.wo_sh);
    a.dbg = (p.algo >> 8) & 0xff;   // profiling switches: algo bits 8.. (8 = no epilogue, 16 = no K loop)
    dim3 grid(a.per_xcd * 8, 1, groups);
    GVFI_EMU_SERIAL(p.stats != nullptr);
    GVFI_LAUNCH_COOP((conv_igemm_glds_kernel<T, BM, BN, WAVES_M, WAVES_N, KB, NSTAGE, PIPE, PPS, BDIR>), grid, dim3(64 * WAVES_M * WAVES_N), stream, a);
    return (int)hipGetLastError();
}

// fused statistics: only the LDS-DMA kernel's slim bf16 store loop accumulates them, and only when a tile cannot
// straddle two images
extern "C" int gvfi_conv2d_stats_ok(const gvfi_conv_params* pp) {
    const gvfi_conv_params& p = *pp;
    int plan[5];
    // the mid-channel halo-staged kernel (conv_p3x3s.hip) accumulates them in its store loop too
    if ((p.algo & 15) == 5 || ((p.algo & 15) == 0 && gvfi_conv2d_p3x3s_eligible(pp) == 1)) return gvfi_conv2d_p3x3s_eligible(pp) != 0;
    if (p.w_layout == 2) return 0;   // (the recurrence layers of the weights-direct variant are not normalised)
    if (p.dtype != GVFI_BF16 || (p.algo & 15) == 1 || !((p.algo & 15) == 2 || gvfi_conv2d_glds_eligible(pp))) return 0;
    if (gvfi_conv2d_glds_plan(pp, plan) != 0 || plan[2] >= 256) return 0;   // (not in the 8-wave tile)
    if (p.epi_mode != GVFI_EPI_STD || p.y_f32 || (p.res && p.res_f32) || p.act1 > GVFI_ACT_PRELU || p.act2 > GVFI_ACT_PRELU)
        return 0;
    if ((p.groups > 1) || ((long long)p.Ho * p.Wo) % plan[1] != 0 || (p.Cout % 8) != 0) return 0;
    if ((((uintptr_t)p.y) & 15) || ((p.ldy * 2) & 15) || (p.res && ((((uintptr_t)p.res) & 15) || ((p.ldr * 2) & 15)))) return 0;
    return 1;
}

// 0 = not eligible, else the K-chunk row size in bytes (128 or 64) the LDS-DMA kernel would use
extern "C" int gvfi_conv2d_glds_eligible(const gvfi_conv_params* pp) {
    const gvfi_conv_params& p = *pp;
    if (p.pad_mode != GVFI_PAD_ZEROS || p.c0 <= 0 || p.KH * p.KW > 32) return 0;   // tap validity is a 32-bit mask
    const int e128 = p.dtype == GVFI_F32 ? 32 : 64;
    if (p.c0 % e128 == 0 && p.c1 % e128 == 0) return 128;
    if (p.w_layout != 0) return 0;   // 64-byte chunking reads the plain [Cout][K] weight image
    if (p.dtype == GVFI_F16) return 0;   // (half operands: only the 128-byte chunk variants are instantiated)
    const int e64 = e128 / 2;
    if (p.c0 % e64 == 0 && p.c1 % e64 == 0) return 64;
    return 0;
}

// One place decides the kernel variant: plan = {algo, BM, BN, KB(bytes), LDS stages} for an eligible problem (also
// reported by gvfi_conv2d_plan).  tile_hint = BN | BM << 10 (0 = auto).
//   BN: 256 (8 waves, 256x256) for Cout >= 192 on >= 65536 pixels, else 128 / 64 / 32 (4 waves);
//   KB: 128-byte K chunks (ring of 2) when the channel counts allow it, else 64-byte chunks (ring of 4; of 2 for BN = 32);
//   BM: 128, except 64 when a 128-row grid would leave CUs without a workgroup (RAFT at 1/8 resolution, Cout <= 128)
//       and 256 (waves 4 x 1) for the BN = 32 full-resolution layers (halves their per-tile fixed cost).
// Variants that were measured and did not pay (3-/4-deep rings at KB = 128, 64-row tiles on larger grids, a 256x128
// tile with two workgroups per CU, 256x64 and 512x32 tall tiles) are listed in DESIGN.md and no longer instantiated.
extern "C" int gvfi_conv2d_glds_plan(const gvfi_conv_params* pp, int* plan) {
    const gvfi_conv_params& p = *pp;
    const int kb = gvfi_conv2d_glds_eligible(pp);
    if (!kb) return -2;
    const int groups = p.groups > 0 ? p.groups : 1;
    const long long M = (long long)p.N * p.Ho * p.Wo / groups;
    if (p.w_layout == 2) {
        // weights-direct variant: fragment-ordered weight image, 64-row tiles, column tile 128 (wave slices of 32) or 256
        // (slices of 64; tile_hint 256), register / LDS ring of 4 chunks
        if (kb != 128 || p.dtype == GVFI_F32 || groups != 1) return -5;
        plan[0] = 6;
        plan[2] = (p.tile_hint & 1023) == 256 ? 256 : 128;
        plan[1] = (((p.tile_hint >> 10) & 1023) == 128 && plan[2] == 128) ? 128 : 64;     // 128-row tiles on request
        plan[3] = 128;
        plan[4] = 4;
        return 0;
    }
    int tile = p.tile_hint & 1023, bm = (p.tile_hint >> 10) & 1023;
    const int ns_hint = (p.tile_hint >> 20) & 15;   // ring depth override (0 = auto), 128-byte chunks only
    if (tile == 0) tile = (p.Cout >= 192 && M >= 256 * 256) ? 256 : (p.Cout > 64 ? 128 : (p.Cout > 32 ? 64 : 32));
    // (IEEE half: the 8-wave tile is instantiated for 128-byte chunks of the chunk-major weight image only -- round 5, the many-row
    // linears of GIMM-VFI-F's Twins encoders under the "enc:f16" policy)
    if (p.dtype == GVFI_F16 && tile >= 256 && (p.w_layout == 0 || kb != 128 || (p.algo & 128))) tile = 128;
    tile = tile >= 256 ? 256 : (tile >= 128 ? 128 : (tile >= 64 ? 64 : 32));
    int k = 64, ns = 2;
    if (tile == 256) k = p.w_layout == 0 ? 64 : 128;
    else if (kb == 128 && !(p.algo & 128)) k = 128;
    else if (p.w_layout != 0) return -5;
    if (tile == 256) {
        bm = 256;
        ns = k == 64 ? 4 : 2;
    } else if (tile == 128) {
        const long long blocks128 = (M + 127) / 128 * ((p.Cout + 127) / 128) * groups;
        if (bm == 0) bm = (k == 128 && blocks128 <= 256) ? 64 : 128;
        bm = (bm <= 64 && k == 128) ? 64 : 128;
        ns = k == 64 ? 4 : 2;
    } else if (tile == 64) {
        bm = 128;
        ns = k == 64 ? 4 : 2;
    } else {
        if (bm == 0) bm = M >= 65536 ? 256 : 128;
        bm = bm >= 256 ? 256 : 128;
    }
    if (ns_hint && k == 128 && tile < 256) ns = ns_hint < 2 ? 2 : (ns_hint > 4 ? 4 : ns_hint);
    plan[0] = 2;
    plan[1] = bm;
    plan[2] = tile;
    plan[3] = k;
    plan[4] = ns;
    return 0;
}

extern "C" int gvfi_conv2d_glds(const gvfi_conv_params* pp, void* stream) {
    const gvfi_conv_params& p = *pp;
    int plan[5];
    const int rc = gvfi_conv2d_glds_plan(pp, plan);
    if (rc) return rc;
    if (((uintptr_t)p.x0 & 15) || ((uintptr_t)p.x1 & 15) || ((uintptr_t)p.w & 15)) return -3;
    if (p.groups > 1 && (p.N % p.groups)) return -4;
    if (p.stats != nullptr && !gvfi_conv2d_stats_ok(pp)) return -6;   // statistics requested but not computable here
    hipStream_t st = (hipStream_t)stream;
    if (plan[0] == 6) {
        if (p.dtype == GVFI_F16) {
            if (plan[2] == 256) return launch_glds<f16_t, 64, 256, 1, 4, 128, 4, false, 1, true>(p, st);
            if (plan[1] == 128) return launch_glds<f16_t, 128, 128, 1, 4, 128, 4, false, 1, true>(p, st);
            return launch_glds<f16_t, 64, 128, 1, 4, 128, 4, false, 1, true>(p, st);
        }
        if (plan[2] == 256) return launch_glds<bf16_t, 64, 256, 1, 4, 128, 4, false, 1, true>(p, st);
        if (plan[1] == 128) return launch_glds<bf16_t, 128, 128, 1, 4, 128, 4, false, 1, true>(p, st);
        return launch_glds<bf16_t, 64, 128, 1, 4, 128, 4, false, 1, true>(p, st);
    }
    const int bm = plan[1], tile = plan[2], k = plan[3], ns = plan[4];
#define GLDS_DISPATCH(TT)                                                                                     \
    if (tile == 256) {                                                                                        \
        if (k == 64) return launch_glds<TT, 256, 256, 2, 4, 64, 4>(p, st);                                    \
        if (p.algo & 32) return launch_glds<TT, 256, 256, 2, 4, 128, 2, true, 1>(p, st);                      \
        return launch_glds<TT, 256, 256, 2, 4, 128, 2, true, 4>(p, st);                                       \
    }                                                                                                         \
    if (tile == 128) {                                                                                        \
        if (k == 64) return launch_glds<TT, 128, 128, 2, 2, 64, 4>(p, st);                                    \
        if (bm == 64 && ns == 3) return launch_glds<TT, 64, 128, 2, 2, 128, 3>(p, st);                        \
        if (bm == 64 && ns == 4) return launch_glds<TT, 64, 128, 2, 2, 128, 4>(p, st);                        \
        if (bm == 64) return launch_glds<TT, 64, 128, 2, 2, 128, 2>(p, st);                                   \
        if (ns == 3) return launch_glds<TT, 128, 128, 2, 2, 128, 3>(p, st);                                   \
        return launch_glds<TT, 128, 128, 2, 2, 128, 2>(p, st);                                                \
    }                                                                                                         \
    if (tile == 64) {                                                                                         \
        if (k == 64) return launch_glds<TT, 128, 64, 2, 2, 64, 4>(p, st);                                     \
        if (ns == 3) return launch_glds<TT, 128, 64, 2, 2, 128, 3>(p, st);                                    \
        if (ns == 4) return launch_glds<TT, 128, 64, 2, 2, 128, 4>(p, st);                                    \
        return launch_glds<TT, 128, 64, 2, 2, 128, 2>(p, st);                                                 \
    }                                                                                                         \
    if (bm == 256) {                                                                                          \
        if (k == 64) return launch_glds<TT, 256, 32, 4, 1, 64, 2>(p, st);                                     \
        return launch_glds<TT, 256, 32, 4, 1, 128, 2>(p, st);                                                 \
    }                                                                                                         \
    if (k == 64) return launch_glds<TT, 128, 32, 4, 1, 64, 2>(p, st);                                         \
    if (ns == 3) return launch_glds<TT, 128, 32, 4, 1, 128, 3>(p, st);                                        \
    if (ns == 4) return launch_glds<TT, 128, 32, 4, 1, 128, 4>(p, st);                                        \
    return launch_glds<TT, 128, 32, 4, 1, 128, 2>(p, st);
    if (p.dtype == GVFI_F32) { GLDS_DISPATCH(float) }
    if (p.dtype == GVFI_F16) {
        // IEEE-half operands: the tiles of the flow estimators' recurrence (small M); everything else of that size takes the
        // 128-wide 4-wave tiles
        if (k != 128) return -7;
        if (tile == 256) {
            if (p.algo & 32) return -7;      // (the one-piece-per-MFMA-group A/B variant exists for bf16 / float only)
            return launch_glds<f16_t, 256, 256, 2, 4, 128, 2, true, 4>(p, st);
        }
        if (tile == 128) {
            if (bm == 64) return launch_glds<f16_t, 64, 128, 2, 2, 128, 2>(p, st);
            return launch_glds<f16_t, 128, 128, 2, 2, 128, 2>(p, st);
        }
        if (tile == 64) return launch_glds<f16_t, 128, 64, 2, 2, 128, 2>(p, st);
        return launch_glds<f16_t, 128, 32, 4, 1, 128, 2>(p, st);
    }
    GLDS_DISPATCH(bf16_t)
#undef GLDS_DISPATCH
}

// Two independent convolutions as ONE launch (conv_igemm_glds_pair_kernel).  Both must be problems the weights-direct 64 x 128
// variant takes (w_layout 2, the same 16-bit dtype, no weight groups, no statistics): 0 = launched; -2 = not such a pair (the
// caller launches them one by one -- never a different arithmetic path, the pair kernel IS the same body).
extern "C" int gvfi_conv2d_pair(const gvfi_conv_params* pa, const gvfi_conv_params* pb, void* stream) {
    const gvfi_conv_params* ps[2] = {pa, pb};
    int plan[5];
    for (int i = 0; i < 2; ++i) {
        const gvfi_conv_params& p = *ps[i];
        if (p.w_layout != 2 || p.dtype != pa->dtype || p.dtype == GVFI_F32 || p.groups > 1 || p.stats != nullptr) return -2;
        if (((p.algo & 15) != 2 && (p.algo & 15) != 0) || gvfi_conv2d_glds_plan(&p, plan) != 0) return -2;
        if (plan[0] != 6 || plan[1] != 64 || plan[2] != 128) return -2;
        if (((uintptr_t)p.x0 & 15) || ((uintptr_t)p.x1 & 15) || ((uintptr_t)p.w & 15)) return -3;
    }
    if (pa->dtype == GVFI_F16) return launch_glds_pair<f16_t>(*pa, *pb, (hipStream_t)stream);
    return launch_glds_pair<bf16_t>(*pa, *pb, (hipStream_t)stream);
}
