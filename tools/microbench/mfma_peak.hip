// Sustained MFMA ceiling of the device, and what the shader clock does under it (evidence for the
// "power limit" reading of the hot convolution's 0.50 of peak).
//
//   pure   : every wave issues independent v_mfma_f32_32x32x16_bf16 back to back, no memory traffic.  A 32x32x16 bf16
//            MFMA occupies its SIMD's matrix pipe for 8 passes = 32 cycles, so with the pipe never idle
//                 sustained clock = (MFMAs per SIMD x 32 cycles) / time
//            -- the figure printed as "implied clock"; 2 500 TFLOP/s is the dense peak quoted at 2.4 GHz.
//   lds    : the same loop with the hot kernel's operand traffic: per 8 MFMAs six ds_read_b128 fragment reads
//            (4 A + 2 B of a 128x64 wave tile), consumed by the MFMAs.
//   lds+dma: additionally one 1 KiB `buffer_load_dwordx4 ... lds` per wave and 8 MFMAs from an L2-resident buffer
//            (conv_p3x3: 37 KB per 256 MFMAs of a workgroup = 1.15 KB per wave and 8 MFMAs).
// Each variant runs for ~0.1 ms, ~1 ms and ~10 ms: the drop from the short to the long run is the DVFS give-back.
// Every variant runs twice: with CONSTANT operands (all lanes the same small values: almost no switching in the
// multiplier arrays) and with pseudo-RANDOM bf16 operands (what a real convolution feeds them) -- dynamic power, and with
// it the sustained clock, depends on the operand data.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ i32x4 make_srd(const void* base) {
    const unsigned long long a = (unsigned long long)base;
    i32x4 r;
    r[0] = (int)(unsigned)a;
    r[1] = (int)((unsigned)(a >> 32) & 0xffffu);
    r[2] = 0x7fffff00;
    r[3] = 0x00020000;
    return r;
}
__device__ __forceinline__ void bufdma16(unsigned voff, i32x4 srd, unsigned soff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds\n\ts_nop 0" ::"v"(voff), "s"(srd), "s"(soff),
                 "s"(lds_addr)
                 : "memory");
}

// MODE 0 pure, 1 lds fragments, 2 lds fragments + LDS-DMA
__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// two bf16 values in [-2, 2) with random mantissas
__device__ __forceinline__ unsigned rnd_bf16x2(unsigned seed) {
    const unsigned h = hash32(seed);
    return (h & 0x807f807fu) | 0x3f803f80u;     // sign + 7 mantissa bits random, exponent of 1.0
}

template <int MODE>
__global__ void __launch_bounds__(512) peak_kernel(float* out, const unsigned char* src, int iters, int random_data) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[112 * 1024];
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 112 * 1024 / 4; i += 512)
        ((unsigned*)smem)[i] = random_data ? rnd_bf16x2(i * 2654435761u + blockIdx.x) : 0x3c003c00u;   // small bf16 values
    __syncthreads();
    bf16x8 a, b, a2, b2;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x & 3); b[j] = (__bf16)1.0f; }
    a2 = a; b2 = b;
    if (random_data) {
        uint4 ra, rb, rc, rd;
        const unsigned t = threadIdx.x * 16u + blockIdx.x * 8192u;
        ra = make_uint4(rnd_bf16x2(t), rnd_bf16x2(t + 1), rnd_bf16x2(t + 2), rnd_bf16x2(t + 3));
        rb = make_uint4(rnd_bf16x2(t + 4), rnd_bf16x2(t + 5), rnd_bf16x2(t + 6), rnd_bf16x2(t + 7));
        rc = make_uint4(rnd_bf16x2(t + 8), rnd_bf16x2(t + 9), rnd_bf16x2(t + 10), rnd_bf16x2(t + 11));
        rd = make_uint4(rnd_bf16x2(t + 12), rnd_bf16x2(t + 13), rnd_bf16x2(t + 14), rnd_bf16x2(t + 15));
        a = __builtin_bit_cast(bf16x8, ra); b = __builtin_bit_cast(bf16x8, rb);
        a2 = __builtin_bit_cast(bf16x8, rc); b2 = __builtin_bit_cast(bf16x8, rd);
    }
    const i32x4 srd = make_srd(src);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)smem);
    // fragment addresses: 16 B per lane, conflict-free (lane-linear), six different 1 KiB blocks per wave, in one of
    // two 32 KiB-apart windows (bytes 0 .. 80 Ki); the DMA lands above them (80 Ki ..)
    const unsigned char* fr = smem + wave * 6144 + lane * 16;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i)     // operands alternate between two register sets, as a real k-step sequence does
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((i & 1) ? a2 : a, (i & 2) ? b2 : b, acc[i], 0, 0, 0);
        } else {
            uint4 f[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) f[q] = *(const uint4*)(fr + q * 1024 + ((it & 1) << 15));   // (address varies: not loop invariant)
            if (MODE == 2)   // 1 KiB per wave into the upper half of the buffer, source walks a 1 MiB window (L2 hits)
                bufdma16((unsigned)lane * 16u, srd, (unsigned)(((it * 8 + wave) & 1023) * 1024), lds0 + 81920 + wave * 1024);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[i]), __builtin_bit_cast(bf16x8, f[4 + j]),
                                                                            acc[i * 2 + j], 0, 0, 0);
        }
    }
    if (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, float* out, const unsigned char* src, int random_data) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = 256;   // one 8-wave workgroup per CU = 2 waves per SIMD (the hot kernel's occupancy)
    for (int iters : {400, 4000, 40000}) {
        float best = 1e30f, last = 0.f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            peak_kernel<MODE><<<blocks, 512>>>(out, src, iters, random_data);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
            last = ms;
        }
        const double mfma_per_simd = 2.0 * iters * 8;                // 2 waves per SIMD
        const double flop = (double)blocks * 8 * iters * 8 * 32768.0;
        printf("%-8s %s iters=%6d  best %8.3f ms (last %8.3f)  %7.1f TFLOP/s = %.3f of 2500   implied clock >= %.2f GHz\n", name, random_data ? "random  " : "constant", iters, best, last,
               flop / best / 1e9, flop / best / 1e9 / 2500.0, mfma_per_simd * 32.0 / (best * 1e-3) / 1e9);
    }
}

int main() {
    float* out;
    unsigned char* src;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&src, 2 << 20);
    {   // DMA source: random bytes too (the LDS-DMA writes toggle the LDS arrays)
        unsigned* h = (unsigned*)malloc(2 << 20);
        unsigned x = 12345u;
        for (int i = 0; i < (2 << 20) / 4; ++i) { x = x * 1664525u + 1013904223u; h[i] = (x & 0x807f807fu) | 0x3f803f80u; }
        hipMemcpy(src, h, 2 << 20, hipMemcpyHostToDevice);
        free(h);
    }
    for (int rd = 0; rd < 2; ++rd) {
        run<0>("pure", out, src, rd);
        run<1>("lds", out, src, rd);
        run<2>("lds+dma", out, src, rd);
    }
    // back-to-back ~10 ms launches for ~1 s: what the clock settles to under a sustained matrix load
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int burst = 0; burst < 6; ++burst) {
        const int rd = burst >= 3;
        hipEventRecord(e0);
        for (int k = 0; k < 20; ++k) peak_kernel<2><<<256, 512>>>(out, src, 40000, rd);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = 20.0 * 256 * 8 * 40000.0 * 8 * 32768.0;
        printf("sustained lds+dma %s burst %d: %.1f ms  %7.1f TFLOP/s  implied clock >= %.2f GHz\n", rd ? "random" : "constant", burst, ms, flop / ms / 1e9,
               20.0 * 2.0 * 40000 * 8 * 32.0 / (ms * 1e-3) / 1e9);
    }
    return 0;
}
