// What does straight-line code cost at kernel start?  Every launch begins with a cold instruction cache; the prologue and
// epilogue of the convolution kernels are ~1 000 instructions each that every wave executes ONCE.  This kernel runs a block
// of N independent-ish VALU instructions twice (a two-trip loop over the same addresses) and stamps s_memtime around each
// trip: trip 0 = cold instruction fetch, trip 1 = warm.  One wave per SIMD (256 threads per workgroup, 1-2 workgroups per CU).
// Build: hipcc --offload-arch=gfx950 -O3 icache.hip -o icache
#include <hip/hip_runtime.h>
#include <cstdio>

#define OP4(a, b, c, d) a = a * 1.0001f + b; b = b * 0.9999f + c; c = c * 1.0002f + d; d = d * 0.9998f + a;
#define OP16 OP4(x0, x1, x2, x3) OP4(x1, x2, x3, x0) OP4(x2, x3, x0, x1) OP4(x3, x0, x1, x2)
#define OP64 OP16 OP16 OP16 OP16
#define OP256 OP64 OP64 OP64 OP64
#define OP1K OP256 OP256 OP256 OP256

template <int KILO>
__global__ void __launch_bounds__(256) icache_kernel(unsigned long long* stamps, float* out, float seed) {
    float x0 = seed + threadIdx.x, x1 = seed * 2.f, x2 = seed * 3.f, x3 = seed * 4.f;
    unsigned long long t[4];
#pragma unroll 1
    for (int trip = 0; trip < 2; ++trip) {
        t[2 * trip] = __builtin_readcyclecounter();
        OP1K
        if (KILO >= 2) { OP1K }
        if (KILO >= 4) { OP1K OP1K }
        if (KILO >= 8) { OP1K OP1K OP1K OP1K }
        asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        t[2 * trip + 1] = __builtin_readcyclecounter();
    }
    if (threadIdx.x == 0) {
        stamps[blockIdx.x * 4 + 0] = t[1] - t[0];
        stamps[blockIdx.x * 4 + 1] = t[3] - t[2];
        stamps[blockIdx.x * 4 + 2] = t[0];
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3;
}

template <int KILO>
static void run(int blocks) {
    unsigned long long* st;
    float* out;
    hipMalloc(&st, blocks * 4 * 8);
    hipMalloc(&out, blocks * 256 * 4);
    unsigned long long* h = (unsigned long long*)malloc(blocks * 4 * 8);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(icache_kernel<KILO>, dim3(blocks), dim3(256), 0, 0, st, out, 1.0f + rep);
        hipDeviceSynchronize();
    }
    hipMemcpy(h, st, blocks * 4 * 8, hipMemcpyDeviceToHost);
    double c0 = 0, c1 = 0;
    unsigned long long tmin = ~0ull;
    for (int b = 0; b < blocks; ++b) tmin = h[b * 4 + 2] < tmin ? h[b * 4 + 2] : tmin;
    // early workgroups (started within 2000 cycles of the first) vs the rest
    double e0 = 0, e1 = 0; int ne = 0;
    for (int b = 0; b < blocks; ++b) {
        c0 += h[b * 4]; c1 += h[b * 4 + 1];
        if (h[b * 4 + 2] - tmin < 2000) { e0 += h[b * 4]; e1 += h[b * 4 + 1]; ++ne; }
    }
    const int ninstr = KILO * 1024;
    printf("%5d instructions (~%3d KB of code), %4d workgroups: cold trip %8.0f cycles (%.2f / instr), warm trip %8.0f (%.2f / instr); "
           "first-wave workgroups (%d): cold %8.0f warm %8.0f\n", ninstr, ninstr * 8 / 1024, blocks, c0 / blocks, c0 / blocks / ninstr,
           c1 / blocks, c1 / blocks / ninstr, ne, ne ? e0 / ne : 0.0, ne ? e1 / ne : 0.0);
    hipFree(st); hipFree(out); free(h);
}

int main() {
    for (int blocks : {256, 512, 1024}) {
        run<1>(blocks);
        run<2>(blocks);
        run<4>(blocks);
        run<8>(blocks);
    }
    return 0;
}
