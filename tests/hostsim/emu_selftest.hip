// TEST INFRASTRUCTURE ONLY -- never part of the product library (tests/hostsim/build.py compiles it into the emulator build).
// Kernels with DELIBERATE hazards: the negative controls of the emulator's schedules (GVFI_EMU_SCHED) and of its adversarial
// LDS-DMA timing (GVFI_EMU_DMA).  Each has a `fixed` switch that adds the missing barrier / wait.
#include "../../gimm-vfi_amd/csrc/conv_mma.h"

// (emulator-only file: the LDS array of the race tests is shared by the kernels that prepare and use it)
__shared__ int st_box[64];
__global__ void emu_selftest_clear_kernel() { st_box[threadIdx.x & 63] = -1; }

// wave 0 produces, wave 1 consumes, no barrier: thread order happens to run the producer first -- REVERSED wave order does not
__global__ void emu_selftest_race_kernel(int* out, int fixed) {
    const int t = threadIdx.x;
    if (t < 64) st_box[t] = t + 1;
    if (fixed) __syncthreads();
    if (t >= 64) out[t - 64] = st_box[t - 64];
}

// wave 1 produces after one wave-level exchange, wave 0 consumes after two, no barrier: waves that advance in step with each
// other get away with it -- a wave that runs ahead alone (DEPTH first) does not
__global__ void emu_selftest_skew_kernel(int* out, int fixed) {
    const int t = threadIdx.x;
    int v = t;
    v = __shfl_xor(v, 1);
    if (t >= 64) st_box[t - 64] = t - 63;
    if (t < 64) v = __shfl_xor(v, 1);
    if (fixed) __syncthreads();
    if (t < 64) out[t] = st_box[t] + (v == t ? 0 : 1000);
}

// one wave fetches 1 KiB by LDS-DMA and reads it back; without the wait the read may come before the data
__global__ void emu_selftest_dma_kernel(const unsigned char* src, int* out, int fixed) {
    __shared__ __attribute__((aligned(16))) unsigned char buf[1024];
    const int lane = threadIdx.x;
    const unsigned base = lds_address(buf);
    bufdma16((unsigned)lane * 16u, make_srd(src), 0u, base);
    if (fixed) glds_wait_n<0>();
    __syncthreads();
    out[lane] = *(const int*)(buf + lane * 16);
}

extern "C" int gvfi_emu_selftest(int which, int fixed, const unsigned char* src, int* out) {
    if (which != 1) GVFI_LAUNCH_COOP(emu_selftest_clear_kernel, dim3(1), dim3(64), nullptr);
    if (which == 0) GVFI_LAUNCH_COOP(emu_selftest_race_kernel, dim3(1), dim3(128), nullptr, out, fixed);
    else if (which == 1) GVFI_LAUNCH_COOP(emu_selftest_dma_kernel, dim3(1), dim3(64), nullptr, src, out, fixed);
    else GVFI_LAUNCH_COOP(emu_selftest_skew_kernel, dim3(1), dim3(128), nullptr, out, fixed);
    return 0;
}
