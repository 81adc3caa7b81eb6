// Probe: does a vector-memory instruction still read its SOURCE VGPRs (address / store data) after the wave has been allowed
// to issue the next VALU instruction?  Each variant issues one VMEM instruction from fixed registers, then NOPS wait states,
// then a VALU write of one of the VMEM instruction's source registers.  Architecturally the VALU write must not be seen by the
// VMEM instruction (for stores of more than 64 bits the ISA asks for 1 wait state; loads have no documented hazard).  The host
// (tools/vmem_src_hazard.py) runs the probe alone and beside LDS-DMA kernels on another stream and counts wrong elements.
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/microbench/libvmem_src_hazard.so tools/microbench/vmem_src_hazard.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NOP0 ""
#define NOP1 "s_nop 0\n\t"
#define NOP2 "s_nop 1\n\t"
#define NOP4 "s_nop 3\n\t"
#define NOP8 "s_nop 7\n\t"
#define NOP16 "s_nop 7\n\ts_nop 7\n\t"
#define NOP32 "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"

// MODE 0: load, 32-bit offset + SGPR base; the offset register is overwritten with 0 afterwards (a hit reads element 0)
#define LOAD_SADDR(NOPS)                                                                                                  \
    asm volatile("v_mov_b32 v20, %4\n\t"                                                                                  \
                 "global_load_dwordx4 v[24:27], v20, %5\n\t" NOPS "v_mov_b32 v20, 0\n\t"                                  \
                 "s_waitcnt vmcnt(0)\n\t"                                                                                 \
                 "v_mov_b32 %0, v24\n\tv_mov_b32 %1, v25\n\tv_mov_b32 %2, v26\n\tv_mov_b32 %3, v27\n\t"                   \
                 : "=v"(a), "=v"(b), "=v"(c), "=v"(d)                                                                     \
                 : "v"(voff), "s"(src)                                                                                    \
                 : "v20", "v24", "v25", "v26", "v27", "memory")
// MODE 1: load, 64-bit address in a VGPR pair; the low half is overwritten with the low half of the base (a hit reads element 0
// of the 4 GB window the element lives in -- the host keeps the buffer inside one window)
#define LOAD_VADDR(NOPS)                                                                                                  \
    asm volatile("v_mov_b32 v20, %4\n\tv_mov_b32 v21, %5\n\t"                                                             \
                 "global_load_dwordx4 v[24:27], v[20:21], off\n\t" NOPS "v_mov_b32 v20, %6\n\t"                           \
                 "s_waitcnt vmcnt(0)\n\t"                                                                                 \
                 "v_mov_b32 %0, v24\n\tv_mov_b32 %1, v25\n\tv_mov_b32 %2, v26\n\tv_mov_b32 %3, v27\n\t"                   \
                 : "=v"(a), "=v"(b), "=v"(c), "=v"(d)                                                                     \
                 : "v"(alo), "v"(ahi), "v"(blo)                                                                           \
                 : "v20", "v21", "v24", "v25", "v26", "v27", "memory")
// MODE 2: 16-byte store; the first and last data registers are overwritten afterwards (a hit stores the marker -7)
#define STORE_X4(NOPS)                                                                                                    \
    asm volatile("v_mov_b32 v20, %0\n\tv_mov_b32 v24, %2\n\tv_mov_b32 v25, %3\n\tv_mov_b32 v26, %4\n\tv_mov_b32 v27, %5\n\t" \
                 "global_store_dwordx4 v20, v[24:27], %1\n\t" NOPS "v_mov_b32 v24, %6\n\tv_mov_b32 v27, %6\n\t"           \
                 "s_waitcnt vmcnt(0)\n\t"                                                                                 \
                 :                                                                                                        \
                 : "v"(voff), "s"(dst), "v"(a), "v"(b), "v"(c), "v"(d), "v"(marker)                                       \
                 : "v20", "v24", "v25", "v26", "v27", "memory")
// MODE 3: 4-byte store; the data register is overwritten afterwards
#define STORE_X1(NOPS)                                                                                                    \
    asm volatile("v_mov_b32 v20, %0\n\tv_mov_b32 v24, %2\n\t"                                                             \
                 "global_store_dword v20, v24, %1\n\t" NOPS "v_mov_b32 v24, %3\n\t"                                       \
                 "s_waitcnt vmcnt(0)\n\t"                                                                                 \
                 :                                                                                                        \
                 : "v"(voff), "s"(dst), "v"(a), "v"(marker)                                                               \
                 : "v20", "v24", "memory")

// MODES 4..7: is load data in the destination VGPRs when s_waitcnt vmcnt(0) releases the wave?  NOPS wait states sit between the
// s_waitcnt and the first VALU read of the destination.  The expected element is idx + 1 (the host compares with src shifted by one).
// MODE 4: A = x3 load of element idx, wait, D = x3 load of element idx + 1 (immediate offset 16) into the SAME registers, wait, read
#define RELOAD_X3(NOPS)                                                                                                   \
    asm volatile("v_mov_b32 v20, %4\n\tv_mov_b32 v21, %5\n\t"                                                             \
                 "global_load_dwordx3 v[24:26], v[20:21], off\n\t"                                                        \
                 "s_waitcnt vmcnt(0)\n\t"                                                                                 \
                 "v_add_f32 v27, v24, v25\n\t"                                                                            \
                 "global_load_dwordx3 v[24:26], v[20:21], off offset:16\n\t"                                              \
                 "s_waitcnt vmcnt(0)\n\t" NOPS                                                                            \
                 "v_mov_b32 %0, v24\n\tv_mov_b32 %1, v25\n\tv_mov_b32 %2, v26\n\tv_mov_b32 %3, v27\n\t"                   \
                 : "=v"(a), "=v"(b), "=v"(c), "=v"(d)                                                                     \
                 : "v"(alo), "v"(ahi)                                                                                     \
                 : "v20", "v21", "v24", "v25", "v26", "v27", "memory")
// MODE 5: the same with x4 loads
#define RELOAD_X4(NOPS)                                                                                                   \
    asm volatile("v_mov_b32 v20, %4\n\tv_mov_b32 v21, %5\n\t"                                                             \
                 "global_load_dwordx4 v[24:27], v[20:21], off\n\t"                                                        \
                 "s_waitcnt vmcnt(0)\n\t"                                                                                 \
                 "v_add_f32 v28, v24, v25\n\t"                                                                            \
                 "global_load_dwordx4 v[24:27], v[20:21], off offset:16\n\t"                                              \
                 "s_waitcnt vmcnt(0)\n\t" NOPS                                                                            \
                 "v_mov_b32 %0, v24\n\tv_mov_b32 %1, v25\n\tv_mov_b32 %2, v26\n\tv_mov_b32 %3, v27\n\t"                   \
                 : "=v"(a), "=v"(b), "=v"(c), "=v"(d)                                                                     \
                 : "v"(alo), "v"(ahi)                                                                                     \
                 : "v20", "v21", "v24", "v25", "v26", "v27", "v28", "memory")
// MODE 6: one x3 load of element idx + 1 into registers that hold the marker (a stale read shows the marker, a dropped immediate
// offset shows element idx)
#define SINGLE_X3(NOPS)                                                                                                   \
    asm volatile("v_mov_b32 v20, %4\n\tv_mov_b32 v21, %5\n\t"                                                             \
                 "v_mov_b32 v24, %6\n\tv_mov_b32 v25, %6\n\tv_mov_b32 v26, %6\n\t"                                        \
                 "global_load_dwordx3 v[24:26], v[20:21], off offset:16\n\t"                                              \
                 "s_waitcnt vmcnt(0)\n\t" NOPS                                                                            \
                 "v_mov_b32 %0, v24\n\tv_mov_b32 %1, v25\n\tv_mov_b32 %2, v26\n\tv_mov_b32 %3, v26\n\t"                   \
                 : "=v"(a), "=v"(b), "=v"(c), "=v"(d)                                                                     \
                 : "v"(alo), "v"(ahi), "v"(marker)                                                                        \
                 : "v20", "v21", "v24", "v25", "v26", "memory")
// MODE 7: x2 load of element i+1 into registers holding a marker, wait, N wait states, then a PACKED fp32 instruction multiplies the pair IN PLACE by 1.0
#define SINGLE_X2_PK(NOPS)                                                                                                \
    asm volatile("v_mov_b32 v20, %4\n\tv_mov_b32 v21, %5\n\t"                                                             \
                 "v_mov_b32 v30, 1.0\n\tv_mov_b32 v31, 1.0\n\t"                                                           \
                 "v_mov_b32 v24, %6\n\tv_mov_b32 v25, %6\n\t"                                                             \
                 "global_load_dwordx2 v[24:25], v[20:21], off offset:16\n\t"                                              \
                 "s_waitcnt vmcnt(0)\n\t" NOPS                                                                            \
                 "v_pk_mul_f32 v[24:25], v[30:31], v[24:25]\n\t"                                                          \
                 "s_nop 7\n\t"                                                                                            \
                 "v_mov_b32 %0, v24\n\tv_mov_b32 %1, v25\n\tv_mov_b32 %2, v24\n\tv_mov_b32 %3, v24\n\t"                   \
                 : "=v"(a), "=v"(b), "=v"(c), "=v"(d)                                                                     \
                 : "v"(alo), "v"(ahi), "v"(marker)                                                                        \
                 : "v20", "v21", "v24", "v25", "v28", "v29", "v30", "v31", "memory")
// MODE 8: the compiler's conditional-tap shape: s_and_saveexec (all lanes on), x2 load, two VALU instructions, wait, N wait states,
// packed fp32 multiply of the pair in place by 1.0
#define COND_X2_PK(NOPS)                                                                                                  \
    asm volatile("v_mov_b32 v20, %4\n\tv_mov_b32 v21, %5\n\t"                                                             \
                 "v_mov_b32 v30, 1.0\n\tv_mov_b32 v31, 1.0\n\t"                                                           \
                 "v_mov_b32 v24, %6\n\tv_mov_b32 v25, %6\n\t"                                                             \
                 "v_cmp_eq_u32_e64 s[22:23], v21, v21\n\t"                                                                \
                 "s_and_saveexec_b64 s[20:21], s[22:23]\n\t"                                                              \
                 "global_load_dwordx2 v[24:25], v[20:21], off offset:16\n\t"                                              \
                 "v_mov_b32 v32, v31\n\t"                                                                                 \
                 "v_pk_mul_f32 v[30:31], v[30:31], v[30:31]\n\t"                                                          \
                 "s_waitcnt vmcnt(0)\n\t" NOPS                                                                            \
                 "v_pk_mul_f32 v[24:25], v[30:31], v[24:25]\n\t"                                                          \
                 "s_nop 0\n\t"                                                                                            \
                 "v_pk_add_f32 v[28:29], v[24:25], v[24:25] neg_lo:[0,1] neg_hi:[0,1]\n\t"                                \
                 "s_or_b64 exec, exec, s[20:21]\n\t"                                                                      \
                 "s_nop 7\n\t"                                                                                            \
                 "v_mov_b32 %0, v24\n\tv_mov_b32 %1, v25\n\tv_mov_b32 %2, v28\n\tv_mov_b32 %3, v29\n\t"                   \
                 : "=v"(a), "=v"(b), "=v"(c), "=v"(d)                                                                     \
                 : "v"(alo), "v"(ahi), "v"(marker)                                                                        \
                 : "v20", "v21", "v24", "v25", "v28", "v29", "v30", "v31", "v32", "s20", "s21", "s22", "s23", "scc", "memory")

#define PROBE(NAME, BODY)                                                                                                 \
    __global__ void __launch_bounds__(256) NAME(const float* __restrict__ src, float* __restrict__ dst, int iters,        \
                                                const unsigned* __restrict__ perm) {                                      \
        const unsigned gid = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;                                    \
        const float marker = -7.0f;                                                                                       \
        (void)marker;                                                                                                     \
        for (int it = 0; it < iters; ++it) {                                                                              \
            const unsigned slot = it * stride + gid;                                                                      \
            const unsigned idx = perm != nullptr ? perm[slot] : slot;      /* (a gather when the host passes a permutation) */ \
            const unsigned voff = idx * 16u;                                                                              \
            const uint64_t addr = (uint64_t)src + voff;                                                                   \
            const unsigned alo = (unsigned)addr, ahi = (unsigned)(addr >> 32), blo = (unsigned)(uint64_t)src;             \
            (void)alo; (void)ahi; (void)blo;                                                                              \
            float a, b, c, d;                                                                                             \
            BODY                                                                                                          \
        }                                                                                                                 \
    }
#define LOAD_BODY(L) L; { float4 v = make_float4(a, b, c, d); *(float4*)(dst + 4ull * slot) = v; }
#define STORE_BODY(S) { const float4 v = *(const float4*)(src + 4ull * idx); a = v.x; b = v.y; c = v.z; d = v.w; } S;

#define ALL_NOPS(X) X(0, NOP0) X(1, NOP1) X(2, NOP2) X(4, NOP4) X(8, NOP8) X(16, NOP16) X(32, NOP32)
#define DEF_M0(N, S) PROBE(probe_m0_n##N, LOAD_BODY(LOAD_SADDR(S)))
#define DEF_M1(N, S) PROBE(probe_m1_n##N, LOAD_BODY(LOAD_VADDR(S)))
#define DEF_M2(N, S) PROBE(probe_m2_n##N, STORE_BODY(STORE_X4(S)))
#define DEF_M3(N, S) PROBE(probe_m3_n##N, STORE_BODY(STORE_X1(S)))
#define DEF_M4(N, S) PROBE(probe_m4_n##N, LOAD_BODY(RELOAD_X3(S)))
#define DEF_M5(N, S) PROBE(probe_m5_n##N, LOAD_BODY(RELOAD_X4(S)))
#define DEF_M6(N, S) PROBE(probe_m6_n##N, LOAD_BODY(SINGLE_X3(S)))
#define DEF_M7(N, S) PROBE(probe_m7_n##N, LOAD_BODY(SINGLE_X2_PK(S)))
#define DEF_M8(N, S) PROBE(probe_m8_n##N, LOAD_BODY(COND_X2_PK(S)))
ALL_NOPS(DEF_M4)
ALL_NOPS(DEF_M5)
ALL_NOPS(DEF_M6)
ALL_NOPS(DEF_M7)
ALL_NOPS(DEF_M8)
ALL_NOPS(DEF_M0)
ALL_NOPS(DEF_M1)
ALL_NOPS(DEF_M2)
ALL_NOPS(DEF_M3)

typedef void (*probe_fn)(const float*, float*, int, const unsigned*);
#define ROW(M) {probe_m##M##_n0, probe_m##M##_n1, probe_m##M##_n2, probe_m##M##_n4, probe_m##M##_n8, probe_m##M##_n16, probe_m##M##_n32}
static probe_fn table[9][7] = {ROW(0), ROW(1), ROW(2), ROW(3), ROW(4), ROW(5), ROW(6), ROW(7), ROW(8)};

// mode 0..8, nops index 0..6 (0,1,2,4,8,16,32 wait states); elements = blocks * 256 * iters float4s
extern "C" int vmem_src_hazard_probe(const float* src, float* dst, int mode, int nops_idx, int blocks, int iters, const unsigned* perm,
                                     void* stream) {
    if (mode < 0 || mode > 8 || nops_idx < 0 || nops_idx > 6) return -2;
    hipLaunchKernelGGL(table[mode][nops_idx], dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dst, iters, perm);
    return (int)hipGetLastError();
}
