"""Host side of tools/microbench/vmem_src_hazard.hip: runs each probe variant (VMEM instruction, N wait states, VALU write of one of
its source registers) alone and beside partner kernels on a second stream, and counts elements that show the VALU write.
usage: python tools/vmem_src_hazard.py"""
import ctypes as C
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "gimm-vfi_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from gimmvfi_hip import lib as L  # noqa: E402
from gimmvfi_hip.ops import ConvLayer, Runtime, View  # noqa: E402

DEV = "cuda:0"
so = C.CDLL(os.path.join(ROOT, "tools", "microbench", "libvmem_src_hazard.so"))
so.vmem_src_hazard_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
rt = Runtime(L.get(), "bf16", DEV)
g = torch.Generator().manual_seed(0)
BLOCKS, ITERS = 2048, 16
n = BLOCKS * 256 * ITERS
src = torch.rand(n + 1, 4, generator=g).to(DEV) + 1.0           # (all > 0, element 0 distinct from the rest; the marker is -7)
lay1 = ConvLayer(rt, torch.randn(256, 256, 1, 1, generator=g) / 16, torch.randn(256, generator=g))
lay64 = ConvLayer(rt, torch.randn(64, 64, 3, 3, generator=g) / 24, torch.randn(64, generator=g))
lay3 = ConvLayer(rt, torch.randn(256, 256, 3, 3, generator=g) / 48, torch.randn(256, generator=g))
px, py = torch.randn(2, 272, 512, 256, device=DEV).to(rt.tdtype), rt.act(2, 272, 512, 256)
qx, qy = torch.randn(2, 544, 1024, 64, device=DEV).to(rt.tdtype), rt.act(2, 544, 1024, 64)
big, big2 = torch.empty(64 << 20, device=DEV), torch.empty(64 << 20, device=DEV)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
GATHER = os.environ.get("GATHER", "0") != "0"
perm = torch.randperm(n, generator=g).to(torch.int32).to(DEV) if GATHER else None       # (element read by slot i)
pl = perm.long() if GATHER else torch.arange(n, device=DEV)


def p_dma():
    for i in range(40):
        if i & 1:
            rt.conv(lay1, View(px, 0, 256), py, algo=2, tile=128)
        else:
            rt.conv(lay64, View(qx, 0, 64), qy)


def p_glds():
    for i in range(40):
        rt.conv(lay1, View(px, 0, 256), py, algo=2, tile=128)


def p_mid():
    for i in range(20):
        rt.conv(lay64, View(qx, 0, 64), qy)


def p_plain():
    for i in range(40):
        rt.conv(lay1, View(px, 0, 256), py, algo=1)


def p_copy():
    for i in range(10):
        big2.copy_(big)


PARTNERS = [("none", None), ("1x1 LDS-DMA 128 tile + mid-channel 3x3", p_dma), ("1x1 LDS-DMA 128 tile", p_glds), ("mid-channel 3x3 (halo kernel)", p_mid),
            ("1x1 register-staged igemm", p_plain), ("256 MB copies", p_copy)]
MODES = ["load x4, 32-bit offset + SGPR base; offset register overwritten", "load x4, 64-bit VGPR address; low half overwritten",
         "store x4; data registers 0 and 3 overwritten", "store x1; data register overwritten",
         "x3 load of element i, wait, x3 load of element i+1 into the same registers, wait, N wait states, read",
         "x4 load of element i, wait, x4 load of element i+1 into the same registers, wait, N wait states, read",
         "x3 load of element i+1 into registers holding a marker, wait, N wait states, read",
         "x2 load of element i+1 into registers holding a marker, wait, N wait states, v_pk_mul_f32 reads the pair",
         "s_and_saveexec (all lanes on), x2 load, two VALU instructions, wait, N wait states, v_pk_mul_f32 reads the pair"]
ONLY = [int(v) for v in os.environ.get("MODES", "0,1,2,3,4,5,6,7,8").split(",")]
NOPS = [0, 1, 2, 4, 8, 16, 32]
REPS = int(os.environ.get("REPS", 6))
for mode, mname in enumerate(MODES):
    if mode not in ONLY:
        continue
    print(f"== {mname}")
    for pname, partner in PARTNERS:
        row = []
        quarters, sample = torch.zeros(4, dtype=torch.long), None
        for ni, nops in enumerate(NOPS):
            bad = 0
            for _ in range(REPS):
                dst = torch.zeros(n, 4, device=DEV)
                torch.cuda.synchronize()
                if partner is not None:
                    with torch.cuda.stream(sb):
                        partner()
                with torch.cuda.stream(sa):
                    rc = so.vmem_src_hazard_probe(src.data_ptr(), dst.data_ptr(), mode, ni, BLOCKS, ITERS, perm.data_ptr() if GATHER else None, torch.cuda.current_stream().cuda_stream)
                    assert rc == 0, rc
                torch.cuda.synchronize()
                if mode == 3:
                    wrong = dst[:, 0] != src[:n, 0]
                elif mode == 2:
                    wrong = (dst != src[:n]).any(1)
                elif mode >= 7:
                    wrong = (dst[:, :2] != src[pl + 1, :2]).any(1)
                elif mode >= 4:
                    wrong = (dst[:, :3] != src[pl + 1, :3]).any(1)
                else:
                    wrong = (dst != src[pl]).any(1)
                bad += int(wrong.sum())
                if mode >= 4 and bool(wrong.any()):
                    lanes = wrong.nonzero().flatten() % 64
                    quarters += torch.bincount((lanes // 16).cpu(), minlength=4)
                    w = int(wrong.nonzero()[0])
                    sample = (dst[w].tolist(), src[w].tolist(), src[w + 1].tolist())
            row.append(bad)
        print(f"   beside {pname:42s}: wrong elements (of {REPS * n}) at {NOPS} wait states: {row}"
              + (f"; by 16-lane quarter {quarters.tolist()}; e.g. got {sample[0]}, element i {sample[1]}, element i+1 {sample[2]}" if sample else ""))
