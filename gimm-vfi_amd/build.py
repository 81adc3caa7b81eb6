"""Builds gimm-vfi_amd/lib/libgimmvfi_hip.so (gfx950) from csrc/*.hip with hipcc.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the tree
to the GPU box.  Objects are cached by source mtime."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "lib")
OBJ_DIR = os.path.join(HERE, "build")
LIB = os.path.join(OUT_DIR, "libgimmvfi_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off"]
# Packed fp32 VALU instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) are switched OFF for the whole library.  Round 6 found
# (tools/concurrency_repro.py, tools/victims2.py, profiles/r6_concurrency_repro.txt): a plain gather kernel whose compiler-generated
# code feeds freshly loaded registers to a packed fp32 instruction computes wrong values in lanes 48..63 of a wave -- a few hundred
# pixels per launch, only while waves of an LDS-DMA kernel (buffer_load ... lds) are resident on the same CU, i.e. only when the
# engine's parallel launch sequences put such kernels side by side.  Forcing s_waitcnt vmcnt(0) after every memory instruction does
# not help; the same source without packed fp32 instructions is exact (two independent kernels, 0 of 200 launches vs 170-190 of 200).
# The scalar forms have the same throughput on CDNA4's fp32 pipe for these bandwidth-bound kernels; the MFMA kernels lose nothing
# measurable (profiles/r6_nopk_ab.txt).  GVFI_PACKED_FP32=1 builds with the compiler's default, for reproducing the finding.
if os.environ.get("GVFI_PACKED_FP32", "0") != "1":
    FLAGS += ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


def _deps():
    return [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "conv_mma.h"), os.path.join(HERE, "..", "include", "gimmvfi_hip.h")]


def _compile(src):
    obj = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
    newest = max(os.path.getmtime(p) for p in [src] + _deps())
    if os.path.exists(obj) and os.path.getmtime(obj) > newest:
        return obj
    cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
    return obj


def build(verbose=True):
    os.makedirs(OUT_DIR, exist_ok=True)
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(_compile, srcs))
    # objects of sources that no longer exist (a deleted kernel file) are pruned: the library is linked from `objs` only, but a
    # stale object in the cache directory reads as product code that nobody can find the source of
    for f in os.listdir(OBJ_DIR):
        if f.endswith(".o") and os.path.join(OBJ_DIR, f) not in objs:
            os.remove(os.path.join(OBJ_DIR, f))
    if (not os.path.exists(LIB)) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build()
