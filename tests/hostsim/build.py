"""TEST INFRASTRUCTURE ONLY: compiles gimm-vfi_amd/csrc/*.hip for the HOST with the
lane-level emulator header (hip_emu.h: lanes as fibers) so CPU tests can exercise kernel index math.
The result (tests/hostsim/_build/libgimmvfi_hostsim.so) is never loaded by the product."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
CSRC = os.path.join(ROOT, "gimm-vfi_amd", "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libgimmvfi_hostsim.so")
CXX = os.environ.get("HOSTSIM_CXX", "/opt/rocm/lib/llvm/bin/clang++")


def _fma_flag():
    """std::fmaf (the emulated fp32 MFMA) as one instruction where the host has it -- exactly rounded either way."""
    try:
        with open("/proc/cpuinfo") as f:
            return ["-mfma"] if " fma " in f.read().replace("\n", " ") else []
    except OSError:
        return []


def _compile(src):
    obj = os.path.join(OUT, os.path.basename(src) + ".o")
    deps = [src, os.path.join(CSRC, "common.h"), os.path.join(CSRC, "conv_mma.h"), os.path.join(HERE, "hip_emu.h"),
            os.path.join(ROOT, "include", "gimmvfi_hip.h")]
    if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(d) for d in deps):
        return obj
    cmd = [CXX, "-std=c++20", "-O2", "-DGVFI_HOSTSIM", "-ffp-contract=off", "-I", HERE, "-x", "c++", "-fPIC", "-pthread"] + _fma_flag() + [
           "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hostsim compile failed:\n" + r.stdout + r.stderr)
    return obj


def build():
    os.makedirs(OUT, exist_ok=True)
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    srcs += sorted(os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".hip"))     # (the emulator's own self-test kernels)
    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(_compile, srcs))
    # objects of sources that no longer exist (a deleted kernel file) are pruned: the library is linked from `objs` only, but a
    # stale object in the cache directory reads as product code that nobody can find the source of
    for f in os.listdir(OUT):
        if f.endswith(".o") and os.path.join(OUT, f) not in objs:
            os.remove(os.path.join(OUT, f))
    if (not os.path.exists(LIB)) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        r = subprocess.run([CXX, "-shared", "-pthread", "-o", LIB] + objs, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hostsim link failed:\n" + r.stdout + r.stderr)
    return LIB


if __name__ == "__main__":
    print(build())
