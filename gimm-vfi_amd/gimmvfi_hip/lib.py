"""ctypes binding of the C ABI declared in include/gimmvfi_hip.h.

The product path loads exactly one library: the in-tree hipcc build
``gimm-vfi_amd/lib/libgimmvfi_hip.so`` (gfx950).  There is no CPU fallback: if
the library is missing or no MI355X is visible, ``get()`` raises.

``HipLib(path)`` binds *a* shared object exporting the ABI; the CPU test-suite
uses it to bind the host emulator build of the same kernels
(tests/hostsim) -- test infrastructure, never reachable from ``get()``.
"""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.abspath(os.path.join(_HERE, "..", ".."))
HEADER = os.path.join(REPO_ROOT, "include", "gimmvfi_hip.h")
LIB_PATH = os.path.abspath(os.path.join(_HERE, "..", "lib", "libgimmvfi_hip.so"))

F32, BF16, F16 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_PRELU, ACT_SIGMOID, ACT_TANH, ACT_SIN, ACT_GELU = range(8)
PAD_ZEROS, PAD_REFLECT = 0, 1
EPI_STD, EPI_GRU_ZR, EPI_GRU_Q = 0, 1, 2


class ConvParams(C.Structure):
    """Mirror of ``gvfi_conv_params`` (include/gimmvfi_hip.h)."""

    _fields_ = [
        ("dtype", C.c_int),
        ("x0", C.c_void_p), ("ld0", C.c_int), ("c0", C.c_int),
        ("x1", C.c_void_p), ("ld1", C.c_int), ("c1", C.c_int),
        ("N", C.c_int), ("H", C.c_int), ("W", C.c_int),
        ("w", C.c_void_p), ("w_group_stride", C.c_longlong), ("groups", C.c_int),
        ("bias", C.c_void_p),
        ("Cout", C.c_int), ("KH", C.c_int), ("KW", C.c_int), ("stride", C.c_int),
        ("pad_h", C.c_int), ("pad_w", C.c_int), ("pad_mode", C.c_int),
        ("Ho", C.c_int), ("Wo", C.c_int),
        ("epi_mode", C.c_int),
        ("act1", C.c_int), ("slope1", C.c_void_p),
        ("res", C.c_void_p), ("ldr", C.c_int), ("res_f32", C.c_int),
        ("act2", C.c_int), ("slope2", C.c_void_p),
        ("out_scale", C.c_float),
        ("y", C.c_void_p), ("ldy", C.c_int), ("y_f32", C.c_int),
        ("y2", C.c_void_p), ("ldy2", C.c_int),
        ("aux0", C.c_void_p), ("lda0", C.c_int),
        ("aux1", C.c_void_p), ("lda1", C.c_int),
        ("stats", C.c_void_p),
        ("tile_hint", C.c_int),
        ("w_layout", C.c_int),
        ("algo", C.c_int),
        ("state_f32", C.c_int),
    ]


class TokenChainParams(C.Structure):
    """Mirror of ``gvfi_token_chain_params`` (include/gimmvfi_hip.h)."""

    _fields_ = [
        ("in0", C.c_void_p), ("ld0", C.c_int),
        ("in1", C.c_void_p), ("ld1", C.c_int),
        ("k0a", C.c_int),
        ("wfrag", C.c_void_p),
        ("bias", C.c_void_p),
        ("ln_g", C.c_void_p), ("ln_b", C.c_void_p), ("eps", C.c_float), ("ln_after", C.c_int),
        ("coords", C.c_void_p), ("period", C.c_longlong),
        ("act0", C.c_int), ("act1", C.c_int),
        ("res0", C.c_void_p), ("ldr0", C.c_int),
        ("res2_from0", C.c_int),
        ("out1", C.c_void_p), ("ldo1", C.c_int),
        ("out2", C.c_void_p), ("ldo2", C.c_int),
        ("rows", C.c_longlong), ("dtype", C.c_int),
    ]


_CTYPES = {
    "int": C.c_int,
    "float": C.c_float,
    "long long": C.c_longlong,
}


def parse_header(path=HEADER):
    """Returns {name: (restype, [argtypes])} for every prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"typedef struct \{.*?\} gvfi_\w+_params;", " ", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"(const char\*|int)\s+(gvfi_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    argtypes.append(C.c_void_p)
                else:
                    ty = a.rsplit(" ", 1)[0]
                    argtypes.append(_CTYPES[ty])
        protos[name] = (C.c_char_p if "char" in ret else C.c_int, argtypes)
    return protos


class HipLib:
    def __init__(self, path):
        if not os.path.isfile(path):
            raise FileNotFoundError(
                f"{path} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()')"
            )
        self.path = path
        self.dll = C.CDLL(path)
        self.protos = parse_header()
        for name, (res, args) in self.protos.items():
            fn = getattr(self.dll, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
            setattr(self, name[len("gvfi_"):], fn)

    def check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed with code {rc}")


_LIB = None


def get() -> HipLib:
    """The product library.  Fails loudly without the gfx950 build or without a GPU."""
    global _LIB
    if _LIB is None:
        # (GVFI_LIB_PATH: another BUILD of the same library, for A/B measurements of compile-time variants)
        lib = HipLib(os.environ.get("GVFI_LIB_PATH", LIB_PATH))
        if lib.device_ok() != 1:
            raise RuntimeError("libgimmvfi_hip.so loaded but no usable gfx950 (MI355X) device is visible")
        _LIB = lib
    return _LIB
