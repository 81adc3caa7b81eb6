"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Imports the *real* reference (GSeanCDAT/GIMM-VFI, read-only at /root/reference)
on CPU so that (a) oracle/gimmvfi_r_oracle.py can be pinned against it and
(b) tests/golden/ fixtures can be generated (oracle/make_golden.py).

/root/reference does not exist on the GPU box: everything that uses this
module must be skipped there (``reference_available()``).

Recipe:
  1. fake ``cupy`` and ``omegaconf`` modules (both absent in this image);
  2. register ``models.generalizable_INR`` as a namespace package without
     running its ``__init__`` (which would pull gimmvfi_f -> timm/yacs/loguru);
  3. replace ``raft.initialize_RAFT`` (it hard-loads
     pretrained_ckpt/raft-things.pth, reference raft/__init__.py:7-24);
  4. replace the CUDA-only ``softsplat_func`` (reference modules/softsplat.py:358-446
     asserts on CPU tensors) by a CPU scatter-add statement of the same kernel.
"""
import importlib
import os
import sys
import types

import torch

REF_ROOT = os.environ.get("GIMMVFI_REFERENCE", "/root/reference")
_PKG = "refsrc_models"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "src", "models", "generalizable_INR"))


class AttrDict(dict):
    """Minimal OmegaConf/easydict stand-in: attribute access + .copy()."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return v

    def __setattr__(self, k, v):
        self[k] = v

    def copy(self):
        return AttrDict({k: (v.copy() if isinstance(v, AttrDict) else v) for k, v in self.items()})


def default_arch_config() -> AttrDict:
    """configs/gimmvfi/gimmvfi_r_arb.yaml:7-27 merged with
    generalizable_INR/configs.py:38-57 defaults."""
    return AttrDict(
        type="gimmvfi_r",
        ema=None,
        ema_value=None,
        fwarp_type="linear",
        rec_weight=0.1,
        raft_iter=20,
        coord_range=[-1.0, 1.0],
        modulated_layer_idxs=[1],
        hyponet=AttrDict(
            type="mlp",
            n_layer=5,
            hidden_dim=[128],
            use_bias=True,
            input_dim=3,
            output_dim=2,
            output_bias=0.5,
            activation=AttrDict(type="siren", siren_w0=1.0),
            initialization=AttrDict(weight_init_type="siren", bias_init_type="siren"),
            normalize_weight=True,
            linear_interpo=False,
        ),
    )


def _cpu_softsplat_out(tenIn: torch.Tensor, tenFlow: torch.Tensor) -> torch.Tensor:
    """CPU statement of kernel softsplat_out (reference modules/softsplat.py:371-421):
    every source element adds in*w to its 4 bilinear neighbours of (x+fx, y+fy);
    taps outside the image are dropped; non-finite targets are skipped."""
    N, C, H, W = tenIn.shape
    dev = tenIn.device
    gy, gx = torch.meshgrid(
        torch.arange(H, device=dev, dtype=torch.float32),
        torch.arange(W, device=dev, dtype=torch.float32),
        indexing="ij",
    )
    fx = gx[None] + tenFlow[:, 0]
    fy = gy[None] + tenFlow[:, 1]
    finite = torch.isfinite(fx) & torch.isfinite(fy)
    fx = torch.where(finite, fx, torch.zeros_like(fx))
    fy = torch.where(finite, fy, torch.zeros_like(fy))
    x0 = torch.floor(fx)
    y0 = torch.floor(fy)
    x1 = x0 + 1
    y1 = y0 + 1
    out = tenIn.new_zeros(N, C, H * W)
    src = tenIn.reshape(N, C, H * W)
    taps = (
        (x0, y0, (x1 - fx) * (y1 - fy)),
        (x1, y0, (fx - x0) * (y1 - fy)),
        (x0, y1, (x1 - fx) * (fy - y0)),
        (x1, y1, (fx - x0) * (fy - y0)),
    )
    for tx, ty, w in taps:
        ok = finite & (tx >= 0) & (tx < W) & (ty >= 0) & (ty < H)
        idx = (ty.clamp(0, H - 1) * W + tx.clamp(0, W - 1)).long().reshape(N, 1, H * W)
        wv = (w * ok).reshape(N, 1, H * W)
        out.scatter_add_(2, idx.expand(N, C, H * W), src * wv)
    return out.reshape(N, C, H, W)


class _SplatFn:
    @staticmethod
    def apply(tenIn, tenFlow):
        return _cpu_softsplat_out(tenIn.float(), tenFlow.float())


def _install_shims():
    if "cupy" not in sys.modules:
        cupy = types.ModuleType("cupy")
        cupy.int32 = int
        cupy.float32 = float
        cupy.ndarray = type("ndarray", (), {})

        def memoize(for_each_device=False):
            def deco(fn):
                return fn

            return deco

        cupy.memoize = memoize
        sys.modules["cupy"] = cupy
    if "omegaconf" not in sys.modules:
        oc = types.ModuleType("omegaconf")
        oc.MISSING = "???"

        class OmegaConf:
            @staticmethod
            def to_object(x):
                return list(x)

            @staticmethod
            def structured(x):
                return x

            @staticmethod
            def merge(*a):
                return a[-1]

        oc.OmegaConf = OmegaConf
        sys.modules["omegaconf"] = oc


_CACHE = {}


def load_reference_modules():
    """Returns the reference's gimmvfi_r module (python module object)."""
    if "mod" in _CACHE:
        return _CACHE["mod"]
    assert reference_available(), "reference checkout not present"
    _install_shims()
    base = os.path.join(REF_ROOT, "src", "models")
    for name, path in (
        (_PKG, base),
        (_PKG + ".generalizable_INR", os.path.join(base, "generalizable_INR")),
    ):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    raft_pkg = importlib.import_module(_PKG + ".generalizable_INR.raft")
    raft_mod = importlib.import_module(_PKG + ".generalizable_INR.raft.raft")

    def initialize_RAFT(model_path=None, device="cpu"):
        import argparse

        args = argparse.ArgumentParser()
        args.small = False
        args.mixed_precision = False
        args.alternate_corr = False
        return raft_mod.RAFT(args)

    raft_pkg.initialize_RAFT = initialize_RAFT
    splat = importlib.import_module(_PKG + ".generalizable_INR.modules.softsplat")
    splat.softsplat_func = _SplatFn
    mod = importlib.import_module(_PKG + ".generalizable_INR.gimmvfi_r")
    mod.initialize_RAFT = initialize_RAFT
    # fi_utils caches a module-level device; force CPU
    fiu = importlib.import_module(_PKG + ".generalizable_INR.modules.fi_utils")
    fiu.device = torch.device("cpu")
    _CACHE["mod"] = mod
    return mod


def build_reference_model(state_dict=None):
    mod = load_reference_modules()
    torch.manual_seed(0)
    model = mod.GIMMVFI_R(default_arch_config())
    if state_dict is not None:
        model.load_state_dict(state_dict, strict=True)
    model.eval()
    return model


def reference_forward(model, img_xs, t_list, ds_factor=None):
    """Mirrors the call sequence of src/video_Nx.py:164-181."""
    B = img_xs.shape[0]
    s_shape = img_xs.shape[-2:]
    ratio = 1.0 if ds_factor is None else ds_factor
    with torch.no_grad():
        coords = [
            (model.sample_coord_input(B, s_shape, [float(t)], device=img_xs.device, upsample_ratio=ratio), None)
            for t in t_list
        ]
        ts = [float(t) * torch.ones(B, dtype=torch.float32) for t in t_list]
        return model(img_xs, coords, t=ts, ds_factor=ds_factor)


if __name__ == "__main__":
    import time

    m = build_reference_model()
    sd = m.state_dict()
    print("keys", len(sd), "params", sum(v.numel() for v in sd.values()))
    x = torch.rand(1, 3, 2, 128, 192)
    t0 = time.time()
    out = reference_forward(m, x, [0.5])
    print("fwd s", time.time() - t0, out["imgt_pred"][0].shape, float(out["imgt_pred"][0].mean()))


# ----------------------------------------------------------------------------------------------
# GIMM-VFI-F (FlowFormer flow estimator).  Extra absent dependencies:
#   timm==0.4.12 (requirements.txt:74), yacs==0.1.6, loguru==0.7.2, and `turtle` (tkinter) which
#   LatentCostFormer/convnext.py:1 imports by accident.
# timm is the one that carries arithmetic: `timm.create_model("twins_svt_large")`
# (flowformer/core/FlowFormer/encoders.py:10).  The reference vendors an adapted copy of timm's
# twins.py (LatentCostFormer/twins.py:814-1290, model kwargs in the comment at :1344-1348); the shim
# below answers create_model with THAT vendored class, and `timm.models.layers.Mlp` with timm 0.4.12's
# published definition (fc1 -> act -> drop -> fc2 -> drop).  Everything else of FlowFormer is the
# reference's own code, run unmodified.  Parity at the timm boundary is therefore pinned against the
# reference's vendored Twins, not against the timm wheel (absent here): "parity unpinned" for timm
# itself, stated in DESIGN.md.
# ----------------------------------------------------------------------------------------------
def _install_shims_f():
    _install_shims()
    import torch.nn as nn

    if "loguru" not in sys.modules:
        lg = types.ModuleType("loguru")

        class _Logger:
            def __getattr__(self, k):
                return lambda *a, **kw: None

        lg.logger = _Logger()
        sys.modules["loguru"] = lg
    if "turtle" not in sys.modules:
        tt = types.ModuleType("turtle")
        tt.forward = lambda *a, **kw: None
        sys.modules["turtle"] = tt
    if "yacs" not in sys.modules:
        yacs = types.ModuleType("yacs")
        ycfg = types.ModuleType("yacs.config")

        class CfgNode(AttrDict):
            def clone(self):
                return CfgNode({k: (v.clone() if isinstance(v, CfgNode) else v) for k, v in self.items()})

        ycfg.CfgNode = CfgNode
        yacs.config = ycfg
        sys.modules["yacs"] = yacs
        sys.modules["yacs.config"] = ycfg
    if "timm" not in sys.modules:
        timm = types.ModuleType("timm")
        data = types.ModuleType("timm.data")
        data.IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
        data.IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)
        models = types.ModuleType("timm.models")
        layers = types.ModuleType("timm.models.layers")
        registry = types.ModuleType("timm.models.registry")
        vit = types.ModuleType("timm.models.vision_transformer")
        helpers = types.ModuleType("timm.models.helpers")

        class Mlp(nn.Module):  # timm 0.4.12 timm/models/layers/mlp.py
            def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
                super().__init__()
                out_features = out_features or in_features
                hidden_features = hidden_features or in_features
                self.fc1 = nn.Linear(in_features, hidden_features)
                self.act = act_layer()
                self.fc2 = nn.Linear(hidden_features, out_features)
                self.drop = nn.Dropout(drop)

            def forward(self, x):
                return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))

        class DropPath(nn.Module):
            def __init__(self, drop_prob=None):
                super().__init__()
                self.drop_prob = drop_prob

            def forward(self, x):
                assert not self.training
                return x

        layers.Mlp = Mlp
        layers.DropPath = DropPath
        layers.to_2tuple = lambda x: tuple(x) if isinstance(x, (tuple, list)) else (x, x)
        layers.trunc_normal_ = torch.nn.init.trunc_normal_
        layers.activations = types.ModuleType("timm.models.layers.activations")
        registry.register_model = lambda fn: fn
        vit.Attention = type("Attention", (nn.Module,), {})
        helpers.build_model_with_cfg = None
        helpers.overlay_external_default_cfg = None

        def create_model(name, pretrained=False, **kw):
            assert name == "twins_svt_large", name
            tw = importlib.import_module(
                _PKG + ".generalizable_INR.flowformer.core.FlowFormer.LatentCostFormer.twins"
            )
            class TimmBlock(tw.Block):
                # timm 0.4.12 twins.py Block.forward(x, size): the vendored Block (twins.py:1094-1097) added a
                # `context` argument that it forwards to every attention class; timm's own does not.
                def forward(self, x, size):
                    x = x + self.drop_path(self.attn(self.norm1(x), size))
                    x = x + self.drop_path(self.mlp(self.norm2(x)))
                    return x

            # kwargs of twins_svt_large: reference twins.py:1344-1348 (comment) == timm 0.4.12 twins.py
            return tw.Twins(
                block_cls=TimmBlock,
                patch_size=4,
                embed_dims=[128, 256, 512, 1024],
                num_heads=[4, 8, 16, 32],
                mlp_ratios=[4, 4, 4, 4],
                depths=[2, 2, 18, 2],
                wss=[7, 7, 7, 7],
                sr_ratios=[8, 4, 2, 1],
            )

        timm.create_model = create_model
        timm.data = data
        timm.models = models
        models.layers = layers
        models.registry = registry
        models.vision_transformer = vit
        models.helpers = helpers
        for n, m in (
            ("timm", timm),
            ("timm.data", data),
            ("timm.models", models),
            ("timm.models.layers", layers),
            ("timm.models.registry", registry),
            ("timm.models.vision_transformer", vit),
            ("timm.models.helpers", helpers),
        ):
            sys.modules[n] = m


def default_arch_config_f() -> AttrDict:
    """configs/gimmvfi/gimmvfi_f_arb.yaml:7-27 merged with configs.py:38-57 defaults."""
    c = default_arch_config()
    c.type = "gimmvfi_f"
    return c


def load_reference_modules_f():
    """Returns the reference's gimmvfi_f module."""
    if "mod_f" in _CACHE:
        return _CACHE["mod_f"]
    load_reference_modules()
    _install_shims_f()
    ffpkg = importlib.import_module(_PKG + ".generalizable_INR.flowformer")

    def initialize_Flowformer():
        # reference flowformer/__init__.py:6-18 without the torch.load of pretrained_ckpt/flowformer_sintel.pth
        cfg = ffpkg.get_cfg()
        return ffpkg.build_flowformer(cfg)

    ffpkg.initialize_Flowformer = initialize_Flowformer
    mod = importlib.import_module(_PKG + ".generalizable_INR.gimmvfi_f")
    mod.initialize_Flowformer = initialize_Flowformer
    _CACHE["mod_f"] = mod
    return mod


def build_reference_model_f(state_dict=None):
    mod = load_reference_modules_f()
    torch.manual_seed(0)
    model = mod.GIMMVFI_F(default_arch_config_f())
    if state_dict is not None:
        model.load_state_dict(state_dict, strict=True)
    model.eval()
    return model
