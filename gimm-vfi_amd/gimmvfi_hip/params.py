"""Parameter registry of GIMM-VFI-R: state_dict key names and shapes.

The key set is part of the drop-in contract: reference checkpoints are loaded
with ``model.load_state_dict(ckpt["state_dict"], strict=True)``
(reference src/video_Nx.py:114-115), so every name below matches the
reference module tree (gimmvfi_r.py:37-124, raft/raft.py:54-70,
raft/extractor.py:122-166, raft/update.py:94-148, modules/fi_components.py).
414 entries / 19,789,980 elements.

``random_state_dict(seed)`` builds seeded random weights of this architecture
for benchmarks and parity tests (no pretrained checkpoints exist offline).
Norm statistics, PReLU slopes and the splat alphas are deliberately
non-trivial so that BN folding / per-channel PReLU / alpha handling are tested.
"""
from collections import OrderedDict
import math

import torch


def _conv(d, name, cout, cin, kh, kw=None):
    kw = kh if kw is None else kw
    d[name + ".weight"] = (cout, cin, kh, kw)
    d[name + ".bias"] = (cout,)


def _bn(d, name, c):
    d[name + ".weight"] = (c,)
    d[name + ".bias"] = (c,)
    d[name + ".running_mean"] = (c,)
    d[name + ".running_var"] = (c,)
    d[name + ".num_batches_tracked"] = ()


def _encoder(d, p, batchnorm, out_dim):
    # raft/extractor.py:122-166
    if batchnorm:
        _bn(d, p + ".norm1", 64)
    _conv(d, p + ".conv1", 64, 3, 7)
    cin = 64
    for li, dim in ((1, 64), (2, 96), (3, 128)):
        for bi in (0, 1):
            q = f"{p}.layer{li}.{bi}"
            stride2 = bi == 0 and li > 1
            _conv(d, q + ".conv1", dim, cin if bi == 0 else dim, 3)
            _conv(d, q + ".conv2", dim, dim, 3)
            if batchnorm:
                _bn(d, q + ".norm1", dim)
                _bn(d, q + ".norm2", dim)
                if stride2:
                    _bn(d, q + ".norm3", dim)
            if stride2:
                _conv(d, q + ".downsample.0", dim, cin, 1)
                if batchnorm:
                    _bn(d, q + ".downsample.1", dim)
        cin = dim
    _conv(d, p + ".conv2", out_dim, 128, 1)


def _convrelu(d, p, cout, cin, k):
    _conv(d, p + ".0", cout, cin, k)
    d[p + ".1.weight"] = (cout,)


def _resblock(d, p, c, side):
    _convrelu(d, p + ".conv1", c, c, 3)
    _convrelu(d, p + ".conv2", side, side, 3)
    _convrelu(d, p + ".conv3", c, c, 3)
    _convrelu(d, p + ".conv4", side, side, 3)
    _conv(d, p + ".conv5", c, c, 3)
    d[p + ".prelu.weight"] = (c,)


def _amt_update(d, p):
    # gimmvfi_r.py:113-124 -> modules/fi_components.py:157-197
    _conv(d, p + ".convc1", 256, 648, 1)
    _conv(d, p + ".convc2", 192, 256, 3)
    _conv(d, p + ".convf1", 128, 4, 7)
    _conv(d, p + ".convf2", 64, 128, 3)
    _conv(d, p + ".conv", 188, 256, 3)
    _conv(d, p + ".gru.0", 192, 320, 3)
    _conv(d, p + ".gru.2", 192, 192, 3)
    _conv(d, p + ".feat_head.0", 192, 192, 3)
    _conv(d, p + ".feat_head.2", 128, 192, 3)
    _conv(d, p + ".flow_head.0", 192, 192, 3)
    _conv(d, p + ".flow_head.2", 4, 192, 3)


def param_spec() -> "OrderedDict[str, tuple]":
    d = OrderedDict()
    d["g_filter"] = (1, 1, 1, 3, 3)
    d["alpha_v"] = (1,)
    d["alpha_fe"] = (1,)
    fe = "flow_estimator"
    _encoder(d, fe + ".fnet", False, 256)
    _encoder(d, fe + ".cnet", True, 256)
    u = fe + ".update_block"
    _conv(d, u + ".encoder.convc1", 256, 324, 1)
    _conv(d, u + ".encoder.convc2", 192, 256, 3)
    _conv(d, u + ".encoder.convf1", 128, 2, 7)
    _conv(d, u + ".encoder.convf2", 64, 128, 3)
    _conv(d, u + ".encoder.conv", 126, 256, 3)
    for n, (kh, kw) in (("1", (1, 5)), ("2", (5, 1))):
        for g in "zrq":
            _conv(d, f"{u}.gru.conv{g}{n}", 128, 384, kh, kw)
    _conv(d, u + ".flow_head.conv1", 256, 128, 3)
    _conv(d, u + ".flow_head.conv2", 2, 256, 3)
    _conv(d, u + ".mask.0", 256, 128, 3)
    _conv(d, u + ".mask.2", 576, 256, 1)
    _conv(d, "amt_last_cproj", 256, 128, 1)
    _conv(d, "amt_second_last_cproj", 128, 96, 1)
    _conv(d, "amt_fproj", 256, 256, 1)
    p = "amt_init_decoder"
    _convrelu(d, p + ".upsample.1", 64, 64, 5)
    for i in (2, 3, 4):
        _convrelu(d, f"{p}.upsample.{i}", 64, 64, 3)
    _convrelu(d, p + ".upsample.5", 128, 64, 3)
    _conv(d, p + ".upsample.6", 128, 128, 1)
    _bn(d, p + ".upsample.7", 128)
    _convrelu(d, p + ".convblock.0", 128, 272, 1)
    for i in (1, 2, 3):
        _resblock(d, f"{p}.convblock.{i}", 128, 64)
    _conv(d, p + ".convblock.4", 133, 128, 3)
    p = "amt_final_decoder"
    _convrelu(d, p + ".upsample.2", 32, 8, 5)
    for i in (3, 4, 5):
        _convrelu(d, f"{p}.upsample.{i}", 32, 32, 3)
    _convrelu(d, p + ".upsample.6", 64, 32, 3)
    _conv(d, p + ".upsample.7", 64, 64, 1)
    _bn(d, p + ".upsample.8", 64)
    _convrelu(d, p + ".convblock.0", 256, 273, 3)
    for i in (1, 2, 3):
        _resblock(d, f"{p}.convblock.{i}", 256, 64)
    _conv(d, p + ".convblock.4", 24, 256, 3)
    _amt_update(d, "amt_update4_low")
    _amt_update(d, "amt_update4_high")
    _conv(d, "amt_comb_block.0", 18, 9, 7)
    d["amt_comb_block.1.weight"] = (18,)
    _conv(d, "amt_comb_block.2", 3, 18, 7)
    _conv(d, "cnn_encoder.0", 16, 2, 3)
    _conv(d, "cnn_encoder.1", 32, 16, 3)
    for i in (3, 4, 5):
        _conv(d, f"cnn_encoder.{i}.layers.0", 32, 32, 3)
        _conv(d, f"cnn_encoder.{i}.layers.2", 32, 32, 3)
    _conv(d, "cnn_encoder.7", 16, 32, 3)
    _conv(d, "res_conv.0", 32, 64, 3)
    _conv(d, "res_conv.1", 64, 32, 3)
    _conv(d, "res_conv.3.layers.0", 64, 64, 3)
    _conv(d, "res_conv.3.layers.2", 64, 64, 3)
    _conv(d, "res_conv.5", 32, 64, 3)
    d["hyponet.params_dict.linear_wb0"] = (36, 128)
    for i in (1, 2, 3):
        d[f"hyponet.params_dict.linear_wb{i}"] = (129, 128)
    d["hyponet.params_dict.linear_wb4"] = (129, 2)
    return d


def random_state_dict(seed: int = 0, gain: float = 1.0) -> "OrderedDict[str, torch.Tensor]":
    """Seeded random weights (CPU generator => identical on every machine).

    conv weights/biases ~ U(+-gain/sqrt(fan_in)) (torch's default Conv2d init
    range), SIREN-style INR weights (modules/utils.py:36-44)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)

    def U(shape, lo, hi):
        return torch.rand(shape, generator=g, dtype=torch.float32) * (hi - lo) + lo

    sd = OrderedDict()
    spec = param_spec()
    for name, shape in spec.items():
        if name == "g_filter":
            sd[name] = torch.tensor(
                [[1 / 16, 1 / 8, 1 / 16], [1 / 8, 1 / 4, 1 / 8], [1 / 16, 1 / 8, 1 / 16]], dtype=torch.float32
            ).reshape(1, 1, 1, 3, 3)
        elif name == "alpha_v":
            sd[name] = torch.tensor([0.8], dtype=torch.float32)
        elif name == "alpha_fe":
            sd[name] = torch.tensor([1.3], dtype=torch.float32)
        elif name.endswith("num_batches_tracked"):
            sd[name] = torch.tensor(0, dtype=torch.long)
        elif name.endswith("running_mean"):
            sd[name] = U(shape, -0.1, 0.1)
        elif name.endswith("running_var"):
            sd[name] = U(shape, 0.6, 1.4)
        elif name.startswith("hyponet"):
            fan_in = shape[0] - 1
            first = name.endswith("wb0")
            std = (1.0 / fan_in) if first else math.sqrt(6.0 / fan_in)
            sd[name] = U(shape, -std, std)
        elif len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            b = gain / math.sqrt(fan_in)
            sd[name] = U(shape, -b, b)
        elif name.endswith(".bias"):
            wname = name[: -len("bias")] + "weight"
            ws = spec.get(wname)
            if ws is not None and len(ws) == 4:
                b = gain / math.sqrt(ws[1] * ws[2] * ws[3])
                sd[name] = U(shape, -b, b)
            else:  # BatchNorm beta
                sd[name] = U(shape, -0.1, 0.1)
        elif name.endswith(".weight"):
            # 1-D weight: BatchNorm gamma or PReLU slope
            is_bn = (name[: -len("weight")] + "running_mean") in spec
            sd[name] = U(shape, 0.8, 1.2) if is_bn else U(shape, 0.1, 0.4)
        else:
            raise KeyError(name)
    return sd


# ---- the motion-only model GIMM (reference generalizable_INR/gimm.py:26-78): the same blocks under the same names
GIMM_KEY_PREFIXES = ("cnn_encoder.", "res_conv.", "hyponet.", "g_filter", "alpha_v", "alpha_fe")


def gimm_param_spec():
    return OrderedDict((k, v) for k, v in param_spec().items() if k.startswith(GIMM_KEY_PREFIXES))


def gimm_state_dict(sd_full):
    """The GIMM subset of a GIMM-VFI-R state_dict."""
    return OrderedDict((k, v) for k, v in sd_full.items() if k.startswith(GIMM_KEY_PREFIXES))


# ---- GIMM-VFI-F (reference generalizable_INR/gimmvfi_f.py:30-112): FlowFormer flow estimator instead of RAFT, no
# amt_*proj 1x1 projections; everything else under the same names.  639 entries / 30,611,132 elements.
def _lin(d, name, cout, cin, bias=True):
    d[name + ".weight"] = (cout, cin)
    if bias:
        d[name + ".bias"] = (cout,)


def _norm(d, name, c):
    d[name + ".weight"] = (c,)
    d[name + ".bias"] = (c,)


def _twins(d, p):
    """timm twins_svt_large cut to two stages (reference flowformer/core/FlowFormer/encoders.py:7-19; classes
    LatentCostFormer/twins.py:814-983,1028-1150).  svt.norm (1024) survives the surgery and is in the checkpoint."""
    dims, srs = (128, 256), (8, 4)
    cin = 3
    for i, (c, patch) in enumerate(zip(dims, (4, 2))):
        _conv(d, f"{p}.svt.patch_embeds.{i}.proj", c, cin, patch)
        _norm(d, f"{p}.svt.patch_embeds.{i}.norm", c)
        cin = c
    for i, (c, sr) in enumerate(zip(dims, srs)):
        for j in (0, 1):
            b = f"{p}.svt.blocks.{i}.{j}"
            _norm(d, b + ".norm1", c)
            if j == 0:
                _lin(d, b + ".attn.qkv", 3 * c, c)
                _lin(d, b + ".attn.proj", c, c)
            else:
                _lin(d, b + ".attn.q", c, c)
                _lin(d, b + ".attn.kv", 2 * c, c)
                _lin(d, b + ".attn.proj", c, c)
                _conv(d, b + ".attn.sr", c, c, sr)
                _norm(d, b + ".attn.norm", c)
            _norm(d, b + ".norm2", c)
            _lin(d, b + ".mlp.fc1", 4 * c, c)
            _lin(d, b + ".mlp.fc2", c, 4 * c)
    for i, c in enumerate(dims):
        _conv(d, f"{p}.svt.pos_block.{i}.proj.0", c, 1, 3)  # depthwise
    _norm(d, p + ".svt.norm", 1024)


def _attn_layer(d, p, qdim, tdim, qk, v, proj_in):
    # encoder.py:282-346 / 214-279, decoder.py:35-120: norm1, norm2, q, k, v, proj, ffn.{0,3}
    _norm(d, p + ".norm1", qdim)
    _norm(d, p + ".norm2", qdim)
    _lin(d, p + ".q", qk, qdim)
    _lin(d, p + ".k", qk, tdim)
    _lin(d, p + ".v", v, tdim)
    _lin(d, p + ".proj", qdim, proj_in)
    _lin(d, p + ".ffn.0", qdim, qdim)
    _lin(d, p + ".ffn.3", qdim, qdim)


def _vertical_block(d, p, local):
    # Block(with_rpe, vert_c_dim=64) -> LocallyGroupedAttnRPEContext / GlobalSubSampleAttnRPEContext (twins.py:331-546)
    _norm(d, p + ".norm1", 128)
    _lin(d, p + ".attn.context_proj", 64, 256)
    _lin(d, p + ".attn.q", 128, 192)
    _lin(d, p + ".attn.k", 128, 192 if local else 128)
    _lin(d, p + ".attn.v", 128, 128)
    _lin(d, p + ".attn.proj", 128, 128)
    if not local:
        _conv(d, p + ".attn.sr_key", 128, 192, 4)
        _conv(d, p + ".attn.sr_value", 128, 128, 4)
        _norm(d, p + ".attn.norm", 128)
    _norm(d, p + ".norm2", 128)
    _lin(d, p + ".mlp.fc1", 512, 128)
    _lin(d, p + ".mlp.fc2", 128, 512)


def _flowformer(d, fe):
    me = fe + ".memory_encoder"
    _twins(d, me + ".feat_encoder")
    d[me + ".channel_convertor.weight"] = (256, 256, 1, 1)
    ce = me + ".cost_perceiver_encoder"
    d[ce + ".latent_tokens"] = (1, 8, 128)
    _conv(d, ce + ".patch_embed.proj.0", 16, 1, 6)
    _conv(d, ce + ".patch_embed.proj.2", 32, 16, 6)
    _conv(d, ce + ".patch_embed.proj.4", 64, 32, 6)
    _conv(d, ce + ".patch_embed.ffn_with_coord.0", 128, 128, 1)
    _conv(d, ce + ".patch_embed.ffn_with_coord.2", 128, 128, 1)
    _norm(d, ce + ".patch_embed.norm", 128)
    _attn_layer(d, ce + ".input_layer", 128, 128, 128, 128, 128)
    for i in range(3):
        _attn_layer(d, f"{ce}.encoder_layers.{i}", 128, 128, 128, 128, 128)
    for i in range(3):
        _vertical_block(d, f"{ce}.vertical_encoder_layers.{i}.local_block", True)
        _vertical_block(d, f"{ce}.vertical_encoder_layers.{i}.global_block", False)
    md = fe + ".memory_decoder"
    _conv(d, md + ".flow_token_encoder.0", 64, 81, 1)
    _conv(d, md + ".flow_token_encoder.2", 64, 64, 1)
    _conv(d, md + ".proj", 256, 256, 1)
    _attn_layer(d, md + ".decoder_layer.cross_attend", 64, 128, 64, 64, 128)
    u = md + ".update_block"
    _conv(d, u + ".encoder.convc1", 256, 145, 1)
    _conv(d, u + ".encoder.convc2", 192, 256, 3)
    _conv(d, u + ".encoder.convf1", 128, 2, 7)
    _conv(d, u + ".encoder.convf2", 64, 128, 3)
    _conv(d, u + ".encoder.conv", 126, 256, 3)
    for n, (kh, kw) in (("1", (1, 5)), ("2", (5, 1))):
        for g in "zrq":
            _conv(d, f"{u}.gru.conv{g}{n}", 128, 512, kh, kw)
    _conv(d, u + ".flow_head.conv1", 256, 128, 3)
    _conv(d, u + ".flow_head.conv2", 2, 256, 3)
    _conv(d, u + ".mask.0", 256, 128, 3)
    _conv(d, u + ".mask.2", 576, 256, 1)
    d[u + ".aggregator.gamma"] = (1,)
    d[u + ".aggregator.to_v.weight"] = (128, 128, 1, 1)
    d[md + ".att.to_qk.weight"] = (256, 128, 1, 1)
    d[md + ".att.pos_emb.rel_ind"] = (160, 160)
    d[md + ".att.pos_emb.rel_height.weight"] = (319, 128)
    d[md + ".att.pos_emb.rel_width.weight"] = (319, 128)
    _twins(d, fe + ".context_encoder")


def param_spec_f() -> "OrderedDict[str, tuple]":
    d = OrderedDict()
    done = False
    for k, v in param_spec().items():
        if k.startswith("flow_estimator.") or k.startswith(("amt_last_cproj", "amt_second_last_cproj", "amt_fproj")):
            if not done:
                _flowformer(d, "flow_estimator")
                done = True
            continue
        d[k] = v
    return d


def random_state_dict_f(seed: int = 0, gain: float = 1.0, flow_head_scale: float = 1.0) -> "OrderedDict[str, torch.Tensor]":
    """Seeded random GIMM-VFI-F weights: the shared blocks exactly as random_state_dict(seed) gives them, the
    FlowFormer from an independent generator (Linear/conv ~ U(+-gain/sqrt(fan_in)), LayerNorm gamma/beta and the
    GMA gamma non-trivial so that every fused/folded path is exercised).  flow_head_scale multiplies the decoder's
    flow head (update_block.flow_head.conv2): the un-trained recurrence then takes 32 small steps instead of 32 large
    ones -- flows of a few pixels, the "well-conditioned" parity fixtures (tests/golden/hr_f_*_fh*.npz)."""
    base = random_state_dict(seed, gain)
    g = torch.Generator(device="cpu")
    g.manual_seed(1000 + seed)

    def U(shape, lo, hi):
        return torch.rand(shape, generator=g, dtype=torch.float32) * (hi - lo) + lo

    spec = param_spec_f()
    sd = OrderedDict()
    for name, shape in spec.items():
        if name in base:
            sd[name] = base[name]
        elif name.endswith("rel_ind"):
            # gma.py:10-14 (buffer of RelPosEmb; unused by the forward)
            dlt = torch.arange(160).view(1, -1) - torch.arange(160).view(-1, 1)
            sd[name] = dlt + 159
        elif name.endswith("latent_tokens"):
            sd[name] = U(shape, -1.0, 1.0)
        elif name.endswith("gamma"):
            sd[name] = torch.tensor([0.35], dtype=torch.float32)
        elif len(shape) >= 2:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            b = gain / math.sqrt(fan_in)
            sd[name] = U(shape, -b, b)
        elif name.endswith(".bias"):
            ws = spec.get(name[: -len("bias")] + "weight")
            if ws is not None and len(ws) >= 2:
                fan_in = 1
                for s in ws[1:]:
                    fan_in *= s
                b = gain / math.sqrt(fan_in)
                sd[name] = U(shape, -b, b)
            else:  # LayerNorm beta
                sd[name] = U(shape, -0.1, 0.1)
        elif name.endswith(".weight"):  # LayerNorm gamma
            sd[name] = U(shape, 0.8, 1.2)
        else:
            raise KeyError(name)
    if flow_head_scale != 1.0:
        for k in (FLOW_HEAD_F + ".weight", FLOW_HEAD_F + ".bias"):
            sd[k] = sd[k] * flow_head_scale
    return sd


FLOW_HEAD_F = "flow_estimator.memory_decoder.update_block.flow_head.conv2"


def random_state_dict_for(model_type: str, seed: int = 0):
    """Seeded random weights of the architecture `config.arch.type` names (the CLIs' --random-init)."""
    t = model_type.lower()
    if t == "gimmvfi_f":
        return random_state_dict_f(seed)
    if t == "gimm":
        return gimm_state_dict(random_state_dict(seed))
    return random_state_dict(seed)
