"""GIMM-VFI-F inference pipeline on the HIP kernels: the FlowFormer flow estimator in front of the shared
motion-INR / frame-synthesis engine (engine.Engine).

Mirrors reference generalizable_INR/gimmvfi_f.py:114-139, 304-384 and flowformer/core/FlowFormer/LatentCostFormer/
{transformer,encoder,decoder,gru,gma,attention,twins}.py + ../encoders.py.  Stage names in ``taps`` are those of
oracle/gimmvfi_f_oracle.py.

Layout: both directions run as one batch of n = 2B images [frame0 of b = 0..B-1, frame1 ...]; direction d of pair b is
image d*B + b.  Token tensors are row matrices [rows, C] (an NHWC tensor flattened); every linear layer is a 1x1
gvfi_conv2d over a [1, 1, rows, C] view.  The 8 latent cost tokens of every cost map live in ONE image-major layout
[(image, token k), pixel p, 128] for the whole encoder, so the "vertical" Twins blocks see plain NHWC images and the
per-map attention over the tokens is a strided gather (gvfi_attn_global) -- the reference permutes the tensor twice
per layer (encoder.py:447-459).

Work the reference performs but never uses is not executed: the feature encoder runs once per image instead of once
per direction (encoder.py:507-509 re-encodes both images for each direction), the mask head + convex upsampling run
for the last of the 32 decoder iterations only (decoder.py:312-314 evaluates all and returns the last).
"""
import os

import torch

from . import lib as L
from .engine import Engine
from .ops import PatchConvLayer, S2DConvLayer, TapSplitConvLayer, TokenChain, View

A = L
K_LAT = 8          # cost_latent_token_num   configs/submission.py:30
TWINS = ((128, 4, 4, 8), (256, 2, 8, 4))   # (embed dim, patch, heads, sr_ratio)   twins.py:1344-1348


def _enc_host(xs, ys, dim):
    """LinearPositionEmbeddingSine (attention.py:170-182) on the host -- weight preparation only (the constant
    key vectors of padded window positions)."""
    fb = torch.linspace(0, dim // 4 - 1, dim // 4)
    x = xs.float()[:, None]
    y = ys.float()[:, None]
    return torch.cat([torch.sin(3.14 * x * fb * (1 / 200)), torch.cos(3.14 * x * fb * (1 / 200)),
                      torch.sin(3.14 * y * fb * (1 / 200)), torch.cos(3.14 * y * fb * (1 / 200))], dim=-1)


FLOW_STAGES = ("enc", "cost", "tok", "upd")


def parse_flow_policy(spec):
    """flow_precision -> {stage: "fp32" | "fp16"} for the stages that leave bf16.  Stages: enc (Twins encoders + channel
    convertor), cost (cost volume + latent cost encoder), and of the 32-iteration decoder tok (flow token: 81-tap cost
    look-up, token encoder, cross-attention against the cost memory -> cost_global) and upd (GMA update block: motion
    encoder, aggregation, ConvGRU, flow head); dec = tok + upd.  Grammar: "bf16" (none), "fp32" (all four in float), "f16"
    (all four on IEEE-half operands: the model default since round 5), or a comma list of stage[:type] with type f16 (IEEE
    half operands, bf16 speed) or fp32 (default), e.g. "dec:f16"."""
    if spec in (None, "", "bf16"):
        return {}
    if spec == "fp32":
        return {st: "fp32" for st in FLOW_STAGES}
    if spec in ("f16", "fp16"):
        return {st: "fp16" for st in FLOW_STAGES}
    pol = {}
    for item in (y.strip() for y in spec.split(",")):
        if not item:
            continue
        name, _, typ = item.partition(":")
        typ = {"": "fp32", "fp32": "fp32", "f32": "fp32", "f16": "fp16", "fp16": "fp16"}.get(typ.strip())
        if typ is None:
            raise ValueError(f"flow_precision: unknown type in {item!r} (f16 or fp32)")
        name = name.strip()
        stages = ("tok", "upd") if name == "dec" else (name,)
        for st in stages:
            if st not in FLOW_STAGES:
                raise ValueError(f"flow_precision: unknown stage {name!r} (stages: {FLOW_STAGES} + 'dec', or 'bf16' / 'fp32')")
            pol[st] = typ
    return pol


class EngineF(Engine):
    def __init__(self, rt, sd, motion_only=False, flow_precision=None, flow_only=False):
        """flow_precision (bf16 runtime only): which stages of the FLOW ESTIMATOR run in float (exact-f32 MFMA) or with
        IEEE-half operands while everything behind it stays bf16 -- see parse_flow_policy.  Those stages run on side
        engines over the same library (`self.side[type]`, flow-estimator layers only); tensors crossing a stage boundary
        are converted by gvfi_copy_channels."""
        self._flow_only = flow_only
        self.flow_policy = parse_flow_policy(flow_precision) if rt.precision == "bf16" else {}
        self.side = {}          # "fp32" / "fp16" -> engine of that precision holding the flow-estimator layers
        super().__init__(rt, sd, motion_only=motion_only)
        if not motion_only:
            for typ in sorted(set(self.flow_policy.values())):
                self.side[typ] = EngineF(rt.sibling(typ), sd, flow_only=True)

    def _build(self, sd):
        if self._flow_only:
            self._build_flow(sd)
            return
        super()._build(sd)

    # ------------------------------------------------------------------ weight preparation
    def _lin(self, sd, key, name=None, **kw):
        kw = {k: v for k, v in kw.items() if k != "wdir" or v}
        w = sd[key + ".weight"]
        b = sd.get(key + ".bias")
        self._add(name or key, w.reshape(w.shape[0], w.shape[1], 1, 1), b, **kw)

    def _lin_cat(self, sd, keys, name, **kw):
        w = torch.cat([sd[k + ".weight"] for k in keys], 0)
        b = torch.cat([sd[k + ".bias"] for k in keys], 0)
        self._add(name, w.reshape(w.shape[0], w.shape[1], 1, 1), b, **kw)

    def _ln(self, sd, key):
        self.ln[key] = (sd[key + ".weight"].float().contiguous().to(self.rt.device),
                        sd[key + ".bias"].float().contiguous().to(self.rt.device))

    def _f32(self, t):
        return t.detach().float().contiguous().to(self.rt.device)

    def _add_s2d(self, k, sd, ksz, ld):
        """filter size == stride, no padding: the same layer as a k x 1 convolution over k * ld contiguous values per patch
        row (ops.S2DConvLayer) when that makes whole K chunks of the LDS-DMA kernel -- a 4 x 4 filter over 3 (8) channels is
        4 taps of 32, an 8 x 8 filter (64 taps: beyond the LDS-DMA kernel's 32-tap masks) 8 taps of 1024."""
        if os.environ.get("GVFI_S2D", "1") != "0" and (ksz * ld) % (4 * self.rt.VE) == 0 and ksz * ksz > 4:
            self.layers[k + ".s2d"] = S2DConvLayer(self.rt, sd[k + ".weight"], sd[k + ".bias"], ld)

    def _conv_k_eq_s(self, k, x, out):
        """x: [N, H, W, ld] tensor (a View of its first channels); the space-to-depth form when the layer has one and the
        tensor is the contiguous whole-pitch tensor it was built for."""
        lay = self.layers.get(k + ".s2d")
        xv = x if isinstance(x, View) else View(x)
        if (lay is not None and xv.coff == 0 and xv.t.is_contiguous() and xv.t.shape[-1] == lay.ld
                and xv.t.shape[1] % lay.k == 0 and xv.t.shape[2] % lay.k == 0):      # (ragged grids: the strided form floors)
            return self.rt.s2d_conv(lay, xv.t, out)
        return self.rt.conv(self.layers[k], xv, out)

    def _build_twins(self, sd, p):
        cin = 3
        for i, (c, patch, heads, sr) in enumerate(TWINS):
            k = f"{p}.svt.patch_embeds.{i}.proj"
            self._add(k, sd[k + ".weight"], sd[k + ".bias"], stride=patch, pad=(0, 0))
            self._add_s2d(k, sd, patch, self.rt.cp(sd[k + ".weight"].shape[1]))
            self._ln(sd, f"{p}.svt.patch_embeds.{i}.norm")
            for j in (0, 1):
                b = f"{p}.svt.blocks.{i}.{j}"
                self._ln(sd, b + ".norm1")
                self._ln(sd, b + ".norm2")
                if j == 0:
                    self._lin(sd, b + ".attn.qkv")
                    bias = sd[b + ".attn.qkv.bias"].float()
                    # padded window positions carry qkv(0) = bias (twins.py:846-857)
                    self.consts[b + ".kpad"] = self._f32(bias[c:2 * c].repeat(49, 1))
                    self.consts[b + ".vpad"] = self._f32(bias[2 * c:].repeat(49, 1))
                else:
                    self._lin(sd, b + ".attn.q")
                    self._lin(sd, b + ".attn.kv")
                    k = b + ".attn.sr"
                    self._add(k, sd[k + ".weight"], sd[k + ".bias"], stride=sr, pad=(0, 0))
                    self._add_s2d(k, sd, sr, c)
                    self._ln(sd, b + ".attn.norm")
                self._lin(sd, b + ".attn.proj")
                self._lin(sd, b + ".mlp.fc1")
                self._lin(sd, b + ".mlp.fc2")
            k = f"{p}.svt.pos_block.{i}.proj.0"
            self.consts[k + ".w"] = self._f32(sd[k + ".weight"].reshape(c, 9).t())   # [9][C]
            self.consts[k + ".b"] = self._f32(sd[k + ".bias"])
            cin = c

    def _build_attn_layer(self, sd, p, merge_qkv, wdir=False):
        self._ln(sd, p + ".norm1")
        self._ln(sd, p + ".norm2")
        if merge_qkv:
            self._lin_cat(sd, (p + ".q", p + ".k", p + ".v"), p + ".qkv")
        else:
            self._lin(sd, p + ".q", wdir=wdir)
            self._lin_cat(sd, (p + ".k", p + ".v"), p + ".kv")
        self._lin(sd, p + ".proj", wdir=wdir)
        self._lin(sd, p + ".ffn.0", wdir=wdir)
        self._lin(sd, p + ".ffn.3", wdir=wdir)

    def _build_vertical(self, sd, p, local):
        self._ln(sd, p + ".norm1")
        self._ln(sd, p + ".norm2")
        a = p + ".attn"
        self._lin(sd, a + ".context_proj")
        self._lin(sd, a + ".v")
        self._lin(sd, a + ".proj")
        self._lin(sd, p + ".mlp.fc1")
        self._lin(sd, p + ".mlp.fc2")
        if local:
            self._lin_cat(sd, (a + ".q", a + ".k"), a + ".qk")
            # window positions beyond the grid: x_qk = 0 + positional code -> k = Wk enc(pos) + bk, v = bv
            # (twins.py:375-407); constants of the weights
            dy, dx = torch.meshgrid(torch.arange(7), torch.arange(7), indexing="ij")
            enc = _enc_host(dx.reshape(-1), dy.reshape(-1), 192)
            wk, bk = sd[a + ".k.weight"].float().cpu(), sd[a + ".k.bias"].float().cpu()
            self.consts[a + ".kpad"] = self._f32(enc @ wk.t() + bk)
            self.consts[a + ".vpad"] = self._f32(sd[a + ".v.bias"].float().repeat(49, 1))
        else:
            self._lin(sd, a + ".q")
            self._lin(sd, a + ".k")
            for k in (a + ".sr_key", a + ".sr_value"):
                self._add(k, sd[k + ".weight"], sd[k + ".bias"], stride=4, pad=(0, 0))
            self._ln(sd, a + ".norm")

    def _build_flow(self, sd):
        """FlowFormer (flowformer/__init__.py:6-18, transformer.py:29-42)."""
        self.ln, self.consts = {}, {}
        self.raft_lanes = int(os.environ.get("GVFI_F_LANES", "2"))   # parallel decoder sequences (see Engine._raft)
        fe = "flow_estimator"
        self._build_twins(sd, fe + ".context_encoder")
        me = fe + ".memory_encoder"
        self._build_twins(sd, me + ".feat_encoder")
        k = me + ".channel_convertor"
        self._add(k, sd[k + ".weight"], None)
        ce = me + ".cost_perceiver_encoder"
        k = ce + ".patch_embed.proj.0"
        self.consts[k + ".w"] = self._f32(sd[k + ".weight"].reshape(16, 36).t())    # [36][16]
        self.consts[k + ".b"] = self._f32(sd[k + ".bias"])
        for k in (ce + ".patch_embed.proj.2", ce + ".patch_embed.proj.4"):
            self._add(k, sd[k + ".weight"], sd[k + ".bias"], stride=2, pad=(2, 2))
        self._conv(sd, ce + ".patch_embed.ffn_with_coord.0")
        self._conv(sd, ce + ".patch_embed.ffn_with_coord.2")
        self._ln(sd, ce + ".patch_embed.norm")
        self.consts["latent"] = self._f32(sd[ce + ".latent_tokens"].reshape(K_LAT, 128))
        self._build_attn_layer(sd, ce + ".input_layer", False)
        for i in range(3):
            self._build_attn_layer(sd, f"{ce}.encoder_layers.{i}", True)
            self._build_vertical(sd, f"{ce}.vertical_encoder_layers.{i}.local_block", True)
            self._build_vertical(sd, f"{ce}.vertical_encoder_layers.{i}.global_block", False)
        md = fe + ".memory_decoder"
        # (K = 81 cost taps padded to 128 zero-weighted channels: two whole K chunks of the LDS-DMA kernel instead of 88
        # channels on the generic one -- 61 -> 12 us per decoder iteration)
        self._conv(sd, md + ".flow_token_encoder.0", cin_pad=128, wdir=True)
        self._conv(sd, md + ".flow_token_encoder.2", wdir=True)
        w, b = sd[md + ".proj.weight"], sd[md + ".proj.bias"]
        self._add("ff.proj_net", w[:128], b[:128])     # tanh half   decoder.py:279-281
        self._add("ff.proj_inp", w[128:], b[128:])     # relu half
        self._build_attn_layer(sd, md + ".decoder_layer.cross_attend", False, wdir=True)
        u = md + ".update_block"
        # the two halves of the flow-token path around its cross-attention as ONE launch each (csrc/token_chain.hip; 16-bit
        # operand types): [flow_token_encoder.0 GELU, .2 (= query), norm1 + position code, q] and [proj + query, norm2,
        # ffn.0 GELU, ffn.3 + x]   decoder.py:84-120, 237-255.  GVFI_F_TOKCHAIN=0 keeps the 5 + 4 separate launches.
        self.chain_a = self.chain_c = None
        # (the whole flow-token path of an iteration as ONE launch was built in round 4 and measured slower -- 74 us against 61 us for
        # the four launches, profiles/r4_tokpath_ab_v2.txt -- and left the library in round 6: tools/experiments/csrc/token_path.hip)
        if self.rt.precision in ("bf16", "fp16") and os.environ.get("GVFI_F_TOKCHAIN", "1") != "0":
            ca_ = md + ".decoder_layer.cross_attend"
            w2 = lambda k: sd[k + ".weight"].reshape(sd[k + ".weight"].shape[0], -1)
            fe0, fe2 = md + ".flow_token_encoder.0", md + ".flow_token_encoder.2"
            self.chain_a = TokenChain(self.rt, [w2(fe0), w2(fe2), w2(ca_ + ".q")],
                                      [sd[fe0 + ".bias"], sd[fe2 + ".bias"], sd[ca_ + ".q.bias"]],
                                      (sd[ca_ + ".norm1.weight"], sd[ca_ + ".norm1.bias"]), 1e-5, 1, act0=A.ACT_GELU)
            self.chain_c = TokenChain(self.rt, [w2(ca_ + ".proj"), w2(ca_ + ".ffn.0"), w2(ca_ + ".ffn.3")],
                                      [sd[ca_ + ".proj.bias"], sd[ca_ + ".ffn.0.bias"], sd[ca_ + ".ffn.3.bias"]],
                                      (sd[ca_ + ".norm2.weight"], sd[ca_ + ".norm2.bias"]), 1e-5, 0, act1=A.ACT_GELU,
                                      res2_from0=True)
        # wdir: the layers of the 32-iteration recurrence take the weights-direct variant of the LDS-DMA kernel
        self._conv(sd, u + ".encoder.convc1", cin_pad=max(self.rt.cp64(145), 192), wdir=True)     # = the pitch of the cost tensor
        self.layers[u + ".encoder.convf1"] = PatchConvLayer(self.rt, sd[u + ".encoder.convf1.weight"],
                                                            sd[u + ".encoder.convf1.bias"], wdir=True)
        for k in ("encoder.convc2", "encoder.convf2", "encoder.conv", "flow_head.conv1"):
            self._conv(sd, f"{u}.{k}", wdir=True)
        for k in ("mask.0", "mask.2"):
            self._conv(sd, f"{u}.{k}")
        k = u + ".flow_head.conv2"
        self.layers[k] = TapSplitConvLayer(self.rt, sd[k + ".weight"], sd[k + ".bias"])
        for n in ("1", "2"):
            # hx = [h(128) | inp(128) | motion(128) | aggregated motion(128)]   gru.py:150-154; the context share (inp)
            # of the gate convolutions is evaluated once per forward, as in the RAFT engine
            wz, wr = sd[f"{u}.gru.convz{n}.weight"], sd[f"{u}.gru.convr{n}.weight"]
            bz, br = sd[f"{u}.gru.convz{n}.bias"], sd[f"{u}.gru.convr{n}.bias"]
            wzr, bzr = torch.cat([wz, wr], 0), torch.cat([bz, br], 0)
            wq, bq = sd[f"{u}.gru.convq{n}.weight"], sd[f"{u}.gru.convq{n}.bias"]
            keep = list(range(0, 128)) + list(range(256, 512))
            self._add(f"gru.zr{n}", wzr[:, keep], None, wdir=True)
            self._add(f"gru.q{n}", wq[:, keep], None, wdir=True)
            self._add(f"gru.zr{n}.ctx", wzr[:, 128:256], bzr)
            self._add(f"gru.q{n}.ctx", wq[:, 128:256], bq)
        # GMA (gma.py:32-115): q pre-scaled by dim_head^-0.5, gamma folded into to_v
        wqk = sd[md + ".att.to_qk.weight"]
        self._add("gma.q", wqk[:128], None)
        self._add("gma.k", wqk[128:], None)
        gamma = float(sd[u + ".aggregator.gamma"].float().item())
        self._wv = (sd[u + ".aggregator.to_v.weight"].float().reshape(128, 128) * gamma).to(self.rt.tdtype).to(self.rt.device)
        self._wv_rep = {}

    # ------------------------------------------------------------------ helpers
    def _tok(self, rows, c):
        return torch.empty((rows, c), dtype=self.rt.tdtype, device=self.rt.device)

    def _tok32(self, rows, c):
        return torch.empty((rows, c), dtype=torch.float32, device=self.rt.device)

    @staticmethod
    def _img(t):
        """[rows, C] token matrix as the [1, 1, rows, C] NHWC tensor gvfi_conv2d reads."""
        return t if t.dim() == 4 else t.view(1, 1, t.shape[0], t.shape[1])

    def _linear(self, name, x, out=None, act=A.ACT_NONE, res=None, x1=None, f32=False):
        """out = act(x [| x1] W^T + b) (+ res); x, out: token matrices or Views of them.  f32: the result is a float
        tensor -- the residual streams of the transformer blocks stay in float (bf16 only rounds the operands of
        the contractions, never the running sum of the block updates)."""
        rt = self.rt
        lay = self.layers[name]
        xv = x if isinstance(x, View) else View(x)
        rows = xv.npix
        if out is None:
            out = self._tok32(rows, lay.cout) if f32 else self._tok(rows, lay.cout)
        ov = out if isinstance(out, View) else View(out)
        mk = lambda v: View(v.t.view(1, 1, rows, v.t.shape[-1]), v.coff, v.c)
        rv = None
        if res is not None:
            rv = mk(res if isinstance(res, View) else View(res))
        x1v = None
        if x1 is not None:
            x1v = mk(x1 if isinstance(x1, View) else View(x1))
        rt.conv(lay, mk(xv), mk(ov), x1=x1v, act1=act, res=rv)
        return out

    def _mlp_res(self, p, x, norm, eps, fc1, fc2):
        """x + fc2(gelu(fc1(LN(x)))) on the float residual stream."""
        y = self.rt.layernorm(x, self.ln[norm], eps)
        h = self._linear(fc1, y, act=A.ACT_GELU)
        return self._linear(fc2, h, res=x, f32=True)

    # ------------------------------------------------------------------ Twins-SVT (two stages)   encoders.py:21-48
    def _twins(self, img, p):
        rt, Ls, C_ = self.rt, self.layers, self.consts
        n = img.shape[0]
        x = View(img, 0, 3)
        feats = []
        for i, (c, patch, heads, sr) in enumerate(TWINS):
            src = x if isinstance(x, View) else View(x)
            H, W = src.t.shape[1:3]
            assert H % patch == 0 and W % patch == 0
            h, w = H // patch, W // patch
            hd = c // heads
            rows = n * h * w
            emb = rt.act(n, h, w, c)
            self._conv_k_eq_s(f"{p}.svt.patch_embeds.{i}.proj", src, emb)
            t = rt.layernorm(emb.view(rows, c), self.ln[f"{p}.svt.patch_embeds.{i}.norm"], 1e-5)
            # ---- block 0: locally-grouped attention (ws 7)   twins.py:814-867
            b = f"{p}.svt.blocks.{i}.0"
            y = rt.layernorm(t, self.ln[b + ".norm1"], 1e-6)
            qkv = self._linear(b + ".attn.qkv", y)
            a = self._tok(rows, c)
            rt.attn_window(View(qkv, 0, c), View(qkv, c, c), View(qkv, 2 * c, c), C_[b + ".kpad"], C_[b + ".vpad"], a,
                           n, h, w, 7, heads, hd)
            t = self._linear(b + ".attn.proj", a, res=t, f32=True)
            t = self._mlp_res(b, t, b + ".norm2", 1e-6, b + ".mlp.fc1", b + ".mlp.fc2")
            # ---- PEG   twins.py:1100-1119
            k = f"{p}.svt.pos_block.{i}.proj.0"
            t = rt.dwconv3x3_res(t.view(n, h, w, c), C_[k + ".w"], C_[k + ".b"]).view(rows, c)
            # ---- block 1: global sub-sampled attention   twins.py:870-925
            b = f"{p}.svt.blocks.{i}.1"
            y = rt.layernorm(t, self.ln[b + ".norm1"], 1e-6)
            q = self._linear(b + ".attn.q", y)
            hs, ws_ = h // sr, w // sr
            m = hs * ws_
            s = rt.act(n, hs, ws_, c)
            self._conv_k_eq_s(b + ".attn.sr", y.view(n, h, w, c), s)
            s = rt.layernorm(s.view(n * m, c), self.ln[b + ".attn.norm"], 1e-5)
            kv = self._linear(b + ".attn.kv", s)
            a = self._tok(rows, c)
            N = h * w
            rt.attn_global(q, (N, 0, 1), View(kv, 0, c), View(kv, c, c), (m, 0, 1), a, (N, 0, 1), n, 1, N, m, heads, hd)
            t = self._linear(b + ".attn.proj", a, res=t, f32=True)
            t = self._mlp_res(b, t, b + ".norm2", 1e-6, b + ".mlp.fc1", b + ".mlp.fc2")
            if t.dtype != rt.tdtype:     # stage output in the activation type: operand of the next convolutions
                t = rt.copy(t, self._tok(rows, c), c).t
            x = t.view(n, h, w, c)
            feats.append(x)
        return feats

    # ------------------------------------------------------------------ cost-volume encoder   encoder.py:349-466
    def _vertical(self, p, x, ctx, n, B, h8, w8, local):
        """Block(with_rpe, vert_c_dim 64) over the [n*K] latent images (twins.py:1028-1097, 331-546)."""
        rt, Ls, C_ = self.rt, self.layers, self.consts
        P8 = h8 * w8
        n_img = n * K_LAT
        rows = n_img * P8
        a_ = p + ".attn"
        y = rt.layernorm(x, self.ln[p + ".norm1"], 1e-5)
        ctxp = self._linear(a_ + ".context_proj", ctx)            # [n*P8, 64]
        att = self._tok(rows, 128)
        if local:
            xqk = self._tok(rows, 192)
            rt.ff_xqk(y, ctxp, xqk, n_img, h8, w8, K_LAT, B, 1, table=self._enc_table(1, h8, w8))
            qk = self._linear(a_ + ".qk", xqk)
            v = self._linear(a_ + ".v", y)
            rt.attn_window(View(qk, 0, 128), View(qk, 128, 128), v, C_[a_ + ".kpad"], C_[a_ + ".vpad"], att, n_img, h8, w8,
                           7, 8, 16)
        else:
            xq = self._tok(rows, 192)
            rt.ff_xqk(y, ctxp, xq, n_img, h8, w8, K_LAT, B, 2, table=self._enc_table(2, h8, w8))
            q = self._linear(a_ + ".q", xq)
            xk = self._tok(rows, 192)
            rt.ff_xqk(y, ctxp, xk, n_img, h8, w8, K_LAT, B, 0)
            # the reference zero-extends the token grid to a multiple of sr_ratio on the right / bottom before the
            # sub-sampling convolutions (twins.py:471-476); queries of the extension are dropped (:540-541)
            hp, wp = (h8 + 3) // 4 * 4, (w8 + 3) // 4 * 4
            xk4, y4 = xk.view(n_img, h8, w8, 192), y.view(n_img, h8, w8, 128)
            if (hp, wp) != (h8, w8):    # memory plumbing only
                xk4 = torch.nn.functional.pad(xk4, (0, 0, 0, wp - w8, 0, hp - h8))
                y4 = torch.nn.functional.pad(y4, (0, 0, 0, wp - w8, 0, hp - h8))
            hs, ws_ = hp // 4, wp // 4
            m = hs * ws_
            sk = rt.act(n_img, hs, ws_, 128)
            rt.conv(Ls[a_ + ".sr_key"], xk4, sk)
            sv = rt.act(n_img, hs, ws_, 128)
            rt.conv(Ls[a_ + ".sr_value"], y4, sv)
            sk = rt.layernorm(sk.view(n_img * m, 128), self.ln[a_ + ".norm"], 1e-5)
            sv = rt.layernorm(sv.view(n_img * m, 128), self.ln[a_ + ".norm"], 1e-5)
            rt.pos_embed(self._grid(hs, ws_), m, 4.0, 0.0, 128, sk, n_img * m, True)
            k = self._linear(a_ + ".k", sk)
            v = self._linear(a_ + ".v", sv)
            rt.attn_global(q, (P8, 0, 1), k, v, (m, 0, 1), att, (P8, 0, 1), n_img, 1, P8, m, 8, 16)
        x = self._linear(a_ + ".proj", att, res=x, f32=True)
        return self._mlp_res(p, x, p + ".norm2", 1e-5, p + ".mlp.fc1", p + ".mlp.fc2")

    def _enc_table(self, mode, h, w):
        """positional code of the window (mode 1) / grid (mode 2) positions, evaluated once per forward"""
        key = ("enc", mode, h, w)
        if key not in self._grids:
            self._grids[key] = self.rt.ff_pos_table(h, w, 192, mode)
        return self._grids[key]

    def _grid(self, h, w):
        key = ("grid", h, w)
        if key not in self._grids:
            self._grids[key] = self.rt.coords_init(1, h, w)
        return self._grids[key]

    def _cost_encoder(self, vol, ctx, n, B, h8, w8, taps):
        rt, Ls, C_ = self.rt, self.layers, self.consts
        P8 = h8 * w8
        maps = n * P8
        ce = "flow_estimator.memory_encoder.cost_perceiver_encoder"
        pe = ce + ".patch_embed"
        # ---- PatchEmbed of the cost maps   encoder.py:30-96
        hp, wp = (h8 + 7) // 8 * 8, (w8 + 7) // 8 * 8
        e1 = rt.cost_embed1(vol, C_[pe + ".proj.0.w"], C_[pe + ".proj.0.b"], maps, h8, w8, hp // 2, wp // 2)
        e2 = rt.act(maps, hp // 4, wp // 4, 32)
        rt.conv(Ls[pe + ".proj.2"], e1, e2, act1=A.ACT_RELU)
        h3, w3 = hp // 8, wp // 8
        T = h3 * w3
        tok = rt.act(maps, h3, w3, 128)
        rt.conv(Ls[pe + ".proj.4"], e2, View(tok, 0, 64))
        rt.pos_embed(self._grid(h3, w3), T, 8.0, 4.0, 64, View(tok, 64, 64), maps * T, False)
        f1 = self._linear(pe + ".ffn_with_coord.0", tok.view(maps * T, 128), act=A.ACT_RELU)
        f2 = self._linear(pe + ".ffn_with_coord.2", f1)
        xt = rt.layernorm(f2, self.ln[pe + ".norm"], 1e-5)
        if taps is not None:
            taps["f01_cost_tokens"] = xt.view(maps, T, 128)[:B * P8]
        # ---- input_layer: the 8 learned latent tokens attend to the 28 tokens of every cost map   encoder.py:282-346
        ip = ce + ".input_layer"
        lat = C_["latent"].to(rt.tdtype)
        qlat = self._linear(ip + ".q", rt.layernorm(lat, self.ln[ip + ".norm1"], 1e-5))
        kv = self._linear(ip + ".kv", xt)
        rows = n * K_LAT * P8
        att = self._tok(rows, 128)
        lay = (K_LAT * P8, 1, P8)          # row of (image b, pixel p, token i) in the image-major latent layout
        rt.attn_global(qlat, (0, 0, 1), View(kv, 0, 128), View(kv, 128, 128), (P8 * T, T, 1), att, lay, n, P8, K_LAT, T,
                       8, 16)
        short = self._tok32(rows, 128)
        rt.tile_rows(C_["latent"], short, rows, P8, K_LAT, 128)
        x = self._linear(ip + ".proj", att, res=short, f32=True)
        y = rt.layernorm(x, self.ln[ip + ".norm2"], 1e-5)
        x = self._linear(ip + ".ffn.3", self._linear(ip + ".ffn.0", y, act=A.ACT_GELU), res=x, f32=True)
        short_cut = x
        if taps is not None:
            taps["f01_latent_in"] = x
        for idx in range(3):
            # ---- SelfAttentionLayer over the 8 tokens of a map   encoder.py:214-279
            ep = f"{ce}.encoder_layers.{idx}"
            y = rt.layernorm(x, self.ln[ep + ".norm1"], 1e-5)
            qkv = self._linear(ep + ".qkv", y)
            att = self._tok(rows, 128)
            rt.attn_global(View(qkv, 0, 128), lay, View(qkv, 128, 128), View(qkv, 256, 128), lay, att, lay, n, P8, K_LAT,
                           K_LAT, 8, 16)
            x = self._linear(ep + ".proj", att, res=x, f32=True)
            y = rt.layernorm(x, self.ln[ep + ".norm2"], 1e-5)
            x = self._linear(ep + ".ffn.3", self._linear(ep + ".ffn.0", y, act=A.ACT_GELU), res=x, f32=True)
            vp = f"{ce}.vertical_encoder_layers.{idx}"
            x = self._vertical(vp + ".local_block", x, ctx, n, B, h8, w8, True)
            x = self._vertical(vp + ".global_block", x, ctx, n, B, h8, w8, False)
            if taps is not None and idx == 0:
                taps["f01_latent_l0"] = x
        mem = self._tok(rows, 128)
        rt.copy(x, mem, 128, add=short_cut)      # cost_encoder_res   encoder.py:462-463
        return mem

    # ------------------------------------------------------------------ FlowFormer (both directions batched)
    def _ff_encode(self, imgA, B, seq):
        """Stage "enc": both Twins encoders + channel convertor -> context features (1/4, 1/8) and matching features."""
        rt, Ls = self.rt, self.layers
        n = 2 * B
        fe = "flow_estimator"
        h8, w8 = imgA.shape[1] // 8, imgA.shape[2] // 8
        if seq and B > 1:
            # consecutive pairs: both Twins encoders are per image, so they run on the B+1 distinct frames (Engine._raft)
            imgU = torch.cat([imgA[:B], imgA[n - 1:n]], 0)
            idx = self._seq_index(B, imgA.device)
            cfeat = [f[idx] for f in self._twins(imgU, fe + ".context_encoder")]
            ff = self._twins(imgU, fe + ".memory_encoder.feat_encoder")[1][idx]
            fmap = rt.act(n, h8, w8, 256)
            rt.conv(Ls[fe + ".memory_encoder.channel_convertor"], ff, fmap)
            return cfeat, fmap
        # the two Twins encoders read the same images and meet only in the cost stage: two parallel launch sequences
        # (as the RAFT encoders of Engine._raft; GVFI_ENC_LANES=0: A/B switch)
        k_enc = 2 if (self.enc_lanes and rt.on_gpu) else 1
        res = {}
        with rt.lanes(k_enc) as lanes:
            with lanes[0]:
                ff = self._twins(imgA, fe + ".memory_encoder.feat_encoder")[1]
                fmap = rt.act(n, h8, w8, 256)
                rt.conv(Ls[fe + ".memory_encoder.channel_convertor"], ff, fmap)
            with lanes[k_enc - 1]:
                res["cfeat"] = self._twins(imgA, fe + ".context_encoder")
        cfeat = res["cfeat"]
        if k_enc > 1 and rt.ev_log is None and not torch.cuda.is_current_stream_capturing():
            cur = torch.cuda.current_stream(rt.device)
            for t_ in cfeat:
                (t_.t if isinstance(t_, View) else t_).record_stream(cur)
        return cfeat, fmap

    def _ff_cost(self, fmap, context, n, B, h8, w8, taps):
        """Stage "cost": all-pairs cost volume + latent cost encoder -> volume (float) and cost memory."""
        rt = self.rt
        self._grids = {}
        P8 = h8 * w8
        ctx_rows = context.view(n * P8, 256)
        # all-pairs cost volume of both directions (encoder.py:489-506; no 1/sqrt(d)): image i against its partner
        # (two launches on the halves of fmap instead of a swapped copy of it)
        vol = rt.f32(n * P8, P8)
        vv = vol.view(n, h8, w8, P8)
        for i0, wt in ((0, fmap[B:]), (B, fmap[:B])):
            rt.conv(None, fmap[i0:i0 + B], View(vv[i0:i0 + B]), groups=B, w_group_stride=P8 * fmap.shape[-1], w_raw=wt, cout=P8)
        mem = self._cost_encoder(vol, ctx_rows, n, B, h8, w8, taps)
        if taps is not None:
            taps["f01_ffeat"] = fmap[:B]
            taps["f01_cost_memory"] = mem.view(n, K_LAT, P8, 128)[:B]
        return vol, mem

    def _ff_decode(self, vol, mem, context, n, B, h8, w8, iters, taps, tok=None, mem_tok=None, cvt=None):
        """Stages "tok" + "upd": MemoryDecoder (decoder.py:257-321) -> full-resolution flows [n,H,W,2] float.  This engine
        runs the update block; `tok` (default: this engine) runs the flow-token path of every iteration on `mem_tok` (the
        cost memory in its activation type) and its cost tensor is converted with `cvt` for the update block."""
        rt, Ls = self.rt, self.layers
        tok = self if tok is None else tok
        mem_tok = mem if mem_tok is None else mem_tok
        rtt = tok.rt
        if not hasattr(self, "_grids"):
            self._grids = {}
        P8 = h8 * w8
        md = "flow_estimator.memory_decoder"
        hA = rt.act(n, h8, w8, 128)
        hB = rt.act(n, h8, w8, 128)
        inp = rt.act(n, h8, w8, 128)
        # bf16 mode: float GRU state beside the bf16 operand copies, as in Engine._raft
        sf = self.gru_state_f32 and rt.precision in ("bf16", "fp16")
        h32A = rt.f32(n, h8, w8, 128) if sf else None
        h32B = rt.f32(n, h8, w8, 128) if sf else None
        if sf:
            rt.conv(Ls["ff.proj_net"], context, h32A, act1=A.ACT_TANH)
            rt.copy(h32A, hA, 128)
        else:
            rt.conv(Ls["ff.proj_net"], context, hA, act1=A.ACT_TANH)
        rt.conv(Ls["ff.proj_inp"], context, inp, act1=A.ACT_RELU)
        # GMA attention (once): softmax(q k^T / sqrt(128))   gma.py:53-76
        gq = rt.act(n, h8, w8, 128)
        gk = rt.act(n, h8, w8, 128)
        rt.conv(Ls["gma.q"], inp, gq, out_scale=128 ** -0.5)
        rt.conv(Ls["gma.k"], inp, gk)
        sim = rt.f32(n * P8, P8)
        rt.conv(None, gq, View(sim.view(n, h8, w8, P8)), groups=n, w_group_stride=P8 * 128, w_raw=gk, cout=P8)
        P8p = rt.cp(P8)              # row pitch of the attention matrix / of V^T: pad columns are zero
        attn = torch.empty((n, 1, P8, P8p), dtype=rt.tdtype, device=rt.device)
        rt.softmax_rows(sim, P8, attn.view(n * P8, P8p), n * P8)
        del sim
        if n not in self._wv_rep:
            self._wv_rep[n] = self._wv.view(1, 1, 128, 128).expand(n, 1, 128, 128).contiguous()
        wv_rep = self._wv_rep[n]
        ca = md + ".decoder_layer.cross_attend"
        kvm = tok._linear(ca + ".kv", mem_tok)                      # [n*K*P8, 128] = [key(64) | value(64)]
        coords = rt.coords_init(n, h8, w8)
        coords_alt = rt.f32(n, h8, w8, 2)
        corr = rt.act(n, h8, w8, 145, zero=True, pitch=max(rt.cp64(145), 192), once="ffdec.corr")    # [cost_global(64) | cost_forward(81) | 0 ...]
        # (mixed policy: the token path fills its own copy in its activation type, converted once per iteration)
        corr_t = corr if tok is self else rtt.act(n, h8, w8, 145, zero=True, pitch=max(rtt.cp64(145), 192), once="ffdec.corr_t")
        flow8 = rt.act(n, h8, w8, 2, zero=True, once="ffdec.flow8")
        X = rt.act(n, h8, w8, 256)          # [motion(126) flow(2) | aggregated motion(128)]   gru.py:150-152
        mfc = rt.act(n, h8, w8, 128)
        vT = rt.act(n, 1, 128, P8, zero=True, pitch=P8p, once="ffdec.vT")      # (columns P8 .. P8p: K padding of the aggregation GEMM)
        c1 = rt.act(n, h8, w8, 256)
        corflo = rt.act(n, h8, w8, 256)
        f1 = rt.act(n, h8, w8, 128)
        zbuf = rt.f32(n, h8, w8, 128) if sf else rt.act(n, h8, w8, 128)
        rh = rt.act(n, h8, w8, 128)
        fh = rt.act(n, h8, w8, 256)
        u = md + ".update_block"
        ctxg = {}
        for key in ("gru.zr1", "gru.q1", "gru.zr2", "gru.q2"):
            lay = Ls[key + ".ctx"]
            ctxg[key] = rt.f32(n, h8, w8, lay.cout)
            rt.conv(lay, inp, ctxg[key])
        fcol = rt.act(n, h8, w8, Ls[u + ".encoder.convf1"].kpad)
        fpart = rt.f32(n, h8, w8, 20)
        lay_q = (P8, 1, 0)
        lay_k = (K_LAT * P8, 1, P8)
        iters = 32 if iters is None else iters

        def chain(a, b):
            """The decoder iterations of images [a, b) (decoder.py:289-314): every tensor of the recurrence is per image
            (cost maps, latent memory, GMA attention matrix, GRU state), so sub-batches are independent sequences."""
            m = b - a
            rows = m * P8
            vol_s = vol[a * P8:b * P8]
            kvm_s = kvm[a * K_LAT * P8:b * K_LAT * P8]
            co, cr, fl, Xs, mf, vt = coords[a:b], corr[a:b], flow8[a:b], X[a:b], mfc[a:b], vT[a:b]
            crt = corr_t[a:b]
            crt_rows = crt.view(rows, crt.shape[-1])
            at, wv = attn[a:b], wv_rep[a:b]
            c1_, cfl, f1_, zb, rh_, fh_ = c1[a:b], corflo[a:b], f1[a:b], zbuf[a:b], rh[a:b], fh[a:b]
            ha, hb, fc, fp = hA[a:b], hB[a:b], fcol[a:b], fpart[a:b]
            h32 = (h32A[a:b], h32B[a:b]) if sf else (None, None)
            cx = {k_: v[a:b] for k_, v in ctxg.items()}
            cr_rows = cr.view(rows, cr.shape[-1])
            fused = rt.fuse_seam and taps is None
            co_home, co_alt = co, coords_alt[a:b]     # (the fused seam updates the coordinates out of place: ping-pong)
            tapl, patl = Ls[u + ".flow_head.conv2"], Ls[u + ".encoder.convf1"]
            for it in range(iters):
                if fused:
                    # one launch: coords1 += flow-head output of the previous iteration, flow activation, 7x7 patch of it
                    co, co_alt = (rt.flow_step(tapl, patl, fp, co, fl, View(Xs, 126, 2), fc, coords_out=co_alt), co) if it > 0 else \
                        (rt.flow_step(tapl, patl, None, co, fl, View(Xs, 126, 2), fc), co_alt)
                # flow token: 81 taps of the query's own cost map   decoder.py:237-255, 293-301
                rtt.cost_lookup(vol_s, co, View(crt, 64, 81), rows, h8, w8)
                if tok.chain_a is not None and taps is None:
                    query, q, a_ = tok._tok(rows, 64), tok._tok(rows, 64), tok._tok(rows, 64)
                    rtt.token_chain(tok.chain_a, View(crt_rows, 64, 128), q, out1=query, coords=co, period=rows)
                    rtt.attn_global(q, lay_q, View(kvm_s, 0, 64), View(kvm_s, 64, 64), lay_k, a_, lay_q, m, P8, 1, K_LAT, 8, 8)
                    rtt.token_chain(tok.chain_c, a_, View(crt_rows, 0, 64), in1=query, res0=query)
                else:
                    t1 = tok._linear(md + ".flow_token_encoder.0", View(crt_rows, 64, 128), act=A.ACT_GELU)   # 81 taps + zeros
                    query = tok._linear(md + ".flow_token_encoder.2", t1)
                    # cross-attention of the one query against the map's 8 latent tokens   decoder.py:84-120
                    qn = rtt.layernorm(query, tok.ln[ca + ".norm1"], 1e-5)
                    rtt.pos_embed(co, rows, 1.0, 0.0, 64, qn, rows, True)
                    q = tok._linear(ca + ".q", qn)
                    a_ = tok._tok(rows, 64)
                    rtt.attn_global(q, lay_q, View(kvm_s, 0, 64), View(kvm_s, 64, 64), lay_k, a_, lay_q, m, P8, 1, K_LAT, 8, 8)
                    x = tok._linear(ca + ".proj", a_, x1=query, res=query)
                    y = rtt.layernorm(x, tok.ln[ca + ".norm2"], 1e-5)
                    tok._linear(ca + ".ffn.3", tok._linear(ca + ".ffn.0", y, act=A.ACT_GELU), out=View(crt_rows, 0, 64), res=x)
                if tok is not self:
                    cvt(View(crt_rows, 0, 145), View(cr_rows, 0, 145), 145)
                # GMAUpdateBlock   gru.py:130-160
                if not fused:
                    rt.flow_pack(co, fl, View(Xs, 126, 2))
                if fused:
                    # (as in Engine._raft: the motion encoder's two branches, gru.py:96-116, as one launch per level)
                    rt.conv_pair(dict(layer=Ls[u + ".encoder.convc1"], x0=cr, out=c1_, act1=A.ACT_RELU),
                                 dict(layer=patl.inner, x0=fc, out=f1_, act1=A.ACT_RELU))
                    rt.conv_pair(dict(layer=Ls[u + ".encoder.convc2"], x0=c1_, out=View(cfl, 0, 192), act1=A.ACT_RELU),
                                 dict(layer=Ls[u + ".encoder.convf2"], x0=f1_, out=View(cfl, 192, 64), act1=A.ACT_RELU))
                else:
                    rt.conv(Ls[u + ".encoder.convc1"], cr, c1_, act1=A.ACT_RELU)
                    rt.conv(Ls[u + ".encoder.convc2"], c1_, View(cfl, 0, 192), act1=A.ACT_RELU)
                    rt.patch_conv(patl, View(fl, 0, 2), f1_, scratch=fc, act1=A.ACT_RELU)
                    rt.conv(Ls[u + ".encoder.convf2"], f1_, View(cfl, 192, 64), act1=A.ACT_RELU)
                rt.conv(Ls[u + ".encoder.conv"], cfl, View(Xs, 0, 126), act1=A.ACT_RELU)
                # global motion aggregation: X[128:256] = mf + gamma * attn @ (mf Wv^T)   gma.py:101-115
                rt.copy(View(Xs, 0, 128), mf, 128)
                rt.conv(None, wv, View(vt, 0, P8), groups=m, w_group_stride=P8 * 128, w_raw=mf, cout=P8)
                Xr = Xs.view(m, 1, P8, 256)
                rt.conv(None, View(at, 0, P8), View(Xr, 128, 128), groups=m, w_group_stride=128 * P8p, w_raw=vt, cout=128,
                        res=View(Xr, 0, 128))
                hc, hn = ha, hb
                sc, sn = h32
                for nn_ in ("1", "2"):
                    rt.conv(Ls["gru.zr" + nn_], hc, zb, x1=Xs, epi=A.EPI_GRU_ZR, y2=rh_, aux0=sc if sf else hc,
                            res=cx["gru.zr" + nn_], state_f32=sf)
                    rt.conv(Ls["gru.q" + nn_], rh_, hn, x1=Xs, epi=A.EPI_GRU_Q, aux0=sc if sf else hc, aux1=zb,
                            y2=sn if sf else None, res=cx["gru.q" + nn_], state_f32=sf)
                    hc, hn = hn, hc
                    sc, sn = sn, sc
                rt.conv(Ls[u + ".flow_head.conv1"], ha, fh_, act1=A.ACT_RELU)
                if fused and it + 1 < iters:
                    rt.conv(tapl.inner, fh_, View(fp, 0, 18))      # per-tap partial sums; summed by the next flow_step
                else:
                    rt.tap_split_conv(tapl, fh_, View(co_home), res=View(co), scratch=fp)    # (-> home tensor)
                if taps is not None and it in (0, iters - 1):
                    taps[f"f01_cost_fwd_it{it}"] = corr[:B, ..., 64:145].clone()
                    taps[f"f01_cost_global_it{it}"] = corr[:B, ..., 0:64].clone()
                    taps[f"f01_net_it{it}"] = hA[:B].clone()
                    taps[f"f01_coords_it{it}"] = coords[:B].clone()

        # parallel launch sequences over image sub-batches, as in Engine._raft (the 32 x 27 launches of the decoder are
        # M = n*h8*w8-row problems that under-fill the chip one at a time)
        k = 1 if taps is not None else max(1, min(self.raft_lanes, n))
        with rt.lanes(k) as lanes:
            for i in range(k):
                with lanes[i]:
                    chain(i * n // k, (i + 1) * n // k)
        rt.conv(Ls[u + ".mask.0"], hA, fh, act1=A.ACT_RELU)
        mask = rt.f32(n, h8, w8, 576)
        rt.conv(Ls[u + ".mask.2"], fh, mask, out_scale=0.25)
        return rt.convex_upsample(coords, mask)

    def _cvt(self, t, eng):
        """t in the activation type of engine `eng` (stage boundary of the precision policy; gvfi_copy_channels converts
        between float and ONE 16-bit type, so bf16 <-> half goes through a float temporary)."""
        want = eng.rt.tdtype
        if t.dtype == want:
            return t
        c = t.shape[-1]
        v = lambda x: View(x.view(-1, c))

        def one(src, dst_dtype):
            rt16 = next(e.rt for e in (self, *self.side.values())
                        if e.rt.tdtype == (src.dtype if src.dtype != torch.float32 else dst_dtype))
            dst = torch.empty(src.shape, dtype=dst_dtype, device=src.device)
            rt16.copy(v(src), v(dst), c)
            return dst

        if t.dtype != torch.float32 and want != torch.float32:
            t = one(t, torch.float32)
        return one(t, want)

    def _cvt_into(self, src, dst, c):
        """per-iteration hand-over of the token path's cost tensor to the update block (Views, c channels)"""
        if src.t.dtype != torch.float32 and dst.t.dtype != torch.float32 and src.t.dtype != dst.t.dtype:
            tmp = torch.empty((src.npix, c), dtype=torch.float32, device=src.t.device)
            self._cvt_into(src, View(tmp), c)
            src = View(tmp)
        d16 = src.t.dtype if src.t.dtype != torch.float32 else dst.t.dtype
        rt16 = next(e.rt for e in (self, *self.side.values()) if e.rt.tdtype == d16)
        rt16.copy(src, dst, c)

    def _flowformer(self, imgA, B, iters, taps, seq=False):
        n = 2 * B
        H, W = imgA.shape[1:3]
        h8, w8 = H // 8, W // 8
        eng = {st: (self.side[self.flow_policy[st]] if st in self.flow_policy else self) for st in FLOW_STAGES}
        img_e = imgA
        if eng["enc"] is not self:      # the side stage reads the float image, not its bf16 rounding
            img_e, _ = eng["enc"].rt.prep_images(self._cur_img_xs)
        cfeat, fmap = eng["enc"]._ff_encode(img_e, B, seq)
        ec = eng["cost"]
        vol, mem = ec._ff_cost(self._cvt(fmap, ec), self._cvt(cfeat[1], ec), n, B, h8, w8, taps)
        ed, et = eng["upd"], eng["tok"]
        flow_up = ed._ff_decode(vol, self._cvt(mem, ed), self._cvt(cfeat[1], ed), n, B, h8, w8, iters, taps, tok=et,
                                mem_tok=self._cvt(mem, et), cvt=self._cvt_into)
        cfeat = [self._cvt(f, self) for f in cfeat]
        fmap = self._cvt(fmap, self)
        if taps is not None:
            taps["f01_context"] = cfeat[1][:B]
            taps["f01_cfeat4"] = cfeat[0][:B]
        return flow_up, fmap, cfeat, (h8, w8)


    def _flow(self, imgA, B, iters, taps, seq=False, front=None, wrap_side=None):
        """(front: the caller's flow-independent work -- run by the caller itself here, after the flow estimator.)
        gimmvfi_f.py:114-139: FlowFormer both ways, BidirCorrBlock on its (converted) features, context features
        of the Twins context encoder at 1/4 and 1/8 -- no projections in this model."""
        flow_up, fmap, cfeat, (h8, w8) = self._flowformer(imgA, B, iters, taps, seq)
        pyr, pyrT = self._bidir_pyramids(fmap, B, h8, w8)
        return flow_up[:B], flow_up[B:], {"pyr": pyr, "pyrT": pyrT, "feat4": cfeat[0], "feat8": cfeat[1]}, (h8, w8)
