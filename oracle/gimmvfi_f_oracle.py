"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the GIMM-VFI-F inference path (FlowFormer flow estimator).

A functional, state_dict-driven restatement (plain torch fp32 on CPU) of
generalizable_INR/gimmvfi_f.py:304-384 and the FlowFormer it calls
(flowformer/core/FlowFormer/LatentCostFormer/*).  Everything behind the flow estimator is shared with the
GIMM-VFI-R oracle (gimmvfi_r_oracle.forward_after_flow).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this file; the product path never does.

Parity status: PINNED against the reference's own FlowFormer code run here on CPU (oracle/ref_harness.py,
tests/test_oracle_pin.py::test_f_oracle_matches_reference_live, bit-exact) and against tests/golden/f_*.npz made
by it (oracle/make_golden_f.py).  One boundary is NOT pinned against its original: the Twins-SVT backbone is
`timm.create_model("twins_svt_large")` in the reference (encoders.py:10, timm==0.4.12, absent from this image
and from /root/reference); it is restated here from the reference's vendored copy of that class
(LatentCostFormer/twins.py:814-983, 1028-1290; model kwargs in the comment at :1344-1348) plus timm's published
Mlp (fc1 -> GELU -> fc2) -- "parity unpinned" for the timm wheel itself.

Paths below are relative to /root/reference/src/models/generalizable_INR/flowformer/core/FlowFormer/.
"""
import torch
import torch.nn.functional as F
from einops import rearrange
from torch import einsum

import gimmvfi_r_oracle as R

LC = "LatentCostFormer/"


def _lin(sd, key, x):
    # .contiguous(): torch's CPU linear takes a different (not bit-identical) path for strided inputs depending on
    # whether the weight is a Parameter that requires grad (the reference's nn.Linear) or a plain tensor (here);
    # on a contiguous input both agree bit for bit.
    return F.linear(x.contiguous(), sd[key + ".weight"], sd.get(key + ".bias"))


def _ln(sd, key, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[key + ".weight"], sd[key + ".bias"], eps)


def _conv(sd, key, x, stride=1, padding=0, groups=1):
    return F.conv2d(x, sd[key + ".weight"], sd.get(key + ".bias"), stride=stride, padding=padding, groups=groups)


def coords_grid(b, h, w):
    # ../utils/utils.py:129-132
    c = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    return torch.stack(c[::-1], dim=0).float()[None].repeat(b, 1, 1, 1)


def bilinear_sampler(img, coords):
    # ../utils/utils.py:83-97
    H, W = img.shape[-2:]
    xg, yg = coords.split([1, 1], dim=-1)
    xg = 2 * xg / (W - 1) - 1
    yg = 2 * yg / (H - 1) - 1
    return F.grid_sample(img, torch.cat([xg, yg], dim=-1), align_corners=True)


def linear_pos_embedding_sine(x, dim=128, nf=1 / 200):
    # LatentCostFormer/attention.py:170-182 (note the literal 3.14)
    fb = torch.linspace(0, dim // 4 - 1, dim // 4)
    return torch.cat(
        [
            torch.sin(3.14 * x[..., -2:-1] * fb * nf),
            torch.cos(3.14 * x[..., -2:-1] * fb * nf),
            torch.sin(3.14 * x[..., -1:] * fb * nf),
            torch.cos(3.14 * x[..., -1:] * fb * nf),
        ],
        dim=-1,
    )


def _mlp(sd, p, x):
    # timm 0.4.12 layers/mlp.py: fc1 -> GELU -> fc2 (dropouts are identities in eval)
    return _lin(sd, p + ".fc2", F.gelu(_lin(sd, p + ".fc1", x)))


# --------------------------------------------------------------------------- Twins-SVT (a25)
def _lsa(sd, p, x, size, heads, ws=7):
    """LocallyGroupedAttn, twins.py:814-867."""
    B, N, C = x.shape
    H, W = size
    x = x.view(B, H, W, C)
    pad_r = (ws - W % ws) % ws
    pad_b = (ws - H % ws) % ws
    x = F.pad(x, (0, 0, 0, pad_r, 0, pad_b))
    _, Hp, Wp, _ = x.shape
    _h, _w = Hp // ws, Wp // ws
    x = x.reshape(B, _h, ws, _w, ws, C).transpose(2, 3)
    qkv = _lin(sd, p + ".qkv", x).reshape(B, _h * _w, ws * ws, 3, heads, C // heads).permute(3, 0, 1, 4, 2, 5)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)) * ((C // heads) ** -0.5)
    attn = attn.softmax(dim=-1)
    attn = (attn @ v).transpose(2, 3).reshape(B, _h, _w, ws, ws, C)
    x = attn.transpose(2, 3).reshape(B, _h * ws, _w * ws, C)
    if pad_r > 0 or pad_b > 0:
        x = x[:, :H, :W, :].contiguous()
    return _lin(sd, p + ".proj", x.reshape(B, N, C))


def _gsa(sd, p, x, size, heads, sr):
    """GlobalSubSampleAttn, twins.py:870-925."""
    B, N, C = x.shape
    q = _lin(sd, p + ".q", x).reshape(B, N, heads, C // heads).permute(0, 2, 1, 3)
    x = x.permute(0, 2, 1).reshape(B, C, *size)
    x = _conv(sd, p + ".sr", x, stride=sr).reshape(B, C, -1).permute(0, 2, 1)
    x = _ln(sd, p + ".norm", x)
    kv = _lin(sd, p + ".kv", x).reshape(B, -1, 2, heads, C // heads).permute(2, 0, 3, 1, 4)
    k, v = kv[0], kv[1]
    attn = (q @ k.transpose(-2, -1)) * ((C // heads) ** -0.5)
    attn = attn.softmax(dim=-1)
    x = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return _lin(sd, p + ".proj", x)


TWINS_STAGES = ((4, 4, 8), (2, 8, 4))  # (patch, heads, sr_ratio); embed dims 128 / 256; twins.py:1344-1348


def twins_svt_large(sd, p, x):
    """encoders.py:7-48 (return_feat=True) over timm's twins_svt_large truncated to two stages.
    Block norms use eps 1e-6 (twins.py:1169), PatchEmbed / GSA norms the LayerNorm default 1e-5."""
    B = x.shape[0]
    feat = []
    for i, (patch, heads, sr) in enumerate(TWINS_STAGES):
        H, W = x.shape[-2:]
        # PatchEmbed, twins.py:1122-1149
        x = _conv(sd, f"{p}.svt.patch_embeds.{i}.proj", x, stride=patch).flatten(2).transpose(1, 2)
        x = _ln(sd, f"{p}.svt.patch_embeds.{i}.norm", x)
        size = (H // patch, W // patch)
        for j in range(2):
            bp = f"{p}.svt.blocks.{i}.{j}"
            # Block (timm: no context argument), twins.py:1094-1097
            y = _ln(sd, bp + ".norm1", x, 1e-6)
            y = _lsa(sd, bp + ".attn", y, size, heads) if j == 0 else _gsa(sd, bp + ".attn", y, size, heads, sr)
            x = x + y
            x = x + _mlp(sd, bp + ".mlp", _ln(sd, bp + ".norm2", x, 1e-6))
            if j == 0:
                # PosConv (PEG), twins.py:1100-1119
                C = x.shape[-1]
                tok = x.transpose(1, 2).view(B, C, *size)
                y = _conv(sd, f"{p}.svt.pos_block.{i}.proj.0", tok, padding=1, groups=C)
                y += tok
                x = y.flatten(2).transpose(1, 2)
        x = x.reshape(B, *size, -1).permute(0, 3, 1, 2).contiguous()
        feat.append(x)
    return x, feat


# --------------------------------------------------------------------------- cost-volume encoder (a26)
def _mha(q, k, v, heads):
    """MultiHeadAttention, LatentCostFormer/attention.py:39-66."""
    scale = (q.shape[-1] / heads) ** -0.5
    B, HW, _ = q.shape
    Q = rearrange(q, "b i (heads d) -> b heads i d", heads=heads)
    K = rearrange(k, "b j (heads d) -> b heads j d", heads=heads)
    dots = einsum("bhid, bhjd -> bhij", Q, K) * scale
    attn = dots.softmax(dim=-1)
    V = rearrange(v, "b j (heads d) -> b heads j d", heads=heads)
    out = einsum("bhij, bhjd -> bhid", attn, V)
    return rearrange(out, "b heads hw d -> b hw (heads d)", b=B, hw=HW)


def _broad_mha(q, k, v, heads):
    """BroadMultiHeadAttention, LatentCostFormer/attention.py:10-36 (one shared query set)."""
    scale = (q.shape[-1] / heads) ** -0.5
    B = k.shape[0]
    N = q.shape[1]
    Q = rearrange(q.squeeze(), "i (heads d) -> heads i d", heads=heads)
    K = rearrange(k, "b j (heads d) -> b heads j d", heads=heads)
    dots = einsum("hid, bhjd -> bhij", Q, K) * scale
    attn = dots.softmax(dim=-1)
    V = rearrange(v, "b j (heads d) -> b heads j d", heads=heads)
    out = einsum("bhij, bhjd -> bhid", attn, V)
    return rearrange(out, "b heads n d -> b n (heads d)", b=B, n=N)


def cost_patch_embed(sd, p, x, patch=8, dim=64):
    """PatchEmbed of the cost maps, LatentCostFormer/encoder.py:30-96."""
    B, C, H, W = x.shape
    x = F.pad(x, (0, (patch - W % patch) % patch, 0, (patch - H % patch) % patch))
    x = F.relu(_conv(sd, p + ".proj.0", x, stride=2, padding=2))
    x = F.relu(_conv(sd, p + ".proj.2", x, stride=2, padding=2))
    x = _conv(sd, p + ".proj.4", x, stride=2, padding=2)
    oh, ow = x.shape[2:]
    pc = coords_grid(B, oh, ow) * patch + patch / 2
    pc = pc.view(B, 2, -1).permute(0, 2, 1)
    enc = linear_pos_embedding_sine(pc, dim=dim).permute(0, 2, 1).view(B, -1, oh, ow)
    x = torch.cat([x, enc], dim=1)
    x = _conv(sd, p + ".ffn_with_coord.2", F.relu(_conv(sd, p + ".ffn_with_coord.0", x)))
    return _ln(sd, p + ".norm", x.flatten(2).transpose(1, 2)), (oh, ow)


def _ffn(sd, p, x):
    return _lin(sd, p + ".3", F.gelu(_lin(sd, p + ".0", x)))


def _context_tokens(sd, p, context, B, H, W):
    # twins.py:366-369 / 465-468
    c = context.repeat(B // context.shape[0], 1, 1, 1)
    c = c.view(B, -1, H * W).permute(0, 2, 1)
    return _lin(sd, p + ".context_proj", c).view(B, H, W, -1)


def _lsa_rpe_ctx(sd, p, x, size, context, heads=8, ws=7, vert_c=64):
    """LocallyGroupedAttnRPEContext, twins.py:331-427."""
    B, N, C = x.shape
    H, W = size
    Cqk = C + vert_c
    ctx = _context_tokens(sd, p, context, B, H, W)
    x = x.view(B, H, W, C)
    x_qk = torch.cat([x, ctx], dim=-1)
    pad_r = (ws - W % ws) % ws
    pad_b = (ws - H % ws) % ws
    x = F.pad(x, (0, 0, 0, pad_r, 0, pad_b))
    x_qk = F.pad(x_qk, (0, 0, 0, pad_r, 0, pad_b))
    _, Hp, Wp, _ = x.shape
    _h, _w = Hp // ws, Wp // ws
    x = x.reshape(B, _h, ws, _w, ws, C).transpose(2, 3)
    x_qk = x_qk.reshape(B, _h, ws, _w, ws, Cqk).transpose(2, 3)
    hd = C // heads
    v = _lin(sd, p + ".v", x).reshape(B, _h * _w, ws * ws, 1, heads, hd).permute(3, 0, 1, 4, 2, 5)[0]
    coords = coords_grid(B, ws, ws).view(B, 2, -1).permute(0, 2, 1)
    enc = linear_pos_embedding_sine(coords, dim=Cqk).view(B, ws, ws, Cqk)
    x_qk = x_qk + enc[:, None, None, :, :, :]
    q = _lin(sd, p + ".q", x_qk).reshape(B, _h * _w, ws * ws, 1, heads, hd).permute(3, 0, 1, 4, 2, 5)[0]
    k = _lin(sd, p + ".k", x_qk).reshape(B, _h * _w, ws * ws, 1, heads, hd).permute(3, 0, 1, 4, 2, 5)[0]
    attn = (q @ k.transpose(-2, -1)) * (hd**-0.5)
    attn = attn.softmax(dim=-1)
    attn = (attn @ v).transpose(2, 3).reshape(B, _h, _w, ws, ws, C)
    x = attn.transpose(2, 3).reshape(B, _h * ws, _w * ws, C)
    if pad_r > 0 or pad_b > 0:
        x = x[:, :H, :W, :].contiguous()
    return _lin(sd, p + ".proj", x.reshape(B, N, C))


def _gsa_rpe_ctx(sd, p, x, size, context, heads=8, sr=4, vert_c=64):
    """GlobalSubSampleAttnRPEContext, twins.py:430-546."""
    B, N, C = x.shape
    H, W = size
    Cqk = C + vert_c
    hd = C // heads
    ctx = _context_tokens(sd, p, context, B, H, W)
    x = x.view(B, H, W, C)
    x_qk = torch.cat([x, ctx], dim=-1)
    pad_r = (sr - W % sr) % sr
    pad_b = (sr - H % sr) % sr
    x = F.pad(x, (0, 0, 0, pad_r, 0, pad_b))
    x_qk = F.pad(x_qk, (0, 0, 0, pad_r, 0, pad_b))
    _, Hp, Wp, _ = x.shape
    pN = Hp * Wp
    x = x.view(B, -1, C)
    x_qk = x_qk.view(B, -1, Cqk)
    coords = coords_grid(B, Hp, Wp).view(B, 2, -1).permute(0, 2, 1)
    enc = linear_pos_embedding_sine(coords, dim=Cqk)
    q = _lin(sd, p + ".q", x_qk + enc).reshape(B, pN, heads, hd).permute(0, 2, 1, 3)
    x = x.permute(0, 2, 1).reshape(B, C, Hp, Wp)
    x_qk = x_qk.permute(0, 2, 1).reshape(B, Cqk, Hp, Wp)
    x = _conv(sd, p + ".sr_value", x, stride=sr).reshape(B, C, -1).permute(0, 2, 1)
    x_qk = _conv(sd, p + ".sr_key", x_qk, stride=sr).reshape(B, C, -1).permute(0, 2, 1)
    x = _ln(sd, p + ".norm", x)
    x_qk = _ln(sd, p + ".norm", x_qk)
    coords = coords_grid(B, Hp // sr, Wp // sr).view(B, 2, -1).permute(0, 2, 1) * sr
    enc = linear_pos_embedding_sine(coords, dim=C)
    n_kv = (Hp // sr) * (Wp // sr)
    k = _lin(sd, p + ".k", x_qk + enc).reshape(B, n_kv, heads, hd).permute(0, 2, 1, 3)
    v = _lin(sd, p + ".v", x).reshape(B, n_kv, heads, hd).permute(0, 2, 1, 3)
    attn = (q @ k.transpose(-2, -1)) * (hd**-0.5)
    attn = attn.softmax(dim=-1)
    x = (attn @ v).transpose(1, 2).reshape(B, Hp, Wp, C)
    if pad_r > 0 or pad_b > 0:
        x = x[:, :H, :W, :].contiguous()
    return _lin(sd, p + ".proj", x.reshape(B, N, C))


def _vertical_block(sd, p, x, size, context, local):
    # Block with the default nn.LayerNorm (eps 1e-5), twins.py:1028-1097; encoder.py:149-204
    y = _ln(sd, p + ".norm1", x)
    y = _lsa_rpe_ctx(sd, p + ".attn", y, size, context) if local else _gsa_rpe_ctx(sd, p + ".attn", y, size, context)
    x = x + y
    return x + _mlp(sd, p + ".mlp", _ln(sd, p + ".norm2", x))


def cost_perceiver_encoder(sd, p, cost_volume, context, depth=3, K=8, taps=None, tag=""):
    """CostPerceiverEncoder.forward, LatentCostFormer/encoder.py:410-466."""
    B, heads, H1, W1, H2, W2 = cost_volume.shape
    cost_maps = cost_volume.permute(0, 2, 3, 1, 4, 5).contiguous().view(B * H1 * W1, 1, H2, W2)
    x, size = cost_patch_embed(sd, p + ".patch_embed", cost_maps)
    if taps is not None:
        taps[tag + "cost_tokens"] = x
    # input_layer: CrossAttentionLayer, encoder.py:282-346
    ip = p + ".input_layer"
    query = sd[p + ".latent_tokens"]
    short = query
    query = _ln(sd, ip + ".norm1", query)
    a = _broad_mha(_lin(sd, ip + ".q", query), _lin(sd, ip + ".k", x), _lin(sd, ip + ".v", x), 8)
    x = short + _lin(sd, ip + ".proj", a)
    x = x + _ffn(sd, ip + ".ffn", _ln(sd, ip + ".norm2", x))
    short_cut = x
    if taps is not None:
        taps[tag + "latent_in"] = x
    for idx in range(depth):
        # SelfAttentionLayer, encoder.py:214-279
        ep = f"{p}.encoder_layers.{idx}"
        sc = x
        y = _ln(sd, ep + ".norm1", x)
        y = _mha(_lin(sd, ep + ".q", y), _lin(sd, ep + ".k", y), _lin(sd, ep + ".v", y), 8)
        x = sc + _lin(sd, ep + ".proj", y)
        x = x + _ffn(sd, ep + ".ffn", _ln(sd, ep + ".norm2", x))
        x = x.view(B, H1 * W1, K, -1).permute(0, 2, 1, 3).reshape(B * K, H1 * W1, -1)
        vp = f"{p}.vertical_encoder_layers.{idx}"
        x = _vertical_block(sd, vp + ".local_block", x, (H1, W1), context, True)
        x = _vertical_block(sd, vp + ".global_block", x, (H1, W1), context, False)
        x = x.view(B, K, H1 * W1, -1).permute(0, 2, 1, 3).reshape(B * H1 * W1, K, -1)
        if taps is not None and idx == 0:
            taps[tag + "latent_l0"] = x
    return x + short_cut, cost_maps, size


def memory_encoder(sd, p, img1, img2, context, taps=None, tag=""):
    """MemoryEncoder.forward, LatentCostFormer/encoder.py:469-539 (cost_heads_num = 1, no 1/sqrt(d))."""
    feats, _ = twins_svt_large(sd, p + ".feat_encoder", torch.cat([img1, img2], dim=0))
    feats = _conv(sd, p + ".channel_convertor", feats)
    B = feats.shape[0] // 2
    feat_s, feat_t = feats[:B], feats[B:]
    _, _, H, W = feat_s.shape
    f1 = rearrange(feat_s, "b (heads d) h w -> b heads (h w) d", heads=1)
    f2 = rearrange(feat_t, "b (heads d) h w -> b heads (h w) d", heads=1)
    corr = einsum("bhid, bhjd -> bhij", f1, f2)
    corr = corr.permute(0, 2, 1, 3).view(B * H * W, 1, H, W)
    corr = corr.view(B, H * W, 1, H * W).permute(0, 2, 1, 3).view(B, 1, H, W, H, W)
    if taps is not None:
        taps[tag + "ffeat"] = feat_s
    mem, cost_maps, size = cost_perceiver_encoder(sd, p + ".cost_perceiver_encoder", corr, context, taps=taps, tag=tag)
    return mem, cost_maps, feat_s


# --------------------------------------------------------------------------- decoder (a27)
def encode_flow_token(cost_maps, coords, r=4):
    """MemoryDecoder.encode_flow_token, LatentCostFormer/decoder.py:237-255 (window axes as in RAFT's lookup)."""
    coords = coords.permute(0, 2, 3, 1)
    b, h1, w1, _ = coords.shape
    d = torch.linspace(-r, r, 2 * r + 1)
    delta = torch.stack(torch.meshgrid(d, d, indexing="ij"), dim=-1).view(1, 2 * r + 1, 2 * r + 1, 2)
    corr = bilinear_sampler(cost_maps, coords.reshape(b * h1 * w1, 1, 1, 2) + delta)
    return corr.view(b, h1, w1, -1).permute(0, 3, 1, 2)


def gma_attention(sd, p, fmap):
    """Attention.forward, LatentCostFormer/gma.py:32-76 (heads 1, dim_head 128; the pos_emb is unused)."""
    b, c, h, w = fmap.shape
    q, k = _conv(sd, p + ".to_qk", fmap).chunk(2, dim=1)
    q, k = (rearrange(t_, "b (h d) x y -> b h x y d", h=1) for t_ in (q, k))
    q = (128**-0.5) * q
    sim = einsum("b h x y d, b h u v d -> b h x y u v", q, k)
    sim = rearrange(sim, "b h x y u v -> b h (x y) (u v)")
    return sim.softmax(dim=-1)


def gma_aggregate(sd, p, attn, fmap):
    """Aggregate.forward, LatentCostFormer/gma.py:79-115 (dim == inner_dim: no project)."""
    b, c, h, w = fmap.shape
    v = rearrange(_conv(sd, p + ".to_v", fmap), "b (h d) x y -> b h (x y) d", h=1)
    out = einsum("b h i j, b h j d -> b h i d", attn, v)
    out = rearrange(out, "b h (x y) d -> b (h d) x y", x=h, y=w)
    return fmap + sd[p + ".gamma"] * out


def _sep_conv_gru(sd, p, h, x):
    # LatentCostFormer/gru.py:35-73
    for s, pad in (("1", (0, 2)), ("2", (2, 0))):
        hx = torch.cat([h, x], dim=1)
        z = torch.sigmoid(_conv(sd, p + ".convz" + s, hx, padding=pad))
        r = torch.sigmoid(_conv(sd, p + ".convr" + s, hx, padding=pad))
        q = torch.tanh(_conv(sd, p + ".convq" + s, torch.cat([r * h, x], dim=1), padding=pad))
        h = (1 - z) * h + z * q
    return h


def gma_update_block(sd, p, net, inp, corr, flow, attention, want_mask):
    """GMAUpdateBlock.forward, LatentCostFormer/gru.py:130-160 (+ BasicMotionEncoder :76-99, FlowHead :6-14)."""
    e = p + ".encoder"
    cor = F.relu(_conv(sd, e + ".convc1", corr))
    cor = F.relu(_conv(sd, e + ".convc2", cor, padding=1))
    flo = F.relu(_conv(sd, e + ".convf1", flow, padding=3))
    flo = F.relu(_conv(sd, e + ".convf2", flo, padding=1))
    out = F.relu(_conv(sd, e + ".conv", torch.cat([cor, flo], dim=1), padding=1))
    mf = torch.cat([out, flow], dim=1)
    mfg = gma_aggregate(sd, p + ".aggregator", attention, mf)
    net = _sep_conv_gru(sd, p + ".gru", net, torch.cat([inp, mf, mfg], dim=1))
    dflow = _conv(sd, p + ".flow_head.conv2", F.relu(_conv(sd, p + ".flow_head.conv1", net, padding=1)), padding=1)
    mask = None
    if want_mask:
        mask = 0.25 * _conv(sd, p + ".mask.2", F.relu(_conv(sd, p + ".mask.0", net, padding=1)))
    return net, mask, dflow


def memory_decoder(sd, p, cost_memory, context, cost_maps, iters=None, taps=None, tag=""):
    """MemoryDecoder.forward, LatentCostFormer/decoder.py:257-321.  Only the last iteration's mask head and
    convex upsampling are evaluated (the reference evaluates all and returns flow_predictions[-1])."""
    B, _, H1, W1 = context.shape
    coords0 = coords_grid(B, H1, W1)
    coords1 = coords_grid(B, H1, W1)
    ctx = _conv(sd, p + ".proj", context)
    net, inp = torch.split(ctx, [128, 128], dim=1)
    net = torch.tanh(net)
    inp = torch.relu(inp)
    attention = gma_attention(sd, p + ".att", inp)
    ca = p + ".decoder_layer.cross_attend"
    key = _lin(sd, ca + ".k", cost_memory)
    value = _lin(sd, ca + ".v", cost_memory)
    iters = 32 if iters is None else iters
    mask = None
    for it in range(iters):
        cost_forward = encode_flow_token(cost_maps, coords1)
        query = _conv(sd, p + ".flow_token_encoder.2", F.gelu(_conv(sd, p + ".flow_token_encoder.0", cost_forward)))
        query = query.permute(0, 2, 3, 1).contiguous().view(B * H1 * W1, 1, 64)
        # decoder CrossAttentionLayer, decoder.py:35-120
        qc = coords1.contiguous().view(B, 2, -1).permute(0, 2, 1)[:, :, None, :].contiguous().view(B * H1 * W1, 1, 2)
        enc = linear_pos_embedding_sine(qc, dim=64)
        short = query
        qn = _ln(sd, ca + ".norm1", query)
        x = _mha(_lin(sd, ca + ".q", qn + enc), key, value, 8)
        x = _lin(sd, ca + ".proj", torch.cat([x, short], dim=2))
        x = short + x
        x = x + _ffn(sd, ca + ".ffn", _ln(sd, ca + ".norm2", x))
        cost_global = x.view(B, H1, W1, 64).permute(0, 3, 1, 2)
        corr = torch.cat([cost_global, cost_forward], dim=1)
        flow = coords1 - coords0
        net, mask, dflow = gma_update_block(sd, p + ".update_block", net, inp, corr, flow, attention, it == iters - 1)
        coords1 = coords1 + dflow
        if taps is not None and it in (0, iters - 1):
            taps[f"{tag}cost_fwd_it{it}"] = cost_forward
            taps[f"{tag}cost_global_it{it}"] = cost_global
            taps[f"{tag}lowflow_it{it}"] = coords1 - coords0
            taps[f"{tag}net_it{it}"] = net
    flow_up = R.convex_upsample(coords1 - coords0, mask)
    return flow_up, coords1 - coords0


def flowformer_forward(sd, p, image1, image2, iters=None, taps=None, tag=""):
    """FlowFormer.forward(return_feat=True), LatentCostFormer/transformer.py:45-74 -> (flow_up, cfeat, ffeat)."""
    image1 = 2 * (image1 / 255.0) - 1.0
    image2 = 2 * (image2 / 255.0) - 1.0
    context, cfeat = twins_svt_large(sd, p + ".context_encoder", image1)
    mem, cost_maps, ffeat = memory_encoder(sd, p + ".memory_encoder", image1, image2, context, taps, tag)
    if taps is not None:
        taps[tag + "context"] = context
        taps[tag + "cfeat4"] = cfeat[0]
        taps[tag + "cost_memory"] = mem
    flow_up, _ = memory_decoder(sd, p + ".memory_decoder", mem, context, cost_maps, iters, taps, tag)
    return flow_up, cfeat, ffeat


sample_coord_input = R.sample_coord_input
psnr = R.psnr


def forward(sd, img_xs, coord, t, ds_factor=None, iters=None, taps=None):
    """GIMMVFI_F.forward, gimmvfi_f.py:304-384 (coord[i][1] is None); flow part = cal_bidirection_flow :114-139."""
    assert isinstance(t, list) and isinstance(coord, list) and len(t) == len(coord)
    full = None
    if ds_factor is not None:
        full = img_xs.clone()
        img_xs = torch.stack([R.resize(img_xs[:, :, 0], ds_factor), R.resize(img_xs[:, :, 1], ds_factor)], 2)
    im0, im1 = 255 * img_xs[:, :, 0], 255 * img_xs[:, :, 1]
    p = "flow_estimator"
    f01, feats0, fnet0 = flowformer_forward(sd, p, im0, im1, iters, taps, "f01_")
    f10, feats1, fnet1 = flowformer_forward(sd, p, im1, im0, iters, taps, "f10_")
    corr_fn = R.BidirCorr(fnet0, fnet1)
    return R.forward_after_flow(sd, img_xs, full, f01, f10, feats0, feats1, corr_fn, coord, t, taps)
