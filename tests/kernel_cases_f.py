"""Per-kernel parity cases of the FlowFormer glue kernels (csrc/flowformer_ops.hip), shared by the CPU emulator tests
and the GPU tests: each drives the C ABI through ``Runtime`` and compares with the corresponding function of the
oracle (oracle/gimmvfi_f_oracle.py) or a plain torch statement."""
import torch
import torch.nn.functional as F

import gimmvfi_f_oracle as forc
from gimmvfi_hip import lib as L
from gimmvfi_hip.ops import TokenChain, View
from kernel_cases import tol


def _r(rt, x):
    return x.to(rt.tdtype).float()


def _g(seed=0):
    return torch.Generator().manual_seed(seed)


def layernorm_case(rt, rows=37, C=128, x_f32=False, eps=1e-6):
    g = _g(1)
    x = torch.randn(rows, C, generator=g) * 2 + 0.3
    gam, bet = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    xd = (x if x_f32 else x.to(rt.tdtype)).to(rt.device)
    out = rt.layernorm(xd, (gam.to(rt.device), bet.to(rt.device)), eps)
    ref = F.layer_norm(xd.float().cpu(), (C,), gam, bet, eps)
    assert out.dtype == rt.tdtype
    assert float((out.float().cpu() - ref).abs().max()) <= tol(rt, float(ref.abs().max()))


def dwconv_case(rt, N=2, H=7, W=9, C=24, f32=False):
    g = _g(2)
    x = torch.randn(N, H, W, C, generator=g)
    w = torch.randn(C, 1, 3, 3, generator=g) * 0.3
    b = torch.randn(C, generator=g) * 0.1
    xd = (x if f32 else x.to(rt.tdtype)).to(rt.device)
    out = rt.dwconv3x3_res(xd, w.reshape(C, 9).t().contiguous().to(rt.device), b.to(rt.device))
    xi = xd.float().cpu().permute(0, 3, 1, 2)
    ref = (F.conv2d(xi, w, b, padding=1, groups=C) + xi).permute(0, 2, 3, 1)
    assert float((out.float().cpu() - ref).abs().max()) <= tol(rt, float(ref.abs().max()))


def pos_embed_case(rt, dim=64):
    g = _g(3)
    coords = torch.rand(11, 2, generator=g) * 50
    rows = 33
    base = _r(rt, torch.randn(rows, dim + 8, generator=g))
    out = base.clone().to(rt.tdtype).to(rt.device)
    rt.pos_embed(coords.to(rt.device), 11, 8.0, 4.0, dim, View(out, 8, dim), rows, True)
    enc = forc.linear_pos_embedding_sine((coords * 8.0 + 4.0)[None], dim=dim)[0]
    ref = base.clone()
    ref[:, 8:] += enc.repeat(3, 1)
    assert float((out.float().cpu() - ref).abs().max()) <= tol(rt, 4.0)
    out2 = torch.zeros(rows, dim, dtype=rt.tdtype, device=rt.device)
    rt.pos_embed(coords.to(rt.device), 11, 1.0, 0.0, dim, out2, rows, False)
    ref2 = forc.linear_pos_embedding_sine(coords[None], dim=dim)[0].repeat(3, 1)
    assert float((out2.float().cpu() - ref2).abs().max()) <= tol(rt, 1.0)


def cost_embed_lookup_case(rt, maps=6, h=5, w=7):
    """First cost-map convolution (zero extension to a multiple of 8) and the 81-tap lookup."""
    g = _g(4)
    vol = torch.randn(maps, h, w, generator=g) * 3
    wt = torch.randn(16, 1, 6, 6, generator=g) * 0.2
    b = torch.randn(16, generator=g) * 0.1
    hp, wp = (h + 7) // 8 * 8, (w + 7) // 8 * 8
    out = rt.cost_embed1(vol.to(rt.device), wt.reshape(16, 36).t().contiguous().to(rt.device), b.to(rt.device), maps, h, w,
                         hp // 2, wp // 2)
    ref = F.relu(F.conv2d(F.pad(vol[:, None], (0, wp - w, 0, hp - h)), wt, b, stride=2, padding=2)).permute(0, 2, 3, 1)
    assert float((out.float().cpu()[..., :16] - ref).abs().max()) <= tol(rt, float(ref.abs().max()))
    # lookup: one cost map per query (maps == B*h*w with B = 1 would need h*w maps; use h*w maps)
    Q = h * w
    vol2 = torch.randn(Q, h, w, generator=g)
    coords = torch.stack([torch.rand(Q, generator=g) * (w + 4) - 2, torch.rand(Q, generator=g) * (h + 4) - 2], -1)
    o = torch.zeros(Q, 88, dtype=rt.tdtype, device=rt.device)
    rt.cost_lookup(vol2.to(rt.device), coords.to(rt.device), View(o, 0, 81), Q, h, w)
    cref = forc.encode_flow_token(vol2[:, None], coords.t().reshape(1, 2, h, w))    # (1, 81, h, w)
    got = o.float().cpu()[:, :81].reshape(h, w, 81).permute(2, 0, 1)[None]
    assert float((got - cref).abs().max()) <= tol(rt, float(cref.abs().max()))
    assert float(o.float().cpu()[:, 81:].abs().max()) == 0.0


def attn_window_case(rt, B=2, H=9, W=10, C=64, heads=4):
    """Against the oracle's LocallyGroupedAttn (twins.py:814-867) without the output projection: ragged grid so the
    padded window positions (key = value = bias) take part."""
    g = _g(5)
    x = _r(rt, torch.randn(B, H * W, C, generator=g))
    sd = {"a.qkv.weight": _r(rt, torch.randn(3 * C, C, generator=g) / C ** 0.5), "a.qkv.bias": torch.randn(3 * C, generator=g) * 0.3,
          "a.proj.weight": torch.eye(C), "a.proj.bias": torch.zeros(C)}
    ref = forc._lsa(sd, "a", x, (H, W), heads)
    qkv = _r(rt, F.linear(x, sd["a.qkv.weight"], sd["a.qkv.bias"])).reshape(B * H * W, 3 * C)
    qd = qkv.to(rt.tdtype).to(rt.device)
    out = torch.empty(B * H * W, C, dtype=rt.tdtype, device=rt.device)
    kpad = sd["a.qkv.bias"][C:2 * C].repeat(49, 1).contiguous().to(rt.device)
    vpad = sd["a.qkv.bias"][2 * C:].repeat(49, 1).contiguous().to(rt.device)
    rt.attn_window(View(qd, 0, C), View(qd, C, C), View(qd, 2 * C, C), kpad, vpad, out, B, H, W, 7, heads, C // heads)
    assert float((out.float().cpu().reshape(B, H * W, C) - ref).abs().max()) <= 2 * tol(rt, float(ref.abs().max()) + 1.0)


def attn_global_mfma_case(rt, hd=16):
    """Shapes that take the MFMA kernel in bf16 (attn_mfma.hip): M = 37 / 112 / 128 keys, ragged query blocks, strided groups."""
    g = _g(16)
    heads = 4
    C = heads * hd
    dev = lambda t: t.reshape(-1, t.shape[-1]).to(rt.tdtype).to(rt.device)
    # (the smallest sizes the MFMA kernel takes, a partial fourth key block, ragged query blocks)
    for (B, N, M) in ((2, 70, 37), (1, 97, 112), (2, 33, 128), (1, 16, 9), (3, 200, 65)):
        q, k, v = (_r(rt, torch.randn(B, n_, C, generator=g)) for n_ in (N, M, M))
        ref = forc._mha(q, k, v, heads)
        # q inside a wider row matrix at a channel offset, k | v in one matrix (as the engine passes them)
        qw = torch.cat([torch.zeros(B, N, 8), q, torch.zeros(B, N, 8)], -1)
        kv = torch.cat([k, v], -1)
        qd, kvd = dev(qw), dev(kv)
        out = torch.zeros(B * N, C + 8, dtype=rt.tdtype, device=rt.device)
        rt.attn_global(View(qd, 8, C), (N, 0, 1), View(kvd, 0, C), View(kvd, C, C), (M, 0, 1), View(out, 0, C), (N, 0, 1), B, 1, N, M, heads, hd)
        got = out.float().cpu()
        assert float(got[:, C:].abs().max()) == 0.0
        err = float((got[:, :C].reshape(B, N, C) - ref).abs().max())
        assert err <= 2 * tol(rt, float(ref.abs().max()) + 1.0), (B, N, M, err)
    # strided groups (G0 > 1): one shared query set against per-map keys, image-major output rows
    n, P, K, T = 2, 5, 20, 12
    lat = _r(rt, torch.randn(1, K, C, generator=g))
    kk, vv = (_r(rt, torch.randn(n * P, T, C, generator=g)) for _ in range(2))
    refb = forc._broad_mha(lat, kk, vv, heads)
    outb = torch.zeros(n * K * P, C, dtype=rt.tdtype, device=rt.device)
    rt.attn_global(dev(lat), (0, 0, 1), dev(kk), dev(vv), (P * T, T, 1), outb, (K * P, 1, P), n, P, K, T, heads, hd)
    gotb = outb.float().cpu().reshape(n, K, P, C).permute(0, 2, 1, 3).reshape(n * P, K, C)
    assert float((gotb - refb).abs().max()) <= 2 * tol(rt, float(refb.abs().max()) + 1.0)


def attn_global_case(rt):
    """(a) batched global attention; (b) one shared query set against per-map keys with the strided image-major
    output; (c) self-attention over the K tokens of a map in the image-major layout."""
    g = _g(6)
    heads, hd = 8, 16
    C = heads * hd
    B, N, M = 2, 13, 5
    q, k, v = (_r(rt, torch.randn(B, n_, C, generator=g)) for n_ in (N, M, M))
    ref = forc._mha(q, k, v, heads)
    dev = lambda t: t.reshape(-1, C).to(rt.tdtype).to(rt.device)
    out = torch.empty(B * N, C, dtype=rt.tdtype, device=rt.device)
    rt.attn_global(dev(q), (N, 0, 1), dev(k), dev(v), (M, 0, 1), out, (N, 0, 1), B, 1, N, M, heads, hd)
    assert float((out.float().cpu().reshape(B, N, C) - ref).abs().max()) <= 2 * tol(rt, float(ref.abs().max()) + 1.0)
    # (b) BroadMultiHeadAttention: latent queries [K, C] vs T tokens of each of n*P maps
    n, P, K, T = 2, 6, 8, 5
    lat = _r(rt, torch.randn(1, K, C, generator=g))
    kk, vv = (_r(rt, torch.randn(n * P, T, C, generator=g)) for _ in range(2))
    refb = forc._broad_mha(lat, kk, vv, heads)                              # (n*P, K, C)
    outb = torch.zeros(n * K * P, C, dtype=rt.tdtype, device=rt.device)
    rt.attn_global(dev(lat), (0, 0, 1), dev(kk), dev(vv), (P * T, T, 1), outb, (K * P, 1, P), n, P, K, T, heads, hd)
    gotb = outb.float().cpu().reshape(n, K, P, C).permute(0, 2, 1, 3).reshape(n * P, K, C)
    assert float((gotb - refb).abs().max()) <= 2 * tol(rt, float(refb.abs().max()) + 1.0)
    # (c) self-attention over the K tokens, image-major rows (b, k, p)
    x = _r(rt, torch.randn(n * P, K, 3 * C, generator=g))
    refc = forc._mha(x[..., :C], x[..., C:2 * C], x[..., 2 * C:], heads)
    xim = x.reshape(n, P, K, 3 * C).permute(0, 2, 1, 3).reshape(n * K * P, 3 * C).to(rt.tdtype).to(rt.device)
    outc = torch.zeros(n * K * P, C, dtype=rt.tdtype, device=rt.device)
    lay = (K * P, 1, P)
    rt.attn_global(View(xim, 0, C), lay, View(xim, C, C), View(xim, 2 * C, C), lay, outc, lay, n, P, K, K, heads, hd)
    gotc = outc.float().cpu().reshape(n, K, P, C).permute(0, 2, 1, 3).reshape(n * P, K, C)
    assert float((gotc - refc).abs().max()) <= 2 * tol(rt, float(refc.abs().max()) + 1.0)


def xqk_case(rt, nb=2, K=3, H=5, W=6):
    """[x | context] with the reference's context.repeat() tiling (twins.py:366) and both positional codes."""
    g = _g(7)
    n = 2 * nb                       # two directions
    P = H * W
    x = _r(rt, torch.randn(n * K, P, 128, generator=g))
    ctx = _r(rt, torch.randn(n, P, 64, generator=g))
    for mode, use_table in ((0, False), (1, False), (2, False), (1, True), (2, True)):
        out = torch.zeros(n * K * P, 192, dtype=rt.tdtype, device=rt.device)
        # with the table: the vector kernel + gvfi_ff_pos_table (what the engine runs); without: per-element evaluation
        table = rt.ff_pos_table(H, W, 192, mode) if use_table else None
        rt.ff_xqk(x.reshape(-1, 128).to(rt.tdtype).to(rt.device), ctx.reshape(-1, 64).to(rt.tdtype).to(rt.device), out,
                  n * K, H, W, K, nb, mode, 7, table=table)
        refs = []
        for d in range(2):           # per direction: x batch (nb*K), context batch nb tiled by repeat
            xd = x[d * nb * K:(d + 1) * nb * K]
            cd = ctx[d * nb:(d + 1) * nb].repeat(K, 1, 1)     # == context.repeat(B // nb, 1, 1, 1) on (nb, P, 64)
            refs.append(torch.cat([xd, cd], -1))
        ref = torch.cat(refs, 0)
        if mode:
            ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
            cs = torch.stack([xs, ys], -1).reshape(P, 2).float()
            if mode == 1:
                cs = cs % 7
            ref = ref + forc.linear_pos_embedding_sine(cs[None], dim=192)
        assert float((out.float().cpu().reshape(n * K, P, 192) - ref).abs().max()) <= tol(rt, float(ref.abs().max()))


def tile_softmax_case(rt):
    g = _g(8)
    K, P, C, n = 3, 5, 16, 2
    table = torch.randn(K, C, generator=g)
    out = torch.zeros(n * K * P, C, dtype=torch.float32, device=rt.device)
    rt.tile_rows(table.to(rt.device), out, n * K * P, P, K, C)
    ref = table[None, :, None, :].expand(n, K, P, C).reshape(-1, C)
    assert float((out.cpu() - ref).abs().max()) == 0.0
    rows, ncol, ld = 70, 40, 48
    x = torch.randn(rows, ncol, generator=g) * 3
    y = torch.full((rows, ld), 7.0, dtype=rt.tdtype, device=rt.device)
    rt.softmax_rows(x.to(rt.device), ncol, y, rows)
    got = y.float().cpu()
    assert float((got[:, :ncol] - x.softmax(-1)).abs().max()) <= tol(rt, 1.0)
    assert float(got[:, ncol:].abs().max()) == 0.0


def token_chain_case(rt, rows=75):
    """gvfi_token_chain (csrc/token_chain.hip) against the unfused arithmetic in torch, rounding to the activation type where
    the separate launches store a tensor: (A) flow-token encoder + norm1 + position code + q, (C) proj + residual, norm2, ffn
    + residual (decoder.py:84-120, 237-255).  Ragged last wave (rows % 32 != 0), inputs / outputs that are channel slices
    of wider tensors."""
    if rt.precision == "fp32":
        return      # 16-bit operand types only (the float engine keeps the separate launches)
    g = _g(7)
    dev = rt.device
    rd = lambda t: _r(rt, t)
    W0, W1, W2 = rd(torch.randn(64, 81, generator=g) / 9), rd(torch.randn(64, 64, generator=g) / 8), rd(torch.randn(64, 64, generator=g) / 8)
    Wp = rd(torch.randn(64, 128, generator=g) / 11)
    b = [torch.randn(64, generator=g) * 0.1 for _ in range(3)]
    gam, bet = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    x = torch.zeros(rows, 192)
    x[:, 64:145] = torch.randn(rows, 81, generator=g)
    xa = x.to(rt.tdtype).to(dev)
    coords = (torch.rand(rows, 2, generator=g) * 40).contiguous()
    # ---- (A)
    ch = TokenChain(rt, [W0, W1, W2], b, (gam, bet), 1e-5, 1, act0=L.ACT_GELU)
    q = torch.full((rows, 72), 3.0, dtype=rt.tdtype, device=dev)
    query = torch.zeros(rows, 64, dtype=rt.tdtype, device=dev)
    rt.token_chain(ch, View(xa, 64, 128), View(q, 8, 64), out1=query, coords=coords.to(dev), period=rows)
    xin = xa.float().cpu()[:, 64:192]
    W0p = torch.zeros(64, 128)
    W0p[:, :81] = W0
    t1 = rd(F.gelu(xin @ W0p.t() + b[0]))
    qr = rd(t1 @ W1.t() + b[1])
    qn = rd(F.layer_norm(qr, (64,), gam, bet, 1e-5))
    c = torch.arange(64)
    part, f = c // 16, (c % 16).float()
    ang = 3.14 * torch.where(part[None] < 2, coords[:, 0:1], coords[:, 1:2]) * f[None] / 200.0
    qn = rd(qn + torch.where((part % 2 == 1)[None], torch.cos(ang), torch.sin(ang)))
    qq = rd(qn @ W2.t() + b[2])
    got_q = q.float().cpu()
    assert float((got_q[:, :8] - 3.0).abs().max()) == 0.0          # nothing outside the slice is written
    t = tol(rt, float(qq.abs().max()))
    assert float((query.float().cpu() - qr).abs().max()) <= t
    assert float((got_q[:, 8:] - qq).abs().max()) <= 2 * t, float((got_q[:, 8:] - qq).abs().max())
    # ---- (C)
    ch2 = TokenChain(rt, [Wp, W1, W2], b, (gam, bet), 1e-5, 0, act1=L.ACT_GELU, res2_from0=True)
    a_ = rd(torch.randn(rows, 64, generator=g)).to(rt.tdtype).to(dev)
    out = torch.full((rows, 192), 5.0, dtype=rt.tdtype, device=dev)
    rt.token_chain(ch2, a_, View(out, 0, 64), in1=query, res0=query)
    qy = query.float().cpu()
    xx = rd(torch.cat([a_.float().cpu(), qy], 1) @ Wp.t() + b[0] + qy)
    y = rd(F.layer_norm(xx, (64,), gam, bet, 1e-5))
    ff = rd(F.gelu(y @ W1.t() + b[1]))
    ref = rd(ff @ W2.t() + b[2] + xx)
    go = out.float().cpu()
    assert float((go[:, 64:] - 5.0).abs().max()) == 0.0
    assert float((go[:, :64] - ref).abs().max()) <= 2 * tol(rt, float(ref.abs().max())), float((go[:, :64] - ref).abs().max())
