"""Per-kernel parity cases shared by the CPU emulator tests and the GPU tests.

Every case drives the C ABI through ``gimmvfi_hip.ops.Runtime`` (the emulator runtime on CPU, the
real libgimmvfi_hip.so on the GPU) and compares with an independent torch / oracle statement.
"""
import ctypes as C

import os

import torch
import torch.nn.functional as F

import gimmvfi_r_oracle as orc
from gimmvfi_hip import lib as L
from gimmvfi_hip.ops import ConvLayer, InrMlp, PatchConvLayer, S2DConvLayer, TapSplitConvLayer, View


def _dev(rt):
    return rt.device


def _to_act(rt, x_nchw, pad_to=None):
    """NCHW float -> NHWC runtime dtype with padded pitch."""
    n, c, h, w = x_nchw.shape
    t = rt.act(n, h, w, c if pad_to is None else pad_to, zero=True)
    t[..., :c] = x_nchw.permute(0, 2, 3, 1).to(t.dtype)
    return t


def _rounded(rt, x):
    return x.to(rt.tdtype).float()


def tol(rt, scale):
    return {"fp32": 2e-5, "bf16": 1.2e-2, "fp16": 1.6e-3}[rt.precision] * scale


def conv_case(rt, N, H, W, Cin, Cout, KH, KW, stride=1, reflect=False, act1=L.ACT_NONE, with_res=False,
              act2=L.ACT_NONE, out_f32=False, split=None, seed=0, out_scale=1.0, tile=0, algo=0, pad=None, pad16=False,
              coff=2, res_f32=False):
    """coff: channel offset of the output slice inside a wider tensor (2 = unaligned: vector store paths fall back;
    0 or 8 = 16-byte aligned rows: the slim store loops run; channels outside the slice must stay untouched)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, KH, KW, generator=g) / (Cin * KH * KW) ** 0.5
    b = torch.randn(Cout, generator=g)
    slope = torch.rand(Cout, generator=g) * 0.3 + 0.1
    x, w = _rounded(rt, x), _rounded(rt, w)
    dev = _dev(rt)
    lay = ConvLayer(rt, w, b, stride=stride, pad=None if pad is None else (pad, pad),
                    pad_mode=L.PAD_REFLECT if reflect else L.PAD_ZEROS, slope=slope, wdir=(algo & 15) == 6)
    if (algo & 15) == 6:
        assert lay.w_frag is not None, "fragment-ordered weight image not packed"
    if split is None:
        xa = _to_act(rt, x).to(dev)
        x0, x1 = View(xa, 0, Cin), None
    else:  # two sources: channels [0,split) from a wider buffer at an offset, the rest from a second tensor
        wide = rt.act(N, H, W, split + 2 * rt.VE, zero=True)
        wide[..., rt.VE:rt.VE + split] = x[:, :split].permute(0, 2, 3, 1).to(wide.dtype)
        xb = _to_act(rt, x[:, split:])
        wide, xb = wide.to(dev), xb.to(dev)
        x0, x1 = View(wide, rt.VE, split), View(xb, 0, Cin - split)
    ph, pw = (KH // 2, KW // 2) if pad is None else (pad, pad)
    Ho, Wo = (H + 2 * ph - KH) // stride + 1, (W + 2 * pw - KW) // stride + 1
    res = None
    r = None
    if with_res:
        r = torch.randn(N, Cout, Ho, Wo, generator=g)
        if res_f32:      # float residual stream: not rounded
            res = r.permute(0, 2, 3, 1).contiguous().to(dev)
        else:
            r = _rounded(rt, r)
            res = _to_act(rt, r).to(dev)
    if pad16:
        # the destination is a tensor of its own whose pad channels (up to the next 16-byte unit) belong to this call:
        # they are pre-filled with garbage and must come back as zeros; nothing beyond them may be touched
        unit = 4 if out_f32 else rt.VE if rt.precision == "bf16" else 4
        cpad = (Cout + unit - 1) // unit * unit
        out = (rt.f32(N, Ho, Wo, cpad + unit) if out_f32 else rt.act(N, Ho, Wo, cpad + unit, zero=False, pitch=cpad + unit))
        out.fill_(7.0)
        if res is not None:
            ru = 4 if res.dtype == torch.float32 else 8
            rp = torch.full((N, Ho, Wo, (Cout + ru - 1) // ru * ru), float("nan"), dtype=res.dtype, device=res.device)
            rp[..., :Cout] = res[..., :Cout]
            res = rp
        rt.conv(lay, x0, View(out, 0, Cout), x1=x1, act1=act1, res=None if res is None else View(res, 0, Cout), act2=act2,
                slope2=lay.slope if act2 == L.ACT_PRELU else None, out_scale=out_scale, tile=tile, algo=algo, pad16=True)
        got = out.float().cpu()
        assert float(got[..., Cout:cpad].abs().max()) == 0.0 if cpad > Cout else True
        assert float((got[..., cpad:] - 7.0).abs().max()) == 0.0
        out = torch.cat([torch.zeros(N, Ho, Wo, 2), got[..., :Cout], torch.zeros(N, Ho, Wo, 1)], -1)
    else:
        wide = Cout + coff + (1 if coff == 2 else 8 + (-Cout) % 8)
        out = (rt.f32(N, Ho, Wo, wide, zero=True) if out_f32 else rt.act(N, Ho, Wo, wide, zero=True, pitch=wide))
        out.fill_(3.0)
        rt.conv(lay, x0, View(out, coff, Cout), x1=x1, act1=act1, res=None if res is None else View(res, 0, Cout), act2=act2,
                slope2=lay.slope if act2 == L.ACT_PRELU else None, out_scale=out_scale, tile=tile, algo=algo)
        got = out.float().cpu()
        assert float((got[..., :coff] - 3.0).abs().max()) == 0.0 if coff else True
        assert float((got[..., coff + Cout:] - 3.0).abs().max()) == 0.0        # nothing beyond the slice is written
        out = torch.cat([torch.zeros(N, Ho, Wo, 2), got[..., coff:coff + Cout], torch.zeros(N, Ho, Wo, 1)], -1)
    xi = F.pad(x, (pw, pw, ph, ph), mode="reflect") if reflect else x
    ref = F.conv2d(xi, w, b, stride=stride, padding=0 if reflect else (ph, pw))

    def act(v, k):
        if k == L.ACT_RELU:
            return F.relu(v)
        if k == L.ACT_LRELU:
            return F.leaky_relu(v, 0.1)
        if k == L.ACT_PRELU:
            return F.prelu(v, slope)
        if k == L.ACT_SIGMOID:
            return torch.sigmoid(v)
        if k == L.ACT_TANH:
            return torch.tanh(v)
        if k == L.ACT_SIN:
            return torch.sin(v)
        if k == L.ACT_GELU:
            return F.gelu(v)
        return v

    ref = act(ref, act1)
    if with_res:
        ref = ref + r
    ref = act(ref, act2) * out_scale
    got = out.float().cpu()
    err = float((got[..., 2:2 + Cout].permute(0, 3, 1, 2) - ref).abs().max())
    assert err <= tol(rt, float(ref.abs().max()) + 1.0), (err, float(ref.abs().max()))
    # channels outside the destination slice must be untouched
    assert float(got[..., :2].abs().max()) == 0.0 and float(got[..., 2 + Cout:].abs().max()) == 0.0
    return err


def conv_pair_case(rt, N=1, H=9, W=12, shapes=((64, 256, 1, 1), (128, 128, 1, 1)), seed=0, expect_pair=True):
    """gvfi_conv2d_pair (two independent weights-direct convolutions in one launch: the branches of the motion encoder,
    raft/update.py:94-112) against the same two problems launched one by one: bit-identical outputs, nothing else written.
    shapes: (Cin, Cout, KH, KW) of problem a / b.  expect_pair False: a pair the library declines (-2) -> the host falls back
    to two launches, same results."""
    g = torch.Generator().manual_seed(seed)
    dev = _dev(rt)
    probs = []
    for Cin, Cout, KH, KW in shapes:
        w = _rounded(rt, torch.randn(Cout, Cin, KH, KW, generator=g) / (Cin * KH * KW) ** 0.5)
        lay = ConvLayer(rt, w, torch.randn(Cout, generator=g), wdir=expect_pair)
        x = torch.randn(N, H, W, Cin, generator=g).to(rt.tdtype).to(dev)
        probs.append((lay, x, Cout))
    outs = {}
    for mode in ("pair", "single"):
        ys = [torch.full((N, H, W, c + 16), 3.0, dtype=rt.tdtype, device=dev) for _, _, c in probs]
        kw = [dict(layer=lay, x0=View(x, 0, x.shape[-1]), out=View(y, 8, c), act1=L.ACT_RELU) for (lay, x, c), y in zip(probs, ys)]
        if mode == "pair":
            keep = rt.pair_launch
            rt.pair_launch = True
            n0 = rt.n_launch
            rt.conv_pair(kw[0], kw[1])
            launches = rt.n_launch - n0
            rt.pair_launch = keep
            assert launches == (1 if expect_pair else 2), launches
        else:
            rt.conv(**kw[0])
            rt.conv(**kw[1])
        outs[mode] = [y.float().cpu() for y in ys]
    for a, b, (lay, x, c) in zip(outs["pair"], outs["single"], probs):
        assert torch.equal(a, b)
        assert float((a[..., :8] - 3.0).abs().max()) == 0.0 and float((a[..., 8 + c:] - 3.0).abs().max()) == 0.0
        wt = lay.w.float().cpu()[..., :x.shape[-1]].permute(0, 3, 1, 2)
        ref = F.relu(F.conv2d(x.float().cpu().permute(0, 3, 1, 2), wt, lay.b.cpu(), padding=(wt.shape[2] // 2, wt.shape[3] // 2)))
        err = float((a[..., 8:8 + c].permute(0, 3, 1, 2) - ref).abs().max())
        assert err <= tol(rt, float(ref.abs().max()) + 1.0), err


def p3x3_equals_glds_case(rt, N, H, W, Cin, Cout, split=None, act1=L.ACT_PRELU, with_res=False, act2=L.ACT_NONE, out_scale=1.0, seed=0, variant=0, algo_new=4, ld_extra=8,
                          slope_hi=0.3):
    """The halo-staged 3x3 kernel (conv_p3x3.hip, algo 4) walks K in the LDS-DMA kernel's order and shares its epilogue
    arithmetic: the two must agree bit for bit.  variant: algo bits 13, 14 (launch form: 0 = the library's choice, 1 = round-2
    kernel, 2 = tile per workgroup with the wave-private epilogue, 3 = persistent stream kernel); slope_hi > 0.9: PReLU slopes
    beyond 1 (the epilogue's general activation path instead of max(t, s t))."""
    g = torch.Generator().manual_seed(seed)
    dev = _dev(rt)
    w = _rounded(rt, torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5)
    lay = ConvLayer(rt, w, torch.randn(Cout, generator=g), slope=torch.rand(Cout, generator=g) * slope_hi + 0.1)
    x = torch.randn(N, H, W, Cin, generator=g).to(rt.tdtype).to(dev)
    if split is None:
        x0, x1 = View(x, 0, Cin), None
    else:
        xa = x[..., :split + 8].contiguous()           # a wider first source: pitch != channel count
        xb = x[..., split:].contiguous()
        x0, x1 = View(xa, 0, split), View(xb, 0, Cin - split)
    res = torch.randn(N, H, W, Cout, generator=g).to(rt.tdtype).to(dev) if with_res else None
    outs = []
    for algo in (algo_new + variant, 2):
        out = torch.full((N, H, W, Cout + ld_extra), 7.0, dtype=rt.tdtype, device=dev)
        rt.conv(lay, x0, View(out, 0, Cout), x1=x1, act1=act1, res=None if res is None else View(res, 0, Cout), act2=act2,
                slope2=lay.slope if act2 == L.ACT_PRELU else None, out_scale=out_scale, algo=algo, tile=256 if algo_new == 4 else 0)
        outs.append(out.float().cpu())
    assert float((outs[0][..., Cout:] - 7.0).abs().max()) == 0.0
    assert torch.equal(outs[0], outs[1]), float((outs[0] - outs[1]).abs().max())


def p3x3_repeat_case(rt, N, H, W, Cin, Cout, reps=5, seed=0):
    """The halo-staged 3x3 kernel in the library's own launch form, `reps` launches of one problem (residual + PReLU): every
    launch must reproduce the first bit for bit."""
    g = torch.Generator().manual_seed(seed)
    dev = _dev(rt)
    lay = ConvLayer(rt, _rounded(rt, torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5), torch.randn(Cout, generator=g),
                    slope=torch.rand(Cout, generator=g) * 0.3 + 0.1)
    x = torch.randn(N, H, W, Cin, generator=g).to(rt.tdtype).to(dev)
    res = torch.randn(N, H, W, Cout, generator=g).to(rt.tdtype).to(dev)
    first = None
    for _ in range(reps):
        out = torch.full((N, H, W, Cout), 7.0, dtype=rt.tdtype, device=dev)
        rt.conv(lay, View(x, 0, Cin), View(out, 0, Cout), act1=L.ACT_PRELU, res=View(res, 0, Cout), act2=L.ACT_PRELU, slope2=lay.slope, algo=4)
        if first is None:
            first = out.clone()
        else:
            assert torch.equal(out, first)


def gru_case(rt, N=1, H=6, W=9, C=16, seed=0, kh=1, kw=5, ctx_split=False, state_f32=False, wdir=False):
    """SepConvGRU half step with the fused epilogues (raft/update.py:58-66).  state_f32: float recurrent state (h, z)
    beside the bf16 operand copy (gvfi_conv_params.state_f32); wdir: weights-direct variant of the LDS-DMA kernel."""
    g = torch.Generator().manual_seed(seed)
    h = _rounded(rt, torch.tanh(torch.randn(N, C, H, W, generator=g)))
    x = _rounded(rt, torch.randn(N, 2 * C, H, W, generator=g))
    wz, wr, wq = (_rounded(rt, torch.randn(C, 3 * C, kh, kw, generator=g) / (3 * C * kh * kw) ** 0.5) for _ in range(3))
    bz, br, bq = (torch.randn(C, generator=g) for _ in range(3))
    dev = _dev(rt)
    lzr = ConvLayer(rt, torch.cat([wz, wr], 0), torch.cat([bz, br], 0), wdir=wdir)
    lq = ConvLayer(rt, wq, bq, wdir=wdir)
    sf = state_f32 and rt.precision == "bf16"
    if sf:
        h = torch.tanh(torch.randn(N, C, H, W, generator=g))        # the float state is NOT rounded to bf16
    ha, xa = _to_act(rt, h).to(dev), _to_act(rt, x).to(dev)         # operand copy of h (bf16-rounded)
    h32 = h.permute(0, 2, 3, 1).contiguous().to(dev) if sf else None
    hn32 = rt.f32(N, H, W, C) if sf else None
    zb = rt.f32(N, H, W, C) if sf else rt.act(N, H, W, C)
    rh, hn = rt.act(N, H, W, C), rt.act(N, H, W, C)
    st = dict(state_f32=True) if sf else {}
    if not ctx_split:
        rt.conv(lzr, ha, zb, x1=xa, epi=L.EPI_GRU_ZR, y2=rh, aux0=h32 if sf else ha, **st)
        rt.conv(lq, rh, hn, x1=xa, epi=L.EPI_GRU_Q, aux0=h32 if sf else ha, aux1=zb, y2=hn32, **st)
    else:
        # the first C channels of x play RAFT's constant context: their share of both convolutions is a separate
        # (bias-carrying) convolution evaluated once, handed to the gate epilogues as a pre-activation term
        keep = list(range(0, C)) + list(range(2 * C, 3 * C))
        wzr, bzr = torch.cat([wz, wr], 0), torch.cat([bz, br], 0)
        lzr_m, lq_m = ConvLayer(rt, wzr[:, keep], None, wdir=wdir), ConvLayer(rt, wq[:, keep], None, wdir=wdir)
        lzr_c, lq_c = ConvLayer(rt, wzr[:, C:2 * C], bzr), ConvLayer(rt, wq[:, C:2 * C], bq)
        czr, cq = rt.f32(N, H, W, 2 * C), rt.f32(N, H, W, C)
        rt.conv(lzr_c, View(xa, 0, C), czr)
        rt.conv(lq_c, View(xa, 0, C), cq)
        xm = View(xa, C, C)
        rt.conv(lzr_m, ha, zb, x1=xm, epi=L.EPI_GRU_ZR, y2=rh, aux0=h32 if sf else ha, res=czr, **st)
        rt.conv(lq_m, rh, hn, x1=xm, epi=L.EPI_GRU_Q, aux0=h32 if sf else ha, aux1=zb, y2=hn32, res=cq, **st)
    pad = (kh // 2, kw // 2)
    hop = _rounded(rt, h)                                       # what the convolutions read
    hx = torch.cat([hop, x], 1)
    z = torch.sigmoid(F.conv2d(hx, wz, bz, padding=pad))
    r = torch.sigmoid(F.conv2d(hx, wr, br, padding=pad))
    rhr = _rounded(rt, r * h)
    q = torch.tanh(F.conv2d(torch.cat([rhr, x], 1), wq, bq, padding=pad))
    zz = z if sf else _rounded(rt, z)
    ref = (1 - zz) * h + zz * q
    err = float((hn.float().cpu().permute(0, 3, 1, 2) - ref).abs().max())
    assert err <= tol(rt, 2.0), err
    if sf:     # the float state itself: no bf16 rounding of h, z or the result (hardware exp / rcp approximations only)
        e32 = float((hn32.cpu().permute(0, 3, 1, 2) - ref).abs().max())
        assert e32 <= 2e-4, e32
        assert torch.equal(hn.float().cpu(), hn32.cpu().to(torch.bfloat16).float())     # operand copy = rounded state
        ez = float((zb.cpu().permute(0, 3, 1, 2) - z).abs().max())
        assert ez <= 1e-4, ez
    return err


def corr_volume_case(rt, B=2, h=5, w=7, C=32, seed=0):
    g = torch.Generator().manual_seed(seed)
    f1 = _rounded(rt, torch.randn(B, C, h, w, generator=g))
    f2 = _rounded(rt, torch.randn(B, C, h, w, generator=g))
    dev = _dev(rt)
    a, b = _to_act(rt, f1).to(dev), _to_act(rt, f2).to(dev)
    P = h * w
    vol = rt.f32(B * P, P)
    rt.conv(None, a, View(vol.view(B, h, w, P)), groups=B, w_group_stride=P * C, w_raw=b, cout=P, out_scale=0.125)
    ref = torch.einsum("bcp,bcq->bpq", f1.reshape(B, C, P), f2.reshape(B, C, P)) * 0.125
    err = float((vol.cpu().view(B, P, P) - ref).abs().max())
    assert err <= tol(rt, float(ref.abs().max())), err


def instnorm_case(rt, N=2, H=9, W=11, C=24):
    g = torch.Generator().manual_seed(1)
    x = _rounded(rt, torch.randn(N, C, H, W, generator=g) * 2 + 0.5)
    r = _rounded(rt, torch.randn(N, C, H, W, generator=g))
    dev = _dev(rt)
    xa, ra = _to_act(rt, x).to(dev), _to_act(rt, r).to(dev)
    o1 = rt.instnorm(xa, C, relu=True).t
    o2 = rt.instnorm(xa, C, relu=True, res=ra).t
    o3 = rt.instnorm(xa, C, relu=False).t
    n = F.instance_norm(x, eps=1e-5)
    for got, ref in ((o1, F.relu(n)), (o2, F.relu(r + F.relu(n))), (o3, n)):
        err = float((got.float().cpu().permute(0, 3, 1, 2)[:, :C] - ref).abs().max())
        assert err <= tol(rt, 4.0), err


def patch_conv_case(rt, N=2, H=10, W=13, Cin=2, Cout=24, KH=7, KW=7):
    """im2col + 1x1 (gvfi_im2col) == the zero-padded KHxKW convolution (raft/update.py:100,107)."""
    g = torch.Generator().manual_seed(11)
    x = _rounded(rt, torch.randn(N, Cin, H, W, generator=g))
    w = _rounded(rt, torch.randn(Cout, Cin, KH, KW, generator=g) / (Cin * KH * KW) ** 0.5)
    b = torch.randn(Cout, generator=g)
    dev = _dev(rt)
    lay = PatchConvLayer(rt, w, b)
    xa = _to_act(rt, x).to(dev)
    out = rt.act(N, H, W, Cout)
    rt.patch_conv(lay, View(xa, 0, Cin), out, act1=L.ACT_RELU)
    ref = F.relu(F.conv2d(x, w, b, padding=(KH // 2, KW // 2)))
    err = float((out.float().cpu().permute(0, 3, 1, 2)[:, :Cout] - ref).abs().max())
    assert err <= tol(rt, 4.0), err


def conv_stats_case(rt, N=2, H=16, W=16, Cin=64, Cout=64, tile=0, algo=0):
    """InstanceNorm statistics fused into the convolution's store loop == sums over the stored tensor."""
    assert rt.precision == "bf16"
    g = torch.Generator().manual_seed(14)
    x = _rounded(rt, torch.randn(N, Cin, H, W, generator=g))
    w = _rounded(rt, torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5)
    b = torch.randn(Cout, generator=g)
    dev = _dev(rt)
    lay = ConvLayer(rt, w, b)
    xa = _to_act(rt, x).to(dev)
    out = rt.act(N, H, W, Cout)
    stats = rt.stats_tensor(N, Cout)
    rt.conv(lay, xa, out, stats=stats, tile=tile, algo=algo)
    assert rt.last_stats_fused
    o = out.float().cpu()
    ref = torch.stack([o.sum((1, 2)), (o * o).sum((1, 2))], -1)          # [N, Cout, 2] of the stored (rounded) values
    err = float((rt.stats_values(stats.cpu()) - ref).abs().max())
    assert err <= 1e-3 * float(ref.abs().max()), err
    # and the normalisation that consumes them equals F.instance_norm of the stored tensor
    y = rt.instnorm(out, Cout, relu=False, stats=stats).t.float().cpu()
    n = F.instance_norm(o.permute(0, 3, 1, 2), eps=1e-5).permute(0, 2, 3, 1)
    assert float((y - n).abs().max()) <= tol(rt, 4.0)
    if algo == 5:      # (the halo-staged kernel's tiles never straddle images)
        return
    # a shape whose tiles straddle images is refused (caller falls back to gvfi_instnorm_stats)
    out2 = rt.act(N, H - 1, W - 5, Cout)
    rt.conv(lay, _to_act(rt, x[:, :, :H - 1, :W - 5]).to(dev), out2, stats=rt.stats_tensor(N, Cout))
    assert not rt.last_stats_fused


def tap_split_conv_case(rt, N=2, H=9, W=12, Cin=64, Cout=2):
    """1x1 to per-tap partial sums + gvfi_tap_sum == the 3x3 zero-padded convolution with residual, in place
    (raft/update.py:6-14, raft/raft.py:157 coords1 = coords1 + delta_flow)."""
    g = torch.Generator().manual_seed(13)
    x = _rounded(rt, torch.randn(N, Cin, H, W, generator=g))
    w = _rounded(rt, torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5)
    b = torch.randn(Cout, generator=g)
    r = torch.randn(N, H, W, Cout, generator=g)
    dev = _dev(rt)
    lay = TapSplitConvLayer(rt, w, b)
    xa = _to_act(rt, x).to(dev)
    out = r.clone().to(dev)
    rt.tap_split_conv(lay, xa, View(out), res=View(out))
    ref = F.conv2d(x, w, b, padding=1).permute(0, 2, 3, 1) + r
    err = float((out.cpu() - ref).abs().max())
    assert err <= tol(rt, 4.0), err


def inr_mlp_case(rt, B=1, H=9, W=61):
    """fused hypo-network (gvfi_inr_mlp) vs the layer chain of modules/hyponet.py:71-146 with bf16-rounded
    weights, inputs and hidden activations (what the layer-by-layer bf16 path computes)."""
    assert rt.precision == "bf16"
    g = torch.Generator().manual_seed(12)
    dims = InrMlp.DIMS
    layers = []
    for o, i in dims:
        w = F.normalize(torch.randn(o, i, generator=g), dim=1)
        layers.append((w, torch.randn(o, generator=g) * 0.3))
    lat = _rounded(rt, torch.randn(B, H, W, 32, generator=g))
    coord = torch.rand(B, 1, H, W, 3, generator=g) * 2 - 1
    dev = _dev(rt)
    mlp = InrMlp(rt, layers)
    lat_d = rt.act(B, H, W, 32, zero=True, pitch=40)
    lat_d[..., :32] = lat.to(lat_d.dtype)
    out = rt.f32(B, H, W, 2)
    rt.inr_mlp(mlp, View(lat_d.to(dev), 0, 32), coord.to(dev).contiguous(), out)
    hcur = torch.cat([lat, _rounded(rt, coord[:, 0])], -1).reshape(-1, 35)
    for li, (w, b) in enumerate(layers):
        hcur = hcur @ _rounded(rt, w).t() + b
        if li < 4:
            hcur = _rounded(rt, torch.sin(hcur))
    err = float((out.cpu().reshape(-1, 2) - hcur).abs().max())
    assert err <= 2e-2, err    # bf16 hidden activations: a 1-ulp flip of a hidden unit moves the output ~4e-3


def resize_warp_shuffle_case(rt):
    g = torch.Generator().manual_seed(2)
    dev = _dev(rt)
    N, C, H, W = 2, 12, 16, 24
    x = _rounded(rt, torch.randn(N, C, H, W, generator=g))
    xa = _to_act(rt, x).to(dev)
    xf = x.permute(0, 2, 3, 1).contiguous().to(dev)
    for s in (0.25, 0.5, 2.0, 4.0):
        ref = F.interpolate(x, scale_factor=s, mode="bilinear", align_corners=False)
        o = rt.resize(xa, C, s).t.float().cpu().permute(0, 3, 1, 2)[:, :C]
        assert float((o - ref).abs().max()) <= tol(rt, 4.0)
        o = rt.resize(xf, C, s, mul=s).t.cpu().permute(0, 3, 1, 2)
        assert float((o - s * ref).abs().max()) <= 2e-5 * 16
    ref = F.interpolate(x, size=(24, 36), mode="bilinear")
    o = rt.resize(xf, C, None, size=(24, 36)).t.cpu().permute(0, 3, 1, 2)
    assert float((o - ref).abs().max()) <= 1e-4
    # warp: includes far out-of-range flows (border clamp)
    flow = torch.randn(N, 2, H, W, generator=g) * 6
    flow[0, :, 0, 0] = torch.tensor([-100.0, 50.0])
    ref = orc.warp(x, flow)
    fa = flow.permute(0, 2, 3, 1).contiguous().to(dev)
    out = rt.act(N, H, W, C + 8, zero=True)
    rt.warp(xa, C, fa, View(out, 8, C))
    o = out.float().cpu().permute(0, 3, 1, 2)[:, 8:8 + C]
    assert float((o - ref).abs().max()) <= tol(rt, 4.0)
    # pixel shuffle
    y = _rounded(rt, torch.randn(N, 16, 5, 6, generator=g))
    o = rt.pixel_shuffle2(_to_act(rt, y).to(dev), 4).float().cpu().permute(0, 3, 1, 2)[:, :4]
    assert float((o - F.pixel_shuffle(y, 2)).abs().max()) == 0.0
    # planes resize (input down-sampling, gimmvfi_r.py:329-337)
    img = torch.rand(2, 3, 2, 16, 24, generator=g)
    o = rt.resize_planes(img.to(dev), 0.5).cpu()
    ref = torch.stack([F.interpolate(img[:, :, f], scale_factor=0.5, mode="bilinear", align_corners=False) for f in (0, 1)], 2)
    assert float((o - ref).abs().max()) <= 1e-6


def corr_lookup_case(rt, B=2, h=16, w=24):
    g = torch.Generator().manual_seed(3)
    dev = _dev(rt)
    vol = torch.randn(B, h, w, 1, h, w, generator=g)
    pyr = orc.corr_pyramid(vol)
    coords = orc.coords_grid(B, h, w) + torch.randn(B, 2, h, w, generator=g) * 4
    coords[0, :, 0, 0] = torch.tensor([-30.0, 3.3])     # fully out of range window
    coords[0, :, 1, 1] = torch.tensor([w - 1.0, h - 1.0])  # exactly on the border
    ref = orc.corr_lookup(pyr, coords)
    dp = [rt.f32(B * h * w, h * w).copy_(vol.reshape(B * h * w, h * w))]
    hh, ww = h, w
    for _ in range(3):
        dp.append(rt.avgpool2(dp[-1], B * h * w, hh, ww))
        hh, ww = hh // 2, ww // 2
    for lvl in range(4):
        assert float((dp[lvl].cpu().reshape(pyr[lvl].shape) - pyr[lvl]).abs().max()) <= 1e-5
    out = rt.act(B, h, w, 324 + 8, zero=True)
    rt.corr_lookup(dp, coords.permute(0, 2, 3, 1).contiguous().to(dev), View(out, 8, 324), B, h, w, h, w)
    o = out.float().cpu().permute(0, 3, 1, 2)[:, 8:8 + 324]
    assert float((o - ref).abs().max()) <= tol(rt, float(ref.abs().max())) + 1e-4
    # the LDS-staged variant (gvfi_corr_lookup_lds): bit-identical outputs, also for coordinates exactly on cell borders
    # and windows partly / fully outside the map
    cd = coords.permute(0, 2, 3, 1).contiguous()
    cd[1, 2, 3] = torch.tensor([5.0, 7.0])
    cd[1, 3, 4] = torch.tensor([-3.0, h + 2.5])
    cd[1, 4, 5] = torch.tensor([1.0e9, -1.0e9])
    cd = cd.to(dev)
    a_ = rt.act(B, h, w, 324 + 8, zero=True)
    b_ = rt.act(B, h, w, 324 + 8, zero=True)
    keep = rt.lookup_lds
    try:
        rt.lookup_lds = False
        rt.corr_lookup(dp, cd, View(a_, 8, 324), B, h, w, h, w)
        rt.lookup_lds = True
        rt.corr_lookup(dp, cd, View(b_, 8, 324), B, h, w, h, w)
    finally:
        rt.lookup_lds = keep
    assert torch.equal(a_.cpu(), b_.cpu())


def s2d_case(rt, N, H, W, Cin, Cout, k, seed=0):
    """filter size == stride, no padding, as a k x 1 convolution over the space-to-depth VIEW of the input
    (ops.S2DConvLayer / Runtime.s2d_conv) against torch's strided convolution; the launched kernel is reported."""
    g = torch.Generator().manual_seed(seed)
    x = _rounded(rt, torch.randn(N, Cin, H, W, generator=g))
    w = _rounded(rt, torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5)
    b = torch.randn(Cout, generator=g)
    xa = _to_act(rt, x).to(_dev(rt))
    lay = S2DConvLayer(rt, w, b, xa.shape[-1])
    out = rt.act(N, H // k, W // k, Cout)
    rt.s2d_conv(lay, xa, out)
    ref = F.conv2d(x, w, b, stride=k).permute(0, 2, 3, 1)
    got = out.float().cpu()[..., :Cout]
    err = float((got - ref).abs().max())
    assert err <= tol(rt, float(ref.abs().max())), (err, float(ref.abs().max()))
    # the strided form of the same layer agrees too (what the engine falls back to on ragged grids)
    lay0 = ConvLayer(rt, w, b, stride=k, pad=(0, 0))
    out0 = rt.act(N, H // k, W // k, Cout)
    rt.conv(lay0, View(xa, 0, Cin), out0)
    assert float((out0.float().cpu()[..., :Cout] - got).abs().max()) <= tol(rt, float(ref.abs().max()))


def flow_step_case(rt, N=2, h=11, w=21, first=False):
    """gvfi_flow_step == gvfi_tap_sum + gvfi_flow_pack + gvfi_im2col in sequence, bit for bit (ragged tiles, image borders;
    first: no pending update)."""
    g = torch.Generator().manual_seed(21)
    dev = _dev(rt)
    w2 = torch.randn(2, 64, 3, 3, generator=g) * 0.1
    b2 = torch.randn(2, generator=g)
    tapl = TapSplitConvLayer(rt, w2, b2)
    patl = PatchConvLayer(rt, torch.randn(16, 2, 7, 7, generator=g), torch.randn(16, generator=g))
    coords0 = (orc.coords_grid(N, h, w) + torch.randn(N, 2, h, w, generator=g) * 3).permute(0, 2, 3, 1).contiguous()
    P = torch.randn(N, h, w, 20, generator=g)
    outs = []
    for fused in (False, True):
        co = coords0.clone().to(dev)
        fp = P.clone().to(dev)
        fl = rt.act(N, h, w, 2, zero=True)
        xb = rt.act(N, h, w, 16, zero=True)
        col = torch.full((N, h, w, patl.kpad), 5.0, dtype=rt.tdtype, device=dev)
        if fused:
            co = rt.flow_step(tapl, patl, None if first else fp, co, fl, View(xb, 8, 2), col, coords_out=torch.empty_like(co))
        else:
            if not first:
                rt._chk(rt.lib.tap_sum(fp.data_ptr(), 20, 2, 3, 3, tapl.b.data_ptr(), co.data_ptr(), 2, co.data_ptr(), 2, N, h, w,
                                       rt.stream()), "tap_sum")
            rt.flow_pack(co, fl, View(xb, 8, 2))
            rt._chk(rt.lib.im2col(fl.data_ptr(), fl.shape[-1], 2, N, h, w, 7, 7, 3, 3, col.data_ptr(), patl.kpad, rt.dtype,
                                  rt.stream()), "im2col")
        outs.append([t.cpu().clone() for t in (co, fl, xb, col)])
    for a_, b_ in zip(*outs):
        assert torch.equal(a_.view(torch.uint8) if a_.dtype != torch.float32 else a_, b_.view(torch.uint8) if b_.dtype != torch.float32 else b_)
    return True


def convex_upsample_case(rt, N=2, h=6, w=9):
    g = torch.Generator().manual_seed(4)
    dev = _dev(rt)
    flow = torch.randn(N, 2, h, w, generator=g) * 3
    mask = torch.randn(N, 576, h, w, generator=g)
    ref = orc.convex_upsample(flow, mask)
    coords = (orc.coords_grid(N, h, w) + flow).permute(0, 2, 3, 1).contiguous().to(dev)
    m = mask.permute(0, 2, 3, 1).contiguous().to(dev)
    o = rt.convex_upsample(coords, m).cpu().permute(0, 3, 1, 2)
    assert float((o - ref).abs().max()) <= 2e-5 * 32


def splat_case(rt, B=2, H=12, W=20, C=16):
    """softmax-splat forward incl. the reference's edge cases: out-of-image targets, non-finite flow
    (skipped), pixels nothing lands on (zero denominator -> 1)."""
    g = torch.Generator().manual_seed(5)
    dev = _dev(rt)
    lat = _rounded(rt, torch.randn(B, C, H, W, generator=g))
    flow = torch.randn(B, 2, H, W, generator=g) * 5
    flow[0, :, 2, 3] = float("nan")
    flow[0, 0, 4, 5] = float("inf")
    flow[1, :, :, :6] = 40.0          # splat a whole band out of the image -> holes
    z = torch.rand(B, 1, H, W, generator=g) + 0.5
    t = torch.tensor([0.3, 0.75])
    for omt in (0, 1):
        ts = (1 - t) if omt else t
        ref = orc.softsplat_linear_zeroeps(lat, flow * ts.view(-1, 1, 1, 1), z)
        acc = rt.f32(B, H, W, C + 1, zero=True)
        la = _to_act(rt, lat).to(dev)
        fd = flow.permute(0, 2, 3, 1).contiguous().to(dev)   # keep references alive across the launch
        zd = z.reshape(B, H, W).contiguous().to(dev)
        td = t.to(dev)
        rt._chk(rt.lib.softsplat_accum(la.data_ptr(), la.shape[-1], C, fd.data_ptr(), zd.data_ptr(), td.data_ptr(), omt,
                                       acc.data_ptr(), B, H, W, rt.dtype, rt.stream()), "softsplat_accum")
        out = rt.act(B, H, W, C)
        rt._chk(rt.lib.softsplat_normalize(acc.data_ptr(), C, out.data_ptr(), out.shape[-1], B * H * W, rt.dtype,
                                           rt.stream()), "softsplat_normalize")
        o = out.float().cpu().permute(0, 3, 1, 2)
        assert torch.isfinite(o).all()
        assert float((o - ref).abs().max()) <= tol(rt, float(ref.abs().max()))


def compose_sbs_case(rt, b=3, N=4, H0=21, W0=30, pad=(1, 1, 5, 6)):
    """gvfi_compose_sbs_u8 against the reference's host statement (src/video_Nx.py:139-151, 198-216): originals converted with
    (x * 255.0).astype(np.uint8) on the un-padded frame, BGR, cv2.hconcat([orig, frame]); with and without the leading frame."""
    import numpy as np

    g = torch.Generator().manual_seed(33)
    dev = _dev(rt)
    l, r, t, bt = pad
    orig = torch.randint(0, 256, (b + 1, 3, H0, W0), generator=g).float() / 255.0          # load_image: uint8 / 255.0
    frames = F.pad(orig, (l, r, t, bt), mode="replicate").contiguous()
    pred = torch.randint(0, 256, (b, N - 1, H0, W0, 3), generator=g, dtype=torch.uint8)
    ob = [(orig[k].numpy().transpose(1, 2, 0) * 255.0)[:, :, ::-1].astype(np.uint8) for k in range(b + 1)]
    for lead in (0, 1):
        got = rt.compose_sbs(frames.to(dev), t, l, pred.to(dev), N, lead).cpu().numpy()
        want = [np.concatenate([ob[0], ob[0]], 1)] if lead else []
        for jj in range(b):
            for i in range(N - 1):
                want.append(np.concatenate([ob[jj], pred[jj, i].numpy()[:, :, ::-1]], 1))
            want.append(np.concatenate([ob[jj + 1], ob[jj + 1]], 1))
        want = np.stack(want)
        assert got.shape == want.shape and (got == want).all()


def col7_planar_case(rt, N=2, H=37, W=45, seed=8):
    """Column kernel, last layer of the combination block (18 -> 3, float residual): the folded finalisation (algo bit 6:
    clamp((y + 1) / 2, 0, 1) stored planar at y2, y untouched) equals the NHWC result + gvfi_finalize_image bit for bit."""
    g = torch.Generator().manual_seed(seed)
    dev = _dev(rt)
    x = _rounded(rt, torch.randn(N, 18, H, W, generator=g))
    w = _rounded(rt, torch.randn(3, 18, 7, 7, generator=g) / (18 * 49) ** 0.5)
    lay = ConvLayer(rt, w, torch.randn(3, generator=g))
    xa = rt.act(N, H, W, 18, zero=True)
    xa[..., :18] = x.permute(0, 2, 3, 1).to(xa.dtype)
    xa = xa.to(dev)
    mean4 = (torch.rand(N, H, W, 4, generator=g) * 2 - 1).to(dev)
    o4 = rt.f32(N, H, W, 4)
    rt.conv(lay, View(xa, 0, 18), View(o4, 0, 3), res=View(mean4, 0, 3), pad16=True, algo=7)
    assert not rt.last_planar
    want = rt.f32(N, 3, H, W)
    rt._chk(rt.lib.finalize_image(o4.data_ptr(), 4, want.data_ptr(), N, H, W, rt.stream()), "finalize_image")
    o4b = rt.f32(N, H, W, 4)
    o4b.fill_(7.0)
    got = rt.f32(N, 3, H, W)
    got.fill_(-5.0)
    rt.conv(lay, View(xa, 0, 18), View(o4b, 0, 3), res=View(mean4, 0, 3), pad16=True, algo=7, planar3=got)
    assert rt.last_planar
    assert torch.equal(got.cpu(), want.cpu())
    assert float((o4b.cpu() - 7.0).abs().max()) == 0.0          # the NHWC tensor is not written
    assert float(want.min()) >= 0.0 and float(want.max()) <= 1.0 and float(want.std()) > 0.05


def splat_gather_case(rt, B=2, H=12, W=20, converge=False):
    """The list-based gather form of the softmax splat (gvfi_softsplat_lists + gvfi_softsplat_gather): both directions in one
    launch, the same edge cases as splat_case, against the oracle; bit-identical over repeated runs (one writer per output,
    fixed order of additions); converge: all sources of a 6 x 6 block (True) or of a converge x (converge + 2) block land in ONE
    cell (lists longer than the register array -> the chunked-selection branch: 36 = 4 full walks + half a one; 9 x 11 = 99
    sources = 12 walks + 3 sources)."""
    g = torch.Generator().manual_seed(15)
    dev = _dev(rt)
    C = 16
    lat = [_rounded(rt, torch.randn(B, C, H, W, generator=g)) for _ in range(2)]
    flow = [torch.randn(B, 2, H, W, generator=g) * 5 for _ in range(2)]
    flow[0][0, :, 2, 3] = float("nan")
    flow[0][0, 0, 4, 5] = float("inf")
    flow[1][1, :, :, :6] = 40.0          # splat a whole band out of the image -> holes
    flow[1][0, :, 0, 0] = float("-inf")
    t = torch.tensor([0.3, 0.75])[:B]
    if converge:
        ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
        bh, bw = (6, 6) if converge is True else (int(converge), int(converge) + 2)
        assert 2 + bh <= H and 3 + bw <= W
        for d in range(2):       # pixels (2..2+bh-1, 3..3+bw-1) of image 0 all move to (5.3, 6.6) at this direction's time scale
            ts0 = float((1 - t[0]) if d else t[0])
            flow[d][0, 0, 2:2 + bh, 3:3 + bw] = (6.6 - xs[2:2 + bh, 3:3 + bw]) / ts0
            flow[d][0, 1, 2:2 + bh, 3:3 + bw] = (5.3 - ys[2:2 + bh, 3:3 + bw]) / ts0
    z = [torch.rand(B, 1, H, W, generator=g) + 0.5 for _ in range(2)]
    latcat = rt.act(B, H, W, 64, zero=True)
    for d in range(2):
        latcat[..., 16 * d:16 * d + 16] = _to_act(rt, lat[d]).to(dev)[..., :16]
    fd = [f.permute(0, 2, 3, 1).contiguous().to(dev) for f in flow]
    zd = [zz.reshape(B, H, W).contiguous().to(dev) for zz in z]
    td = t.to(dev)
    outs = []
    for _ in range(3):
        head = torch.full((2, B, H + 1, W + 1), -1, dtype=torch.int32, device=dev)
        nxt = torch.empty((2, B, H, W), dtype=torch.int32, device=dev)
        latcat[..., 32:] = 7.0
        rt._chk(rt.lib.softsplat_lists(fd[0].data_ptr(), fd[1].data_ptr(), td.data_ptr(), head.data_ptr(), nxt.data_ptr(), B, H, W,
                                       rt.stream()), "softsplat_lists")
        rt._chk(rt.lib.softsplat_gather(latcat.data_ptr(), latcat.shape[-1], fd[0].data_ptr(), fd[1].data_ptr(), zd[0].data_ptr(),
                                        zd[1].data_ptr(), td.data_ptr(), head.data_ptr(), nxt.data_ptr(),
                                        View(latcat, 32, 32).ptr, latcat.shape[-1], B, H, W, rt.dtype, rt.stream()), "softsplat_gather")
        outs.append(latcat.float().cpu().clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])        # deterministic
    assert torch.equal(outs[0][..., :32], torch.cat([_to_act(rt, lat[0])[..., :16], _to_act(rt, lat[1])[..., :16]], -1).float().cpu())
    for d in range(2):
        ts = (1 - t) if d else t
        ref = orc.softsplat_linear_zeroeps(lat[d], flow[d] * ts.view(-1, 1, 1, 1), z[d])
        o = outs[0][..., 32 + 16 * d:48 + 16 * d].permute(0, 3, 1, 2)
        assert torch.isfinite(o).all()
        assert float((o - ref).abs().max()) <= tol(rt, float(ref.abs().max())), (d, float((o - ref).abs().max()))


def splat_nchw_case(rt, N=2, C=5, H=13, W=21):
    """The reference's native op contract (softsplat_func.forward, softsplat.py:358-446): NCHW f32 in / out, the
    caller zero-initialises tenOut; checked against the O(P) restatement of the CuPy kernel in the oracle."""
    g = torch.Generator().manual_seed(12)
    dev = _dev(rt)
    ten_in = torch.randn(N, C, H, W, generator=g)
    flow = torch.randn(N, 2, H, W, generator=g) * 4
    flow[0, :, 1, 2] = float("nan")
    flow[1, 1, 3, 4] = float("-inf")
    flow[1, :, :, -3:] = 25.0
    ref = orc.splat_sum(ten_in, flow)
    a, f = ten_in.to(dev).contiguous(), flow.to(dev).contiguous()
    out = rt.f32(N, C, H, W, zero=True)
    rt._chk(rt.lib.softsplat_out_nchw(a.data_ptr(), f.data_ptr(), out.data_ptr(), N, C, H, W, rt.stream()),
            "softsplat_out_nchw")
    assert float((out.cpu() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))


def combine_warps_up_case(rt, B=2, H=12, W=20, scale=2):
    """gvfi_combine_warps_up (up-sampling of the decoder output + multi_flow_combine front half + planar flows in one
    pass) against the separate passes it replaces, bit for bit, and those against torch (fi_components.py:57-88,
    gimmvfi_r.py:294-303)."""
    g = torch.Generator().manual_seed(21 + scale)
    dev = _dev(rt)
    Hf, Wf = H * scale, W * scale
    dec = torch.randn(B, H, W, 24, generator=g)
    dec[..., :12] *= 3.0
    dec[..., 12:15] = torch.sigmoid(dec[..., 12:15])
    i0 = torch.zeros(B, Hf, Wf, 4)
    i1 = torch.zeros(B, Hf, Wf, 4)
    i0[..., :3] = torch.rand(B, Hf, Wf, 3, generator=g) * 2 - 1
    i1[..., :3] = torch.rand(B, Hf, Wf, 3, generator=g) * 2 - 1
    dec_d, i0d, i1d = dec.to(dev), i0.to(dev), i1.to(dev)
    # separate passes
    if scale != 1:
        decf = rt.f32(B, Hf, Wf, 24)
        rt.resize(View(dec_d, 0, 12), 12, float(scale), mul=float(scale), out=View(decf, 0, 12))
        rt.resize(View(dec_d, 12, 12), 12, float(scale), out=View(decf, 12, 12))
    else:
        decf = dec_d
    cw = rt.act(B, Hf, Wf, 9, zero=False)
    mean4 = rt.f32(B, Hf, Wf, 4)
    rt._chk(rt.lib.combine_warps(i0d.data_ptr(), i1d.data_ptr(), decf.data_ptr(), 24, cw.data_ptr(), cw.shape[-1],
                                 cw.shape[-1], mean4.data_ptr(), B, Hf, Wf, rt.dtype, rt.stream()), "combine_warps")
    f0 = rt.nhwc_to_nchw(View(decf, 0, 6), 6)
    f1 = rt.nhwc_to_nchw(View(decf, 6, 6), 6)
    # fused
    cw2 = rt.act(B, Hf, Wf, 9, zero=False)
    cw2.fill_(3.0)
    mean42 = rt.f32(B, Hf, Wf, 4)
    f02, f12 = rt.f32(B, 3, 2, Hf, Wf), rt.f32(B, 3, 2, Hf, Wf)
    rt._chk(rt.lib.combine_warps_up(i0d.data_ptr(), i1d.data_ptr(), dec_d.data_ptr(), 24, H, W, cw2.data_ptr(),
                                    cw2.shape[-1], cw2.shape[-1], mean42.data_ptr(), f02.data_ptr(), f12.data_ptr(), B, 0,
                                    Hf, Wf, rt.dtype, rt.stream()), "combine_warps_up")
    assert torch.equal(cw2.cpu(), cw.cpu())
    # timestep-batched form: the decoder output of 2 timesteps [t][b] against the SAME B source images (src_B = B)
    # == the two timesteps run separately
    dec2 = torch.cat([dec_d, dec_d.flip(-3).contiguous()], 0).contiguous()
    cw3 = rt.act(2 * B, Hf, Wf, 9, zero=False)
    mean43 = rt.f32(2 * B, Hf, Wf, 4)
    f03, f13 = rt.f32(2 * B, 3, 2, Hf, Wf), rt.f32(2 * B, 3, 2, Hf, Wf)
    rt._chk(rt.lib.combine_warps_up(i0d.data_ptr(), i1d.data_ptr(), dec2.data_ptr(), 24, H, W, cw3.data_ptr(),
                                    cw3.shape[-1], cw3.shape[-1], mean43.data_ptr(), f03.data_ptr(), f13.data_ptr(), 2 * B, B,
                                    Hf, Wf, rt.dtype, rt.stream()), "combine_warps_up")
    cw4 = rt.act(B, Hf, Wf, 9, zero=False)
    mean44 = rt.f32(B, Hf, Wf, 4)
    rt._chk(rt.lib.combine_warps_up(i0d.data_ptr(), i1d.data_ptr(), dec2[B:].data_ptr(), 24, H, W, cw4.data_ptr(),
                                    cw4.shape[-1], cw4.shape[-1], mean44.data_ptr(), None, None, B, 0,
                                    Hf, Wf, rt.dtype, rt.stream()), "combine_warps_up")
    assert torch.equal(cw3[:B].cpu(), cw.cpu()) and torch.equal(cw3[B:].cpu(), cw4.cpu())
    assert torch.equal(mean43[:B].cpu(), mean4.cpu()) and torch.equal(mean43[B:].cpu(), mean44.cpu())
    assert torch.equal(f03[:B].cpu(), f02.cpu())
    assert torch.equal(mean42.cpu(), mean4.cpu())
    assert torch.equal(f02.cpu().reshape(B, 6, Hf, Wf), f0.cpu())
    assert torch.equal(f12.cpu().reshape(B, 6, Hf, Wf), f1.cpu())
    # torch statement of the up-sampled flows (the warps are covered by the end-to-end goldens)
    d = dec.permute(0, 3, 1, 2)
    if scale != 1:
        up = F.interpolate(d[:, :12], scale_factor=float(scale), mode="bilinear", align_corners=False) * scale
    else:
        up = d[:, :12]
    assert float((f02.cpu().reshape(B, 6, Hf, Wf) - up[:, :6]).abs().max()) <= 1e-4
    assert float((f12.cpu().reshape(B, 6, Hf, Wf) - up[:, 6:12]).abs().max()) <= 1e-4


def splat_weights_and_norm_case(rt, sd, B=2, H=16, W=20):
    g = torch.Generator().manual_seed(6)
    dev = _dev(rt)
    # rough (non-smooth) flows keep the variance away from the ill-conditioned zero-variance regime
    f01 = torch.randn(B, 2, H, W, generator=g) * 3
    f10 = torch.randn(B, 2, H, W, generator=g) * 3
    w1, w2 = orc.cal_splatting_weights(sd, f01, f10)
    a, b = f01.permute(0, 2, 3, 1).contiguous().to(dev), f10.permute(0, 2, 3, 1).contiguous().to(dev)
    z0, z1 = rt.f32(B, H, W), rt.f32(B, H, W)
    g9 = sd["g_filter"].reshape(9).contiguous().to(dev)
    rt._chk(rt.lib.splat_weights(a.data_ptr(), b.data_ptr(), g9.data_ptr(), float(sd["alpha_v"]), float(sd["alpha_fe"]),
                                 z0.data_ptr(), z1.data_ptr(), B, H, W, rt.stream()), "splat_weights")
    assert float((z0.cpu() - w1[:, 0]).abs().max()) <= 1e-4
    assert float((z1.cpu() - w2[:, 0]).abs().max()) <= 1e-4
    # normalisation round trip  (fi_utils.py:52-64)
    nref, sref = orc.normalize_flow(torch.stack([f01, -f10], 2))
    sc = rt.f32(B, zero=True)
    rt._chk(rt.lib.flow_absmax(a.data_ptr(), b.data_ptr(), sc.data_ptr(), B, H * W, rt.stream()), "flow_absmax")
    assert float((sc.cpu() - sref.reshape(B)).abs().max()) == 0.0
    nf = rt.act(2 * B, H, W, 2, zero=True)
    nfl = rt.f32(B, 2, 2, H, W)
    rt._chk(rt.lib.flow_normalize(a.data_ptr(), b.data_ptr(), sc.data_ptr(), nf.data_ptr(), nf.shape[-1], nf.shape[-1],
                                  nfl.data_ptr(), B, H, W, rt.dtype, rt.stream()), "flow_normalize")
    assert float((nfl.cpu() - nref).abs().max()) <= 1e-6
    got = nf.float().cpu()
    assert float((got[:B, ..., :2].permute(0, 3, 1, 2) - nref[:, :, 0]).abs().max()) <= tol(rt, 1.0)
    assert float((got[B:, ..., :2].permute(0, 3, 1, 2) - nref[:, :, 1]).abs().max()) <= tol(rt, 1.0)
    ninr = torch.rand(B, H, W, 2, generator=g).to(dev)
    ft, nn_ = rt.f32(B, H, W, 2), rt.f32(B, 2, 1, H, W)
    rt._chk(rt.lib.flow_unnormalize(ninr.data_ptr(), sc.data_ptr(), ft.data_ptr(), nn_.data_ptr(), B, H * W, rt.stream()),
            "flow_unnormalize")
    ref = orc.unnormalize_flow(ninr.cpu().permute(0, 3, 1, 2).unsqueeze(2), sref)
    assert float((ft.cpu().permute(0, 3, 1, 2) - ref[:, :, 0]).abs().max()) <= 1e-5


def flow_to_image_case(rt, n=3, h=37, w=53):
    """gvfi_flow_to_image == the CLI's numpy flow_viz.flow_to_image (reference src/utils/flow_viz.py), per image normalisation."""
    import importlib.util
    import os

    import numpy as np

    spec = importlib.util.spec_from_file_location(
        "gvfi_cli_flow_viz", os.path.join(os.path.dirname(__file__), "..", "gimm-vfi_amd", "src", "utils", "flow_viz.py"))
    fv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fv)
    flow_to_image, make_colorwheel = fv.flow_to_image, fv.make_colorwheel
    g = torch.Generator().manual_seed(5)
    flows = torch.randn(n, 2, h, w, generator=g) * torch.tensor([0.3, 4.0, 40.0][:n]).view(n, 1, 1, 1)
    flows[0, :, 0, 0] = 0.0
    dev = _dev(rt)
    wheel = torch.from_numpy(make_colorwheel()).float().to(dev)
    got = rt.flow_to_image(flows.to(dev).contiguous(), wheel, bgr=True).cpu().numpy().astype(np.int32)
    bad = 0
    for i in range(n):
        ref = flow_to_image(flows[i].permute(1, 2, 0).numpy(), convert_to_bgr=True).astype(np.int32)
        d = np.abs(got[i] - ref)
        assert d.max() <= 1, d.max()                      # atan2f may differ by an ulp from numpy's: a floor() can flip
        bad += int((d > 0).sum())
    assert bad <= 1e-3 * got.size, bad
    return bad
