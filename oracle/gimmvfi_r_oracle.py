"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the GIMM-VFI-R inference path.

A functional, state_dict-driven restatement (plain torch fp32 on CPU, NCHW) of
the reference algorithm  generalizable_INR/gimmvfi_r.py:324-407  and everything
it calls.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this file; the product path (gimm-vfi_amd/) never does and fails
loudly when the HIP library is missing.

Parity status: PINNED.  tests/test_oracle_pin.py checks this file against the
real reference imported from /root/reference (oracle/ref_harness.py) when that
checkout is present, and against tests/golden/*.pt fixtures that were produced
by the reference itself (oracle/make_golden.py) everywhere else.  The reference
ships no tests or golden vectors of its own.

Every function cites the reference file:line it follows (paths relative to
/root/reference/src/models/generalizable_INR/).  ``taps`` (a dict) collects the
stage-boundary tensors of SURVEY.md section 8a so HIP kernels can be compared
stage by stage.
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- helpers
def _conv(sd, key, x, stride=1, padding=0, padding_mode="zeros"):
    w = sd[key + ".weight"]
    b = sd.get(key + ".bias")
    if padding_mode == "reflect":
        p = padding
        x = F.pad(x, (p, p, p, p), mode="reflect")
        padding = 0
    return F.conv2d(x, w, b, stride=stride, padding=padding)


def _bn(sd, key, x, eps=1e-5):
    # eval-mode BatchNorm2d (raft/extractor.py:21-24, modules/fi_components.py:225-226)
    return F.batch_norm(
        x, sd[key + ".running_mean"], sd[key + ".running_var"], sd[key + ".weight"], sd[key + ".bias"], False, 0.0, eps
    )


def _inorm(x, eps=1e-5):
    # nn.InstanceNorm2d defaults: affine=False, no running stats (raft/extractor.py:26-30)
    return F.instance_norm(x, eps=eps)


def _prelu(sd, key, x):
    return F.prelu(x, sd[key + ".weight"])


def _lrelu(x):
    return F.leaky_relu(x, 0.1)


def resize(x, scale_factor):
    # modules/fi_utils.py:67-70
    return F.interpolate(x, scale_factor=scale_factor, mode="bilinear", align_corners=False)


def warp(img, flow):
    # modules/fi_utils.py:19-49 : bilinear, border padding, align_corners=True,
    # flow in pixels of the *input* tensor
    n, _, h, w = flow.shape
    gx = torch.linspace(-1.0, 1.0, w).view(1, 1, 1, w).expand(n, -1, h, -1)
    gy = torch.linspace(-1.0, 1.0, h).view(1, 1, h, 1).expand(n, -1, -1, w)
    fx = flow[:, 0:1] / ((img.shape[3] - 1.0) / 2.0)
    fy = flow[:, 1:2] / ((img.shape[2] - 1.0) / 2.0)
    g = torch.cat([gx + fx, gy + fy], 1).permute(0, 2, 3, 1)
    return F.grid_sample(img, g, mode="bilinear", padding_mode="border", align_corners=True)


def coords_grid(n, h, w):
    # raft/utils/utils.py:83-88 : channel 0 = x, channel 1 = y
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    return torch.stack([xs, ys], 0).float()[None].repeat(n, 1, 1, 1)


# --------------------------------------------------------------------------- RAFT encoders
def _residual_block(sd, p, x, norm, stride):
    # raft/extractor.py:6-58
    def nrm(name, t):
        if norm == "instance":
            return _inorm(t)
        return _bn(sd, p + "." + name, t)

    y = F.relu(nrm("norm1", _conv(sd, p + ".conv1", x, stride=stride, padding=1)))
    y = F.relu(nrm("norm2", _conv(sd, p + ".conv2", y, padding=1)))
    if stride != 1:
        x = _conv(sd, p + ".downsample.0", x, stride=stride)
        x = _inorm(x) if norm == "instance" else _bn(sd, p + ".downsample.1", x)
    return F.relu(x + y)


def basic_encoder(sd, p, x, norm):
    """raft/extractor.py:122-220 (BasicEncoder.forward); returns (out, [f@H/2, f@H/4, f@H/8])."""
    x = _conv(sd, p + ".conv1", x, stride=2, padding=3)
    x = _inorm(x) if norm == "instance" else _bn(sd, p + ".norm1", x)
    x = F.relu(x)
    feats = []
    for li, stride in ((1, 1), (2, 2), (3, 2)):
        x = _residual_block(sd, f"{p}.layer{li}.0", x, norm, stride)
        x = _residual_block(sd, f"{p}.layer{li}.1", x, norm, 1)
        feats.append(x)
    x = _conv(sd, p + ".conv2", x)
    return x, feats


# --------------------------------------------------------------------------- correlation
def corr_volume(f1, f2):
    # raft/corr.py:167-175 : fmap1^T fmap2 / sqrt(dim)
    b, d, h, w = f1.shape
    c = torch.matmul(f1.view(b, d, h * w).transpose(1, 2), f2.view(b, d, h * w))
    return c.view(b, h, w, 1, h, w) / math.sqrt(d)


def corr_pyramid(vol, levels=4):
    # raft/corr.py:127-142
    b, h1, w1, d, h2, w2 = vol.shape
    c = vol.reshape(b * h1 * w1, d, h2, w2)
    pyr = [c]
    for _ in range(levels - 1):
        c = F.avg_pool2d(c, 2, stride=2)
        pyr.append(c)
    return pyr


def _bilinear_sampler(img, coords):
    # raft/utils/utils.py:66-80 : pixel coords, zeros padding, align_corners=True
    H, W = img.shape[-2:]
    xg, yg = coords.split([1, 1], dim=-1)
    xg = 2 * xg / (W - 1) - 1
    yg = 2 * yg / (H - 1) - 1
    return F.grid_sample(img, torch.cat([xg, yg], -1), align_corners=True)


def corr_lookup(pyr, coords, r=4):
    """raft/corr.py:144-165.  NB the window is transposed: delta = stack(meshgrid(dy, dx))
    is added to (x, y), so the FIRST window axis moves x."""
    coords = coords.permute(0, 2, 3, 1)
    b, h1, w1, _ = coords.shape
    d = torch.linspace(-r, r, 2 * r + 1)
    delta = torch.stack(torch.meshgrid(d, d, indexing="ij"), -1).view(1, 2 * r + 1, 2 * r + 1, 2)
    out = []
    for i, c in enumerate(pyr):
        cen = coords.reshape(b * h1 * w1, 1, 1, 2) / 2**i
        s = _bilinear_sampler(c, cen + delta)
        out.append(s.view(b, h1, w1, -1))
    return torch.cat(out, -1).permute(0, 3, 1, 2).contiguous().float()


# --------------------------------------------------------------------------- RAFT update
def _raft_update(sd, p, net, inp, corr, flow, want_mask):
    """raft/update.py:131-154 (BasicUpdateBlock) + :94-112 (BasicMotionEncoder)
    + :35-73 (SepConvGRU) + :6-14 (FlowHead)."""
    e = p + ".encoder"
    cor = F.relu(_conv(sd, e + ".convc1", corr))
    cor = F.relu(_conv(sd, e + ".convc2", cor, padding=1))
    flo = F.relu(_conv(sd, e + ".convf1", flow, padding=3))
    flo = F.relu(_conv(sd, e + ".convf2", flo, padding=1))
    out = F.relu(_conv(sd, e + ".conv", torch.cat([cor, flo], 1), padding=1))
    x = torch.cat([inp, out, flow], 1)

    g = p + ".gru"
    h = net
    for sfx, pad in (("1", (0, 2)), ("2", (2, 0))):
        hx = torch.cat([h, x], 1)
        z = torch.sigmoid(F.conv2d(hx, sd[f"{g}.convz{sfx}.weight"], sd[f"{g}.convz{sfx}.bias"], padding=pad))
        r = torch.sigmoid(F.conv2d(hx, sd[f"{g}.convr{sfx}.weight"], sd[f"{g}.convr{sfx}.bias"], padding=pad))
        q = torch.tanh(
            F.conv2d(torch.cat([r * h, x], 1), sd[f"{g}.convq{sfx}.weight"], sd[f"{g}.convq{sfx}.bias"], padding=pad)
        )
        h = (1 - z) * h + z * q
    d = F.relu(_conv(sd, p + ".flow_head.conv1", h, padding=1))
    dflow = _conv(sd, p + ".flow_head.conv2", d, padding=1)
    mask = None
    if want_mask:
        m = F.relu(_conv(sd, p + ".mask.0", h, padding=1))
        mask = 0.25 * _conv(sd, p + ".mask.2", m)
    return h, mask, dflow


def convex_upsample(flow, mask):
    # raft/raft.py:86-97
    n, _, h, w = flow.shape
    mask = torch.softmax(mask.view(n, 1, 9, 8, 8, h, w), dim=2)
    up = F.unfold(8 * flow, [3, 3], padding=1).view(n, 2, 9, 1, 1, h, w)
    up = torch.sum(mask * up, dim=2).permute(0, 1, 4, 2, 5, 3)
    return up.reshape(n, 2, 8 * h, 8 * w)


def raft_forward(sd, p, image1, image2, iters=20, taps=None, tag=""):
    """raft/raft.py:99-167 with return_feat=True -> (flow_up, feats[1:], fmap1).
    Only the last iteration's mask/upsample is evaluated (the reference evaluates
    all 20 and discards 19, raft.py:156-167 -- same result)."""
    image1 = 2 * (image1 / 255.0) - 1.0
    image2 = 2 * (image2 / 255.0) - 1.0
    fm, _ = basic_encoder(sd, p + ".fnet", torch.cat([image1, image2], 0), "instance")
    b = image1.shape[0]
    fmap1, fmap2 = fm[:b], fm[b:]
    pyr = corr_pyramid(corr_volume(fmap1, fmap2))
    cnet, feats = basic_encoder(sd, p + ".cnet", image1, "batch")
    net, inp = torch.split(cnet, [128, 128], dim=1)
    net = torch.tanh(net)
    inp = torch.relu(inp)
    n, _, H, W = image1.shape
    coords0 = coords_grid(n, H // 8, W // 8)
    coords1 = coords_grid(n, H // 8, W // 8)
    if taps is not None:
        taps[tag + "fmap1"] = fmap1
        taps[tag + "net0"] = net
        taps[tag + "inp"] = inp
        taps[tag + "corr_l0"] = pyr[0]
        taps[tag + "corr_l3"] = pyr[3]
    mask = None
    for it in range(iters):
        corr = corr_lookup(pyr, coords1)
        flow = coords1 - coords0
        net, mask, dflow = _raft_update(sd, p + ".update_block", net, inp, corr, flow, it == iters - 1)
        coords1 = coords1 + dflow
        if taps is not None and it in (0, iters - 1):
            taps[f"{tag}corr_it{it}"] = corr
            taps[f"{tag}lowflow_it{it}"] = coords1 - coords0
            taps[f"{tag}net_it{it}"] = net
    flow_up = convex_upsample(coords1 - coords0, mask)
    return flow_up, feats[1:], fmap1


# --------------------------------------------------------------------------- BidirCorr
class BidirCorr:
    """raft/corr.py:23-93."""

    def __init__(self, f1, f2, radius=4):
        vol = corr_volume(f1, f2)
        volT = vol.clone().permute(0, 4, 5, 3, 1, 2)
        self.pyr = corr_pyramid(vol)
        self.pyrT = corr_pyramid(volT)
        self.r = radius

    def __call__(self, coords0, coords1):
        return corr_lookup(self.pyr, coords0, self.r), corr_lookup(self.pyrT, coords1, self.r)


# --------------------------------------------------------------------------- flow (un)normalisation
def normalize_flow(flows):
    # modules/fi_utils.py:52-60 : per-sample abs-max over both directions & components
    s = torch.max(torch.abs(flows).flatten(1), dim=-1)[0].reshape(-1, 1, 1, 1, 1)
    return (flows / s + 1.0) / 2.0, s


def unnormalize_flow(flows, s):
    # modules/fi_utils.py:63-64
    return (flows * 2.0 - 1.0) * s


# --------------------------------------------------------------------------- GIMM pieces
def cal_splatting_weights(sd, f01, f10):
    # gimmvfi_r.py:444-492
    b = f01.shape[0]
    fl = torch.cat([f01, f10], 0)
    g = sd["g_filter"]
    sm = F.conv3d(F.pad(torch.cat([fl**2, fl], 1), (1, 1, 1, 1), mode="reflect").unsqueeze(1), g).squeeze(1)
    sq_mean, mean = torch.split(sm, 2, dim=1)
    var = (sq_mean - mean**2).clamp(1e-9, None).sqrt().mean(1).unsqueeze(1)
    var01, var10 = var[:b], var[b:]
    err01 = (-warp(f10, f01) - f01).abs().mean(1).unsqueeze(1)
    err10 = (-warp(f01, f10) - f10).abs().mean(1).unsqueeze(1)
    w1 = 1 / (1 + err01 * sd["alpha_fe"]) + 1 / (1 + var01 * sd["alpha_v"])
    w2 = 1 / (1 + err10 * sd["alpha_fe"]) + 1 / (1 + var10 * sd["alpha_v"])
    return w1, w2


def _lateral(sd, p, x):
    # modules/fi_components.py:17-29
    y = _lrelu(_conv(sd, p + ".layers.0", x, padding=1))
    return _conv(sd, p + ".layers.2", y, padding=1) + x


def cnn_encoder(sd, x):
    # gimmvfi_r.py:84-97
    p = "cnn_encoder"
    x = _conv(sd, p + ".0", x, padding=1)
    x = _lrelu(_conv(sd, p + ".1", x, padding=1))
    for i in (3, 4, 5):
        x = _lateral(sd, f"{p}.{i}", x)
    x = _lrelu(x)
    return _conv(sd, p + ".7", x, padding=1, padding_mode="reflect")


def res_conv(sd, x):
    # gimmvfi_r.py:98-109
    p = "res_conv"
    x = _conv(sd, p + ".0", x, padding=1)
    x = _lrelu(_conv(sd, p + ".1", x, padding=1))
    x = _lrelu(_lateral(sd, p + ".3", x))
    return _conv(sd, p + ".5", x, padding=1, padding_mode="reflect")


def splat_sum(ten_in, flow):
    """Kernel softsplat_out, modules/softsplat.py:371-421 (summation splat with
    bilinear weights, OOB taps dropped, non-finite targets skipped)."""
    N, C, H, W = ten_in.shape
    gy, gx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    fx = gx[None] + flow[:, 0]
    fy = gy[None] + flow[:, 1]
    fin = torch.isfinite(fx) & torch.isfinite(fy)
    fx = torch.where(fin, fx, torch.zeros_like(fx))
    fy = torch.where(fin, fy, torch.zeros_like(fy))
    x0, y0 = torch.floor(fx), torch.floor(fy)
    x1, y1 = x0 + 1, y0 + 1
    out = ten_in.new_zeros(N, C, H * W)
    src = ten_in.reshape(N, C, H * W)
    for tx, ty, wgt in (
        (x0, y0, (x1 - fx) * (y1 - fy)),
        (x1, y0, (fx - x0) * (y1 - fy)),
        (x0, y1, (x1 - fx) * (fy - y0)),
        (x1, y1, (fx - x0) * (fy - y0)),
    ):
        ok = fin & (tx >= 0) & (tx < W) & (ty >= 0) & (ty < H)
        idx = (ty.clamp(0, H - 1) * W + tx.clamp(0, W - 1)).long().reshape(N, 1, H * W)
        out.scatter_add_(2, idx.expand(N, C, H * W), src * (wgt * ok).reshape(N, 1, H * W))
    return out.reshape(N, C, H, W)


def softsplat_linear_zeroeps(ten_in, flow, metric):
    # modules/softsplat.py:286-352 with strMode "linear-zeroeps"
    o = splat_sum(torch.cat([ten_in * metric, metric], 1), flow)
    nrm = o[:, -1:].clone()
    nrm[nrm == 0.0] = 1.0
    return o[:, :-1] / nrm


def hyponet_forward(sd, coord, pixel_latent, output_bias=0.5, w0=1.0):
    """modules/hyponet.py:71-146 (use_bias, normalize_weight, siren activation;
    no modulation).  coord (B,1,H',W',3) ordered (t,y,x); pixel_latent (B,H,W,32)."""
    B = coord.shape[0]
    shp = coord.shape[1:-1]
    lat = F.interpolate(pixel_latent.permute(0, 3, 1, 2), size=(shp[1], shp[2]), mode="bilinear").permute(0, 2, 3, 1)
    h = torch.cat([lat.reshape(B, -1, lat.shape[-1]), coord.reshape(B, -1, coord.shape[-1])], -1)
    n_layer = 5
    for i in range(n_layer):
        wb = sd[f"hyponet.params_dict.linear_wb{i}"]
        w = F.normalize(wb[:-1], dim=0)  # fan_in axis (dim=1 of the batched (b,n,m) tensor)
        h = h @ w + wb[-1:]
        if i < n_layer - 1:
            h = torch.sin(w0 * h)
    return (h + output_bias).view(B, *shp, -1)


# --------------------------------------------------------------------------- frame synthesis
def _convrelu(sd, p, x, padding):
    # modules/fi_components.py:32-54 (conv + per-channel PReLU)
    return _prelu(sd, p + ".1", _conv(sd, p + ".0", x, padding=padding))


def _resblock(sd, p, x, side):
    # modules/fi_components.py:97-154
    out = _convrelu(sd, p + ".conv1", x, 1)
    out = torch.cat([out[:, :-side], _convrelu(sd, p + ".conv2", out[:, -side:], 1)], 1)
    out = _convrelu(sd, p + ".conv3", out, 1)
    out = torch.cat([out[:, :-side], _convrelu(sd, p + ".conv4", out[:, -side:], 1)], 1)
    out = _conv(sd, p + ".conv5", out, padding=1)
    return _prelu(sd, p + ".prelu", x + out)


def init_decoder_upsample(sd, f):
    # modules/fi_components.py:234-244
    p = "amt_init_decoder.upsample"
    f = F.pixel_shuffle(f, 2)
    f = _convrelu(sd, p + ".1", f, 2)
    for i in (2, 3, 4, 5):
        f = _convrelu(sd, f"{p}.{i}", f, 1)
    return F.relu(_bn(sd, p + ".7", _conv(sd, p + ".6", f)))


def init_decoder(sd, f0, f1, flow0_in, flow1_in, img0, img1):
    # modules/fi_components.py:255-276
    f0 = init_decoder_upsample(sd, f0)
    f1 = init_decoder_upsample(sd, f1)
    f_in = torch.cat([warp(f0, flow0_in), warp(f1, flow1_in), flow0_in, flow1_in], 1)
    sf = f_in.shape[2] / img0.shape[2]
    i0, i1 = resize(img0, sf), resize(img1, sf)
    f_in = torch.cat([f_in, i0, i1, warp(i0, flow0_in), warp(i1, flow1_in)], 1)
    p = "amt_init_decoder.convblock"
    out = _convrelu(sd, p + ".0", f_in, 0)
    for i in (1, 2, 3):
        out = _resblock(sd, f"{p}.{i}", out, 64)
    out = _conv(sd, p + ".4", out, padding=1)
    return flow0_in + out[:, :2], flow1_in + out[:, 2:4], out[:, 4:]


def final_decoder_upsample(sd, f):
    # modules/fi_components.py:284-295
    p = "amt_final_decoder.upsample"
    f = F.pixel_shuffle(F.pixel_shuffle(f, 2), 2)
    f = _convrelu(sd, p + ".2", f, 2)
    for i in (3, 4, 5, 6):
        f = _convrelu(sd, f"{p}.{i}", f, 1)
    return F.relu(_bn(sd, p + ".8", _conv(sd, p + ".7", f)))


def final_decoder(sd, ft_, f0, f1, flow0, flow1, mask, img0, img1, n=3):
    # modules/fi_components.py:307-340
    f0 = final_decoder_upsample(sd, f0)
    f1 = final_decoder_upsample(sd, f1)
    flow0 = 4.0 * resize(flow0, 4.0)
    flow1 = 4.0 * resize(flow1, 4.0)
    ft_ = resize(ft_, 4.0)
    mask = resize(mask, 4.0)
    f_in = torch.cat(
        [ft_, warp(f0, flow0), warp(f1, flow1), flow0, flow1, mask, img0, img1, warp(img0, flow0), warp(img1, flow1)], 1
    )
    p = "amt_final_decoder.convblock"
    out = _convrelu(sd, p + ".0", f_in, 1)
    for i in (1, 2, 3):
        out = _resblock(sd, f"{p}.{i}", out, 64)
    out = _conv(sd, p + ".4", out, padding=1)
    d0, d1, dm, res = torch.split(out, [2 * n, 2 * n, n, 3 * n], 1)
    mask = torch.sigmoid(dm + mask.repeat(1, n, 1, 1))
    return d0 + flow0.repeat(1, n, 1, 1), d1 + flow1.repeat(1, n, 1, 1), mask, res


def amt_update(sd, p, net, flow, corr, scale_factor):
    # modules/fi_components.py:157-222
    if scale_factor is not None:
        net = resize(net, 1 / scale_factor)
    cor = _lrelu(_conv(sd, p + ".convc1", corr))
    cor = _lrelu(_conv(sd, p + ".convc2", cor, padding=1))
    flo = _lrelu(_conv(sd, p + ".convf1", flow, padding=3))
    flo = _lrelu(_conv(sd, p + ".convf2", flo, padding=1))
    inp = _lrelu(_conv(sd, p + ".conv", torch.cat([cor, flo], 1), padding=1))
    inp = torch.cat([inp, flow, net], 1)
    out = _conv(sd, p + ".gru.2", _lrelu(_conv(sd, p + ".gru.0", inp, padding=1)), padding=1)
    dnet = _conv(sd, p + ".feat_head.2", _lrelu(_conv(sd, p + ".feat_head.0", out, padding=1)), padding=1)
    dflow = _conv(sd, p + ".flow_head.2", _lrelu(_conv(sd, p + ".flow_head.0", out, padding=1)), padding=1)
    if scale_factor is not None:
        dnet = resize(dnet, scale_factor)
        dflow = scale_factor * resize(dflow, scale_factor)
    return dnet, dflow


def multi_flow_combine(sd, img0, img1, flow0, flow1, mask, img_res):
    # modules/fi_components.py:57-94 + gimmvfi_r.py:60-64
    b, c, h, w = flow0.shape
    nf = c // 2
    flow0 = flow0.reshape(b * nf, 2, h, w)
    flow1 = flow1.reshape(b * nf, 2, h, w)
    mask = mask.reshape(b * nf, 1, h, w)
    img_res = img_res.reshape(b * nf, 3, h, w)
    i0 = torch.stack([img0] * nf, 1).reshape(-1, 3, h, w)
    i1 = torch.stack([img1] * nf, 1).reshape(-1, 3, h, w)
    warps = (mask * warp(i0, flow0) + (1 - mask) * warp(i1, flow1) + img_res).reshape(b, nf, 3, h, w)
    x = warps.reshape(b, -1, h, w)
    res = _conv(sd, "amt_comb_block.2", _prelu(sd, "amt_comb_block.1", _conv(sd, "amt_comb_block.0", x, padding=3)), padding=3)
    return (warps.mean(1) + res + 1.0) / 2


def frame_synthesize(sd, img_xs, flow_t, features0, features1, corr_fn, cur_t, full_img=None, taps=None, tag=""):
    """gimmvfi_r.py:222-322."""
    B = img_xs.shape[0]
    img0 = 2 * img_xs[:, :, 0] - 1.0
    img1 = 2 * img_xs[:, :, 1] - 1.0
    H, W = img0.shape[-2:]
    lookup = coords_grid(B, H // 8, W // 8)
    ft0_full = flow_t * (-cur_t)
    ft1_full = flow_t * (1.0 - cur_t)
    ft0_inr4 = 0.25 * resize(ft0_full, 0.25)
    ft1_inr4 = 0.25 * resize(ft1_full, 0.25)
    flowt0_4, flowt1_4, ft_4 = init_decoder(sd, features0[-1], features1[-1], ft0_inr4, ft1_inr4, img0, img1)
    mask_4, ft_4 = ft_4[:, :1], ft_4[:, 1:]
    # warp_w_mask (gimmvfi_r.py:213-220, 259-261)
    f0u, f1u = 4 * resize(flowt0_4, 4), 4 * resize(flowt1_4, 4)
    m4 = resize(mask_4, 4).sigmoid()
    img_warp_4 = torch.clamp((m4 * warp(img0, f0u) + (1 - m4) * warp(img1, f1u) + 1.0) / 2, 0, 1)
    if taps is not None:
        taps[tag + "init_flow0_4"] = flowt0_4
        taps[tag + "init_ft_4"] = ft_4
    # _amt_corr_scale_lookup (gimmvfi_r.py:494-507), downsample=2
    fl0 = 0.5 * resize(flowt0_4, 0.5)
    fl1 = 0.5 * resize(flowt1_4, 0.5)
    c0, c1 = corr_fn(lookup + fl1 * (1.0 / (1.0 - cur_t)), lookup + fl0 * (1.0 / cur_t))
    corr_4 = torch.cat([c0, c1], 1)
    flow_4_lr = torch.cat([fl0, fl1], 1)
    dft, dfl = amt_update(sd, "amt_update4_low", ft_4, flow_4_lr, corr_4, 2.0)
    flowt0_4 = flowt0_4 + dfl[:, :2]
    flowt1_4 = flowt1_4 + dfl[:, 2:4]
    ft_4 = ft_4 + dft
    corr_4 = resize(corr_4, 2.0)
    dft, dfl = amt_update(sd, "amt_update4_high", ft_4, torch.cat([flowt0_4, flowt1_4], 1), corr_4, None)
    flowt0_4 = flowt0_4 + dfl[:, :2]
    flowt1_4 = flowt1_4 + dfl[:, 2:4]
    ft_4 = ft_4 + dft
    if taps is not None:
        taps[tag + "upd_flow0_4"] = flowt0_4
        taps[tag + "upd_ft_4"] = ft_4
    flowt0_1, flowt1_1, mask, img_res = final_decoder(
        sd, ft_4, features0[0], features1[0], flowt0_4, flowt1_4, mask_4, img0, img1
    )
    if taps is not None:
        taps[tag + "final_flow0_1"] = flowt0_1
        taps[tag + "final_mask"] = mask
        taps[tag + "final_res"] = img_res
    if full_img is not None:
        img0 = 2 * full_img[:, :, 0] - 1.0
        img1 = 2 * full_img[:, :, 1] - 1.0
        inv = img1.shape[2] / flowt0_1.shape[2]
        flowt0_1 = inv * resize(flowt0_1, inv)
        flowt1_1 = inv * resize(flowt1_1, inv)
        mask = resize(mask, inv)
        img_res = resize(img_res, inv)
    pred = torch.clamp(multi_flow_combine(sd, img0, img1, flowt0_1, flowt1_1, mask, img_res), 0, 1)
    hh, ww = img0.shape[-2:]
    return (
        pred,
        [flowt0_1.reshape(B, 3, 2, hh, ww), flowt0_4],
        [flowt1_1.reshape(B, 3, 2, hh, ww), flowt1_4],
        [img_warp_4],
    )


# --------------------------------------------------------------------------- top level
def sample_coord_input(batch_size, s_shape, t_ids, upsample_ratio=1.0):
    # modules/coord_sampler.py:15-43 ; coord_range (-1, 1)
    cs = [torch.tensor(t_ids, dtype=torch.float32) / 1.0]
    for n in s_shape:
        n = int(n * upsample_ratio)
        cs.append(-1.0 + 2.0 * ((0.5 + torch.arange(n)) / n))
    g = torch.stack(torch.meshgrid(*cs, indexing="ij"), -1)
    return g.unsqueeze(0).repeat(batch_size, 1, 1, 1, 1)


def forward(sd, img_xs, coord, t, ds_factor=None, iters=20, taps=None):
    """gimmvfi_r.py:324-407 (GIMMVFI_R.forward) for coord[i][1] is None."""
    assert isinstance(t, list) and isinstance(coord, list) and len(t) == len(coord)
    full = None
    if ds_factor is not None:
        full = img_xs.clone()
        img_xs = torch.stack([resize(img_xs[:, :, 0], ds_factor), resize(img_xs[:, :, 1], ds_factor)], 2)
    im0, im1 = 255 * img_xs[:, :, 0], 255 * img_xs[:, :, 1]
    p = "flow_estimator"
    # cal_bidirection_flow (gimmvfi_r.py:126-156)
    f01, feats0, fnet0 = raft_forward(sd, p, im0, im1, iters, taps, "r01_")
    f10, feats1, fnet1 = raft_forward(sd, p, im1, im0, iters, taps, "r10_")
    corr_fn = BidirCorr(_conv(sd, "amt_fproj", fnet0), _conv(sd, "amt_fproj", fnet1))
    feats0 = [_conv(sd, "amt_second_last_cproj", feats0[0]), _conv(sd, "amt_last_cproj", feats0[1])]
    feats1 = [_conv(sd, "amt_second_last_cproj", feats1[0]), _conv(sd, "amt_last_cproj", feats1[1])]
    return forward_after_flow(sd, img_xs, full, f01, f10, feats0, feats1, corr_fn, coord, t, taps)


def forward_after_flow(sd, img_xs, full, f01, f10, feats0, feats1, corr_fn, coord, t, taps=None):
    """Everything of GIMMVFI_{R,F}.forward behind the flow estimator (gimmvfi_r.py:143-156, 352-407;
    identical in gimmvfi_f.py:124-139, 329-384): flow normalisation, predict_flow, frame_synthesize."""
    nflows, scal = normalize_flow(torch.stack([f01, -f10], 2))
    flows = torch.stack([f01, f10], 2)
    # predict_flow (gimmvfi_r.py:158-211)
    w1, w2 = cal_splatting_weights(sd, f01, f10)
    pl0 = cnn_encoder(sd, nflows[:, :, 0])
    pl1 = cnn_encoder(sd, nflows[:, :, 1])
    if taps is not None:
        taps["f01"], taps["f10"] = f01, f10
        taps["w1"], taps["w2"] = w1, w2
        taps["pl0"], taps["pl1"] = pl0, pl1
        taps["feat0_4"], taps["feat0_8"] = feats0
    ninr = []
    for i, cur_t in enumerate(t):
        ct = cur_t.reshape(-1, 1, 1, 1)
        s0 = softsplat_linear_zeroeps(pl0, f01 * ct, w1)
        s1 = softsplat_linear_zeroeps(pl1, f10 * (1 - ct), w2)
        tmp = torch.cat([s0, s1], 1)
        tmp = tmp + res_conv(sd, torch.cat([pl0, pl1, tmp], 1))
        lat = tmp.permute(0, 2, 3, 1)
        c = coord[i]
        assert c[1] is None
        assert c[0][0, 0, 0, 0, 0] == cur_t[0].squeeze()
        out = hyponet_forward(sd, c[0], lat).permute(0, 4, 1, 2, 3)
        ninr.append(out)
        if taps is not None:
            taps[f"t{i}_splat0"] = s0
            taps[f"t{i}_latent"] = tmp
    flow_t = [unnormalize_flow(o, scal).squeeze() for o in ninr]
    preds, f0p, f1p, others = [], [], [], []
    for i in range(len(coord)):
        ft = flow_t[i]
        if ft.ndim != 4:
            ft = ft.unsqueeze(0)
        a, b_, c_, d = frame_synthesize(
            sd, img_xs, ft, feats0, feats1, corr_fn, t[i].reshape(-1, 1, 1, 1), full, taps, f"t{i}_"
        )
        preds.append(a)
        f0p.append(b_)
        f1p.append(c_)
        others.append(d)
    return {
        "imgt_pred": preds,
        "other_pred": others,
        "flowt0_pred": f0p,
        "flowt1_pred": f1p,
        "raft_flow": flows,
        "ninrflow": ninr,
        "nflow": nflows,
        "flowt": flow_t,
    }


def psnr(a, b):
    # src/SNU_FILM_arb.py:53-55
    mse = torch.mean((a.double() - b.double()) ** 2).item()
    return float("inf") if mse == 0 else -10.0 * math.log10(mse)
