"""GIMM-VFI-R inference pipeline on the HIP kernels (NHWC, one launch list per forward).

Mirrors reference generalizable_INR/gimmvfi_r.py:324-407 stage by stage; the stage names in
``taps`` are those of oracle/gimmvfi_r_oracle.py so tests can compare stage boundaries.

Work the reference performs but never uses is not executed (BASELINE.md "minimal" figure):
the duplicate fnet pass of the second RAFT direction, the mask head + convex upsampling of RAFT
iterations 1..19, and the t-independent decoder `upsample` stacks inside the timestep loop.
"""
import functools
import math
import os

import torch

from . import lib as L
from .ops import ConvLayer, InrMlp, PatchConvLayer, Runtime, TapSplitConvLayer, V, View

A = L  # activation / epilogue constants


def _fold_bn(w, b, sd, p, eps=1e-5):
    g = sd[p + ".weight"].float()
    beta = sd[p + ".bias"].float()
    mean = sd[p + ".running_mean"].float()
    var = sd[p + ".running_var"].float()
    s = g / torch.sqrt(var + eps)
    return w.float() * s.view(-1, 1, 1, 1), (b.float() - mean) * s + beta


def _on_device(fn):
    """Run the launch list with the runtime's device current: the ctypes launches use the device's current stream, and
    a model placed on a device that is not torch's current one (model.to("cuda:1") without set_device) would otherwise
    launch on the wrong context."""

    @functools.wraps(fn)
    def wrap(self, *a, **kw):
        if self.rt.on_gpu:
            with torch.cuda.device(self.rt.device):
                return fn(self, *a, **kw)
        return fn(self, *a, **kw)

    return wrap


class Engine:
    def __init__(self, rt: Runtime, sd, motion_only=False):
        """motion_only: build only the GIMM blocks (splat metric, cnn_encoder, res_conv, hypo-network) -- the
        reference's motion-only model `GIMM` (generalizable_INR/gimm.py)."""
        self.rt = rt
        self.motion_only = motion_only
        sd = {k: v.detach() for k, v in sd.items()}
        self.alpha_v = float(sd["alpha_v"].float().cpu().item())
        self.alpha_fe = float(sd["alpha_fe"].float().cpu().item())
        self.g9 = sd["g_filter"].float().reshape(9).contiguous().to(rt.device)
        # number of parallel launch sequences the flow estimator's recurrence is split into (sub-batches of images)
        self.raft_lanes = int(os.environ.get("GVFI_RAFT_LANES", "2"))
        # float GRU state in bf16 mode (gvfi_conv_params.state_f32), OFF by default.  Built to test whether the bf16 rounding
        # of the recurrent state is what costs GIMM-VFI-F its reference fidelity at 2K / 4K: it is not (demo-2K 32.47 dB
        # with the float state against 32.26 dB without, profiles/r3_f_policy.md -- the loss is the rounding of the update
        # block's MFMA operands, amplified by the un-trained recurrence) and it costs 1.3 % (326.2 vs 330.7 frames/s)
        self.gru_state_f32 = os.environ.get("GVFI_GRU_STATE_F32", "0") == "1"
        # frame synthesis of several timesteps of a pair as ONE batch (gimmvfi_r.py:376-396 runs them one by one): at 2K / 4K
        # a pair is B = 1 and the 1/8- and 1/4-resolution layers of the decoders launch 68-272 workgroups on 256 CUs; the
        # timesteps are independent once their flows exist.  Upper bound on T*B*H*W working-resolution pixels per batch
        # (activation memory: ~6 KB per pixel); 0 = one timestep at a time
        self.t_batch_pix = int(float(os.environ.get("GVFI_T_BATCH_PIX", "4.1e6")))
        # softmax splat as a deterministic gather over per-cell source lists (csrc/gimm_ops.hip); 0 = the float-atomic scatter
        self.splat_gather = os.environ.get("GVFI_SPLAT_GATHER", "1") != "0"
        # (round 4 built a third launch lane beside the recurrence for everything behind the encoders that does not depend on the
        # flow -- BidirCorrBlock's volumes, the context projections, the decoders' t-independent up-sampling stacks -- and measured
        # it neutral: 347.6 vs 348.3 frames/s at 448x256, 111.8 vs 111.5 at 2K, 96.4 vs 97.1 at 4K, profiles/r4_side_lane_ab.txt;
        # the switch also left side-stream allocations un-recorded on the main stream, so it is gone: that work
        # runs after the recurrence on the main stream)
        # launch sequences beside each other (round 5; same arithmetic, parallel branches of the hipGraph): the two encoders of the flow
        # estimator (RAFT fnet || cnet, FlowFormer's two Twins passes), and the flow-independent work behind the recurrence beside
        # the motion path.  profiles/r5_lanes_ab.txt, same box: R 448x256 351.1 -> 355.5 -> 359.5 frames/s (+1.3 %, +2.4 %), F 190.0 ->
        # 193.9 -> 194.7, R 4K 100.6 -> 101.9, F 4K 67.7 -> 69.1.  =0: A/B switches
        # ... and the independent branches inside the AMT update blocks (profiles/r5_synth_lanes_ab.txt: R 448x256 356.1 -> 359.2, +0.9 %)
        self.synth_lanes = os.environ.get("GVFI_SYNTH_LANES", "1") != "0"
        self.misc_lanes = os.environ.get("GVFI_MISC_LANES", "0") == "1"      # (A/B switch: two more small fork / joins, see _raft / _synthesize)
        self.enc_lanes = os.environ.get("GVFI_ENC_LANES", "1") != "0"
        self.post_lanes = os.environ.get("GVFI_POST_LANES", "1") != "0"
        self._lane_defaults = None
        self._tb_mem = {}
        self.layers = {}
        self._build(sd)

    def set_serial_launch(self, serial=True):
        """serial=True: every parallel launch sequence off -- the captured forward is a LINEAR graph that runs entirely on its
        launch stream (the branches of a forked graph run on internal streams of the HIP runtime).  3-7 % slower for one forward
        at a time; one of the two slot kinds StepsInFlight.calibrate() chooses between.  serial=False restores the switches."""
        names = ("raft_lanes", "synth_lanes", "misc_lanes", "enc_lanes", "post_lanes")
        if self._lane_defaults is None:
            self._lane_defaults = {n: getattr(self, n) for n in names}
        if serial:
            self.raft_lanes, self.synth_lanes, self.misc_lanes, self.enc_lanes, self.post_lanes = 1, False, False, False, False
        else:
            for n, v in self._lane_defaults.items():
                setattr(self, n, v)

    # ------------------------------------------------------------------ weight preparation
    def _add(self, name, w, b, **kw):
        self.layers[name] = ConvLayer(self.rt, w, b, **kw)

    def _patch_conv(self, sd, key, wdir=False):
        self.layers[key] = PatchConvLayer(self.rt, sd[key + ".weight"], sd[key + ".bias"], wdir=wdir)

    def _conv(self, sd, key, name=None, bn=None, slope=None, **kw):
        w, b = sd[key + ".weight"], sd[key + ".bias"]
        if bn is not None:
            w, b = _fold_bn(w, b, sd, bn)
        self._add(name or key, w, b, slope=None if slope is None else sd[slope], **kw)

    def _build_encoder(self, sd, p, batchnorm):
        # raft/extractor.py:122-166
        self._conv(sd, p + ".conv1", bn=(p + ".norm1") if batchnorm else None, stride=2)
        for li in (1, 2, 3):
            for bi in (0, 1):
                q = f"{p}.layer{li}.{bi}"
                s2 = li > 1 and bi == 0
                self._conv(sd, q + ".conv1", bn=(q + ".norm1") if batchnorm else None, stride=2 if s2 else 1)
                self._conv(sd, q + ".conv2", bn=(q + ".norm2") if batchnorm else None)
                if s2:
                    self._conv(sd, q + ".downsample.0", bn=(q + ".downsample.1") if batchnorm else None, stride=2)

    def _build_resblock(self, sd, p):
        for i in (1, 2, 3, 4):
            self._conv(sd, f"{p}.conv{i}.0", slope=f"{p}.conv{i}.1.weight")
        self._conv(sd, p + ".conv5", slope=p + ".prelu.weight")

    def _build_update(self, sd, p):
        self._conv(sd, f"{p}.convc1", cin_pad=self.rt.cp64(648))   # padded to a K chunk -> LDS-DMA kernel
        self._patch_conv(sd, f"{p}.convf1")        # 4 -> 128, 7x7: im2col + 1x1
        for k in ("convc2", "convf2", "conv", "gru.0", "gru.2", "feat_head.0", "feat_head.2",
                  "flow_head.0", "flow_head.2"):
            self._conv(sd, f"{p}.{k}")

    def _build_motion(self, sd):
        """gimmvfi_r.py:84-124 == gimm.py:36-78: motion encoder, latent refiner, hypo-network."""
        self._conv(sd, "cnn_encoder.0")
        self._conv(sd, "cnn_encoder.1")
        for i in (3, 4, 5):
            self._conv(sd, f"cnn_encoder.{i}.layers.0")
            self._conv(sd, f"cnn_encoder.{i}.layers.2")
        self._conv(sd, "cnn_encoder.7", pad_mode=L.PAD_REFLECT)
        self._conv(sd, "res_conv.0")
        self._conv(sd, "res_conv.1")
        self._conv(sd, "res_conv.3.layers.0")
        self._conv(sd, "res_conv.3.layers.2")
        self._conv(sd, "res_conv.5", pad_mode=L.PAD_REFLECT)
        # INR: weights L2-normalised along fan_in once (constant at inference)  modules/hyponet.py:124-128
        inr = []
        for i in range(5):
            wb = sd[f"hyponet.params_dict.linear_wb{i}"].float()
            w = torch.nn.functional.normalize(wb[:-1], dim=0)
            b = wb[-1].clone()
            if i == 4:
                b = b + 0.5  # output_bias, hyponet.py:143
            self._add(f"inr.{i}", w.t().reshape(w.shape[1], w.shape[0], 1, 1).contiguous(), b)
            inr.append((w.t().contiguous(), b))
        # bf16 mode: the five layers run as ONE kernel with register-resident activations (csrc/inr_mlp.hip)
        self.inr_mlp = InrMlp(self.rt, inr) if InrMlp.supported(self.rt, inr) else None

    def _build(self, sd):
        self._build_motion(sd)
        if self.motion_only:
            return
        self._build_flow(sd)
        self._build_synth(sd)

    def _build_flow(self, sd):
        """RAFT flow estimator + the 1x1 projections of its features (gimmvfi_r.py:44-53)."""
        fe = "flow_estimator"
        self._build_encoder(sd, fe + ".fnet", False)
        self._conv(sd, fe + ".fnet.conv2")
        self._build_encoder(sd, fe + ".cnet", True)
        w, b = sd[fe + ".cnet.conv2.weight"], sd[fe + ".cnet.conv2.bias"]
        self._add("cnet.out_net", w[:128], b[:128])   # tanh half   raft/raft.py:134-136
        self._add("cnet.out_inp", w[128:], b[128:])   # relu half
        u = fe + ".update_block"
        # wdir: the layers of the 20-iteration recurrence take the weights-direct variant of the LDS-DMA kernel
        self._conv(sd, f"{u}.encoder.convc1", cin_pad=self.rt.cp64(324), wdir=True)
        self._patch_conv(sd, f"{u}.encoder.convf1", wdir=True)   # 2 -> 128, 7x7: im2col + 1x1
        for k in ("encoder.convc2", "encoder.convf2", "encoder.conv", "flow_head.conv1"):
            self._conv(sd, f"{u}.{k}", wdir=True)
        for k in ("mask.0", "mask.2"):
            self._conv(sd, f"{u}.{k}")
        # 256 -> 2, 3x3 on the iteration's critical path: 1x1 to the 18 per-tap partial sums + tap gather
        k = f"{u}.flow_head.conv2"
        self.layers[k] = TapSplitConvLayer(self.rt, sd[k + ".weight"], sd[k + ".bias"])
        for n in ("1", "2"):
            wz, wr = sd[f"{u}.gru.convz{n}.weight"], sd[f"{u}.gru.convr{n}.weight"]
            bz, br = sd[f"{u}.gru.convz{n}.bias"], sd[f"{u}.gru.convr{n}.bias"]
            # hx = [h(128) | inp(128) | motion(126) | flow(2)]  (raft/update.py:143-144, 58-73).  The context features
            # `inp` do not change over the 20 iterations, so their share of every gate convolution is evaluated once
            # per forward ("ctx" layers, bias included) and enters the recurrence as a pre-activation term; the
            # per-iteration convolution only reads [h | motion | flow] (256 of the 384 channels).
            wzr, bzr = torch.cat([wz, wr], 0), torch.cat([bz, br], 0)
            wq, bq = sd[f"{u}.gru.convq{n}.weight"], sd[f"{u}.gru.convq{n}.bias"]
            keep = list(range(0, 128)) + list(range(256, 384))
            self._add(f"gru.zr{n}", wzr[:, keep], None, wdir=True)
            self._add(f"gru.q{n}", wq[:, keep], None, wdir=True)
            self._add(f"gru.zr{n}.ctx", wzr[:, 128:256], bzr)
            self._add(f"gru.q{n}.ctx", wq[:, 128:256], bq)
        for k in ("amt_last_cproj", "amt_second_last_cproj", "amt_fproj"):
            self._conv(sd, k)

    def _build_synth(self, sd):
        """Frame-synthesis decoders, identical in GIMM-VFI-R and -F (gimmvfi_r.py:55-64,113-124)."""
        p = "amt_init_decoder"
        for i in (1, 2, 3, 4, 5):
            self._conv(sd, f"{p}.upsample.{i}.0", slope=f"{p}.upsample.{i}.1.weight")
        self._conv(sd, p + ".upsample.6", bn=p + ".upsample.7")
        self._conv(sd, p + ".convblock.0.0", slope=p + ".convblock.0.1.weight", cin_pad=self.rt.cp64(272))
        for i in (1, 2, 3):
            self._build_resblock(sd, f"{p}.convblock.{i}")
        w, b = sd[p + ".convblock.4.weight"], sd[p + ".convblock.4.bias"]
        self._add("init.head5", w[:5], b[:5])     # [dflow0(2) dflow1(2) mask(1)]  fi_components.py:272-276
        self._add("init.ft", w[5:], b[5:])
        p = "amt_final_decoder"
        for i in (2, 3, 4, 5, 6):
            self._conv(sd, f"{p}.upsample.{i}.0", slope=f"{p}.upsample.{i}.1.weight")
        self._conv(sd, p + ".upsample.7", bn=p + ".upsample.8")
        self._conv(sd, p + ".convblock.0.0", slope=p + ".convblock.0.1.weight", cin_pad=self.rt.cp64(273))
        for i in (1, 2, 3):
            self._build_resblock(sd, f"{p}.convblock.{i}")
        self._conv(sd, p + ".convblock.4")
        self._build_update(sd, "amt_update4_low")
        self._build_update(sd, "amt_update4_high")
        self._conv(sd, "amt_comb_block.0", slope="amt_comb_block.1.weight")
        self._conv(sd, "amt_comb_block.2")

    # ------------------------------------------------------------------ building blocks
    def _enc(self, x, p, norm, B2):
        """raft/extractor.py:168-220.  x: prepared images [2B,H,W,8]."""
        rt, Ls = self.rt, self.layers
        n, H, W = x.shape[:3]
        inst = norm == "instance"
        # the InstanceNorm statistics of all 15 normalised convolutions live in ONE zero-filled arena (one fill per encoder
        # pass instead of one per layer): 64 + 4 x 64 + 5 x 96 + 5 x 128 = 1440 channels x (sum, sum of squares) per image
        # (64-bit fixed point: 4 float-sized words per image and channel; order-independent accumulation, gvfi_conv_params.stats)
        arena = rt.f32(n * 1440 * 4, zero=True) if inst else None
        used = [0]

        def cn(name, src, h, w, cout, res=None, final_relu=True, first=True):
            # conv (+norm) + relu ; for the block's second conv: relu(res + relu(norm(conv)))
            lay = Ls[name]
            if inst:
                raw = rt.act(n, h, w, cout)
                stats = arena[used[0]:used[0] + n * cout * 4].view(n, cout, 4)
                used[0] += n * cout * 4
                rt.conv(lay, src, raw, stats=stats)     # statistics fused into the convolution where possible
                return rt.instnorm(raw, cout, relu=final_relu, res=res, stats=stats if rt.last_stats_fused else None).t
            out = rt.act(n, h, w, cout)
            rt.conv(lay, src, out, act1=A.ACT_RELU if final_relu else A.ACT_NONE, res=res,
                    act2=A.ACT_RELU if res is not None else A.ACT_NONE)
            return out

        h, w = H // 2, W // 2
        y = cn(p + ".conv1", View(x, 0, 3), h, w, 64)
        feats = []
        cin = 64
        for li, dim in ((1, 64), (2, 96), (3, 128)):
            for bi in (0, 1):
                q = f"{p}.layer{li}.{bi}"
                s2 = li > 1 and bi == 0
                if s2:
                    h, w = h // 2, w // 2
                y1 = cn(q + ".conv1", y, h, w, dim)
                if s2:
                    sc = cn(q + ".downsample.0", y, h, w, dim, final_relu=False)
                else:
                    sc = y
                y = cn(q + ".conv2", y1, h, w, dim, res=sc)
            feats.append(y)
            cin = dim
        return y, feats, (h, w)

    def _corr_pyramids(self, fa, fb, n, h8, w8):
        """All-pairs volume fa^T fb / sqrt(256) + 3 pooled levels (raft/corr.py:127-142,167-175)
        as a grouped 1x1 'convolution' whose weights are the other frame's features."""
        rt = self.rt
        P8 = h8 * w8
        vol = rt.f32(n * P8, P8)
        out = vol.view(n, h8, w8, P8)
        # fb = one tensor of n partner maps, or a list of (first image, count, tensor) pieces: the bidirectional callers pair image
        # i with image i +- B of the SAME tensor, which is two launches on its halves instead of a concatenated copy of it
        pieces = fb if isinstance(fb, list) else [(0, n, fb)]
        for i0, cnt, wt in pieces:
            rt.conv(None, fa[i0:i0 + cnt], View(out[i0:i0 + cnt]), groups=cnt, w_group_stride=P8 * wt.shape[-1], w_raw=wt, cout=P8,
                    out_scale=1.0 / math.sqrt(256.0))
        pyr = [vol]
        hh, ww = h8, w8
        for _ in range(3):
            pyr.append(rt.avgpool2(pyr[-1], n * P8, hh, ww))
            hh, ww = hh // 2, ww // 2
        return pyr

    def _resblock(self, p, x, C, side=64):
        """modules/fi_components.py:97-154 with the channel concatenations expressed as two-source convs."""
        rt, Ls = self.rt, self.layers
        n, h, w = x.shape[:3]
        o1 = rt.act(n, h, w, C)
        rt.conv(Ls[p + ".conv1.0"], x, o1, act1=A.ACT_PRELU)
        s2 = rt.act(n, h, w, side)
        rt.conv(Ls[p + ".conv2.0"], View(o1, C - side, side), s2, act1=A.ACT_PRELU)
        o3 = rt.act(n, h, w, C)
        rt.conv(Ls[p + ".conv3.0"], View(o1, 0, C - side), o3, x1=s2, act1=A.ACT_PRELU)
        s4 = rt.act(n, h, w, side)
        rt.conv(Ls[p + ".conv4.0"], View(o3, C - side, side), s4, act1=A.ACT_PRELU)
        out = rt.act(n, h, w, C)
        lay = Ls[p + ".conv5"]
        rt.conv(lay, View(o3, 0, C - side), out, x1=s4, res=x, act2=A.ACT_PRELU, slope2=lay.slope)
        return out

    # ------------------------------------------------------------------ RAFT (both directions batched)
    @staticmethod
    def _seq_index(B, device):
        """Image i of the [frame0 of pair 0..B-1 | frame1 of pair 0..B-1] layout as an index into the B+1 distinct frames
        of B consecutive pairs (pair b = frames b, b+1)."""
        return torch.cat([torch.arange(B, device=device), torch.arange(1, B + 1, device=device)])

    def _raft(self, imgA, B, iters, taps, seq=False, side=None):
        """side(fmap, cfeats): the caller's launch sequence that only needs the encoders' outputs; it runs behind the update
        iterations on the main stream."""
        rt, Ls = self.rt, self.layers
        n = 2 * B
        H, W = imgA.shape[1:3]
        fe = "flow_estimator"
        if seq and B > 1:
            # consecutive pairs share frames: both encoders are per image (InstanceNorm / folded BatchNorm),
            # so they run on the B+1 distinct frames and the 2B-image layout is an index gather of their outputs
            imgU = torch.cat([imgA[:B], imgA[n - 1:n]], 0)
            idx = self._seq_index(B, imgA.device)
            f128, _, (h8, w8) = self._enc(imgU, fe + ".fnet", "instance", B + 1)
            fmapU = rt.act(B + 1, h8, w8, 256)
            rt.conv(Ls[fe + ".fnet.conv2"], f128, fmapU)
            fmap = fmapU[idx]
            c128, cfeats, _ = self._enc(imgU, fe + ".cnet", "batch", B + 1)
            c128, cfeats = c128[idx], [f[idx] for f in cfeats]
        else:
            # the two encoders read the same images and meet only in the recurrence: they run as two parallel launch sequences
            # (their 1/4- and 1/8-resolution layers launch 224-448 workgroups each; GVFI_ENC_LANES=0: A/B switch); the context
            # encoder's outputs are allocated on the side stream and handed to the main one explicitly
            k_enc = 2 if (self.enc_lanes and taps is None and rt.on_gpu) else 1
            res = {}
            with rt.lanes(k_enc) as lanes:
                with lanes[0]:
                    f128, _, (h8, w8) = self._enc(imgA, fe + ".fnet", "instance", n)
                    fmap = rt.act(n, h8, w8, 256)
                    rt.conv(Ls[fe + ".fnet.conv2"], f128, fmap)
                with lanes[k_enc - 1]:
                    res["c128"], res["cfeats"], _ = self._enc(imgA, fe + ".cnet", "batch", n)
            c128, cfeats = res["c128"], res["cfeats"]
            if k_enc > 1 and rt.ev_log is None and not torch.cuda.is_current_stream_capturing():
                cur = torch.cuda.current_stream(rt.device)
                for t_ in (c128, *cfeats):      # (allocated while the side stream was current: the main stream reads them from here on)
                    (t_.t if isinstance(t_, View) else t_).record_stream(cur)
        hA = rt.act(n, h8, w8, 128)
        hB = rt.act(n, h8, w8, 128)
        xbuf = rt.act(n, h8, w8, 256)     # [inp(128) | motion(126) | flow(2)]  raft/update.py:143-144
        # bf16 mode: the GRU state h (and the gate z) live in FLOAT beside the bf16 operand copies the convolutions read --
        # the state is an accumulator over the iterations, the one place where bf16 rounding would pile up
        sf = self.gru_state_f32 and rt.precision == "bf16"
        h32A = rt.f32(n, h8, w8, 128) if sf else None
        h32B = rt.f32(n, h8, w8, 128) if sf else None
        u = fe + ".update_block"
        # context share of the four gate convolutions (f32, evaluated once): see _build
        ctx = {key: rt.f32(n, h8, w8, Ls[key + ".ctx"].cout) for key in ("gru.zr1", "gru.q1", "gru.zr2", "gru.q2")}
        # the matching features feed the correlation pyramids, the context features the initial state / the constant GRU input and
        # its four pre-activation terms: two independent launch sequences (GVFI_MISC_LANES=1, A/B switch; everything the second
        # one writes is allocated above)
        k_pre = 2 if (self.misc_lanes and taps is None and rt.on_gpu) else 1
        with rt.lanes(k_pre) as lanes:
            with lanes[k_pre - 1]:
                if sf:
                    rt.conv(Ls["cnet.out_net"], c128, h32A, act1=A.ACT_TANH)
                    rt.copy(h32A, hA, 128)
                else:
                    rt.conv(Ls["cnet.out_net"], c128, hA, act1=A.ACT_TANH)
                rt.conv(Ls["cnet.out_inp"], c128, View(xbuf, 0, 128), act1=A.ACT_RELU)
                for key in ctx:
                    rt.conv(Ls[key + ".ctx"], View(xbuf, 0, 128), ctx[key])
            with lanes[0]:
                # correlation pyramids of both directions (0->1 for images [0,B), 1->0 for [B,2B)): image i against its partner (i +- B)
                pyr_ab = self._corr_pyramids(fmap, [(0, B, fmap[B:]), (B, B, fmap[:B])], n, h8, w8)
        pyr_a = [p[:B * h8 * w8] for p in pyr_ab]
        if taps is not None:
            taps["r01_fmap1"] = fmap[:B]
            taps["r01_net0"] = hA[:B].clone()
            taps["r01_inp"] = xbuf[:B, ..., :128].clone()
            taps["r01_corr_l0"] = pyr_a[0]
            taps["r01_corr_l3"] = pyr_a[3]
        coords = rt.coords_init(n, h8, w8)
        coords_alt = rt.f32(n, h8, w8, 2)
        corrf = rt.act(n, h8, w8, 324, zero=True, pitch=rt.cp64(324), zero_pad_only=True, once="raft.corrf")
        flow8 = rt.act(n, h8, w8, 2, zero=True, once="raft.flow8")
        c1 = rt.act(n, h8, w8, 256)
        corflo = rt.act(n, h8, w8, 256)
        f1 = rt.act(n, h8, w8, 128)
        zbuf = rt.f32(n, h8, w8, 128) if sf else rt.act(n, h8, w8, 128)
        rh = rt.act(n, h8, w8, 128)
        fh = rt.act(n, h8, w8, 256)
        fcol = rt.act(n, h8, w8, Ls[u + ".encoder.convf1"].kpad)
        fpart = rt.f32(n, h8, w8, 20)   # 9 taps x 2 partial sums of the flow head (+ pad)
        P8 = h8 * w8

        def chain(a, b):
            """The 20 update iterations of images [a, b) (raft/raft.py:144-161): every tensor of the recurrence is
            per image, so sub-batches are independent launch sequences."""
            m = b - a
            pyr_s = [p[a * P8:b * P8] for p in pyr_ab]
            co, cf, fl, xb = coords[a:b], corrf[a:b], flow8[a:b], xbuf[a:b]
            c1_, cfl, f1_, zb, rh_, fh_ = c1[a:b], corflo[a:b], f1[a:b], zbuf[a:b], rh[a:b], fh[a:b]
            ha, hb, fc, fp = hA[a:b], hB[a:b], fcol[a:b], fpart[a:b]
            h32 = (h32A[a:b], h32B[a:b]) if sf else (None, None)
            cx = {k: v[a:b] for k, v in ctx.items()}
            fused = rt.fuse_seam and taps is None
            co_home, co_alt = co, coords_alt[a:b]     # (the fused seam updates the coordinates out of place: ping-pong)
            tapl, patl = Ls[u + ".flow_head.conv2"], Ls[u + ".encoder.convf1"]
            for it in range(iters):
                if fused:
                    # one launch: coords1 += flow-head output of the previous iteration, flow activation, 7x7 patch of it
                    co, co_alt = (rt.flow_step(tapl, patl, fp, co, fl, View(xb, 254, 2), fc, coords_out=co_alt), co) if it > 0 else \
                        (rt.flow_step(tapl, patl, None, co, fl, View(xb, 254, 2), fc), co_alt)
                rt.corr_lookup(pyr_s, co, cf, m, h8, w8, h8, w8)
                if not fused:
                    rt.flow_pack(co, fl, View(xb, 254, 2))
                # (measured r2: running the flow branch of the motion encoder as a parallel graph branch is worth
                # nothing -- the CUs already hold the 2 workgroups their LDS admits -- and forks nested inside lanes()
                # crash hipStreamEndCapture on ROCm 7.0, so the branches are launched in sequence)
                if fused:
                    # the two branches of the motion encoder (raft/update.py:94-112) meet only in `conv`: convc1 || convf1 and
                    # convc2 || convf2 are one launch each (gvfi_conv2d_pair: the flow branch's 224 workgroups are the tail of
                    # the correlation branch's grid instead of two more launches on the iteration's critical path)
                    rt.conv_pair(dict(layer=Ls[u + ".encoder.convc1"], x0=cf, out=c1_, act1=A.ACT_RELU),
                                 dict(layer=patl.inner, x0=fc, out=f1_, act1=A.ACT_RELU))
                    rt.conv_pair(dict(layer=Ls[u + ".encoder.convc2"], x0=c1_, out=View(cfl, 0, 192), act1=A.ACT_RELU),
                                 dict(layer=Ls[u + ".encoder.convf2"], x0=f1_, out=View(cfl, 192, 64), act1=A.ACT_RELU))
                else:
                    rt.conv(Ls[u + ".encoder.convc1"], cf, c1_, act1=A.ACT_RELU)
                    rt.conv(Ls[u + ".encoder.convc2"], c1_, View(cfl, 0, 192), act1=A.ACT_RELU)
                    rt.patch_conv(patl, View(fl, 0, 2), f1_, scratch=fc, act1=A.ACT_RELU)
                    rt.conv(Ls[u + ".encoder.convf2"], f1_, View(cfl, 192, 64), act1=A.ACT_RELU)
                rt.conv(Ls[u + ".encoder.conv"], cfl, View(xb, 128, 126), act1=A.ACT_RELU)
                hc, hn = ha, hb
                sc, sn = h32
                for nn_ in ("1", "2"):  # SepConvGRU horizontal then vertical  raft/update.py:58-73
                    xm = View(xb, 128, 128)   # [motion(126) | flow(2)]
                    # the z | r and q convolutions with their gate epilogues (one launch per half was built in round 5 and measured
                    # neutral: tools/experiments/csrc/gru_fused.hip)
                    rt.conv(Ls["gru.zr" + nn_], hc, zb, x1=xm, epi=A.EPI_GRU_ZR, y2=rh_, aux0=sc if sf else hc,
                            res=cx["gru.zr" + nn_], state_f32=sf)
                    rt.conv(Ls["gru.q" + nn_], rh_, hn, x1=xm, epi=A.EPI_GRU_Q, aux0=sc if sf else hc, aux1=zb,
                            y2=sn if sf else None, res=cx["gru.q" + nn_], state_f32=sf)
                    hc, hn = hn, hc
                    sc, sn = sn, sc
                # after two passes the state is back in hA
                rt.conv(Ls[u + ".flow_head.conv1"], ha, fh_, act1=A.ACT_RELU)
                if fused and it + 1 < iters:
                    rt.conv(tapl.inner, fh_, View(fp, 0, 18))      # per-tap partial sums; summed by the next flow_step
                else:
                    rt.tap_split_conv(tapl, fh_, View(co_home), res=View(co), scratch=fp)   # coords1 += delta (-> home tensor)
                if taps is not None and it in (0, iters - 1):
                    taps[f"r01_corr_it{it}"] = corrf[:B, ..., :324].clone()
                    taps[f"r01_net_it{it}"] = hA[:B].clone()
                    taps[f"r01_coords_it{it}"] = coords[:B].clone()

        # k sub-batches of images run their recurrences as k parallel launch sequences (hipGraph branches): each launch
        # of the recurrence under-fills the chip (M = n*h8*w8 rows -> 224-448 workgroups with serial phases), so
        # independent sequences overlap each other's prologues, tails and epilogues.  Same arithmetic per image.
        k = 1 if taps is not None else max(1, min(self.raft_lanes, n))
        with rt.lanes(k) as lanes:
            for i in range(k):
                with lanes[i]:
                    chain(i * n // k, (i + 1) * n // k)
        if side is not None:
            side(fmap, cfeats)
        rt.conv(Ls[u + ".mask.0"], hA, fh, act1=A.ACT_RELU)
        mask = rt.f32(n, h8, w8, 576)
        rt.conv(Ls[u + ".mask.2"], fh, mask, out_scale=0.25)
        flow_up = rt.convex_upsample(coords, mask)
        return flow_up, fmap, cfeats, (h8, w8)

    # ------------------------------------------------------------------ AMT-style update block
    def _amt_update(self, p, net, flow4_f32, corr, B, h, w, st4, ft_4, low):
        """modules/fi_components.py:199-222.  net: [B,h,w,128] (already down-sampled for the low block)."""
        rt, Ls = self.rt, self.layers
        # (GVFI_SYNTH_LANES=0: A/B switch) the block's independent branches -- correlation branch || flow branch of its motion
        # encoder, feature head || flow head -- as two parallel launch sequences (every tensor that crosses a join is allocated
        # outside the branches; a branch's temporaries live and die on its own stream)
        k = 2 if (self.synth_lanes and rt.on_gpu) else 1
        c1 = rt.act(B, h, w, 256)
        corflo = rt.act(B, h, w, 256)
        flo = rt.act(B, h, w, 4, zero=True, once=p + ".flo")
        with rt.lanes(k) as lanes:
            with lanes[0]:
                rt.conv(Ls[p + ".convc1"], corr, c1, act1=A.ACT_LRELU)
                rt.conv(Ls[p + ".convc2"], c1, View(corflo, 0, 192), act1=A.ACT_LRELU)
            with lanes[k - 1]:
                rt.copy(View(flow4_f32, 0, 4), View(flo, 0, 4), 4)
                f1 = rt.act(B, h, w, 128)
                rt.patch_conv(Ls[p + ".convf1"], View(flo, 0, 4), f1, act1=A.ACT_LRELU)
                rt.conv(Ls[p + ".convf2"], f1, View(corflo, 192, 64), act1=A.ACT_LRELU)
                del f1
        inp = rt.act(B, h, w, 192)          # [inp(188) | flow(4)] ; net(128) is the second conv source
        rt.conv(Ls[p + ".conv"], corflo, View(inp, 0, 188), act1=A.ACT_LRELU)
        rt.copy(View(flow4_f32, 0, 4), View(inp, 188, 4), 4)
        g0 = rt.act(B, h, w, 192)
        rt.conv(Ls[p + ".gru.0"], inp, g0, x1=net, act1=A.ACT_LRELU)
        out = rt.act(B, h, w, 192)
        rt.conv(Ls[p + ".gru.2"], g0, out)
        with rt.lanes(k) as lanes:
            with lanes[0]:
                fh0 = rt.act(B, h, w, 192)
                rt.conv(Ls[p + ".feat_head.0"], out, fh0, act1=A.ACT_LRELU)
                if low:
                    dnet = rt.act(B, h, w, 128)
                    rt.conv(Ls[p + ".feat_head.2"], fh0, dnet)
                    up = rt.resize(dnet, 128, 2.0)
                    rt.copy(up, ft_4, 128, add=ft_4)
                else:
                    rt.conv(Ls[p + ".feat_head.2"], fh0, ft_4, res=ft_4)
            with lanes[k - 1]:
                lh0 = rt.act(B, h, w, 192)
                rt.conv(Ls[p + ".flow_head.0"], out, lh0, act1=A.ACT_LRELU)
                if low:
                    dflow = rt.f32(B, h, w, 4)
                    rt.conv(Ls[p + ".flow_head.2"], lh0, dflow)
                    upf = rt.resize(dflow, 4, 2.0, mul=2.0)
                    rt.copy(upf, View(st4, 0, 4), 4, add=View(st4, 0, 4))
                    del dflow, upf
                else:
                    rt.conv(Ls[p + ".flow_head.2"], lh0, View(st4, 0, 4), res=View(st4, 0, 4))
                del lh0

    # ------------------------------------------------------------------ motion INR (shared by GIMM-VFI-R and GIMM)
    def _start_side(self):
        """Forks the deferred side sequence(s) of forward() -- see `deferred` there -- from the current stream."""
        pend, self._side_pending = getattr(self, "_side_pending", []), []
        if not pend:
            return
        rt = self.rt
        cur = torch.cuda.current_stream(rt.device)
        ss = rt._lane_streams(cur, 1)[0]
        ss.wait_stream(cur)
        held = []
        with torch.cuda.stream(ss):
            for fn, a in pend:
                fn(*a)
                held.append(a)
        # the tensors the side sequence READS (allocated on this stream: feature maps, context features) must outlive its
        # kernels: once the caller drops them the caching allocator -- also the capture pool -- may hand their blocks to a
        # later allocation on THIS stream with no ordering against the side stream's reads.  They are held until the join.
        self._side_join = (cur, ss, held)

    def _motion_encode(self, nfA, f01, f10, B, H, W):
        """Splat metric (gimmvfi_r.py:444-492 == gimm.py:80-127) and the motion encoder on both normalised flows
        (gimmvfi_r.py:164-167 == gimm.py:139-140).  nfA: [2B,H,W,2(+pad)] activation, f01/f10: [B,H,W,2] f32.
        Returns z0, z1 [B,H,W] f32 and latcat [B,H,W,64] = [pl0 | pl1 | (splat0) | (splat1)]."""
        rt, Ls, lib, st = self.rt, self.layers, self.rt.lib, self.rt.stream
        n = 2 * B
        z0, z1 = rt.f32(B, H, W), rt.f32(B, H, W)
        rt._chk(lib.splat_weights(f01.data_ptr(), f10.data_ptr(), self.g9.data_ptr(), self.alpha_v, self.alpha_fe,
                                  z0.data_ptr(), z1.data_ptr(), B, H, W, st()), "splat_weights")
        e0 = rt.act(n, H, W, 16)
        rt.conv(Ls["cnn_encoder.0"], View(nfA, 0, 2), e0)
        e = rt.act(n, H, W, 32)
        rt.conv(Ls["cnn_encoder.1"], e0, e, act1=A.ACT_LRELU)
        tA = rt.act(n, H, W, 32)
        for i in (3, 4, 5):
            rt.conv(Ls[f"cnn_encoder.{i}.layers.0"], e, tA, act1=A.ACT_LRELU)
            e2 = rt.act(n, H, W, 32)
            rt.conv(Ls[f"cnn_encoder.{i}.layers.2"], tA, e2, res=e, act2=A.ACT_LRELU if i == 5 else A.ACT_NONE)
            e = e2
        latcat = rt.act(B, H, W, 64)   # [pl0 | pl1 | splat0 | splat1]  gimmvfi_r.py:187-192
        rt.conv(Ls["cnn_encoder.7"], e[:B], View(latcat, 0, 16))
        rt.conv(Ls["cnn_encoder.7"], e[B:], View(latcat, 16, 16))
        return z0, z1, latcat

    def _motion_inr(self, latcat, f01, f10, z0, z1, cg, tv, B, H, W, taps=None, tag=""):
        """One timestep: softmax-splat both latents to t, refine, evaluate the hypo-network on the coordinate grid
        cg [B,1,Hc,Wc,3] (gimmvfi_r.py:171-205 == gimm.py:147-179).  Returns the normalised flow [B,Hc,Wc,2] f32."""
        rt, Ls, lib, st = self.rt, self.layers, self.rt.lib, self.rt.stream
        HW = H * W
        Hc, Wc = cg.shape[2], cg.shape[3]
        # softmax splatting of the two latents to time t   gimmvfi_r.py:171-193
        if self.splat_gather:
            # deterministic gather over per-cell source lists, both directions and the normalisation in one launch
            head = torch.full((2, B, H + 1, W + 1), -1, dtype=torch.int32, device=rt.device)
            nxt = torch.empty((2, B, H, W), dtype=torch.int32, device=rt.device)
            rt._chk(lib.softsplat_lists(f01.data_ptr(), f10.data_ptr(), tv.data_ptr(), head.data_ptr(), nxt.data_ptr(), B, H, W,
                                        st()), "softsplat_lists")
            rt._chk(lib.softsplat_gather(latcat.data_ptr(), latcat.shape[-1], f01.data_ptr(), f10.data_ptr(), z0.data_ptr(),
                                         z1.data_ptr(), tv.data_ptr(), head.data_ptr(), nxt.data_ptr(), View(latcat, 32, 32).ptr,
                                         latcat.shape[-1], B, H, W, rt.dtype, st()), "softsplat_gather")
        else:
            for d, (fl, zz) in enumerate(((f01, z0), (f10, z1))):
                acc = rt.f32(B, H, W, 17, zero=True)
                rt._chk(lib.softsplat_accum(View(latcat, 16 * d, 16).ptr, latcat.shape[-1], 16, fl.data_ptr(),
                                            zz.data_ptr(), tv.data_ptr(), d, acc.data_ptr(), B, H, W, rt.dtype, st()),
                        "softsplat_accum")
                rt._chk(lib.softsplat_normalize(acc.data_ptr(), 16, View(latcat, 32 + 16 * d, 16).ptr,
                                                latcat.shape[-1], B * HW, rt.dtype, st()), "softsplat_normalize")
        r0 = rt.act(B, H, W, 32)
        rt.conv(Ls["res_conv.0"], latcat, r0)
        r1 = rt.act(B, H, W, 64)
        rt.conv(Ls["res_conv.1"], r0, r1, act1=A.ACT_LRELU)
        r2 = rt.act(B, H, W, 64)
        rt.conv(Ls["res_conv.3.layers.0"], r1, r2, act1=A.ACT_LRELU)
        r3 = rt.act(B, H, W, 64)
        rt.conv(Ls["res_conv.3.layers.2"], r2, r3, res=r1, act2=A.ACT_LRELU)
        lat = rt.act(B, H, W, 32)
        rt.conv(Ls["res_conv.5"], r3, lat, res=View(latcat, 32, 32))
        if taps is not None:
            taps[f"{tag}splat0"] = latcat[..., 32:48].clone()
            taps[f"{tag}latent"] = lat
        # INR   modules/hyponet.py:71-146
        if (Hc, Wc) != (H, W):
            lat = rt.resize(lat, 32, None, size=(Hc, Wc)).t
        ninr = rt.f32(B, Hc, Wc, 2)
        if self.inr_mlp is not None:
            rt.inr_mlp(self.inr_mlp, View(lat, 0, 32), cg, ninr)
        else:
            xin = rt.act(B, Hc, Wc, 35, zero=True, once="inr.xin")
            rt._chk(lib.inr_pack(lat.data_ptr(), lat.shape[-1], 32, cg.data_ptr(), xin.data_ptr(), xin.shape[-1],
                                 xin.shape[-1], B * Hc * Wc, rt.dtype, st()), "inr_pack")
            hcur = View(xin, 0, 35)
            for li in range(4):
                hn = rt.act(B, Hc, Wc, 128)
                rt.conv(Ls[f"inr.{li}"], hcur, hn, act1=A.ACT_SIN)
                hcur = hn
            rt.conv(Ls["inr.4"], hcur, ninr)
        return ninr

    @torch.no_grad()
    @_on_device
    def forward_motion(self, xs, coord, ori_flow, timesteps):
        """The reference's motion-only model GIMM.forward (gimm.py:129-214, keep_xs_shape=True): xs = normalised flows
        (B,2,2,H,W) [channel, frame], ori_flow = raw flows (B,2,2,H,W) [frame 0: 0->1, frame 1: 1->0], coord /
        timesteps one tensor each or equally long lists.  Returns (a list of) normalised flows (B,2,1,H',W')."""
        rt = self.rt
        xs = xs.to(device=rt.device, dtype=torch.float32)
        ori_flow = ori_flow.to(device=rt.device, dtype=torch.float32)
        B, _, _, H, W = xs.shape
        # layout plumbing only: NCHW planes -> the NHWC tensors the kernels read
        f01 = ori_flow[:, :, 0].permute(0, 2, 3, 1).contiguous()
        f10 = ori_flow[:, :, 1].permute(0, 2, 3, 1).contiguous()
        nfA = rt.act(2 * B, H, W, 2, zero=True)
        nfA[:B, ..., :2] = xs[:, :, 0].permute(0, 2, 3, 1).to(nfA.dtype)
        nfA[B:, ..., :2] = xs[:, :, 1].permute(0, 2, 3, 1).to(nfA.dtype)
        z0, z1, latcat = self._motion_encode(nfA, f01, f10, B, H, W)
        single = not isinstance(timesteps, list)
        if single:
            coord, timesteps = [coord], [timesteps]
        assert isinstance(coord, list) and len(coord) == len(timesteps)
        outs = []
        for c, cur_t in zip(coord, timesteps):
            cg = c.to(device=rt.device, dtype=torch.float32).contiguous()
            tv = cur_t.to(device=rt.device, dtype=torch.float32).reshape(-1).contiguous()
            if tv.numel() == 1 and B > 1:
                tv = tv.expand(B).contiguous()     # gimm.py:184 broadcasts a scalar time over the batch
            ninr = self._motion_inr(latcat, f01, f10, z0, z1, cg, tv, B, H, W)
            outs.append(rt.nhwc_to_nchw(ninr, 2).unsqueeze(2))   # (B,2,1,H',W')
        return outs[0] if single else outs

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    @_on_device
    def forward(self, img_xs, coord, t, iters=20, ds_factor=None, taps=None, want_aux=True, seq=False):
        """seq: the B pairs of img_xs are consecutive pairs of one frame sequence (img_xs[b, :, 1] IS img_xs[b+1, :, 0]):
        per-frame encoder work is done once per distinct frame.  Same outputs."""
        rt, Ls, lib = self.rt, self.layers, self.rt.lib
        st = rt.stream
        assert isinstance(t, list) and isinstance(coord, list) and len(t) == len(coord)
        img_xs = img_xs.to(device=rt.device, dtype=torch.float32).contiguous()
        B, _, _, Hf, Wf = img_xs.shape
        img4_full = None
        if ds_factor is not None:
            # gimmvfi_r.py:329-337
            _, img4_full = rt.prep_images(img_xs)
            img_xs = rt.resize_planes(img_xs, ds_factor)
        imgA, img4 = rt.prep_images(img_xs)
        self._cur_img_xs = img_xs      # (the float stages of GIMM-VFI-F's precision policy prepare their own copy)
        H, W = img_xs.shape[-2:]
        assert H % 8 == 0 and W % 8 == 0 and H >= 128 and W >= 128, "working resolution must be >=128 and /8"
        n = 2 * B
        HW = H * W

        # ---- cal_bidirection_flow (gimmvfi_r.py:126-156)
        # t-independent decoder front ends, hoisted out of the timestep loop
        pre = {}

        def front(feat4_, feat8_):
            pre["up8"] = self._init_upsample(feat8_)      # [2B,h4,w4,128]
            pre["up4"] = self._final_upsample(feat4_)     # [2B,H,W,64]
            pre["i0q"] = rt.resize(View(img4[:B], 0, 4), 4, 0.25)   # fi_components.py:265-267
            pre["i1q"] = rt.resize(View(img4[B:], 0, 4), 4, 0.25)

        # (GVFI_POST_LANES=0: A/B switch) everything between the recurrence and frame synthesis that does not need the flow -- the
        # projections, BidirCorrBlock's volumes, the decoders' up-sampling stacks (`front`) -- runs as a second launch sequence beside
        # the mask head, convex up-sampling, flow normalisation, motion encoder and the per-timestep INR passes, and is joined in
        # front of the first synthesis batch.  (Round 4 had tried it beside the RECURRENCE, whose workgroups hold every slot: neutral.)
        self._side_join = None
        defer = self.post_lanes and taps is None and rt.on_gpu and rt.ev_log is None

        self._side_pending = []

        def deferred(fn):
            def run(*a):
                # (launched by _start_side(), right behind the flow estimator: the fork needs the main stream's position there)
                self._side_pending.append((fn, a))
            return run if defer else fn

        # (side: what the deferred sequence produces -- pyr, pyrT, feat4, feat8 --, filled when _start_side() runs it)
        f01, f10, side_out, (h8, w8) = self._flow(imgA, B, iters, taps, seq, front=front, wrap_side=deferred)
        if "feat4" in side_out and not pre:      # (GIMM-VFI-F: the flow estimator's own features; only `front` is deferred)
            deferred(front)(side_out["feat4"], side_out["feat8"])
        self._start_side()
        h4, w4 = H // 4, W // 4
        scaler = rt.f32(B, zero=True)
        rt._chk(lib.flow_absmax(f01.data_ptr(), f10.data_ptr(), scaler.data_ptr(), B, HW, st()), "flow_absmax")
        nfA = rt.act(n, H, W, 2, zero=True, once="nfA")
        nflow = rt.f32(B, 2, 2, H, W)
        rt._chk(lib.flow_normalize(f01.data_ptr(), f10.data_ptr(), scaler.data_ptr(), nfA.data_ptr(), nfA.shape[-1],
                                   nfA.shape[-1], nflow.data_ptr(), B, H, W, rt.dtype, st()), "flow_normalize")
        raft_flow = torch.stack([rt.nhwc_to_nchw(f01, 2), rt.nhwc_to_nchw(f10, 2)], dim=2)

        # ---- predict_flow (gimmvfi_r.py:158-211): splat metric + latent encoder
        z0, z1, latcat = self._motion_encode(nfA, f01, f10, B, H, W)
        pyr, pyrT, feat4, feat8 = side_out["pyr"], side_out["pyrT"], side_out["feat4"], side_out["feat8"]
        if taps is not None:
            taps["f01"], taps["f10"] = f01, f10
            taps["w1"], taps["w2"] = z0, z1
            taps["pl0"] = latcat[..., 0:16].clone()
            taps["feat0_4"], taps["feat0_8"] = feat4[:B], feat8[:B]

        up8, up4, i0q, i1q = pre["up8"], pre["up4"], pre["i0q"], pre["i1q"]

        out = {k: [] for k in ("imgt_pred", "other_pred", "flowt0_pred", "flowt1_pred", "ninrflow", "flowt")}
        T = len(t)
        G = self._timesteps_per_batch(T, B, HW, Hf * Wf)            # timesteps per synthesis batch
        for i0 in range(0, T, G):
            g = min(G, T - i0)
            flow_all = rt.f32(g * B, H, W, 2)                       # [t][b] order
            tvs = []
            for k in range(g):
                i = i0 + k
                c, cur_t = coord[i], t[i]
                assert isinstance(c, tuple) and c[1] is None, "sub-sampled coordinates are a training feature"
                cg = c[0].to(device=rt.device, dtype=torch.float32).contiguous()
                tv = cur_t.to(device=rt.device, dtype=torch.float32).reshape(-1).contiguous()
                assert cg.shape[0] == B and cg.shape[1] == 1 and cg.shape[-1] == 3 and tv.numel() == B
                Hc, Wc = cg.shape[2], cg.shape[3]
                ninr = self._motion_inr(latcat, f01, f10, z0, z1, cg, tv, B, H, W, taps, f"t{i}_")
                assert (Hc, Wc) == (H, W), "frame synthesis needs the INR grid at the working resolution"
                flow_t = flow_all[k * B:(k + 1) * B]
                ninr_nchw = rt.f32(B, 2, 1, Hc, Wc)
                rt._chk(lib.flow_unnormalize(ninr.data_ptr(), scaler.data_ptr(), flow_t.data_ptr(), ninr_nchw.data_ptr(),
                                             B, Hc * Wc, st()), "flow_unnormalize")
                out["ninrflow"].append(ninr_nchw)
                ft_nchw = rt.nhwc_to_nchw(flow_t, 2)
                out["flowt"].append(ft_nchw.squeeze())     # B==1 squeeze quirk, gimmvfi_r.py:364-372
                tvs.append(tv)
            tv_all = tvs[0] if g == 1 else torch.cat(tvs)
            if self._side_join is not None:      # the deferred side sequence: joined where its results are first read
                cur, ss, _held = self._side_join
                cur.wait_stream(ss)
                if not torch.cuda.is_current_stream_capturing():
                    for t_ in (*pyr, *pyrT, feat4, feat8, *pre.values()):     # (allocated on the side stream, read on this one)
                        (t_.t if isinstance(t_, View) else t_).record_stream(cur)
                self._side_join = None
                up8, up4, i0q, i1q = pre["up8"], pre["up4"], pre["i0q"], pre["i1q"]
            pred, f0p, f1p, oth = self._synthesize(g * B, B, H, W, Hf, Wf, img4, img4_full, flow_all, tv_all, up8, up4, i0q, i1q,
                                                   pyr, pyrT, taps, i0, want_aux)
            for k in range(g):
                sl = slice(k * B, (k + 1) * B)
                out["imgt_pred"].append(pred[sl])
                out["flowt0_pred"].append([f[sl] for f in f0p])
                out["flowt1_pred"].append([f[sl] for f in f1p])
                out["other_pred"].append([o[sl] for o in oth])
        if self._side_join is not None:      # (no synthesis batch ran: the side sequence still has to re-join)
            self._side_join[0].wait_stream(self._side_join[1])
            self._side_join = None
        out["raft_flow"] = raft_flow
        out["nflow"] = nflow
        return out

    # activation bytes one timestep of a synthesis batch keeps alive, per pixel: ~6 KB at the working resolution (the 256-channel
    # decoder stacks), and at the frame resolution cw 32 + cb 48 + o4 16 + mean4 16 + f01 / f11 48 + pred 12 + the output clones 60
    _T_BYTES_WORK, _T_BYTES_FULL = 6144, 240

    def _timesteps_per_batch(self, T, B, HW, HfWf):
        """How many timesteps of a pair run through frame synthesis as one batch: GVFI_T_BATCH_PIX working pixels at most, and
        at most what 60 % of the device memory that is free when the signature is first seen holds (a smaller or shared GPU
        degrades to the per-timestep loop instead of running out of memory).  Cached per signature: the capture
        pass of a hipGraph must take the same decision as its warm-up pass, and may not query the device."""
        G = max(1, min(T, self.t_batch_pix // max(B * HW, 1)))
        if G > 1 and self.rt.on_gpu:
            key = (T, B, HW, HfWf)
            lim = self._tb_mem.get(key)
            if lim is None and not torch.cuda.is_current_stream_capturing():
                free, _ = torch.cuda.mem_get_info(self.rt.device)
                free += torch.cuda.memory_reserved(self.rt.device) - torch.cuda.memory_allocated(self.rt.device)
                per_t = B * (HW * self._T_BYTES_WORK + HfWf * self._T_BYTES_FULL)
                lim = self._tb_mem[key] = max(1, int(0.6 * free) // per_t)
            if lim is not None:
                G = min(G, lim)
        return G

    def _flow(self, imgA, B, iters, taps, seq=False, front=None, wrap_side=None):
        """Bidirectional flow + what frame synthesis needs from the flow estimator (gimmvfi_r.py:126-141): flows
        [B,H,W,2] f32 of both directions, the two correlation pyramids of BidirCorrBlock, context features at 1/4
        (128 ch) and 1/8 (256 ch) for both frames.  front(feat4, feat8): the caller's flow-independent work on the context
        features; with the projections and the volumes it is the `side` sequence of Engine._raft."""
        rt, Ls = self.rt, self.layers
        n = 2 * B
        H, W = imgA.shape[1:3]
        h4, w4 = H // 4, W // 4
        so = {}

        def side(fmap, cfeats):
            h8, w8 = fmap.shape[1:3]
            g = rt.act(n, h8, w8, 256)
            rt.conv(Ls["amt_fproj"], fmap, g)
            so["pyr"], so["pyrT"] = self._bidir_pyramids(g, B, h8, w8)
            so["feat4"] = rt.act(n, h4, w4, 128)
            rt.conv(Ls["amt_second_last_cproj"], cfeats[1], so["feat4"])
            so["feat8"] = rt.act(n, h8, w8, 256)
            rt.conv(Ls["amt_last_cproj"], cfeats[2], so["feat8"])
            if front is not None:
                front(so["feat4"], so["feat8"])

        flow_up, fmap, cfeats, (h8, w8) = self._raft(imgA, B, iters, taps, seq, side=side if wrap_side is None else wrap_side(side))
        return flow_up[:B], flow_up[B:], so, (h8, w8)

    def _bidir_pyramids(self, g, B, h8, w8):
        """BidirCorrBlock (raft/corr.py:23-45): volume + transposed volume, each with its pooled pyramid."""
        pyr2 = self._corr_pyramids(g, [(0, B, g[B:]), (B, B, g[:B])], 2 * B, h8, w8)
        pyr = [p[:B * h8 * w8] for p in pyr2]       # corr
        pyrT = [p[B * h8 * w8:] for p in pyr2]      # corr_T (raft/corr.py:32)
        return pyr, pyrT

    def _init_upsample(self, feat8):
        # modules/fi_components.py:234-244
        rt, Ls = self.rt, self.layers
        p = "amt_init_decoder.upsample"
        x = rt.pixel_shuffle2(feat8, 64)
        n, h, w = x.shape[:3]
        for i, co in ((1, 64), (2, 64), (3, 64), (4, 64), (5, 128)):
            y = rt.act(n, h, w, co)
            rt.conv(Ls[f"{p}.{i}.0"], x, y, act1=A.ACT_PRELU)
            x = y
        y = rt.act(n, h, w, 128)
        rt.conv(Ls[p + ".6"], x, y, act1=A.ACT_RELU)
        return y

    def _final_upsample(self, feat4):
        # modules/fi_components.py:284-295
        rt, Ls = self.rt, self.layers
        p = "amt_final_decoder.upsample"
        x = rt.pixel_shuffle2(feat4, 32)
        x = rt.pixel_shuffle2(x, 8)
        n, h, w = x.shape[:3]
        for i, co in ((2, 32), (3, 32), (4, 32), (5, 32), (6, 64)):
            y = rt.act(n, h, w, co)
            rt.conv(Ls[f"{p}.{i}.0"], View(x, 0, x.shape[-1] if i > 2 else 8), y, act1=A.ACT_PRELU)
            x = y
        y = rt.act(n, h, w, 64)
        rt.conv(Ls[p + ".7"], x, y, act1=A.ACT_RELU)
        return y

    def _synthesize(self, B, sb, H, W, Hf, Wf, img4, img4_full, flow_t, tv, up8, up4, i0q, i1q, pyr, pyrT, taps, ti0,
                    want_aux):
        """gimmvfi_r.py:222-322 for B = g * sb images: g timesteps [t][b] of the sb pairs in one batch.  flow_t [B,H,W,2],
        tv [B]; the t-independent sources (img4, up8, up4, i0q / i1q: 2 * sb images [frame 0 | frame 1]; pyr / pyrT: sb
        volumes) are read modulo sb by the warp / copy / look-up / combine kernels.  ti0: index of the first timestep (tap
        names).  Returns tensors of B images in the same [t][b] order."""
        rt, Ls, lib = self.rt, self.layers, self.rt.lib
        st = rt.stream
        HW = H * W
        h4, w4, h8, w8 = H // 4, W // 4, H // 8, W // 8
        ft0, ft1 = rt.f32(B, H, W, 2), rt.f32(B, H, W, 2)
        rt._chk(lib.flow_split_t(flow_t.data_ptr(), tv.data_ptr(), ft0.data_ptr(), ft1.data_ptr(), B, HW, st()),
                "flow_split_t")
        # quarter-resolution flows  gimmvfi_r.py:242-244 ; fl4in = [F_t0/4, F_t1/4, 0..] doubles as the head residual
        fl4in = rt.f32(B, h4, w4, 8, zero=True, once="synth.fl4in")
        rt.resize(ft0, 2, 0.25, mul=0.25, out=View(fl4in, 0, 2))
        rt.resize(ft1, 2, 0.25, mul=0.25, out=View(fl4in, 2, 2))
        # ---- NewInitDecoder  fi_components.py:255-276
        f_in = rt.act(B, h4, w4, 272, zero=True, pitch=rt.cp64(272), zero_pad_only=True, once="synth.f_in")
        k_asm = 2 if (self.misc_lanes and taps is None and rt.on_gpu) else 1      # (the seven writers of f_in's channel slices: two sequences)
        with rt.lanes(k_asm) as lanes:
            with lanes[0]:
                rt.warp(up8[:sb], 128, View(fl4in, 0, 2), View(f_in, 0, 128))
                rt.copy(View(fl4in, 0, 4), View(f_in, 256, 4), 4)
                rt.copy(View(i0q.t, 0, 3), View(f_in, 260, 3), 3)
                rt.warp(View(i0q.t, 0, 3), 3, View(fl4in, 0, 2), View(f_in, 266, 3))
            with lanes[k_asm - 1]:
                rt.warp(up8[sb:], 128, View(fl4in, 2, 2), View(f_in, 128, 128))
                rt.copy(View(i1q.t, 0, 3), View(f_in, 263, 3), 3)
                rt.warp(View(i1q.t, 0, 3), 3, View(fl4in, 2, 2), View(f_in, 269, 3))
        p = "amt_init_decoder.convblock"
        x = rt.act(B, h4, w4, 128)
        rt.conv(Ls[p + ".0.0"], f_in, x, act1=A.ACT_PRELU)
        for i in (1, 2, 3):
            x = self._resblock(f"{p}.{i}", x, 128)
        st4 = rt.f32(B, h4, w4, 8, zero=True, once="synth.st4")    # [flowt0_4(2) flowt1_4(2) mask_4(1) pad]
        rt.conv(Ls["init.head5"], x, View(st4, 0, 5), res=View(fl4in, 0, 5))
        ft_4 = rt.act(B, h4, w4, 128)
        rt.conv(Ls["init.ft"], x, ft_4)
        mask_4 = View(st4, 4, 1)
        def tap(name, ten):
            if taps is not None:
                for k in range(B // sb):
                    taps[f"t{ti0 + k}_{name}"] = ten[k * sb:(k + 1) * sb].clone()

        tap("init_flow0_4", st4[..., 0:2])
        tap("init_ft_4", ft_4)
        others = []
        if want_aux:
            # warp_w_mask at scale 4  gimmvfi_r.py:213-220, 259-261
            f0u = rt.resize(View(st4, 0, 2), 2, 4.0, mul=4.0)
            f1u = rt.resize(View(st4, 2, 2), 2, 4.0, mul=4.0)
            m4u = rt.resize(mask_4, 1, 4.0)
            iw4 = rt.f32(B, 3, H, W)
            rt._chk(lib.warp_blend(img4[:sb].data_ptr(), img4[sb:].data_ptr(), f0u.t.data_ptr(), f1u.t.data_ptr(),
                                   m4u.t.data_ptr(), iw4.data_ptr(), B, 0 if sb == B else sb, H, W, st()), "warp_blend")
            others = [iw4]
        # ---- _amt_corr_scale_lookup (downsample=2)  gimmvfi_r.py:494-507
        fl0 = rt.resize(View(st4, 0, 2), 2, 0.5, mul=0.5).t
        fl1 = rt.resize(View(st4, 2, 2), 2, 0.5, mul=0.5).t
        c0, c1 = rt.f32(B, h8, w8, 2), rt.f32(B, h8, w8, 2)
        rt._chk(lib.lookup_coords(fl0.data_ptr(), fl1.data_ptr(), tv.data_ptr(), c0.data_ptr(), c1.data_ptr(), B, h8,
                                  w8, st()), "lookup_coords")
        corr = rt.act(B, h8, w8, 648, zero=True, pitch=rt.cp64(648), zero_pad_only=True, once="synth.corr")
        rt.corr_lookup(pyr, c0, View(corr, 0, 324), B, h8, w8, h8, w8, src_n=0 if sb == B else sb)
        rt.corr_lookup(pyrT, c1, View(corr, 324, 324), B, h8, w8, h8, w8, src_n=0 if sb == B else sb)
        flow_lr = rt.f32(B, h8, w8, 4)
        rt.copy(fl0, View(flow_lr, 0, 2), 2)
        rt.copy(fl1, View(flow_lr, 2, 2), 2)
        net_lr = rt.resize(ft_4, 128, 0.5).t
        self._amt_update("amt_update4_low", net_lr, flow_lr, corr, B, h8, w8, st4, ft_4, low=True)
        corr_up = rt.act(B, h4, w4, 648, zero=True, pitch=rt.cp64(648), zero_pad_only=True, once="synth.corr_up")
        rt.resize(View(corr, 0, 648), 648, 2.0, out=View(corr_up, 0, 648))
        flow4 = rt.f32(B, h4, w4, 4)
        rt.copy(View(st4, 0, 4), flow4, 4)
        self._amt_update("amt_update4_high", ft_4, flow4, corr_up, B, h4, w4, st4, ft_4, low=False)
        tap("upd_flow0_4", st4[..., 0:2])
        tap("upd_ft_4", ft_4)
        # ---- NewMultiFlowDecoder  fi_components.py:307-340
        fl0u = rt.resize(View(st4, 0, 2), 2, 4.0, mul=4.0).t
        fl1u = rt.resize(View(st4, 2, 2), 2, 4.0, mul=4.0).t
        mku = rt.resize(mask_4, 1, 4.0).t
        fin = rt.act(B, H, W, 273, zero=True, pitch=rt.cp64(273), zero_pad_only=True, once="synth.fin")
        rt.resize(ft_4, 128, 4.0, out=View(fin, 0, 128))
        rt.warp(up4[:sb], 64, fl0u, View(fin, 128, 64))
        rt.warp(up4[sb:], 64, fl1u, View(fin, 192, 64))
        rt.copy(fl0u, View(fin, 256, 2), 2)
        rt.copy(fl1u, View(fin, 258, 2), 2)
        rt.copy(mku, View(fin, 260, 1), 1)
        rt.copy(View(img4[:sb], 0, 3), View(fin, 261, 3), 3)
        rt.copy(View(img4[sb:], 0, 3), View(fin, 264, 3), 3)
        rt.warp(View(img4[:sb], 0, 3), 3, fl0u, View(fin, 267, 3))
        rt.warp(View(img4[sb:], 0, 3), 3, fl1u, View(fin, 270, 3))
        p = "amt_final_decoder.convblock"
        x = rt.act(B, H, W, 256)
        rt.conv(Ls[p + ".0.0"], fin, x, act1=A.ACT_PRELU)
        for i in (1, 2, 3):
            x = self._resblock(f"{p}.{i}", x, 256)
        dec = rt.f32(B, H, W, 24)
        rt.conv(Ls[p + ".4"], x, dec)
        rt._chk(lib.decoder_head(dec.data_ptr(), 24, fl0u.data_ptr(), fl1u.data_ptr(), mku.data_ptr(), B * HW, st()),
                "decoder_head")
        tap("final_flow0_1", dec[..., 0:6])
        tap("final_mask", dec[..., 12:15])
        tap("final_res", dec[..., 15:24])
        # ---- multi_flow_combine + comb_block  fi_components.py:57-94, gimmvfi_r.py:294-308.  With DS_SCALE < 1 the decoder
        # output lives at the working resolution: its bilinear up-sampling (flows x Hf/H), the six warps + blends and the
        # planar copies of the up-sampled flows for the return dict are one pass over the full-resolution pixels
        i0f, i1f = (img4[:sb], img4[sb:]) if img4_full is None else (img4_full[:sb], img4_full[sb:])
        cw = rt.act(B, Hf, Wf, 9, zero=False)
        mean4 = rt.f32(B, Hf, Wf, 4)
        f01 = rt.f32(B, 3, 2, Hf, Wf)
        f11 = rt.f32(B, 3, 2, Hf, Wf)
        rt._chk(lib.combine_warps_up(i0f.data_ptr(), i1f.data_ptr(), dec.data_ptr(), 24, H, W, cw.data_ptr(), cw.shape[-1],
                                     cw.shape[-1], mean4.data_ptr(), f01.data_ptr(), f11.data_ptr(), B, 0 if sb == B else sb, Hf, Wf,
                                     rt.dtype, st()), "combine_warps_up")
        cb = rt.act(B, Hf, Wf, 18, zero=True, once="synth.cb")
        # (pad16: cb's channels 18..23 and o4's / mean4's channel 3 are padding owned here -> whole 16-byte stores)
        rt.conv(Ls["amt_comb_block.0"], View(cw, 0, 9), View(cb, 0, 18), act1=A.ACT_PRELU, pad16=True, algo=rt.comb_algo)
        o4 = rt.f32(B, Hf, Wf, 4)
        pred = rt.f32(B, 3, Hf, Wf)
        # (the column kernel stores clamp((y + 1) / 2, 0, 1) straight into the planar frame; other kernels leave o4 to finalize_image)
        rt.conv(Ls["amt_comb_block.2"], View(cb, 0, 18), View(o4, 0, 3), res=View(mean4, 0, 3), pad16=True, algo=rt.comb_algo,
                planar3=pred if rt.fold_finalize else None)
        if not rt.last_planar:
            rt._chk(lib.finalize_image(o4.data_ptr(), 4, pred.data_ptr(), B, Hf, Wf, st()), "finalize_image")
        f04 = rt.nhwc_to_nchw(View(st4, 0, 2), 2)
        f14 = rt.nhwc_to_nchw(View(st4, 2, 2), 2)
        return pred, [f01, f04], [f11, f14], others
