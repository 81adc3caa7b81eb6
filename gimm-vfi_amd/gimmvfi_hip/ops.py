"""Host-side launch wrappers over the C ABI (include/gimmvfi_hip.h).

torch is used here only for device memory (allocation, views) and the current
HIP stream; every arithmetic operation of the path is a HIP kernel behind the
C ABI.  ``Runtime`` carries the library handle, the element type of the
activation tensors (bf16 fast path / fp32 validation path) and the device.
"""
import ctypes as C
import os
import math

import torch

from . import lib as L


def roundup(x, m):
    return (x + m - 1) // m * m


class View:
    """A channel slice [coff, coff+c) of an NHWC tensor t[N,H,W,ld]."""

    __slots__ = ("t", "coff", "c")

    def __init__(self, t, coff=0, c=None):
        self.t = t
        self.coff = coff
        self.c = (t.shape[-1] - coff) if c is None else c

    @property
    def ld(self):
        return self.t.shape[-1]

    @property
    def ptr(self):
        return self.t.data_ptr() + self.coff * self.t.element_size()

    @property
    def is_f32(self):
        return 1 if self.t.dtype == torch.float32 else 0

    @property
    def npix(self):
        return self.t.numel() // self.t.shape[-1]

    def imgs(self, a, b):
        """Sub-batch [a, b) of images (leading dimension)."""
        return View(self.t[a:b], self.coff, self.c)


def V(t, coff=0, c=None):
    return t if isinstance(t, View) else View(t, coff, c)


class ConvLayer:
    """Packed weights [Cout][KH][KW][cin_pad] (+ float bias / PReLU slope) of one convolution."""

    def __init__(self, rt, w, b, stride=1, pad=None, pad_mode=L.PAD_ZEROS, slope=None, cin_pad=None, wdir=False):
        """wdir: also pack the MFMA-fragment-ordered image (w_layout = 2) of the weights-direct variant of the LDS-DMA
        kernel -- for the layers of the flow estimators' recurrences (small M, launch time = one workgroup's K chain)."""
        cout, cin, kh, kw = w.shape
        cp = roundup(cin, rt.VE) if cin_pad is None else cin_pad
        pk = torch.zeros(cout, kh, kw, cp, dtype=torch.float32, device=w.device)
        pk[..., :cin] = w.detach().float().permute(0, 2, 3, 1)
        self.w = pk.to(rt.tdtype).contiguous().to(rt.device)
        # second image for the LDS-DMA kernel: K-chunk major + pre-swizzled == the LDS tile, so the weight
        # tile of a K chunk is one contiguous block (conv_igemm_glds.hip, w_layout = 1)
        bke = 8 * rt.VE
        self.w_glds = None
        # (the LDS-DMA kernel keeps tap validity in a 32-bit mask: filters with more than 32 taps take the generic kernel)
        if cp % bke == 0 and pad_mode == L.PAD_ZEROS and kh * kw <= 32 and not os.environ.get("GVFI_NO_WGLDS"):
            k = kh * kw * cp
            # K chunks in the kernel's walk order: channel chunk outer, filter tap inner (L2 locality of the taps)
            wk = pk.reshape(cout, kh * kw, cp // bke, 8, rt.VE).permute(0, 2, 1, 3, 4).reshape(cout, k // bke, 8, rt.VE)
            sw = (torch.arange(cout, device=pk.device) >> 1) & 7                # (row>>1)&7 with row == n (tile bases are /16)
            src_slot = torch.arange(8, device=pk.device)[None, :] ^ sw[:, None]    # dest slot s holds source slot s^sw
            wk = torch.gather(wk, 2, src_slot[:, None, :, None].expand(cout, k // bke, 8, rt.VE))
            self.w_glds = wk.permute(1, 0, 2, 3).contiguous().to(rt.tdtype).to(rt.device)   # [chunk][n][slot][ve]
        self.w_frag = None
        self.use_wdir = bool(wdir) and os.environ.get("GVFI_WDIR", "1") != "0"
        if (self.use_wdir and rt.precision in ("bf16", "fp16") and cp % 64 == 0 and pad_mode == L.PAD_ZEROS
                and kh * kw <= 32):
            k = kh * kw * cp
            nb = (cout + 31) // 32
            wp = torch.zeros(nb * 32, kh * kw, cp, dtype=torch.float32, device=pk.device)
            wp[:cout] = pk.reshape(cout, kh * kw, cp)
            # K chunks in the kernel's walk order (channel chunk outer, tap inner), then [chunk][col block][k-step][half][col][8]
            wk = wp.reshape(nb, 32, kh * kw, cp // 64, 4, 2, 8).permute(3, 2, 0, 4, 5, 1, 6)
            self.w_frag = wk.contiguous().reshape(k // 64, nb, 4, 64, 8).to(rt.tdtype).to(rt.device)
        self.b = None if b is None else b.detach().float().contiguous().to(rt.device)
        self.slope = None if slope is None else slope.detach().float().contiguous().to(rt.device)
        self.cout, self.cin, self.cin_pad, self.kh, self.kw = cout, cin, cp, kh, kw
        self.stride = stride
        self.pad = (kh // 2, kw // 2) if pad is None else pad
        self.pad_mode = pad_mode


class PatchConvLayer:
    """A KHxKW stride-1 zero-padded convolution over a few channels, run as im2col (gvfi_im2col) + 1x1 convolution:
    K = KH*KW*cin real products, padded once to a whole K chunk instead of padding every tap's channels."""

    def __init__(self, rt, w, b, slope=None, wdir=False):
        cout, cin, kh, kw = w.shape
        self.kh, self.kw, self.cin, self.cout = kh, kw, cin, cout
        self.kpad = roundup(kh * kw * cin, 8 * rt.VE)
        w2 = torch.zeros(cout, self.kpad, 1, 1, dtype=torch.float32, device=w.device)
        w2[:, :kh * kw * cin, 0, 0] = w.detach().float().permute(0, 2, 3, 1).reshape(cout, -1)   # K order (kh, kw, c)
        self.inner = ConvLayer(rt, w2, b, slope=slope, wdir=wdir)
        self.inner.cin = kh * kw * cin      # real products per output (FLOP accounting)


class S2DConvLayer:
    """A convolution whose filter size equals its stride, without padding (patch embeddings, sub-sampling convolutions of
    the Twins encoder: twins.py:720-745, 870-925): non-overlapping k x k patches.  In NHWC the k pixels of one patch row
    are ONE contiguous run of k * ld values, so the input [N, H, W, ld] IS the tensor [N * H/k, k, W/k, k * ld] and the
    layer is a k x 1 stride-1 convolution over k * ld "channels" with ONE output row per image -- no data moves, and a
    filter with more than 32 taps (8 x 8: the generic kernel at 12 TFLOP/s) becomes 8 taps of the LDS-DMA kernel.
    ld = channel pitch of the input tensor (whole pitch: the view must start at channel 0)."""

    def __init__(self, rt, w, b, ld):
        cout, cin, kh, kw = w.shape
        assert kh == kw and cin <= ld
        self.k, self.ld, self.cin, self.cout = kh, ld, cin, cout
        w2 = torch.zeros(cout, kw * ld, kh, 1, dtype=torch.float32, device=w.device)
        # w2[o, kx * ld + c, ky, 0] = w[o, c, ky, kx]
        w2.view(cout, kw, ld, kh)[:, :, :cin, :] = w.detach().float().permute(0, 3, 1, 2)
        self.inner = ConvLayer(rt, w2, b, stride=1, pad=(0, 0))
        self.inner.cin = kh * kw * cin      # real products per output (FLOP accounting)


class TokenChain:
    """Three linears 128 -> 64 -> 64 -> 64 (+ LayerNorm parameters) packed for gvfi_token_chain (csrc/token_chain.hip):
    weights in MFMA-fragment order, fragment (layer, 32-row block mb, k-step kk) = 64 lanes x 8 values with lane l holding
    W[32 mb + (l & 31)][16 kk + 8 (l >> 5) .. +8]; biases [3][64]; gamma / beta [64].  ws[0]: [64, K0 <= 128] (zero-padded
    to 128 input features), ws[1], ws[2]: [64, 64]."""

    def __init__(self, rt, ws, bs, ln, eps, ln_after, act0=L.ACT_NONE, act1=L.ACT_NONE, res2_from0=False):
        assert rt.precision in ("bf16", "fp16") and len(ws) == 3
        frs = []
        for li, w in enumerate(ws):
            k = 128 if li == 0 else 64
            wp = torch.zeros(64, k, dtype=torch.float32)
            assert w.shape[0] == 64 and w.shape[1] <= k, w.shape
            wp[:, :w.shape[1]] = w.detach().float().cpu()
            # [mb][row 32][kk][half][8] -> [mb][kk][half][row][8]
            frs.append(wp.view(2, 32, k // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous().reshape(-1))
        self.wfrag = torch.cat(frs).to(rt.tdtype).contiguous().to(rt.device)
        self.bias = torch.stack([(torch.zeros(64) if b is None else b.detach().float().cpu()) for b in bs]).contiguous().to(rt.device)
        self.ln_g = ln[0].detach().float().contiguous().to(rt.device)
        self.ln_b = ln[1].detach().float().contiguous().to(rt.device)
        self.eps, self.ln_after, self.act0, self.act1, self.res2_from0 = eps, ln_after, act0, act1, res2_from0


class TapSplitConvLayer:
    """A KHxKW zero-padded stride-1 convolution with very few output channels, run as a 1x1 convolution to the
    KH*KW*Cout per-tap partial sums + gvfi_tap_sum (Runtime.tap_split_conv)."""

    def __init__(self, rt, w, b):
        cout, cin, kh, kw = w.shape
        self.cout, self.cin, self.kh, self.kw = cout, cin, kh, kw
        w2 = w.detach().float().permute(2, 3, 0, 1).reshape(kh * kw * cout, cin, 1, 1)   # row = tap * cout + c
        self.inner = ConvLayer(rt, w2, None)
        self.b = None if b is None else b.detach().float().contiguous().to(rt.device)


class InrMlp:
    """Packed weights of the fused hypo-network kernel (gvfi_inr_mlp): layers = [(w [out,in], b [out])] x 5 with
    dims 35 -> 128 -> 128 -> 128 -> 128 -> 2.  The packing is the library's own host routine."""

    DIMS = [(128, 35), (128, 128), (128, 128), (128, 128), (2, 128)]

    @staticmethod
    def supported(rt, layers):
        return rt.precision == "bf16" and [tuple(w.shape) for w, _ in layers] == InrMlp.DIMS

    def __init__(self, rt, layers):
        assert InrMlp.supported(rt, layers)
        ws = [w.detach().float().cpu().contiguous() for w, _ in layers]
        bs = [b.detach().float().cpu().contiguous() for _, b in layers]
        nb, nf = C.c_int(), C.c_int()
        rt.lib.inr_mlp_pack_sizes(C.byref(nb), C.byref(nf))
        wfrag = torch.zeros(nb.value // 2, dtype=torch.bfloat16)
        bias = torch.zeros(nf.value, dtype=torch.float32)
        wp = (C.c_void_p * 5)(*[t.data_ptr() for t in ws])
        bp = (C.c_void_p * 5)(*[t.data_ptr() for t in bs])
        rc = rt.lib.inr_mlp_pack(wp, bp, wfrag.data_ptr(), bias.data_ptr())
        if rc != 0:
            raise RuntimeError(f"gvfi_inr_mlp_pack failed with code {rc}")
        self.wfrag = wfrag.to(rt.device)
        self.bias = bias.to(rt.device)
        self.layers = list(zip(ws, bs))   # un-packed copies (host): kept for inspection / test restatements


class _Lanes:
    """`with rt.lanes(k) as lanes:` then `with lanes[i]: ...` for i < k: k independent launch sequences (e.g. the
    recurrences of k sub-batches), each on its own HIP stream, all started after the work enqueued so far and joined
    when the outer block exits.  Inside a hipGraph capture they become k parallel branches, so the under-filled
    launches of one sequence (224-448 workgroups, each with a serial prologue / K loop / epilogue) overlap those of
    the others.  Lane 0 is the current stream.  Without a GPU, or while launches are being timed one by one
    (``rt.ev_log``), the lanes simply run one after another on the current stream.  Do not nest: a stream forked from
    a forked stream crashes hipStreamEndCapture (ROCm 7.0, tools/lanes_probe.py)."""

    def __init__(self, rt, k):
        self.rt, self.k = rt, k
        self.cur = None
        self.streams = []

    def __enter__(self):
        rt = self.rt
        if rt.on_gpu and rt.ev_log is None and self.k > 1:
            self.cur = torch.cuda.current_stream(rt.device)
            self.streams = rt._lane_streams(self.cur, self.k - 1)
            for st in self.streams:
                st.wait_stream(self.cur)
        return self

    def __getitem__(self, i):
        assert 0 <= i < self.k
        if not self.streams or i == 0:
            return _NullCtx()
        return torch.cuda.stream(self.streams[i - 1])

    def __exit__(self, *exc):
        for st in self.streams:
            self.cur.wait_stream(st)
        return False


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class Runtime:
    def __init__(self, lib, precision, device):
        self.lib = lib
        self.precision = precision
        if precision == "fp32":
            self.dtype, self.tdtype, self.VE = L.F32, torch.float32, 4
        elif precision == "bf16":
            self.dtype, self.tdtype, self.VE = L.BF16, torch.bfloat16, 8
        elif precision == "fp16":
            # IEEE half activations / weights (GVFI_F16): the decoder of GIMM-VFI-F's flow estimator -- not a mode of the
            # whole model (the halo-staged 3x3 / patch / fused-INR kernels of the synthesis path are bf16 / float only)
            self.dtype, self.tdtype, self.VE = L.F16, torch.float16, 8
        else:
            raise ValueError(f"precision must be 'bf16', 'fp16' or 'fp32', got {precision}")
        self.device = torch.device(device)
        self.on_gpu = self.device.type == "cuda"
        self.n_launch = 0
        self.ev_log = None   # list => conv launches are bracketed by HIP events (bench.py)
        self.lookup_lds = os.environ.get("GVFI_LOOKUP_LDS", "0") == "1"   # A/B switch: LDS-staged correlation look-up
        self.fuse_seam = os.environ.get("GVFI_FUSE_SEAM", "1") != "0"     # A/B switch: gvfi_flow_step between iterations
        # A/B switch: 0 keeps the combination block (7x7, 9 -> 18 -> 3) on the patch kernel instead of conv_col7.hip
        self.comb_algo = 0 if os.environ.get("GVFI_COL7", "1") != "0" else 3
        self.use_p3x3 = os.environ.get("GVFI_P3X3", "1") != "0"   # A/B switch: 0 keeps the LDS-DMA kernel on the hot 3x3 layers
        # A/B switch of the halo-staged 3x3 kernel's launch form (algo bits 13, 14 of gvfi_conv2d_p3x3): 0 = auto (persistent stream
        # kernel where it applies), 1 = the round-2 tile-per-workgroup kernel, 2 = tile per workgroup + wave-private epilogue
        self.p3x3_form = int(os.environ.get("GVFI_P3X3_FORM", "0")) & 3
        self.ev_shapes = False   # tags carry the problem shape (bench.py --shapes: per-shape table)
        self._lanes = {}         # stream id -> extra streams for lanes()
        self.last_stats_fused = False
        self.last_planar = False
        self.fold_finalize = os.environ.get("GVFI_FOLD_FINALIZE", "1") != "0"   # A/B switch: finalize_image inside the last 7x7 layer
        # zero-once buffers (act(once=...)): name/shape -> tensor, and which input signature (`once_scope`, set by the models around
        # a forward) uses which -- release_once(scope) when that signature's graph is evicted.  ONE forward at a time per runtime:
        # these buffers (and the engines' caches) are shared by every forward of the runtime, two forwards on different streams
        # would race on them.
        self._once_scopes = {}
        self.once_scope = None
        self.zero_once = os.environ.get("GVFI_ZERO_ONCE", "1") != "0"           # A/B switch: persistent zero-once buffers (act(once=...))
        self.pair_launch = os.environ.get("GVFI_CONV_PAIR", "1") != "0"         # A/B switch: two independent convolutions per launch
        self._once = {}
        # PROFILING ONLY (results are garbage): the kernel's phase-skip switches on every weights-direct launch of the recurrences --
        # 8 = no epilogue, 16 = no K loop, 24 = neither: how much of the recurrence's wall time is the fixed per-launch cost
        # (launch + prologue [+ epilogue]) and how much the contraction (tools/evidence.sh wdir-floor, profiles/r5_wdir_floor.txt)
        self.wdir_dbg = int(os.environ.get("GVFI_WDIR_DBG", "0")) & 24
        if self.wdir_dbg:
            # a profiling switch of tools/ (phases of every weights-direct launch are SKIPPED: times are real, results are not)
            import warnings

            warnings.warn(f"gimmvfi_hip: GVFI_WDIR_DBG={self.wdir_dbg} is set -- the weights-direct convolutions skip phases of their "
                          "work (profiling only): EVERY FRAME OF THIS RUNTIME IS GARBAGE", RuntimeWarning, stacklevel=2)
        self.wdir_bm128 = os.environ.get("GVFI_WDIR_BM128", "0") == "1"       # A/B switch: 128-row weights-direct tiles (see conv())
        self.wdir_bm128_max = int(os.environ.get("GVFI_WDIR_BM128_MAX", "16384"))     # (largest pixel count that takes them)
        self.wdir_bm128_cout = int(os.environ.get("GVFI_WDIR_BM128_COUT", "129"))     # (smallest Cout that takes them)

    def sibling(self, precision):
        """A runtime of another precision over the same library and device (GIMM-VFI-F's float flow-estimator stages)."""
        return Runtime(self.lib, precision, self.device)

    # ------------------------------------------------------------------ memory
    def stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream if self.on_gpu else 0

    def lanes(self, k):
        return _Lanes(self, k)

    def _lane_streams(self, cur, k):
        lst = self._lanes.setdefault(cur.cuda_stream, [])
        while len(lst) < k:
            lst.append(torch.cuda.Stream(device=self.device))
        return lst[:k]

    def cp(self, c):
        return roundup(c, self.VE)

    def cp64(self, c):
        """Channel count padded to one K chunk of the LDS-DMA convolution (128 bytes: 64 bf16 / 32 f32)."""
        return roundup(c, 8 * self.VE)

    def act(self, n, h, w, c, zero=None, pitch=None, zero_pad_only=False, once=None):
        """Activation tensor in the runtime element type with padded channel pitch.  zero_pad_only: the caller
        writes every real channel [0, c) before the tensor is read, so only the pad channels are cleared (the 320-pitch
        decoder input at full resolution is 587 MB per timestep -- clearing all of it was a 76 us memset).
        once: a call-site name -- the zero state is only ever needed ONCE (pad channels nobody writes, channels every forward
        overwrites before reading): the tensor is then a persistent buffer of this runtime, zero-filled when it is first
        created and handed out again for the same (name, shape) -- no fill kernel per forward (round 5: the 33 ATen fills /
        copies of a captured forward were 0.2 ms of its 23 ms at 448x256, 0.9 ms at 4K).  Intermediates only: the next
        forward overwrites them."""
        cpad = self.cp(c) if pitch is None else pitch
        z = (cpad != c) if zero is None else zero
        if z and once is not None:
            t = self._zero_once(once, (n, h, w, cpad), self.tdtype)
            if t is not None:
                return t
        if z and zero_pad_only and cpad > c:
            t = torch.empty((n, h, w, cpad), dtype=self.tdtype, device=self.device)
            t[..., c:].zero_()
            return t
        f = torch.zeros if z else torch.empty
        return f((n, h, w, cpad), dtype=self.tdtype, device=self.device)

    def f32(self, *shape, zero=False, once=None):
        if zero and once is not None:
            t = self._zero_once(once, tuple(shape), torch.float32)
            if t is not None:
                return t
        return (torch.zeros if zero else torch.empty)(shape, dtype=torch.float32, device=self.device)

    def _zero_once(self, name, shape, dtype):
        """The persistent zero-initialised buffer (name, shape, dtype) of this runtime, or None when it cannot be created
        right now (switch off; first seen inside a hipGraph capture -- the warm-up pass of a capture creates it before)."""
        if not self.zero_once:
            return None
        key = (name, shape, dtype)
        t = self._once.get(key)
        if t is None:
            if self.on_gpu and torch.cuda.is_current_stream_capturing():
                return None
            t = self._once[key] = torch.zeros(shape, dtype=dtype, device=self.device)
        if self.once_scope is not None:
            self._once_scopes.setdefault(self.once_scope, set()).add(key)
        return t

    def release_once(self, scope):
        """Drops the zero-once buffers that only the input signature `scope` used (the models call it when they evict that
        signature's captured graph: the buffers live outside the graph's private pool -- the largest activations of a forward,
        hundreds of MB per timestep at 4K -- and would otherwise pile up under shape churn).  Buffers another live
        signature also uses stay."""
        keys = self._once_scopes.pop(scope, set())
        live = set().union(*self._once_scopes.values()) if self._once_scopes else set()
        for k in keys - live:
            self._once.pop(k, None)

    def once_bytes(self):
        return sum(t.numel() * t.element_size() for t in self._once.values())

    def _chk(self, rc, name):
        self.n_launch += 1
        if rc != 0:
            raise RuntimeError(f"gvfi_{name} failed with code {rc}")

    # ------------------------------------------------------------------ convolution
    def conv(self, layer, x0, out, x1=None, act1=L.ACT_NONE, res=None, act2=L.ACT_NONE, out_scale=1.0,
             slope1=None, slope2=None, epi=L.EPI_STD, y2=None, aux0=None, aux1=None, groups=1,
             w_group_stride=0, w_raw=None, cout=None, tile=0, algo=0, stats=None, pad16=False, state_f32=False, planar3=None,
             _defer=False):
        """_defer: build and route the problem but do not launch it; returns (params, flops, layer meta) for conv_pair.
        planar3: float (N, 3, Ho, Wo) tensor -- when the column kernel takes this launch (3 float output channels) the result
        leaves finalised, clamp((y + 1) / 2, 0, 1), in planar form there and `out` is NOT written (gvfi_finalize_image folded
        in); ``self.last_planar`` says whether that happened (else the caller finalises `out` itself).
        state_f32 (GRU epilogues, bf16 mode): the recurrent state tensors are float (gvfi_conv_params.state_f32).
        stats: optional zero-initialised f32 [N, cout, 4] (= two 64-bit fixed-point sums per image and channel, stats_tensor());
        when the library can fuse the InstanceNorm statistics
        into this convolution (gvfi_conv2d_stats_ok) they are accumulated there and True is returned in
        ``self.last_stats_fused``, else the caller computes them with instnorm_stats."""
        x0 = V(x0)
        out = V(out)
        p = L.ConvParams()
        p.dtype = self.dtype
        n, h, w_ = x0.t.shape[0], x0.t.shape[1], x0.t.shape[2]
        p.x0, p.ld0, p.c0 = x0.ptr, x0.ld, self.cp(x0.c)
        if x1 is not None:
            x1 = V(x1)
            assert x1.t.shape[:3] == x0.t.shape[:3]
            assert x0.c % self.VE == 0, "first source of a two-source conv must be vector aligned"
            p.x1, p.ld1, p.c1 = x1.ptr, x1.ld, self.cp(x1.c)
        else:
            p.x1, p.ld1, p.c1 = None, 0, 0
        assert x0.coff % self.VE == 0 and (x1 is None or x1.coff % self.VE == 0)
        p.N, p.H, p.W = n, h, w_
        if layer is not None:
            assert p.c0 + p.c1 == layer.cin_pad, (p.c0, p.c1, layer.cin_pad)
            bke_ = 8 * self.VE
            want = algo & 15
            aligned = layer.w_glds is not None and p.c0 % bke_ == 0 and p.c1 % bke_ == 0
            # the mid-channel 3x3 kernel (conv_p3x3s.hip) reads the plain weight image; decided below once p is complete
            small3 = (want in (0, 5) and self.use_p3x3 and x1 is None and layer.kh == 3 and layer.kw == 3 and layer.stride == 1
                      and layer.cout <= 64 and p.c0 in (32, 64) and p.c0 == layer.cin_pad and self.dtype == L.BF16
                      and (want == 5 or n * h * w_ >= 65536))
            if want == 5:
                p.w, p.w_layout = layer.w.data_ptr(), 0
            elif want in (0, 6) and layer.w_frag is not None and layer.use_wdir and p.c0 % 64 == 0 and p.c1 % 64 == 0 and groups == 1:
                p.w, p.w_layout = layer.w_frag.data_ptr(), 2      # weights-direct variant of the LDS-DMA kernel
                algo = 2 | (algo & ~15) | (self.wdir_dbg << 8)
                # 128-row tiles for the layers with two 128-column tiles (Cout > 128) while the 64-row grid would exceed half of the
                # chip's 512 workgroup slots: half the weight stream per pixel, and both lanes' launches co-reside (GVFI_WDIR_BM128)
                if self.wdir_bm128 and tile == 0 and layer.cout >= self.wdir_bm128_cout and 12288 <= n * h * w_ <= self.wdir_bm128_max:
                    tile = 128 | (128 << 10)
                want = 2
            elif want in (0, 2, 4) and aligned and not (algo & 128):
                p.w, p.w_layout = layer.w_glds.data_ptr(), 1
                algo = (4 if want == 4 else 2) | (algo & ~15)
            else:
                p.w, p.w_layout = layer.w.data_ptr(), 0
            p.bias = None if layer.b is None else layer.b.data_ptr()
            kh, kw, st, (ph, pw), pm = layer.kh, layer.kw, layer.stride, layer.pad, layer.pad_mode
            p.Cout = layer.cout
        else:  # weights are another activation tensor (correlation volume): [groups][cout][c0]
            p.w = w_raw.data_ptr()
            p.bias = None
            kh = kw = st = 1
            ph = pw = 0
            pm = L.PAD_ZEROS
            p.Cout = cout
        p.w_group_stride, p.groups = w_group_stride, groups
        p.KH, p.KW, p.stride, p.pad_h, p.pad_w, p.pad_mode = kh, kw, st, ph, pw, pm
        p.Ho = (h + 2 * ph - kh) // st + 1
        p.Wo = (w_ + 2 * pw - kw) // st + 1
        assert out.t.shape[0] == n and out.t.shape[1] == p.Ho and out.t.shape[2] == p.Wo, (out.t.shape, n, p.Ho, p.Wo)
        p.epi_mode = epi
        p.act1, p.act2 = act1, act2
        s1 = slope1 if slope1 is not None else (layer.slope if (layer is not None and act1 == L.ACT_PRELU) else None)
        p.slope1 = None if s1 is None else s1.data_ptr()
        p.slope2 = None if slope2 is None else slope2.data_ptr()
        if res is not None:
            res = V(res)
            p.res, p.ldr, p.res_f32 = res.ptr, res.ld, res.is_f32
        else:
            p.res, p.ldr, p.res_f32 = None, 0, 0
        p.out_scale = out_scale
        p.y, p.ldy, p.y_f32 = out.ptr, out.ld, out.is_f32
        if y2 is not None:
            y2 = V(y2)
            p.y2, p.ldy2 = y2.ptr, y2.ld
        if aux0 is not None:
            aux0 = V(aux0)
            p.aux0, p.lda0 = aux0.ptr, aux0.ld
        if aux1 is not None:
            aux1 = V(aux1)
            p.aux1, p.lda1 = aux1.ptr, aux1.ld
        if pad16:
            # the caller owns the pad channels of `out` (and `res`) up to the next 16-byte boundary: check that they exist
            unit = 4 if out.is_f32 else 8
            assert out.coff % unit == 0 and out.coff + roundup(p.Cout, unit) <= out.ld
            if res is not None:
                ru = 4 if res.is_f32 else 8
                assert res.coff % ru == 0 and res.coff + roundup(p.Cout, ru) <= res.ld
            algo |= 16
        p.tile_hint = tile
        p.algo = algo
        p.state_f32 = 1 if (state_f32 and self.dtype != L.F32) else 0
        p.stats = None
        if layer is not None and want == 0 and p.w_layout == 1 and stats is None and self.use_p3x3 \
                and self.lib.conv2d_p3x3_eligible(C.byref(p)) == 1:
            p.algo = 4 | (algo & ~15)       # halo-staged 3x3 kernel (conv_p3x3.hip) ahead of the LDS-DMA kernel
            if not (p.algo >> 13) & 3:
                p.algo |= self.p3x3_form << 13
        if layer is not None and want == 0 and small3:
            keep = (p.w, p.w_layout)
            p.w, p.w_layout = layer.w.data_ptr(), 0
            if self.lib.conv2d_p3x3s_eligible(C.byref(p)) == 1:
                p.algo = 5 | (algo & ~15)   # mid-channel sibling (conv_p3x3s.hip)
            else:
                p.w, p.w_layout = keep
        if not self.use_p3x3 and layer is not None and (p.algo & 15) == 0:
            # A/B switch GVFI_P3X3=0: with algo 0 the LIBRARY would still route to the halo-staged / patch kernels by itself;
            # an explicit algo (LDS-DMA where it is eligible, else generic) makes the baseline really theirs
            p.algo = (2 if self.lib.conv2d_glds_eligible(C.byref(p)) else 1) | (p.algo & ~15)
        self.last_planar = False
        if planar3 is not None and layer is not None:
            assert planar3.dtype == torch.float32 and planar3.is_contiguous() and tuple(planar3.shape) == (n, 3, p.Ho, p.Wo)
            keep_algo, keep_y2 = p.algo, p.y2
            p.algo, p.y2 = p.algo | 64, planar3.data_ptr()
            plan = (C.c_int * 5)()
            if self.lib.conv2d_col7_eligible(C.byref(p)) >= 1 and self.lib.conv2d_plan(C.byref(p), plan) == 0 and plan[0] == 7:
                self.last_planar = True
            else:
                p.algo, p.y2 = keep_algo, keep_y2
        self.last_stats_fused = False
        if stats is not None:
            p.stats = stats.data_ptr()
            if self.lib.conv2d_stats_ok(C.byref(p)) == 1:
                self.last_stats_fused = True
            else:
                p.stats = None
        self.last_algo = p.algo & 15         # (tests: which kernel family an explicit request really got)
        if _defer:
            cin_real = (layer.cin if layer is not None else x0.c)
            return p, 2.0 * n * p.Ho * p.Wo * p.Cout * kh * kw * cin_real
        if self.ev_log is None:
            self._chk(self.lib.conv2d(C.byref(p), self.stream()), "conv2d")
        else:
            # measurement mode (bench.py): HIP events on the launch stream around this kernel
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            self._chk(self.lib.conv2d(C.byref(p), self.stream()), "conv2d")
            e1.record()
            plan = (C.c_int * 5)()
            self._chk(self.lib.conv2d_plan(C.byref(p), plan), "conv2d_plan")   # the library says which kernel it ran
            cin_real = (layer.cin if layer is not None else x0.c)
            flops = 2.0 * n * p.Ho * p.Wo * p.Cout * kh * kw * cin_real
            kname = {1: "conv_igemm_kernel", 2: "conv_igemm_glds_kernel", 3: "conv_patch_kernel", 4: "conv_p3x3_kernel", 5: "conv_p3x3s_kernel",
                     6: "conv_igemm_glds_kernel[wdir]", 7: "conv_col7_kernel"}[plan[0]]
            if plan[0] == 4 and self.lib.conv2d_p3x3_form(C.byref(p)) == 3:
                kname = "conv_p3x3_stream_kernel"      # (the persistent form: its own kernel in a rocprofv3 trace)
            tag = f"{kname}<{ {L.F32: 'float', L.BF16: 'bf16', L.F16: 'f16'}[self.dtype] },{plan[1]},{plan[2]},kb{plan[3]},s{plan[4]}>"
            if self.ev_shapes:
                tag += f" {n}x{h}x{w_} {cin_real}->{p.Cout} {kh}x{kw}s{st}"
            self.ev_log.append((tag, flops, e0, e1))
        return out

    def conv_pair(self, a, b):
        """Two independent convolutions (keyword dicts of Runtime.conv) as ONE launch where the library takes the pair
        (gvfi_conv2d_pair: both on the weights-direct 64 x 128 variant), else one after the other -- the same kernel body
        either way, bit-identical results.  GVFI_CONV_PAIR=0: always one by one (A/B switch)."""
        if not self.pair_launch:
            self.conv(**a)
            self.conv(**b)
            return
        pa, fa = Runtime.conv(self, **a, _defer=True)      # (Runtime.conv: a test runtime may override conv())
        pb, fb = Runtime.conv(self, **b, _defer=True)
        if self.ev_log is not None:
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
        rc = self.lib.conv2d_pair(C.byref(pa), C.byref(pb), self.stream())
        if rc in (-2, -3):
            # not a pair the weights-direct variant takes (nothing was launched): two ordinary launches through conv() -- a
            # subclass's override included --, each logged as what it is (the fallback used to be logged as one
            # '[wdir-pair]' entry, and -3 raised)
            self.conv(**a)
            self.conv(**b)
            return
        self._chk(rc, "conv2d_pair")
        if self.ev_log is not None:
            e1.record()
            tag = f"conv_igemm_glds_kernel[wdir-pair]<{ {L.F32: 'float', L.BF16: 'bf16', L.F16: 'f16'}[self.dtype] },64,128,kb128,s4>"
            if self.ev_shapes:
                la, lb = a["layer"], b["layer"]
                tag += f" {pa.N}x{pa.H}x{pa.W} {la.cin}->{pa.Cout} {pa.KH}x{pa.KW} || {lb.cin}->{pb.Cout} {pb.KH}x{pb.KW}"
            self.ev_log.append((tag, fa + fb, e0, e1))

    def resize_planes(self, src, scale):
        *lead, h, w = src.shape
        ho, wo = int(math.floor(h * scale)), int(math.floor(w * scale))
        dst = self.f32(*lead, ho, wo)
        planes = src.numel() // (h * w)
        self._chk(self.lib.resize_planes_f32(src.data_ptr(), dst.data_ptr(), planes, h, w, ho, wo,
                                             float(1.0 / scale), self.stream()), "resize_planes_f32")
        return dst

    def prep_images(self, img_xs):
        b, _, _, h, w = img_xs.shape
        act = torch.empty((2 * b, h, w, 8), dtype=self.tdtype, device=self.device)
        img4 = self.f32(2 * b, h, w, 4)
        self._chk(self.lib.prep_images(img_xs.data_ptr(), act.data_ptr(), img4.data_ptr(), b, h, w, self.dtype,
                                       self.stream()), "prep_images")
        return act, img4

    def stats_tensor(self, n, c):
        """Zeroed InstanceNorm statistics of n images x c channels: two 64-bit fixed-point sums each (gvfi_conv_params.stats)."""
        return self.f32(n, c, 4, zero=True)

    @staticmethod
    def stats_values(stats):
        """[n, c, 2] float (sum, sum of squares) of a statistics tensor (tests / diagnostics)."""
        q = stats.contiguous().view(torch.int64).to(torch.float64)
        return torch.stack([q[..., 0] / 2.0 ** 24, q[..., 1] / 2.0 ** 20], -1).float()

    def instnorm(self, x, c, relu, res=None, out=None, stats=None):
        """stats: [n, c, 4] (stats_tensor) already accumulated by the producing convolution (Runtime.conv(stats=...)), else None."""
        x = V(x)
        n, h, w = x.t.shape[:3]
        if stats is None:
            stats = self.stats_tensor(n, c)
            self._chk(self.lib.instnorm_stats(x.ptr, x.ld, c, n, h * w, stats.data_ptr(), self.dtype, self.stream()),
                      "instnorm_stats")
        out = V(self.act(n, h, w, c) if out is None else out)
        r = None if res is None else V(res)
        self._chk(self.lib.instnorm_apply(x.ptr, x.ld, c, n, h * w, stats.data_ptr(), 1 if relu else 0,
                                          None if r is None else r.ptr, 0 if r is None else r.ld, out.ptr, out.ld,
                                          self.dtype, self.stream()), "instnorm_apply")
        return out

    def avgpool2(self, src, maps, h, w):
        dst = self.f32(maps, h // 2, w // 2)
        self._chk(self.lib.avgpool2_f32(src.data_ptr(), dst.data_ptr(), maps, h, w, self.stream()), "avgpool2_f32")
        return dst

    def corr_lookup(self, pyr, coords, out, n, h, w, h2, w2, radius=4, src_n=0):
        """src_n: the pyramid holds src_n images and query image i reads image i % src_n (0: one map per query image)."""
        out = V(out)
        fn = self.lib.corr_lookup_lds if (self.lookup_lds and radius == 4) else self.lib.corr_lookup
        self._chk(fn(pyr[0].data_ptr(), pyr[1].data_ptr(), pyr[2].data_ptr(), pyr[3].data_ptr(),
                                       coords.data_ptr(), out.ptr, out.ld, self.dtype, n, src_n, h, w, h2, w2, radius,
                                       self.stream()), "corr_lookup")

    def coords_init(self, n, h, w):
        c = self.f32(n, h, w, 2)
        self._chk(self.lib.coords_init(c.data_ptr(), n, h, w, self.stream()), "coords_init")
        return c

    def token_chain(self, ch, in0, out2, in1=None, res0=None, out1=None, coords=None, period=0):
        """in0 / in1 / res0 / out1 / out2: Views of [rows, ld] token matrices (activation type); see gvfi_token_chain."""
        p = L.TokenChainParams()
        in0, out2 = V(in0), V(out2)
        rows = in0.npix
        p.in0, p.ld0, p.k0a = in0.ptr, in0.ld, in0.c
        if in1 is not None:
            in1 = V(in1)
            assert in0.c + in1.c == 128 and in1.npix == rows
            p.in1, p.ld1 = in1.ptr, in1.ld
        else:
            assert in0.c == 128
            p.in1, p.ld1 = None, 0
        p.wfrag, p.bias = ch.wfrag.data_ptr(), ch.bias.data_ptr()
        p.ln_g, p.ln_b, p.eps, p.ln_after = ch.ln_g.data_ptr(), ch.ln_b.data_ptr(), ch.eps, ch.ln_after
        if coords is not None:
            assert coords.dtype == torch.float32 and coords.is_contiguous()
            p.coords, p.period = coords.data_ptr(), period
        else:
            p.coords, p.period = None, 0
        p.act0, p.act1, p.res2_from0 = ch.act0, ch.act1, int(ch.res2_from0)
        if res0 is not None:
            res0 = V(res0)
            assert res0.c == 64 and not res0.is_f32
            p.res0, p.ldr0 = res0.ptr, res0.ld
        else:
            p.res0, p.ldr0 = None, 0
        if out1 is not None:
            out1 = V(out1)
            p.out1, p.ldo1 = out1.ptr, out1.ld
        else:
            p.out1, p.ldo1 = None, 0
        assert out2.c == 64 and out2.npix == rows
        p.out2, p.ldo2 = out2.ptr, out2.ld
        p.rows, p.dtype = rows, self.dtype
        self._chk(self.lib.token_chain(C.byref(p), self.stream()), "token_chain")

    def s2d_conv(self, layer, x, out, **kw):
        """x: contiguous [N, H, W, ld] activation tensor (H, W multiples of layer.k), out: [N, H/k, W/k, cout]."""
        k = layer.k
        n, h, w_, ld = x.shape
        assert x.is_contiguous() and ld == layer.ld and h % k == 0 and w_ % k == 0, (x.shape, layer.ld, k)
        out = V(out)
        assert out.t.is_contiguous() and out.t.shape[:3] == (n, h // k, w_ // k)
        xv = x.view(n * (h // k), k, w_ // k, k * ld)
        ov = View(out.t.view(n * (h // k), 1, w_ // k, out.t.shape[-1]), out.coff, out.c)
        return self.conv(layer.inner, View(xv, 0, k * ld), ov, **kw)

    def patch_conv(self, layer, src, out, scratch=None, **kw):
        """conv of a PatchConvLayer: im2col into `scratch` [N,H,W,kpad] then a 1x1 convolution."""
        src = V(src)
        n, h, w = src.t.shape[:3]
        if scratch is None:
            scratch = torch.empty((n, h, w, layer.kpad), dtype=self.tdtype, device=self.device)
        self._chk(self.lib.im2col(src.ptr, src.ld, layer.cin, n, h, w, layer.kh, layer.kw, layer.kh // 2, layer.kw // 2,
                                  scratch.data_ptr(), layer.kpad, self.dtype, self.stream()), "im2col")
        return self.conv(layer.inner, scratch, out, **kw)

    def tap_split_conv(self, layer, src, out, res=None, scratch=None):
        """out (f32 view) = conv(src) [+ res]; res may alias out.  scratch: f32 [N,H,W,>=KH*KW*cout]."""
        src = V(src)
        out = V(out)
        n, h, w = src.t.shape[:3]
        k = layer.kh * layer.kw * layer.cout
        if scratch is None:
            scratch = self.f32(n, h, w, roundup(k, 4))
        self.conv(layer.inner, src, View(scratch, 0, k))
        r = None if res is None else V(res)
        assert out.is_f32 and (r is None or r.is_f32)
        self._chk(self.lib.tap_sum(scratch.data_ptr(), scratch.shape[-1], layer.cout, layer.kh, layer.kw,
                                   None if layer.b is None else layer.b.data_ptr(),
                                   None if r is None else r.ptr, 0 if r is None else r.ld, out.ptr, out.ld, n, h, w,
                                   self.stream()), "tap_sum")
        return out

    def flow_step(self, tap_layer, patch_layer, pending, coords1, fl, xb, col, coords_out=None):
        """The seam between two update iterations as one launch (gvfi_flow_step): coords_out = coords1 + tap sum of `pending`
        (the per-tap partial sums the flow head's 1x1 convolution of the PREVIOUS iteration wrote; None: no update), flow ->
        fl / xb, 7x7 im2col of the flow -> col (the input of patch_layer.inner).  Returns the tensor that now holds coords1."""
        fl, xb = V(fl), (None if xb is None else V(xb))
        n, h, w = coords1.shape[:3]
        assert (patch_layer.kh, patch_layer.kw, patch_layer.cin) == (7, 7, 2) and col.shape[-1] == patch_layer.kpad
        assert (tap_layer.kh, tap_layer.kw, tap_layer.cout) == (3, 3, 2)
        self._chk(self.lib.flow_step(None if pending is None else pending.data_ptr(), 0 if pending is None else pending.shape[-1],
                                     None if tap_layer.b is None else tap_layer.b.data_ptr(), coords1.data_ptr(),
                                     None if pending is None else coords_out.data_ptr(), fl.ptr, fl.ld,
                                     fl.ld, None if xb is None else xb.ptr, 0 if xb is None else xb.ld, col.data_ptr(),
                                     col.shape[-1], n, h, w, self.dtype, self.stream()), "flow_step")
        return coords1 if pending is None else coords_out

    def inr_mlp(self, mlp, lat, coord, out):
        """out[B,H,W,2] (f32) = hypo-network(lat[..., :32], coord[B,1,H,W,3])."""
        lat = V(lat)
        npix = out.numel() // 2
        self._chk(self.lib.inr_mlp(lat.ptr, lat.ld, coord.data_ptr(), mlp.wfrag.data_ptr(), mlp.bias.data_ptr(),
                                   out.data_ptr(), npix, self.dtype, self.stream()), "inr_mlp")
        return out

    def flow_pack(self, coords1, dst0, dst1):
        n, h, w = coords1.shape[:3]
        dst0, dst1 = V(dst0), V(dst1)
        self._chk(self.lib.flow_pack(coords1.data_ptr(), dst0.ptr, dst0.ld, dst0.ld, dst1.ptr, dst1.ld, n, h, w,
                                     self.dtype, self.stream()), "flow_pack")

    def convex_upsample(self, coords1, mask):
        n, h, w = coords1.shape[:3]
        mask = V(mask)
        out = self.f32(n, 8 * h, 8 * w, 2)
        self._chk(self.lib.convex_upsample(coords1.data_ptr(), mask.ptr, mask.ld, mask.is_f32, out.data_ptr(), n, h, w,
                                           self.dtype, self.stream()), "convex_upsample")
        return out

    def resize(self, src, c, scale, mul=1.0, out=None, out_f32=None, rscale=None, size=None):
        """dst = mul * bilinear_resize(src) (align_corners=False).  Either scale (torch scale_factor
        semantics) or size=(Ho,Wo) (torch size semantics: rscale = in/out)."""
        src = V(src)
        n, h, w = src.t.shape[:3]
        if size is None:
            ho, wo = int(math.floor(h * scale)), int(math.floor(w * scale))
            rs = float(1.0 / scale)
            rs_h = rs_w = rs
        else:
            ho, wo = size
            rs_h, rs_w = h / ho, w / wo
            assert abs(rs_h - rs_w) < 1e-12 or True
        if out is None:
            f32o = src.is_f32 if out_f32 is None else out_f32
            out = self.f32(n, ho, wo, c) if f32o else self.act(n, ho, wo, c)
        out = V(out)
        assert out.t.shape[1] == ho and out.t.shape[2] == wo
        if size is not None and rs_h != rs_w:
            raise NotImplementedError("anisotropic resize")
        self._chk(self.lib.resize_nhwc(src.ptr, src.ld, src.is_f32, out.ptr, out.ld, out.is_f32, c, n, h, w, ho, wo,
                                       float(rs_h), float(mul), self.dtype, self.stream()), "resize_nhwc")
        return out

    def warp(self, src, c, flow, out, fmul=1.0):
        """A source with fewer images than `out` is read modulo its batch (out image n <- src image n % src_n)."""
        src, out, flow = V(src), V(out), V(flow)
        n, h, w = out.t.shape[:3]
        assert src.t.shape[1] == h and src.t.shape[2] == w and flow.t.shape[1] == h and flow.t.shape[0] == n
        sn = src.t.shape[0]
        assert sn == n or (0 < sn < n and n % sn == 0), (sn, n)
        self._chk(self.lib.warp_nhwc(src.ptr, src.ld, src.is_f32, flow.ptr, flow.ld, float(fmul), out.ptr, out.ld,
                                     out.is_f32, c, n, 0 if sn == n else sn, h, w, self.dtype, self.stream()), "warp_nhwc")
        return out

    def pixel_shuffle2(self, src, cout):
        src = V(src)
        n, h, w = src.t.shape[:3]
        out = self.act(n, 2 * h, 2 * w, cout)
        self._chk(self.lib.pixel_shuffle2(src.ptr, src.ld, out.data_ptr(), out.shape[-1], cout, n, h, w, self.dtype,
                                          self.stream()), "pixel_shuffle2")
        return out

    def copy(self, src, dst, c, mul=1.0, add=None):
        """A source with fewer pixels than `dst` (a whole divisor) is read modulo its size: broadcast over timesteps."""
        src, dst = V(src), V(dst)
        a = None if add is None else V(add)
        sp = src.npix
        assert sp >= dst.npix or (sp > 0 and dst.npix % sp == 0), (sp, dst.npix)
        self._chk(self.lib.copy_channels(src.ptr, src.ld, src.is_f32, None if a is None else a.ptr,
                                         0 if a is None else a.ld, 0 if a is None else a.is_f32, dst.ptr, dst.ld,
                                         dst.is_f32, c, float(mul), dst.npix, 0 if sp >= dst.npix else sp, self.dtype,
                                         self.stream()), "copy_channels")
        return dst

    # ------------------------------------------------------------------ FlowFormer glue (csrc/flowformer_ops.hip)
    def layernorm(self, x, gb, eps, out=None):
        """nn.LayerNorm over the channels of a token matrix / NHWC tensor (gb = (gamma, beta) float tensors)."""
        x = V(x)
        if out is None:
            out = torch.empty(x.t.shape, dtype=self.tdtype, device=self.device)
        o = V(out)
        assert not o.is_f32 or self.dtype == L.F32
        self._chk(self.lib.layernorm(x.ptr, x.ld, x.is_f32, gb[0].data_ptr(), gb[1].data_ptr(), float(eps), o.ptr, o.ld,
                                     x.npix, x.c, self.dtype, self.stream()), "layernorm")
        return out

    def dwconv3x3_res(self, x, w9c, bias):
        n, h, w, c = x.shape
        out = torch.empty_like(x)
        self._chk(self.lib.dwconv3x3_res(x.data_ptr(), c, w9c.data_ptr(), bias.data_ptr(), out.data_ptr(), c,
                                         1 if x.dtype == torch.float32 else 0, n, h, w, c, self.dtype, self.stream()),
                  "dwconv3x3_res")
        return out

    def pos_embed(self, coords, period, scale, offset, dim, out, rows, accumulate):
        o = V(out)
        self._chk(self.lib.pos_embed(coords.data_ptr(), period, float(scale), float(offset), dim, o.ptr, o.ld, rows,
                                     1 if accumulate else 0, self.dtype, self.stream()), "pos_embed")

    def cost_embed1(self, vol, w, b, maps, h, w_, ho, wo):
        out = self.act(maps, ho, wo, 16)
        self._chk(self.lib.cost_embed1(vol.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), out.shape[-1], maps,
                                       h, w_, ho, wo, self.dtype, self.stream()), "cost_embed1")
        return out

    def cost_lookup(self, vol, coords, out, q, h, w, radius=4):
        o = V(out)
        self._chk(self.lib.cost_lookup(vol.data_ptr(), coords.data_ptr(), o.ptr, o.ld, q, h, w, radius, self.dtype,
                                       self.stream()), "cost_lookup")

    def attn_window(self, q, k, v, kpad, vpad, out, n_img, h, w, ws, heads, hd):
        q, k, v, o = V(q), V(k), V(v), V(out)
        self._chk(self.lib.attn_window(q.ptr, q.ld, k.ptr, k.ld, v.ptr, v.ld, kpad.data_ptr(), vpad.data_ptr(), o.ptr,
                                       o.ld, n_img, h, w, ws, heads, hd, float(hd ** -0.5), self.dtype, self.stream()),
                  "attn_window")
        return out

    def attn_global(self, q, qrow, k, v, krow, out, orow, g1, g0, nq, m, heads, hd):
        """qrow / krow / orow = (b1, b0, s): row = g1*b1 + g0*b0 + i*s (see gvfi_attn_global)."""
        q, k, v, o = V(q), V(k), V(v), V(out)
        self._chk(self.lib.attn_global(q.ptr, q.ld, *qrow, k.ptr, k.ld, v.ptr, v.ld, *krow, o.ptr, o.ld, *orow, g1, g0, nq,
                                       m, heads, hd, float(hd ** -0.5), self.dtype, self.stream()), "attn_global")
        return out

    def ff_xqk(self, x, ctx, out, n_img, h, w, k, nb, enc_mode, ws=7, table=None):
        x, c, o = V(x), V(ctx), V(out)
        self._chk(self.lib.ff_xqk(x.ptr, x.ld, x.c, c.ptr, c.ld, c.c, o.ptr, o.ld, n_img, h, w, k, nb, enc_mode, ws,
                                  None if table is None else table.data_ptr(), self.dtype, self.stream()), "ff_xqk")
        return out

    def ff_pos_table(self, h, w, ct, enc_mode, ws=7):
        t = self.f32(ws * ws if enc_mode == 1 else h * w, ct)
        self._chk(self.lib.ff_pos_table(t.data_ptr(), h, w, ct, enc_mode, ws, self.stream()), "ff_pos_table")
        return t

    def tile_rows(self, table, out, rows, p, k, c):
        o = V(out)
        self._chk(self.lib.tile_rows(table.data_ptr(), o.ptr, o.ld, o.is_f32, rows, p, k, c, self.dtype, self.stream()),
                  "tile_rows")
        return out

    def softmax_rows(self, x, n, out, rows):
        o = V(out)
        self._chk(self.lib.softmax_rows(x.data_ptr(), n, o.ptr, o.ld, rows, self.dtype, self.stream()), "softmax_rows")
        return out

    def flow_to_image(self, flows, wheel, bgr=True):
        """flows: [n, 2, h, w] float (contiguous) -> [n, h, w, 3] uint8 pictures (reference flow_viz.flow_to_image per image)."""
        n, _, h, w = flows.shape
        assert flows.dtype == torch.float32 and flows.is_contiguous(), "flow_to_image reads raw [n,2,h,w] float planes"
        out = torch.empty((n, h, w, 3), dtype=torch.uint8, device=self.device)
        scratch = torch.zeros(n, dtype=torch.int32, device=self.device)
        self._chk(self.lib.flow_to_image(flows.data_ptr(), 2 * h * w, n, h, w, wheel.data_ptr(), scratch.data_ptr(), out.data_ptr(),
                                         1 if bgr else 0, self.stream()), "flow_to_image")
        return out

    def frames_to_u8(self, frames_nchw):
        b, _, h, w = frames_nchw.shape
        out = torch.empty((b, h, w, 3), dtype=torch.uint8, device=self.device)
        self._chk(self.lib.frames_to_u8(frames_nchw.data_ptr(), out.data_ptr(), b, h, w, self.stream()), "frames_to_u8")
        return out

    def compose_sbs(self, frames, pad_top, pad_left, pred_u8, n_interp, lead):
        """[orig | interpolated] video frames of a block of consecutive pairs (gvfi_compose_sbs_u8): frames (b+1, 3, Hp, Wp) float,
        pred_u8 [b, N-1, H0, W0, 3] RGB -> [b * N + lead, H0, 2 * W0, 3] BGR uint8."""
        b, nm1, h0, w0, _ = pred_u8.shape
        assert frames.dtype == torch.float32 and frames.is_contiguous() and frames.shape[0] == b + 1 and frames.shape[1] == 3
        assert pred_u8.dtype == torch.uint8 and pred_u8.is_contiguous() and nm1 == n_interp - 1
        out = torch.empty((b * n_interp + int(lead), h0, 2 * w0, 3), dtype=torch.uint8, device=self.device)
        self._chk(self.lib.compose_sbs_u8(frames.data_ptr(), b + 1, frames.shape[2], frames.shape[3], int(pad_top), int(pad_left),
                                          pred_u8.data_ptr(), b, n_interp, int(lead), out.data_ptr(), h0, w0, self.stream()),
                  "compose_sbs_u8")
        return out

    def nhwc_to_nchw(self, src, c):
        src = V(src)
        assert src.is_f32
        n, h, w = src.t.shape[:3]
        out = self.f32(n, c, h, w)
        self._chk(self.lib.nhwc_to_nchw_f32(src.ptr, src.ld, out.data_ptr(), c, n, h, w, self.stream()),
                  "nhwc_to_nchw_f32")
        return out
