"""``GIMMVFI_R``: drop-in module for reference generalizable_INR/gimmvfi_r.py:34-442.

Same constructor config fields, same ``forward(img_xs, coord, t, iters, ds_factor)`` /
``sample_coord_input`` signatures and return dict, and the same 414 state_dict keys (so
``load_state_dict(ckpt["state_dict"], strict=True)`` of reference checkpoints works,
src/video_Nx.py:114-115).  Differences by design:

* inference only (no autograd graph); compute runs on the HIP kernels of
  libgimmvfi_hip.so via ``Engine`` -- there is no torch/CPU fallback;
* the constructor does not read ``pretrained_ckpt/raft-things.pth`` (reference
  raft/__init__.py:7-24): RAFT weights are part of the GIMM-VFI checkpoint loaded afterwards;
* ``precision``: "bf16" (default, MFMA bf16 with fp32 accumulation; flows, coordinates,
  correlation volumes and splat sums stay fp32) or "fp32" (exact-f32 MFMA validation mode).
"""
import os

import torch
import torch.nn as nn

from . import lib as L
from .engine import Engine
from .ops import Runtime
from .params import (gimm_param_spec, gimm_state_dict, param_spec, param_spec_f, random_state_dict,
                     random_state_dict_f)


class _Node(nn.Module):
    """Pure container; mirrors the reference module tree so state_dict keys match."""


_BUFFER_SUFFIXES = ("running_mean", "running_var", "num_batches_tracked")


def _cfg_get(cfg, key, default=None):
    if cfg is None:
        return default
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


def check_supported_config(config):
    """The kernels hard-wire the architecture values of the reference's shipped configs
    (configs/gimmvfi/gimmvfi_{r,f}_arb.yaml, configs/gimm/gimm.yaml + the dataclass defaults of
    generalizable_INR/configs.py:38-57, modules/module_config.py:28-41).  Anything else would silently compute a
    different function than the reference does with that yaml, so it is rejected here."""
    if config is None:
        return
    want = {"fwarp_type": "linear"}
    for k, v in want.items():
        got = _cfg_get(config, k, v)
        if got != v:
            raise ValueError(f"gimmvfi_hip supports arch.{k} = {v!r} only (got {got!r})")
    hyp = _cfg_get(config, "hyponet")
    if hyp is None:
        return
    hwant = {"type": "mlp", "n_layer": 5, "use_bias": True, "input_dim": 3, "output_dim": 2, "output_bias": 0.5,
             "normalize_weight": True, "linear_interpo": False}
    for k, v in hwant.items():
        got = _cfg_get(hyp, k, v)
        if got != v:
            raise ValueError(f"gimmvfi_hip supports arch.hyponet.{k} = {v!r} only (got {got!r})")
    hd = _cfg_get(hyp, "hidden_dim", [128])
    if list(hd) != [128]:
        raise ValueError(f"gimmvfi_hip supports arch.hyponet.hidden_dim = [128] only (got {list(hd)!r})")
    act = _cfg_get(hyp, "activation")
    if act is not None:
        if _cfg_get(act, "type", "siren") != "siren" or float(_cfg_get(act, "siren_w0", 1.0)) != 1.0:
            raise ValueError("gimmvfi_hip supports arch.hyponet.activation = {type: siren, siren_w0: 1.0} only")
    mod = _cfg_get(config, "modulated_layer_idxs")
    if mod is not None and list(mod) != [1]:
        raise ValueError(f"gimmvfi_hip supports arch.modulated_layer_idxs = [1] only (got {list(mod)!r})")


class GIMMVFI_R(nn.Module):
    _spec = staticmethod(param_spec)
    _init_sd = staticmethod(random_state_dict)
    _engine_cls = Engine

    def __init__(self, config=None, precision=None):
        super().__init__()
        check_supported_config(config)
        self.config = config
        self.raft_iter = 20  # gimmvfi_r.py:41 (config.raft_iter is ignored by the reference too)
        cfg_prec = None
        if config is not None:
            cfg_prec = config.get("precision") if isinstance(config, dict) else getattr(config, "precision", None)
        self.precision = precision or cfg_prec or os.environ.get("GIMMVFI_PRECISION", "bf16")
        self.coord_range = (-1.0, 1.0)
        if config is not None:
            cr = config.get("coord_range") if isinstance(config, dict) else getattr(config, "coord_range", None)
            if cr is not None:
                self.coord_range = (float(cr[0]), float(cr[1]))
        sd0 = self._init_sd(0)
        for name, shape in self._spec().items():
            parts = name.split(".")
            node = self
            for p in parts[:-1]:
                if p not in node._modules:
                    node.add_module(p, _Node())
                node = node._modules[p]
            val = sd0[name].clone()
            if parts[-1] in _BUFFER_SUFFIXES or not val.is_floating_point():
                node.register_buffer(parts[-1], val)
            else:
                node.register_parameter(parts[-1], nn.Parameter(val, requires_grad=False))
        self._engine = None
        self._engine_key = None
        # hipGraph replay of the whole forward (one graph per input signature): the launch list is ~1000 kernels,
        # so eager mode is host-bound as soon as the kernels are fast.  GIMMVFI_GRAPH=0 disables it.
        self.use_graph = os.environ.get("GIMMVFI_GRAPH", "1") != "0"
        self._graphs = {}
        self.max_graphs = 4      # captured graphs keep their intermediates alive (GBs at 2K): bounded cache
        # Graph replays return fresh tensors (clones), like the reference's modules.  A caller that consumes the outputs on
        # the launch stream before its next forward() of the same signature -- the CLI, bench.py: both convert them to uint8
        # frames right away -- can opt out: the outputs are then the graph's own static tensors, valid until that next
        # forward (at 4K x 7 timesteps the clones were 87 copies / 1.4 ms of a 72 ms step; 38 copies / 0.15 ms at 448x256).
        self.static_outputs = os.environ.get("GIMMVFI_STATIC_OUTPUTS", "0") == "1"
        # serial_launch: the engine's parallel launch sequences off (Engine.set_serial_launch): a linear graph per forward.  Slower for
        # one forward at a time (-3 ... -7 %); one of the slot kinds of StepsInFlight.
        self.serial_launch = False

    # ---- engine cache invalidation: weights are folded/packed for the kernels lazily
    def load_state_dict(self, state_dict, strict=True, **kw):
        r = super().load_state_dict(state_dict, strict=strict, **kw)
        self._engine = None
        self._graphs = {}
        return r

    def _apply(self, fn, *a, **kw):
        r = super()._apply(fn, *a, **kw)
        self._engine = None
        self._graphs = {}
        return r

    def engine(self, device=None, runtime=None):
        device = torch.device(device) if device is not None else next(self.parameters()).device
        key = (str(device), self.precision, id(runtime), getattr(self, "flow_precision", None))
        if self._engine is None or self._engine_key != key:
            if runtime is None:
                if device.type != "cuda":
                    raise RuntimeError(
                        "GIMMVFI_R runs only on an MI355X through libgimmvfi_hip.so: move the model and inputs to "
                        "'cuda' (there is no CPU fallback)"
                    )
                runtime = Runtime(L.get(), self.precision, device)
            self._engine = self._make_engine(runtime)
            self._engine_key = key
            self._graphs = {}     # graphs captured on the previous engine replay ITS buffers and packed weights
        if getattr(self._engine, "_serial_applied", None) != self.serial_launch:
            self._engine.set_serial_launch(self.serial_launch)
            self._engine._serial_applied = self.serial_launch
            self._graphs = {}     # (captured with the other launch structure)
        return self._engine

    def _make_engine(self, runtime):
        return self._engine_cls(runtime, self.state_dict())

    # ---- reference API -------------------------------------------------------------------
    def forward(self, img_xs, coord=None, t=None, iters=None, ds_factor=None, _seq=False):
        assert isinstance(t, list)
        assert isinstance(coord, list)
        assert len(t) == len(coord)
        iters = self._iters()
        eng = self.engine(img_xs.device)
        if not (self.use_graph and img_xs.is_cuda and eng.rt.ev_log is None):
            self._eager_scope(eng, img_xs, coord, t, ds_factor, _seq)
            return eng.forward(img_xs, coord, t, iters=iters, ds_factor=ds_factor, seq=_seq)
        try:
            return self._forward_graph(eng, img_xs, coord, t, iters, ds_factor, _seq)
        except RuntimeError as e:
            # capture can fail for reasons outside this package (another capture in progress, allocator limits);
            # the eager launch list is the same kernels -- never a different arithmetic path
            if "capture" not in str(e).lower() and "graph" not in str(e).lower():
                raise
            import warnings

            warnings.warn(f"gimmvfi_hip: hipGraph capture failed ({e}); this model now launches eagerly (same kernels)")
            self.use_graph = False
            self._graphs = {}
            torch.cuda.synchronize(img_xs.device)
            return eng.forward(img_xs, coord, t, iters=iters, ds_factor=ds_factor, seq=_seq)

    def _eager_scope(self, eng, img_xs, coord, t, ds_factor, seq):
        """Eager launches keep the runtime's zero-once buffers per input signature like the captured graphs do: at most
        `max_graphs` signatures, the least recently seen one is released (Runtime.release_once)."""
        key = ("eager", tuple(img_xs.shape), tuple(tuple(c[0].shape) for c in coord), len(t), ds_factor, seq)
        lru = self.__dict__.setdefault("_eager_sigs", [])
        if key in lru:
            lru.remove(key)
        lru.append(key)
        while len(lru) > self.max_graphs:
            eng.rt.release_once(lru.pop(0))
        eng.rt.once_scope = key

    def forward_sequence(self, frames, coord=None, t=None, ds_factor=None):
        """Addition to the reference API for video: `frames` (B+1, 3, H, W) are consecutive frames; returns what
        forward() returns for the B adjacent pairs (frames[b], frames[b+1]) as one batch, with the per-frame encoder work
        (RAFT fnet / cnet, Twins) done once per frame instead of once per pair end (src/video_Nx.py feeds every interior
        frame to the reference twice)."""
        assert frames.dim() == 4 and frames.shape[0] >= 2
        img_xs = torch.stack([frames[:-1], frames[1:]], dim=2)
        return GIMMVFI_R.forward(self, img_xs, coord=coord, t=t, ds_factor=ds_factor, _seq=True)

    def _iters(self):
        return self.raft_iter  # gimmvfi_r.py:127-132 hard-codes 20

    def _ctor_kwargs(self):
        return {"config": self.config, "precision": self.precision}

    def replica(self):
        """A second instance of this model on the same device with the same weights and switches and its OWN engine (packed
        weights, buffers, captured graphs) -- what StepsInFlight keeps per slot."""
        r = type(self)(**self._ctor_kwargs())
        r.load_state_dict(self.state_dict(), strict=True)
        r = r.to(next(self.parameters()).device).eval()
        r.use_graph, r.static_outputs, r.max_graphs, r.raft_iter = self.use_graph, self.static_outputs, self.max_graphs, self.raft_iter
        r.serial_launch = self.serial_launch
        return r

    def _forward_graph(self, eng, img_xs, coord, t, iters, ds_factor, seq=False):
        """Capture once per input signature, then replay; inputs are copied into the graph's static buffers and
        the outputs are returned as fresh tensors (clones), like the eager path."""
        for c in coord:
            assert isinstance(c, tuple) and c[1] is None, "sub-sampled coordinates are a training feature"
        key = (tuple(img_xs.shape), str(img_xs.device), tuple(tuple(c[0].shape) for c in coord), len(t), ds_factor, seq)
        ent = self._graphs.get(key)
        if ent is None and len(self._graphs) >= self.max_graphs:
            old = next(iter(self._graphs))
            self._graphs.pop(old)            # oldest signature: its private memory pool is released ...
            eng.rt.release_once(old)         # ... and the runtime's zero-once buffers only it used (they live outside the pool)
        eng.rt.once_scope = key
        if ent is None:
            dev = img_xs.device
            sx = img_xs.detach().to(torch.float32).contiguous().clone()
            sc = [c[0].detach().to(device=dev, dtype=torch.float32).contiguous().clone() for c in coord]
            st = [ti.detach().to(device=dev, dtype=torch.float32).reshape(-1).contiguous().clone() for ti in t]
            coords = [(c, None) for c in sc]
            cur = torch.cuda.current_stream(dev)
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):   # warm-up outside capture (one-time attribute / allocator work)
                eng.forward(sx, coords, st, iters=iters, ds_factor=ds_factor, seq=seq)
            cur.wait_stream(side)
            torch.cuda.synchronize(dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = eng.forward(sx, coords, st, iters=iters, ds_factor=ds_factor, seq=seq)
            ent = (g, sx, sc, st, out)
            self._graphs[key] = ent
        g, sx, sc, st, out = ent
        sx.copy_(img_xs)
        for a, c in zip(sc, coord):
            a.copy_(c[0])
        for a, ti in zip(st, t):
            a.copy_(ti.reshape(-1))
        g.replay()

        keep = self.static_outputs

        def fresh(o):       # (containers are always rebuilt: the caller may edit the dict / lists it gets)
            if isinstance(o, torch.Tensor):
                return o if keep else o.clone()
            if isinstance(o, (list, tuple)):
                return type(o)(fresh(v) for v in o)
            if isinstance(o, dict):
                return {k: fresh(v) for k, v in o.items()}
            return o

        return fresh(out)

    def sample_coord_input(self, batch_size, s_shape, t_ids, coord_range=None, upsample_ratio=1.0, device=None):
        """modules/coord_sampler.py:15-43,73-91 (list t_ids) -> (B, T, H', W', 3) ordered (t, y, x).
        Pure index generation (host plumbing, no arithmetic of the hot path)."""
        assert device is not None
        assert coord_range is None
        lo, hi = self.coord_range
        assert isinstance(t_ids, list)
        cs = [torch.tensor(t_ids, device=device, dtype=torch.float32) / 1.0]
        for n in s_shape:
            n = int(n * upsample_ratio)
            c = (0.5 + torch.arange(n, device=device)) / n
            cs.append(lo + (hi - lo) * c)
        g = torch.stack(torch.meshgrid(*cs, indexing="ij"), dim=-1)
        return g.unsqueeze(0).repeat(batch_size, 1, 1, 1, 1)

    def compute_psnr(self, preds, targets, reduction="mean"):
        # gimmvfi_r.py:412-426
        assert reduction in ["mean", "sum", "none"]
        b = preds.shape[0]
        mse = torch.reshape((preds - targets) ** 2, (b, -1)).mean(dim=-1)
        psnr = -10 * torch.log10(mse)
        return psnr.mean() if reduction == "mean" else (psnr.sum() if reduction == "sum" else psnr)


class GIMMVFI_F(GIMMVFI_R):
    """Drop-in for reference generalizable_INR/gimmvfi_f.py:27-419: the same model with the FlowFormer flow estimator
    (Twins-SVT encoders, latent cost-volume encoder, 32-iteration GMA decoder) instead of RAFT.  Same 639
    state_dict keys (``load_state_dict(strict=True)`` of reference checkpoints), ``forward(img_xs, coord, t,
    ds_factor)`` without an ``iters`` argument (gimmvfi_f.py:304).  The constructor does not read
    ``pretrained_ckpt/flowformer_sintel.pth`` (flowformer/__init__.py:10): those weights are part of the GIMM-VFI-F
    checkpoint loaded afterwards.  Working resolutions are multiples of 8, >= 128 (as for GIMM-VFI-R); grids that are not
    multiples of the window / sub-sampling sizes take the reference's zero-extension branches."""

    _spec = staticmethod(param_spec_f)
    _init_sd = staticmethod(random_state_dict_f)

    def __init__(self, config=None, precision=None, flow_precision=None):
        """flow_precision (with precision "bf16"): precision policy of the FlowFormer flow estimator only -- a comma list
        of stage[:type] entries for the stages that leave bf16: enc (Twins encoders), cost (cost volume + latent cost
        encoder), tok / upd (flow-token path / GMA update block of the 32-iteration decoder; dec = both); type f16 = IEEE
        half operands (same MFMA rate as bf16, 11 instead of 8 significand bits) or fp32 (exact-f32 MFMA, the default
        type); "fp32" alone = all stages in float, "bf16" = none.  Default: config.flow_precision,
        $GIMMVFI_F_FLOW_PRECISION, else "f16" (= "enc:f16,cost:f16,dec:f16": the whole flow estimator on IEEE-half operands).
        Measured against the reference's own outputs (profiles/r3_f_policy.md, r5_f_policy_enc_cost.txt, r5_f_policy_all.txt):
        the update block is the stage whose bf16 operand rounding moves the flow by pixels when the flows are large -- all-bf16
        32.3-39.8 dB at 2K / 4K with 40-50 px flows (>= 52 dB once the flows stay below 10 px), "dec:f16" 41.6-50.4 dB at the
        same speed; the Twins encoders' rounding costs the two hardest fixtures another 2 - 2.7 dB: "f16" 43.7-53.0 dB at
        -1.2 % (448x256, B = 8) / -0.3 % (4K) once the MFMA attentions and the 8-wave tile run on half operands (round 5);
        "dec" (float decoder) 42.9-51.9 dB at half the speed, "fp32" 46.2 dB on the hardest fixture.  DESIGN.md section 9."""
        super().__init__(config, precision)
        cfg_fp = _cfg_get(config, "flow_precision")
        self.flow_precision = flow_precision or cfg_fp or os.environ.get("GIMMVFI_F_FLOW_PRECISION", "f16")

    def _make_engine(self, runtime):
        return self._engine_cls(runtime, self.state_dict(), flow_precision=self.flow_precision)

    def _ctor_kwargs(self):
        return {"config": self.config, "precision": self.precision, "flow_precision": self.flow_precision}

    @property
    def _engine_cls(self):
        from .engine_f import EngineF

        return EngineF

    def _iters(self):
        return None   # decoder_depth = 32 (flowformer/configs/submission.py:52, decoder.py:289-290)

    def forward(self, img_xs, coord=None, t=None, ds_factor=None, _seq=False):
        return super().forward(img_xs, coord=coord, t=t, iters=None, ds_factor=ds_factor, _seq=_seq)


class StepsInFlight:
    """Addition to the reference API for throughput: `depth` independent steps in flight on one GPU.  Slot k is a replica of the
    model (its own engine, buffers and captured graphs) with its own stream; submit() launches one forward on the next slot's
    stream and returns at once, so consecutive steps overlap on the device at kernel granularity -- two latency-bound launch
    chains fill each other's gaps (profiles/r6_steps_in_flight.txt, r6_queue_probe*.txt: +5 ... +14 % at 448x256, batch 8; every
    step's output bit-identical to the same step run alone -- the forward is bit-reproducible, tests/test_gpu_e2e.py).  The
    arithmetic of a step is untouched: the same kernels as model.forward().

        pipe = StepsInFlight(model, depth=2)
        pipe.calibrate(img_xs, coord, t)      # once per input signature (set-up, like capturing the graphs): see calibrate()
        h = pipe.submit(img_xs, coord, t, ds_factor=None, then=lambda out, m: to_u8(out["imgt_pred"][0]))
        ...                                   # submit more; the host never blocks
        frames = pipe.wait(h)                 # the caller's stream waits for that step (device-side wait, no host sync)

    `then(out, replica)` runs on the slot's stream right behind the forward (the place to convert / pack the outputs).  With
    model.static_outputs the tensors a step returns are the slot's own graph outputs: valid until that slot's next submit, i.e.
    for `depth` more submits."""

    def __init__(self, model, depth=2, serial=None):
        """serial: True = the slots launch LINEAR graphs (model.serial_launch: the engine's parallel launch sequences off), False =
        the model's own (forked) graphs, None = the model's own until calibrate() has measured both kinds."""
        assert depth >= 1
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("StepsInFlight needs the model on an MI355X ('cuda'): there is no CPU path")
        self.device = dev
        self.model = model
        self._auto = serial is None
        self.serial = model.serial_launch if serial is None else bool(serial)
        self.replicas = self._make_replicas(depth, self.serial)
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
        self.calibration = None
        self._n = 0

    def _make_replicas(self, depth, serial):
        if serial == self.model.serial_launch:
            return [self.model] + [self.model.replica() for _ in range(depth - 1)]
        reps = []                                    # (the model handed in keeps its own setting: every slot is a replica then)
        for _ in range(depth):
            r = self.model.replica()
            r.serial_launch = serial
            reps.append(r)
        return reps

    @property
    def depth(self):
        return len(self.replicas)

    @staticmethod
    def _forward(m, img_xs, coord, t, ds_factor, sequence):
        return m.forward_sequence(img_xs, coord=coord, t=t, ds_factor=ds_factor) if sequence else m(img_xs, coord, t=t, ds_factor=ds_factor)

    def submit(self, img_xs, coord, t, ds_factor=None, then=None, sequence=False):
        k = self._n % len(self.replicas)
        self._n += 1
        m, s = self.replicas[k], self.streams[k]
        s.wait_stream(torch.cuda.current_stream(self.device))          # the inputs were produced on the caller's stream
        with torch.cuda.stream(s):
            out = self._forward(m, img_xs, coord, t, ds_factor, sequence)
            res = then(out, m) if then is not None else out
            ev = torch.cuda.Event()
            ev.record(s)
        return ev, res

    def prime(self, img_xs, coord, t, ds_factor=None, sequence=False):
        """One forward per slot, slot 0 first, each on its own stream: captures every slot's graph (set-up)."""
        for m, st in zip(self.replicas, self.streams):
            with torch.cuda.stream(st):
                self._forward(m, img_xs, coord, t, ds_factor, sequence)
            st.synchronize()

    def wait(self, handle):
        ev, res = handle
        torch.cuda.current_stream(self.device).wait_event(ev)
        return res

    def drain(self):
        for s in self.streams:
            s.synchronize()

    def calibrate(self, img_xs, coord, t, ds_factor=None, steps=8, max_pairs=8, extra_pairs=2, good_enough=1.03, sequence=False):
        """Choose how the two slots are run BY MEASUREMENT (depth 2), because whether two captured forwards in flight overlap
        on this runtime is decided by state nobody controls.  Round 6 measured (profiles/r6_queue_probe.txt), for the same two
        graphs and nothing changed but the pair of launch streams: 199 / 201 / 226 frames/s (F 448x256) and 362 / 372 / 386 / 397
        (R); pairs of streams that alias onto one of HIP's 4 hardware queues do not overlap at all, some other pairs run SLOWER
        than one step at a time (a slot's launch queue colliding with the internal streams the other graph's branches run on),
        whole processes in which no pair lets two forked F graphs overlap, and linear graphs (no branches) that overlap on
        about every second pair of fresh streams.  The outcome is stable for a given pair of graphs and streams, so it can be
        measured once.  Linear slots first (unless the constructor fixed the kind): primed on a fresh pair of streams and timed
        for `steps` steps there, then on further fresh pairs until one beats the model alone by `good_enough` or `max_pairs` are
        spent; then the other kind on the streams it is primed on + `extra_pairs` fresh pairs; the best configuration stays if it
        beats the model alone, one step at a time, by more than 1 % -- otherwise the pipeline degenerates to exactly that
        (depth 1).  Set-up work like capturing the graphs: call it once per input signature before the steady state.  Returns
        the table (steps/s)."""
        import gc
        import time

        if len(self.replicas) != 2:
            return {}
        dev = self.device

        def rate(n=steps):
            best = 0.0
            for rep in range(2):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(n):
                    self.submit(img_xs, coord, t, ds_factor=ds_factor, sequence=sequence)
                torch.cuda.synchronize(dev)
                if rep:                              # (the first repetition warms the configuration up)
                    best = n / (time.perf_counter() - t0)
            return best

        report = {}
        keep = (self.replicas, self.streams, self.serial)
        self.replicas, self.streams = [self.model], [torch.cuda.Stream(device=dev)]
        self.prime(img_xs, coord, t, ds_factor=ds_factor, sequence=sequence)
        base = report["model alone, one step at a time"] = rate()
        self.replicas, self.streams, self.serial = keep
        best = (base * 1.01, None)
        tried = []
        kinds = [True, False] if self._auto else [self.serial]
        for n_kind, kind in enumerate(kinds):
            name = "linear graphs" if kind else "forked graphs"
            reps = keep[0] if kind == keep[2] else self._make_replicas(2, kind)      # (the slots the pipeline was built with / new ones)
            self.replicas = reps
            self.streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
            self.prime(img_xs, coord, t, ds_factor=ds_factor, sequence=sequence)
            tab = report[name] = {}
            for trial in range(max_pairs if n_kind == 0 else 1 + extra_pairs):
                if trial:
                    self.streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
                where = "streams as primed" if trial == 0 else f"fresh stream pair {trial}"
                r = tab[where] = rate()
                if r > best[0]:
                    best = (r, (reps, list(self.streams), kind, name + ", " + where))
                if n_kind == 0 and r >= good_enough * base:
                    break
            tried.append(reps)
            for old in tried:                        # (memory: a captured 4K forward holds tens of GB -- only the leader's graphs stay)
                if best[1] is None or best[1][0] is not old:
                    for m in old:
                        if m is not self.model:
                            m._graphs = {}
            gc.collect()
            torch.cuda.empty_cache()
        if best[1] is None:
            self.replicas, self.streams, self.serial = [self.model], [torch.cuda.Stream(device=dev)], self.model.serial_launch
            report["picked"] = "model alone, one step at a time"
        else:
            self.replicas, self.streams, self.serial, report["picked"] = best[1]
        self.prime(img_xs, coord, t, ds_factor=ds_factor, sequence=sequence)      # (a kind whose graphs were dropped above is re-captured)
        self.calibration = report
        return report


class GIMM(nn.Module):
    """Drop-in for the reference's motion-only model (generalizable_INR/gimm.py:26-253, `create_model` type "gimm",
    used by src/VTF.py / src/VSF.py): flows in, flow at time t out -- the splat metric, motion encoder, softmax
    splat, latent refiner and hypo-network kernels of the GIMM-VFI-R path without flow estimation and synthesis.
    Same 36 state_dict keys, same ``forward(xs, coord, keep_xs_shape, ori_flow, timesteps)`` /
    ``sample_coord_input`` / ``compute_loss`` surface; inference only, no CPU fallback."""

    def __init__(self, config=None, precision=None):
        super().__init__()
        check_supported_config(config)
        self.config = config
        cfg_prec = None
        if config is not None:
            cfg_prec = config.get("precision") if isinstance(config, dict) else getattr(config, "precision", None)
        self.precision = precision or cfg_prec or os.environ.get("GIMMVFI_PRECISION", "bf16")
        self.coord_range = (-1.0, 1.0)
        if config is not None:
            cr = config.get("coord_range") if isinstance(config, dict) else getattr(config, "coord_range", None)
            if cr is not None:
                self.coord_range = (float(cr[0]), float(cr[1]))
        sd0 = gimm_state_dict(random_state_dict(0))
        for name in gimm_param_spec():
            parts = name.split(".")
            node = self
            for p in parts[:-1]:
                if p not in node._modules:
                    node.add_module(p, _Node())
                node = node._modules[p]
            node.register_parameter(parts[-1], nn.Parameter(sd0[name].clone(), requires_grad=False))
        self._engine = None
        self._engine_key = None

    def load_state_dict(self, state_dict, strict=True, **kw):
        r = super().load_state_dict(state_dict, strict=strict, **kw)
        self._engine = None
        return r

    def _apply(self, fn, *a, **kw):
        r = super()._apply(fn, *a, **kw)
        self._engine = None
        return r

    def engine(self, device=None, runtime=None):
        device = torch.device(device) if device is not None else next(self.parameters()).device
        key = (str(device), self.precision, id(runtime))
        if self._engine is None or self._engine_key != key:
            if runtime is None:
                if device.type != "cuda":
                    raise RuntimeError(
                        "GIMM runs only on an MI355X through libgimmvfi_hip.so: move the model and inputs to 'cuda' "
                        "(there is no CPU fallback)"
                    )
                runtime = Runtime(L.get(), self.precision, device)
            self._engine = Engine(runtime, self.state_dict(), motion_only=True)
            self._engine_key = key
        return self._engine

    def forward(self, xs, coord=None, keep_xs_shape=True, ori_flow=None, timesteps=None):
        # gimm.py:129-214
        assert keep_xs_shape, "keep_xs_shape=False is not used by the reference's drivers"
        assert coord is not None and ori_flow is not None and timesteps is not None
        return self.engine(xs.device).forward_motion(xs, coord, ori_flow, timesteps)

    sample_coord_input = GIMMVFI_R.sample_coord_input

    def compute_loss(self, preds, targets, reduction="mean", single=False):
        # gimm.py:216-238
        assert reduction in ["mean", "sum", "none"]
        assert preds.shape[2] == 1 and targets.shape[2] == 1
        b = preds.shape[0]
        mse = torch.reshape((preds[:, :, 0] - targets[:, :, 0]) ** 2, (b, -1)).mean(dim=-1)
        if reduction == "mean":
            total, psnr = mse.mean(), (-10 * torch.log10(mse)).mean()
        elif reduction == "sum":
            total, psnr = mse.sum(), (-10 * torch.log10(mse)).sum()
        else:
            total, psnr = mse, -10 * torch.log10(mse)
        return {"loss_total": total, "mse": total, "psnr": psnr}
