"""Host-side logic on the CPU: pair sharding + result gather over gloo (world_size 2), the CLI's
config / padding / flow-colour helpers, and the synthetic data generator."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gimmvfi_hip import shard
from util import ROOT

SRC = os.path.join(ROOT, "gimm-vfi_amd", "src")


def test_pair_range_partitions_contiguously():
    for m in (0, 1, 7, 8, 9, 31):
        for w in (1, 2, 3, 8):
            rs = [shard.pair_range(m, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == m
            assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, num_pairs, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a, b = shard.pair_range(num_pairs, rank, world)
    # frame j of this rank's pairs is filled with the global pair index
    local = torch.stack([torch.full((3, 4, 5, 3), j, dtype=torch.uint8) for j in range(a, b)]) if b > a else \
        torch.zeros((0, 3, 4, 5, 3), dtype=torch.uint8)
    out = shard.gather_frames(local, num_pairs, rank, world)
    # streaming schedule of the CLI: rounds of world x 2 pairs, two result tensors per forward (frames + flow pictures),
    # ragged / empty last sub-blocks; rank 0 re-assembles the video in order from the per-round pieces
    ppf = 2
    rg = shard.RoundGather(rank, world)
    pieces = []
    for rnd in shard.round_schedule(num_pairs, ppf, world):
        j0, b = rnd[rank]
        fr = torch.stack([torch.full((3, 4, 5, 3), j, dtype=torch.uint8) for j in range(j0, j0 + b)]) if b else \
            torch.zeros((0, 3, 4, 5, 3), dtype=torch.uint8)
        pics = (fr[:, :, :2, :2].reshape(-1, 2, 2, 3) + 100) if b else torch.zeros((0, 2, 2, 3), dtype=torch.uint8)
        got = rg.gather([fr, pics], [(ppf, 3, 4, 5, 3), (ppf * 3, 2, 2, 3)],
                        [[c for _, c in rnd], [3 * c for _, c in rnd]])
        if rank == 0:
            for r in range(world):
                assert got[0][r].shape[0] == rnd[r][1] and got[1][r].shape[0] == 3 * rnd[r][1]
                assert bool((got[1][r].reshape(-1) == (got[0][r][:, :, :2, :2].reshape(-1) + 100)).all())
                pieces.append(got[0][r].clone())
        else:
            assert got is None
    if rank == 0:
        assert torch.equal(out, torch.cat(pieces, 0))
        q.put(out.numpy())
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("num_pairs", [7, 5, 2, 1])
def test_gather_frames_gloo_world2(num_pairs):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, num_pairs, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got.shape == (num_pairs, 3, 4, 5, 3)
    for j in range(num_pairs):
        assert (got[j] == j).all()


def test_round_schedule_covers_every_pair_once_in_order():
    for m in (0, 1, 7, 8, 9, 31, 64):
        for w in (1, 2, 3, 8):
            for ppf in (1, 2, 8):
                seen = []
                for rnd in shard.round_schedule(m, ppf, w):
                    assert len(rnd) == w
                    for j0, b in rnd:
                        assert 0 <= b <= ppf
                        seen += list(range(j0, j0 + b))
                assert seen == list(range(m))           # rank-major inside a round == video order


def _cli_round_worker(rank, world, port, num_pairs, bsz, N, q):
    """One rank of the CLI's gather bookkeeping (src/video_Nx.py main loop) with stand-in results: frame f of the video holds
    the value f % 251, picture g of flow.mp4 the value (g + 100) % 251 -- what arrives where is checked on rank 0."""
    import numpy as np

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, SRC)
    import video_Nx

    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    rounds = shard.round_schedule(num_pairs, bsz, world)
    gat = shard.RoundGather(rank, world)
    H0, W0, hf, wf = 3, 4, 2, 2
    out, flow = {}, {}
    for k, rnd in enumerate(rounds):
        j0, b = rnd[rank]
        lead = 1 if (b > 0 and j0 == 0) else 0
        comp = torch.zeros((b * N + lead, H0, 2 * W0, 3), dtype=torch.uint8)
        if lead:
            comp[0] = 0
        for jj in range(b):
            for i in range(N):
                comp[lead + jj * N + i] = (1 + (j0 + jj) * N + i) % 251
        pics = torch.zeros((b * (N - 1), hf, wf, 3), dtype=torch.uint8)
        for g in range(b * (N - 1)):
            pics[g] = (j0 * (N - 1) + g + 100) % 251
        cnt = [c for _, c in rnd]
        got = gat.gather([comp, pics], [(bsz * N + 1, H0, 2 * W0, 3), (bsz * (N - 1), hf, wf, 3)],
                         [[c * N + (1 if (c > 0 and jb == 0) else 0) for jb, c in rnd], [c * (N - 1) for c in cnt]])
        if rank == 0:
            blocks = [blk for blk in rnd if blk[1] > 0]
            hosts = []
            for r, blk in enumerate(rnd):
                if blk[1] > 0:
                    hosts += [got[0][r].numpy().copy(), got[1][r].numpy().copy()]
            for kind, idx, img in video_Nx.round_frames(blocks, hosts, N, num_pairs):
                (out if kind == "out" else flow)[idx] = int(np.asarray(img).flat[0])
                assert (np.asarray(img) == np.asarray(img).flat[0]).all()
    if rank == 0:
        q.put((out, flow))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("num_pairs,bsz", [(7, 2), (4, 4), (9, 1)])
def test_cli_round_gather_and_frame_order_gloo_world2(num_pairs, bsz):
    """The multi-GPU CLI path that no hardware was available for: round schedule -> every rank's composed side-by-side
    frames (+ the video's leading frame on the rank that owns pair 0) and flow pictures -> RoundGather -> round_frames:
    every frame of output.mp4 / flow.mp4 arrives exactly once, at its position, the very last frame dropped."""
    N = 4
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cli_round_worker, args=(r, 2, port, num_pairs, bsz, N, q)) for r in range(2)]
    for p in procs:
        p.start()
    out, flow = q.get(timeout=180)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(out) == list(range(num_pairs * N))                  # 1 + pairs * N frames, the last one dropped
    assert all(out[f] == f % 251 for f in out)
    assert sorted(flow) == list(range(num_pairs * (N - 1)))
    assert all(flow[g] == (g + 100) % 251 for g in flow)


def test_bench_launches_itself_world2_stub():
    """`python bench.py --gpus 2` without a torchrun world starts its own two ranks (the driver's invocation);
    `--stub` swaps the GPU step for a CPU/gloo stand-in so the launcher, the barrier / max-over-ranks timing and the
    one-JSON-line contract are exercised here."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["scaling"] == "weak"
    assert rec["config"]["world_size_rccl"] == 2 and rec["config"]["parallelism"] == "pair-sharded x2"
    assert rec["value"] > 0 and rec["higher_is_better"] is True
    # without the stub, more GPUs than the node has is a clear error, not an assertion deep inside
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1"],
                        capture_output=True, text=True, timeout=300, env=env)
    assert r2.returncode == 2 and "GPU(s) are visible" in r2.stderr


def test_cli_helpers_config_padder_flowviz():
    sys.path.insert(0, SRC)
    try:
        from utils.flow_viz import flow_to_image, make_colorwheel
        from utils.setup import load_config
        from utils.utils import InputPadder
    finally:
        sys.path.remove(SRC)
    cfg = load_config(os.path.join(ROOT, "gimm-vfi_amd", "configs", "gimmvfi", "gimmvfi_r_arb.yaml"))
    assert cfg.arch.type == "gimmvfi_r" and cfg.arch.hyponet.activation.siren_w0 == 1.0
    assert cfg.arch.fwarp_type == "linear" and cfg.arch.raft_iter == 20 and cfg.arch.hyponet.output_bias == 0.5
    x = torch.arange(2 * 3 * 45 * 70, dtype=torch.float32).reshape(2, 3, 45, 70)
    pad = InputPadder(x.shape, 32)
    xp = pad.pad(x)
    assert xp.shape[-2:] == (64, 96)
    assert torch.equal(pad.unpad(xp), x)
    assert torch.equal(xp[..., 0, :], xp[..., 9, :])     # replicate padding, centred (9 rows on top)
    wheel = make_colorwheel()
    assert wheel.shape == (55, 3) and wheel[0].tolist() == [255, 0, 0] and wheel[15].tolist() == [255, 255, 0]
    flow = np.stack(np.meshgrid(np.linspace(-3, 3, 16), np.linspace(-2, 2, 12)), -1).astype(np.float32)
    img = flow_to_image(flow, convert_to_bgr=True)
    assert img.shape == (12, 16, 3) and img.dtype == np.uint8
    rgb = flow_to_image(flow)
    assert (img[..., ::-1] == rgb).all()


def test_synthetic_pairs_are_seeded_and_quantised():
    from gimmvfi_hip.synth import synthetic_pairs

    a = synthetic_pairs(2, 64, 96, seed=5)
    b = synthetic_pairs(2, 64, 96, seed=5)
    assert torch.equal(a, b) and a.shape == (2, 3, 2, 64, 96)
    assert float(a.min()) >= 0 and float(a.max()) <= 1
    assert torch.equal(torch.round(a * 255), a * 255 + 0 * a) or float((torch.round(a * 255) - a * 255).abs().max()) < 1e-4
    assert not torch.equal(a[0], a[1])


def test_cli_checkpoint_branches(tmp_path):
    """src/video_Nx.py:99-115: {"state_dict": ...} strict=True, and the legacy branch for paths containing "ours"
    (module.feature_bone.* -> frame_encoder.*, every other key dropped, strict=False)."""
    sys.path.insert(0, SRC)
    try:
        import video_Nx
    finally:
        sys.path.remove(SRC)

    class Legacy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.frame_encoder = torch.nn.Linear(3, 2)
            self.other = torch.nn.Linear(2, 2)

    m = Legacy()
    other0 = m.other.weight.detach().clone()
    w = torch.arange(6, dtype=torch.float32).reshape(2, 3)
    p_ours = str(tmp_path / "ours_legacy.pth")
    torch.save({"module.feature_bone.weight": w, "module.feature_bone.bias": torch.ones(2), "module.other.weight": torch.zeros(2, 2)}, p_ours)
    r = video_Nx.load_checkpoint(m, p_ours)
    assert torch.equal(m.frame_encoder.weight, w) and torch.equal(m.other.weight, other0)
    assert sorted(r.missing_keys) == ["other.bias", "other.weight"] and not r.unexpected_keys
    p_std = str(tmp_path / "gimmvfi.pth")
    sd = {k: v + 1 for k, v in Legacy().state_dict().items()}
    torch.save({"state_dict": sd}, p_std)
    video_Nx.load_checkpoint(m, p_std)
    assert torch.equal(m.other.weight, sd["other.weight"])
    bad = dict(sd)
    bad.pop("other.bias")
    torch.save({"state_dict": bad}, p_std)
    with pytest.raises(RuntimeError):
        video_Nx.load_checkpoint(m, p_std)


def _cli_main_worker(rank, world, port, src, out, N, bsz, fail, q):
    """One rank of the REAL src/video_Nx.py main() with every rank a CPU stand-in (GVFI_CLI_DRY=2: no model, results =
    [orig | orig] frames of the real shape): schedule, decode, per-rank PNG encoding, BytesGather, abort flag, sinks."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      GVFI_CLI_DRY="2", GVFI_CLI_TEST_FAIL=fail)
    sys.path.insert(0, SRC)
    import video_Nx

    try:
        video_Nx.main(["--source-path", src, "--output-path", out, "--N", str(N), "--ds-factor", "1.0", "--batch", str(bsz),
                       "-m", os.path.join(ROOT, "gimm-vfi_amd", "configs", "gimmvfi", "gimmvfi_r_arb.yaml"), "--random-init", "--eval"])
        q.put((rank, "ok"))
    except BaseException as e:     # noqa: BLE001
        q.put((rank, f"{type(e).__name__}: {e}"))
        raise


def _write_frames(d, n, H=40, W=56):
    from PIL import Image

    rng = np.random.default_rng(3)
    frames = []
    for i in range(n):
        f = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        f[:, :, 0] = i                      # frame index in the red channel of every pixel
        Image.fromarray(f).save(os.path.join(d, f"{i:04d}.png"))
        frames.append(f)
    return frames


def _run_cli_world(tmp_path, world, n_frames, N, bsz, fail=""):
    src, out = str(tmp_path / "in"), str(tmp_path / "out")
    os.makedirs(src)
    frames = _write_frames(src, n_frames)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cli_main_worker, args=(r, world, port, src, out, N, bsz, fail, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
    return frames, out, res, [p.exitcode for p in procs]


@pytest.mark.parametrize("world,n_frames,bsz", [(2, 8, 2), (2, 6, 1), (2, 3, 2), (1, 5, 2)])
def test_cli_main_encoded_gather_gloo_world2(tmp_path, world, n_frames, bsz):
    """Round 5: the multi-GPU CLI with per-rank PNG encoding + the gather of the compressed bytes (shard.BytesGather), the real
    main() on two gloo ranks: every frame of output / flow arrives exactly once at its position, byte-exact (PNG is lossless:
    decoded == what the owning rank composed), the video's very last frame dropped (reference video_Nx.py:225-246)."""
    from PIL import Image

    N = 3
    frames, out, res, codes = _run_cli_world(tmp_path, world, n_frames, N, bsz)      # (world 1: the same encode path, no collective)
    assert codes == [0] * world and all(v == "ok" for v in res.values()), (codes, res)
    num_pairs = n_frames - 1
    od, fd = os.path.join(out, "output_frames"), os.path.join(out, "flow_frames")
    assert sorted(os.listdir(od)) == [f"{i:04d}.png" for i in range(num_pairs * N)]
    assert sorted(os.listdir(fd)) == [f"{i:04d}.png" for i in range(num_pairs * (N - 1))]
    W = frames[0].shape[1]
    for idx in range(num_pairs * N):
        img = np.array(Image.open(os.path.join(od, f"{idx:04d}.png")))           # RGB
        if idx == 0:
            j, i = 0, None
        else:
            j, i = (idx - 1) // N, (idx - 1) % N
        assert (img[:, :W] == frames[j]).all(), idx                              # left half: orig_j
        right = frames[j + 1] if i == N - 1 else frames[j]
        assert (img[1:, W:] == right[1:]).all(), idx
        if i is not None:
            assert tuple(img[0, W]) == (i, i, i), (idx, img[0, W])               # the slot marker the owning rank wrote
    for g in range(num_pairs * (N - 1)):
        img = np.array(Image.open(os.path.join(fd, f"{g:04d}.png")))
        j, i = g // (N - 1), g % (N - 1)
        assert (img[1:] == frames[j][1:]).all() and tuple(img[0, 0]) == (i, i, i), g


@pytest.mark.parametrize("fail", ["1:1", "0:2", "sink:1"])
def test_cli_main_abort_reaches_every_rank_gloo_world2(tmp_path, fail):
    """A failure on ONE rank (a forward on rank 1, on rank 0, rank 0's sink) ends BOTH ranks within a round -- nobody is left
    blocked in a collective until the backend's timeout (VERDICT r4 missing #2 / ADVICE r3)."""
    import time

    t0 = time.perf_counter()
    _, _, res, codes = _run_cli_world(tmp_path, 2, 12, 3, 1, fail=fail)
    assert time.perf_counter() - t0 < 120
    assert all(c not in (0, None) for c in codes), (codes, res)
    assert all(v != "ok" for v in res.values()), res
    assert any("injected" in v for v in res.values()) and any(("aborted" in v) or ("injected" in v) for v in res.values())


def test_steps_in_flight_refuses_a_model_that_is_not_on_the_gpu():
    """StepsInFlight is a device-side pipeline: like the model itself it has no CPU path and says so."""
    import pytest
    from gimmvfi_hip.model import GIMMVFI_R, StepsInFlight

    m = GIMMVFI_R(precision="bf16")
    with pytest.raises(RuntimeError, match="MI355X"):
        StepsInFlight(m, depth=2)
    m.serial_launch = True
    r = m.replica()
    assert type(r) is GIMMVFI_R and r is not m and r.precision == m.precision and r.serial_launch
    assert all((a == b).all() for a, b in zip(m.state_dict().values(), r.state_dict().values()))


def test_compact_bench_line_stays_below_2_kb_also_for_eight_ranks():
    """bench.py prints the compact form of its record: the driver keeps a bounded tail of stdout, so the line must stay below 2 KB --
    also with the per-rank step times of an 8-GPU run and every in-run probe in it.  Checked on the committed record of the final
    run (profiles/r6_bench_all_final_box2.json) blown up to eight ranks."""
    import importlib.util
    import json

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    full = json.load(open(os.path.join(root, "profiles", "r6_bench_all_final_box2.json")))
    one = json.dumps(bench.compact_line(full, "gpurun_out/bench_full.json"))
    assert len(one) < 2048, len(one)
    full["n_gpus"] = 8
    full["config"]["parallelism"] = "pair-sharded x8"
    full["config"]["world_size_rccl"] = 8
    full["config"]["ms_per_step_per_rank"] = [20.123 + 0.011 * r for r in range(8)]
    full["config"]["gather_bytes_per_rank_per_step"] = 2752512
    full["configs"] = full["configs"][:2]                 # (N > 1 times the two configurations BASELINE defines on 8 GPUs)
    eight = json.loads(json.dumps(bench.compact_line(full, "gpurun_out/bench_full.json")))
    assert len(json.dumps(eight)) < 2048, len(json.dumps(eight))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in eight, k
    assert eight["config"]["steps_in_flight"] == 2 and len(eight["config"]["ms_per_step_per_rank"]) == 8
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(eight["roofline"])
