"""Nx video frame interpolation CLI -- drop-in for reference src/video_Nx.py (same flags, same outputs):

    python src/video_Nx.py --source-path DIR --output-path DIR --ds-factor F --N K -m CFG.yaml -l CKPT --eval

writes OUT/output.mp4 (side-by-side [original | interpolated], fps 2N) and OUT/flow.mp4 (colour-coded
flow_t).  Differences: runs on the MI355X HIP kernels (no CuPy / CUDA), and under
``python -m torch.distributed.run --nproc-per-node G`` the frame pairs are sharded over G GPUs in ROUNDS of G x `--batch`
consecutive pairs (gimmvfi_hip/shard.py:round_schedule); after every round the device-resident uint8 result frames and
flow pictures are gathered to rank 0 over RCCL (the path's only collective) and written to the videos in order while the
next round computes -- memory on every rank is O(one round), and a rank decodes only its own frames.  Every rank runs `--batch`
consecutive pairs per forward and encodes each frame once (model.forward_sequence); the [original | interpolated] frames are
composed on the device from the resident input frames (gvfi_compose_sbs_u8); frame decode / upload, result download and
video writing run in a host pipeline beside the GPU (gimmvfi_hip/io_pipeline.py).  Without OpenCV the frames are written as PNGs
(OUT/output_frames, OUT/flow_frames) and encoded with ffmpeg when it is on PATH.
``--random-init`` (addition) runs with seeded random weights when no checkpoint is available."""
import argparse
import os
import shutil
import subprocess
import sys
import time

import numpy as np
import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from models import create_model  # noqa: E402
from utils.flow_viz import make_colorwheel  # noqa: E402
from utils.setup import single_setup  # noqa: E402
from utils.utils import InputPadder, set_seed  # noqa: E402

from gimmvfi_hip import shard  # noqa: E402  (models/__init__ put the package root on sys.path)
from gimmvfi_hip.io_pipeline import FramePrefetcher, ResultDrain, VideoSink  # noqa: E402

try:
    import cv2
except Exception:  # OpenCV is optional
    cv2 = None
try:
    from tqdm import tqdm
except Exception:
    def tqdm(x):
        return x


def default_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("-m", "--model-config", type=str, default="./configs/gimmvfi/gimmvfi_r_arb.yaml")
    parser.add_argument("--source-path", type=str, default="")
    parser.add_argument("--output-path", type=str, default="")
    parser.add_argument("--N", type=int, default=8)
    parser.add_argument("--ds-factor", type=float, default=1.0)
    parser.add_argument("-r", "--result-path", type=str, default="./results.tmp")
    parser.add_argument("-l", "--load-path", type=str, default="")
    parser.add_argument("-p", "--postfix", type=str, default="")
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--eval", action="store_true")
    parser.add_argument("--random-init", action="store_true", help="seeded random weights instead of a checkpoint")
    parser.add_argument("--precision", type=str, default=None, choices=[None, "bf16", "fp32"])
    parser.add_argument("--batch", type=int, default=0, help="consecutive frame pairs per forward (0 = by frame size)")
    return parser


def parse_args(argv=None):
    return default_parser().parse_known_args(argv)


def load_image(img_path):
    # reference video_Nx.py:46-50
    raw = np.array(Image.open(img_path).convert("RGB"))
    return (torch.from_numpy(raw.copy()).permute(2, 0, 1) / 255.0).to(torch.float).unsqueeze(0)


def images_to_video(imgs, output_video_path, fps=15):
    # reference video_Nx.py:53-84 (cv2.VideoWriter, ffmpeg for > 2048); PNG fallback without OpenCV.  Kept for callers of the
    # reference's module surface; main() streams through gimmvfi_hip.io_pipeline.VideoSink instead of collecting a list
    height, width, _ = imgs[0].shape
    big = max(height, width // 2) > 2048
    if cv2 is not None and not big:
        video = cv2.VideoWriter(output_video_path, cv2.VideoWriter_fourcc(*"mp4v"), fps, (width, height))
        for img in imgs:
            video.write(img)
        video.release()
        return output_video_path
    frame_dir = os.path.splitext(output_video_path)[0] + "_frames"
    os.makedirs(frame_dir, exist_ok=True)
    from concurrent.futures import ThreadPoolExecutor

    def save(item):     # (zlib releases the GIL: the frames compress in parallel; level 1 is ~4x faster than PIL's default 6)
        idx, img = item
        Image.fromarray(np.ascontiguousarray(img[:, :, ::-1])).save(os.path.join(frame_dir, f"{idx:04d}.png"), compress_level=1)

    with ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(save, enumerate(imgs)))
    if shutil.which("ffmpeg"):
        subprocess.run(["ffmpeg", "-y", "-framerate", f"{fps}", "-i", f"{frame_dir}/%04d.png", "-c:v", "libx264",
                        "-pix_fmt", "yuv420p", output_video_path])
        shutil.rmtree(frame_dir, ignore_errors=True)
        return output_video_path
    return frame_dir


def round_frames(blocks, hosts, N, num_pairs):
    """Video positions of what one round delivered.  blocks: [(first pair j0, pairs b)] in rank order; hosts: per block the
    composed side-by-side frames [b*N (+1 when j0 == 0), H, 2W, 3] and the flow pictures [b*(N-1), h, w, 3] (numpy).  Yields
    ("out", index in output.mp4, frame) and ("flow", index in flow.mp4, picture) in the reference's order (video_Nx.py:225-246):
    [orig0|orig0], then per pair N-1 x [orig_j | interp] and [orig_j+1 | orig_j+1]; the video's very last frame is dropped."""
    for k, (j0, b) in enumerate(blocks):
        comp, pics = hosts[2 * k], hosts[2 * k + 1]
        lead = 1 if j0 == 0 else 0
        assert comp.shape[0] == b * N + lead and pics.shape[0] == b * (N - 1), (comp.shape, pics.shape, j0, b)
        if lead:
            yield "out", 0, comp[0]
        for jj in range(b):
            j = j0 + jj
            for i in range(N):
                if i == N - 1 and j + 1 >= num_pairs:
                    continue
                yield "out", 1 + j * N + i, comp[lead + jj * N + i]
            for i in range(N - 1):
                yield "flow", j * (N - 1) + i, pics[jj * (N - 1) + i]


def load_checkpoint(model, load_path):
    """Reference src/video_Nx.py:99-115.  A GIMM-VFI checkpoint is {"state_dict": ...} loaded strict=True; a path containing
    "ours" is the reference's legacy branch: a flat dict whose `module.feature_bone.*` keys are renamed to `frame_encoder.*`
    (every other key dropped) and loaded strict=False -- on the GIMM-VFI models, which have no `frame_encoder`, that loads
    nothing, exactly as in the reference; the (missing, unexpected) key lists are returned for the caller / tests."""
    ckpt = torch.load(load_path, map_location="cpu")
    if "ours" in load_path:
        ckpt = {k.replace("module.feature_bone", "frame_encoder"): v for k, v in ckpt.items() if "feature_bone" in k}
        return model.load_state_dict(ckpt, strict=False)
    return model.load_state_dict(ckpt["state_dict"], strict=True)


def encode_block(block, comp, pics, N, num_pairs, flow_hw, pool):
    """One rank's results of one round as encoded video frames: [(kind, index in that video, PNG bytes)].  comp / pics: numpy
    uint8 [b*N (+1), H, 2W, 3] / [b*(N-1), h, w, 3] BGR (what round_frames takes); flow_hw: size of a flow.mp4 frame (the
    pictures are resized to it when the flow lives at a down-scaled working resolution, reference video_Nx.py:199-207)."""
    from gimmvfi_hip.io_pipeline import png_bytes_rgb

    def one(kind, idx, img):
        if kind == "flow" and tuple(img.shape[:2]) != tuple(flow_hw):
            img = np.array(Image.fromarray(img).resize((flow_hw[1], flow_hw[0]), Image.BILINEAR))
        return kind, idx, png_bytes_rgb(img[:, :, ::-1])

    futs = [pool.submit(one, kind, idx, img) for kind, idx, img in round_frames([block], [comp, pics], N, num_pairs)]
    return [f.result() for f in futs]


def main(argv=None):
    args, extra_args = parse_args(argv)
    set_seed(args.seed)
    config = single_setup(args, extra_args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # GVFI_CLI_DRY=1 (world > 1): rehearsal of the multi-GPU result path on a ONE-GPU box -- rank 0 is the real thing, ranks
    # 1.. are CPU stand-ins that decode their frames, compose [orig | orig] stand-in results of the real shape, ENCODE them on
    # their own host threads and take part in every collective (gloo, host tensors).  Never a throughput measurement of the
    # model; it exercises the schedule, the encoded gather, the abort path and rank 0's host load with real byte counts.
    # GVFI_CLI_DRY=2: EVERY rank is a stand-in (no GPU at all): the whole host side of the CLI -- schedule, decode, encode, gather,
    # abort, sinks -- runs on CPU / gloo (tests/test_host_logic.py).
    dry_level = int(os.environ.get("GVFI_CLI_DRY", "0") or 0)
    dry = dry_level == 2 or (dry_level == 1 and world > 1)
    stub = dry and (rank > 0 or dry_level == 2)
    # test hook: "R:K" raises on rank R in round K; "sink:K" kills rank 0's output sink in round K (the abort path's tests)
    fail_at = os.environ.get("GVFI_CLI_TEST_FAIL", "")
    device = torch.device("cpu") if stub else torch.device("cuda", 0 if dry else local)
    if not stub:
        torch.cuda.set_device(device)
    xdev = torch.device("cpu") if dry else device          # where the exchanged tensors live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dry:
            torch.distributed.init_process_group(backend="gloo")
        else:
            torch.distributed.init_process_group(backend="nccl", device_id=device)

    os.makedirs(args.output_path, exist_ok=True)
    if args.precision is not None:
        config.arch["precision"] = args.precision
    model = None
    if not stub:
        model, _ = create_model(config.arch)
        if args.load_path != "":
            load_checkpoint(model, args.load_path)
        elif args.random_init:
            from gimmvfi_hip.params import random_state_dict_for

            mtype = config.arch["type"] if isinstance(config.arch, dict) else config.arch.type
            model.load_state_dict(random_state_dict_for(mtype, args.seed), strict=True)
        elif args.eval:
            raise ValueError("--load-path must be specified in evaluation or resume mode")
        model = model.to(device).eval()
        # every output of a forward is turned into uint8 frames / flow pictures on the launch stream before the next forward is
        # enqueued (frames_to_u8, compose_sbs, flow_to_image below), so the graph's own output tensors are used without clones
        model.static_outputs = True

    img_list = sorted(os.listdir(args.source_path))
    num_pairs = len(img_list) - 1
    N = args.N
    ds_factor = args.ds_factor
    # host pipeline: frames are decoded once, ahead of the GPU, into pinned memory and uploaded on a side stream; results
    # come back asynchronously, are composed / colour-coded in worker threads and written WHILE the GPU runs (VideoSink)
    first = load_image(os.path.join(args.source_path, img_list[0]))
    padder = InputPadder(first.shape, 32)
    H0, W0 = first.shape[-2:]
    paths = [os.path.join(args.source_path, f) for f in img_list]
    # pairs per forward: consecutive pairs run as ONE batch whose per-frame encoder work is shared
    # (model.forward_sequence); the default keeps ~1 Mpixel of frames per forward
    bsz = args.batch if args.batch > 0 else max(1, min(8, (1 << 20) // max(1, H0 * W0)))
    # streaming schedule (gimmvfi_hip/shard.py): round k = the next world x bsz pairs of the video, rank r its r-th block;
    # the results of a round are gathered to rank 0 (the path's only collective) and written in order -- memory on
    # every rank is O(one round), whatever the length of the video
    rounds = shard.round_schedule(num_pairs, bsz, world)
    my_blocks = [rnd[rank] for rnd in rounds]
    # (only this rank's frames are decoded: block (j0, b) reads frames j0 .. j0 + b)
    my_frames = [j for j0, b in my_blocks if b > 0 for j in range(j0, j0 + b + 1)]
    ncpu = os.cpu_count() or 8
    # (a 2K PNG takes ~0.1 s to decode and the model consumes ~12 frames / s / GPU at 2K 8x: 8 decode threads, 8 frames ahead)
    frames_in = FramePrefetcher(paths, device, pad_fn=padder.pad, decode=load_image, lookahead=bsz + 7, workers=8, order=my_frames)
    drain = ResultDrain(device, depth=12, workers=8)      # composing + resizing the frames of a block: ~0.4 s at 2K
    rt = None if stub else model.engine(device).rt
    wheel = None if stub else torch.from_numpy(make_colorwheel()).float().to(device)
    # size of a flow picture: flow_t lives at the working resolution and is cropped by the FULL-resolution pad amounts
    # (padder.unpad on the down-scaled field -- the reference's behaviour, video_Nx.py:199-207)
    pw_, ph_ = padder._pad[0] + padder._pad[1], padder._pad[2] + padder._pad[3]
    hf, wf = (H0, W0) if ds_factor == 1.0 else (int((H0 + ph_) * ds_factor) - ph_, int((W0 + pw_) * ds_factor) - pw_)
    # Result path of the multi-GPU run.  PNG sinks (no OpenCV, or frames beyond the mp4v writer): every rank ENCODES its own
    # frames on its own host threads and the collective moves the compressed bytes (shard.BytesGather) -- rank 0 only writes
    # files.  OpenCV sinks need the raw frames in order on rank 0: the raw uint8 gather (shard.RoundGather) stays.
    # (one GPU takes the same encode path -- its sink then only writes files too, and the pinned D2H buffers are recycled)
    encoded = not (VideoSink.uses_cv2((H0, 2 * W0)) or VideoSink.uses_cv2((H0, W0)))
    gatherer = shard.RoundGather(rank, world) if (world > 1 and not encoded) else None
    bgather = shard.BytesGather(rank, world, xdev) if (encoded and world > 1) else None
    LAG = 2           # rounds between a forward and the exchange of its encoded frames: the encode threads' head start
    from concurrent.futures import Future, ThreadPoolExecutor
    enc_pool = ThreadPoolExecutor(max_workers=max(2, min(32, ncpu // max(1, world)))) if encoded else None
    enc_futs = []

    t_start = time.perf_counter()
    sinks = None
    if rank == 0 and num_pairs > 0:
        sinks = (VideoSink(os.path.join(args.output_path, "output.mp4"), N * 2, num_pairs * N, (H0, 2 * W0)),
                 VideoSink(os.path.join(args.output_path, "flow.mp4"), N * 2, num_pairs * (N - 1), (H0, W0)))

    prof = {"decode_wait": 0.0, "forward": 0.0, "enqueue": 0.0, "submit": 0.0, "post": 0.0, "exchange": 0.0} if os.environ.get("GVFI_CLI_TIMING") else None

    def post(blocks):
        """blocks: [(first pair, pairs)] of the tensors handed to the drain, in order (one per contributing rank)."""
        def fn(*hosts):
            tq = time.perf_counter()
            # hosts: per block (comp_u8 [b*N (+1), H, 2W, 3] BGR side-by-side frames, composed on the GPU from the resident input
            # frames by gvfi_compose_sbs_u8; pics_u8 [b*(N-1), h, w, 3] BGR flow pictures, colour-coded on the GPU by
            # gvfi_flow_to_image).  Output frame order of the reference (video_Nx.py:225-246): [orig0|orig0], then per pair its
            # N-1 [orig_j | interp] frames and [orig_j+1 | orig_j+1]; the very last frame is dropped
            out_sink, flow_sink = sinks
            for kind, idx, img in round_frames(blocks, [h_.numpy() for h_ in hosts], N, num_pairs):
                if kind == "out":
                    out_sink.put(idx, img)
                else:
                    if ds_factor != 1.0:   # flow_t lives at the working resolution; resize the picture for the video
                        img = np.array(Image.fromarray(img).resize((W0, H0), Image.BILINEAR))
                    flow_sink.put(idx, img)
            if prof is not None:
                prof["post"] += time.perf_counter() - tq
            return None
        return fn

    def post_encode(block, fut):
        """(encoded path) this rank's own block: compose -> PNG bytes on this rank's threads; the result goes to `fut`."""
        def fn(comp_h, pics_h):
            tq = time.perf_counter()
            try:
                fut.set_result(encode_block(block, comp_h.numpy(), pics_h.numpy(), N, num_pairs, (H0, W0), enc_pool))
            except BaseException as e:      # noqa: BLE001 -- reaches the main loop through the future (-> abort flag)
                fut.set_exception(e)
            if prof is not None:
                prof["post"] += time.perf_counter() - tq
            return None
        return fn

    def stub_results(frames, j0, b):
        """(dry rehearsal, CPU ranks) results of the real SHAPE and a realistic byte count: every [orig | interp] slot shows
        [orig | orig], the flow pictures a crop of the frame."""
        fr = (padder.unpad(frames).clamp(0, 1) * 255).to(torch.uint8).permute(0, 2, 3, 1).flip(-1).contiguous()      # BGR
        lead = 1 if j0 == 0 else 0
        comp = torch.empty((b * N + lead, H0, 2 * W0, 3), dtype=torch.uint8)
        if lead:
            comp[0] = torch.cat([fr[0], fr[0]], 1)
        for jj in range(b):
            for i in range(N):
                comp[lead + jj * N + i] = torch.cat([fr[jj], fr[jj + 1] if i == N - 1 else fr[jj]], 1)
                comp[lead + jj * N + i, 0, W0] = i          # (first pixel of the right half: which of the pair's N slots this is)
        pics = fr[:b, :hf, :wf].repeat_interleave(N - 1, 0).contiguous()
        for g in range(b * (N - 1)):
            pics[g, 0, 0] = g % (N - 1)
        return comp, pics

    def deliver(kk, failed):
        """(encoded path) exchange of round kk's encoded frames; returns True when the run is to stop (some rank failed)."""
        payload, entries = None, None
        if not failed:
            try:
                entries = enc_futs[kk].result(timeout=600)
                if world > 1:
                    payload = shard.pack_entries(entries)
            except BaseException as e:      # noqa: BLE001
                failed = e
        if rank == 0 and not failed and sinks is not None:
            failed = sinks[0].err or sinks[1].err or False
        if world == 1:          # one rank: its own encoded frames go straight to the sinks
            if failed:
                return failed
            for kind, idx, data in entries:
                sinks[0 if kind == "out" else 1].put_png(idx, data)
            return False
        got, stop = bgather.exchange(payload, abort=bool(failed))
        if stop:
            return failed or True
        if rank == 0:
            for buf in got:
                for kind, idx, data in shard.unpack_entries(buf):
                    sinks[0 if kind == "out" else 1].put_png(idx, data)
        return False

    coord_cache = {}
    copied = [None, None]      # D2H-complete events of the gather staging buffers (two parities)
    t_warm, pairs_warm = None, 0
    failure, stopped = None, False
    for k, (j0, b) in enumerate(tqdm(my_blocks)):
        if k == 1 and not stub:    # the first round pays model packing + graph capture: steady state starts here
            torch.cuda.synchronize(device)
            t_warm, pairs_warm = time.perf_counter(), sum(c for _, c in rounds[0])
        try:
            if fail_at == f"{rank}:{k}":
                raise RuntimeError(f"injected failure on rank {rank} in round {k} (GVFI_CLI_TEST_FAIL)")
            if fail_at == f"sink:{k}" and sinks is not None:
                sinks[0].err = RuntimeError(f"injected sink failure in round {k} (GVFI_CLI_TEST_FAIL)")
            comp_u8 = torch.zeros((0, H0, 2 * W0, 3), dtype=torch.uint8, device=device)
            pics_u8 = torch.zeros((0, hf, wf, 3), dtype=torch.uint8, device=device)
            if b > 0:
                tp = time.perf_counter()
                frames = torch.cat([frames_in.get(j) for j in range(j0, j0 + b + 1)], 0)      # (b+1, 3, Hp, Wp), each decoded once
                if prof is not None:
                    prof["decode_wait"] += time.perf_counter() - tp
                    tp = time.perf_counter()
                s_shape = frames.shape[-2:]
                if stub:
                    comp_u8, pics_u8 = stub_results(frames, j0, b)
                else:
                    with torch.no_grad():
                        key = (b, tuple(s_shape))
                        if key not in coord_cache:     # the coordinate grids only depend on the batch and frame size
                            coord_cache[key] = (
                                [(model.sample_coord_input(b, s_shape, [1 / N * i], device=device, upsample_ratio=ds_factor), None)
                                 for i in range(1, N)],
                                [i * 1 / N * torch.ones(b, device=device, dtype=torch.float) for i in range(1, N)])
                        coord_inputs, timesteps = coord_cache[key]
                        out = model.forward_sequence(frames, coord_inputs, t=timesteps, ds_factor=None if ds_factor == 1.0 else ds_factor)
                        if prof is not None:        # (host time of the graph launch: input copies + hipGraphLaunch)
                            prof["forward"] += time.perf_counter() - tp
                            tp = time.perf_counter()
                        preds = torch.stack([padder.unpad(out["imgt_pred"][i]) for i in range(N - 1)], 1)       # [b, N-1, 3, H, W]
                        pred_u8 = rt.frames_to_u8(preds.reshape(-1, *preds.shape[2:]).contiguous()).reshape(b, N - 1, H0, W0, 3)
                        # [orig | interp] video frames from the frames already resident (reference: a second decode + cv2.hconcat per frame)
                        comp_u8 = rt.compose_sbs(frames, padder._pad[2], padder._pad[0], pred_u8, N, lead=(j0 == 0))
                        flows = []
                        for i in range(N - 1):
                            u = padder.unpad(out["flowt"][i])
                            flows.append(u.reshape(b, 2, *u.shape[-2:]))
                        flows = torch.stack(flows, 1).contiguous()                                              # [b, N-1, 2, h, w]
                        pics_u8 = rt.flow_to_image(flows.reshape(-1, 2, *flows.shape[-2:]), wheel, bgr=True)     # reference :199-207
                if prof is not None:
                    prof["enqueue"] += time.perf_counter() - tp
        except BaseException as e:      # noqa: BLE001 -- a failure on ONE rank must end EVERY rank: raised below, once the flag went round
            failure = e
        # ---- result path.  The abort flag travels with the round's collective (encoded: in the size exchange; raw: one all_gather
        # in front of the gather), so a dead sink on rank 0 or a failed forward anywhere stops all ranks at this round instead of
        # leaving them blocked in a gather until the RCCL timeout.
        tp = time.perf_counter()
        if encoded:
            fut = Future()
            enc_futs.append(fut)
            try:
                if failure is not None:
                    raise failure
                if b > 0:
                    drain.submit(k, [comp_u8, pics_u8], post_encode((j0, b), fut), recycle=True)
                else:
                    fut.set_result([])
            except BaseException as e:      # noqa: BLE001
                failure = e
                if not fut.done():
                    fut.set_exception(e)
            if k >= LAG or failure is not None:
                stop = deliver(max(0, k - LAG), failure or False)
                if stop:
                    failure = stop if isinstance(stop, BaseException) else (failure or RuntimeError("video_Nx: another rank aborted the run"))
                    stopped = True
                    break
        elif world > 1:
            if rank == 0 and failure is None and sinks is not None:
                failure = sinks[0].err or sinks[1].err or None
            if shard.any_abort(failure is not None, xdev, world):
                failure = failure or RuntimeError("video_Nx: another rank aborted the run")
                stopped = True
                break
            try:
                # one gather per round and result kind (uint8, device resident); only rank 0 copies anything to the host
                if dry:
                    comp_u8, pics_u8 = comp_u8.cpu(), pics_u8.cpu()      # (rehearsal: the collective runs over gloo)
                par = gatherer.parity
                if copied[par] is not None:
                    torch.cuda.current_stream(device).wait_event(copied[par])    # staging buffers of round k-2 fully drained
                cnt = [c for _, c in rounds[k]]
                got = gatherer.gather([comp_u8, pics_u8], [(bsz * N + 1, H0, 2 * W0, 3), (bsz * (N - 1), hf, wf, 3)],
                                      [[c * N + (1 if (c > 0 and jb == 0) else 0) for jb, c in rounds[k]], [c * (N - 1) for c in cnt]])
                if rank == 0:
                    blocks = [blk for blk in rounds[k] if blk[1] > 0]
                    tens = []
                    for r, blk in enumerate(rounds[k]):
                        if blk[1] > 0:
                            tens += [got[0][r], got[1][r]]
                    copied[par] = drain.submit(k, tens, post(blocks))
            except BaseException as e:      # noqa: BLE001 -- (a failed writer surfaces in submit) flagged at the next round's exchange
                failure = e
        else:
            if failure is not None:
                break
            if b > 0:
                drain.submit(k, [comp_u8, pics_u8], post([(j0, b)]))
        if prof is not None:
            prof["exchange"] += time.perf_counter() - tp
    if world > 1 and not encoded and not stopped:
        # (raw path) a failure of the LAST round's gather / submit still has to reach the other ranks
        if shard.any_abort(failure is not None, xdev, world):
            failure = failure or RuntimeError("video_Nx: another rank aborted the run")
    if encoded and not stopped:
        for kk in range(max(0, len(my_blocks) - LAG), len(my_blocks)):        # the last LAG rounds are still to be exchanged
            stop = deliver(kk, False)
            if stop:
                failure = stop if isinstance(stop, BaseException) else RuntimeError("video_Nx: another rank aborted the run")
                break
    if failure is not None:
        frames_in.close()
        if enc_pool is not None:
            enc_pool.shutdown(wait=False, cancel_futures=True)
        if world > 1:
            torch.distributed.destroy_process_group()
        raise failure
    tp = time.perf_counter()
    drain.finish()
    frames_in.close()
    if enc_pool is not None:
        enc_pool.shutdown(wait=True)
    if prof is not None:
        print(f"[video_Nx] rank {rank} host seconds of the main loop ({len(my_blocks)} rounds): " + ", ".join(f"{k_}: {v:.3f}" for k_, v in prof.items())
              + f", final drain {time.perf_counter() - tp:.3f}"
              + (f"; encoded gather: {bgather.bytes_moved / 1e6:.1f} MB received" if (bgather is not None and rank == 0) else ""))
    if t_warm is not None and rank == 0:
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t_warm
        nfr = (num_pairs - pairs_warm) * (N - 1)
        print(f"[video_Nx] steady state: {nfr} interpolated frames ({W0}x{H0}, {N}x, {bsz} pairs/forward x {world} GPU(s)) in {dt:.3f} s = "
              f"{nfr / dt:.1f} frames/s incl. PNG decode, H2D, gather, D2H and flow colour-coding (video encoding runs beside it)"
              + (" -- DRY rehearsal: ranks 1.. are CPU stand-ins, not a throughput figure" if dry else ""))
    if rank == 0 and sinks is not None:
        o1, o2 = sinks[0].close(), sinks[1].close()
        wall = time.perf_counter() - t_start
        print(f"[video_Nx] wall clock incl. video writing: {num_pairs * (N - 1)} interpolated frames in {wall:.2f} s = "
              f"{num_pairs * (N - 1) / wall:.1f} frames/s (first forward includes weight packing + graph capture)")
        print("=========================Interpolation Finished=========================")
        print(num_pairs * N + 1, o1, o2)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
