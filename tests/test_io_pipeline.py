"""Host I/O pipeline of the CLI: ordering, decode-once, windowing, error propagation.
Device-free here (CPU tensors); the CUDA streams / pinned buffers are exercised by the CLI test on the GPU."""
import threading
import time

import pytest
import torch

from gimmvfi_hip.io_pipeline import FramePrefetcher, ResultDrain


def test_prefetcher_decodes_each_frame_once_in_order():
    calls = []
    lock = threading.Lock()

    def decode(path):
        with lock:
            calls.append(path)
        time.sleep(0.002)
        return torch.full((1, 3, 4, 6), float(path))

    pf = FramePrefetcher(list(range(9)), "cpu", pad_fn=lambda t: torch.nn.functional.pad(t, (1, 1, 0, 0)), lookahead=3,
                         workers=3, decode=decode)
    for j in range(8):                       # the CLI's access pattern: (j, j+1) for consecutive pairs
        a, b = pf.get(j), pf.get(j + 1)
        assert a.shape == (1, 3, 4, 8) and float(a[0, 0, 0, 1]) == j and float(b[0, 0, 0, 1]) == j + 1
    pf.close()
    assert sorted(calls) == list(range(9)) and pf.decodes == 9          # every frame decoded exactly once
    assert len(pf.ready) <= 3                                           # sliding window, not the whole clip


def test_result_drain_runs_post_processing_off_thread_and_keeps_keys():
    drain = ResultDrain("cpu", depth=2)
    main_thread = threading.get_ident()
    seen = []

    def post(a, b):
        seen.append(threading.get_ident())
        return float(a.sum() + b.sum())

    for k in range(6):
        drain.submit(k, [torch.full((2, 2), float(k)), torch.ones(3)], post)
    out = drain.finish()
    assert out == {k: 4.0 * k + 3.0 for k in range(6)}
    assert all(t != main_thread for t in seen)


def test_result_drain_surfaces_errors():
    drain = ResultDrain("cpu")
    drain.submit(0, [torch.zeros(1)], lambda a: (_ for _ in ()).throw(ValueError("boom")))
    with pytest.raises(ValueError):
        drain.finish()


def test_result_drain_bounds_items_in_flight():
    """At most `depth` items are between submit() and the end of their post-processing (with the posts handed
    to an unbounded pool the pinned buffers in flight were no longer bounded by the queue depth)."""
    depth = 3
    drain = ResultDrain("cpu", depth=depth, workers=2)
    lock = threading.Lock()
    live, peak = [0], [0]
    gate = threading.Event()

    def post(a):
        with lock:
            live[0] += 1
            peak[0] = max(peak[0], live[0])
        gate.wait(timeout=5)
        with lock:
            live[0] -= 1
        return None

    done = []

    def producer():
        for k in range(10):
            drain.submit(k, [torch.zeros(4)], post)
            done.append(k)

    t = threading.Thread(target=producer)
    t.start()
    time.sleep(0.3)
    assert len(done) <= depth          # the producer is blocked by the bound while the posts hang
    gate.set()
    t.join(timeout=20)
    assert len(done) == 10
    drain.finish()
    assert peak[0] <= 2                # (the pool has 2 workers)


def test_video_sink_writes_frames_by_index_from_many_threads(tmp_path):
    """PNG mode of the incremental video writer: frames arrive out of order from several threads, every index is written
    once under its own name; close() checks completeness."""
    import os

    import numpy as np
    from PIL import Image

    from gimmvfi_hip.io_pipeline import VideoSink

    total, h, w = 17, 6, 10
    sink = VideoSink(str(tmp_path / "output.mp4"), 8, total, (h, w), use_cv2=False)
    frames = {i: np.full((h, w, 3), i * 7 % 251, dtype=np.uint8) for i in range(total)}
    for i in frames:
        frames[i][0, 0] = (1, 2, 3)          # BGR marker -> RGB (3, 2, 1) on disk
    order = list(range(total))[::-1]

    def work(idxs):
        for i in idxs:
            sink.put(i, frames[i])

    ts = [threading.Thread(target=work, args=(order[k::3],)) for k in range(3)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    out = sink.close()
    if os.path.isdir(out):                   # no ffmpeg here: the numbered PNGs stay
        names = sorted(os.listdir(out))
        assert names == [f"{i:04d}.png" for i in range(total)]
        im = np.array(Image.open(os.path.join(out, "0005.png")))
        assert im.shape == (h, w, 3) and tuple(im[0, 0]) == (3, 2, 1) and int(im[1, 1, 0]) == 5 * 7 % 251
    sink2 = VideoSink(str(tmp_path / "flow.mp4"), 8, 3, (h, w), use_cv2=False)
    sink2.put(0, frames[0])
    with pytest.raises(AssertionError):
        sink2.close()                        # two frames missing


def test_png_writer_is_lossless_and_standard():
    """png_bytes_rgb (Sub filter + zlib Z_RLE, written by hand for speed) decodes with PIL to exactly the frame, for random
    content, gradients (wrap-around of the byte differences), one-pixel-wide and one-pixel-high images."""
    import io

    import numpy as np
    from PIL import Image

    from gimmvfi_hip.io_pipeline import png_bytes_rgb

    rng = np.random.default_rng(0)
    cases = [rng.integers(0, 256, (37, 53, 3), dtype=np.uint8),
             np.stack([np.add.outer(np.arange(40), 7 * np.arange(90)) % 256] * 3, -1).astype(np.uint8),
             rng.integers(0, 256, (9, 1, 3), dtype=np.uint8), rng.integers(0, 256, (1, 11, 3), dtype=np.uint8)]
    cases.append(np.ascontiguousarray(cases[0][:, ::-1])[:, :, ::-1])        # a non-contiguous view, as the CLI passes (BGR -> RGB)
    for f in cases:
        b = png_bytes_rgb(f)
        im = Image.open(io.BytesIO(b))
        im.load()
        assert im.mode == "RGB" and im.size == (f.shape[1], f.shape[0])
        assert np.array_equal(np.array(im), f)


def test_prefetcher_under_the_round_schedule_decodes_only_the_ranks_frames():
    """with world > 1 a rank asks for non-contiguous blocks of the video.  The prefetcher is given the
    rank's own frame order: it decodes those frames only (about frames / world, not the whole video) and holds a bounded
    number of futures / ready frames at any time -- rank 3 of 8 over 160 pairs used to decode 158 of 161 frames and keep
    128 pinned tensors it never popped."""
    from gimmvfi_hip import shard

    num_pairs, bsz, world, rank = 160, 4, 8, 3
    rounds = shard.round_schedule(num_pairs, bsz, world)
    my_blocks = [rnd[rank] for rnd in rounds]
    my_frames = [j for j0, b in my_blocks if b > 0 for j in range(j0, j0 + b + 1)]
    calls, lock = [], threading.Lock()

    def decode(path):
        with lock:
            calls.append(path)
        return torch.full((1, 3, 2, 2), float(path))

    lookahead = bsz + 3
    pf = FramePrefetcher(list(range(num_pairs + 1)), "cpu", lookahead=lookahead, workers=3, decode=decode, order=my_frames)
    worst = 0
    for j0, b in my_blocks:
        for j in range(j0, j0 + b + 1):
            assert float(pf.get(j)[0, 0, 0, 0]) == j
            worst = max(worst, len(pf.futures) + len(pf.ready))
    pf.close()
    assert sorted(calls) == sorted(set(my_frames)) and pf.decodes == len(set(my_frames))
    assert len(set(my_frames)) == (bsz + 1) * len(rounds)            # 25 of 161 frames for this rank
    assert worst <= lookahead + 3, worst                             # look-ahead + the sliding pair window
    assert not pf.futures                                            # nothing left behind
    with pytest.raises(KeyError):
        pf.get(0)                                                    # a frame of another rank


def test_video_sink_cv2_branch_writes_in_order_and_surfaces_writer_errors(tmp_path, monkeypatch):
    """The reference's container path (video_Nx.py:53-84: cv2.VideoWriter, mp4v) with a stand-in `cv2` module -- OpenCV is
    not in the image, so this branch (in-order writer thread over out-of-order put()s) had never executed."""
    import sys
    import types

    import numpy as np

    from gimmvfi_hip.io_pipeline import VideoSink

    log = {"frames": [], "released": 0, "args": None}

    class Writer:
        def __init__(self, path, fourcc, fps, size):
            log["args"] = (path, fourcc, fps, size)

        def write(self, frame):
            if log.get("fail_at") == len(log["frames"]):
                raise IOError("disk full")
            log["frames"].append(int(frame[0, 0, 0]))

        def release(self):
            log["released"] += 1

    fake = types.ModuleType("cv2")
    fake.VideoWriter = Writer
    fake.VideoWriter_fourcc = lambda *a: "".join(a)
    monkeypatch.setitem(sys.modules, "cv2", fake)
    path = str(tmp_path / "output.mp4")
    sink = VideoSink(path, 16, 12, (8, 20))
    assert sink.cv2 is fake and log["args"] == (path, "mp4v", 16, (20, 8))          # (width, height) as cv2 wants it
    for idx in (3, 0, 1, 2, 7, 6, 5, 4, 11, 10, 9, 8):                                # producers finish out of order
        sink.put(idx, np.full((8, 20, 3), idx, np.uint8))
    assert sink.close() == path
    assert log["frames"] == list(range(12)) and log["released"] == 1 and sink.written == 12
    # frames larger than 2048 px go to the PNG + ffmpeg path even with OpenCV present (reference video_Nx.py:62-84)
    big = VideoSink(str(tmp_path / "big.mp4"), 16, 1, (2176, 2 * 4096), png_workers=1)
    assert big.cv2 is None
    big.put(0, np.zeros((4, 4, 3), np.uint8))
    assert big.close() is not None
    # a writer error stops the sink: later put()s raise instead of queueing behind a dead writer, close() re-raises
    log.update(frames=[], fail_at=2)
    bad = VideoSink(str(tmp_path / "bad.mp4"), 16, 6, (8, 20))
    for idx in range(3):
        bad.put(idx, np.full((8, 20, 3), idx, np.uint8))
    deadline = time.time() + 5.0
    while bad.err is None and time.time() < deadline:
        time.sleep(0.01)
    with pytest.raises(IOError):
        bad.put(3, np.zeros((8, 20, 3), np.uint8))
    with pytest.raises(IOError):
        bad.close()
