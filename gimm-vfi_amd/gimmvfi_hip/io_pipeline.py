"""Host I/O pipeline around the hot loop of src/video_Nx.py.

The reference's loop (video_Nx.py:134-216) is synchronous: PIL decode of both frames of a pair (every interior frame is
decoded twice), pad, blocking H2D, forward, blocking D2H per output frame.  At MI355X speeds (a 448x256 pair in a few
milliseconds) that loop, not the model, sets the frames/s of the CLI.  Here:

* ``FramePrefetcher``: a small thread pool decodes + converts + pads each frame ONCE, a bounded look-ahead ahead of the
  consumer, into pinned host memory; the H2D copy is enqueued on a side HIP stream and handed over with an event, so
  decode and upload of pair j+1 overlap the forward of pair j;
* ``ResultDrain``: result tensors are copied D2H asynchronously into pinned buffers on a second side stream; the
  consumer thread does the CPU post-processing (colour-coding of flow, BGR conversion) while the GPU runs ahead.  At most
  ``depth`` items are between `submit` and the end of their post-processing, so the pinned buffers in flight are bounded;
* ``VideoSink``: the output video is written WHILE the GPU computes: frames are handed over by index from the
  post-processing threads and encoded in order (cv2.VideoWriter) or as numbered PNGs (any order) by the sink -- the
  reference collects every frame of the video in a list and writes it after the loop (video_Nx.py:236-250).

Pure host plumbing (threads, pinned buffers, streams, events): no arithmetic of the hot path lives here.
"""
import queue
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch


def decode_rgb01(path):
    """reference video_Nx.py:46-50: RGB uint8 -> float CHW in [0,1] with a leading batch axis."""
    from PIL import Image

    raw = np.array(Image.open(path).convert("RGB"))
    return (torch.from_numpy(raw.copy()).permute(2, 0, 1) / 255.0).to(torch.float).unsqueeze(0)


class FramePrefetcher:
    """frames[i] -> device tensor (1,3,Hp,Wp), decoded once, ``lookahead`` frames ahead of the consumer.

    ``pad_fn(t) -> t`` is applied on the host (the CLI passes InputPadder.pad).  ``get(i)`` must be called with
    non-decreasing i (each frame may be requested any number of times while it is within the window).

    ``order``: the frame indices this consumer will ask for, ascending (default: every frame).  Under the multi-GPU round
    schedule a rank owns non-contiguous blocks of the video: only ITS frames are decoded (decode work and pinned memory
    per rank are O(video / world) and O(look-ahead), not O(video)), and the look-ahead runs along that order."""

    def __init__(self, paths, device, pad_fn=None, lookahead=4, workers=4, decode=decode_rgb01, order=None):
        self.paths, self.device, self.pad_fn, self.decode = list(paths), torch.device(device), pad_fn, decode
        self.order = list(range(len(self.paths))) if order is None else sorted(set(int(i) for i in order))
        assert all(0 <= i < len(self.paths) for i in self.order)
        self.pos = {i: k for k, i in enumerate(self.order)}
        self.lookahead = max(1, lookahead)
        self.pool = ThreadPoolExecutor(max_workers=workers)
        self.on_gpu = self.device.type == "cuda"
        self.copy_stream = torch.cuda.Stream(device=self.device) if self.on_gpu else None
        self.futures = {}      # index -> Future[(host tensor)]
        self.ready = {}        # index -> (device tensor, event)
        self.next_submit = 0   # position in self.order of the next frame to hand to the decode pool
        self.decodes = 0       # statistics (tests): every frame is decoded exactly once
        self._lock = threading.Lock()

    def _load(self, i):
        t = self.decode(self.paths[i])
        if self.pad_fn is not None:
            t = self.pad_fn(t)
        t = t.contiguous()
        if self.on_gpu:
            t = t.pin_memory()
        with self._lock:
            self.decodes += 1
        return t

    def _submit_upto(self, hi_pos):
        hi_pos = min(hi_pos, len(self.order) - 1)
        while self.next_submit <= hi_pos:
            i = self.order[self.next_submit]
            self.futures[i] = self.pool.submit(self._load, i)
            self.next_submit += 1

    def get(self, i):
        if i not in self.pos:
            raise KeyError(f"frame {i} is not in this prefetcher's order")
        self._submit_upto(self.pos[i] + self.lookahead)
        for k in [k for k in self.futures if k < i]:        # never requested and now behind the consumer: drop
            self.futures.pop(k).cancel()
        if i not in self.ready:
            host = self.futures.pop(i).result()
            if self.on_gpu:
                with torch.cuda.stream(self.copy_stream):
                    dev = host.to(self.device, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(self.copy_stream)
                self.ready[i] = (dev, ev, host)     # keep the pinned source alive until the copy is consumed
            else:
                self.ready[i] = (host, None, None)
        dev, ev, _ = self.ready[i]
        if ev is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            # the block was allocated on copy_stream's pool: tell the caching allocator that the consumer stream reads
            # it, otherwise dropping the entry below hands the block back to copy_stream while kernels queued on `cur`
            # (the host runs several pairs ahead of the GPU) have not read it yet and a later upload overwrites it
            dev.record_stream(cur)
        for k in [k for k in self.ready if k < i - 1]:      # frames behind the sliding pair window are done
            del self.ready[k]
        return dev

    def close(self):
        self.pool.shutdown(wait=False, cancel_futures=True)


class ResultDrain:
    """Asynchronous D2H of result tensors + CPU post-processing in a consumer thread.

    ``submit(key, tensors, post)``: ``tensors`` (device) are copied to pinned host buffers on a side stream after the
    work already enqueued on the current stream; ``post(*host_tensors)`` runs in the consumer thread once the copy has
    landed.  ``results()`` returns {key: post result} after ``finish()``."""

    def __init__(self, device, depth=8, workers=4):
        self.slots = threading.Semaphore(max(1, depth))     # items between submit() and the end of their post()
        self.device = torch.device(device)
        self.on_gpu = self.device.type == "cuda"
        self.stream = torch.cuda.Stream(device=self.device) if self.on_gpu else None
        self.q = queue.Queue(maxsize=depth)
        self.out = {}
        self.err = None
        # the copies are awaited in order by one thread; the CPU post-processing of different items (colour-coding,
        # image encoding: numpy / PIL release the GIL) runs in a small pool so that it keeps up with the GPU
        self.pool = ThreadPoolExecutor(max_workers=max(1, workers))
        self.pending = []
        self._free = {}                      # (shape, dtype) -> recycled pinned host buffers
        self._pool_lock = threading.Lock()
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def _pinned(self, t):
        """A pinned host buffer for a copy of `t`: from the free list of recycled buffers of that shape, else a new one
        (hipHostMalloc of a 100 MB buffer costs milliseconds of the submitting thread -- the CLI's main loop)."""
        key = (tuple(t.shape), t.dtype)
        with self._pool_lock:
            lst = self._free.get(key)
            if lst:
                return lst.pop()
        return torch.empty(t.shape, dtype=t.dtype, device="cpu", pin_memory=True)

    def submit(self, key, tensors, post, recycle=False):
        """Returns the event that marks the end of the D2H copies (None on CPU): a caller that re-uses `tensors` as
        staging buffers waits for it before overwriting them.  recycle: `post` is DONE with the host tensors when it returns
        (it keeps no views of them) -- their pinned buffers go back to a free list for later submits."""
        self.slots.acquire()
        if self.err is None:
            # a post() that already failed (sink.put, an encoder) surfaces at the next submit, not only at finish(): the GPU
            # loop must not run the rest of the video into a dead writer
            for _, fut in self.pending:
                if fut.done() and fut.exception() is not None:
                    self.err = fut.exception()
                    break
        if self.err is not None:
            self.slots.release()
            raise self.err
        ev = None
        if self.on_gpu:
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(self.device))
            hosts = []
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(done)
                for t in tensors:
                    h = self._pinned(t) if recycle else torch.empty(t.shape, dtype=t.dtype, device="cpu", pin_memory=True)
                    h.copy_(t, non_blocking=True)
                    hosts.append(h)
                ev = torch.cuda.Event()
                ev.record(self.stream)
            self.q.put((key, hosts, ev, (post, recycle), tensors))     # `tensors` kept alive until the copy has completed
        else:
            self.q.put((key, [t.clone() for t in tensors], None, (post, False), None))
        return ev

    def _post(self, post, hosts):
        post, recycle = post
        try:
            return post(*hosts)
        finally:
            if recycle:
                with self._pool_lock:
                    for h in hosts:
                        self._free.setdefault((tuple(h.shape), h.dtype), []).append(h)
            self.slots.release()

    def _run(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            key, hosts, ev, post, _keep = item
            try:
                if ev is not None:
                    ev.synchronize()
                self.pending.append((key, self.pool.submit(self._post, post, hosts)))
                self.pending = [(k, f) for k, f in self.pending if not (f.done() and f.exception() is None and f.result() is None)] \
                    if len(self.pending) > 64 else self.pending       # posts that return nothing leave nothing behind
            except Exception as e:      # surfaced by finish() / the next submit()
                self.err = e
                self.slots.release()

    def finish(self):
        self.q.put(None)
        self.thread.join()
        for key, fut in self.pending:
            try:
                self.out[key] = fut.result()
            except Exception as e:
                self.err = self.err or e
        self.pool.shutdown(wait=True)
        if self.err is not None:
            raise self.err
        return self.out


def png_bytes_rgb(rgb):
    """A lossless 8-bit RGB PNG of ``rgb`` (H, W, 3) uint8, built for speed: PNG filter 1 (Sub: byte minus the byte of the
    pixel to its left, one vectorised numpy subtraction) + zlib level 1 with the Z_RLE strategy (what zlib recommends for
    filtered PNG data).  A 2K side-by-side frame (1088 x 4096) takes ~0.17 s and 7.9 MB; PIL's encoder at
    compress_level=1 ~1.0 s and 8.9 MB (adaptive filter search + default deflate): the CLI's writer threads were the
    bottleneck of the 2K / 4K runs.  zlib and numpy release the GIL, so a thread pool scales."""
    import struct
    import zlib

    rgb = np.ascontiguousarray(rgb)
    h, w, c = rgb.shape
    assert c == 3 and rgb.dtype == np.uint8
    flat = rgb.reshape(h, w * 3)
    raw = np.empty((h, 1 + w * 3), np.uint8)
    raw[:, 0] = 1
    raw[:, 1:4] = flat[:, :3]
    np.subtract(flat[:, 3:], flat[:, :-3], out=raw[:, 4:])
    co = zlib.compressobj(1, zlib.DEFLATED, 15, 9, zlib.Z_RLE)
    data = co.compress(memoryview(raw).cast("B")) + co.flush()      # (no copy of the 13 MB; zlib releases the GIL)

    def chunk(tag, payload):
        return struct.pack(">I", len(payload)) + tag + payload + struct.pack(">I", zlib.crc32(tag + payload) & 0xffffffff)

    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", data) + chunk(b"IEND", b"")


class VideoSink:
    """Incremental writer of one output video (reference images_to_video, video_Nx.py:53-84, fed frame by frame).

    ``put(index, frame_hwc_bgr_u8)`` may be called from any thread in any order; ``close()`` returns the path that was
    written.  With OpenCV (and frames the mp4v writer accepts) frames are encoded in index order by one writer thread --
    out-of-order frames wait in a dict whose size the producers bound (ResultDrain depth); otherwise every frame becomes
    a PNG ``<stem>_frames/<index>.png`` (png_bytes_rgb: Sub filter + zlib level 1 / Z_RLE, ~0.17 s per 2K side-by-side frame
    against ~1 s for PIL's encoder) written by a pool of threads -- the files are independent;
    ``put`` blocks once workers + 32 frames are waiting (bounded memory) -- and ffmpeg, when present, encodes them at
    close.  ``total``: number of frames, known up front."""

    @staticmethod
    def uses_cv2(frame_hw, use_cv2=None):
        """Whether a sink for frames of this size encodes through cv2.VideoWriter (needs the raw frames, in order) or writes
        PNG files (independent, may arrive already encoded: put_png).  Same answer on every rank of a node: the multi-GPU
        CLI picks its result path with it."""
        try:
            import cv2  # noqa: F401
        except Exception:
            return False
        h, w = frame_hw
        return max(h, w // 2) <= 2048 and use_cv2 is not False

    def __init__(self, path, fps, total, frame_hw, use_cv2=None, png_workers=None):
        import os

        self.path, self.fps, self.total = path, fps, total
        cv2 = None
        if VideoSink.uses_cv2(frame_hw, use_cv2):
            import cv2
        h, w = frame_hw
        self.cv2 = cv2
        self.written = 0
        self.err = None
        self._lock = threading.Condition()
        self._pending = {}
        self._closed = False
        if self.cv2 is not None:
            self.writer = self.cv2.VideoWriter(path, self.cv2.VideoWriter_fourcc(*"mp4v"), fps, (w, h))
            self.thread = threading.Thread(target=self._run, daemon=True)
            self.thread.start()
        else:
            self.frame_dir = os.path.splitext(path)[0] + "_frames"
            os.makedirs(self.frame_dir, exist_ok=True)
            # (one thread encodes ~6 side-by-side 2K frames per second; the GPU delivers ~200 output + flow frames per second)
            nw = png_workers if png_workers is not None else max(1, min(96, (os.cpu_count() or 4) // 2))
            self.png_pool = ThreadPoolExecutor(max_workers=nw)
            self.png_slots = threading.BoundedSemaphore(nw + 32)

    def _save_png(self, index, frame):
        import os

        try:
            with open(os.path.join(self.frame_dir, f"{index:04d}.png"), "wb") as f:
                f.write(png_bytes_rgb(frame[:, :, ::-1]))
            with self._lock:
                self.written += 1
        except Exception as e:          # surfaced by close()
            self.err = self.err or e
        finally:
            self.png_slots.release()

    def _write_png(self, index, data):
        import os

        try:
            with open(os.path.join(self.frame_dir, f"{index:04d}.png"), "wb") as f:
                f.write(data)
            with self._lock:
                self.written += 1
        except Exception as e:
            self.err = self.err or e
        finally:
            self.png_slots.release()

    def put_png(self, index, data):
        """PNG sink only: a frame that arrives already encoded (the multi-GPU CLI: every rank encodes its own frames,
        gimmvfi_hip/shard.py:BytesGather) -- this sink just writes the file."""
        assert self.cv2 is None and 0 <= index < self.total
        if self.err is not None:
            raise self.err
        self.png_slots.acquire()
        self.png_pool.submit(self._write_png, index, bytes(data))

    def put(self, index, frame):
        assert 0 <= index < self.total
        if self.cv2 is None:
            if self.err is not None:
                raise self.err
            self.png_slots.acquire()
            self.png_pool.submit(self._save_png, index, frame)
            return
        if self.err is not None:            # the ordered writer died: do not pile frames up behind it
            raise self.err
        with self._lock:
            self._pending[index] = frame
            self._lock.notify_all()

    def _run(self):
        nxt = 0
        while nxt < self.total:
            with self._lock:
                while nxt not in self._pending and not self._closed:
                    self._lock.wait(timeout=1.0)
                if nxt not in self._pending:
                    return                      # closed early (error path)
                frame = self._pending.pop(nxt)
            try:
                self.writer.write(np.ascontiguousarray(frame))
            except Exception as e:
                self.err = e
                return
            nxt += 1
            self.written = nxt

    def close(self):
        import os
        import shutil
        import subprocess

        if self.cv2 is not None:
            with self._lock:
                missing = self.total - self.written - len(self._pending)
                if missing > 0:
                    self._closed = True
                self._lock.notify_all()
            self.thread.join()
            self.writer.release()
            if self.err is not None:
                raise self.err
            assert self.written == self.total, (self.written, self.total)
            return self.path
        self.png_pool.shutdown(wait=True)
        if self.err is not None:
            raise self.err
        assert self.written == self.total, (self.written, self.total)
        if shutil.which("ffmpeg"):
            # yuv420p needs even dimensions: pad odd frames by one row / column instead of failing
            r = subprocess.run(["ffmpeg", "-y", "-framerate", f"{self.fps}", "-i", f"{self.frame_dir}/%04d.png",
                                "-vf", "pad=ceil(iw/2)*2:ceil(ih/2)*2", "-c:v", "libx264", "-pix_fmt", "yuv420p", self.path])
            if r.returncode == 0 and os.path.isfile(self.path) and os.path.getsize(self.path) > 0:
                shutil.rmtree(self.frame_dir, ignore_errors=True)
                return self.path
            print(f"[VideoSink] ffmpeg failed (exit code {r.returncode}); the frames are kept in {self.frame_dir}")
        return self.frame_dir
