"""Host I/O pipeline around the hot loop of src/video_Nx.py.

The reference's loop (video_Nx.py:134-216) is synchronous: PIL decode of both frames of a pair (every interior frame is
decoded twice), pad, blocking H2D, forward, blocking D2H per output frame.  At MI355X speeds (a 448x256 pair in a few
milliseconds) that loop, not the model, sets the frames/s of the CLI.  Here:

* ``FramePrefetcher``: a small thread pool decodes + converts + pads each frame ONCE, a bounded look-ahead ahead of the
  consumer, into pinned host memory; the H2D copy is enqueued on a side HIP stream and handed over with an event, so
  decode and upload of pair j+1 overlap the forward of pair j;
* ``ResultDrain``: result tensors are copied D2H asynchronously into pinned buffers on a second side stream; the
  consumer thread does the CPU post-processing (colour-coding of flow, BGR conversion) while the GPU runs ahead.

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
    non-decreasing i (each frame may be requested any number of times while it is within the window)."""

    def __init__(self, paths, device, pad_fn=None, lookahead=4, workers=4, decode=decode_rgb01):
        self.paths, self.device, self.pad_fn, self.decode = list(paths), torch.device(device), pad_fn, decode
        self.lookahead = max(1, lookahead)
        self.pool = ThreadPoolExecutor(max_workers=workers)
        self.on_gpu = self.device.type == "cuda"
        self.copy_stream = torch.cuda.Stream(device=self.device) if self.on_gpu else None
        self.futures = {}      # index -> Future[(host tensor)]
        self.ready = {}        # index -> (device tensor, event)
        self.next_submit = 0
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

    def _submit_upto(self, hi):
        hi = min(hi, len(self.paths) - 1)
        while self.next_submit <= hi:
            self.futures[self.next_submit] = self.pool.submit(self._load, self.next_submit)
            self.next_submit += 1

    def get(self, i):
        self._submit_upto(i + self.lookahead)
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
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def submit(self, key, tensors, post):
        if self.on_gpu:
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(self.device))
            hosts = []
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(done)
                for t in tensors:
                    h = torch.empty(t.shape, dtype=t.dtype, device="cpu", pin_memory=True)
                    h.copy_(t, non_blocking=True)
                    hosts.append(h)
                ev = torch.cuda.Event()
                ev.record(self.stream)
            self.q.put((key, hosts, ev, post, tensors))     # `tensors` kept alive until the copy has completed
        else:
            self.q.put((key, [t.clone() for t in tensors], None, post, None))

    def _run(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            key, hosts, ev, post, _keep = item
            try:
                if ev is not None:
                    ev.synchronize()
                self.pending.append((key, self.pool.submit(post, *hosts)))
            except Exception as e:      # surfaced by finish()
                self.err = e

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
