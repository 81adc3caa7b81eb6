"""Frame-pair sharding over the GPUs of one node.

Adjacent frame pairs are independent forwards (reference src/video_Nx.py:134-181 carries no state
across pairs), so the path shards with no data-path collective.  Two schedules: `pair_range` gives rank r one
CONTIGUOUS range of pair indices (bench-style batch jobs: `gather_frames` collects everything once); `round_schedule`
+ `RoundGather` stream a video: contiguous blocks per ROUND, gathered round by round, so that rank 0 writes the output
in order with bounded memory (the CLI).  The only collective is the gather of the result frames to rank 0 (RCCL over xGMI
on GPUs, gloo in the CPU tests): raw uint8 frames (`RoundGather`, the OpenCV / mp4 sink) or -- round 5, the PNG sink -- the
frames each rank has already ENCODED on its own host threads (`BytesGather`), whose size exchange also carries the abort flag
that ends every rank within one round when any of them fails."""
import numpy as np
import torch
import torch.distributed as dist


def pair_range(num_pairs: int, rank: int, world: int):
    """Contiguous [start, stop) of pair indices owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(num_pairs, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def gather_frames(local: torch.Tensor, num_pairs: int, rank: int, world: int, dst: int = 0):
    """local: uint8 [n_local, ...] frames of this rank's pairs (n_local = size of its pair_range).
    Returns on `dst` the concatenation over ranks in pair order ([num_pairs, ...]), None elsewhere."""
    if world == 1:
        return local
    counts = [pair_range(num_pairs, r, world)[1] - pair_range(num_pairs, r, world)[0] for r in range(world)]
    assert local.shape[0] == counts[rank], (local.shape, counts, rank)
    cmax = max(counts)
    pad = torch.zeros((cmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)


def round_schedule(num_pairs: int, pairs_per_forward: int, world: int):
    """Streaming schedule of the CLI: round k covers the CONTIGUOUS block of world * pairs_per_forward pairs starting at
    k * world * pairs_per_forward, rank r runs the r-th sub-block of it as one forward (consecutive pairs, so the per-frame
    encoder work of shared frames is still reused inside a forward).  Returns [[(first pair, pairs) per rank] per round];
    a rank beyond the end of the video gets (num_pairs, 0).  After round k every pair below (k+1) * world * ppf is done, so
    rank 0 can write the output video in order with O(one round) of frames in memory -- a contiguous range per rank
    (pair_range) would make it hold the whole video."""
    ppf = max(1, pairs_per_forward)
    per_round = ppf * world
    rounds = []
    for base in range(0, num_pairs, per_round):
        rounds.append([(min(num_pairs, base + r * ppf), max(0, min(ppf, num_pairs - base - r * ppf))) for r in range(world)])
    return rounds


class RoundGather:
    """The path's only collective, one round at a time: every rank contributes the uint8 result tensors of ITS forward of
    the round (any number of tensors with leading dimension = its pair count x something), rank `dst` receives all of them
    into persistent, double-buffered staging tensors (RCCL over xGMI on GPUs, gloo in the CPU tests).  Memory on `dst` is
    2 x world x one forward's results, whatever the length of the video."""

    def __init__(self, rank: int, world: int, dst: int = 0):
        self.rank, self.world, self.dst = rank, world, dst
        self._send = {}
        self._recv = {}
        self.parity = 0

    def _bufs(self, slot, shape, dtype, device):
        key = (slot, self.parity, tuple(shape), dtype)
        if key not in self._send:
            self._send[key] = torch.zeros(shape, dtype=dtype, device=device)
            if self.rank == self.dst:
                self._recv[key] = [torch.empty(shape, dtype=dtype, device=device) for _ in range(self.world)]
        return self._send[key], self._recv.get(key)

    def gather(self, tensors, full_shapes, counts):
        """tensors[s]: this rank's s-th result (leading dim n_s <= full_shapes[s][0]; may be empty), full_shapes[s] the
        padded per-rank shape (same on every rank), counts[s][r] the valid leading length of rank r's contribution.
        Returns on dst [[rank r's tensor s trimmed to counts[s][r]] per s], None elsewhere.  The returned views alias the
        staging buffers of this round's parity: consume (or copy) them before the round after next."""
        out = []
        for s, (t, shp) in enumerate(zip(tensors, full_shapes)):
            send, recv = self._bufs(s, shp, t.dtype, t.device)
            send[: t.shape[0]].copy_(t)
            dist.gather(send, recv, dst=self.dst)
            if self.rank == self.dst:
                out.append([recv[r][: counts[s][r]] for r in range(self.world)])
        self.parity ^= 1
        return out if self.rank == self.dst else None


# ------------------------------------------------------------------ encoded results: per-rank encoding, gather of the compressed bytes
# Round 5.  RoundGather moves RAW uint8 frames: every frame of every rank is then PNG-encoded (or muxed) by rank 0's host threads,
# which on ONE GPU already held the 2K run at 72 % of the model rate -- at 8 GPUs the sink would be an 8x bottleneck.  With the PNG
# sink each rank encodes its own frames on its own host threads and the collective carries the compressed bytes: rank 0 only
# orders and writes them.  The size exchange in front of the payload also carries an ABORT flag, so an error on any rank (a dead
# sink on rank 0, a failed forward elsewhere) ends every rank at the next round instead of leaving the others blocked in a gather
# until the RCCL timeout.
_KINDS = ("out", "flow")


def pack_entries(entries):
    """entries: [(kind "out" | "flow", index in that video, encoded bytes)] -> one uint8 array:
    [n u32][n x (kind u32, index u32, length u32)][payloads back to back]."""
    n = len(entries)
    head = np.zeros(1 + 3 * n, np.uint32)
    head[0] = n
    total = 0
    for i, (kind, idx, data) in enumerate(entries):
        head[1 + 3 * i:4 + 3 * i] = (_KINDS.index(kind), idx, len(data))
        total += len(data)
    buf = np.empty(head.nbytes + total, np.uint8)
    buf[:head.nbytes] = head.view(np.uint8)
    off = head.nbytes
    for _, _, data in entries:
        buf[off:off + len(data)] = np.frombuffer(data, np.uint8)
        off += len(data)
    return buf


def unpack_entries(buf):
    """Inverse of pack_entries; the payloads are views into `buf`."""
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    if buf.size == 0:
        return []
    n = int(buf[:4].view(np.uint32)[0])
    head = buf[4:4 + 12 * n].view(np.uint32).reshape(n, 3)
    off = 4 + 12 * n
    out = []
    for kind, idx, ln in head.tolist():
        out.append((_KINDS[kind], idx, buf[off:off + ln]))
        off += ln
    assert off == buf.size, (off, buf.size)
    return out


def any_abort(flag: bool, device, world: int):
    """True on every rank when ANY rank raised its flag (one all_gather of a byte-sized tensor: the abort path of the raw-frame
    gather; BytesGather carries the flag in its size exchange)."""
    if world == 1:
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
    allf = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(allf, t)
    return bool(int(torch.stack(allf).sum().item()) > 0)


class BytesGather:
    """Gather of one variable-length byte payload per rank and round to `dst` (RCCL over xGMI on GPUs; gloo on CPU tensors in
    the tests / the dry rehearsal).  Every round: (1) all_gather of [length, abort flag] -- every rank learns every length and
    whether anybody wants to stop; (2) unless stopping, one dist.gather of the payloads padded to the round's longest (persistent
    staging tensors, grown in 1 MiB steps).  The payload arrives on `dst` as host numpy arrays, in rank order."""

    def __init__(self, rank: int, world: int, device, dst: int = 0):
        self.rank, self.world, self.dst, self.device = rank, world, dst, torch.device(device)
        self._send = None
        self._recv = None
        self.bytes_moved = 0        # statistics: payload bytes received on dst

    def _cap(self, cur, need, n=1):
        if cur is not None and cur[0].numel() >= need:
            return cur
        return [torch.empty(need, dtype=torch.uint8, device=self.device) for _ in range(n)]

    def exchange(self, payload, abort: bool = False):
        """payload: uint8 numpy array (or None = nothing this round).  Returns (payloads per rank on dst | None, stop): stop is
        True on EVERY rank as soon as one rank passed abort=True; no payload moves in that round."""
        n = 0 if payload is None else int(payload.size)
        meta = torch.tensor([n, 1 if abort else 0], dtype=torch.int64, device=self.device)
        allm = [torch.zeros_like(meta) for _ in range(self.world)]
        if self.world > 1:
            dist.all_gather(allm, meta)
        else:
            allm = [meta]
        m = torch.stack(allm).cpu()
        sizes = [int(v) for v in m[:, 0]]
        if int(m[:, 1].sum()) > 0:
            return None, True
        if self.world == 1:
            return ([payload if payload is not None else np.zeros(0, np.uint8)], False)
        cap = max(1, (max(sizes) + (1 << 20) - 1) >> 20 << 20)
        self._send = self._cap(self._send, cap)
        send = self._send[0][:cap]
        if n:
            send[:n].copy_(torch.from_numpy(payload))
        recv = None
        if self.rank == self.dst:
            self._recv = self._cap(self._recv, cap, self.world)
            recv = [t[:cap] for t in self._recv]
        dist.gather(send, recv, dst=self.dst)
        if self.rank != self.dst:
            return None, False
        self.bytes_moved += sum(sizes)
        return [recv[r][:sizes[r]].cpu().numpy() for r in range(self.world)], False
