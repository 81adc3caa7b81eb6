"""Frame-pair sharding over the GPUs of one node.

Adjacent frame pairs are independent forwards (reference src/video_Nx.py:134-181 carries no state
across pairs), so the path shards with no data-path collective.  Two schedules: `pair_range` gives rank r one
CONTIGUOUS range of pair indices (bench-style batch jobs: `gather_frames` collects everything once); `round_schedule`
+ `RoundGather` stream a video: contiguous blocks per ROUND, gathered round by round, so that rank 0 writes the output
in order with bounded memory (the CLI).  The only collective is the gather of the uint8 result frames to rank 0 (RCCL
over xGMI on GPUs, gloo in the CPU tests)."""
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
