"""Frame-pair sharding over the GPUs of one node.

Adjacent frame pairs are independent forwards (reference src/video_Nx.py:134-181 carries no state
across pairs), so the path shards with no data-path collective: rank r owns a CONTIGUOUS range of
pair indices (contiguous so that per-frame encoder work of a shared frame could be reused on one
rank).  The only collective is the gather of the uint8 result frames to rank 0 (RCCL over xGMI on
GPUs, gloo in the CPU tests)."""
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


def gather_frames_chunked(local: torch.Tensor, num_pairs: int, rank: int, world: int, dst: int = 0,
                          chunk_pairs: int = 16):
    """gather_frames in rounds of at most `chunk_pairs` pairs per rank, so the padded staging buffers stay bounded for
    long videos at 2K / 4K (a 4K pair at 8x is 186 MB of uint8 frames).  Same result."""
    if world == 1:
        return local
    counts = [pair_range(num_pairs, r, world)[1] - pair_range(num_pairs, r, world)[0] for r in range(world)]
    rounds = (max(counts) + chunk_pairs - 1) // chunk_pairs
    parts = [[] for _ in range(world)]
    for c in range(rounds):
        lo, hi = c * chunk_pairs, (c + 1) * chunk_pairs
        n_r = [max(0, min(cnt, hi) - lo) for cnt in counts]
        cmax = max(n_r)
        pad = torch.zeros((cmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        mine = local[lo:min(hi, local.shape[0])]
        pad[: mine.shape[0]] = mine
        bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
        dist.gather(pad, bufs, dst=dst)
        if rank == dst:
            for r in range(world):
                parts[r].append(bufs[r][: n_r[r]])
    if rank != dst:
        return None
    return torch.cat([torch.cat(p, 0) if p else local[:0] for p in parts], dim=0)
