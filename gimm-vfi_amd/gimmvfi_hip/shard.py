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
