"""RCCL smoke run on ONE GPU (world size 1): `backend="nccl"` IS RCCL on ROCm.  Proves that librccl loads on this box, that the
process group comes up over 127.0.0.1 and that the collectives the path uses take DEVICE tensors: the round gather of uint8 result
frames (shard.RoundGather -> dist.gather), the one-word abort exchange (dist.all_gather) and the barrier / max-reduce of bench.py's
timed region.  A one-rank group moves no bytes over xGMI -- what it shows is the call path, not the fabric (no multi-GPU node has
been available to the builder in any round: DESIGN.md section 6).  Prints `RCCL OK ...` and exits 0 on success."""
import os
import socket
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "gimm-vfi_amd"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from gimmvfi_hip import shard  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s.getsockname()[1])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
    dev = torch.device("cuda", torch.cuda.current_device())
    dist.init_process_group("nccl", rank=rank, world_size=world)
    try:
        # (1) the result gather of a round: 3 frames of 64 x 96 RGB per rank
        rg = shard.RoundGather(rank, world)
        frames = (torch.arange(3 * 64 * 96 * 3, device=dev) % 251).to(torch.uint8).reshape(3, 64, 96, 3) + rank
        got = rg.gather([frames], [(3, 64, 96, 3)], [[3] * world])
        if rank == 0:
            assert len(got) == 1 and len(got[0]) == world
            for r in range(world):
                assert got[0][r].is_cuda and torch.equal(got[0][r], (frames - rank + r).to(torch.uint8)), r
        # (2) the abort word (shard.any_abort takes a shortcut at world size 1: the collective itself here)
        t = torch.tensor([rank + 5], dtype=torch.int32, device=dev)
        allf = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allf, t)
        assert [int(v) for v in torch.stack(allf).flatten().tolist()] == [r + 5 for r in range(world)]
        # (3) bench.py's timed-region bracket: barrier + max over ranks
        dist.barrier()
        ms = torch.tensor([1.5 + rank], dtype=torch.float64, device=dev)
        dist.all_reduce(ms, op=dist.ReduceOp.MAX)
        assert float(ms) == 1.5 + world - 1
        torch.cuda.synchronize()
        ver = ".".join(str(v) for v in torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else "?"
        if rank == 0:
            print(f"RCCL OK world={world} backend={dist.get_backend()} version={ver} device={torch.cuda.get_device_name(dev)}")
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
