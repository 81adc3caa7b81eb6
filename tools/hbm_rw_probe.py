"""HBM write / copy / read rates of this box with plain torch kernels (fill, copy, sum) on 4 GiB tensors -- the yardstick for the
write-dominated kernels (combine_warps_up writes 96 B per pixel).  usage: python tools/hbm_rw_probe.py"""
import torch

n = 1 << 30          # 4 GiB of float32
a = torch.empty(n, dtype=torch.float32, device="cuda")
b = torch.empty(n, dtype=torch.float32, device="cuda")


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


gb = n * 4 / 1e9
t = timed(lambda: a.fill_(1.0))
print(f"fill  (write only)   : {gb / t / 1e3:.2f} TB/s")
t = timed(lambda: b.copy_(a))
print(f"copy  (read + write) : {2 * gb / t / 1e3:.2f} TB/s of traffic ({gb / t / 1e3:.2f} TB/s each way)")
t = timed(lambda: a.sum())
print(f"sum   (read only)    : {gb / t / 1e3:.2f} TB/s")
