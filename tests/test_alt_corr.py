"""Volume-free correlation lookup: the reference's native module alt_cuda_corr.

* the CPU restatement (oracle/alt_corr_oracle.py) is pinned against the CorrBlock lookup of the main oracle, which is
  itself bit-exact against the reference (raft/corr.py documents both blocks as equivalent);
* the HIP kernel is checked against the restatement: in the host emulator here, through the C ABI on the GPU.
Tolerance: 2e-5 * |corr|max (float accumulation order differs: lanes stride the channel axis).
"""
import ctypes as C

import pytest
import torch

import alt_corr_oracle as ao
import gimmvfi_r_oracle as orc


def _case(B=2, Cc=96, h=16, w=24, seed=0, spread=3.0):
    g = torch.Generator().manual_seed(seed)
    f1 = torch.randn(B, Cc, h, w, generator=g)
    f2 = torch.randn(B, Cc, h, w, generator=g)
    base = torch.stack(torch.meshgrid(torch.arange(w).float(), torch.arange(h).float(), indexing="xy"), 0)
    coords = base[None].repeat(B, 1, 1, 1) + torch.randn(B, 2, h, w, generator=g) * spread
    coords[0, :, 0, 0] = torch.tensor([-7.3, 2.5])        # window partly / fully outside fmap2
    coords[0, :, 1, 1] = torch.tensor([w + 9.0, h + 9.0])
    return f1, f2, coords


def test_oracle_alt_corr_equals_pinned_corrblock_lookup():
    f1, f2, coords = _case()
    a = ao.alternate_corr_block(f1, f2, coords, 4, 4)
    b = orc.corr_lookup(orc.corr_pyramid(orc.corr_volume(f1, f2)), coords, 4)
    assert a.shape == b.shape == (2, 324, 16, 24)
    assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())


def _run_lib(lib, f1_nhwc, f2_nhwc, coords5, r, dtype_flag, stream):
    B, N, H, W = coords5.shape[:4]
    rd = 2 * r + 1
    corr = torch.full((B, N, rd * rd, H, W), float("nan"), dtype=torch.float32, device=f1_nhwc.device)
    rc = lib.alt_corr_forward(f1_nhwc.data_ptr(), f2_nhwc.data_ptr(), coords5.data_ptr(), corr.data_ptr(), B, N, H, W,
                              f2_nhwc.shape[1], f2_nhwc.shape[2], f1_nhwc.shape[3], r, dtype_flag, stream)
    assert rc == 0
    return corr


@pytest.mark.parametrize("r,N", [(4, 1), (2, 3)])
def test_emulated_kernel_matches_restatement(r, N):
    from sim_runtime import hostsim_lib

    lib = hostsim_lib()
    f1, f2, coords = _case(B=1, Cc=80, h=6, w=9, seed=3)
    a = f1.permute(0, 2, 3, 1).contiguous()
    b = f2.permute(0, 2, 3, 1).contiguous()
    c5 = coords.permute(0, 2, 3, 1).reshape(1, 1, 6, 9, 2).repeat(1, N, 1, 1, 1).contiguous()
    c5[:, 1:] += 0.37
    got = _run_lib(lib, a, b, c5, r, 0, None)
    ref = ao.alt_corr_forward(a, b, c5, r)
    assert torch.isfinite(got).all()      # every element written (output was NaN-filled)
    assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    # rejects what the kernel cannot hold
    assert lib.alt_corr_forward(a.data_ptr(), b.data_ptr(), c5.data_ptr(), got.data_ptr(), 1, N, 6, 9, 6, 9, 80, 6, 0, None) == -2


def test_emulated_avgpool2_nhwc():
    from sim_runtime import hostsim_lib

    lib = hostsim_lib()
    x = torch.randn(2, 7, 10, 12)
    y = torch.empty(2, 3, 5, 12)
    assert lib.avgpool2_nhwc(x.data_ptr(), y.data_ptr(), 2, 7, 10, 12, 0, None) == 0
    ref = torch.nn.functional.avg_pool2d(x.permute(0, 3, 1, 2), 2, stride=2).permute(0, 2, 3, 1)
    assert float((y - ref).abs().max()) <= 1e-6


@pytest.mark.gpu
def test_gpu_alt_corr_forward_and_block():
    from gimmvfi_hip import alt_corr

    f1, f2, coords = _case(B=2, Cc=256, h=32, w=56, seed=5, spread=6.0)
    a = f1.permute(0, 2, 3, 1).contiguous().cuda()
    b = f2.permute(0, 2, 3, 1).contiguous().cuda()
    c5 = coords.permute(0, 2, 3, 1).reshape(2, 1, 32, 56, 2).contiguous().cuda()
    (got,) = alt_corr.forward(a, b, c5, 4)
    ref = ao.alt_corr_forward(a.cpu(), b.cpu(), c5.cpu(), 4)
    assert float((got.cpu() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    blk = alt_corr.AlternateCorrBlock(a, b, num_levels=4, radius=4)
    full = blk(coords.cuda())
    pinned = orc.corr_lookup(orc.corr_pyramid(orc.corr_volume(f1, f2)), coords, 4)
    assert full.shape == pinned.shape
    assert float((full.cpu() - pinned).abs().max()) <= 5e-5 * float(pinned.abs().max())
    with pytest.raises(RuntimeError):
        alt_corr.forward(a.cpu(), b, c5, 4)                      # CHECK_CUDA
    with pytest.raises(RuntimeError):
        alt_corr.forward(a.transpose(1, 2), b, c5, 4)            # CHECK_CONTIGUOUS
