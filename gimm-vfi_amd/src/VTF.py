"""Vimeo-Triplet-Flow benchmark of the motion-only model GIMM -- drop-in for reference src/VTF.py (same flags):

    python src/VTF.py -m configs/gimm/gimm.yaml -l CKPT --eval

For every triplet of DATA/tri_testlist.txt the bidirectional flows im1<->im3 (DATA/flow_sequences/<name>/*.flo) go
through GIMM at t = 0.5 and are scored against the flows of the middle frame: PSNR on normalised flows and end-point
error in pixels (reference VTF.py:64-159).  Additions: ``--data-root`` (the reference hard-codes
data/vimeo90k/vimeo_triplet), ``--random-init`` (seeded weights when no checkpoint is at hand), ``--precision``.
Runs on the MI355X kernels through the model API; there is no CPU path."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from models import create_model  # noqa: E402
from utils.frame_utils import readFlow  # noqa: E402
from utils.setup import single_setup  # noqa: E402
from utils.utils import set_seed  # noqa: E402


def default_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("-m", "--model-config", type=str, default="configs/gimm/gimm.yaml")
    parser.add_argument("-l", "--load-path", type=str, default="")
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--eval", action="store_true")
    parser.add_argument("--data-root", type=str, default="data/vimeo90k/vimeo_triplet")
    parser.add_argument("--random-init", action="store_true")
    parser.add_argument("--precision", type=str, default=None, choices=[None, "bf16", "fp32"])
    return parser


def process_flow(path):
    # VTF.py:35-38: (H,W,2) -> (1,2,H,W) float
    return torch.from_numpy(readFlow(path).copy()).permute(2, 0, 1).unsqueeze(0).to(torch.float32)


def grid_coords(h, w, device, t):
    """(1,1,h,w,3) coordinate grid at normalised time t (a slice of the reference's `xytshape2coordinate` grid over
    [0,1], VTF.py:92-121 / VSF.py:98-127), spatial axes at pixel centres in [-1,1]."""
    ys = -1.0 + 2.0 * (0.5 + torch.arange(h, device=device)) / h
    xs = -1.0 + 2.0 * (0.5 + torch.arange(w, device=device)) / w
    g = torch.stack(torch.meshgrid(torch.tensor([float(t)], device=device), ys, xs, indexing="ij"), dim=-1)
    return g.unsqueeze(0)


def mid_coords(h, w, device):
    return grid_coords(h, w, device, 0.5)


# protocol: (list file, first frame, last frame, [(middle frame, coordinate time, timestep)])
TRIPLET = ("tri_testlist.txt", 1, 3, [(2, 0.5, 0.5)])                                   # VTF.py:61-139
# the septuplet driver pairs the grid slice (t_id-1)/6 with the time step t_id/6 (VSF.py:128-148): kept as is
SEPTUPLET = ("sep_testlist.txt", 1, 7, [(k, (k - 1) / 6, k / 6) for k in range(2, 7)])   # VSF.py:61-148


def evaluate(model, data_root, device, limit=None, protocol=TRIPLET):
    listfile, a, b, mids = protocol
    with open(os.path.join(data_root, listfile)) as f:
        names = [ln for ln in f.read().splitlines() if ln.strip()]
    if limit:
        names = names[:limit]
    psnrs, epes = [], []
    for name in names:
        d = os.path.join(data_root, "flow_sequences", name)
        f01 = process_flow(os.path.join(d, f"im{a}_im{b}.flo")).unsqueeze(2)
        f10 = process_flow(os.path.join(d, f"im{b}_im{a}.flo")).unsqueeze(2)
        xs = torch.cat((f01, -f10), dim=2).to(device)                       # VTF.py:90
        scaler = xs.abs().max().reshape(1, 1)                                # VTF.py:124-129
        ori = torch.cat((xs[:, :, :1], -xs[:, :, 1:2]), dim=2)              # VTF.py:137
        xn = (xs / scaler + 1.0) / 2.0
        for k, tc, ts in mids:
            gt = (process_flow(os.path.join(d, f"im{k}_im{b}.flo")) - process_flow(os.path.join(d, f"im{k}_im{a}.flo"))).unsqueeze(2)
            with torch.no_grad():
                out = model(xn, grid_coords(xs.shape[3], xs.shape[4], device, tc), ori_flow=ori,
                            timesteps=torch.tensor([ts], device=device, dtype=torch.float32))
                target = (gt.to(device) / scaler + 1.0) / 2.0
                psnrs.append(float(model.compute_loss(out, target, reduction="sum")["psnr"]))
                flow = (out * 2.0 - 1.0) * scaler                            # un-normalise, VTF.py:151-157
                epes.append(float(((flow[0, :, 0] - gt[0, :, 0].to(device)) ** 2).sum(0).sqrt().mean()))
    return float(np.mean(psnrs)), float(np.mean(epes)), len(psnrs)


def main(argv=None, protocol=TRIPLET, default_root="data/vimeo90k/vimeo_triplet"):
    args, extra = default_parser().parse_known_args(argv)
    if args.data_root == "data/vimeo90k/vimeo_triplet":
        args.data_root = default_root
    set_seed(args.seed)
    config = single_setup(args, extra)
    device = torch.device("cuda")
    if args.precision is not None:
        config.arch["precision"] = args.precision
    model, _ = create_model(config.arch)
    if args.load_path != "":
        model.load_state_dict(torch.load(args.load_path, map_location="cpu")["state_dict"], strict=False)
    elif args.random_init:
        from gimmvfi_hip.params import gimm_state_dict, random_state_dict

        model.load_state_dict(gimm_state_dict(random_state_dict(args.seed)), strict=True)
    else:
        raise ValueError("--load-path must be specified in evaluation mode")
    model = model.to(device).eval()
    psnr, epe, n = evaluate(model, args.data_root, device, protocol=protocol)
    print("Avg PSNR: {} EPE: {}".format(psnr, epe))
    return psnr, epe, n


if __name__ == "__main__":
    main()
