"""X4K1000FPS (XTEST) benchmark -- drop-in for reference src/X4K.py (same flags):

    python src/X4K.py -m configs/gimmvfi/gimmvfi_r_arb.yaml -l CKPT -p OUT --eval

Protocol of the reference (X4K.py:42-63, 87-197): every scene folder of DATA/<type>/<scene>/*.png contributes, per
window of ``t_step_size`` = 32 frames, the 7 in-between targets of 8x interpolation (t = k/8); the set is scored twice:
"XTEST-2k" (all three frames area-resampled to 2048x1080, flow estimated at DS 0.5) and "XTEST-4k" (native size, DS
0.25).  The prediction is un-padded, rounded to uint8 and scored by PSNR against the ground-truth frame; predictions are
written to the ``-p`` directory.  LPIPS is not reported: its AlexNet weights (reference utils/lpips) are not available
offline.  Additions: ``--data-root`` (the reference hard-codes ./data/x4k/test), ``--modes``, ``--multiple``,
``--t-step-size``, ``--size-2k``, ``--random-init``, ``--precision``.  MI355X kernels only."""
import argparse
import glob
import os
import sys

import numpy as np
import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from models import create_model  # noqa: E402
from utils.setup import single_setup  # noqa: E402
from utils.utils import InputPadder, set_seed  # noqa: E402

MODES = {"XTEST-2k": 0.5, "XTEST-4k": 0.25}     # X4K.py:100-128: DS factor per mode


def default_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("-m", "--model-config", type=str, default="configs/gimmvfi/gimmvfi_r_arb.yaml")
    parser.add_argument("-p", "--pred_save_path", type=str, default="./eval_output/x4k")
    parser.add_argument("-l", "--load-path", type=str, default="")
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--eval", action="store_true")
    parser.add_argument("--data-root", type=str, default="./data/x4k/test")
    parser.add_argument("--modes", type=str, default="XTEST-2k,XTEST-4k")
    parser.add_argument("--multiple", type=int, default=8)
    parser.add_argument("--t-step-size", type=int, default=32)
    parser.add_argument("--size-2k", type=str, default="2048x1080", help="WxH of the XTEST-2k resampling")
    parser.add_argument("--random-init", action="store_true")
    parser.add_argument("--precision", type=str, default=None, choices=[None, "bf16", "fp32"])
    return parser


def getXVFI(dir, multiple=8, t_step_size=32):
    """[[I0 path, I1 path, It path, t], ...] over DIR/<type>/<scene>/*.png (X4K.py:42-63): windows start every
    ``t_step_size`` frames, the last frame only closes a window, targets sit t_step_size // multiple apart."""
    samples = []
    t = np.linspace(1 / multiple, 1 - 1 / multiple, multiple - 1)
    for type_folder in sorted(glob.glob(os.path.join(dir, "*", ""))):
        for scene_folder in sorted(glob.glob(os.path.join(type_folder, "*", ""))):
            frames = sorted(glob.glob(scene_folder + "*.png"))
            for idx in range(0, len(frames), t_step_size):
                if idx == len(frames) - 1:
                    break
                for mul in range(multiple - 1):
                    samples.append([frames[idx], frames[idx + t_step_size],
                                    frames[idx + (t_step_size // multiple) * (mul + 1)], float(t[mul])])
    return samples


def load_image(path):
    # X4K.py:90-98: 8-bit RGB in [0, 1], (1,3,H,W)
    raw = np.array(Image.open(path).convert("RGB"))
    return (torch.from_numpy(raw.copy()).permute(2, 0, 1).to(torch.float) * (1.0 / 255.0)).unsqueeze(0)


def area_resize(img, size_wh):
    """cv2.resize(..., dsize=(W, H), interpolation=INTER_AREA) of the reference (X4K.py:101-121) for shrinking: every
    output pixel is the area-weighted mean of the source pixels it covers (the 4K -> 2K case is the 2x2 box mean)."""
    W, H = size_wh
    h, w = img.shape[-2:]
    if (h, w) == (H, W):
        return img
    if h % H == 0 and w % W == 0:
        return torch.nn.functional.avg_pool2d(img, (h // H, w // W))

    def weights(n_in, n_out):
        # row i of the [n_out, n_in] matrix = overlap of output cell i with every input cell / cell size
        s = n_in / n_out
        lo = torch.arange(n_out, dtype=torch.float64)[:, None] * s
        j = torch.arange(n_in, dtype=torch.float64)[None, :]
        return ((torch.minimum(lo + s, j + 1) - torch.maximum(lo, j)).clamp(min=0) / s).to(torch.float32)

    return torch.einsum("Hh,nchw,Ww->ncHW", weights(h, H), img, weights(w, W))


def calculate_psnr(a, b):
    return float(-10 * torch.log10(((a - b) * (a - b)).mean()))     # X4K.py:165-167


def evaluate_mode(model, samples, mode, device, size_2k, save_dir=None):
    ds = MODES[mode]
    psnrs = []
    for p0, p2, pt, t in samples:
        I0, I2, I1 = load_image(p0), load_image(p2), load_image(pt)
        if mode == "XTEST-2k":
            I0, I2, I1 = (area_resize(x, size_2k) for x in (I0, I2, I1))
        padder = InputPadder(I0.shape, 32)
        I0p, I2p = padder.pad(I0, I2)
        xs = torch.stack((I0p, I2p), dim=2).to(device)
        b, s_shape = xs.shape[0], xs.shape[-2:]
        assert t <= 1
        coords = [(model.sample_coord_input(b, s_shape, [t], device=device, upsample_ratio=ds), None)]
        ts = [t * torch.ones(b, device=device, dtype=torch.float)]
        with torch.no_grad():
            pred = padder.unpad(model(xs, coords, t=ts, ds_factor=ds)["imgt_pred"][0])
        # X4K.py:150-163: the prediction is quantised to 8 bits before it is scored
        u8 = (pred[0].float().cpu().numpy().transpose(1, 2, 0) * 255.0).clip(0.0, 255.0).round().astype(np.uint8)
        pred_q = torch.from_numpy(u8.transpose(2, 0, 1)[None].copy()).to(torch.float) / 255.0
        psnrs.append(calculate_psnr(I1, pred_q))
        if save_dir is not None:
            tag = os.path.basename(os.path.dirname(pt)) + "_" + os.path.basename(pt)
            Image.fromarray(u8).save(os.path.join(save_dir, tag))
    return float(np.mean(psnrs)) if psnrs else float("nan"), len(psnrs)


def main(argv=None):
    args, extra = default_parser().parse_known_args(argv)
    set_seed(args.seed)
    config = single_setup(args, extra)
    device = torch.device("cuda")
    os.makedirs(args.pred_save_path, exist_ok=True)
    if args.precision is not None:
        config.arch["precision"] = args.precision
    model, _ = create_model(config.arch)
    if args.load_path != "":
        model.load_state_dict(torch.load(args.load_path, map_location="cpu")["state_dict"], strict=True)
    elif args.random_init:
        from gimmvfi_hip.params import random_state_dict_for

        mtype = config.arch["type"] if isinstance(config.arch, dict) else config.arch.type
        model.load_state_dict(random_state_dict_for(mtype, args.seed), strict=True)
    else:
        raise ValueError("--load-path must be specified in evaluation mode")
    model = model.to(device).eval()
    samples = getXVFI(args.data_root, args.multiple, args.t_step_size)
    size_2k = tuple(int(v) for v in args.size_2k.lower().split("x"))
    results = {}
    for mode in [m for m in args.modes.split(",") if m]:
        psnr, n = evaluate_mode(model, samples, mode, device, size_2k, args.pred_save_path)
        results[mode] = (psnr, n)
        print(f"{mode}  PSNR: {psnr}  ({n} frames)")
    return results


if __name__ == "__main__":
    main()
