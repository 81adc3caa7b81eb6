"""SNU-FILM arbitrary-timestep benchmark -- drop-in for reference src/SNU_FILM_arb.py (same flags):

    python src/SNU_FILM_arb.py -m configs/gimmvfi/gimmvfi_r_arb.yaml -l CKPT -p OUT --eval

For every entry of DATA/test-{medium,hard,extreme}.txt (4x / 8x / 16x interpolation) the frames between the two listed
frames are predicted in one forward and scored by PSNR against the ground-truth frames on disk; predictions are written
to the ``-p`` directory (reference SNU_FILM_arb.py:78-175).  LPIPS is not reported: its AlexNet weights
(reference utils/lpips) are not available offline.  Additions: ``--data-root`` (the reference hard-codes ./data/SNU-FILM
and list paths relative to the CWD), ``--splits``, ``--random-init``, ``--precision``.  MI355X kernels only."""
import argparse
import os
import sys

import numpy as np
import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from models import create_model  # noqa: E402
from utils.setup import single_setup  # noqa: E402
from utils.utils import InputPadder, set_seed  # noqa: E402

STEPS = {"medium": 4, "hard": 8, "extreme": 16}     # SNU_FILM_arb.py:84-90


def default_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("-m", "--model-config", type=str, default="configs/gimmvfi/gimmvfi_r_arb.yaml")
    parser.add_argument("-p", "--pred_save_path", type=str, default="./eval_output/snu_film_arb")
    parser.add_argument("-l", "--load-path", type=str, default="")
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--eval", action="store_true")
    parser.add_argument("--data-root", type=str, default="./data/SNU-FILM")
    parser.add_argument("--splits", type=str, default="medium,hard,extreme")
    parser.add_argument("--random-init", action="store_true")
    parser.add_argument("--precision", type=str, default=None, choices=[None, "bf16", "fp32"])
    return parser


def load_image(path):
    # SNU_FILM_arb.py:46-50
    raw = np.array(Image.open(path).convert("RGB"))
    return (torch.from_numpy(raw.copy()).permute(2, 0, 1) / 255.0).to(torch.float).unsqueeze(0)


def calculate_psnr(a, b):
    return float(-10 * torch.log10(((a - b) * (a - b)).mean()))     # SNU_FILM_arb.py:53-55


def between(path0, k):
    """File name of the k-th frame after path0: the zero-padded stem is incremented (SNU_FILM_arb.py:108-115)."""
    d, base = os.path.split(path0)
    stem, ext = os.path.splitext(base)
    return os.path.join(d, "{:0>{w}}{}".format(int(stem) + k, ext, w=len(stem)))


def evaluate_split(model, root, split, device, save_dir=None):
    T = STEPS[split]
    with open(os.path.join(root, f"test-{split}.txt")) as f:
        entries = [ln.strip().split(" ") for ln in f if ln.strip()]
    psnrs = []
    for name in entries:
        p0, p2 = (os.path.join(root, name[0]), os.path.join(root, name[2]))
        I0, I2 = load_image(p0), load_image(p2)
        gts = [load_image(between(p0, k)).to(device) for k in range(1, T)]
        padder = InputPadder(I0.shape, 32)
        I0p, I2p = padder.pad(I0, I2)
        xs = torch.stack((I0p, I2p), dim=2).to(device)
        b, s_shape = xs.shape[0], xs.shape[-2:]
        coords = [(model.sample_coord_input(b, s_shape, [k / T], device=device), None) for k in range(1, T)]
        ts = [k / T * torch.ones(b, device=device) for k in range(1, T)]
        with torch.no_grad():
            preds = [padder.unpad(im) for im in model(xs, coords, t=ts)["imgt_pred"]]
        for k, (gt, pr) in enumerate(zip(gts, preds), 1):
            psnrs.append(calculate_psnr(gt, pr))
            if save_dir is not None:
                tag = os.path.basename(os.path.dirname(p0)) + "_" + os.path.basename(between(p0, k))
                img = (pr[0].clamp(0, 1).cpu().numpy().transpose(1, 2, 0) * 255.0).astype(np.uint8)
                Image.fromarray(img).save(os.path.join(save_dir, tag))
    return float(np.mean(psnrs)), len(psnrs)


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
        model.load_state_dict(torch.load(args.load_path, map_location="cpu")["state_dict"], strict=False)
    elif args.random_init:
        from gimmvfi_hip.params import random_state_dict_for

        mtype = config.arch["type"] if isinstance(config.arch, dict) else config.arch.type
        model.load_state_dict(random_state_dict_for(mtype, args.seed), strict=True)
    else:
        raise ValueError("--load-path must be specified in evaluation mode")
    model = model.to(device).eval()
    results = {}
    for split in [s for s in args.splits.split(",") if s]:
        psnr, n = evaluate_split(model, args.data_root, split, device, args.pred_save_path)
        results[split] = (psnr, n)
        print(f"[SNU-FILM] [{split}] psnr: {psnr:.02f}, interpolation_step: {STEPS[split]: 02d}  ({n} frames)")
    return results


if __name__ == "__main__":
    main()
