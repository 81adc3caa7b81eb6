"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's motion-only model `GIMM`
(src/models/generalizable_INR/gimm.py:129-214), SURVEY.md 8f row 3.

It reuses the stage functions of gimmvfi_r_oracle.py (splat metric, cnn_encoder, softmax splat, res_conv, hypo-network:
the GIMM class defines the same blocks, gimm.py:36-78 vs gimmvfi_r.py:84-124) and restates only the orchestration.
Pinned bit-exact against the real reference class on CPU by tests/test_gimm_model.py (reference present) and through
tests/golden/gimm_*.npz (everywhere).
"""
import torch

import gimmvfi_r_oracle as orc


def forward(sd, xs, coord, ori_flow, timesteps, keep_xs_shape=True):
    """gimm.py:129-214.  xs: normalised flows (B,2,2,H,W) [channel, frame]; ori_flow: raw flows (B,2,2,H,W);
    coord / timesteps: either one tensor each or equally long lists.  Returns (list of) (B,2,1,H',W')."""
    f01, f10 = ori_flow[:, :, 0], ori_flow[:, :, 1]                      # :133-134
    w1, w2 = orc.cal_splatting_weights(sd, f01, f10)                      # :137
    pl0 = orc.cnn_encoder(sd, xs[:, :, 0])                                # :139-140
    pl1 = orc.cnn_encoder(sd, xs[:, :, 1])

    def latent(cur_t):
        cur_t = cur_t.reshape(-1, 1, 1, 1)
        s0 = orc.softsplat_linear_zeroeps(pl0, f01 * cur_t, w1)           # :149-160
        s1 = orc.softsplat_linear_zeroeps(pl1, f10 * (1 - cur_t), w2)
        lat = torch.cat([s0, s1], 1)
        lat = lat + orc.res_conv(sd, torch.cat([pl0, pl1, lat], 1))       # :161-166
        return lat.permute(0, 2, 3, 1)

    def inr(c, lat):
        out = orc.hyponet_forward(sd, c, lat)                             # :170-179 / :201-210  (B,1,H,W,2)
        return out.permute(0, 4, 1, 2, 3) if keep_xs_shape else out

    if isinstance(timesteps, list):
        assert isinstance(coord, list) and len(coord) == len(timesteps)
        return [inr(c, latent(t)) for c, t in zip(coord, timesteps)]
    return inr(coord, latent(timesteps))


GIMM_KEY_PREFIXES = ("cnn_encoder.", "res_conv.", "hyponet.", "g_filter", "alpha_v", "alpha_fe")


def gimm_state_dict(sd_full):
    """The GIMM subset of a GIMM-VFI-R state_dict (same block definitions and names)."""
    return {k: v for k, v in sd_full.items() if k.startswith(GIMM_KEY_PREFIXES)}
