"""Drop-in for reference src/utils/setup.py:105-110 + src/utils/config.py:26-44,108-117 (eval path only):
yaml -> attribute dict with the GIMMVFIConfig defaults merged in (generalizable_INR/configs.py:38-57).
PyYAML only; the reference's omegaconf/easydict dependencies are not needed."""
from pathlib import Path

import yaml


class Config(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def copy(self):
        return _wrap(dict(self))


def _wrap(o):
    if isinstance(o, dict):
        return Config({k: _wrap(v) for k, v in o.items()})
    if isinstance(o, list):
        return [_wrap(v) for v in o]
    return o


def _merge(defaults, over):
    out = dict(defaults)
    for k, v in over.items():
        out[k] = _merge(out[k], v) if isinstance(v, dict) and isinstance(out.get(k), dict) else v
    return out


ARCH_DEFAULTS = {
    "type": "gimmvfi", "ema": False, "ema_value": None, "fwarp_type": "linear", "rec_weight": 0.1, "raft_iter": 20,
    "modulated_layer_idxs": None,
    "hyponet": {
        "type": "mlp", "n_layer": 5, "use_bias": True, "input_dim": 2, "output_dim": 3, "output_bias": 0.5,
        "activation": {"type": "relu", "siren_w0": 30.0},
        "initialization": {"weight_init_type": "kaiming_uniform", "bias_init_type": "zero"},
        "normalize_weight": True, "linear_interpo": False,
    },
}


def load_config(config_path):
    with open(config_path) as f:
        cfg = yaml.safe_load(f)
    cfg["arch"] = _merge(ARCH_DEFAULTS, cfg.get("arch", {}))
    return _wrap(cfg)


def single_setup(args, extra_args=(), train=True):
    assert args.eval
    args.model_config = Path(args.model_config).absolute().resolve().as_posix()
    config = load_config(args.model_config)
    if "seed" not in config:
        config["seed"] = args.seed
    return config
