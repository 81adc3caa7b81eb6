"""Drop-in for reference src/models/__init__.py:15-37: ``create_model(config.arch) -> (model, model_ema)``."""
import os
import sys

_PKG = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from .generalizable_INR import gimm, gimmvfi_f, gimmvfi_r  # noqa: E402


def create_model(config, ema=False):
    model_type = (config["type"] if isinstance(config, dict) else config.type).lower()
    if model_type == "gimmvfi_r":
        model = gimmvfi_r(config)
    elif model_type == "gimm":
        model = gimm(config)
    elif model_type == "gimmvfi_f":
        model = gimmvfi_f(config)
    else:
        raise ValueError(f"{model_type} is invalid..")
    if ema:
        raise NotImplementedError("EMA is a training feature (reference src/models/ema.py); inference only here")
    return model, None
