"""Drop-in for reference src/models/generalizable_INR/__init__.py."""
from gimmvfi_hip.model import GIMMVFI_R


def gimmvfi_r(config):
    return GIMMVFI_R(config)
