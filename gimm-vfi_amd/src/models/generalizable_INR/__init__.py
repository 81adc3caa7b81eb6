"""Drop-in for reference src/models/generalizable_INR/__init__.py."""
from gimmvfi_hip.model import GIMM, GIMMVFI_F, GIMMVFI_R


def gimmvfi_r(config):
    return GIMMVFI_R(config)


def gimmvfi_f(config):
    return GIMMVFI_F(config)


def gimm(config):
    return GIMM(config)
