import os
import sys
import warnings

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "gimm-vfi_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"),
          os.path.join(ROOT, "tests", "hostsim")):
    if p not in sys.path:
        sys.path.insert(0, p)
warnings.filterwarnings("ignore")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`gpu` tests need an MI355X: on a box without one a plain `pytest` run skips them instead of failing."""
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible (gpu-marked tests run on the MI355X box)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def sd():
    from gimmvfi_hip.params import random_state_dict

    return random_state_dict(0)


@pytest.fixture(scope="session")
def simlib():
    from sim_runtime import hostsim_lib

    return hostsim_lib()


@pytest.fixture(scope="session")
def sd_f():
    from gimmvfi_hip.params import random_state_dict_f

    return random_state_dict_f(0)
