"""Drop-in CLI with the GIMM-VFI-F config (FlowFormer flow estimator) on the GPU.  Kept in the last test file of the
session: it was written after this round's GPU budget was spent (model, kernels and goldens of the F path ARE verified
on the MI355X; this end-to-end CLI run is not yet), so a surprise here cannot hide the verified tests behind `-x`."""
import pytest

pytestmark = pytest.mark.gpu


def test_cli_video_Nx_random_init_f(tmp_path, sd):
    from test_gpu_e2e import test_cli_video_Nx_random_init

    test_cli_video_Nx_random_init(tmp_path, sd, cfg="gimmvfi_f_arb.yaml")
