import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "diffusion-model_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def weights_dir():
    return os.path.join(ROOT, "weights")


@pytest.fixture(scope="session")
def real_spec(weights_dir):
    from isingdiff.keras_file import load_unet_spec
    return load_unet_spec(os.path.join(weights_dir, "b_diffusion_model_2d_ising_realspace_unet.keras"))


@pytest.fixture(scope="session")
def fft_spec(weights_dir):
    from isingdiff.keras_file import load_unet_spec
    return load_unet_spec(os.path.join(weights_dir, "b_diffusion_model_2d_ising_unet.keras"))


@pytest.fixture(scope="session")
def ref_real(real_spec):
    from oracle import ref_cpu
    return ref_cpu.RefModel(real_spec)


@pytest.fixture(scope="session")
def gpu_real(real_spec):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible")
    from isingdiff.model import ScoreModel
    return ScoreModel(real_spec, device=0, precision="fp32")
