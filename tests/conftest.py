import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def weights():
    from human_dynamics_amd import assets
    return assets.make_synthetic_weights(0)


@pytest.fixture(scope="session")
def smpl_consts():
    from human_dynamics_amd import assets
    return assets.make_synthetic_smpl(2)


@pytest.fixture(scope="session")
def golden_window():
    return dict(np.load(os.path.join(GOLDEN, "window_b1_t20.npz")))


@pytest.fixture(scope="session")
def golden_video():
    return dict(np.load(os.path.join(GOLDEN, "video_n24_b2_t20.npz")))


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    return "cuda:0"


class Config(object):
    """Duck-typed stand-in for the reference's absl flags object (src/config.py)."""
    def __init__(self, **kw):
        self.load_path = "synthetic:0"
        self.batch_size = 8
        self.sequence_length = 20
        self.pred_mode = "pred"
        self.num_conv_layers = 3
        self.delta_t_values = ["-5", "5"]
        self.smpl_model_path = "synthetic:2"
        self.num_kps = 25
        self.__dict__.update(kw)
