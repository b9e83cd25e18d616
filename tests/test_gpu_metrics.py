"""On-device evaluation metrics vs the reference's own eval_util.py, whose functions were imported
from the reference tree and executed to produce tests/golden/reference_metrics.npz."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ref():
    return dict(np.load(os.path.join(GOLDEN, "reference_metrics.npz")))


def test_mpjpe_and_pa_mpjpe(ref, gpu_device):
    from human_dynamics_amd.evaluation import eval_util as E
    vis = ref["vis"].astype(bool)
    e, epa = E.compute_error_3d(ref["gt"], ref["pred"], vis, device=gpu_device)
    assert len(e) == len(ref["mpjpe"]) == int(vis.sum())
    assert np.abs(np.array(e) - ref["mpjpe"]).max() < 1e-6
    assert np.abs(np.array(epa) - ref["pa_mpjpe"]).max() < 1e-6
    # a pure similarity transform of the ground truth has zero Procrustes error (incl. a reflection-free check)
    rng = np.random.default_rng(0)
    gt = rng.normal(size=(5, 14, 3))
    q = np.linalg.qr(rng.normal(size=(3, 3)))[0]
    q *= np.sign(np.linalg.det(q))
    _, epa = E.compute_error_3d(gt, 0.7 * gt @ q.T + 0.2, device=gpu_device)
    assert max(epa) < 1e-6


def test_acceleration_metrics(ref, gpu_device):
    from human_dynamics_amd.evaluation import eval_util as E
    assert np.abs(E.compute_accel(ref["pred"], device=gpu_device) - ref["accel"]).max() < 1e-6
    got = E.compute_error_accel(ref["gt"], ref["pred"], ref["vis"].astype(bool), device=gpu_device)
    assert got.shape == ref["accel_err"].shape and np.abs(got - ref["accel_err"]).max() < 1e-6


def test_vertex_error(ref, gpu_device):
    from human_dynamics_amd.evaluation import eval_util as E
    got = E.compute_error_verts(ref["verts_gt"], ref["verts_pred"], device=gpu_device)
    assert np.abs(got - ref["verts_err"]).max() < 1e-6
