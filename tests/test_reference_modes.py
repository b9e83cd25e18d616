"""The NON-Tester configurations of the path against the reference's own source executed (tests/golden/
make_modes_golden.py): batch_pred_omega / call_hmr_ief with every (use_optcam, use_delta_from_pred) combination and a
per-row omega_mean (src/models.py:233-267, 299-377), batch_global_rigid_transformation(rotate_base=True)
(src/tf_smpl/batch_lbs.py:151-158).  CPU: the oracle equals the reference; GPU: the HIP path equals the reference."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from human_dynamics_amd import assets
from oracle import hmmr_oracle as O

F64 = torch.float64
COMBOS = [(True, True), (True, False), (False, True), (False, False)]


@pytest.fixture(scope="module")
def ref():
    return dict(np.load(os.path.join(GOLDEN, "reference_modes.npz")))


@pytest.mark.parametrize("optcam,from_pred", COMBOS)
def test_oracle_ief_modes_equal_reference(ref, optcam, from_pred):
    w = assets.make_synthetic_ief_weights(7, delta_nd=72 if optcam else 75)
    phi = torch.tensor(ref["strips"]).reshape(-1, 2048)
    omega, deltas = O.call_hmr_ief(phi, torch.tensor(ref["omega_mean"]), w, dtype=F64, use_optcam=optcam,
                                   use_delta_from_pred=from_pred)
    tag = "optcam%d_frompred%d" % (optcam, from_pred)
    assert np.abs(omega.numpy().reshape(2, 3, 85) - ref["omega_" + tag]).max() < 1e-12
    for dt, key in ((-5, "delta_m5_"), (5, "delta_p5_")):
        assert np.abs(deltas[dt].numpy().reshape(2, 3, 85) - ref[key + tag]).max() < 1e-12, (dt, tag)
    if optcam:      # the fixed camera and the beta of the delta's starting omega (models.py:367-371)
        assert np.array_equal(ref["delta_m5_" + tag][..., :3], np.broadcast_to([1.0, 0.0, 0.0], (2, 3, 3)))
    src = ref["omega_" + tag] if from_pred else ref["omega_mean"].reshape(2, 3, 85)
    assert np.array_equal(ref["delta_p5_" + tag][..., 75:], src[..., 75:])


def test_oracle_rotate_base_equals_reference(ref):
    for rb in (0, 1):
        nj, A = O.batch_global_rigid_transformation(torch.tensor(ref["fk_Rs"]), torch.tensor(ref["fk_Js"]),
                                                    [int(p) for p in assets.SMPL_PARENTS], rotate_base=bool(rb))
        assert np.abs(nj.numpy() - ref["fk_new_j_rb%d" % rb]).max() < 1e-13
        assert np.abs(A.numpy() - ref["fk_A_rb%d" % rb]).max() < 1e-13
    assert np.abs(ref["fk_A_rb1"] - ref["fk_A_rb0"]).max() > 0.1        # the flag does something


def test_pack_ief_reads_use_optcam_from_the_checkpoint_shapes():
    from human_dynamics_amd import _lib, packing
    for nd, flag in ((72, 0), (75, 1)):
        iw, keys = packing.pack_ief(assets.make_synthetic_ief_weights(1, delta_nd=nd), _lib.HMMR_F32, packing.DeviceStore("cpu"))
        assert keys == [0, -5, 5] and iw.no_optcam == flag and [iw.reg[r].nd for r in range(3)] == [85, nd, nd]
    bad = assets.make_synthetic_ief_weights(1, delta_nd=72)
    bad["single_view_ief_past5/3D_module/fc1/weights"] = np.zeros((2048 + 70, 1024), np.float32)
    with pytest.raises(ValueError):
        packing.pack_ief(bad, _lib.HMMR_F32, packing.DeviceStore("cpu"))


# ------------------------------------------------------------------------------------------------ the HIP path
@pytest.mark.gpu
@pytest.mark.parametrize("dt,tol", [("f32", 2e-6), ("f16x3", 5e-5)])
@pytest.mark.parametrize("optcam,from_pred", COMBOS)
def test_hip_batch_pred_omega_modes_equal_reference(ref, gpu_device, optcam, from_pred, dt, tol):
    from human_dynamics_amd.engine import HmmrEngine
    from human_dynamics_amd.models import batch_pred_omega
    w = assets.make_synthetic_ief_weights(7, delta_nd=72 if optcam else 75)
    eng = HmmrEngine(w, None, dtype=dt, device=gpu_device)
    assert eng.use_optcam == optcam
    strips = torch.tensor(ref["strips"], dtype=torch.float32, device=gpu_device)
    omega, deltas = batch_pred_omega(input_features=strips, batch_size=2, sequence_length=3, num_output=85,
                                     is_training=False, omega_mean=ref["omega_mean"].astype(np.float32),
                                     scope="single_view_ief", engine=eng, predict_delta_keys=[0, -5, 5],
                                     use_optcam=optcam, use_delta_from_pred=from_pred)
    tag = "optcam%d_frompred%d" % (optcam, from_pred)
    assert np.abs(omega.cpu().numpy() - ref["omega_" + tag]).max() < tol
    for k, key in ((-5, "delta_m5_"), (5, "delta_p5_")):
        got = deltas[k].cpu().numpy()
        assert np.abs(got - ref[key + tag]).max() < tol, (k, tag)
        if optcam:
            assert np.array_equal(got[..., :3], np.broadcast_to(np.float32([1, 0, 0]), (2, 3, 3)))
    with pytest.raises(ValueError):                      # the argument is checked against the checkpoint's regressor widths
        batch_pred_omega(input_features=strips, batch_size=2, sequence_length=3, num_output=85, is_training=False,
                         omega_mean=None, scope="single_view_ief", engine=eng, predict_delta_keys=[0, -5, 5],
                         use_optcam=not optcam, use_delta_from_pred=from_pred)


@pytest.mark.gpu
def test_hip_rotate_base_equals_reference(ref, gpu_device):
    from human_dynamics_amd.tf_smpl.batch_lbs import batch_global_rigid_transformation
    for rb in (0, 1):
        nj, A = batch_global_rigid_transformation(ref["fk_Rs"].astype(np.float32), ref["fk_Js"].astype(np.float32),
                                                  assets.SMPL_PARENTS, rotate_base=bool(rb))
        assert np.abs(nj.cpu().numpy() - ref["fk_new_j_rb%d" % rb]).max() < 2e-6
        assert np.abs(A.cpu().numpy() - ref["fk_A_rb%d" % rb]).max() < 2e-6
