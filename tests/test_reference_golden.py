"""Parity against golden vectors produced by EXECUTING THE REFERENCE'S OWN SOURCE
(tests/golden/make_reference_golden.py: src/tf_smpl/*, src/omega.py, Tester.predict_all_images and
src/evaluation/eval_util.py imported from the reference tree and run on a NumPy stand-in for the
elementary TF ops they call).  These fixtures pin (i) the CPU oracle and (ii) the HIP path for the
SMPL / projection / container / sliding-window rows of the hot path."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, Config
from human_dynamics_amd import assets
from human_dynamics_amd import dist as hd
from oracle import hmmr_oracle as O

VSUB = 8
F64 = torch.float64


@pytest.fixture(scope="module")
def ref():
    return dict(np.load(os.path.join(GOLDEN, "reference_smpl.npz")))


@pytest.fixture(scope="module")
def ref_windows():
    return dict(np.load(os.path.join(GOLDEN, "reference_windows.npz")))


# ------------------------------------------------------------------ the oracle vs the reference
def test_oracle_smpl_equals_reference_code(ref, smpl_consts):
    v, j, R = O.smpl_forward(ref["beta"], ref["theta"], smpl_consts, F64)
    k = O.batch_orth_proj_idrot(j, torch.tensor(ref["cams"]))
    assert np.abs(v.numpy()[:, ::VSUB] - ref["verts"]).max() < 1e-12
    assert np.abs(j.numpy() - ref["joints"]).max() < 1e-12
    assert np.abs(R.numpy() - ref["Rs"]).max() < 1e-13
    assert np.abs(k.numpy() - ref["kps"]).max() < 1e-12


def test_oracle_containers_equal_reference_omegaspred(ref, weights, smpl_consts):
    """OmegasPred as driven by build_test_model: delta containers project with omega_0's camera,
    keep [1,0,0] in their raw omega (tester.py:208-213, omega.py:263-304)."""
    t = O.OracleTester(weights, smpl_consts, dtype=F64)
    om0 = torch.tensor(ref["omg_omega0"]).reshape(-1, 85)
    main = t.smpl_outputs(om0, om0[:, :3])
    for k in ("cams", "joints", "kps", "poses", "shapes", "omegas"):
        assert np.abs(main[k].numpy().reshape(ref["omg_" + k].shape) - ref["omg_" + k]).max() < 1e-12, k
    assert np.abs(main["verts"].numpy()[:, ::VSUB].reshape(ref["omg_verts"].shape) - ref["omg_verts"]).max() < 1e-12
    for tag, key in (("-5", "omg_delta_m5"), ("+5", "omg_delta_p5")):
        d = t.smpl_outputs(torch.tensor(ref[key]).reshape(-1, 85), om0[:, :3])
        for k in ("cams", "joints", "kps", "poses", "shapes", "omegas"):
            g = ref["omg_%s_delta_%s" % (k, tag)]
            assert np.abs(d[k].numpy().reshape(g.shape) - g).max() < 1e-12, (k, tag)


def test_window_logic_equals_reference_predict_all_images(ref_windows):
    """What the reference feeds the network (every window slot) and what it keeps, for several
    (N, B): ShardPlan / window_plan reproduce both, on one rank and split over ranks."""
    for key in [k for k in ref_windows if k.startswith("fed_")]:
        n, B = [int(x[1:]) for x in key.split("_")[1:]]
        fed = ref_windows[key].astype(np.int64)                  # [count*B, T], 1-based ids, -1 = zero image
        kept = ref_windows["kept_n%d_b%d" % (n, B)]
        assert np.array_equal(kept, np.arange(1, n + 1))         # every frame exactly once, in order
        p = hd.ShardPlan(n, B, 20, 13, 1, 0)
        idx = p.window_frame_index()
        assert np.array_equal(np.where(idx >= 0, idx + 1, -1), fed)
        for world in (2, 3):
            parts = []
            for r in range(world):
                q = hd.ShardPlan(n, B, 20, 13, world, r)
                ix = q.window_frame_index()
                parts.append(np.where(ix >= 0, ix + q.f0 + 1, -1))
            assert np.array_equal(np.concatenate(parts, 0), fed)


def test_oracle_predict_all_images_uses_the_reference_windows(ref_windows, weights, smpl_consts):
    from human_dynamics_amd.evaluation.tester import window_plan
    for key in [k for k in ref_windows if k.startswith("fed_")]:
        n, B = [int(x[1:]) for x in key.split("_")[1:]]
        margin, g, count, num_fill = window_plan(n, B, 20, 13)
        assert (margin, g) == (6, 8) and count * B == ref_windows[key].shape[0]
        assert O.window_plan(n, B, 20, 13) == (margin, g, count, num_fill)


# ------------------------------------------------------------------ the HIP path vs the reference
@pytest.mark.gpu
def test_hip_smpl_equals_reference_code(ref, smpl_consts, gpu_device):
    """BASELINE metric 'SMPL verts max-abs-err' against the reference's own SMPL source: <= 1e-4."""
    from human_dynamics_amd.engine import HmmrEngine
    eng = HmmrEngine(None, smpl_consts, device=gpu_device)
    verts, joints, kps, rs = eng.smpl(ref["theta"].astype(np.float32), ref["beta"].astype(np.float32),
                                      ref["cams"].astype(np.float32))
    errs = {"verts": np.abs(verts.cpu().numpy()[:, ::VSUB] - ref["verts"]).max(),
            "joints": np.abs(joints.cpu().numpy() - ref["joints"]).max(),
            "kps": np.abs(kps.cpu().numpy() - ref["kps"]).max(),
            "Rs": np.abs(rs.cpu().numpy() - ref["Rs"]).max()}
    print("HIP SMPL vs reference source (float64):", {k: "%.2e" % v for k, v in errs.items()})
    assert all(e < 1e-4 for e in errs.values()), errs
    assert errs["verts"] < 1e-5


@pytest.mark.gpu
def test_hip_containers_equal_reference_omegaspred(ref, weights, smpl_consts, gpu_device):
    from human_dynamics_amd.engine import HmmrEngine
    from human_dynamics_amd.omega import OmegasPred
    eng = HmmrEngine(None, smpl_consts, device=gpu_device)
    B, T = ref["omg_omega0"].shape[:2]
    cfg = Config(batch_size=B)
    reg = []
    preds = {0: OmegasPred(cfg, eng, use_optcam=False, vis_max_batch=B, registry=reg)}
    for dt in (-5, 5):
        preds[dt] = OmegasPred(cfg, eng, use_optcam=True, vis_max_batch=B, registry=reg)
    preds[0].append_batched(eng.to_device(ref["omg_omega0"]))
    for dt, key in ((-5, "omg_delta_m5"), (5, "omg_delta_p5")):
        preds[dt].append_batched(eng.to_device(ref[key]))
        preds[dt].set_cams(preds[0].get_cams())
    OmegasPred.compute_all_smpl(reg)
    from human_dynamics_amd.evaluation.tester import Tester
    for dt in (0, -5, 5):
        got = Tester.make_fetch_dict(preds[dt], suffix="_delta" if dt else "")
        for k, v in got.items():
            g = ref["omg_%s%s" % (k, ("_%+d" % dt) if dt else "")]
            a = v.float().cpu().numpy()
            if "verts" in k:
                a = a[..., ::VSUB, :]
            assert a.shape == g.shape, (k, a.shape, g.shape)
            assert np.abs(a - g).max() < 1e-4, (k, dt, np.abs(a - g).max())


# ------------------------------------------------------------------ f_movie + IEF wiring (src/models.py)
@pytest.fixture(scope="module")
def ref_ti():
    return dict(np.load(os.path.join(GOLDEN, "reference_temporal_ief.npz")))


def test_checkpoint_variable_names_are_the_ones_the_reference_asks_for(ref_ti, weights):
    """The reference's own variable scopes (models.py run under the shim) resolve inside the
    weight dict contract of assets.py / SURVEY App. B."""
    used = [str(u) for u in ref_ti["used_variables"]]
    assert len(used) == 42 and all(u in weights for u in used)
    assert "AZ_FC_block_preact_gn1block_0/gamma" in used and "AZ_FC_block2_conv2block_2/weights" in used
    assert "single_view_ief/3D_module/fc1/weights" in used
    assert "single_view_ief_past5/3D_module/fc3/biases" in used and "single_view_ief_future5/3D_module/fc2/weights" in used


def test_oracle_temporal_and_ief_equal_reference_wiring(ref_ti, weights):
    strips = O.az_fc2_groupnorm(ref_ti["phi"], weights, 3, F64)
    assert np.abs(strips.numpy() - ref_ti["strips"]).max() < 1e-9
    mean = torch.tensor(weights["mean_param"], dtype=F64).reshape(1, 85).expand(40, 85)
    om, deltas = O.call_hmr_ief(torch.tensor(ref_ti["strips"]).reshape(40, -1), mean, weights, (-5, 5), F64)
    assert np.abs(om.numpy().reshape(2, 20, 85) - ref_ti["omega"]).max() < 1e-9
    assert np.abs(deltas[-5].numpy().reshape(2, 20, 85) - ref_ti["delta_m5"]).max() < 1e-9
    assert np.abs(deltas[5].numpy().reshape(2, 20, 85) - ref_ti["delta_p5"]).max() < 1e-9


@pytest.mark.gpu
def test_hip_temporal_and_ief_equal_reference_wiring(ref_ti, weights, smpl_consts, gpu_device):
    from human_dynamics_amd.engine import HmmrEngine
    eng = HmmrEngine(weights, smpl_consts, dtype="f32", device=gpu_device)
    strips = eng.temporal(ref_ti["phi"].astype(np.float32)).cpu().numpy()
    e1 = np.abs(strips - ref_ti["strips"]).max()
    om = eng.ief(ref_ti["strips"].astype(np.float32).reshape(40, -1)).cpu().numpy().reshape(3, 2, 20, 85)
    e2 = max(np.abs(om[0] - ref_ti["omega"]).max(), np.abs(om[1] - ref_ti["delta_m5"]).max(),
             np.abs(om[2] - ref_ti["delta_p5"]).max())
    print("HIP (fp32 operands) vs reference models.py wiring: strips %.2e  omegas %.2e" % (e1, e2))
    assert e1 < 1e-4 and e2 < 1e-4


# ------------------------------------------------------------------ image encoder (src/models.py:50-77) + fc2_res
# tests/golden/make_resnet_golden.py EXECUTES the reference's encoder_resnet on a transcription of slim's
# resnet_v2.py / resnet_utils.py (oracle/slim_resnet_v2.py) and the shim's slim layers: NumPy arithmetic that
# shares no code with the oracle's PyTorch restatement.
@pytest.fixture(scope="module")
def ref_resnet():
    return dict(np.load(os.path.join(GOLDEN, "reference_resnet.npz")))


def _resnet_frames():
    frames = assets.make_synthetic_frames(2, seed=1)
    frames[1] = 0.0                        # the zero padding image of predict_all_images
    return frames


def _sample(a):
    sh, sc = max(1, a.shape[1] // 8), max(1, a.shape[3] // 32)
    return a[:, ::sh, ::sh, ::sc]


def test_resnet_variable_names_are_the_ones_slim_asks_for(ref_resnet, weights):
    used = [str(u) for u in ref_resnet["used_variables"]]
    assert len(used) == 270 and all(u in weights for u in used)
    assert len([k for k in weights if k.startswith("resnet_v2_50/")]) == 270      # and nothing is left over
    assert "resnet_v2_50/block3/unit_6/bottleneck_v2/conv2/BatchNorm/moving_variance" in used
    assert "resnet_v2_50/block4/unit_1/bottleneck_v2/shortcut/biases" in used and "resnet_v2_50/postnorm/gamma" in used


def test_oracle_resnet_equals_reference_encoder_resnet(ref_resnet, weights):
    """phi and every bottleneck unit's output: the PyTorch restatement against the reference's
    encoder_resnet executed on the slim transcription (float64 both)."""
    phi, ep = O.resnet_v2_50(_resnet_frames(), weights, F64, return_endpoints=True)
    assert np.abs(phi.numpy() - ref_resnet["phi"]).max() < 1e-12
    names = [str(n) for n in ref_resnet["end_points"]]
    assert len(names) == 73
    checked = 0
    for key, t in ep.items():
        alias = {"conv1": "resnet_v2_50/conv1"}.get(key, "resnet_v2_50/%s/bottleneck_v2" % key)
        if alias not in names:
            continue                        # pool1 is not one of slim's collected end points
        got = _sample(t.permute(0, 2, 3, 1).numpy())
        assert got.shape == ref_resnet["ep:" + alias].shape, alias
        assert np.abs(got - ref_resnet["ep:" + alias]).max() < 1e-11, alias
        checked += 1
    assert checked == 17                    # conv1 + 16 units


def test_oracle_fc2_res_equals_reference(weights):
    g = dict(np.load(os.path.join(GOLDEN, "reference_fc2_res.npz")))
    wh = assets.make_synthetic_weights(0, with_hallucinator=True)
    assert sorted(str(u) for u in g["used_variables"]) == sorted(k for k in wh if k.startswith("fc2_res/"))
    assert np.abs(O.fc2_res(g["phi"], wh, F64).numpy() - g["out"]).max() < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("dt,tol", [("f32", 2e-5), ("f16x3", 2e-4)])
def test_hip_resnet_equals_reference_encoder_resnet(ref_resnet, weights, gpu_device, dt, tol):
    from human_dynamics_amd.engine import HmmrEngine
    eng = HmmrEngine(weights, None, dtype=dt, device=gpu_device)
    phi = eng.resnet(_resnet_frames()).cpu().numpy()
    err = np.abs(phi - ref_resnet["phi"]).max()
    rel = np.linalg.norm(phi - ref_resnet["phi"]) / np.linalg.norm(ref_resnet["phi"])
    print("HIP ResNet (%s) vs reference encoder_resnet: max-abs %.2e rel-L2 %.2e" % (dt, err, rel))
    assert err < tol
    phi_z = eng.resnet(_resnet_frames()[:1], n_zero=1).cpu().numpy()          # the n_zero tail == an explicit zero image
    assert np.array_equal(phi_z, phi)


@pytest.mark.gpu
def test_hip_hallucinator_equals_reference_fc2_res(gpu_device):
    from human_dynamics_amd.engine import HmmrEngine
    g = dict(np.load(os.path.join(GOLDEN, "reference_fc2_res.npz")))
    wh = assets.make_synthetic_weights(0, with_hallucinator=True)
    for dt, tol in (("f32", 2e-5), ("f16x3", 1e-4)):
        eng = HmmrEngine(wh, None, dtype=dt, device=gpu_device)
        out = eng.hallucinate(g["phi"].astype(np.float32)).cpu().numpy()
        err = np.abs(out - g["out"]).max()
        print("HIP fc2_res (%s) vs reference: %.2e" % (dt, err))
        assert err < tol
