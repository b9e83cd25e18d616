"""Pins the CPU oracle: algebraic known answers + the committed golden vectors
(the reference has no tests or fixtures of its own)."""
import numpy as np
import torch

from human_dynamics_amd import assets
from oracle import hmmr_oracle as O

F64 = torch.float64


def test_rodrigues_zero_is_identity():
    R = O.batch_rodrigues(torch.zeros(5, 3, dtype=F64))
    assert np.allclose(R.numpy(), np.eye(3)[None], atol=1e-7)


def test_rodrigues_matches_matrix_exponential():
    rng = np.random.default_rng(0)
    th = torch.tensor(rng.normal(size=(16, 3)))
    R = O.batch_rodrigues(th)
    ref = torch.linalg.matrix_exp(O.batch_skew(th))
    assert np.allclose(R.numpy(), ref.numpy(), atol=1e-7)     # the 1e-8 epsilon perturbs at ~1e-8
    eye = (R @ R.transpose(1, 2)).numpy()
    assert np.allclose(eye, np.eye(3)[None], atol=1e-7)


def test_smpl_zero_pose_is_shaped_template(smpl_consts):
    rng = np.random.default_rng(1)
    beta = rng.normal(size=(3, 10))
    verts, joints, Rs = O.smpl_forward(beta, np.zeros((3, 72)), smpl_consts, F64)
    v_shaped = (beta @ smpl_consts["shapedirs"].astype(np.float64)).reshape(3, -1, 3) + smpl_consts["v_template"]
    assert np.allclose(verts.numpy(), v_shaped, atol=1e-6)
    kj = np.einsum("nvc,vk->nkc", v_shaped, smpl_consts["cocoplus_regressor"].astype(np.float64))
    assert np.allclose(joints.numpy(), kj, atol=1e-6)


def test_smpl_root_rotation_rotates_about_root_joint(smpl_consts):
    # rotating only the root joint rotates the whole zero-pose mesh rigidly about J_0
    beta = np.zeros((1, 10))
    th = np.zeros((1, 72)); th[0, :3] = [0.3, -0.2, 0.5]
    v0, _, _ = O.smpl_forward(beta, np.zeros((1, 72)), smpl_consts, F64)
    v1, _, Rs = O.smpl_forward(beta, th, smpl_consts, F64)
    J0 = (smpl_consts["J_regressor"].astype(np.float64).T @ smpl_consts["v_template"].astype(np.float64))[0]
    R = Rs[0, 0].numpy()
    # pose blend shapes ignore the root rotation (pose_feature uses joints 1..23)
    expect = (v0[0].numpy() - J0) @ R.T + J0
    assert np.allclose(v1[0].numpy(), expect, atol=1e-9)


def test_group_norm_constant_input_gives_beta():
    x = torch.full((2, 20, 2048), 3.25, dtype=F64)
    g = torch.rand(2048, dtype=F64) + 0.5
    b = torch.randn(2048, dtype=F64)
    y = O.group_norm_time(x, g, b)
    assert np.allclose(y.numpy(), b.numpy()[None, None].repeat(2, 0).repeat(20, 1), atol=1e-6)


def test_group_norm_statistics_span_time():
    torch.manual_seed(0)
    x = torch.randn(3, 20, 2048, dtype=F64) * 2 + 1
    y = O.group_norm_time(x, torch.ones(2048, dtype=F64), torch.zeros(2048, dtype=F64))
    yg = y.reshape(3, 20, 32, 64)
    assert np.allclose(yg.mean(dim=(1, 3)).numpy(), 0, atol=1e-9)
    assert np.allclose(yg.var(dim=(1, 3), unbiased=False).numpy(), 1, atol=1e-4)


def test_temporal_conv_delta_kernel_is_a_shift():
    torch.manual_seed(0)
    x = torch.randn(2, 20, 16, dtype=F64)
    for k, shift in ((0, -1), (1, 0), (2, 1)):
        w = torch.zeros(3, 1, 16, 16, dtype=F64)
        w[k, 0] = torch.eye(16, dtype=F64)
        y = O.temporal_conv3(x, w, torch.zeros(16, dtype=F64))
        ref = torch.zeros_like(x)
        if shift == 0:
            ref = x.clone()
        elif shift == -1:
            ref[:, 1:] = x[:, :-1]          # out[t] = x[t-1], zero at the window edge
        else:
            ref[:, :-1] = x[:, 1:]
        assert torch.equal(y, ref)


def test_maxpool_same_padding_is_bottom_right_only():
    # a one-hot at the last row/col must survive 3x3/2 SAME pooling of a 112 map into cell 55
    x = torch.full((1, 1, 112, 112), -5.0)
    x[0, 0, 111, 111] = 7.0
    y = torch.nn.functional.max_pool2d(torch.nn.functional.pad(x, (0, 1, 0, 1), value=float("-inf")), 3, stride=2)
    assert y.shape[-1] == 56 and y[0, 0, 55, 55] == 7.0 and y[0, 0, 0, 0] == -5.0


def test_window_plan_matches_reference_arithmetic():
    # tester.py:281-289 with the demo defaults B=8, T=20, fov=13
    assert O.window_plan(100, 8, 20, 13) == (6, 8, 2, 48)
    assert O.window_plan(64, 8, 20, 13) == (6, 8, 1, 20)
    assert O.window_plan(65, 8, 20, 13) == (6, 8, 2, 83)


def test_delta_omega_layout(weights):
    torch.manual_seed(0)
    phi = torch.randn(4, 2048, dtype=F64)
    mean = torch.tensor(weights["mean_param"], dtype=F64).expand(4, 85)
    om, deltas = O.call_hmr_ief(phi, mean, weights, (-5, 5), F64)
    for dt in (-5, 5):
        d = deltas[dt]
        assert torch.equal(d[:, 0], torch.ones(4, dtype=F64)) and torch.equal(d[:, 1:3], torch.zeros(4, 2, dtype=F64))
        assert torch.equal(d[:, 75:], om[:, 75:])
    assert not torch.allclose(deltas[-5][:, 3:75], deltas[5][:, 3:75])


def test_oracle_reproduces_golden_window_stages(weights, smpl_consts, golden_window):
    """float32 oracle vs the committed float64 golden, stage by stage (cheap stages only)."""
    T = O.OracleTester(weights, smpl_consts, batch_size=1, dtype=torch.float32)
    strips = T.movie_strips(torch.tensor(golden_window["phi"]).reshape(1, 20, -1)).reshape(20, -1)
    assert np.abs(strips.numpy() - golden_window["strips"]).max() < 2e-4
    om0, deltas = T.omegas(golden_window["strips"])
    got = torch.stack([om0] + [deltas[k] for k in sorted(deltas)]).numpy()
    assert np.abs(got - golden_window["omegas_all"]).max() < 2e-5
    out = T.smpl_outputs(torch.tensor(golden_window["omegas"][0]), torch.tensor(golden_window["cams"][0]))
    assert np.abs(out["verts"].numpy() - golden_window["verts"][0]).max() < 1e-5
    assert np.abs(out["joints"].numpy() - golden_window["joints"][0]).max() < 1e-5
    assert np.abs(out["kps"].numpy() - golden_window["kps"][0]).max() < 1e-5


def test_oracle_resnet_reproduces_golden_phi(weights, golden_window):
    frames = assets.make_synthetic_frames(4, seed=1)
    phi = O.resnet_v2_50(frames, weights, torch.float32).numpy()
    err = np.abs(phi - golden_window["phi"][:4]).max()
    assert err < 1e-4, err


def test_dedup_windowing_equals_literal_on_features(weights, smpl_consts, golden_video):
    """Per-frame independence of the (inference) ResNet: running it once per frame
    and windowing the FEATURES reproduces the literal predict_all_images."""
    T = O.OracleTester(weights, smpl_consts, batch_size=2, sequence_length=20, dtype=torch.float64)
    frames = assets.make_synthetic_frames(24, seed=7)
    phi = T.features(frames)
    phi0 = T.features(np.zeros((1, 224, 224, 3), np.float32))
    margin, g, count, num_fill = O.window_plan(24, 2, 20, T.fov)
    padded = torch.cat([phi0.expand(margin, -1), phi, phi0.expand(num_fill, -1)])
    wins = torch.stack([padded[i * g:i * g + 20] for i in range(count * 2)])
    strips = T.movie_strips(wins)[:, margin:-margin].reshape(-1, 2048)[:24]
    om0, _ = T.omegas(strips)
    assert np.abs(om0.numpy() - golden_video["omegas"]).max() < 1e-5


def test_quantize_models_the_two_storage_formats():
    """oracle.quantize: bf16 = 8 mantissa bits, f16x3 = an fp16 hi/lo pair (22 bits for values whose lo half is a normal
    fp16); identical to human_dynamics_amd.packing.to_split / from_split on the host side; filter banks are scaled per
    output channel like packing.row_pow2 scales them."""
    from human_dynamics_amd.packing import from_split, to_split
    x = torch.randn(4, 64, generator=torch.Generator().manual_seed(0), dtype=torch.float64) * 7
    assert O.quantize(x, None) is x
    q16, q3 = O.quantize(x, "bf16"), O.quantize(x, "f16x3")
    assert float(((q16 - x).abs() / x.abs()).max()) < 2.0 ** -8
    assert bool(((q3 - x).abs() <= torch.maximum(x.abs() * 2.0 ** -21, torch.tensor(6.0e-8, dtype=torch.float64))).all())
    assert torch.equal(q3.to(torch.float32), from_split(to_split(x.to(torch.float32))))
    assert torch.equal(O.quantize(q3, "f16x3"), q3)            # idempotent
    # a small-valued filter bank [in, out]: unscaled, its lo halves are fp16 subnormals; scaled per output channel they are not
    from human_dynamics_amd import packing
    w = torch.randn(256, 24, generator=torch.Generator().manual_seed(1), dtype=torch.float64) * 0.004
    w[:, 3] *= 40.0
    plain, scaled = O.quantize(w, "f16x3"), O.quantize(w, "f16x3", weight=True)
    assert float(((scaled - w).abs() / w.abs()).max()) < 2.0 ** -20 < float(((plain - w).abs() / w.abs()).max())
    k = packing.row_pow2(w.numpy().T)                           # rows = output channels
    host = packing.scale_rows(w.numpy().T, k)
    back = from_split(to_split(torch.from_numpy(np.ascontiguousarray(host)))).double().numpy() / np.exp2(k)[:, None]
    assert np.array_equal(back.T, scaled.numpy())
    assert 2.0 ** 13 <= np.abs(host).max(axis=1).min() and np.abs(host).max() < 2.0 ** 14


def test_storage_emulating_resnet(weights):
    """resnet_v2_50_emulated: with no rounding it is the plain restatement (folded fp32 BN constants:
    1e-7); with bf16 / f16x3 storage it predicts the error SIZE of those HIP modes; and the bf16 chain is
    chaotic -- a relative 1e-7 nudge before each rounding moves phi by a large fraction of the bf16 error,
    which is why the GPU tests gate that mode on the error size, not on element-wise agreement."""
    frames = assets.make_synthetic_frames(1, seed=1)
    exact = O.resnet_v2_50(frames, weights, torch.float64)
    rel = lambda a, b: float(torch.linalg.norm(a - b) / torch.linalg.norm(b))
    assert rel(O.resnet_v2_50_emulated(frames, weights, None), exact) < 1e-6
    e16 = O.resnet_v2_50_emulated(frames, weights, "bf16")
    e3 = O.resnet_v2_50_emulated(frames, weights, "f16x3")
    assert 1e-3 < rel(e16, exact) < 1e-2 and rel(e3, exact) < 2e-5
    assert rel(e16, exact) > 200 * rel(e3, exact)
    orig, g = O.quantize, torch.Generator().manual_seed(0)
    try:
        O.quantize = lambda x, em, weight=False: orig(x if em is None else x * (1 + 1e-7 * torch.randn(x.shape, generator=g, dtype=x.dtype)), em, weight)
        nudged = O.resnet_v2_50_emulated(frames, weights, "bf16")
    finally:
        O.quantize = orig
    assert rel(nudged, e16) > 0.2 * rel(e16, exact)


def test_emulated_resnet_with_folded_shortcut_differs_by_one_rounding(weights):
    """f16x3 emulation: accumulating the conv shortcut inside conv3 (what the HIP path does, csrc/gemm_conv.hip in2) instead
    of storing it first removes ONE 16-bit rounding of the shortcut tensor in four units -- the features move by ~1e-6
    relative, and both stay within the mode's distance of the unrounded graph."""
    frames = assets.make_synthetic_frames(1, seed=4)
    a = O.resnet_v2_50_emulated(frames, weights, "f16x3").numpy()
    b = O.resnet_v2_50_emulated(frames, weights, "f16x3", fold_shortcut=False).numpy()
    ref = O.resnet_v2_50(frames, weights, torch.float64).numpy()
    n = np.linalg.norm(ref)
    assert 0 < np.linalg.norm(a - b) / n < 2e-5
    assert np.linalg.norm(a - ref) / n < 5e-5 and np.linalg.norm(b - ref) / n < 5e-5


def test_why_the_split_format_has_fp16_halves(smpl_consts):
    """The storage-emulating oracle on the hard BatchNorm / GroupNorm weight set (oracle/hard_weights.py), one 20-frame
    window, whole path: bf16 halves (the split format of rounds 1-2, 16-17 bits) leave the 1e-4 tolerance, fp16 halves with
    per-output-channel scaled filters (22 bits, the format of csrc/common.h) stay an order of magnitude inside it -- at the
    same MFMA rate and the same bytes."""
    from oracle import hard_weights as H
    w = H.make_hard_weights(3)
    plain = assets.make_synthetic_weights(3)
    for k in plain:
        if k.endswith("fc3/weights"):
            w[k] = plain[k]
    frames = assets.make_synthetic_frames(20, seed=5)[None]
    ref = O.OracleTester(w, smpl_consts, batch_size=1, dtype=torch.float64).predict(frames)
    err = {}
    for em in ("bf16x3", "f16x3"):
        e = O.OracleTester(w, smpl_consts, batch_size=1, dtype=torch.float64, emulate=em).predict(frames)
        err[em] = max(float(np.abs(e[k] - ref[k]).max()) for k in ("verts", "verts_delta"))
    print("hard set, vertices: bf16 halves %.2e, fp16 halves + scaled filters %.2e" % (err["bf16x3"], err["f16x3"]))
    assert err["bf16x3"] > 1e-4 > 10 * err["f16x3"]
