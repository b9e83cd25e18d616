"""Model assets for the HMMR hot path: variable names/shapes and synthetic
generators.

The reference restores its variables from two TF checkpoints and a SMPL pickle
(src/evaluation/tester.py:92-152, src/tf_smpl/batch_smpl.py:27-87); none of
those files ship with the reference tree.  This module defines the *contract*
(variable names and shapes) as plain ``dict[str, ndarray]``
and provides deterministic synthetic generators with the right shapes and
statistics, so parity and throughput can be measured without the checkpoints.

Everything here is NumPy only (PCG64 streams are bit-reproducible across
machines), so the GPU box regenerates exactly the tensors the golden fixtures
were produced from.
"""
from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------- #
# Architecture constants (src/config.py:43-69 defaults are part of the contract)
# --------------------------------------------------------------------------- #
IMG_SIZE = 224
NUM_VERTS = 6890
NUM_JOINTS = 24
NUM_KPS = 25
NUM_BETAS = 10
NUM_THETA = 85           # [cam(3), pose(72), shape(10)], src/omega.py:231-235
FEAT_DIM = 2048
BN_EPS = 1e-5            # slim resnet_arg_scope batch_norm_epsilon
GN_EPS = 1e-6            # tf.contrib.layers.group_norm default
GN_GROUPS = 32

# slim resnet_v2_50: (name, base_depth, num_units, stride-on-last-unit)
RESNET_BLOCKS = (("block1", 64, 3, 2), ("block2", 128, 4, 2),
                 ("block3", 256, 6, 2), ("block4", 512, 3, 1))

# SMPL kinematic tree (kintree_table[0]; root parent stored as -1 here).
SMPL_PARENTS = np.array([-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13,
                         14, 16, 17, 18, 19, 20, 21], dtype=np.int32)


def resnet_units():
    """Yield (scope, c_in, base, depth, stride, has_shortcut_conv) per
    bottleneck unit, in execution order (slim resnet_v2.bottleneck)."""
    c_in = 64
    for bname, base, n_units, bstride in RESNET_BLOCKS:
        depth = 4 * base
        for u in range(1, n_units + 1):
            stride = bstride if u == n_units else 1
            scope = "resnet_v2_50/%s/unit_%d/bottleneck_v2" % (bname, u)
            yield scope, c_in, base, depth, stride, (c_in != depth)
            c_in = depth


def temporal_scopes(i):
    """Variable scopes of temporal block i (src/models.py:159,182,192,219:
    scope strings are concatenated with the block name, no separator)."""
    n = "block_%d" % i
    return ("AZ_FC_block_preact_gn1" + n, "AZ_FC_block2_conv1" + n,
            "AZ_FC_block_preact_gn2" + n, "AZ_FC_block2_conv2" + n)


def ief_scopes(delta_t_values=(-5, 5)):
    """IEF regressor scopes: key -> (scope, theta_dim) (src/models.py:344-347)."""
    out = {0: ("single_view_ief", 85)}
    for dt in delta_t_values:
        if dt > 0:
            out[dt] = ("single_view_ief_future%d" % dt, 72)
        else:
            out[dt] = ("single_view_ief_past%d" % abs(dt), 72)
    return out


# --------------------------------------------------------------------------- #
# Synthetic network weights
# --------------------------------------------------------------------------- #
def _conv_w(rng, kh, kw, cin, cout, gain=2.0):
    std = np.sqrt(gain / (kh * kw * cin))
    return (rng.standard_normal((kh, kw, cin, cout), dtype=np.float32) * std)


def _bn(rng, c, prefix, out):
    out[prefix + "/gamma"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
    out[prefix + "/beta"] = (rng.standard_normal(c) * 0.1).astype(np.float32)
    out[prefix + "/moving_mean"] = (rng.standard_normal(c) * 0.1).astype(np.float32)
    out[prefix + "/moving_variance"] = rng.uniform(0.5, 1.5, c).astype(np.float32)


def make_synthetic_weights(seed=0, num_conv_layers=3, delta_t_values=(-5, 5),
                           with_hallucinator=False):
    """All trainable/non-trainable network variables with the checkpoint names
    and shapes of SURVEY.md App. B.  BN/GN affine parameters and moving
    statistics are randomised so that a wrong fold cannot hide behind an
    identity transform."""
    rng = np.random.Generator(np.random.PCG64(seed))
    w = {}
    # ---- ResNet-v2-50 (slim) ------------------------------------------------
    w["resnet_v2_50/conv1/weights"] = _conv_w(rng, 7, 7, 3, 64)
    w["resnet_v2_50/conv1/biases"] = (rng.standard_normal(64) * 0.1).astype(np.float32)
    for scope, c_in, base, depth, stride, has_sc in resnet_units():
        _bn(rng, c_in, scope + "/preact", w)
        if has_sc:
            w[scope + "/shortcut/weights"] = _conv_w(rng, 1, 1, c_in, depth, gain=1.0)
            w[scope + "/shortcut/biases"] = (rng.standard_normal(depth) * 0.1).astype(np.float32)
        w[scope + "/conv1/weights"] = _conv_w(rng, 1, 1, c_in, base)
        _bn(rng, base, scope + "/conv1/BatchNorm", w)
        w[scope + "/conv2/weights"] = _conv_w(rng, 3, 3, base, base)
        _bn(rng, base, scope + "/conv2/BatchNorm", w)
        # residual branch a bit smaller than the trunk so 16 units stay O(1)
        w[scope + "/conv3/weights"] = _conv_w(rng, 1, 1, base, depth, gain=0.5)
        w[scope + "/conv3/biases"] = (rng.standard_normal(depth) * 0.1).astype(np.float32)
    _bn(rng, 2048, "resnet_v2_50/postnorm", w)
    # the trunk's scale grows to ~10 over 16 un-normalised residual units; give
    # postnorm statistics of that size so phi is O(1) like a trained network's
    w["resnet_v2_50/postnorm/moving_variance"] *= np.float32(64.0)
    # ---- f_movie temporal encoder (src/models.py:121-228) -------------------
    for i in range(num_conv_layers):
        gn1, c1, gn2, c2 = temporal_scopes(i)
        for gn in (gn1, gn2):
            w[gn + "/gamma"] = rng.uniform(0.5, 1.5, FEAT_DIM).astype(np.float32)
            w[gn + "/beta"] = (rng.standard_normal(FEAT_DIM) * 0.1).astype(np.float32)
        w[c1 + "/weights"] = _conv_w(rng, 3, 1, FEAT_DIM, FEAT_DIM, gain=2.0)
        w[c1 + "/biases"] = (rng.standard_normal(FEAT_DIM) * 0.1).astype(np.float32)
        w[c2 + "/weights"] = _conv_w(rng, 3, 1, FEAT_DIM, FEAT_DIM, gain=0.2)
        w[c2 + "/biases"] = (rng.standard_normal(FEAT_DIM) * 0.1).astype(np.float32)
    # ---- IEF regressors (src/models.py:80-116, 380-415) ---------------------
    for _, (scope, nd) in sorted(ief_scopes(delta_t_values).items()):
        p = scope + "/3D_module"
        w[p + "/fc1/weights"] = (rng.standard_normal((FEAT_DIM + nd, 1024), dtype=np.float32)
                                 * np.sqrt(2.0 / (FEAT_DIM + nd)))
        w[p + "/fc1/biases"] = (rng.standard_normal(1024) * 0.1).astype(np.float32)
        w[p + "/fc2/weights"] = (rng.standard_normal((1024, 1024), dtype=np.float32)
                                 * np.sqrt(2.0 / 1024))
        w[p + "/fc2/biases"] = (rng.standard_normal(1024) * 0.1).astype(np.float32)
        # large enough that an upstream error moves theta visibly
        w[p + "/fc3/weights"] = (rng.standard_normal((1024, nd), dtype=np.float32)
                                 * (0.08 / np.sqrt(1024)))
        w[p + "/fc3/biases"] = (rng.standard_normal(nd) * 0.01).astype(np.float32)
    if with_hallucinator:   # src/models.py:270-296 (pred_mode == 'hal')
        for k in ("fc1", "fc2", "fc3"):
            g = 2.0 if k != "fc3" else 0.1
            w["fc2_res/%s/weights" % k] = (rng.standard_normal((FEAT_DIM, FEAT_DIM), dtype=np.float32)
                                           * np.sqrt(g / FEAT_DIM))
            w["fc2_res/%s/biases" % k] = (rng.standard_normal(FEAT_DIM) * 0.1).astype(np.float32)
    # ---- mean theta (src/evaluation/tester.py:118-152) ----------------------
    w["mean_param"] = make_mean_theta(seed + 1000)[None, :]
    # checkpoint variables are float32 (NumPy-2 scalar promotion would otherwise
    # leave some of the products above in float64)
    return {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in w.items()}


def make_synthetic_ief_weights(seed=0, delta_t_values=(-5, 5), delta_nd=72):
    """Only the IEF variables + mean_param, with `delta_nd`-wide delta regressors: 72 = use_optcam=True (what
    make_synthetic_weights builds), 75 = use_optcam=False (camera + pose, src/models.py:333-336)."""
    rng = np.random.Generator(np.random.PCG64([seed, delta_nd]))
    w = {}
    for key, (scope, nd) in sorted(ief_scopes(delta_t_values).items()):
        nd = nd if key == 0 else delta_nd
        p = scope + "/3D_module"
        w[p + "/fc1/weights"] = rng.standard_normal((FEAT_DIM + nd, 1024), dtype=np.float32) * np.sqrt(2.0 / (FEAT_DIM + nd))
        w[p + "/fc1/biases"] = (rng.standard_normal(1024) * 0.1).astype(np.float32)
        w[p + "/fc2/weights"] = rng.standard_normal((1024, 1024), dtype=np.float32) * np.sqrt(2.0 / 1024)
        w[p + "/fc2/biases"] = (rng.standard_normal(1024) * 0.1).astype(np.float32)
        w[p + "/fc3/weights"] = rng.standard_normal((1024, nd), dtype=np.float32) * (0.08 / np.sqrt(1024))
        w[p + "/fc3/biases"] = (rng.standard_normal(nd) * 0.01).astype(np.float32)
    w["mean_param"] = make_mean_theta(seed + 1000)[None, :]
    return {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in w.items()}


def make_mean_theta(seed=1000):
    """[cam(0.9,0,0), pose (root = pi,0,0), shape] as load_mean_params builds it."""
    rng = np.random.Generator(np.random.PCG64(seed))
    pose = (rng.standard_normal(72) * 0.2).astype(np.float32)
    pose[:3] = 0.0
    pose[0] = np.pi
    shape = (rng.standard_normal(10) * 0.5).astype(np.float32)
    return np.concatenate([np.array([0.9, 0.0, 0.0], np.float32), pose, shape]).astype(np.float32)


# --------------------------------------------------------------------------- #
# Synthetic SMPL constants in the src/tf_smpl parameter layout
# --------------------------------------------------------------------------- #
def _sparse_rows_sum1(rng, rows, cols, nnz):
    """[rows, cols] non-negative matrix, <= nnz non-zeros per row, rows sum to 1."""
    m = np.zeros((rows, cols), np.float32)
    for r in range(rows):
        idx = rng.choice(cols, size=nnz, replace=False)
        v = rng.uniform(0.05, 1.0, nnz)
        m[r, idx] = (v / v.sum()).astype(np.float32)
    return m


def make_synthetic_smpl(seed=2, lbs_nnz=4):
    """SMPL-shaped constants with the layout of batch_smpl.py:35-80:
    ``shapedirs`` [10, 6890*3] (column = 3*v + c), ``posedirs`` [207, 6890*3],
    ``J_regressor`` [6890, 24] and ``cocoplus_regressor`` [6890, 25] stored
    transposed, ``lbs_weights`` [6890, 24] (<= lbs_nnz non-zeros per vertex,
    rows sum to 1), ``parents`` int32[24]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    s = {}
    s["v_template"] = (rng.standard_normal((NUM_VERTS, 3)) * 0.3).astype(np.float32)
    s["shapedirs"] = (rng.standard_normal((NUM_BETAS, NUM_VERTS * 3)) * 0.01).astype(np.float32)
    s["posedirs"] = (rng.standard_normal((207, NUM_VERTS * 3)) * 0.001).astype(np.float32)
    s["J_regressor"] = np.ascontiguousarray(_sparse_rows_sum1(rng, NUM_JOINTS, NUM_VERTS, 32).T)
    s["cocoplus_regressor"] = np.ascontiguousarray(_sparse_rows_sum1(rng, NUM_KPS, NUM_VERTS, 48).T)
    s["lbs_weights"] = _sparse_rows_sum1(rng, NUM_VERTS, NUM_JOINTS, lbs_nnz)
    s["parents"] = SMPL_PARENTS.copy()
    return s


def make_synthetic_frames(n, seed=1, img_size=IMG_SIZE, start=0):
    """Frames [n, H, W, 3] float32 in [-1, 1].  Frame i depends only on
    (seed, start + i), so any rank can regenerate its own shard (BASELINE
    config 5).  Each frame is a per-channel offset + a blocky low-frequency
    pattern + uniform noise, so that ResNet features differ from frame to
    frame (pure noise images pool to nearly identical features)."""
    out = np.empty((n, img_size, img_size, 3), np.float32)
    cell = img_size // 7
    for i in range(n):
        rng = np.random.Generator(np.random.PCG64([seed, start + i]))
        off = rng.uniform(-0.4, 0.4, 3).astype(np.float32)
        coarse = rng.uniform(-0.5, 0.5, (7, 7, 3)).astype(np.float32)
        amp = np.float32(rng.uniform(0.1, 0.4))
        noise = rng.random((img_size, img_size, 3), dtype=np.float32) * 2.0 - 1.0
        pat = np.repeat(np.repeat(coarse, cell, axis=0), cell, axis=1)
        out[i] = np.clip(off + pat + amp * noise, -1.0, 1.0)
    return out
