"""CPU oracle for HMMR's inference hot path -- TEST INFRASTRUCTURE ONLY.

*** PARITY: PINNED TO THE REFERENCE'S OWN SOURCE, EXECUTED; TF LAYER SEMANTICS RESTATED ***
The reference (akanazawa/human_dynamics) has no tests and no golden vectors, and TensorFlow 1.8 /
tf-slim cannot be installed in this image.  This file is a restatement of the
reference graph in PyTorch-CPU (float64 or float32), written from the reference sources and the TF-1.8
semantics of the un-vendored ops it calls.  It is pinned as follows.

  * Reference source EXECUTED (tests/golden/make_reference_golden.py, make_resnet_golden.py import the
    files from /root/reference and run them unmodified, in float64, on oracle/tf_shim.py, a NumPy
    stand-in for the TF ops and tf.contrib layers they call); this oracle agrees with the outputs to 1e-9
    or better (tests/test_reference_golden.py):
      - src/tf_smpl/*, src/omega.py: smpl_forward / batch_rodrigues / batch_global_rigid_transformation
        / batch_orth_proj_idrot and the OmegasPred container semantics (smpl_outputs);
      - Tester.predict_all_images: the sliding-window arithmetic;
      - src/models.py: az_fc2_groupnorm / az_fc_block2, batch_pred_omega / call_hmr_ief / hmr_ief
        (wiring + checkpoint variable names), fc2_res, and encoder_resnet -- the latter on
        oracle/slim_resnet_v2.py, a function-for-function transcription of slim's resnet_v2.py /
        resnet_utils.py kept apart from this file: phi and all 17 collected unit outputs agree to 1e-12.
  * What remains a restatement (it lives inside TensorFlow, which cannot run here): the SEMANTICS of the
    tf.contrib layers -- conv2d SAME/VALID, batch_norm (inference), max_pool2d SAME, group_norm,
    fully_connected -- stated once in oracle/tf_shim.py (NumPy, explicit padding arithmetic) and once here
    (PyTorch conv2d / pooling): two independent implementations that agree, plus the algebraic
    known-answer tests of tests/test_oracle.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product path (human_dynamics_amd) never does.

All citations are relative to /root/reference.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
GN_EPS = 1e-6

_BLOCKS = (("block1", 64, 3, 2), ("block2", 128, 4, 2),
           ("block3", 256, 6, 2), ("block4", 512, 3, 1))


def _t(a, dtype):
    return torch.as_tensor(np.asarray(a)).to(dtype)


# --------------------------------------------------------------------------- #
# ResNet-v2-50 as invoked at src/models.py:50-77 (tf.contrib.slim.nets.resnet_v2)
# --------------------------------------------------------------------------- #
def _bn(x, w, prefix, dtype):
    """Inference batch norm on NCHW: gamma*(x-mean)*rsqrt(var+eps)+beta."""
    g = _t(w[prefix + "/gamma"], dtype)
    b = _t(w[prefix + "/beta"], dtype)
    m = _t(w[prefix + "/moving_mean"], dtype)
    v = _t(w[prefix + "/moving_variance"], dtype)
    inv = torch.rsqrt(v + BN_EPS) * g
    return x * inv.view(1, -1, 1, 1) + (b - m * inv).view(1, -1, 1, 1)


def _conv(x, w_hwio, dtype, stride=1, pad=0, bias=None):
    wt = _t(w_hwio, dtype).permute(3, 2, 0, 1).contiguous()   # HWIO -> OIHW
    b = None if bias is None else _t(bias, dtype)
    return F.conv2d(x, wt, b, stride=stride, padding=pad)


# --------------------------------------------------------------------------- #
# Storage-precision emulation (NOT part of the reference): the HIP path's reduced-precision modes
# round the GEMM operands and every tensor they store between launches; an oracle that rounds at
# exactly those points (and nowhere else, accumulating in float64) turns "bf16-sized error" into a
# tight comparison.  emulate = None | 'bf16' | 'f16x3' | 'bf16x3'.
# --------------------------------------------------------------------------- #
def quantize(x, emulate, weight=False):
    """x as the HIP path stores it: bf16 (round to nearest even), or the f16x3 hi/lo pair
    (hi = fp16(x), lo = fp16(x - hi); x ~ hi + lo; filters scaled, see below).  'bf16x3' = the bf16 hi/lo pair of
    rounds 1-2, kept for comparison."""
    if emulate is None:
        return x
    x32 = x.to(torch.float32)
    hi = x32.to(torch.bfloat16).to(torch.float32)
    if emulate == "bf16":
        return hi.to(x.dtype)
    if emulate == "bf16x3":
        return (hi.to(torch.float64) + (x32 - hi).to(torch.bfloat16).to(torch.float64)).to(x.dtype)
    if emulate == "f16x3":
        # the split format of csrc/common.h: fp16 halves (11 + 11 bits; fp16 subnormals kept, as the f16 MFMA keeps them),
        # values clamped to +-65504.  Filter banks (weight=True; output channel = last axis, HWIO / [in, out]) are scaled
        # per output channel by a power of two to a maximum in [2^13, 2^14) before the split, exactly undone afterwards --
        # packing.row_pow2: it keeps their lo halves out of the fp16 subnormal range.
        x32 = x32.clamp(-65504.0, 65504.0)
        scale = None
        if weight:
            m = x32.abs().reshape(-1, x32.shape[-1]).max(dim=0).values.to(torch.float64)
            k = torch.where(m > 0, 13.0 - torch.floor(torch.log2(torch.where(m > 0, m, torch.ones_like(m)))), torch.zeros_like(m))
            scale = torch.pow(torch.tensor(2.0, dtype=torch.float64), k).to(torch.float32)
            x32 = x32 * scale
        h16 = x32.to(torch.float16).to(torch.float32)
        out = h16.to(torch.float64) + (x32 - h16).to(torch.float16).to(torch.float64)
        if scale is not None:
            out = out / scale.to(torch.float64)
        return out.to(x.dtype)
    raise ValueError("emulate must be None, 'bf16', 'f16x3' or 'bf16x3' (the split format of rounds 1-2)")


def _fold_bn32(w, prefix):
    """Inference BN folded the way human_dynamics_amd/packing.py folds it (float64 -> float32
    scale and shift), which is what the HIP epilogues apply as one fused multiply-add."""
    g = np.asarray(w[prefix + "/gamma"], np.float64)
    b = np.asarray(w[prefix + "/beta"], np.float64)
    m = np.asarray(w[prefix + "/moving_mean"], np.float64)
    v = np.asarray(w[prefix + "/moving_variance"], np.float64)
    scale = g / np.sqrt(v + BN_EPS)
    return scale.astype(np.float32), (b - m * scale).astype(np.float32)


def resnet_v2_50_emulated(images_nhwc, w, emulate, fold_shortcut=None):
    """resnet_v2_50 below with the rounding points of csrc/resnet.hip in mode `emulate`: image and
    filters as operands, and every tensor the launch sequence stores (stem conv + bias, each unit's
    pre-activation, h1, h2, conv shortcut, trunk).  Arithmetic between rounding points is float64
    (the HIP path accumulates in fp32, so agreement is to fp32 roundoff plus the rare operand that
    such a difference tips over a rounding boundary)."""
    dt = torch.float64
    q = lambda t: quantize(t, emulate)
    if fold_shortcut is None:      # split modes: the conv shortcut of a stride-1 unit is accumulated inside conv3's GEMM, never stored
        fold_shortcut = emulate in ("bf16x3", "f16x3")

    def conv(x, name, stride=1, pad=0):
        wt = quantize(_t(w[name], dt), emulate, weight=True).permute(3, 2, 0, 1).contiguous()
        return F.conv2d(x, wt, None, stride=stride, padding=pad)

    def affine(x, scale, shift):
        return x * _t(scale, dt).view(1, -1, 1, 1) + _t(shift, dt).view(1, -1, 1, 1)

    x = q(_t(images_nhwc, dt)).permute(0, 3, 1, 2).contiguous()
    x = q(conv(x, "resnet_v2_50/conv1/weights", 2, 3) + _t(w["resnet_v2_50/conv1/biases"], dt).view(1, -1, 1, 1))
    x = F.max_pool2d(F.pad(x, (0, 1, 0, 1), value=float("-inf")), 3, stride=2)
    raw, c_in = x, 64
    for bname, base, n_units, bstride in _BLOCKS:
        depth = 4 * base
        for u in range(1, n_units + 1):
            stride = bstride if u == n_units else 1
            sc = "resnet_v2_50/%s/unit_%d/bottleneck_v2" % (bname, u)
            preact = q(torch.relu(affine(raw, *_fold_bn32(w, sc + "/preact"))))
            if c_in == depth:
                shortcut = raw if stride == 1 else raw[:, :, ::stride, ::stride]
            else:
                shortcut = (conv(preact, sc + "/shortcut/weights", stride)
                            + _t(w[sc + "/shortcut/biases"], dt).view(1, -1, 1, 1))
                if not (fold_shortcut and stride == 1):
                    shortcut = q(shortcut)
            r = q(torch.relu(affine(conv(preact, sc + "/conv1/weights"), *_fold_bn32(w, sc + "/conv1/BatchNorm"))))
            r = q(torch.relu(affine(conv(r, sc + "/conv2/weights", stride, 1), *_fold_bn32(w, sc + "/conv2/BatchNorm"))))
            r = conv(r, sc + "/conv3/weights") + _t(w[sc + "/conv3/biases"], dt).view(1, -1, 1, 1)
            raw = q(shortcut + r)
            c_in = depth
    x = torch.relu(affine(raw, *_fold_bn32(w, "resnet_v2_50/postnorm")))
    return x.mean(dim=(2, 3))


def resnet_v2_50(images_nhwc, w, dtype=torch.float64, return_endpoints=False):
    """images [N,224,224,3] -> phi [N,2048].

    slim resnet_v2_50(num_classes=None, is_training=False), SURVEY App. A:
    stem conv2d_same 7x7/2 with bias and no BN/ReLU; max_pool 3x3/2 SAME
    (pad bottom/right only); bottleneck_v2 units with the stride on the LAST
    unit of each block (conv2d_same: explicit symmetric pad + VALID when
    stride > 1); postnorm BN+ReLU; mean over H,W."""
    ep = {}
    x = _t(images_nhwc, dtype).permute(0, 3, 1, 2).contiguous()
    x = _conv(x, w["resnet_v2_50/conv1/weights"], dtype, stride=2, pad=3,
              bias=w["resnet_v2_50/conv1/biases"])
    ep["conv1"] = x
    # TF SAME for k=3,s=2 on 112: total pad 1, all of it after.
    x = F.max_pool2d(F.pad(x, (0, 1, 0, 1), value=float("-inf")), 3, stride=2)
    ep["pool1"] = x
    c_in = 64
    for bname, base, n_units, bstride in _BLOCKS:
        depth = 4 * base
        for u in range(1, n_units + 1):
            stride = bstride if u == n_units else 1
            sc = "resnet_v2_50/%s/unit_%d/bottleneck_v2" % (bname, u)
            preact = torch.relu(_bn(x, w, sc + "/preact", dtype))
            if c_in == depth:
                shortcut = x if stride == 1 else x[:, :, ::stride, ::stride]
            else:
                shortcut = _conv(preact, w[sc + "/shortcut/weights"], dtype,
                                 stride=stride, bias=w[sc + "/shortcut/biases"])
            r = _conv(preact, w[sc + "/conv1/weights"], dtype)
            r = torch.relu(_bn(r, w, sc + "/conv1/BatchNorm", dtype))
            r = _conv(r, w[sc + "/conv2/weights"], dtype, stride=stride, pad=1)
            r = torch.relu(_bn(r, w, sc + "/conv2/BatchNorm", dtype))
            r = _conv(r, w[sc + "/conv3/weights"], dtype, bias=w[sc + "/conv3/biases"])
            x = shortcut + r
            c_in = depth
            ep["%s/unit_%d" % (bname, u)] = x
    x = torch.relu(_bn(x, w, "resnet_v2_50/postnorm", dtype))
    phi = x.mean(dim=(2, 3))
    if return_endpoints:
        return phi, ep
    return phi


# --------------------------------------------------------------------------- #
# f_movie: az_fc2_groupnorm / az_fc_block2, src/models.py:121-228
# --------------------------------------------------------------------------- #
def group_norm_time(x, gamma, beta, groups=32):
    """tf.contrib.layers.group_norm(channels_axis=-1, reduction_axes=(-3,-2))
    on [B,T,1,C] (src/models.py:155-161): statistics per (b, group) over
    (T, 1, C/groups), population variance, eps 1e-6.
    x: [B,T,C]."""
    B, T, C = x.shape
    xg = x.reshape(B, T, groups, C // groups)
    mean = xg.mean(dim=(1, 3), keepdim=True)
    var = ((xg - mean) ** 2).mean(dim=(1, 3), keepdim=True)
    gain = torch.rsqrt(var + GN_EPS) * gamma.view(1, 1, groups, C // groups)
    offset = -mean * gain + beta.view(1, 1, groups, C // groups)
    return (xg * gain + offset).reshape(B, T, C)


def temporal_conv3(x, w_hwio, bias):
    """tf.contrib.layers.conv2d kernel [3,1], stride 1, SAME, bias, no act on
    [B,T,1,C] (src/models.py:173-184): out[t] = sum_k x[t+k-1] W[k] + b with
    zero rows outside the window.  x: [B,T,C]; w: [3,1,Cin,Cout]."""
    B, T, C = x.shape
    xp = F.pad(x, (0, 0, 1, 1))
    out = bias.view(1, 1, -1).expand(B, T, w_hwio.shape[3]).clone()
    for k in range(3):
        out = out + xp[:, k:k + T, :] @ w_hwio[k, 0]
    return out


def az_fc2_groupnorm(phi_btc, w, num_conv_layers=3, dtype=torch.float64, emulate=None):
    """emulate: round the two GEMM operands of each temporal conv as csrc/temporal.hip stores them (the
    GroupNorm+ReLU output and the filters); trunk, statistics and conv outputs stay unrounded."""
    q = lambda t: quantize(t, emulate)
    net = _t(phi_btc, dtype)
    for i in range(num_conv_layers):
        n = "block_%d" % i
        gn1, c1 = "AZ_FC_block_preact_gn1" + n, "AZ_FC_block2_conv1" + n
        gn2, c2 = "AZ_FC_block_preact_gn2" + n, "AZ_FC_block2_conv2" + n
        h = q(torch.relu(group_norm_time(net, _t(w[gn1 + "/gamma"], dtype), _t(w[gn1 + "/beta"], dtype))))
        h = temporal_conv3(h, quantize(_t(w[c1 + "/weights"], dtype), emulate, weight=True), _t(w[c1 + "/biases"], dtype))
        h = q(torch.relu(group_norm_time(h, _t(w[gn2 + "/gamma"], dtype), _t(w[gn2 + "/beta"], dtype))))
        h = temporal_conv3(h, quantize(_t(w[c2 + "/weights"], dtype), emulate, weight=True), _t(w[c2 + "/biases"], dtype))
        net = h + net                                   # src/models.py:226
    return net


def fc2_res(phi_btc, w, dtype=torch.float64):
    """Hallucinator, src/models.py:270-296 (pred_mode == 'hal')."""
    phi = _t(phi_btc, dtype)
    h = torch.relu(phi @ _t(w["fc2_res/fc1/weights"], dtype) + _t(w["fc2_res/fc1/biases"], dtype))
    h = torch.relu(h @ _t(w["fc2_res/fc2/weights"], dtype) + _t(w["fc2_res/fc2/biases"], dtype))
    h = h @ _t(w["fc2_res/fc3/weights"], dtype) + _t(w["fc2_res/fc3/biases"], dtype)
    return h + phi


# --------------------------------------------------------------------------- #
# IEF: hmr_ief / call_hmr_ief / batch_pred_omega, src/models.py:233-415
# --------------------------------------------------------------------------- #
def hmr_ief(phi, omega_start, w, scope, num_stage=3, dtype=torch.float64, emulate=None):
    """emulate: the rounding points of csrc/ief.hip -- phi and the phi rows of fc1, fc2, fc3 as operands;
    the stored phi.W1 + b1, h1 and h2; the theta state and its rows of fc1 stay fp32."""
    q = lambda t: quantize(t, emulate)
    p = scope + "/3D_module"
    W1, b1 = _t(w[p + "/fc1/weights"], dtype), _t(w[p + "/fc1/biases"], dtype)
    W2, b2 = _t(w[p + "/fc2/weights"], dtype), _t(w[p + "/fc2/biases"], dtype)
    W3, b3 = _t(w[p + "/fc3/weights"], dtype), _t(w[p + "/fc3/biases"], dtype)
    theta = omega_start
    if emulate is not None:
        nphi = phi.shape[1]
        qw = lambda t: quantize(t, emulate, weight=True)
        pre = q(q(phi) @ qw(W1[:nphi]) + b1)
        for _ in range(num_stage):
            h = q(torch.relu(pre + theta @ W1[nphi:]))
            h = q(torch.relu(h @ qw(W2) + b2))
            theta = theta + (h @ qw(W3) + b3)
        return theta
    for _ in range(num_stage):
        state = torch.cat([phi, theta], dim=1)          # models.py:402
        h = torch.relu(state @ W1 + b1)                 # dropout = identity at test
        h = torch.relu(h @ W2 + b2)
        theta = theta + (h @ W3 + b3)                   # models.py:410
    return theta


def call_hmr_ief(phi, omega_start, w, delta_t_values=(-5, 5), dtype=torch.float64, emulate=None,
                 use_optcam=True, use_delta_from_pred=True):
    """models.py:299-377.  Defaults = use_optcam=True, use_delta_from_pred=True as tester.py:196-207 calls it."""
    theta_here = hmr_ief(phi, omega_start, w, "single_view_ief", dtype=dtype, emulate=emulate)
    nd = 72 if use_optcam else 3 + 72                   # models.py:333-336
    deltas = {}
    for dt in delta_t_values:
        scope = "single_view_ief" + ("_future%d" % dt if dt > 0 else "_past%d" % abs(dt))
        start_full = theta_here if use_delta_from_pred else omega_start      # models.py:349
        beta = start_full[:, -10:]
        start = start_full[:, 3:3 + nd] if use_optcam else start_full[:, :nd]   # models.py:353-357
        d = hmr_ief(phi, start, w, scope, dtype=dtype, emulate=emulate)
        n = d.shape[0]
        if use_optcam:
            deltas[dt] = torch.cat([torch.ones(n, 1, dtype=dtype), torch.zeros(n, 2, dtype=dtype),
                                    d, beta], dim=1)    # models.py:367-371
        else:
            deltas[dt] = torch.cat([d[:, :75], beta], dim=1)                  # models.py:372-373
    return theta_here, deltas


# --------------------------------------------------------------------------- #
# SMPL: src/tf_smpl/batch_lbs.py, batch_smpl.py, projection.py
# --------------------------------------------------------------------------- #
def batch_skew(vec):
    """batch_lbs.py:15-39."""
    n = vec.shape[0]
    z = torch.zeros(n, dtype=vec.dtype)
    return torch.stack([z, -vec[:, 2], vec[:, 1],
                        vec[:, 2], z, -vec[:, 0],
                        -vec[:, 1], vec[:, 0], z], dim=1).reshape(n, 3, 3)


def batch_rodrigues(theta):
    """batch_lbs.py:42-60: angle = ||theta + 1e-8||, r = theta / angle."""
    angle = torch.linalg.norm(theta + 1e-8, dim=1, keepdim=True)
    r = theta / angle
    angle = angle.unsqueeze(-1)
    cos, sin = torch.cos(angle), torch.sin(angle)
    outer = r.unsqueeze(2) * r.unsqueeze(1)
    eye = torch.eye(3, dtype=theta.dtype).unsqueeze(0)
    return cos * eye + (1 - cos) * outer + sin * batch_skew(r)


def batch_global_rigid_transformation(Rs, Js, parents, rotate_base=False):
    """batch_lbs.py:133-194."""
    N = Rs.shape[0]
    dtype = Rs.dtype
    if rotate_base:                                     # batch_lbs.py:151-158
        rot_x = torch.tensor([[1, 0, 0], [0, -1, 0], [0, 0, -1]], dtype=dtype)
        Rs = torch.cat([(Rs[:, 0] @ rot_x).unsqueeze(1), Rs[:, 1:]], dim=1)

    def make_A(R, t):
        top = torch.cat([R, t.reshape(N, 3, 1)], dim=2)
        bot = torch.tensor([0, 0, 0, 1], dtype=dtype).expand(N, 1, 4)
        return torch.cat([top, bot], dim=1)

    results = [make_A(Rs[:, 0], Js[:, 0])]
    for i in range(1, len(parents)):
        j_here = Js[:, i] - Js[:, parents[i]]
        results.append(results[parents[i]] @ make_A(Rs[:, i], j_here))
    results = torch.stack(results, dim=1)                       # [N,24,4,4]
    new_J = results[:, :, :3, 3]
    Js_w0 = torch.cat([Js, torch.zeros(N, 24, 1, dtype=dtype)], dim=2).unsqueeze(-1)
    init_bone = results @ Js_w0                                 # [N,24,4,1]
    init_bone = F.pad(init_bone, (3, 0))
    return new_J, results - init_bone


def smpl_forward(beta, theta, smpl, dtype=torch.float64):
    """SMPL.__call__(beta[N,10], theta[N,72], get_skin=True), batch_smpl.py:89-162.
    Returns verts [N,6890,3], joints [N,25,3], Rs [N,24,3,3]."""
    beta = _t(beta, dtype)
    theta = _t(theta, dtype).reshape(-1, 72)
    N = beta.shape[0]
    v_template = _t(smpl["v_template"], dtype)
    shapedirs = _t(smpl["shapedirs"], dtype)
    posedirs = _t(smpl["posedirs"], dtype)
    J_regressor = _t(smpl["J_regressor"], dtype)
    weights = _t(smpl["lbs_weights"], dtype)
    kreg = _t(smpl["cocoplus_regressor"], dtype)
    parents = [int(p) for p in np.asarray(smpl["parents"])]
    nv = v_template.shape[0]

    v_shaped = (beta @ shapedirs).reshape(N, nv, 3) + v_template          # :110-112
    J = torch.stack([v_shaped[:, :, c] @ J_regressor for c in range(3)], dim=2)   # :115-118
    Rs = batch_rodrigues(theta.reshape(-1, 3)).reshape(N, 24, 3, 3)       # :123-124
    pose_feature = (Rs[:, 1:] - torch.eye(3, dtype=dtype)).reshape(N, 207)  # :127-128
    v_posed = (pose_feature @ posedirs).reshape(N, nv, 3) + v_shaped      # :131-133
    _, A = batch_global_rigid_transformation(Rs, J, parents)              # :136-137
    T = (weights @ A.reshape(N, 24, 16)).reshape(N, nv, 4, 4)             # :141-146
    v_homo = torch.cat([v_posed, torch.ones(N, nv, 1, dtype=dtype)], dim=2)
    verts = (T @ v_homo.unsqueeze(-1))[:, :, :3, 0]                       # :147-151
    joints = torch.stack([verts[:, :, c] @ kreg for c in range(3)], dim=2)  # :154-157
    return verts, joints, Rs


def batch_orth_proj_idrot(X, camera):
    """projection.py:16-29: s * (X_xy + t)."""
    camera = camera.reshape(-1, 1, 3)
    return camera[:, :, 0:1] * (X[:, :, :2] + camera[:, :, 1:])


# --------------------------------------------------------------------------- #
# Tester.predict / predict_all_images, src/evaluation/tester.py:169-312
# --------------------------------------------------------------------------- #
class OracleTester(object):
    """CPU restatement of ``Tester`` (pred_mode 'pred' or 'hal')."""

    def __init__(self, weights, smpl, batch_size=8, sequence_length=20,
                 num_conv_layers=3, delta_t_values=(-5, 5), pred_mode="pred",
                 dtype=torch.float64, emulate=None):
        self.w, self.smpl = weights, smpl
        self.batch_size, self.sequence_length = batch_size, sequence_length
        self.num_conv_layers = num_conv_layers
        self.fov = num_conv_layers * 4 + 1                      # tester.py:48
        self.delta_t_values = [int(d) for d in delta_t_values]
        self.pred_mode = pred_mode
        self.dtype = dtype
        self.emulate = emulate          # None | 'bf16' | 'f16x3' | 'bf16x3': storage-precision emulation of the HIP modes
        if emulate is not None and (dtype != torch.float64 or pred_mode != "pred"):
            raise ValueError("storage-precision emulation runs in float64, pred_mode 'pred'")
        self.img_size = 224

    # -- stages, exposed separately so each HIP stage can be checked alone ----
    def features(self, frames_nhwc, chunk=16):
        out = []
        for i in range(0, len(frames_nhwc), chunk):
            if self.emulate is not None:
                out.append(resnet_v2_50_emulated(frames_nhwc[i:i + chunk], self.w, self.emulate))
            else:
                out.append(resnet_v2_50(frames_nhwc[i:i + chunk], self.w, self.dtype))
        return torch.cat(out, dim=0)

    def movie_strips(self, phi_btc):
        if self.pred_mode == "pred":
            return az_fc2_groupnorm(phi_btc, self.w, self.num_conv_layers, self.dtype, self.emulate)
        if self.pred_mode == "hal":
            return fc2_res(phi_btc, self.w, self.dtype)
        raise Exception("Pred mode {} not recognized".format(self.pred_mode))

    def omegas(self, strips_nc):
        n = strips_nc.shape[0]
        mean = _t(self.w["mean_param"], self.dtype).reshape(1, 85).expand(n, 85)
        return call_hmr_ief(_t(strips_nc, self.dtype), mean, self.w, self.delta_t_values, self.dtype, self.emulate)

    def smpl_outputs(self, omega, cams):
        """One OmegasPred.compute_smpl (src/omega.py:263-304)."""
        verts, joints, Rs = smpl_forward(omega[:, 75:85], omega[:, 3:75], self.smpl, self.dtype)
        kps = batch_orth_proj_idrot(joints, cams)
        return {"cams": cams, "joints": joints, "kps": kps, "poses": Rs,
                "shapes": omega[:, 75:85], "verts": verts, "omegas": omega}

    def predict(self, images):
        """images [B,T,224,224,3] -> dict of float32 arrays (tester.py:229-258)."""
        B, T = images.shape[0], images.shape[1]
        phi = self.features(np.asarray(images).reshape(B * T, 224, 224, 3)).reshape(B, T, -1)
        strips = self.movie_strips(phi).reshape(B * T, -1)
        omega0, deltas = self.omegas(strips)
        res = {k: v.reshape((B, T) + v.shape[1:]) for k, v in
               self.smpl_outputs(omega0, omega0[:, :3]).items()}
        dres = [self.smpl_outputs(deltas[dt], omega0[:, :3]) for dt in sorted(self.delta_t_values)]
        if dres:
            for k in list(dres[0].keys()):                      # stacked on axis 2, tester.py:253
                res[k + "_delta"] = torch.stack(
                    [d[k].reshape((B, T) + d[k].shape[1:]) for d in dres], dim=2)
        return {k: v.to(torch.float32).numpy() for k, v in res.items()}

    def predict_all_images(self, all_images):
        """Sliding window, tester.py:260-312."""
        B, T = self.batch_size, self.sequence_length
        N = len(all_images)
        H = W = self.img_size
        margin = (self.fov - 1) // 2
        g = T - 2 * margin
        count = int(np.ceil(N / (g * B)))
        num_fill = count * B * g + T - N
        padded = np.concatenate((np.zeros((margin, H, W, 3), np.float32),
                                 np.asarray(all_images, np.float32),
                                 np.zeros((num_fill, H, W, 3), np.float32)), axis=0)
        results = {}
        for c in range(count):
            batch = np.stack([padded[i * g:i * g + T] for i in range(c * B, (c + 1) * B)])
            pred = self.predict(batch)
            for k, v in pred.items():
                results.setdefault(k, []).append(v)
        out = {}
        for k, v in results.items():
            v = np.array(v)[:, :, margin:-margin]
            out[k] = v.reshape((-1,) + v.shape[3:])[:N]
        return out


def window_plan(n_frames, batch_size, sequence_length, fov):
    """Index arithmetic of predict_all_images (tester.py:281-295), returned as
    integers so host code can be checked against it."""
    margin = (fov - 1) // 2
    g = sequence_length - 2 * margin
    count = int(math.ceil(n_frames / float(g * batch_size)))
    num_fill = count * batch_size * g + sequence_length - n_frames
    return margin, g, count, num_fill
