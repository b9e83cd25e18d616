"""Mirror of the error metrics of src/evaluation/eval_util.py, evaluated on the device
(csrc/eval_metrics.hip) so predictions can be scored where the SMPL stage left them.

Same function names, arguments and return values as the reference (lists / arrays of per-frame
errors); inputs may be NumPy arrays or device tensors.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _lib as L

LEFT_HIP, RIGHT_HIP = 3, 2            # LSP order, eval_util.py:166-167


def _dev(x, device):
    if isinstance(x, torch.Tensor):
        return x.to(device, torch.float32).contiguous()
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(device)


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


def _joint_metrics(gt, pred, device, want_err=True, want_accel=True):
    lib = L.load()
    pred = _dev(pred, device)
    n, k = pred.shape[0], pred.shape[1]
    gt = _dev(gt, device) if gt is not None else None
    mp = torch.empty(n, device=device) if (want_err and gt is not None) else None
    pa = torch.empty(n, device=device) if (want_err and gt is not None) else None
    ac = torch.empty(max(n - 2, 0), device=device) if want_accel else None
    ae = torch.empty(max(n - 2, 0), device=device) if (want_accel and gt is not None) else None
    L.check(lib.hmmr_eval_joints(L.ptr(gt), pred.data_ptr(), n, k, LEFT_HIP, RIGHT_HIP, L.ptr(mp), L.ptr(pa),
                                 L.ptr(ac), L.ptr(ae), _stream(device)), "hmmr_eval_joints")
    return mp, pa, ac, ae


def compute_accel(joints, device="cuda:0"):
    """Acceleration of 3D joints (Nx25x3) -> (N-2) (eval_util.py:14-27)."""
    return _joint_metrics(None, joints, device, want_err=False)[2].cpu().numpy()


def compute_error_3d(gt3ds, preds, vis=None, device="cuda:0"):
    """MPJPE after pelvis alignment and after Procrustes, per visible frame (eval_util.py:30-60)."""
    assert len(gt3ds) == len(preds)
    mp, pa, _, _ = _joint_metrics(np.asarray(gt3ds).reshape(len(gt3ds), -1, 3) if not isinstance(gt3ds, torch.Tensor)
                                  else gt3ds, preds, device, want_accel=False)
    mp, pa = mp.cpu().numpy(), pa.cpu().numpy()
    keep = np.ones(len(mp), bool) if vis is None else np.asarray(vis).astype(bool)
    return list(mp[keep]), list(pa[keep])


def compute_error_accel(joints_gt, joints_pred, vis=None, device="cuda:0"):
    """Acceleration error per interior frame, dropping every frame whose 3-frame stencil touches an
    invisible frame (eval_util.py:63-94)."""
    ae = _joint_metrics(joints_gt, joints_pred, device, want_err=False)[3].cpu().numpy()
    if vis is None:
        new_vis = np.ones(len(ae), dtype=bool)
    else:
        invis = np.logical_not(np.asarray(vis).astype(bool))
        new_invis = np.logical_or(invis, np.logical_or(np.roll(invis, -1), np.roll(invis, -2)))[:-2]
        new_vis = np.logical_not(new_invis)
    return ae[new_vis]


def compute_error_verts(verts_gt, verts_pred, device="cuda:0"):
    """Mean per-vertex distance per frame (eval_util.py:140-155)."""
    lib = L.load()
    g, p = _dev(verts_gt, device), _dev(verts_pred, device)
    assert g.shape == p.shape
    n, nv = g.shape[0], g.shape[1]
    out = torch.empty(n, device=device)
    L.check(lib.hmmr_eval_verts(g.data_ptr(), nv * 3, p.data_ptr(), nv * 3, n, nv, out.data_ptr(), _stream(device)),
            "hmmr_eval_verts")
    return out.cpu().numpy()
