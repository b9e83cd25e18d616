"""Mirror of src/tf_smpl/batch_lbs.py for device tensors.

`batch_rodrigues` is evaluated by smpl_pose_kernel, `batch_global_rigid_transformation` by
smpl_fk_kernel (csrc/smpl.hip: the same chain arithmetic, exposed for arbitrary Rs / Js); the
functions here carry the reference's signatures.
"""
from __future__ import annotations

import numpy as np
import torch


def batch_rodrigues(theta, engine):
    """theta [N,3] -> R [N,3,3] (src/tf_smpl/batch_lbs.py:42-60).  N must be a
    multiple of 24 (whole poses), as on the reference's hot path."""
    theta = engine.to_device(theta).reshape(-1, 72)
    beta = torch.zeros((theta.shape[0], 10), dtype=torch.float32, device=engine.device)
    _, _, _, rs = engine.smpl(theta, beta, None, want_rs=True)
    return rs.reshape(-1, 3, 3)


def batch_global_rigid_transformation(Rs, Js, parent, rotate_base=False, engine=None):
    """Rs [N,24,3,3], Js [N,24,3], parent [24] -> (new_J [N,24,3], A [N,24,4,4]): absolute joint
    locations and the relative joint transforms for LBS (src/tf_smpl/batch_lbs.py:133-194).
    rotate_base: the root rotation is multiplied by diag(1, -1, -1) (batch_lbs.py:151-158; False on the hot path,
    batch_smpl.py:136)."""
    import ctypes as C  # noqa: F401
    from .. import _lib as L
    lib = L.load()
    dev = engine.device if engine is not None else torch.device("cuda:0")
    to = (engine.to_device if engine is not None else
          (lambda a: torch.as_tensor(np.asarray(a, np.float32)).to(dev).contiguous()))
    Rs, Js = to(Rs).reshape(-1, 24, 3, 3).contiguous(), to(Js).reshape(-1, 24, 3).contiguous()
    par = torch.as_tensor(np.asarray(parent).astype(np.int32)).to(dev)
    n = Rs.shape[0]
    new_j = torch.empty((n, 24, 3), dtype=torch.float32, device=dev)
    A = torch.empty((n, 24, 4, 4), dtype=torch.float32, device=dev)
    L.check(lib.hmmr_global_rigid_transformation(Rs.data_ptr(), Js.data_ptr(), par.data_ptr(), n, new_j.data_ptr(),
                                                 A.data_ptr(), int(bool(rotate_base)), torch.cuda.current_stream(dev).cuda_stream),
            "hmmr_global_rigid_transformation")
    return new_j, A
