"""Mirror of src/tf_smpl/projection.py:16-29 (batch_orth_proj_idrot).

On the hot path the projection is fused into smpl_joints_kernel; this
standalone form exists for callers that project other point sets.
"""
from __future__ import annotations

import torch


def batch_orth_proj_idrot(X, camera):
    """X [N,P,3], camera [N,3] = [s, tx, ty] -> [N,P,2] = s * (X_xy + t)."""
    camera = camera.reshape(-1, 1, 3)
    return camera[:, :, 0:1] * (X[:, :, :2] + camera[:, :, 1:])
