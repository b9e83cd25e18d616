"""Mirror of src/tf_smpl/batch_smpl.py: ``SMPL(pkl_path, joint_type)`` and
``smpl(beta, theta, get_skin)`` backed by the HIP SMPL stage.

The constants keep the reference's parameter layout (batch_smpl.py:35-80):
v_template [6890,3], shapedirs [10,20670], J_regressor [6890,24] (transposed),
posedirs [207,20670], lbs_weights [6890,24], cocoplus_regressor [6890,K]
(transposed), parents int32[24].
"""
from __future__ import annotations

import os
import pickle

import numpy as np
import torch

from .. import assets
from ..engine import HmmrEngine


def _undo_chumpy(x):
    return x if isinstance(x, np.ndarray) else np.asarray(x.r)


def load_smpl_constants(path):
    """'synthetic[:seed]' | .npz in the tf_smpl layout | SMPL .pkl (needs chumpy
    importable for the official pickles, like the reference)."""
    if path is None or str(path).startswith("synthetic"):
        seed = int(str(path).split(":")[1]) if path and ":" in str(path) else 2
        return assets.make_synthetic_smpl(seed)
    if not os.path.exists(path):
        raise FileNotFoundError("SMPL model %s does not exist" % path)
    if path.endswith(".npz"):
        return {k: v for k, v in np.load(path).items()}
    with open(path, "rb") as f:
        dd = pickle.load(f, encoding="latin1")
    nb = dd["shapedirs"].shape[-1]
    out = {
        "v_template": _undo_chumpy(dd["v_template"]).astype(np.float32),
        "shapedirs": np.reshape(_undo_chumpy(dd["shapedirs"]), [-1, nb]).T.astype(np.float32),
        "J_regressor": np.asarray(dd["J_regressor"].T.todense(), np.float32),
        "posedirs": np.reshape(_undo_chumpy(dd["posedirs"]), [-1, dd["posedirs"].shape[-1]]).T.astype(np.float32),
        "parents": dd["kintree_table"][0].astype(np.int32),
        "lbs_weights": _undo_chumpy(dd["weights"]).astype(np.float32),
        "cocoplus_regressor": np.asarray(dd["cocoplus_regressor"].T.todense(), np.float32),
    }
    return out


class SMPL(object):
    def __init__(self, pkl_path, joint_type="cocoplus", dtype=None, engine=None, device="cuda:0"):
        if joint_type not in ("cocoplus", "lsp"):
            raise ValueError('Unknown joint type: %s, it must be either "cocoplus" or "lsp"' % joint_type)
        consts = pkl_path if isinstance(pkl_path, dict) else load_smpl_constants(pkl_path)
        self.consts = consts
        self.joint_type = joint_type
        self.size = [consts["v_template"].shape[0], 3]
        self.num_betas = consts["shapedirs"].shape[0]
        self.parents = np.asarray(consts["parents"]).astype(np.int32)
        # an engine that only carries the SMPL constants unless one is shared in
        self.engine = engine if engine is not None else HmmrEngine(None, consts, device=device,
                                                                   joint_type=joint_type)

    def __call__(self, beta, theta, get_skin=False, name=None):
        """beta [N,10], theta [N,72] or [N,24,3] -> joints [N,K,3]
        (+ verts [N,6890,3], Rs [N,24,3,3] if get_skin), as device tensors."""
        e = self.engine
        theta = e.to_device(theta).reshape(-1, 72)
        beta = e.to_device(beta).reshape(-1, 10)
        verts, joints, _, rs = e.smpl(theta, beta, None, want_rs=True)
        if get_skin:
            return verts, joints, rs
        return joints
