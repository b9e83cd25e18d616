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


class _ChStub(object):
    """Stand-in for chumpy.Ch while unpickling: the SMPL pickles store leaf arrays as chumpy
    objects whose state holds the ndarray under 'x'; `.r` is all the reference uses
    (batch_smpl.py:22-23)."""
    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {})

    @property
    def r(self):
        return np.asarray(self.__dict__["x"])


class _SmplUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.split(".")[0] == "chumpy":
            return _ChStub
        if module.startswith("scipy.sparse."):      # py2-era pickles name scipy.sparse.csc.csc_matrix: a deprecated module path
            import scipy.sparse
            return getattr(scipy.sparse, name)
        return super().find_class(module, name)


_CKPT_SMPL_VARS = ("v_template", "shapedirs", "J_regressor", "posedirs", "lbs_weights", "cocoplus_regressor")


def load_smpl_constants(path, checkpoint_vars=None):
    """'synthetic[:seed]' | .npz in the tf_smpl layout | SMPL .pkl (chumpy NOT required) |
    the SMPL variables the reference saves inside its checkpoint (non-trainable tf.Variables,
    batch_smpl.py:35-80), when `path` does not exist but `checkpoint_vars` has them."""
    if path is not None and str(path).startswith("synthetic"):
        seed = int(str(path).split(":")[1]) if ":" in str(path) else 2
        return assets.make_synthetic_smpl(seed)
    if path is None or not os.path.exists(path):
        if checkpoint_vars is not None and all(k in checkpoint_vars for k in _CKPT_SMPL_VARS):
            out = {k: np.asarray(checkpoint_vars[k], np.float32) for k in _CKPT_SMPL_VARS}
            out["parents"] = assets.SMPL_PARENTS.copy()      # kintree_table[0] of every SMPL model
            return out
        raise FileNotFoundError("SMPL model %s does not exist" % path)
    if str(path).endswith(".npz"):
        return {k: v for k, v in np.load(path).items()}
    with open(path, "rb") as f:
        dd = _SmplUnpickler(f, encoding="latin1").load()
    nb = _undo_chumpy(dd["shapedirs"]).shape[-1]
    posedirs = _undo_chumpy(dd["posedirs"])

    def dense_t(m):
        return np.asarray(m.T.todense() if hasattr(m, "todense") else np.asarray(m).T, np.float32)
    out = {
        "v_template": _undo_chumpy(dd["v_template"]).astype(np.float32),
        "shapedirs": np.reshape(_undo_chumpy(dd["shapedirs"]), [-1, nb]).T.astype(np.float32),
        "J_regressor": dense_t(dd["J_regressor"]),
        "posedirs": np.reshape(posedirs, [-1, posedirs.shape[-1]]).T.astype(np.float32),
        "parents": np.asarray(dd["kintree_table"])[0].astype(np.int64).astype(np.int32),
        "lbs_weights": _undo_chumpy(dd["weights"]).astype(np.float32),
        "cocoplus_regressor": dense_t(dd["cocoplus_regressor"]),
    }
    # The reference creates these six tensors as (non-trainable) tf.Variables, so `Saver.restore` of a checkpoint that
    # holds them OVERWRITES the pkl values (batch_smpl.py:35-80 + tester.py:92-116): the checkpoint wins there, and here.
    if checkpoint_vars is not None and all(k in checkpoint_vars for k in _CKPT_SMPL_VARS):
        for k in _CKPT_SMPL_VARS:
            v = np.asarray(checkpoint_vars[k], np.float32)
            if v.shape != out[k].shape:
                raise ValueError("checkpoint variable %s has shape %s, the SMPL model %s" % (k, v.shape, out[k].shape))
            out[k] = v
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
