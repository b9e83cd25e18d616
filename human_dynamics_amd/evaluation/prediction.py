"""Prediction cache after the path: the on-disk wire format the reference's
evaluation and demo drivers exchange predictions in (`src/evaluation/prediction.py`).

Layout (prediction.py:22-61):  <pred_dir>/<basename(load_path)>/<dataset>-<video>-P<p_id>.pkl holds
every output of `predict_all_images` except the vertices plus the bookkeeping keys `tf_path` and
`p_id`; `...-P<p_id>-verts.pkl` holds the `verts*` arrays.  Both are plain pickled dicts of float32
ndarrays, so files written by either implementation load in the other.

`TubeCache` is the implementation; the module-level functions keep the reference's names and
argument order for its callers (`eval.py`, `demo_video.py`).
"""
from __future__ import annotations

import os
import pickle

import numpy as np

PRED_DIR = "predictions_cache"


def _stem(tf_path):
    return os.path.basename(tf_path).replace(".tfrecord", "")


def _model_dir(pred_dir, load_path):
    return os.path.join(pred_dir, os.path.basename(load_path))


class TubeCache(object):
    """The two pickle files of one (model, tfrecord, person tube)."""

    def __init__(self, load_path, tf_path, p_id, pred_dir=PRED_DIR):
        # the dataset name is the directory two levels above the tfrecord (prediction.py:44-46)
        dataset = os.path.basename(os.path.dirname(os.path.dirname(tf_path)))
        self.tf_path, self.p_id = tf_path, p_id
        self.folder = _model_dir(pred_dir, load_path)
        self.base = "%s-%s-P%s" % (dataset, _stem(tf_path), p_id)

    def name(self, verts=False):
        return self.base + ("-verts" if verts else "") + ".pkl"

    def path(self, verts=False):
        os.makedirs(self.folder, exist_ok=True)
        return os.path.join(self.folder, self.name(verts))

    def complete(self, need_verts):
        return os.path.exists(self.path()) and (not need_verts or os.path.exists(self.path(True)))

    def load(self, with_verts):
        with open(self.path(), "rb") as f:
            out = pickle.load(f)
        if with_verts:
            with open(self.path(True), "rb") as f:
                out.update(pickle.load(f))
        return out

    def store(self, preds, with_verts):
        """Writes the vertex-free dict always and the vertices on request; returns what the
        reference returns (vertices merged back only if they were asked for)."""
        small, verts = split_preds(dict(preds, tf_path=self.tf_path, p_id=self.p_id))
        with open(self.path(), "wb") as f:
            pickle.dump(small, f)
        if with_verts:
            with open(self.path(True), "wb") as f:
                pickle.dump(verts, f)
            small.update(verts)
        return small


def split_preds(preds):
    """(everything else, vertices): keys containing 'vert' go to the second dict (prediction.py:105-116)."""
    verts = {k: preds[k] for k in preds if "vert" in k}
    return {k: preds[k] for k in preds if k not in verts}, verts


def get_pred_path_name(load_path, tf_path, p_id, pred_dir=PRED_DIR, incl_verts=False):
    """(path, file name) of one tube's cached predictions (prediction.py:22-61)."""
    c = TubeCache(load_path, tf_path, p_id, pred_dir)
    return c.path(incl_verts), c.name(incl_verts)


def get_result_path_name(split, load_path, pred_mode, datasets, pred_dir=PRED_DIR):
    """prediction.py:64-80."""
    return os.path.join(_model_dir(pred_dir, load_path), "results_%s_%s_%s.json" % (split, pred_mode, "-".join(datasets)))


def get_eval_path_name(load_path, pred_mode, tf_path, p_id, pred_dir=PRED_DIR, min_visible=0):
    """prediction.py:83-102."""
    tail = "_min-vis%s" % min_visible if min_visible > 0 else ""
    return os.path.join(_model_dir(pred_dir, load_path), "results_%s_%s_P%s%s.pkl" % (pred_mode, _stem(tf_path), p_id, tail))


def get_predictions(model, images, load_path, tf_path, p_id, pred_dir=PRED_DIR, incl_verts=False):
    """One tube's predictions: from the cache when both needed files exist, otherwise from
    `model.predict_all_images` (filling the cache), prediction.py:119-165.  Frames given in
    [0, 255] are mapped to [-1, 1] first, as the reference does."""
    cache = TubeCache(load_path, tf_path, p_id, pred_dir)
    if cache.complete(incl_verts):
        return cache.load(incl_verts)
    frames = np.asarray(images)
    if frames.max() > 1.1:
        frames = (frames / 255) * 2 - 1
    return cache.store(model.predict_all_images(frames), incl_verts)
