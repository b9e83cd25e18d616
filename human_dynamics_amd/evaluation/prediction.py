"""Prediction cache after the path: the on-disk wire format the reference's
evaluation and demo drivers exchange predictions in (`src/evaluation/prediction.py`).

Layout (prediction.py:22-61):  <pred_dir>/<basename(load_path)>/<dataset>-<video>-P<p_id>.pkl holds
every output of `predict_all_images` except the vertices plus the bookkeeping keys `tf_path` and
`p_id`; `...-P<p_id>-verts.pkl` holds the `verts*` arrays.  Both are plain pickled dicts of float32
ndarrays, so files written by either implementation load in the other.
"""
from __future__ import annotations

import os
import pickle

import numpy as np

PRED_DIR = "predictions_cache"


def _cache_dir(pred_dir, load_path):
    out = os.path.join(pred_dir, os.path.basename(load_path))
    os.makedirs(out, exist_ok=True)
    return out


def get_pred_path_name(load_path, tf_path, p_id, pred_dir=PRED_DIR, incl_verts=False):
    """(path, file name) of one tube's cached predictions (prediction.py:22-61): the dataset is the
    directory two levels above the tfrecord, the video id its basename without `.tfrecord`."""
    video = os.path.basename(tf_path).replace(".tfrecord", "")
    dataset = os.path.basename(os.path.dirname(os.path.dirname(tf_path)))
    name = "%s-%s-P%s%s.pkl" % (dataset, video, p_id, "-verts" if incl_verts else "")
    return os.path.join(_cache_dir(pred_dir, load_path), name), name


def get_result_path_name(split, load_path, pred_mode, datasets, pred_dir=PRED_DIR):
    """prediction.py:64-80."""
    name = "results_%s_%s_%s.json" % (split, pred_mode, "-".join(datasets))
    return os.path.join(pred_dir, os.path.basename(load_path), name)


def get_eval_path_name(load_path, pred_mode, tf_path, p_id, pred_dir=PRED_DIR, min_visible=0):
    """prediction.py:83-102."""
    video = os.path.basename(tf_path).replace(".tfrecord", "")
    name = "results_%s_%s_P%s" % (pred_mode, video, p_id)
    if min_visible > 0:
        name += "_min-vis%s" % min_visible
    return os.path.join(pred_dir, os.path.basename(load_path), name) + ".pkl"


def split_preds(preds):
    """(everything else, vertices): keys containing 'vert' go to the second dict (prediction.py:105-116)."""
    rest = {k: v for k, v in preds.items() if "vert" not in k}
    verts = {k: v for k, v in preds.items() if "vert" in k}
    return rest, verts


def get_predictions(model, images, load_path, tf_path, p_id, pred_dir=PRED_DIR, incl_verts=False):
    """Load one tube's predictions from the cache, or run `model.predict_all_images` and fill it
    (prediction.py:119-165).  Images in [0, 255] are mapped to [-1, 1] first, as the reference does."""
    pred_path, _ = get_pred_path_name(load_path, tf_path, p_id, pred_dir, incl_verts=False)
    vert_path, _ = get_pred_path_name(load_path, tf_path, p_id, pred_dir, incl_verts=True)
    if os.path.exists(pred_path) and (not incl_verts or os.path.exists(vert_path)):
        with open(pred_path, "rb") as f:
            preds = pickle.load(f)
        if incl_verts:
            with open(vert_path, "rb") as f:
                preds.update(pickle.load(f))
        return preds
    if np.max(images) > 1.1:
        images = (np.array(images) / 255) * 2 - 1
    preds = model.predict_all_images(images)
    preds.update({"tf_path": tf_path, "p_id": p_id})
    preds, verts = split_preds(preds)
    with open(pred_path, "wb") as f:
        pickle.dump(preds, f)
    if incl_verts:
        preds.update(verts)
        with open(vert_path, "wb") as f:
            pickle.dump(verts, f)
    return preds
