"""ORACLE (test infrastructure only -- never imported by the product path).

NumPy restatement of the reference's hand-off to the rasteriser:
  * `orig_camera`      src/util/render/nmr_renderer.py:356-401 (visualize_img_orig): the camera and the
                       2-D keypoints moved from the 224x224 crop to the squared original image;
  * `project`          src/util/render/nmr_renderer.py:139-144 + torch_utils.py:11-29
                       (orthographic_proj_withz_idrot, then y *= -1).
PARITY: PINNED -- checked against tests/golden/reference_handoff.npz, which
tests/golden/make_handoff_golden.py produced by executing the reference's own functions.
(The reference's intermediate precision depends on the NumPy version -- float32 scalars stay float32
under NumPy 2 -- so the pin is to 1e-6 relative, not bitwise; float64 throughout here.)
"""
import numpy as np


def squared_size(h, w, max_img_size):
    """Side of the squared (make_square) original image after the optional down-scale (resize_img's floor)."""
    if max(h, w) > max_img_size:
        s = max_img_size / float(max(h, w))
        h, w = int(np.floor(h * s)), int(np.floor(w * s))
        return max(h, w), s
    return max(h, w), 1.0


def orig_camera(cam, kp_pred, start_pt, scale, proc_size, orig_shape, max_img_size=300):
    cam = np.asarray(cam, np.float64)
    img_size, scale_orig = squared_size(orig_shape[0], orig_shape[1], max_img_size)
    undo = (1.0 / float(scale)) * scale_orig
    start_pt = np.asarray(start_pt, np.float64)
    pred_joint = ((np.asarray(kp_pred, np.float64) + 1) * 0.5) * proc_size
    pred_joint_orig = (pred_joint + start_pt - proc_size) * undo
    kp_orig = 2 * (pred_joint_orig / img_size) - 1
    cam_crop = np.hstack([proc_size * cam[0] * 0.5, cam[1:] + (2.0 / cam[0]) * 0.5])
    cam_orig = np.hstack([cam_crop[0] * undo, cam_crop[1:] + (start_pt - proc_size) / cam_crop[0]])
    new_cam = np.hstack([cam_orig[0] * (2.0 / img_size), cam_orig[1:] - (1 / ((2.0 / img_size) * cam_orig[0]))])
    return new_cam, kp_orig, img_size


def project(verts, cam):
    """verts [n,V,3], cam [n,3] -> [n,V,3] = [s(x+tx), -s(y+ty), z]."""
    verts = np.asarray(verts)
    cam = np.asarray(cam, verts.dtype)
    out = verts.copy()
    out[..., :2] = cam[:, None, :1] * (verts[..., :2] + cam[:, None, 1:3])
    out[..., 1] *= -1
    return out
