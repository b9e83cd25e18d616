"""Hand-off of the path's outputs to a rasteriser.

The reference renders each frame with pytorch-NMR (`src/util/render/nmr_renderer.py`); what the
rasteriser itself receives is `proj_verts` (`VisRenderer.__call__`, :139-144) computed from the
vertices and a camera that `visualize_img_orig` (:333-409) first moves from the 224x224 crop to the
squared original image.  The reference does this per frame in NumPy after the whole prediction
dict has crossed PCIe; here `rasteriser_inputs` does it for all frames in one launch of
`hmmr_render_handoff`, reading cams / verts / kps in place inside the packed per-frame records
(`Tester.predict_records`, `dist.record_layout`), so nothing leaves the device before the
rasteriser.  NMR itself (third-party CUDA, PyTorch 0.4) is out of scope; any rasteriser taking
(proj_verts [n,V,3], faces [F,3]) can consume the result.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ... import _lib as L


def orig_image_geometry(image_og_params, orig_shape, max_img_size=300):
    """One `geom` row {undo_scale, start_x, start_y, proc_size, img_size} from the per-frame dict of
    `process_image(s)` (`start_pt`, `scale`, `im_shape`) and the original frame's (H, W): the scalars
    of visualize_img_orig (nmr_renderer.py:356-374): optional down-scale to max_img_size with
    resize_img's floor (common.py:8), then make_square's padding to the longer side."""
    h, w = int(orig_shape[0]), int(orig_shape[1])
    scale = float(np.asarray(image_og_params["scale"]).reshape(-1)[0])
    if max(h, w) > max_img_size:
        scale_orig = max_img_size / float(max(h, w))
        h, w = int(np.floor(h * scale_orig)), int(np.floor(w * scale_orig))     # resize_img
        undo_scale = (1.0 / scale) * scale_orig
    else:
        undo_scale = 1.0 / scale
    start = np.asarray(image_og_params["start_pt"], np.float64).reshape(2)
    proc = float(image_og_params["im_shape"][0])
    return np.array([undo_scale, start[0], start[1], proc, float(max(h, w))], np.float64)


def rasteriser_inputs(cams, verts, kps=None, geom=None, stream=None):
    """cams [n,>=3], verts [n,V,3], kps [n,K,2] (device float32; any row stride, e.g. views into the
    packed records) -> dict(cams [n,3], proj_verts [n,V,3], kps [n,K,2]) on the device.
    geom: [n,5] rows of `orig_image_geometry` (original-image rendering) or None (crop rendering)."""
    lib = L.load()
    dev = verts.device
    if dev.type != "cuda":
        raise L.HmmrError("rasteriser_inputs needs device tensors (the HIP library has no CPU path)")
    n, nv = verts.shape[0], verts.shape[1]

    def rows(t):
        t2 = t.reshape(n, -1) if t.dim() > 2 else t
        if t2.dtype != torch.float32 or t2.stride(1) != 1:
            t2 = t2.float().contiguous()
        return t2
    cams2, verts2 = rows(cams), rows(verts)
    kps2 = rows(kps) if kps is not None else None
    nk = kps2.shape[1] // 2 if kps2 is not None else 0
    g = None
    if geom is not None:
        g = torch.as_tensor(np.asarray(geom, np.float64).reshape(n, 5), dtype=torch.float32).to(dev)
    new_cam = torch.empty((n, 3), dtype=torch.float32, device=dev)
    proj = torch.empty((n, nv, 3), dtype=torch.float32, device=dev)
    kp_out = torch.empty((n, nk, 2), dtype=torch.float32, device=dev) if nk else None
    st = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream
    L.check(lib.hmmr_render_handoff(
        cams2.data_ptr(), cams2.stride(0), verts2.data_ptr(), verts2.stride(0),
        kps2.data_ptr() if nk else None, kps2.stride(0) if nk else 0,
        g.data_ptr() if g is not None else None, n, nv, nk,
        new_cam.data_ptr(), proj.data_ptr(), kp_out.data_ptr() if nk else None, st), "hmmr_render_handoff")
    return {"cams": new_cam, "proj_verts": proj, "kps": kp_out}


def rasteriser_inputs_from_records(records, layout, geom=None):
    """Same, straight from packed per-frame records [n, rec_len] (`Tester.predict_records`,
    `dist.record_layout`): the present-frame container's cams / verts / kps are read in place."""
    get = {k: (off, size, shp) for k, shp, off, size in layout}
    oc, sc, _ = get["cams"]
    ov, sv, shp_v = get["verts"]
    ok, sk, _ = get["kps"]
    return rasteriser_inputs(records[:, oc:oc + sc], records[:, ov:ov + sv].unflatten(1, tuple(shp_v)),
                             records[:, ok:ok + sk], geom)
