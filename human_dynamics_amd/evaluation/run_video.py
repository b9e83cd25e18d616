"""Device mirror of the crop in src/evaluation/run_video.py (`process_image`, :56-107).

    images, infos = process_images(frames_uint8[n,H,W,3], bbox_params[n,3])

returns the [n,224,224,3] float32 crops in [-1,1] as a DEVICE tensor (ready for
Tester.predict_all_images / ShardedPredictor) plus, per frame, the dict fields of the reference
(`im_shape`, `center`, `scale`, `start_pt`).  Decoding (imread) stays with the caller.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib as L

IMG_SIZE = 224


def crop_geometry(h, w, bbox_param):
    """The integers of process_image / resize_img for one frame, in the reference's float64
    arithmetic: scaled size, centre after scaling (x uses the HEIGHT factor and y the WIDTH factor,
    as `center * scale_factors` does at run_video.py:74), crop origin."""
    center = np.asarray(bbox_param[:2], np.float64)
    scale = float(bbox_param[2])
    new_size = (np.floor(np.array([h, w]) * scale)).astype(int)                 # common.py:8
    factors = [new_size[0] / float(h), new_size[1] / float(w)]
    center_scaled = np.round(center * factors).astype(int) + IMG_SIZE           # in the padded image
    start_pt = center_scaled - IMG_SIZE // 2
    hp, wp = new_size[0] + 2 * IMG_SIZE, new_size[1] + 2 * IMG_SIZE
    end_pt = np.array([min(center_scaled[0] + IMG_SIZE // 2, wp), min(center_scaled[1] + IMG_SIZE // 2, hp)])
    if new_size.min() < 1 or (end_pt - start_pt != IMG_SIZE).any() or (start_pt < 0).any():
        raise ValueError("bbox %s does not yield a full 224x224 crop of a %dx%d frame" % (bbox_param, h, w))
    return {"hs": int(new_size[0]), "ws": int(new_size[1]), "u0": int(start_pt[0] - IMG_SIZE),
            "v0": int(start_pt[1] - IMG_SIZE), "center": center_scaled - start_pt, "scale": scale,
            "start_pt": start_pt, "im_shape": [IMG_SIZE, IMG_SIZE]}


def process_images(frames, bbox_params, device="cuda:0"):
    lib = L.load()
    if isinstance(frames, torch.Tensor):
        fr = frames.to(device).contiguous()
    else:
        fr = torch.from_numpy(np.ascontiguousarray(frames)).to(device)
    assert fr.dtype == torch.uint8 and fr.dim() == 4 and fr.shape[3] == 3, "frames: [n,H,W,3] uint8 (RGB)"
    n, h, w = fr.shape[:3]
    infos = [crop_geometry(h, w, bp) for bp in np.asarray(bbox_params, np.float64)]
    geom = torch.tensor([[g["hs"], g["ws"], g["u0"], g["v0"]] for g in infos], dtype=torch.int32).to(device)
    out = torch.empty((n, IMG_SIZE, IMG_SIZE, 3), dtype=torch.float32, device=device)
    L.check(lib.hmmr_crop_frames(fr.data_ptr(), geom.data_ptr(), n, h, w, out.data_ptr(),
                                 torch.cuda.current_stream(device).cuda_stream), "hmmr_crop_frames")
    return out, [{k: g[k] for k in ("im_shape", "center", "scale", "start_pt")} for g in infos]
