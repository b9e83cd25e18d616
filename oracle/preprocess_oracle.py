"""CPU restatement of the crop that precedes the hot path -- TEST INFRASTRUCTURE ONLY.

`process_image` (src/evaluation/run_video.py:56-107): uint8 frame -> [-1, 1] float64 ->
`resize_img` (src/util/common.py:7-14: cv2.resize to floor(shape * scale), bilinear) -> np.pad(224,
mode='edge') -> 224 x 224 crop around round(center * scale_factors).

PARITY: the pad / crop / rounding logic is pinned to the reference (its `process_image` is executed
in tests/golden/make_reference_golden.py); `cv2_resize_linear` below RESTATES OpenCV's INTER_LINEAR
for floating-point images (cv2 is not installable here): pixel-centre alignment
sx = (dx + 0.5) * (src / dst) - 0.5 evaluated in float32, floor, clamp to the border, float32
weights (1 - f, f), accumulation in float64 -- "parity unpinned" for that one function.
"""
from __future__ import annotations

import numpy as np

IMG_SIZE = 224


def _taps(src, dst):
    scale = float(src) / float(dst)
    d = np.arange(dst)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo], s[lo] = 0.0, 0
    hi = s >= src - 1
    f[hi], s[hi] = 0.0, src - 1
    s1 = np.minimum(s + 1, src - 1)
    return s, s1, (np.float32(1.0) - f).astype(np.float64), f.astype(np.float64)


def cv2_resize_linear(img, dsize):
    """cv2.resize(img, (width, height)) with the default INTER_LINEAR, float images."""
    w, h = int(dsize[0]), int(dsize[1])
    H, W = img.shape[:2]
    x0, x1, a0, a1 = _taps(W, w)
    y0, y1, b0, b1 = _taps(H, h)
    img = img.astype(np.float64)
    rows = img[:, x0] * a0[None, :, None] + img[:, x1] * a1[None, :, None]      # horizontal pass
    return rows[y0] * b0[:, None, None] + rows[y1] * b1[:, None, None]          # vertical pass


def resize_img(img, scale_factor):
    """src/util/common.py:7-14."""
    new_size = (np.floor(np.array(img.shape[0:2]) * scale_factor)).astype(int)
    new_img = cv2_resize_linear(img, (new_size[1], new_size[0]))
    actual_factor = [new_size[0] / float(img.shape[0]), new_size[1] / float(img.shape[1])]
    return new_img, actual_factor


def process_image(image_u8, bbox_param):
    """run_video.py:56-107 on an already decoded uint8 frame.  Returns the crop and the dict fields."""
    center = np.asarray(bbox_param[:2], np.float64)
    scale = float(bbox_param[2])
    image = ((image_u8 / 255.) - 0.5) * 2
    image_scaled, scale_factors = resize_img(image, scale)
    center_scaled = np.round(center * scale_factors).astype(int)      # (sic) x * height factor, y * width factor
    image_padded = np.pad(image_scaled, ((IMG_SIZE,), (IMG_SIZE,), (0,)), mode="edge")
    height, width = image_padded.shape[:2]
    center_scaled += IMG_SIZE
    margin = IMG_SIZE // 2
    start_pt = (center_scaled - margin).astype(int)
    end_pt = (center_scaled + margin).astype(int)
    end_pt[0] = min(end_pt[0], width)
    end_pt[1] = min(end_pt[1], height)
    crop = image_padded[start_pt[1]:end_pt[1], start_pt[0]:end_pt[0], :]
    center_scaled -= start_pt
    return {"image": crop, "im_shape": list(crop.shape[:2]), "center": center_scaled, "scale": scale,
            "start_pt": start_pt}
