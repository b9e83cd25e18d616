"""Generate the committed golden fixtures from the CPU oracle (float64).

    python tests/golden/make_golden.py

The reference itself cannot run here (no TensorFlow), so these vectors pin the
ORACLE, not the reference (parity unpinned, see oracle/hmmr_oracle.py).  Inputs
are regenerated from seeds (human_dynamics_amd.assets), only outputs are stored.

  window_b1_t20.npz     BASELINE config 1: one [1,20,224,224,3] window through
                        predict(); per-stage intermediates phi / strips / omegas
  video_n24_b2_t20.npz  24-frame video through predict_all_images(), B=2, T=20
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from human_dynamics_amd import assets          # noqa: E402
from oracle import hmmr_oracle as O            # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
VSUB = 16      # vertex sub-sampling of the delta meshes (keeps the fixtures small)


def pack(res, out):
    for k, v in res.items():
        if k == "verts_delta":
            out[k + "_sub"] = v[..., ::VSUB, :].astype(np.float32)
        else:
            out[k] = v.astype(np.float32)


def main():
    w = assets.make_synthetic_weights(0)
    s = assets.make_synthetic_smpl(2)
    # ---- config 1 -----------------------------------------------------------
    frames = assets.make_synthetic_frames(20, seed=1)
    T = O.OracleTester(w, s, batch_size=1, sequence_length=20, dtype=torch.float64)
    phi = T.features(frames)
    strips = T.movie_strips(phi.reshape(1, 20, -1)).reshape(20, -1)
    om0, deltas = T.omegas(strips)
    out = {"phi": phi.numpy().astype(np.float32), "strips": strips.numpy().astype(np.float32),
           "omegas_all": torch.stack([om0] + [deltas[k] for k in sorted(deltas)]).numpy().astype(np.float32)}
    pack(T.predict(frames[None]), out)
    np.savez_compressed(os.path.join(HERE, "window_b1_t20.npz"), **out)
    # ---- short video through the sliding window ------------------------------
    frames = assets.make_synthetic_frames(24, seed=7)
    T = O.OracleTester(w, s, batch_size=2, sequence_length=20, dtype=torch.float64)
    res = T.predict_all_images(frames)
    out = {}
    pack(res, out)
    out["verts_sub"] = out.pop("verts")[:, ::VSUB]
    np.savez_compressed(os.path.join(HERE, "video_n24_b2_t20.npz"), **out)
    for f in ("window_b1_t20.npz", "video_n24_b2_t20.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
