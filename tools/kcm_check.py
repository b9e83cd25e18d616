"""Chunk-major K order (hmmr_conv_desc_t.k_order = 1) against the tap-major order on the 3x3 shapes of blocks 2-4
(dev aid; FIRST thing to run on the GPU next round -- the kernel path was written without GPU minutes left):

    python tools/kcm_check.py [frames]

Same products in another summation order: the two outputs agree to fp32 rounding (max |a - b| / max |a| ~ 1e-6); the
timings show whether the L2 locality pays (FETCH_SIZE / TCC hit rate: tools/pmc_conv_study.sh with KORD=1)."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from human_dynamics_amd import _lib as L
from human_dynamics_amd import engine as E
from human_dynamics_amd import packing

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = np.random.default_rng(0)
bad = 0
for h, c, stride in ((28, 128, 1), (28, 128, 2), (14, 256, 1), (14, 256, 2), (7, 512, 1)):
    x = (rng.normal(size=(n, h, h, c)) * 0.5).astype(np.float32)
    w = (rng.normal(size=(3, 3, c, c)) / (9 * c) ** 0.5).astype(np.float32)
    sc = (rng.random(c) + 0.5).astype(np.float32)
    sh = rng.normal(size=c).astype(np.float32)
    xs = packing.to_split(torch.from_numpy(x).cuda())
    for tile in (1, 2, 3, 5, 6, 7, 8):
        if tile == 8 and c % 256:
            continue
        out = []
        for ko in (0, 1):
            o, _ = E.conv_gemm(xs, w, stride=stride, pad=1, scale=sc, shift=sh, relu=True, in_dtype=L.HMMR_BF16X3,
                               out_dtype=L.HMMR_BF16X3, tile=tile, k_order=ko)
            out.append(np.asarray(o, np.float64))
        err = np.abs(out[0] - out[1]).max() / max(np.abs(out[0]).max(), 1e-30)
        ok = err < 2e-5
        bad += not ok
        print("h %2d c %3d s %d tile %d: rel diff %.2e %s" % (h, c, stride, tile, err, "ok" if ok else "MISMATCH"))
print("FAILED" if bad else "all equal to rounding")
sys.exit(1 if bad else 0)
