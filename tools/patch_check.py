"""Dev check of the 3x3 patch kernel (hmmr_conv_desc_t.k_order = 1, tiles 9 / 10) against a float64 convolution and
against the ring tiles on the same split operands.   python tools/patch_check.py"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from human_dynamics_amd import _lib as L  # noqa: E402
from human_dynamics_amd import engine as E  # noqa: E402

CASES = [  # n, h, w, cin, cout, tile
    (5, 14, 14, 256, 256, 10), (9, 7, 7, 512, 512, 10), (3, 28, 28, 128, 128, 9), (4, 5, 9, 64, 256, 10),
    (1, 14, 14, 256, 256, 10), (2, 3, 3, 32, 256, 10), (7, 7, 7, 128, 384, 9), (2, 28, 28, 128, 256, 9),
    (3, 14, 14, 64, 512, 0),
]


def main():
    rng = np.random.default_rng(0)
    bad = 0
    for n, h, w, cin, cout, tile in CASES:
        x = rng.normal(size=(n, h, w, cin)).astype(np.float32)
        wt = (rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
        sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
        sh = rng.normal(size=cout).astype(np.float32)
        ref = torch.nn.functional.conv2d(torch.from_numpy(x).double().permute(0, 3, 1, 2),
                                         torch.from_numpy(wt).double().permute(3, 2, 0, 1), padding=1)
        ref = torch.relu(ref * torch.from_numpy(sc).double()[None, :, None, None] +
                         torch.from_numpy(sh).double()[None, :, None, None]).permute(0, 2, 3, 1).numpy()
        kw = dict(pad=1, scale=sc, shift=sh, relu=True, in_dtype=L.HMMR_F16X3, out_dtype=L.HMMR_F16X3)
        a, _ = E.conv_gemm(x, wt, tile=tile, k_order=1, **kw)
        b, _ = E.conv_gemm(x, wt, tile=0, **kw)
        ea, eb = np.abs(a - ref).max(), np.abs(b - ref).max()
        ok = ea < 3e-6 * max(1.0, np.abs(ref).max())
        bad += not ok
        print("n=%d %dx%d %d->%d tile %d: patch err %.2e, ring err %.2e, patch vs ring %.2e %s"
              % (n, h, w, cin, cout, tile, ea, eb, np.abs(a - b).max(), "ok" if ok else "BAD"), flush=True)
    print("FAILED" if bad else "all ok")
    return bad


if __name__ == "__main__":
    sys.exit(main())
