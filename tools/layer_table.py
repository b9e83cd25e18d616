"""Per-layer ResNet table on one GPU: ms, TFLOP/s and algorithmic TB/s per launch (development aid).

    python tools/layer_table.py [n_frames] [dtype] [reps]
"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from human_dynamics_amd import assets
from human_dynamics_amd.engine import HmmrEngine


def layers(n, esz, fused_stem=True):
    """(name, flops, bytes) per profile slot, in launch order (csrc/resnet.hip)."""
    if fused_stem:
        out = [("stem(fused)", 2.0 * n * 112 * 112 * 64 * 147, n * (224 * 224 * 3 * 4 + 56 * 56 * 64 * esz)),
               ("-", 0, 0), ("-", 0, 0)]
    else:
        out = [("stem re-pack", 0, n * (224 * 224 * 3 * 4 + 230 * 232 * 4 * esz)),
               ("stem 7x7/2 (8 taps x 32)", 2.0 * n * 112 * 112 * 64 * 147, n * (230 * 232 * 4 + 112 * 112 * 64) * esz),
               ("pool1 + preact", 0, n * (112 * 112 * 64 + 56 * 56 * 64) * esz)]
    h = 56
    for scope, c_in, base, depth, stride, has_sc in assets.resnet_units():
        u = scope.split("/")[1][5:] + "." + scope.split("/")[2][5:]
        px, ho = n * h * h, h // stride
        pxo = n * ho * ho
        if has_sc:
            out.append((u + " sc  1x1 %d->%d" % (c_in, depth), 2.0 * pxo * c_in * depth,
                        (pxo * c_in + pxo * depth) * esz + c_in * depth * esz))
        out.append((u + " c1  1x1 %d->%d" % (c_in, base), 2.0 * px * c_in * base, (px * c_in + px * base) * esz))
        out.append((u + " c2  3x3 %d/s%d" % (base, stride), 2.0 * pxo * 9 * base * base,
                    (px * base + pxo * base) * esz + 9 * base * base * esz))
        out.append((u + " c3  1x1 %d->%d" % (base, depth), 2.0 * pxo * base * depth,
                    (pxo * base + 2 * pxo * depth) * esz))
        h = ho
    out.append(("pool5", 0, n * 49 * 2048 * esz))
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 257
    dt = sys.argv[2] if len(sys.argv) > 2 else "bf16"
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    eng = HmmrEngine(assets.make_synthetic_weights(0), assets.make_synthetic_smpl(2), dtype=dt)
    x = torch.rand((n, 224, 224, 3), device="cuda") * 2 - 1
    for _ in range(2):
        eng.resnet(x, prof=True)
    acc = None
    for _ in range(reps):
        _, prof = eng.resnet(x, prof=True)
        p = np.asarray(prof, dtype=np.float64)
        acc = p if acc is None else np.minimum(acc, p)
    tab = layers(n, 2 if dt == "bf16" else 4, fused_stem=(dt != "f32"))
    tot_ms = tot_f = tot_b = 0.0
    print("%-26s %8s %8s %8s %7s" % ("layer", "ms", "TFLOP/s", "TB/s", "GB"))
    for i, (name, fl, by) in enumerate(tab):
        ms = float(acc[i])
        if fl == 0 and by == 0:
            continue
        print("%-26s %8.4f %8.1f %8.2f %7.3f" % (name, ms, fl / ms / 1e9 if ms > 0 else 0, by / ms / 1e9 if ms > 0 else 0, by / 1e9))
        tot_ms += ms; tot_f += fl; tot_b += by
    print("%-26s %8.4f %8.1f %8.2f %7.3f" % ("TOTAL", tot_ms, tot_f / tot_ms / 1e9, tot_b / tot_ms / 1e9, tot_b / 1e9))


if __name__ == "__main__":
    main()
