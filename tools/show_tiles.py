"""Print the per-layer tiles the autotuner picks for a batch size (development aid)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from human_dynamics_amd import assets
from human_dynamics_amd.engine import HmmrEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dt = sys.argv[2] if len(sys.argv) > 2 else "bf16"
x = torch.rand((n, 224, 224, 3), device="cuda") * 2 - 1
res = {}
for tune in (False, True):
    eng = HmmrEngine(assets.make_synthetic_weights(0), None, dtype=dt, autotune=tune)
    t0 = time.perf_counter()
    eng.resnet(x, n_zero=1)
    torch.cuda.synchronize()
    first = (time.perf_counter() - t0) * 1e3
    for _ in range(3):
        eng.resnet(x, n_zero=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        eng.resnet(x, n_zero=1)
    torch.cuda.synchronize()
    res[tune] = (time.perf_counter() - t0) * 100
    print("autotune=%s first call %.1f ms, steady %.3f ms/pass" % (tune, first, res[tune]))
    if tune:
        tab = eng._tiles[n + 1]
        for _, u, nm in eng._resnet_layers():
            if tab[(u, nm)]:
                print("  unit %2d %-8s -> tile %d" % (u, nm, tab[(u, nm)]))
