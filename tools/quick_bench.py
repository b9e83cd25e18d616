"""Per-stage timing on one GPU (development aid; bench.py is the contract)."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from human_dynamics_amd import assets
from human_dynamics_amd.engine import HmmrEngine
from human_dynamics_amd import _lib as L


def timed(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters * 1e3


def layer_names():
    names = ["stem_repack", "stem_conv", "pool1"]
    for scope, c_in, base, depth, stride, has_sc in assets.resnet_units():
        u = scope.split("/")[1] + "/" + scope.split("/")[2]
        if has_sc:
            names.append(u + "/shortcut")
        names += [u + "/conv1", u + "/conv2", u + "/conv3"]
    names.append("pool5")
    return names


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    w = assets.make_synthetic_weights(0)
    s = assets.make_synthetic_smpl(2)
    out = {}
    for dt in ("bf16", "f32"):
        eng = HmmrEngine(w, s, dtype=dt)
        x = torch.rand((n, 224, 224, 3), device="cuda") * 2 - 1
        ms = timed(lambda: eng.resnet(x))
        out["resnet_%s_ms" % dt] = ms
        out["resnet_%s_fps" % dt] = n / ms * 1e3
        out["resnet_%s_tflops" % dt] = n * 6.9604e9 / (ms * 1e-3) / 1e12
        _, prof = eng.resnet(x, prof=True)
        _, prof = eng.resnet(x, prof=True)
        names = layer_names()
        out["resnet_%s_layers" % dt] = {names[i]: round(float(prof[i]), 4) for i in range(len(names))}
        phi = torch.randn((n // 2, 20, 2048), device="cuda")
        out["temporal_%s_ms(b=%d,t=20)" % (dt, n // 2)] = timed(lambda: eng.temporal(phi))
        st = torch.randn((n * 4, 2048), device="cuda")
        out["ief_%s_ms(m=%d)" % (dt, n * 4)] = timed(lambda: eng.ief(st))
        if dt == "bf16":
            th = torch.randn((n * 12, 72), device="cuda") * 0.3
            be = torch.randn((n * 12, 10), device="cuda")
            cm = torch.rand((n * 12, 3), device="cuda")
            out["smpl_ms(m=%d)" % (n * 12)] = timed(lambda: eng.smpl(th, be, cm))
        del eng
        torch.cuda.empty_cache()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
