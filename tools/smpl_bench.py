"""SMPL stage timing (dev aid): the vector-unit blend (default) against the matrix-core form, and the one-launch-set record path.
    python tools/smpl_bench.py"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from human_dynamics_amd import assets
from human_dynamics_amd.engine import HmmrEngine, set_debug

eng = HmmrEngine(None, assets.make_synthetic_smpl(2), device="cuda:0")
rng = np.random.default_rng(0)
for m in (256, 768):
    theta = torch.from_numpy((rng.normal(size=(m, 72)) * 0.5).astype(np.float32)).cuda()
    beta = torch.from_numpy(rng.normal(size=(m, 10)).astype(np.float32)).cuda()
    cams = torch.from_numpy(rng.normal(size=(m, 3)).astype(np.float32)).cuda()
    for name, mf in (("split-fp16 mfma (default)", 0), ("exact-fp32 mfma", 1), ("packed-fma valu", 2)):
        set_debug(smpl_blend_mfma=mf)
        for _ in range(3):
            eng.smpl(theta, beta, cams)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            eng.smpl(theta, beta, cams)
        e1.record()
        torch.cuda.synchronize()
        print("m %4d %s: %.1f us per SMPL call (pose + verts + joints)" % (m, name, e0.elapsed_time(e1) / 20 * 1e3))
set_debug()
