"""Operand-mode selection: `dtype="auto"`, the default of `Tester`.

The reference graph is fp32 (src/evaluation/tester.py:64-66); the north-star tolerance is 1e-4 on vertices / joints.
bf16x3 (split-bf16 operands, ~16-17 mantissa bits) meets it with an order of magnitude to spare on well-conditioned
weights, but its margin is a property of the WEIGHTS: BatchNorm channels with a large offset over their spread, a wide
gamma range or a large fc3 gain amplify operand rounding (oracle/hard_weights.py builds such a set; on it the ResNet in
bf16x3 alone moves the vertices by 1.2e-4).  So the default does not assume, it measures -- on the device, once per
`Tester`, against the exact-fp32 MFMA mode of the same kernels:

    ladder (cheapest first)      resnet    f_movie   IEF
      bf16x3                     bf16x3    bf16x3    bf16x3
      f32 resnet                 f32       bf16x3    bf16x3
      f32                        f32       f32       f32

A rung is accepted when, on `PROBE_WINDOWS` synthetic 20-frame windows, the vertices and joints of all three containers
(present, past, future) stay within `PROBE_FRACTION` x tolerance of the f32 rung's.  The probe frames are synthetic (the
caller's video is not needed): conditioning is a property of the weights, and the fraction leaves room for the
frame-to-frame spread (x 1.5-2 between the probe and the worst frame of a 256-frame video, tests/test_gpu_stress.py)
and for the f32 mode's own distance from the exact graph (1e-6 ... 1e-5).  The last rung is always accepted: it IS the
reference arithmetic (exact fp32 products, fp32 accumulation).

An explicit dtype ('bf16x3', 'bf16', 'f32') skips all of this.
"""
from __future__ import annotations

import numpy as np
import torch

from . import assets
from .engine import DTYPE_NAMES, HmmrEngine

TOLERANCE = 1e-4
PROBE_FRACTION = 0.3
PROBE_WINDOWS = 2
LADDER = (("bf16x3", "bf16x3", "bf16x3"), ("f32", "bf16x3", "bf16x3"), ("f32", "f32", "f32"))


def describe(engine):
    """'bf16x3', or e.g. 'f32 resnet + bf16x3 f_movie / IEF' for a mixed engine."""
    r, t, i = (DTYPE_NAMES[d] for d in (engine.dtype, engine.temporal_dtype, engine.ief_dtype))
    if r == t == i:
        return r
    return "%s resnet + %s f_movie + %s IEF" % (r, t, i)


def probe_outputs(engine, frames, T=20, pred_mode="pred"):
    """frames [W*T,224,224,3] (device) -> {'verts': [R, W*T, V, 3], 'joints': [R, W*T, K, 3]} through every stage."""
    phi = engine.resnet(frames).reshape(-1, T, 2048)
    strips = (engine.temporal(phi) if pred_mode == "pred" else engine.hallucinate(phi)).reshape(-1, 2048)
    om = engine.ief(strips)
    verts, joints = [], []
    for r in range(om.shape[0]):
        v, j, _, _ = engine.smpl(om[r][:, 3:75], om[r][:, 75:85], om[0][:, :3], want_rs=False)
        verts.append(v)
        joints.append(j)
    return {"verts": torch.stack(verts), "joints": torch.stack(joints)}


def choose_engine(weights, smpl, device, pred_mode="pred", **engine_kw):
    """Walk LADDER; returns (engine, report).  report = {'operands', 'probe_tolerance', 'rungs': [{'operands', 'verts',
    'joints', 'accepted'}]} -- what bench.py prints and Tester.precision holds."""
    dev = torch.device(device)
    frames = torch.from_numpy(assets.make_synthetic_frames(20 * PROBE_WINDOWS, seed=4242)).to(dev)
    tol = PROBE_FRACTION * TOLERANCE

    def make(rung):
        return HmmrEngine(weights, smpl, dtype=rung[0], temporal_dtype=rung[1], ief_dtype=rung[2], device=device, **engine_kw)
    ref_engine = make(LADDER[-1])
    ref = probe_outputs(ref_engine, frames, pred_mode=pred_mode)
    rungs, chosen = [], None
    for rung in LADDER[:-1]:
        eng = make(rung)
        got = probe_outputs(eng, frames, pred_mode=pred_mode)
        errs = {k: float((got[k] - ref[k]).abs().max()) for k in ("verts", "joints")}
        ok = bool(np.isfinite(list(errs.values())).all() and max(errs.values()) <= tol)
        rungs.append(dict(operands=describe(eng), accepted=ok, **errs))
        if ok:
            chosen = eng
            break
        del eng, got
        torch.cuda.empty_cache()
    if chosen is None:
        chosen = ref_engine
        rungs.append(dict(operands=describe(ref_engine), accepted=True, verts=0.0, joints=0.0))
    else:
        del ref_engine
    del ref
    torch.cuda.empty_cache()
    return chosen, {"operands": describe(chosen), "probe_tolerance": tol, "probe_frames": 20 * PROBE_WINDOWS,
                    "against": "the f32 rung (exact fp32 MFMA) on the same device", "rungs": rungs}
