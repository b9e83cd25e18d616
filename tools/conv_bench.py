"""Micro-benchmark of hmmr_conv_gemm on the ResNet-50 / temporal / IEF layer shapes (dev aid).

    python tools/conv_bench.py [batch] [dtype]
"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from human_dynamics_amd import _lib as L
from human_dynamics_amd import packing

DT = {"bf16": (L.HMMR_BF16, torch.bfloat16), "f32": (L.HMMR_F32, torch.float32), "f16x3": (L.HMMR_F16X3, packing.SPLIT)}


def cast(t, tdt):
    return packing.to_split(t) if tdt is packing.SPLIT else t.to(tdt)


def run(lib, name, n, h, cin, cout, k, stride, kind, dt, tile, iters=20):
    code, tdt = DT[dt]
    dev = "cuda"
    pad = 1 if k == 3 else 0
    ho = (h + 2 * pad - k) // stride + 1
    x = cast(torch.randn((n, h, h, cin), device=dev) * 0.5, tdt)
    w = cast(torch.randn((max(cout, 128) if cout % 128 else cout, k * k * cin), device=dev) / (k * k * cin) ** 0.5, tdt)
    if w.shape[0] % 128:
        w = torch.cat([w, torch.zeros((128 - w.shape[0] % 128, w.shape[1]), device=dev, dtype=tdt)])
    sc = torch.rand(w.shape[0], device=dev) + 0.5
    sh = torch.randn(w.shape[0], device=dev)
    out = packing.empty_act((n, ho, ho, cout), code, dev)
    d = L.ConvDesc()
    d.in_, d.w, d.out = x.data_ptr(), w.data_ptr(), out.data_ptr()
    d.in_dtype = d.out_dtype = code
    d.n_img, d.hin, d.win, d.cin = n, h, h, cin
    d.in_img_stride, d.in_row_stride, d.in_px_stride = h * h * cin, h * cin, cin
    d.kh = d.kw = k; d.sy = d.sx = stride; d.py = d.px = pad
    d.ho = d.wo = ho; d.cout = cout; d.ldo = cout; d.tile = tile
    if tile in (9, 10, 11):
        d.k_order = 1        # the 3x3 patch kernel (timing only: the random filter needs no re-packing)
    keep = [x, w, sc, sh, out]
    nbytes = x.numel() * x.element_size() / (stride * stride if k == 1 else 1) + out.numel() * out.element_size()
    if kind in ("bnrelu", "pro"):
        d.scale, d.shift, d.relu = sc.data_ptr(), sh.data_ptr(), 1
        if kind == "pro":                     # conv1 / shortcut: fused pre-activation of the A operand
            ps = torch.rand(cin, device=dev) + 0.5
            pb = torch.randn(cin, device=dev)
            d.pro_scale, d.pro_shift = ps.data_ptr(), pb.data_ptr()
            keep += [ps, pb]
    elif kind == "bias":
        d.shift = sh.data_ptr()
    elif kind == "res":                       # conv3: bias + residual
        res = cast(torch.randn((n, ho, ho, cout), device=dev), tdt)
        d.shift, d.res, d.ldr = sh.data_ptr(), res.data_ptr(), cout
        keep += [res]
        nbytes += out.numel() * out.element_size()
    elif kind == "res2":                      # conv3: bias + residual + second output
        res = cast(torch.randn((n, ho, ho, cout), device=dev), tdt)
        out2 = torch.empty_like(out)
        d.shift, d.res, d.ldr = sh.data_ptr(), res.data_ptr(), cout
        d.out2, d.scale2, d.shift2 = out2.data_ptr(), sc.data_ptr(), sh.data_ptr()
        keep += [res, out2]
        nbytes += 2 * out.numel() * out.element_size()
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        L.check(lib.hmmr_conv_gemm(C.byref(d), st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.hmmr_conv_gemm(C.byref(d), st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * n * ho * ho * cout * k * k * cin
    return {"layer": name, "tile": tile, "ms": round(ms, 4), "TF": round(flops / ms / 1e9, 1),
            "GBs": round(nbytes / ms / 1e6, 0)}


SHAPES = [  # name, h, cin, cout, k, stride, kind
    ("b1.conv1(256->64)", 56, 256, 64, 1, 1, "pro"),
    ("b1.conv2(3x3 64)", 56, 64, 64, 3, 1, "bnrelu"),
    ("b1.conv3(64->256)", 56, 64, 256, 1, 1, "res"),
    ("b1.short(64->256)", 56, 64, 256, 1, 1, "bias"),
    ("b2.conv1(512->128)", 28, 512, 128, 1, 1, "pro"),
    ("b2.conv2(3x3 128)", 28, 128, 128, 3, 1, "bnrelu"),
    ("b2.conv3(128->512)", 28, 128, 512, 1, 1, "res"),
    ("b3.conv1(1024->256)", 14, 1024, 256, 1, 1, "pro"),
    ("b3.conv2(3x3 256)", 14, 256, 256, 3, 1, "bnrelu"),
    ("b3.conv3(256->1024)", 14, 256, 1024, 1, 1, "res"),
    ("b4.conv1(2048->512)", 7, 2048, 512, 1, 1, "pro"),
    ("b4.conv2(3x3 512)", 7, 512, 512, 3, 1, "bnrelu"),
    ("b4.conv3(512->2048)", 7, 512, 2048, 1, 1, "res"),
]


TILES = (1, 5, 2, 6, 3)


def main():
    global TILES
    if len(sys.argv) > 3:
        TILES = tuple(int(t) for t in sys.argv[3].split(','))
    only = sys.argv[4].split(',') if len(sys.argv) > 4 else None
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dt = sys.argv[2] if len(sys.argv) > 2 else "bf16"
    lib = L.load()
    from human_dynamics_amd import engine as E
    E._debug_from_env()              # HMMR_GEMM_PROBE (probe build only): K loop without MFMAs (1) / loads (2) / barriers (4)
    rows = []
    for name, h, cin, cout, k, stride, kind in SHAPES:
        if only and not any(o in name for o in only):
            continue
        for tile in TILES:
            if (tile in (1, 5, 7, 9, 11) and cout % 128) or (tile in (8, 10) and cout % 256):
                continue
            if tile in (9, 10, 11) and not (k == 3 and stride == 1 and dt in ("f16x3", "split")):
                continue
            rows.append(run(lib, name, n, h, cin, cout, k, stride, kind, dt, tile))
            print(json.dumps(rows[-1]), flush=True)


if __name__ == "__main__":
    main()
