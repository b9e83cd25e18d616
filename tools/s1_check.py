"""Timing of csrc/conv1x1_stream.hip on the layer shapes it serves (development aid; results are not checked here --
tests/test_gpu_conv1x1_stream.py does that).

    [HMMR_LIB_PATH=human_dynamics_amd/libhmmr_hip_s1probe_<bits>.so] python tools/s1_check.py [frames] [tiles, comma separated] [k_order]
"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from human_dynamics_amd import _lib as L
from human_dynamics_amd import packing

SHAPES = [  # name, h, cin, cout, n_split (or: "c3", cin2, res, out2 -- the conv3 form)
    ("4.1 c1", 7, 1024, 512, 0),
    ("4.2 c1", 7, 2048, 512, 0),
    ("3.1 sc+c1", 14, 512, 1280, 1024),
    ("2.1 c1", 28, 256, 128, 0),
    ("4.1 c3", 7, 512, 2048, ("c3", 1024, False, True)),
    ("4.2 c3", 7, 512, 2048, ("c3", 0, True, True)),
    ("4.3 c3", 7, 512, 2048, ("c3", 0, True, False)),
]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 257
    tiles = [int(t) for t in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 22, 23, 24, 25]
    k_order = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    lib = L.load()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    for name, h, cin, cout, n_split in SHAPES:
        c3 = n_split if isinstance(n_split, tuple) else None
        n_split = 0 if c3 else n_split
        cin2 = c3[1] if c3 else 0
        x = packing.to_split(torch.randn((n, h, h, cin), device=dev))
        w = (rng.normal(size=(1, 1, cin + cin2, cout)) / np.sqrt(cin + cin2)).astype(np.float32)
        if k_order == 2:
            wt = packing.pack_conv1x1_stream(w).to(dev)
        else:
            wt = packing.to_split(torch.from_numpy(packing.scale_rows(packing.pack_conv_weight(w), packing.row_pow2(packing.pack_conv_weight(w)))).to(dev))
        sc = torch.full((cout,), 2.0 ** -12, device=dev)
        sh = torch.zeros(cout, device=dev)
        out = packing.empty_act((n, h, h, n_split or cout), L.HMMR_F16X3, dev)
        out_b = packing.empty_act((n, h, h, cout - n_split), L.HMMR_F16X3, dev) if n_split else None
        d = L.ConvDesc()
        d.in_, d.w, d.out = x.data_ptr(), wt.data_ptr(), out.data_ptr()
        d.scale, d.shift, d.relu = sc.data_ptr(), sh.data_ptr(), 0 if n_split else 1
        d.in_dtype = d.out_dtype = L.HMMR_F16X3
        d.n_img, d.hin, d.win, d.cin = n, h, h, cin
        d.in_img_stride, d.in_row_stride, d.in_px_stride = h * h * cin, h * cin, cin
        d.kh = d.kw = 1; d.sy = d.sx = 1
        d.ho = d.wo = h; d.cout = cout; d.ldo = n_split or cout; d.k_order = k_order
        if n_split:
            d.out_b, d.ldo_b, d.n_split, d.relu_b = out_b.data_ptr(), cout - n_split, n_split, 1
        keep = []
        if c3:
            d.relu = 0
            if cin2:
                x2 = packing.to_split(torch.randn((n, h, h, cin2), device=dev))
                d.in2, d.cin2 = x2.data_ptr(), cin2
                keep.append(x2)
            if c3[2]:
                r = packing.to_split(torch.randn((n, h, h, cout), device=dev))
                d.res, d.ldr = r.data_ptr(), cout
                keep.append(r)
            if c3[3]:
                o2 = packing.empty_act((n, h, h, cout), L.HMMR_F16X3, dev)
                s2 = torch.ones(cout, device=dev)
                d.out2, d.scale2, d.shift2 = o2.data_ptr(), s2.data_ptr(), sh.data_ptr()
                keep += [o2, s2]
        flops = 2.0 * n * h * h * (cin + cin2) * cout
        for tile in tiles:
            d.tile = tile
            st = torch.cuda.current_stream(dev).cuda_stream
            try:
                L.check(lib.hmmr_conv_gemm(C.byref(d), st), "hmmr_conv_gemm")
            except L.HmmrError as e:
                print("%-10s tile %2d: %s" % (name, tile, str(e)[:80]))
                continue
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for rep in range(3):
                e0.record()
                for _ in range(10):
                    lib.hmmr_conv_gemm(C.byref(d), st)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 10)
            print("%-10s tile %2d: %.4f ms  %.0f TFLOP/s" % (name, tile, best, flops / best / 1e9), flush=True)


if __name__ == "__main__":
    main()
