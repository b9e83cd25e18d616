"""The one-wave-per-SIMD 3x3 kernel (csrc/conv3x3_stream.hip, k_order 2) alone: every tile against a float64 convolution and
against each other (bit for bit), timed next to the 8-wave patch kernel (k_order 1, tile 11).
    python tools/s3_check.py b2|b3|b4 [frames] [tiles ...]"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from human_dynamics_amd import _lib as L, packing, engine  # noqa: E402

lib = L.load()
ts = None
if "probe" in L.LIB_PATH:
    ts = torch.zeros((4096, 4, 8), dtype=torch.int64, device="cuda")
    d_ = L.Debug()
    d_.reserved[0], d_.reserved[1] = ts.data_ptr() & 0xffffffff, ts.data_ptr() >> 32
    lib.hmmr_set_debug(C.byref(d_))
blk = sys.argv[1] if len(sys.argv) > 1 else "b3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 257
tiles = [int(t) for t in sys.argv[3:]] or [12, 13, 14, 15, 16, 17, 18]
cch, hw = {"b1": (64, 56), "b2": (128, 28), "b3": (256, 14), "b4": (512, 7)}[blk]
if blk == "b1" and len(sys.argv) <= 3:
    tiles = [19, 20]
dev = "cuda"
X3 = L.HMMR_F16X3
g = torch.Generator(device="cpu").manual_seed(5)
x = torch.randn(n, hw, hw, cch, generator=g).clamp_(min=0)
w = (torch.randn(3, 3, cch, cch, generator=g) / (9 * cch) ** 0.5).numpy()
sc = (torch.rand(cch, generator=g) + 0.5).numpy()
sh = (torch.randn(cch, generator=g) * 0.3).numpy()
xs = packing.to_split(x.to(dev))
xv = packing.from_split(xs).double()                                    # the values the kernels see
nref = min(n, 6)
ref = torch.nn.functional.conv2d(xv[:nref].permute(0, 3, 1, 2), torch.from_numpy(w).double().permute(3, 2, 0, 1).to(dev), padding=1)
ref = torch.relu(ref.permute(0, 2, 3, 1) * torch.from_numpy(sc).double().to(dev) + torch.from_numpy(sh).double().to(dev))
rtail = torch.nn.functional.conv2d(xv[-2:].permute(0, 3, 1, 2), torch.from_numpy(w).double().permute(3, 2, 0, 1).to(dev), padding=1)
rtail = torch.relu(rtail.permute(0, 2, 3, 1) * torch.from_numpy(sc).double().to(dev) + torch.from_numpy(sh).double().to(dev))


def run(k_order, tile, reps=10):
    # engine.conv_gemm packs and launches once; re-launch the same descriptor for the timing
    store = packing.DeviceStore(dev)
    wp = packing.pack_conv_weight(w, k_order if k_order != 2 else 0)
    wp = wp[:cch]
    k = packing.row_pow2(wp)
    scale = store.vec((sc.astype(np.float64) * np.exp2(-k.astype(np.float64))).astype(np.float32))
    shift = store.vec(sh)
    wt = store.put_tensor(packing.pack_conv3x3_stream(w, k)) if k_order == 2 else store.put(packing.scale_rows(wp, k), packing.SPLIT)
    out = packing.empty_act((n, hw, hw, cch), X3, dev, zero=True)
    d = L.ConvDesc()
    d.in_, d.w, d.out, d.scale, d.shift = xs.data_ptr(), wt.data_ptr(), out.data_ptr(), scale.data_ptr(), shift.data_ptr()
    d.in_dtype = d.out_dtype = X3
    d.n_img, d.hin, d.win, d.cin = n, hw, hw, cch
    d.in_img_stride, d.in_row_stride, d.in_px_stride = hw * hw * cch, hw * cch, cch
    d.kh = d.kw = 3
    d.sy = d.sx = d.py = d.px = 1
    d.ho = d.wo = hw
    d.cout = d.ldo = cch
    d.relu, d.tile, d.k_order = 1, tile, k_order
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        L.check(lib.hmmr_conv_gemm(C.byref(d), st), "conv")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.check(lib.hmmr_conv_gemm(C.byref(d), st), "conv")
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    if ts is not None and k_order == 2:
        bm = {12: 448, 13: 256, 14: 512, 15: 384, 16: 320, 17: 512, 18: 384, 19: 640, 20: 512, 21: 224}[tile]
        nb = ((n * hw * hw + bm - 1) // bm) * max(1, cch // 128)
        t = ts[:nb].cpu().numpy().astype(np.float64)
        t0 = t[:, :, 0].min()
        print("      stamps (100 MHz ticks) over %d workgroups: start spread %.0f | prologue %.0f | loop %.0f (min %.0f max %.0f) | epilogue %.0f | span %.0f" % (
            nb, t[:, :, 0].max() - t0, (t[:, :, 1] - t[:, :, 0]).mean(), (t[:, :, 2] - t[:, :, 1]).mean(), (t[:, :, 2] - t[:, :, 1]).min(),
            (t[:, :, 2] - t[:, :, 1]).max(), (t[:, :, 3] - t[:, :, 2]).mean(), t[:, :, 3].max() - t0))
    return out, ms


fl = 2.0 * n * hw * hw * 9 * cch * cch
o11, ms11 = run(1, 11) if blk != "b1" else run(0, 0)
print("%s, %d frames (%d px): %s  %.4f ms = %.0f TFLOP/s" % (blk, n, n * hw * hw, "k_order 1 tile 11" if blk != "b1" else "k_order 0", ms11, fl / ms11 / 1e9))
first = None
for t in tiles:
    o, ms = run(2, t)
    f = packing.from_split(o).double()
    err = float((f[:nref] - ref).abs().max())
    errt = float((f[-2:] - rtail).abs().max())
    e11 = float((f - packing.from_split(o11).double()).abs().max())
    same = "" if first is None else ("   == tile %d: %s" % (tiles[0], torch.equal(o, first)))
    if first is None:
        first = o
    print("  tile %2d  %.4f ms = %.0f TFLOP/s   max err vs float64: %.2e (first frames) %.2e (last frames; max |ref| %.2f)   vs tile 11: %.2e%s"
          % (t, ms, fl / ms / 1e9, err, errt, float(ref.abs().max()), e11, same))
flags = C.c_uint(0)
lib.hmmr_run_flags(C.byref(flags), 1)
print("run flags:", flags.value)
