"""The register-resident unit pair (csrc/unit_pair.hip) alone: bit-compare with the two hmmr_conv_gemm launches it replaces
(conv3 + shortcut -> trunk; fused-preact conv1 -> h1') and time both.   python tools/pair_check.py b3|b2|b2f [frames]"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from human_dynamics_amd import _lib as L, packing  # noqa: E402

lib = L.load()
from human_dynamics_amd import devflags, engine  # noqa: E402
blk = sys.argv[1] if len(sys.argv) > 1 else "b3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 257
form = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # hmmr_debug_t.pair_form: 0 = one wave per SIMD (round 4, the default), 2 = wave-specialised (round 6)
ws = form == 2 and blk != "b2f"
ts = None
if "probe" in L.LIB_PATH:
    ts = torch.zeros((4097, 8 if ws else 4, 8), dtype=torch.int64, device="cuda")
    d_ = L.Debug()
    d_.gemm_probe = int(devflags.get("GEMM_PROBE") or 0)
    d_.reserved[0], d_.reserved[1] = ts.data_ptr() & 0xffffffff, ts.data_ptr() >> 32
    d_.pair_form = form
    lib.hmmr_set_debug(C.byref(d_))
else:
    engine.set_debug(pair_form=form)
print("pair_form %d%s" % (form, "  (%s)" % L.LIB_PATH if ts is not None else ""))
cm, depth, n2, hw, cxp = {"b2": (128, 512, 128, 28, 0), "b3": (256, 1024, 256, 14, 0), "b2f": (128, 512, 128, 28, 256)}[blk]
m = n * hw * hw
dev = "cuda"
X3 = L.HMMR_F16X3
g = torch.Generator(device="cpu").manual_seed(3)
rnd = lambda *s: torch.randn(*s, generator=g)
h2 = packing.to_split(rnd(m, cm).clamp_(min=0).to(dev))
xp = packing.to_split(rnd(m, cxp).clamp_(min=0).to(dev)) if cxp else None
res = None if cxp else packing.to_split(rnd(m, depth).to(dev))
K3 = cm + cxp
W3 = (rnd(depth, K3) / K3 ** 0.5).numpy()
W1 = (rnd(n2, depth) / depth ** 0.5).numpy()
b3 = rnd(depth).to(dev)
ps, pb = (torch.rand(depth, generator=g) + 0.5).to(dev), (rnd(depth) * 0.3).to(dev)
s1, b1 = (torch.rand(n2, generator=g) + 0.5), (rnd(n2) * 0.3).to(dev)
store = packing.DeviceStore(dev)
k3, k1 = packing.row_pow2(W3), packing.row_pow2(W1)
w3p = store.put(packing.scale_rows(W3, k3), packing.SPLIT)           # [depth][K3] rows, K-contiguous (what hmmr_conv_gemm reads)
w1p = store.put(packing.scale_rows(W1, k1), packing.SPLIT)
sc3 = torch.from_numpy(np.exp2(-k3.astype(np.float64)).astype(np.float32)).to(dev)
sc1 = torch.from_numpy((s1.double().numpy() * np.exp2(-k1.astype(np.float64))).astype(np.float32)).to(dev)
stream = packing.pack_pair_stream(W3, W1).to(dev)
assert stream.numel() * 2 == lib.hmmr_pair_stream_bytes(K3 // 16, depth, n2)
st = torch.cuda.current_stream().cuda_stream


def reference():
    trunk = packing.empty_act((m, depth), X3, dev, zero=True)
    h1 = packing.empty_act((m, n2), X3, dev, zero=True)
    d = L.ConvDesc()
    d.in_, d.w, d.scale, d.shift, d.out = h2.data_ptr(), w3p.data_ptr(), sc3.data_ptr(), b3.data_ptr(), trunk.data_ptr()
    d.in_dtype = d.out_dtype = X3
    d.n_img, d.hin, d.win, d.cin = 1, 1, m, cm
    d.in_img_stride, d.in_row_stride, d.in_px_stride = m * cm, m * cm, cm
    d.kh = d.kw = d.sy = d.sx = 1
    d.ho, d.wo, d.cout, d.ldo = 1, m, depth, depth
    if cxp:
        d.in2, d.cin2 = xp.data_ptr(), cxp
    else:
        d.res, d.ldr = res.data_ptr(), depth
    d.tile = 5
    e = L.ConvDesc()
    e.in_, e.w, e.scale, e.shift, e.out = trunk.data_ptr(), w1p.data_ptr(), sc1.data_ptr(), b1.data_ptr(), h1.data_ptr()
    e.in_dtype = e.out_dtype = X3
    e.n_img, e.hin, e.win, e.cin = 1, 1, m, depth
    e.in_img_stride, e.in_row_stride, e.in_px_stride = m * depth, m * depth, depth
    e.kh = e.kw = e.sy = e.sx = 1
    e.ho, e.wo, e.cout, e.ldo = 1, m, n2, n2
    e.relu, e.pro_scale, e.pro_shift, e.tile = 1, ps.data_ptr(), pb.data_ptr(), 5

    def run():
        L.check(lib.hmmr_conv_gemm(C.byref(d), st), "conv3")
        L.check(lib.hmmr_conv_gemm(C.byref(e), st), "conv1'")
    return trunk, h1, run


def pair():
    trunk = packing.empty_act((m, depth), X3, dev, zero=True)
    h1 = packing.empty_act((m, n2), X3, dev, zero=True)
    t = L.TailDesc()
    t.dtype, t.h2, t.m, t.c_mid, t.depth = X3, h2.data_ptr(), m, cm, depth
    t.scale3, t.shift3, t.out = sc3.data_ptr(), b3.data_ptr(), trunk.data_ptr()
    if cxp:
        t.xp, t.c_xp = xp.data_ptr(), cxp
    else:
        t.res, t.ldr = res.data_ptr(), depth
    t.pre_scale, t.pre_shift = ps.data_ptr(), pb.data_ptr()
    t.scale1, t.shift1, t.relu1, t.n2, t.out_h1 = sc1.data_ptr(), b1.data_ptr(), 1, n2, h1.data_ptr()
    t.pair_stream = stream.data_ptr()

    def run():
        L.check(lib.hmmr_bottleneck_tail(C.byref(t), st), "unit pair")
    return trunk, h1, run


def timed(run, reps=10):
    for _ in range(2):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


tr, hr, run_ref = reference()
tp, hp, run_pair = pair()
run_ref()
run_pair()
torch.cuda.synchronize()
for name, a, b, width in (("trunk", tr, tp, depth), ("h1'", hr, hp, n2)):
    same = torch.equal(a, b)
    fa, fb = packing.from_split(a), packing.from_split(b)
    diff = (fa - fb).abs()
    print("%s: %s   max |diff| %.3e (max |ref| %.3e)" % (name, "bit-identical" if same else "DIFFERS", float(diff.max()), float(fa.abs().max())))
    if not same:
        bad = (a != b).nonzero()
        rows = torch.unique(bad[:, 0])
        cols = torch.unique(bad[:, 1])
        print("   %d words differ in %d rows (first %s) and %d columns (first %s); non-finite in pair: %d" % (
            bad.shape[0], rows.numel(), rows[:8].tolist(), cols.numel(), cols[:16].tolist(), int((~torch.isfinite(fb)).sum())))
        r0 = int(rows[0])
        print("   row %d ref  %s" % (r0, fa[r0, :8].tolist()))
        print("   row %d pair %s" % (r0, fb[r0, :8].tolist()))
ms_ref, ms_pair = timed(run_ref), timed(run_pair)
gb = (h2.numel() + (xp.numel() if cxp else res.numel()) + tr.numel() + hr.numel()) * 4 / 1e9
fl = 2.0 * m * (K3 * depth + depth * n2)
print("%s (%d px): two launches %.4f ms | unit pair %.4f ms = %.0f TFLOP/s, %.2f TB/s of tensor traffic" % (
    blk, m, ms_ref, ms_pair, fl / ms_pair / 1e9, gb / ms_pair))
import os
if os.environ.get("PAIR_POWER"):              # ~1 s of back-to-back launches under bench.SmiSampler: socket power and the XCDs' clocks while ONLY this kernel runs
    import time
    import bench
    for label, fn, ms_ in (("unit pair", run_pair, ms_pair), ("two launches", run_ref, ms_ref)):
        reps = int(1.0 / (ms_ * 1e-3))
        smi = bench.SmiSampler(0)
        smi.start()
        t0 = time.perf_counter()
        for i in range(reps):
            fn()
            if i % 64 == 63:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        pw = smi.stop() or {}
        print("power, %s x %d back to back: %.4f ms each | %s W mean (max %s, cap %s) | clocks %s MHz mean, slowest XCD %s | %.3f mJ per launch" % (
            label, reps, el / reps * 1e3, pw.get("socket_w_mean"), pw.get("socket_w_max"), pw.get("power_cap_w"), pw.get("gfxclk_mhz_mean"), pw.get("gfxclk_mhz_min_xcd"),
            (pw.get("joules") or 0) / reps * 1e3))

if ts is not None and ws:
    run_pair()
    torch.cuda.synchronize()
    nb = (m + 127) // 128
    t = ts[:nb].cpu().numpy().astype(np.float64)
    t0 = t[:, :, 0].min()
    for b in (0, min(255, nb - 1), nb - 1):
        for w in (0, 4):
            print("block %d wave %d (%s): start %+8.0f | prologue issue %6.0f | drain + sync %6.0f | loop %8.0f | tail %6.0f | in waits + barriers %8.0f | B block %7.0f   (shader-clock ticks)" % (
                b, w, "A" if w < 4 else "B", t[b, w, 0] - t0, t[b, w, 1] - t[b, w, 0], t[b, w, 2] - t[b, w, 1], t[b, w, 3] - t[b, w, 2], t[b, w, 4] - t[b, w, 3], t[b, w, 5], t[b, w, 6]))
    A, Bw = t[:, :4], t[:, 4:]
    print("kernel span (ticks): %.0f | mean loop A %.0f B %.0f | mean prologue %.0f | mean tail A %.0f B %.0f | mean in waits + barriers A %.0f B %.0f | B block %.0f" % (
        t[:, :, 4].max() - t0, (A[:, :, 3] - A[:, :, 2]).mean(), (Bw[:, :, 3] - Bw[:, :, 2]).mean(), (t[:, :, 2] - t[:, :, 0]).mean(),
        (A[:, :, 4] - A[:, :, 3]).mean(), (Bw[:, :, 4] - Bw[:, :, 3]).mean(), A[:, :, 5].mean(), Bw[:, :, 5].mean(), Bw[:, :, 6].mean()))
    # s_memtime counts shader clocks, s_memrealtime (slots 6 / 7) the constant 100 MHz reference: the clock each wave ran at, first-round and second-round workgroups
    dt_, dr_ = Bw[:, :, 4] - Bw[:, :, 0], Bw[:, :, 7] - Bw[:, :, 6]
    ok = dr_ > 0
    if ok.any():
        first = np.arange(nb)[:, None].repeat(4, 1) < 256
        f_ = lambda sel: (dt_[sel & ok].sum() / dr_[sel & ok].sum() * 100.0) if (sel & ok).any() else float("nan")
        print("clock of the B waves (shader ticks per 100 MHz tick): all %.0f MHz | workgroups 0-255 %.0f MHz (%.1f us each) | the rest %.0f MHz (%.1f us each)" % (
            f_(ok), f_(first), dr_[first & ok].mean() / 100.0, f_(~first), dr_[~first & ok].mean() / 100.0 if (~first & ok).any() else float("nan")))
elif ts is not None:
    run_pair()
    torch.cuda.synchronize()
    nb = (m + 127) // 128
    t = ts[:nb].cpu().numpy().astype(np.float64)
    t0 = t[:, :, 0].min()
    for b in (0, min(255, nb - 1), nb - 1):
        for w in (0, 3):
            print("block %d wave %d: start %+8.0f | consts+h2 issue %6.0f | drain %6.0f | loop %8.0f | last wait %5.0f | h1' out %6.0f   (100 MHz ticks x 21 ~ cycles)" % (
                b, w, t[b, w, 0] - t0, t[b, w, 1] - t[b, w, 0], t[b, w, 2] - t[b, w, 1], t[b, w, 3] - t[b, w, 2], t[b, w, 4] - t[b, w, 3], t[b, w, 5] - t[b, w, 4]))
    print("kernel span (ticks): %.0f; mean loop %.0f; mean prologue %.0f; mean tail %.0f" % (
        t[:, :, 5].max() - t0, (t[:, :, 3] - t[:, :, 2]).mean(), (t[:, :, 2] - t[:, :, 0]).mean(), (t[:, :, 5] - t[:, :, 3]).mean()))
    ut = ts[4096].reshape(-1).cpu().numpy()
    if ut.any():
        print("per-unit cycles summed over the iterations (block 0, wave 0): units %s | head %d tail %d step-boundary %d" % (ut[:16].tolist(), ut[16], ut[17], ut[18]))
