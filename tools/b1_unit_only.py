"""Run the whole-unit kernel of block 1 (csrc/b1_unit.hip) alone at ResNet sizes, both forms (timing only: random filters; dev aid).
    [HMMR_LIB_PATH=human_dynamics_amd/libhmmr_hip_b1p<name>.so] python tools/b1_unit_only.py [n_frames]"""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from human_dynamics_amd import _lib as L, packing
lib = L.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 257
m = n * 56 * 56
dev = "cuda"
X3 = L.HMMR_F16X3
rng = np.random.default_rng(0)
h1 = packing.to_split(torch.randn((m, 64), device=dev).clamp_(min=0))
xp = packing.to_split(torch.randn((m, 64), device=dev).clamp_(min=0))
res = packing.to_split(torch.randn((m, 256), device=dev))
out = packing.empty_act((m, 256), X3, dev)
h1n = packing.empty_act((m, 64), X3, dev)
w2 = (rng.normal(size=(3, 3, 64, 64)) * 0.06).astype(np.float32)
k2 = packing.row_pow2(packing.pack_conv_weight(w2)[:64])
vec = lambda v: torch.tensor(np.asarray(v, np.float32), device=dev)
s2, b2 = vec(np.exp2(-k2.astype(np.float64))), vec(rng.normal(size=64) * 0.1)
ps, pb = vec(rng.uniform(0.5, 1.5, 256)), vec(rng.normal(size=256) * 0.1)
b3 = vec(rng.normal(size=256) * 0.1)
b1 = vec(rng.normal(size=64) * 0.1)
ts = None
if "b1p64" in L.LIB_PATH or "b1p66" in L.LIB_PATH:              # stamp build: [tiles][4 waves][16] s_memtime values (100 MHz)
    ts = torch.zeros(((m + 127) // 128, 4, 16), dtype=torch.int64, device=dev)
    d_ = L.Debug()
    d_.reserved[0], d_.reserved[1] = ts.data_ptr() & 0xffffffff, ts.data_ptr() >> 32
    lib.hmmr_set_debug(C.byref(d_))
st = torch.cuda.current_stream().cuda_stream
for folded in (True, False):
    K3 = 128 if folded else 64
    w3 = (rng.normal(size=(256, K3)) / K3 ** 0.5).astype(np.float32)
    w1 = (rng.normal(size=(64, 256)) / 16).astype(np.float32)
    stream = packing.pack_b1_unit_stream(w2, k2, w3, w1).to(dev)
    s3 = vec(np.exp2(-packing.row_pow2(w3).astype(np.float64)))
    s1 = vec(np.exp2(-packing.row_pow2(w1).astype(np.float64)))
    d = L.TailDesc()
    d.dtype, d.m, d.c_mid, d.depth, d.n2 = X3, m, 64, 256, 64
    d.h1, d.hin, d.win, d.ho, d.wo = h1.data_ptr(), 56, 56, 56, 56
    d.unit_stream, d.scale2, d.shift2 = stream.data_ptr(), s2.data_ptr(), b2.data_ptr()
    d.scale3, d.shift3, d.pre_scale, d.pre_shift = s3.data_ptr(), b3.data_ptr(), ps.data_ptr(), pb.data_ptr()
    d.scale1, d.shift1, d.relu1, d.out, d.out_h1 = s1.data_ptr(), b1.data_ptr(), 1, out.data_ptr(), h1n.data_ptr()
    if folded:
        d.xp, d.c_xp = xp.data_ptr(), 64
    else:
        d.res, d.ldr = res.data_ptr(), 256
    for _ in range(3):
        L.check(lib.hmmr_bottleneck_tail(C.byref(d), st), "b1 unit")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        lib.hmmr_bottleneck_tail(C.byref(d), st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    gb = (h1.numel() + (xp.numel() if folded else res.numel()) + out.numel() + h1n.numel()) * 4 / 1e9
    fl = 2.0 * m * (9 * 64 * 64 + K3 * 256 + 256 * 64)
    if ts is not None:
        t = ts.cpu().numpy().astype(np.float64)                  # s_memtime ticks of the LAST launch
        t = t[:, :, :13]
        d0 = (t - t[:, :, :1]) * 10.0                            # ns since the workgroup's start (100 MHz counter)
        names = ["prologue", "conv2", "c2 epi"] + ["chunk %d" % c for c in range(8)] + ["h1' epi"]
        med = np.median(np.diff(d0, axis=2).reshape(-1, 12), axis=0)
        print("   per phase, median ns per wave: " + ", ".join("%s %.0f" % (nm, v) for nm, v in zip(names, med)))
        print("   workgroup lifetime: median %.1f us, launch span %.1f us" % (np.median(d0[:, :, 12]) / 1e3, (t[:, :, 12].max() - t[:, :, 0].min()) * 10.0 / 1e3))
    print("b1 unit %s: %.4f ms per launch, %.2f GB -> %.2f TB/s, %.0f TFLOP/s" % ("folded  " if folded else "identity", ms, gb, gb / ms, fl / ms / 1e9))
