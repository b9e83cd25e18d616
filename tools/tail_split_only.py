"""Run the split-operand hmmr_bottleneck_tail (conv3 + add + next preact + conv1) alone at ResNet sizes (timing only: random
fragment-major filters; dev aid).   python tools/tail_split_only.py b2|b3"""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from human_dynamics_amd import _lib as L, packing
lib = L.load()
blk = sys.argv[1] if len(sys.argv) > 1 else "b2"
n = 257
cm, depth, n2, hw = {"b1": (64, 256, 64, 56), "b2": (128, 512, 128, 28), "b3": (256, 1024, 256, 14)}[blk]
m = n * hw * hw
dev = "cuda"
X3 = L.HMMR_F16X3
h2 = packing.to_split(torch.randn((m, cm), device=dev).clamp_(min=0))
res = packing.to_split(torch.randn((m, depth), device=dev))
out = packing.empty_act((m, depth), X3, dev)
h1 = packing.empty_act((m, n2), X3, dev)
w3 = packing.pack_frag_major((torch.randn((depth, cm)) / cm ** 0.5).numpy()).to(dev)
w1 = packing.pack_frag_major((torch.randn((n2, depth)) / depth ** 0.5).numpy()).to(dev)
b3 = torch.randn(depth, device=dev)
s3 = torch.full((depth,), 2.0 ** -12, device=dev)
ps, pb = torch.rand(depth, device=dev) + 0.5, torch.randn(depth, device=dev)
s1, b1 = (torch.rand(n2, device=dev) + 0.5) * 2.0 ** -12, torch.randn(n2, device=dev)
d = L.TailDesc()
d.dtype, d.h2, d.m, d.c_mid, d.depth = X3, h2.data_ptr(), m, cm, depth
d.w3, d.scale3, d.shift3, d.res, d.ldr, d.out = w3.data_ptr(), s3.data_ptr(), b3.data_ptr(), res.data_ptr(), depth, out.data_ptr()
d.pre_scale, d.pre_shift, d.w1, d.scale1, d.shift1, d.relu1, d.n2, d.out_h1 = ps.data_ptr(), pb.data_ptr(), w1.data_ptr(), s1.data_ptr(), b1.data_ptr(), 1, n2, h1.data_ptr()
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    L.check(lib.hmmr_bottleneck_tail(C.byref(d), st), "tail")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    lib.hmmr_bottleneck_tail(C.byref(d), st)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
gb = (h2.numel() + res.numel() + out.numel() + h1.numel()) * 4 / 1e9
print("%s tail: %.4f ms per launch, %.2f GB -> %.2f TB/s, %.0f TFLOP/s (conv3 + conv1')" % (blk, ms, gb, gb / ms, 2.0 * m * (cm * depth + depth * n2) / ms / 1e9))
