"""Run hmmr_bottleneck_tail alone at ResNet sizes (for rocprofv3 counter passes; development aid)."""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from human_dynamics_amd import _lib as L
lib = L.load()
blk = sys.argv[1] if len(sys.argv) > 1 else "b2"
n = 257
cm, depth, n2, hw = (64, 256, 64, 56) if blk == "b1" else (128, 512, 128, 28)
m = n * hw * hw
bf = torch.bfloat16
dev = "cuda"
h2 = torch.randn((m, cm), device=dev).clamp_(min=0).to(bf)
w3 = (torch.randn((depth, cm), device=dev) / 8).to(bf)
b3 = torch.randn(depth, device=dev)
res = torch.randn((m, depth), device=dev).to(bf)
out = torch.empty((m, depth), device=dev, dtype=bf)
ps, pb = torch.rand(depth, device=dev) + 0.5, torch.randn(depth, device=dev)
w1 = (torch.randn((n2, depth), device=dev) / 16).to(bf)
s1, b1 = torch.rand(n2, device=dev) + 0.5, torch.randn(n2, device=dev)
h1 = torch.empty((m, n2), device=dev, dtype=bf)
d = L.TailDesc()
d.dtype, d.h2, d.m, d.c_mid, d.depth = L.HMMR_BF16, h2.data_ptr(), m, cm, depth
d.w3, d.shift3, d.res, d.ldr, d.out = w3.data_ptr(), b3.data_ptr(), res.data_ptr(), depth, out.data_ptr()
d.pre_scale, d.pre_shift, d.w1, d.scale1, d.shift1, d.relu1, d.n2, d.out_h1 = ps.data_ptr(), pb.data_ptr(), w1.data_ptr(), s1.data_ptr(), b1.data_ptr(), 1, n2, h1.data_ptr()
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    L.check(lib.hmmr_bottleneck_tail(C.byref(d), st))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    lib.hmmr_bottleneck_tail(C.byref(d), st)
e1.record()
torch.cuda.synchronize()
print(blk, "ms per launch %.4f" % (e0.elapsed_time(e1) / 10))
