"""Dev aid (round 6): the shader clock over ResNet passes -- bench.ClockSampler's samples as a time series (one line per ~100 us) and the same for
a pure MFMA loop and for an idle GPU.    python tools/clock_trace.py [frames] [dtype]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from human_dynamics_amd import assets, _lib as L
from human_dynamics_amd.engine import HmmrEngine
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 257
dt = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
eng = HmmrEngine(assets.make_synthetic_weights(0), None, dtype=dt)
eng.resnet_streams = 1
x = torch.rand((n - 1, 224, 224, 3), device="cuda") * 2 - 1
for _ in range(3):
    eng.resnet(x, n_zero=1)
torch.cuda.synchronize()


def series(label, work, every=20):
    s = bench.ClockSampler(eng)
    s.start()
    work()
    res = s.stop()
    b = s.buf.cpu().numpy()
    b = b[b[:, 1] > 0][1:]
    t = (b[:, 1] - b[0, 1]) / 100.0                       # us
    w = b[::every]
    mhz = (w[1:, 0] - w[:-1, 0]) / np.maximum(w[1:, 1] - w[:-1, 1], 1) * 100.0
    print("%s: %s" % (label, res))
    print("   t_us:mhz  " + " ".join("%d:%d" % (t[::every][i + 1], mhz[i]) for i in range(len(mhz))))


series("%d-frame ResNet passes x 3 (%s, one stream)" % (n, dt), lambda: [eng.resnet(x, n_zero=1) for _ in range(3)])
cus = torch.cuda.get_device_properties(0).multi_processor_count
series("pure MFMA loop (constant operands) x 20", lambda: [L.check(eng.lib.hmmr_mfma_rate_probe(cus, 2500, None, torch.cuda.current_stream().cuda_stream), "p") for _ in range(20)], every=10)
series("idle (a 10 ms sleep on the host)", lambda: __import__("time").sleep(0.01), every=40)
