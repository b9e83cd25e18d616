"""Dev aid (round 6): the ResNet pass (one stream, shipped tile tables / library choices) over the batch sizes the reference and BASELINE.json
produce: ms, frames/s and the fraction of the split mode's nominal MFMA ceiling.  python tools/resnet_by_batch.py [dtype]"""
import sys
import torch
sys.path.insert(0, ".")
from human_dynamics_amd import assets
from human_dynamics_amd.engine import HmmrEngine

dt = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
peak = {"f16x3": 2.5e15 / 3, "bf16": 2.5e15, "f32": 157.3e12}[dt]
eng = HmmrEngine(assets.make_synthetic_weights(0), None, dtype=dt)
eng.resnet_streams = 1
print("%6s %9s %10s %8s   (%s, one stream; 6.9604 GFLOP per frame)" % ("frames", "ms", "frames/s", "frac", dt))
for n in (8, 20, 40, 64, 96, 128, 160, 192, 256, 257, 384, 512, 768, 1024):
    x = torch.rand((n, 224, 224, 3), device="cuda") * 2 - 1
    for _ in range(3):
        eng.resnet(x)
    reps = max(4, min(40, int(2000 / n)))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        eng.resnet(x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("%6d %9.3f %10.0f %8.4f" % (n, ms, n / ms * 1e3, 6.9604e9 * n / (ms * 1e-3) / peak))
    del x
print("tuning passes run by this engine:", eng.tune_log)
