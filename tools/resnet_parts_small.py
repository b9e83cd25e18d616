"""Dev aid (round 6): does the two-part ResNet (two contiguous halves on concurrent streams, engine.resnet(parts=2)) pay below the 128-frame switch?
At 64 frames blocks 3-4 are 98 - 392 wave tiles for 1 024 SIMDs: two independent launch sequences could fill each other's idle CUs.
    python tools/resnet_parts_small.py [dtype]"""
import sys
import torch
sys.path.insert(0, ".")
from human_dynamics_amd import assets
from human_dynamics_amd.engine import HmmrEngine

dt = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
eng = HmmrEngine(assets.make_synthetic_weights(0), None, dtype=dt)
eng._SPLIT_MIN_FRAMES = 1
print("%6s %12s %12s %12s   (%s; ms per pass)" % ("frames", "one stream", "two parts", "three parts", dt))
for n in [int(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else (20, 40, 64, 96, 128, 160):
    x = torch.rand((n, 224, 224, 3), device="cuda") * 2 - 1
    row = []
    ref = None
    for parts in (1, 2, 3):
        for _ in range(4):
            phi = eng.resnet(x, parts=parts)
        torch.cuda.synchronize()
        if ref is None:
            ref = phi.clone()
        else:
            assert torch.equal(phi, ref), "parts change no bits"
        reps = 30
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            eng.resnet(x, parts=parts)
        e1.record(); torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / reps)
    print("%6d %12.3f %12.3f %12.3f" % (n, *row))
