"""Per-stage timing of one bench step (dev aid): resnet / temporal / ief / smpl, frames=256."""
import sys, time, json
import torch
sys.path.insert(0, ".")
from human_dynamics_amd import assets
from human_dynamics_amd.engine import HmmrEngine

def timed(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t) / iters * 1e3, 4)

w = assets.make_synthetic_weights(0); s = assets.make_synthetic_smpl(2)
eng = HmmrEngine(w, s, dtype=sys.argv[1] if len(sys.argv) > 1 else "bf16")
x = torch.rand((256, 224, 224, 3), device="cuda") * 2 - 1
phi = torch.randn((32, 20, 2048), device="cuda")
st = torch.randn((256, 2048), device="cuda")
th = torch.randn((256, 72), device="cuda") * 0.3; be = torch.randn((256, 10), device="cuda"); cm = torch.rand((256, 3), device="cuda")
out = {"resnet257_ms": timed(lambda: eng.resnet(x, n_zero=1), 10), "temporal32x20_ms": timed(lambda: eng.temporal(phi)),
       "ief256_ms": timed(lambda: eng.ief(st)), "smpl256_ms(x1 of 3)": timed(lambda: eng.smpl(th, be, cm))}
print(json.dumps(out))
if len(sys.argv) > 2:       # temporal tiles
    for t in (5, 6, 7, 8):
        for i in range(3):
            eng.tw.block[i].conv1.tile = eng.tw.block[i].conv2.tile = t
        print("temporal tile", t, timed(lambda: eng.temporal(phi)))
