"""Dev aid: the ResNet pass at sizes the shipped tile tables do not hold (they borrow the nearest size) against freshly measured tables for exactly
those sizes.  python tools/table_new_sizes_ab.py new_tables.json [dtype]"""
import json, sys
import torch
sys.path.insert(0, ".")
from human_dynamics_amd import assets
from human_dynamics_amd.engine import HmmrEngine, DTYPES

new = json.load(open(sys.argv[1]))
dt = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
eng = HmmrEngine(assets.make_synthetic_weights(0), None, dtype=dt, autotune=False)
eng.resnet_streams = 1
shipped = dict(eng._tiles)


def timed(x, nz, reps=12):
    for _ in range(3):
        eng.resnet(x, n_zero=nz)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        eng.resnet(x, n_zero=nz)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for key in sorted(k for k in new if k.startswith("%d:" % DTYPES[dt])):
    nt = int(key.split(":")[1])
    tab = {(int(k.split(":")[0]), k.split(":")[1]): int(v) for k, v in new[key].items()}
    nz = nt % 2
    x = torch.rand((nt - nz, 224, 224, 3), device="cuda") * 2 - 1
    a, b = [], []
    for rnd in range(4):
        eng._tiles = dict(shipped); a.append(timed(x, nz))
        eng._tiles = dict(shipped); eng._tiles[nt] = tab; b.append(timed(x, nz))
    print("%s %4d frames: shipped (nearest size) %s min %.4f | own table %s min %.4f" % (dt, nt, [round(v, 3) for v in a], min(a), [round(v, 3) for v in b], min(b)))
