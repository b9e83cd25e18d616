"""Dev aid: the ResNet pass under two tile tables, alternating (same box, same process).  python tools/table_ab.py a.json b.json [frames] [dtype]"""
import json, sys
import torch
sys.path.insert(0, ".")
from human_dynamics_amd import assets
from human_dynamics_amd.engine import HmmrEngine, DTYPES

a, b = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
nt = int(sys.argv[3]) if len(sys.argv) > 3 else 257
dt = sys.argv[4] if len(sys.argv) > 4 else "f16x3"
key = "%d:%d" % (DTYPES[dt], nt)
ta, tb = [{(int(k.split(":")[0]), k.split(":")[1]): int(v) for k, v in t[key].items()} for t in (a, b)]
eng = HmmrEngine(assets.make_synthetic_weights(0), None, dtype=dt, autotune=False)
eng.resnet_streams = 1
n_zero = nt % 2
x = torch.rand((nt - n_zero, 224, 224, 3), device="cuda") * 2 - 1


def timed(tab, reps=10):
    eng._tiles = {nt: tab}
    for _ in range(2):
        eng.resnet(x, n_zero=n_zero)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        eng.resnet(x, n_zero=n_zero)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


res = {"a": [], "b": []}
for rnd in range(6):
    res["a"].append(round(timed(ta), 4)); res["b"].append(round(timed(tb), 4))
print(key, "a (%s):" % sys.argv[1], res["a"], "min %.4f" % min(res["a"]))
print(key, "b (%s):" % sys.argv[2], res["b"], "min %.4f" % min(res["b"]))
# per-layer: which entries of b beat a (instrumented passes)
import numpy as np
def prof(tab):
    eng._tiles = {nt: tab}
    eng.resnet(x, n_zero=n_zero, prof=True)
    return np.min([eng.resnet(x, n_zero=n_zero, prof=True)[1] for _ in range(5)], axis=0)
pa, pb = prof(ta), prof(tb)
best = dict(ta)
for slot, u, nm in eng._resnet_layers():
    if ta[(u, nm)] != tb[(u, nm)]:
        print("  unit %2d %-8s a tile %2d %.4f ms | b tile %2d %.4f ms" % (u, nm, ta[(u, nm)], pa[slot], tb[(u, nm)], pb[slot]))
        if pb[slot] < pa[slot] * 0.97:
            best[(u, nm)] = tb[(u, nm)]
print("merged (b where it wins by 3 %%): %.4f ms" % min(timed(best) for _ in range(4)))
json.dump({key: {"%d:%s" % k: v for k, v in sorted(best.items())}}, open("gpurun_out/table_ab_merged.json", "w"))
