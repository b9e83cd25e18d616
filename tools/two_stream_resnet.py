"""Experiment: ResNet of one 257-frame batch as two concurrent half-batches on two HIP streams
(development aid; tests whether stream-level concurrency fills the tile-quantisation tails)."""
import sys, time
import torch
sys.path.insert(0, ".")
from human_dynamics_amd import assets
from human_dynamics_amd.engine import HmmrEngine

w = assets.make_synthetic_weights(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 2
x = torch.rand((n, 224, 224, 3), device="cuda") * 2 - 1
engs = [HmmrEngine(w, None, dtype="bf16") for _ in range(parts)]
streams = [torch.cuda.Stream() for _ in range(parts)]
cuts = [round(i * n / parts) for i in range(parts + 1)]

def one():
    return engs[0].resnet(x, n_zero=1)

def split():
    outs = []
    cur = torch.cuda.current_stream()
    for i in range(parts):
        streams[i].wait_stream(cur)
        with torch.cuda.stream(streams[i]):
            outs.append(engs[i].resnet(x[cuts[i]:cuts[i + 1]], n_zero=1 if i == parts - 1 else 0))
    for s in streams:
        cur.wait_stream(s)
    return outs

def timed(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters * 1e3

a = one(); b = torch.cat(split(), 0)
print("identical:", torch.equal(a, b))
print("one stream   %.3f ms" % timed(one))
print("%d streams    %.3f ms" % (parts, timed(split)))
print("one stream   %.3f ms" % timed(one))
print("%d streams    %.3f ms" % (parts, timed(split)))
