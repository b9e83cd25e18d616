"""Dev aid (round 6): where the time goes at the reference's own batch sizes -- Tester.predict's B=8 x T=20 = 160 frames, FeatureExtractor's
64, one 20-frame window.  ResNet as one launch sequence and as the engine's two concurrent half-batches, and the stages of the tail."""
import sys, time, json
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tools")
from human_dynamics_amd import assets
from human_dynamics_amd.evaluation.tester import Tester
from conftest_cfg import Config


def ev(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / iters, 4)


w, s = assets.make_synthetic_weights(0), assets.make_synthetic_smpl(2)
out = {}
for B, n in ((8, 160), (1, 20), (3, 64)):
    t = Tester(Config(batch_size=B), weights=w, smpl=s, dtype="f16x3")
    eng = t.engine
    x = torch.rand((n, 224, 224, 3), device="cuda") * 2 - 1
    r = {"resnet_parts1_ms": ev(lambda: eng.resnet(x, parts=1)), "resnet_parts2_ms": ev(lambda: eng.resnet(x, parts=2))}
    if n % 20 == 0:
        phi = eng.resnet(x).reshape(B, 20, -1)
        strips = t._movie_strips(phi)
        om = eng.ief(strips.reshape(n, -1))
        r.update(temporal_ms=ev(lambda: t._movie_strips(phi)), ief_ms=ev(lambda: eng.ief(strips.reshape(n, -1))),
                 records_from_omegas_ms=ev(lambda: t.records_from_omegas(om)),
                 predict_device_ms=ev(lambda: t.predict_device(x.reshape(B, 20, 224, 224, 3))))
    out["%d frames" % n] = r
    del t, eng
print(json.dumps(out, indent=1))
