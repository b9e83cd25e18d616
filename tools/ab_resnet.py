"""A/B the conv staging variants inside ONE box/run (separate processes, interleaved)."""
import json, os, subprocess, sys
code = r'''
import sys, json, time, torch
sys.path.insert(0, ".")
from human_dynamics_amd import assets
from human_dynamics_amd.engine import HmmrEngine
w = assets.make_synthetic_weights(0)
n = int(sys.argv[1]); dt = sys.argv[2]
eng = HmmrEngine(w, None, dtype=dt)
x = torch.rand((n, 224, 224, 3), device="cuda") * 2 - 1
for _ in range(3): eng.resnet(x)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): eng.resnet(x)
torch.cuda.synchronize(); ms = (time.perf_counter() - t) / 10 * 1e3
phi = torch.randn((32, 20, 2048), device="cuda")
for _ in range(3): eng.temporal(phi)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): eng.temporal(phi)
torch.cuda.synchronize(); tms = (time.perf_counter() - t) / 10 * 1e3
print(json.dumps({"resnet_ms": round(ms, 3), "temporal_ms": round(tms, 3)}))
'''
n = sys.argv[1] if len(sys.argv) > 1 else "256"
var = sys.argv[2] if len(sys.argv) > 2 else "HMMR_LIB_PATH"
vals = sys.argv[3:] or [""]
for rep in range(2):
    for dt in ("bf16", "f32"):
        for v in vals:
            env = dict(os.environ)
            env[var] = os.path.abspath(v) if var == "HMMR_LIB_PATH" and v else v
            if var == "HMMR_LIB_PATH" and not v:
                env.pop(var)
            out = subprocess.run([sys.executable, "-c", code, n, dt], env=env, capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            print(rep, dt, "%s=%s" % (var, v), line[-1] if line else out.stderr[-300:], flush=True)
