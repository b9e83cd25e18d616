"""Dev aid: what amdsmi reports on this box while the ResNet runs (clocks per XCD, socket power, the cap, throttle state)."""
import sys, time, threading
import torch
sys.path.insert(0, ".")
import amdsmi
amdsmi.amdsmi_init()
hs = amdsmi.amdsmi_get_processor_handles()
print("handles:", len(hs))
h = hs[0]
for name in ("amdsmi_get_power_cap_info", "amdsmi_get_power_info", "amdsmi_get_clock_info"):
    try:
        f = getattr(amdsmi, name)
        print(name, f(h, amdsmi.AmdSmiClkType.GFX) if "clock" in name else f(h))
    except Exception as e:
        print(name, "failed:", repr(e)[:200])
try:
    m = amdsmi.amdsmi_get_gpu_metrics_info(h)
    print({k: v for k, v in m.items() if not isinstance(v, (list, tuple)) or len(v) <= 8})
except Exception as e:
    print("metrics failed", repr(e)[:300])

from human_dynamics_amd import assets
from human_dynamics_amd.engine import HmmrEngine
eng = HmmrEngine(assets.make_synthetic_weights(0), None, dtype="f16x3")
x = torch.rand((256, 224, 224, 3), device="cuda") * 2 - 1
for _ in range(3):
    eng.resnet(x, n_zero=1)
torch.cuda.synchronize()
samples, run = [], [True]


def poll():
    while run[0]:
        t = time.perf_counter()
        try:
            m = amdsmi.amdsmi_get_gpu_metrics_info(h)
            samples.append((t, m.get("current_gfxclks"), m.get("average_gfxclk_frequency"), m.get("average_socket_power"), m.get("current_socket_power"), m.get("throttle_status"), m.get("indep_throttle_status"), m.get("temperature_hotspot")))
        except Exception as e:
            samples.append((t, repr(e)[:100]))
        time.sleep(0.002)


th = threading.Thread(target=poll); th.start()
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < 3.0:
    for _ in range(10):
        eng.resnet(x, n_zero=1)
    torch.cuda.synchronize(); n += 10
el = time.perf_counter() - t0
time.sleep(0.3)
run[0] = False; th.join()
print("%d passes in %.3f s = %.3f ms each; %d samples" % (n, el, el / n * 1e3, len(samples)))
for s in samples[:3] + samples[len(samples) // 2: len(samples) // 2 + 6] + samples[-3:]:
    print("  %.4f" % (s[0] - t0), s[1:])
