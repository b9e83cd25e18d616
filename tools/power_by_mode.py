"""Dev aid (round 6): socket power, shader clocks and energy per frame of the ResNet pass in the three operand modes, of the bare MFMA streams of
csrc/probe.hip (constant / changing operands) and of an idle GPU -- bench.SmiSampler (amdsmi, a host thread) around ~1.5 s of each.
    python tools/power_by_mode.py [frames]"""
import sys, time
import torch
sys.path.insert(0, ".")
from human_dynamics_amd import assets, _lib as L
from human_dynamics_amd.engine import HmmrEngine
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 257
w = assets.make_synthetic_weights(0)
x = torch.rand((n - 1, 224, 224, 3), device="cuda") * 2 - 1
cus = torch.cuda.get_device_properties(0).multi_processor_count


def region(label, work, seconds=1.5, frames_per_call=0, flop_per_call=0.0):
    for _ in range(3):
        work()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); work(); torch.cuda.synchronize(); one = time.perf_counter() - t0
    reps = max(3, int(seconds / max(one, 1e-5)))
    s = bench.SmiSampler(0)
    s.start()
    t0 = time.perf_counter()
    for i in range(reps):
        work()
        if i % 16 == 15:
            torch.cuda.synchronize()            # (keep the launch queue short)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    p = s.stop() or {}
    extra = ""
    if frames_per_call:
        extra = " | %.3f ms per pass = %.0f frames/s | %.4f J per frame" % (el / reps * 1e3, frames_per_call * reps / el, (p.get("joules") or 0) / (frames_per_call * reps))
    if flop_per_call:
        extra = " | %.0f TFLOP/s | %.3f pJ per FLOP (socket, everything included)" % (flop_per_call * reps / el / 1e12, (p.get("joules") or 0) / (flop_per_call * reps) * 1e12)
    print("%-44s %5.0f W mean (max %4.0f, cap %4.0f) | clocks %4.0f MHz mean, slowest XCD %4.0f | hotspot %s C%s" % (
        label, p.get("socket_w_mean") or 0, p.get("socket_w_max") or 0, p.get("power_cap_w") or 0, p.get("gfxclk_mhz_mean") or 0, p.get("gfxclk_mhz_min_xcd") or 0,
        p.get("hotspot_c_max"), extra), flush=True)


region("idle (host sleeps)", lambda: time.sleep(0.05), seconds=1.0)
for dt in ("f16x3", "bf16", "f32"):
    eng = HmmrEngine(w, None, dtype=dt)
    eng.resnet_streams = 1
    region("ResNet, %d frames, %s, one stream" % (n, dt), lambda: eng.resnet(x, n_zero=1), frames_per_call=n)
    if dt == "f16x3":
        eng.resnet_streams = 2
        region("ResNet, %d frames, %s, two parts" % (n, dt), lambda: eng.resnet(x, n_zero=1), frames_per_call=n)
        lib = eng.lib
        st = torch.cuda.current_stream().cuda_stream
        region("bare MFMA stream, constant operands", lambda: L.check(lib.hmmr_mfma_rate_probe(cus, 2500, None, st), "p"), flop_per_call=cus * 4 * 8 * 2500 * 32768.0)
        region("bare MFMA stream, changing operands", lambda: L.check(lib.hmmr_mfma_rate_probe(cus, -2500, None, st), "p"), flop_per_call=cus * 4 * 8 * 2500 * 32768.0)
    del eng
    torch.cuda.empty_cache()
