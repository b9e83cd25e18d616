"""Host timeline of the streamed predict_all_images (dev aid): python tools/stream_trace.py [dtype] [frames] [u8] [ramp]"""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from human_dynamics_amd import assets
from human_dynamics_amd.evaluation.tester import Tester
from human_dynamics_amd.evaluation.streaming import HostStreamer


class Config(object):
    load_path, batch_size, sequence_length, pred_mode, num_conv_layers = "synthetic:0", 8, 20, "pred", 3
    delta_t_values, smpl_model_path, num_kps = ["-5", "5"], "synthetic:2", 25

dt = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
u8 = len(sys.argv) > 3 and sys.argv[3] == "u8"
ramp = not (len(sys.argv) > 4 and sys.argv[4] == "noramp")
w = assets.make_synthetic_weights(0); s = assets.make_synthetic_smpl(2)
t = Tester(Config(), weights=w, smpl=s, dtype=dt, device="cuda:0")
fr = assets.make_synthetic_frames(min(n, 256), seed=1)
fr = np.concatenate([fr] * ((n + len(fr) - 1) // len(fr)))[:n]
if u8:
    fr = np.clip(np.rint((fr + 1) * 127.5), 0, 255).astype(np.uint8)
t._streamer = HostStreamer(t)
for i in range(2):
    t.predict_all_images(fr)
best = 1e9
for i in range(3):
    t0 = time.perf_counter(); t.predict_all_images(fr); best = min(best, time.perf_counter() - t0)
print("dtype %s frames %d u8 %s ramp %s: %.2f ms = %.0f fps" % (dt, n, u8, ramp, best * 1e3, n / best))
os.environ["HMMR_STREAM_TRACE"] = "1"
t.predict_all_images(fr)
if u8:
    import torch
    st = t._streamer
    d = torch.from_numpy(fr[:256]).cuda()
    for i in range(3): st._to_float(d, 256, st._float_buf(0))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(10): st._to_float(d, 256, st._float_buf(0))
    torch.cuda.synchronize(); print("crop 256 frames: %.3f ms" % ((time.perf_counter() - t0) * 100))
