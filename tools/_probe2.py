import sys, os, time
import numpy as np, torch
sys.path.insert(0, ".")
from human_dynamics_amd import assets
from human_dynamics_amd.evaluation.streaming import HostStreamer
from human_dynamics_amd.evaluation.tester import Tester
from bench import Cfg
w, s = assets.make_synthetic_weights(0), assets.make_synthetic_smpl(2)
te = Tester(Cfg(), weights=w, smpl=s, dtype="bf16x3", device="cuda:0")
xs = np.random.default_rng(0).random((1024, 224, 224, 3), dtype=np.float32) * 2 - 1
print(open("/sys/fs/cgroup/cpu.max").read() if os.path.exists("/sys/fs/cgroup/cpu.max") else "no cpu.max", os.cpu_count(), len(os.sched_getaffinity(0)))
for chunk in (64, 256):
    te._streamer = HostStreamer(te, chunk=chunk)
    te.predict_all_images(xs); te.predict_all_images(xs)
    os.environ["HMMR_STREAM_TRACE"] = "1"
    print("==== chunk", chunk)
    te.predict_all_images(xs)
    del os.environ["HMMR_STREAM_TRACE"]
