"""Where the host-in / host-out time of predict_all_images goes (development aid).

    python tools/pcie_probe.py [dtype]
"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from human_dynamics_amd import assets
from human_dynamics_amd.evaluation.streaming import HostStreamer
from human_dynamics_amd.evaluation.tester import Tester
from bench import Cfg


def t(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dt = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
    dev = torch.device("cuda:0")
    x = np.random.default_rng(0).random((256, 224, 224, 3), dtype=np.float32) * 2 - 1
    pin = torch.empty((256, 224, 224, 3), dtype=torch.float32, pin_memory=True)
    d = torch.empty((256, 224, 224, 3), dtype=torch.float32, device=dev)
    print("threads", torch.get_num_threads())
    print("host copy pageable->pinned 154 MB: %.2f ms" % t(lambda: pin.copy_(torch.from_numpy(x))))
    print("H2D pinned 154 MB:                 %.2f ms" % t(lambda: d.copy_(pin, non_blocking=True)))
    print("H2D pageable 154 MB:               %.2f ms" % t(lambda: d.copy_(torch.from_numpy(x))))
    print("pinned alloc 65 MB:                %.2f ms" % t(lambda: torch.empty((256, 63327), dtype=torch.float32, pin_memory=True)))
    r = torch.empty((256, 63327), dtype=torch.float32, device=dev)
    ph = torch.empty((256, 63327), dtype=torch.float32, pin_memory=True)
    print("D2H pinned 65 MB:                  %.2f ms" % t(lambda: ph.copy_(r, non_blocking=True)))
    w, s = assets.make_synthetic_weights(0), assets.make_synthetic_smpl(2)
    te = Tester(Cfg(), weights=w, smpl=s, dtype=dt, device="cuda:0")
    for n in (256, 1024):
        xs = np.concatenate([x] * (n // 256))
        for chunk, staged in ((256, False), (256, True), (256, "auto")):
            te._streamer = HostStreamer(te, chunk=chunk, staged=staged)
            print("staged=%s" % (staged,), end=" ")
            ms = t(lambda: te.predict_all_images(xs), 2)
            ms2 = t(lambda: te.predict_all_images(xs, want=("joints", "omegas", "cams")), 2)
            print("%s N=%d chunk=%d: %.1f ms = %.0f fps; without verts %.1f ms = %.0f fps" % (dt, n, chunk, ms, n / ms * 1e3, ms2, n / ms2 * 1e3))
        ms = t(lambda: te.predict_all_images(xs, stream=False), 2)
        print("%s N=%d one-shot: %.1f ms = %.0f fps" % (dt, n, ms, n / ms * 1e3))
        u8 = (np.random.default_rng(1).integers(0, 256, size=xs.shape, dtype=np.uint8))
        te._streamer = HostStreamer(te, chunk=128)
        ms = t(lambda: te.predict_all_images(u8), 2)
        print("%s N=%d uint8 chunk=128: %.1f ms = %.0f fps" % (dt, n, ms, n / ms * 1e3))


if __name__ == "__main__":
    main()
