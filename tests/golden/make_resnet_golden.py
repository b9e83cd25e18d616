"""Golden vectors for the image encoder and the hallucinator, produced by EXECUTING the reference's
own `encoder_resnet` (src/models.py:50-77) and `fc2_res` (src/models.py:270-296).

    python tests/golden/make_resnet_golden.py          # needs /root/reference (this container only)

TensorFlow 1.8 cannot be installed here.  `oracle/tf_shim.py` stands in for the TF ops and the slim /
contrib LAYERS (conv2d, batch_norm, max_pool2d, fully_connected, arg_scope: their TF-1.8 semantics,
restated), and `oracle/slim_resnet_v2.py` is a function-for-function transcription of slim's
`resnet_utils.py` / `resnet_v2.py` (the network DEFINITION that `encoder_resnet` imports).  The
reference function itself is imported from /root/reference and run unmodified, in float64, on the
synthetic weight dict (variable names = SURVEY.md App. B; every name the code looks up must exist).

This is independent arithmetic (NumPy matmuls over taps, explicit TF padding rules, slim's own
control flow) against which the oracle's PyTorch restatement (oracle/hmmr_oracle.resnet_v2_50) and the
HIP path are both checked: tests/test_reference_golden.py.

Output (committed): reference_resnet.npz -- phi of two frames (one of them all-zero: the padding image
of predict_all_images) and strided samples of slim's end points; reference_fc2_res.npz.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from human_dynamics_amd import assets          # noqa: E402
from oracle import tf_shim                     # noqa: E402


def sample(a):
    """A strided sample of an NHWC end point that keeps the fixture small but touches borders."""
    a = np.asarray(a)
    sh, sc = max(1, a.shape[1] // 8), max(1, a.shape[3] // 32)
    return a[:, ::sh, ::sh, ::sc]


def main():
    added = tf_shim.install(np.float64)
    sys.path.insert(0, REF)
    try:
        from src import models as ref_models
        import tensorflow as tf
    finally:
        sys.path.remove(REF)
    w = assets.make_synthetic_weights(0)
    tf_shim.WEIGHTS = w
    del tf_shim.USED_VARIABLES[:]
    tf_shim.COLLECTIONS.clear()

    frames = assets.make_synthetic_frames(2, seed=1)
    frames[1] = 0.0                                                   # the zero padding image
    net, scope = ref_models.encoder_resnet(tf.constant(frames.astype(np.float64)), is_training=False, reuse=False)
    phi = np.asarray(net)
    assert phi.shape == (2, 2048) and scope == "resnet_v2_50"
    (coll, items), = tf_shim.COLLECTIONS.items()
    out = {"frames_seed": np.array(1), "phi": phi, "used_variables": np.array(sorted(set(tf_shim.USED_VARIABLES)))}
    names = []
    for alias, t in items:
        out["ep:" + alias] = sample(t).astype(np.float64)
        names.append(alias)
    out["end_points"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "reference_resnet.npz"), **out)
    print("encoder_resnet: %d variables looked up, %d end points" % (len(out["used_variables"]), len(names)))

    # ---- fc2_res (pred_mode == 'hal')
    tf_shim.WEIGHTS = assets.make_synthetic_weights(0, with_hallucinator=True)
    del tf_shim.USED_VARIABLES[:]
    phi_in = np.load(os.path.join(HERE, "window_b1_t20.npz"))["phi"].astype(np.float64).reshape(1, 20, 2048)
    hal = ref_models.fc2_res(tf.constant(phi_in))
    np.savez_compressed(os.path.join(HERE, "reference_fc2_res.npz"), phi=phi_in, out=np.asarray(hal),
                        used_variables=np.array(sorted(set(tf_shim.USED_VARIABLES))))
    tf_shim.uninstall(added)
    for f in ("reference_resnet.npz", "reference_fc2_res.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
