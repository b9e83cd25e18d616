"""Golden vectors for the NON-Tester configurations of the path, produced by EXECUTING THE REFERENCE'S OWN SOURCE on
oracle/tf_shim.py (as tests/golden/make_reference_golden.py does for the Tester configuration):

    python tests/golden/make_modes_golden.py              # needs /root/reference (this container only)

  * src/models.py  batch_pred_omega / call_hmr_ief with every (use_optcam, use_delta_from_pred) combination and a
                   per-row omega_mean (lines 233-267, 299-377)
  * src/tf_smpl/batch_lbs.py  batch_global_rigid_transformation(rotate_base=True) (lines 133-194)

Output (committed): reference_modes.npz.  Weights are regenerated from seeds by the tests
(assets.make_synthetic_ief_weights), only inputs and outputs are stored.
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


def main():
    added = tf_shim.install(np.float64)
    sys.path.insert(0, REF)
    try:
        from src import models as ref_models
        from src.tf_smpl import batch_lbs as ref_lbs
        import tensorflow as tf
    finally:
        sys.path.remove(REF)
    out = {}
    rng = np.random.default_rng(21)
    B, T = 2, 3
    strips = rng.normal(size=(B, T, 2048)) * 0.7
    mean = assets.make_mean_theta(1007).astype(np.float64)
    omega_mean = np.tile(mean, (B * T, 1)) + rng.normal(size=(B * T, 85)) * 0.05      # a different start per row
    out["strips"], out["omega_mean"] = strips, omega_mean
    for optcam in (True, False):
        tf_shim.WEIGHTS = assets.make_synthetic_ief_weights(7, delta_nd=72 if optcam else 75)
        for from_pred in (True, False):
            del tf_shim.USED_VARIABLES[:]
            omega, deltas = ref_models.batch_pred_omega(
                input_features=tf.constant(strips), batch_size=B, sequence_length=T, num_output=85, is_training=False,
                omega_mean=tf.constant(omega_mean), scope="single_view_ief", predict_delta_keys=[0, -5, 5],
                use_optcam=optcam, use_delta_from_pred=from_pred)
            tag = "optcam%d_frompred%d" % (optcam, from_pred)
            out["omega_" + tag] = np.asarray(omega)
            out["delta_m5_" + tag], out["delta_p5_" + tag] = np.asarray(deltas[-5]), np.asarray(deltas[5])
    # rotate_base
    m = 5
    theta = rng.normal(size=(m * 24, 3)) * 0.5
    Rs = np.asarray(ref_lbs.batch_rodrigues(tf.constant(theta))).reshape(m, 24, 3, 3)
    Js = rng.normal(size=(m, 24, 3)) * 0.3
    for rb in (False, True):
        nj, A = ref_lbs.batch_global_rigid_transformation(tf.constant(Rs), tf.constant(Js), assets.SMPL_PARENTS, rotate_base=rb)
        out["fk_new_j_rb%d" % rb], out["fk_A_rb%d" % rb] = np.asarray(nj), np.asarray(A)
    out["fk_Rs"], out["fk_Js"] = Rs, Js
    tf_shim.uninstall(added)
    path = os.path.join(HERE, "reference_modes.npz")
    np.savez_compressed(path, **{k: np.asarray(v, np.float64) for k, v in out.items()})
    print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
