"""Golden vectors produced by EXECUTING THE REFERENCE'S OWN SOURCE (akanazawa/human_dynamics).

    python tests/golden/make_reference_golden.py          # needs /root/reference (this container only)

TensorFlow 1.8 is not installable here, so `oracle/tf_shim.py` (a NumPy implementation of the
elementary TF ops those files call) is registered as `tensorflow`, and the following reference
code is imported from /root/reference and run unmodified, in float64:

  * src/tf_smpl/batch_smpl.py   SMPL.__init__ / SMPL.__call__         (on a synthetic SMPL pickle
  * src/tf_smpl/batch_lbs.py    batch_rodrigues, batch_skew,           written in the model's own
                                batch_global_rigid_transformation      format: chumpy-free arrays +
  * src/tf_smpl/projection.py   batch_orth_proj_idrot                  scipy-sparse regressors)
  * src/omega.py                OmegasPred: append_batched, set_cams, compute_all_smpl, getters
  * src/evaluation/tester.py    Tester.make_fetch_dict and Tester.predict_all_images (the sliding
                                window arithmetic, lines 260-312), driven with a stub `predict`
  * src/evaluation/eval_util.py compute_accel, compute_error_3d, compute_error_verts, ...

  * src/models.py               az_fc2_groupnorm / az_fc_block2, batch_pred_omega / call_hmr_ief /
                                hmr_ief / encoder_fc3_dropout (on the shim's slim / contrib layers)

  * src/evaluation/run_video.py process_image (crop before the path; cv2.resize restated, see (6))

Outputs (committed): reference_crops.npz, reference_smpl.npz, reference_windows.npz, reference_metrics.npz,
reference_temporal_ief.npz.
The GPU box never runs this script; the tests there only read the fixtures.
"""
import os
import pickle
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from human_dynamics_amd import assets          # noqa: E402
from oracle import tf_shim                     # noqa: E402

VSUB = 8


def write_smpl_pickle(path, c):
    """The layout SMPL.__init__ expects (batch_smpl.py:33-80)."""
    nv = c["v_template"].shape[0]
    dd = {
        "v_template": c["v_template"].astype(np.float64),
        "shapedirs": c["shapedirs"].T.reshape(nv, 3, 10).astype(np.float64),
        "posedirs": c["posedirs"].T.reshape(nv, 3, 207).astype(np.float64),
        "J_regressor": sp.csc_matrix(c["J_regressor"].T.astype(np.float64)),
        "cocoplus_regressor": sp.csc_matrix(c["cocoplus_regressor"].T.astype(np.float64)),
        "weights": c["lbs_weights"].astype(np.float64),
        "kintree_table": np.stack([np.where(c["parents"] < 0, 2 ** 32 - 1, c["parents"]).astype(np.uint32),
                                   np.arange(24, dtype=np.uint32)]),
    }
    with open(path, "wb") as f:
        pickle.dump(dd, f, protocol=2)


class Cfg(object):
    batch_size = 2
    num_kps = 25


def main():
    added = tf_shim.install(np.float64)
    sys.path.insert(0, REF)
    try:
        from src.tf_smpl.batch_smpl import SMPL
        from src.tf_smpl.projection import batch_orth_proj_idrot
        from src.omega import OmegasPred
        from src.evaluation.tester import Tester
        from src.evaluation import eval_util
        import tensorflow as tf
    finally:
        sys.path.remove(REF)

    consts = assets.make_synthetic_smpl(2)
    pkl = "/tmp/hmmr_synthetic_smpl.pkl"
    write_smpl_pickle(pkl, consts)
    smpl = SMPL(pkl)                                               # reference class, reference code

    # ---- (1) SMPL.__call__ + projection on random theta/beta ------------------------------------
    rng = np.random.default_rng(11)
    m = 6
    theta = (rng.normal(size=(m, 72)) * 0.4).astype(np.float32)
    theta[:, 0] += np.pi
    theta[0, 3:6] = 0.0                                            # the 1e-8 epsilon branch of Rodrigues
    theta[1] = 0.0                                                 # zero pose
    beta = rng.normal(size=(m, 10)).astype(np.float32)
    cams = np.concatenate([rng.uniform(0.5, 1.5, (m, 1)), rng.normal(size=(m, 2)) * 0.2], 1).astype(np.float32)
    verts, joints, Rs = smpl(tf.constant(beta.astype(np.float64)), tf.constant(theta.astype(np.float64)),
                             get_skin=True)
    kps = batch_orth_proj_idrot(joints, tf.constant(cams.astype(np.float64)))
    out = {"theta": theta, "beta": beta, "cams": cams, "verts": np.asarray(verts), "joints": np.asarray(joints),
           "Rs": np.asarray(Rs), "kps": np.asarray(kps)}

    # ---- (2) OmegasPred containers exactly as build_test_model drives them (tester.py:196-214) ----
    B, T = 2, 3
    OmegasPred.omega_instances = []
    om0 = np.concatenate([cams, theta, beta], 1)[:B * T].reshape(B, T, 85)
    deltas = {}
    for dt in (-5, 5):
        d = om0.copy()
        d[..., 3:75] += (rng.normal(size=(B, T, 72)) * 0.1).astype(np.float32)
        d[..., :3] = [1.0, 0.0, 0.0]
        deltas[dt] = d
    cfg = Cfg()
    preds = {0: OmegasPred(config=cfg, smpl=smpl, use_optcam=False, vis_max_batch=B, is_training=False)}
    for dt in (-5, 5):
        preds[dt] = OmegasPred(config=cfg, smpl=smpl, use_optcam=True, vis_max_batch=B, is_training=False)
    preds[0].append_batched(tf.constant(om0.astype(np.float64)))
    for dt in (-5, 5):
        preds[dt].append_batched(tf.constant(deltas[dt].astype(np.float64)))
        preds[dt].set_cams(preds[0].get_cams())
    OmegasPred.compute_all_smpl()
    fetch = Tester.make_fetch_dict(None, preds[0])
    for k, v in fetch.items():
        out["omg_" + k] = np.asarray(v)
    for dt in (-5, 5):
        for k, v in Tester.make_fetch_dict(None, preds[dt], suffix="_delta").items():
            out["omg_%s_%+d" % (k, dt)] = np.asarray(v)
    out["omg_omega0"] = om0
    out["omg_delta_m5"], out["omg_delta_p5"] = deltas[-5], deltas[5]
    for k in list(out):
        if "verts" in k:
            out[k] = out[k][..., ::VSUB, :]                         # keep the fixture small
    np.savez_compressed(os.path.join(HERE, "reference_smpl.npz"), **{k: np.asarray(v, np.float64) for k, v in out.items()})

    # ---- (3) the sliding window of Tester.predict_all_images, reference lines 260-312 ------------
    class FakeTester(object):
        """`self` for the unbound reference method: geometry attributes + a predict() that tags
        every (window, slot) with the index of the padded frame it was fed."""
        def __init__(self, B, T, fov):
            self.batch_size, self.sequence_length, self.fov, self.img_size = B, T, fov, 4

            self.fed = []

        def predict(self, images):
            tag = images[:, :, 0, 0, 0]                             # frame id planted in pixel (0,0,0)
            nz = (np.abs(images).sum(axis=(2, 3, 4)) > 0)
            ids = np.where(nz, tag, -1.0)
            self.fed.append(ids)                                    # what the reference fed to the network
            return {"frame_id": ids}

    wins = {}
    for n, B in ((1, 8), (24, 2), (64, 8), (65, 8), (100, 3), (256, 8)):
        ft = FakeTester(B, 20, 13)
        frames = np.zeros((n, 4, 4, 3), np.float32)
        frames[:, 0, 0, 0] = np.arange(1, n + 1)                    # ids 1..n (0 = padding)
        res = Tester.predict_all_images(ft, frames)
        wins["kept_n%d_b%d" % (n, B)] = res["frame_id"]
        wins["fed_n%d_b%d" % (n, B)] = np.concatenate(ft.fed, axis=0)   # [count*B, T] frame ids, -1 = zero image
    np.savez_compressed(os.path.join(HERE, "reference_windows.npz"), **wins)

    # ---- (4) evaluation metrics (src/evaluation/eval_util.py; pure NumPy) -------------------------
    rng = np.random.default_rng(5)
    N = 12
    gt = rng.normal(size=(N, 14, 3)) * 0.3
    R = np.linalg.qr(rng.normal(size=(3, 3)))[0]
    pred = (gt @ R.T) * 1.1 + rng.normal(size=(N, 14, 3)) * 0.02 + 0.3
    vis = (rng.random(N) > 0.2)
    e, epa = eval_util.compute_error_3d(gt, pred, vis)
    met = {"gt": gt, "pred": pred, "vis": vis.astype(np.float64), "mpjpe": np.array(e), "pa_mpjpe": np.array(epa),
           "accel": eval_util.compute_accel(pred[:, :, :]), "accel_err": eval_util.compute_error_accel(gt, pred, vis)}
    vg, vp = rng.normal(size=(N, 50, 3)), rng.normal(size=(N, 50, 3))
    met["verts_gt"], met["verts_pred"] = vg, vp
    met["verts_err"] = np.array(eval_util.compute_error_verts(vg, vp))
    met["pa_aligned0"] = eval_util.compute_similarity_transform(eval_util.align_by_pelvis(pred[0]),
                                                               eval_util.align_by_pelvis(gt[0]))
    np.savez_compressed(os.path.join(HERE, "reference_metrics.npz"), **met)

    # ---- (5) f_movie and the IEF regressors: src/models.py run on the shim's contrib layers --------
    # (pins the reference's wiring and its checkpoint variable names; the layer semantics themselves
    #  are TF's and are restated in oracle/tf_shim.py)
    sys.path.insert(0, REF)
    try:
        from src import models as ref_models
    finally:
        sys.path.remove(REF)
    w = assets.make_synthetic_weights(0)
    tf_shim.WEIGHTS = w
    del tf_shim.USED_VARIABLES[:]
    phi = np.load(os.path.join(HERE, "window_b1_t20.npz"))["phi"].astype(np.float64).reshape(1, 20, 2048)
    phi2 = np.concatenate([phi, phi[:, ::-1]], 0)                       # two different windows
    f_temporal = ref_models.get_temporal_encoder()
    strips = f_temporal(is_training=False, net=tf.constant(phi2), num_conv_layers=3)
    B, T = 2, 20
    omega_mean = np.tile(np.asarray(w["mean_param"], np.float64), (B * T, 1))
    omega, deltas = ref_models.batch_pred_omega(
        input_features=strips, batch_size=B, sequence_length=T, num_output=85, is_training=False,
        omega_mean=tf.constant(omega_mean), scope="single_view_ief", predict_delta_keys=[0, -5, 5],
        use_optcam=True, use_delta_from_pred=True)
    used = sorted(set(tf_shim.USED_VARIABLES))
    np.savez_compressed(os.path.join(HERE, "reference_temporal_ief.npz"), phi=phi2, strips=np.asarray(strips),
                        omega=np.asarray(omega), delta_m5=np.asarray(deltas[-5]), delta_p5=np.asarray(deltas[5]),
                        used_variables=np.array(used))
    print("variables the reference code looked up:", len(used))

    # ---- (6) the crop before the path: process_image (src/evaluation/run_video.py:56-107) ----------
    # executed from the reference with stubs for what cannot be installed: skimage.io.imread returns
    # our synthetic frame, cv2.resize is the restated bilinear of oracle/preprocess_oracle.py, the NMR
    # renderer module is an empty stand-in, and np.int (removed in NumPy 2) is int.
    import types
    from oracle import preprocess_oracle as PO
    if not hasattr(np, "int"):
        np.int = int
    cv2 = sys.modules["cv2"]
    cv2.resize = lambda img, dsize: PO.cv2_resize_linear(img, dsize)
    frames = {}
    skio = types.ModuleType("skimage.io"); skio.imread = lambda path: frames[path]
    sk = types.ModuleType("skimage"); sk.io = skio
    nmr = types.ModuleType("src.util.render.nmr_renderer")
    nmr.VisRenderer = nmr.visualize_img = nmr.visualize_img_orig = None
    for name, mod in (("skimage", sk), ("skimage.io", skio), ("src.util.render.nmr_renderer", nmr)):
        if name not in sys.modules:
            sys.modules[name] = mod
            added.append(name)
    sys.path.insert(0, REF)
    try:
        from src.evaluation import run_video as ref_rv
    finally:
        sys.path.remove(REF)
    rng = np.random.default_rng(9)
    crops, params, srcs = [], [], []
    H, W = 96, 128
    for i, (cx, cy, sc) in enumerate([(64.3, 40.2, 1.7), (5.0, 90.0, 2.3), (120.0, 3.0, 0.9), (70.0, 50.0, 3.1)]):
        fr = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
        frames["f%d" % i] = fr
        out = ref_rv.process_image("f%d" % i, np.array([cx, cy, sc]))
        assert out["image"].shape == (224, 224, 3), out["image"].shape
        crops.append(out["image"]); srcs.append(fr)
        params.append([cx, cy, sc, out["center"][0], out["center"][1], out["start_pt"][0], out["start_pt"][1]])
    np.savez_compressed(os.path.join(HERE, "reference_crops.npz"), frames=np.stack(srcs),
                        crops=np.stack(crops).astype(np.float32), params=np.array(params, np.float64))

    tf_shim.uninstall(added)
    for f in ("reference_crops.npz", "reference_smpl.npz", "reference_windows.npz", "reference_metrics.npz", "reference_temporal_ief.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
