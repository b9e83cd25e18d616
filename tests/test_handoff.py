"""the rasteriser hand-off and the prediction-cache wire format, pinned to
outputs of the reference's own code (tests/golden/make_handoff_golden.py)."""
import json
import os
import pickle
import shutil

import numpy as np
import pytest

from oracle import handoff_oracle as HO

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "reference_handoff.npz"))


def test_oracle_camera_change_matches_reference(gold):
    for i, (h, w, sx, sy, scale, max_img) in enumerate(gold["params"]):
        cam, kp, size = HO.orig_camera(gold["cams"][i], gold["kps"][i], [sx, sy], scale, 224, (int(h), int(w)), int(max_img))
        assert size == gold["img_size"][i]
        assert np.allclose(cam, gold["new_cams"][i], rtol=2e-6, atol=2e-6), i
        assert np.allclose(kp, gold["kp_orig"][i], rtol=2e-6, atol=2e-6), i
    assert np.array_equal(HO.project(gold["verts"], gold["new_cams"]), gold["proj_verts"])
    assert np.array_equal(HO.project(gold["verts"], gold["cams"]), gold["proj_verts_crop"])


def test_host_geometry_rows(gold):
    from human_dynamics_amd.util.render.handoff import orig_image_geometry
    for i, (h, w, sx, sy, scale, max_img) in enumerate(gold["params"]):
        g = orig_image_geometry({"start_pt": [sx, sy], "scale": scale, "im_shape": [224, 224]}, (int(h), int(w)), int(max_img))
        size, s_orig = HO.squared_size(int(h), int(w), int(max_img))
        assert g[4] == gold["img_size"][i] == size
        assert g[0] == (1.0 / scale) * s_orig and tuple(g[1:4]) == (sx, sy, 224.0)


def test_prediction_cache_reads_and_writes_the_reference_format(tmp_path):
    from human_dynamics_amd.evaluation import prediction as P
    names = json.load(open(os.path.join(GOLD, "reference_predcache.json")))
    args = names["args"]
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        # path helpers
        assert list(P.get_pred_path_name(pred_dir="predictions_cache", incl_verts=False, **args)) == names["pred"]
        assert list(P.get_pred_path_name(pred_dir="predictions_cache", incl_verts=True, **args)) == names["verts"]
        assert P.get_result_path_name("test", args["load_path"], "pred", ["3dpw", "h36m"], pred_dir="predictions_cache") == names["result"]
        assert P.get_eval_path_name(args["load_path"], "pred", args["tf_path"], 1, pred_dir="predictions_cache") == names["eval"]
        assert P.get_eval_path_name(args["load_path"], "pred", args["tf_path"], 1, pred_dir="predictions_cache",
                                    min_visible=6) == names["eval_minvis"]

        class Model(object):           # the fake model of the golden generator
            calls = 0

            def predict_all_images(self, images):
                Model.calls += 1
                assert float(np.max(images)) <= 1.0 and float(np.min(images)) >= -1.0    # mapped to [-1, 1]
                n = len(images)
                r = np.random.default_rng(3)
                out = {k: r.normal(size=(n,) + s).astype(np.float32) for k, s in
                       (("cams", (3,)), ("joints", (25, 3)), ("kps", (25, 2)), ("poses", (24, 3, 3)), ("shapes", (10,)),
                        ("verts", (11, 3)), ("omegas", (85,)), ("cams_delta", (2, 3)), ("verts_delta", (2, 11, 3)))}
                out["mean_image"] = np.float32(np.mean(images))
                return out
        images = np.load(os.path.join(GOLD, "predcache_images.npy"))
        mine = P.get_predictions(Model(), images, pred_dir="mine", incl_verts=True, **args)
        assert Model.calls == 1
        ref_dir = os.path.join(GOLD, "predcache", "hmmr_model.ckpt-1119816")
        for fn in sorted(os.listdir(ref_dir)):
            ref = pickle.load(open(os.path.join(ref_dir, fn), "rb"))
            got = pickle.load(open(os.path.join("mine", "hmmr_model.ckpt-1119816", fn), "rb"))
            assert sorted(ref) == sorted(got), fn
            for k in ref:
                assert np.array_equal(np.asarray(ref[k]), np.asarray(got[k])), (fn, k)
        # a cache written by the reference is a hit for us (no model call), verts merged on request
        shutil.copytree(os.path.join(GOLD, "predcache"), "predictions_cache", dirs_exist_ok=True)
        hit = P.get_predictions(None, images, pred_dir="predictions_cache", incl_verts=True, **args)
        assert sorted(hit) == sorted(mine)
        for k in hit:
            assert np.array_equal(np.asarray(hit[k]), np.asarray(mine[k])), k
        no_verts = P.get_predictions(None, images, pred_dir="predictions_cache", incl_verts=False, **args)
        assert not any("vert" in k for k in no_verts) and no_verts["p_id"] == 1
    finally:
        os.chdir(cwd)


@pytest.mark.gpu
def test_render_handoff_kernel_matches_reference(gold, gpu_device):
    import torch
    from human_dynamics_amd.util.render.handoff import orig_image_geometry, rasteriser_inputs
    cams = torch.from_numpy(gold["cams"]).to(gpu_device)
    verts = torch.from_numpy(gold["verts"]).to(gpu_device)
    kps = torch.from_numpy(gold["kps"]).to(gpu_device)
    geom = np.stack([orig_image_geometry({"start_pt": [sx, sy], "scale": sc, "im_shape": [224, 224]}, (int(h), int(w)), int(mx))
                     for h, w, sx, sy, sc, mx in gold["params"]])
    out = rasteriser_inputs(cams, verts, kps, geom)
    assert np.allclose(out["cams"].cpu().numpy(), gold["new_cams"], rtol=2e-6, atol=2e-6)
    assert np.allclose(out["kps"].cpu().numpy(), gold["kp_orig"], rtol=2e-6, atol=2e-6)
    assert np.allclose(out["proj_verts"].cpu().numpy(), gold["proj_verts"], rtol=1e-5, atol=1e-5)
    # the projection itself is bit-exact: fp32 mul(add), no fma contraction (crop rendering keeps the camera)
    crop = rasteriser_inputs(cams, verts, kps, None)
    assert np.array_equal(crop["proj_verts"].cpu().numpy(), gold["proj_verts_crop"])
    assert np.array_equal(crop["cams"].cpu().numpy(), gold["cams"]) and np.array_equal(crop["kps"].cpu().numpy(), gold["kps"])
    assert np.array_equal(out["proj_verts"].cpu().numpy(), HO.project(gold["verts"], out["cams"].cpu().numpy()))


@pytest.mark.gpu
def test_render_handoff_reads_the_packed_records_in_place(weights, smpl_consts, gpu_device):
    """End of the path -> rasteriser inputs without leaving the device: records of Tester.predict_records
    go straight into hmmr_render_handoff through row strides."""
    import torch
    from conftest import Config
    from human_dynamics_amd import assets, dist as hd
    from human_dynamics_amd.evaluation.tester import Tester
    from human_dynamics_amd.util.render.handoff import rasteriser_inputs_from_records
    t = Tester(Config(batch_size=2), weights=weights, smpl=smpl_consts, dtype="f32", device=gpu_device)
    frames = torch.from_numpy(assets.make_synthetic_frames(24, seed=4)).to(gpu_device)
    sp = hd.ShardedPredictor(t, 24, 0, 1)
    rec = sp.run(frames)
    rng = np.random.default_rng(2)
    geom = np.stack([[rng.uniform(0.4, 2.0), rng.integers(100, 400), rng.integers(100, 400), 224.0, rng.integers(200, 720)]
                     for _ in range(24)])
    out = rasteriser_inputs_from_records(rec, sp.layout, geom)
    un = hd.unpack_outputs(rec, sp.layout)
    cams, verts, kps = (un[k].cpu().numpy() for k in ("cams", "verts", "kps"))
    for i in range(24):
        # geometry rows -> the oracle's arguments (undo_scale = 1/scale with no down-scale)
        cam, kp, _ = HO.orig_camera(cams[i], kps[i], geom[i, 1:3], 1.0 / geom[i, 0], 224, (int(geom[i, 4]), int(geom[i, 4])), 10 ** 6)
        assert np.allclose(out["cams"][i].cpu().numpy(), cam, rtol=2e-6, atol=2e-6)
        assert np.allclose(out["kps"][i].cpu().numpy(), kp, rtol=2e-6, atol=1e-5)
    assert np.array_equal(out["proj_verts"].cpu().numpy(), HO.project(verts, out["cams"].cpu().numpy()))
