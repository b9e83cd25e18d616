"""Golden vectors for the f-3 row, produced by EXECUTING the reference's own source here
(/root/reference is not on the GPU box, so the outputs are committed as fixtures):

  reference_handoff.npz   src/util/render/nmr_renderer.py `visualize_img_orig` (camera / keypoint
                          change to the original image) and `VisRenderer.__call__`'s projection
                          (`orthographic_proj_withz_idrot` + y flip), with stand-ins for what cannot
                          be installed: neural_renderer, skimage.io, cv2 (only its output SHAPE
                          matters here).  The call that would rasterise (`visualize_img`) is replaced
                          by a recorder of its arguments.
  predcache/              files written by src/evaluation/prediction.py `get_predictions` for a fake
                          model, plus reference_predcache.json with the path helpers' outputs.

    python tests/golden/make_handoff_golden.py
"""
import json
import os
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def main():
    if not hasattr(np, "int"):
        np.int = int
    added = []

    def stub(name, mod):
        if name not in sys.modules:
            sys.modules[name] = mod
            added.append(name)
    nr = types.ModuleType("neural_renderer")
    skio = types.ModuleType("skimage.io"); skio.imread = lambda p: None
    sk = types.ModuleType("skimage"); sk.io = skio
    cv2 = types.ModuleType("cv2")
    cv2.resize = lambda img, dsize: np.zeros((dsize[1], dsize[0]) + img.shape[2:], img.dtype)
    for n, m in (("neural_renderer", nr), ("skimage", sk), ("skimage.io", skio), ("cv2", cv2)):
        stub(n, m)
    sys.path.insert(0, REF)
    try:
        from src.util.render import nmr_renderer as R
        from src.util.render.torch_utils import orthographic_proj_withz_idrot
        from src.evaluation import prediction as P
    finally:
        sys.path.remove(REF)

    # ---- (1) camera change + projection ---------------------------------------------------------
    seen = {}

    def recorder(img, cam, kp_pred, vert, renderer, **kw):
        seen.update(cam=np.array(cam), kps=np.array(kp_pred), img_size=img.shape[0])
        return None
    R.visualize_img = recorder

    class FakeRenderer(object):
        class renderer(object):
            image_size = 0
    rng = np.random.default_rng(11)
    nv, nk = 97, 25
    shapes = [(240, 320), (720, 1280), (96, 64), (300, 300), (1080, 810), (301, 120)]
    cams, kps, verts, params, new_cams, kp_orig, sizes = [], [], [], [], [], [], []
    for i, (h, w) in enumerate(shapes):
        cam = np.array([rng.uniform(0.4, 1.3), rng.uniform(-0.4, 0.4), rng.uniform(-0.4, 0.4)], np.float32)
        kp = rng.uniform(-1.2, 1.2, size=(nk, 2)).astype(np.float32)
        v = rng.normal(size=(nv, 3)).astype(np.float32)
        start_pt = rng.integers(100, 500, size=2)
        scale = float(rng.uniform(0.3, 2.5))
        max_img = 300 if i % 2 == 0 else 720
        R.visualize_img_orig(cam=cam, kp_pred=kp, vert=v, renderer=FakeRenderer(), start_pt=start_pt, scale=scale,
                             proc_img_shape=[224, 224], img=np.zeros((h, w, 3)), max_img_size=max_img)
        cams.append(cam); kps.append(kp); verts.append(v)
        params.append([h, w, start_pt[0], start_pt[1], scale, max_img])
        new_cams.append(seen["cam"]); kp_orig.append(seen["kps"]); sizes.append(seen["img_size"])
    new_cams = np.stack(new_cams)
    assert new_cams.dtype == np.float32
    proj = orthographic_proj_withz_idrot(torch.from_numpy(np.stack(verts)), torch.from_numpy(new_cams), offset_z=0)
    proj[:, :, 1] *= -1
    proj_crop = orthographic_proj_withz_idrot(torch.from_numpy(np.stack(verts)), torch.from_numpy(np.stack(cams)), offset_z=0)
    proj_crop[:, :, 1] *= -1
    np.savez_compressed(os.path.join(HERE, "reference_handoff.npz"), cams=np.stack(cams), kps=np.stack(kps),
                        verts=np.stack(verts), params=np.array(params, np.float64), new_cams=new_cams,
                        kp_orig=np.stack(kp_orig).astype(np.float64), img_size=np.array(sizes),
                        proj_verts=proj.numpy(), proj_verts_crop=proj_crop.numpy())

    # ---- (2) prediction cache --------------------------------------------------------------------
    class FakeModel(object):
        def predict_all_images(self, images):
            n = len(images)
            r = np.random.default_rng(3)
            out = {k: r.normal(size=(n,) + s).astype(np.float32) for k, s in
                   (("cams", (3,)), ("joints", (25, 3)), ("kps", (25, 2)), ("poses", (24, 3, 3)), ("shapes", (10,)),
                    ("verts", (11, 3)), ("omegas", (85,)), ("cams_delta", (2, 3)), ("verts_delta", (2, 11, 3)))}
            out["mean_image"] = np.float32(np.mean(images))
            return out
    tmp = tempfile.mkdtemp()
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        images = np.random.default_rng(4).integers(0, 256, size=(5, 4, 4, 3)).astype(np.float64)
        args = dict(load_path="models/hmmr_model.ckpt-1119816", tf_path="/data/tf_datasets/3dpw/test/downtown_arguing_00.tfrecord", p_id=1)
        P.get_predictions(FakeModel(), images, pred_dir="predictions_cache", incl_verts=True, **args)
        dst = os.path.join(HERE, "predcache")
        shutil.rmtree(dst, ignore_errors=True)
        shutil.copytree(os.path.join(tmp, "predictions_cache"), dst)
        names = {
            "pred": P.get_pred_path_name(pred_dir="predictions_cache", incl_verts=False, **args),
            "verts": P.get_pred_path_name(pred_dir="predictions_cache", incl_verts=True, **args),
            "result": P.get_result_path_name("test", args["load_path"], "pred", ["3dpw", "h36m"], pred_dir="predictions_cache"),
            "eval": P.get_eval_path_name(args["load_path"], "pred", args["tf_path"], 1, pred_dir="predictions_cache"),
            "eval_minvis": P.get_eval_path_name(args["load_path"], "pred", args["tf_path"], 1, pred_dir="predictions_cache", min_visible=6),
            "args": args,
        }
        json.dump(names, open(os.path.join(HERE, "reference_predcache.json"), "w"), indent=1)
        np.save(os.path.join(HERE, "predcache_images.npy"), images)
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)
    for n in added:
        sys.modules.pop(n, None)
    for f in ("reference_handoff.npz", "reference_predcache.json"):
        print(f, os.path.getsize(os.path.join(HERE, f)))
    for root, _, files in os.walk(os.path.join(HERE, "predcache")):
        for f in files:
            print(os.path.join(root, f), os.path.getsize(os.path.join(root, f)))


if __name__ == "__main__":
    main()
