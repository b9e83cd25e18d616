"""Drop-in for src/evaluation/tester.py's ``Tester`` on MI355X.

Same constructor, attributes and methods as the reference:

    Tester(config, pretrained_resnet_path='', sequence_length=None)
    .predict(images[B,T,224,224,3]) -> dict of float32 ndarrays   (tester.py:229-258)
    .predict_all_images(all_images[N,224,224,3]) -> dict[N,...]   (tester.py:260-312)
    .fov .batch_size .sequence_length .img_size .num_output

``config`` is duck-typed (load_path, batch_size, sequence_length, pred_mode,
num_conv_layers, delta_t_values, smpl_model_path, num_kps).  Differences, all
deliberate: errors raise instead of dropping into ipdb; weights are read from the
TF checkpoint-V2 files natively (no TensorFlow; or an ``.npz`` with the same
variable names, or ``synthetic[:seed]``);
``predict_all_images`` pushes every frame through the ResNet once instead of
T/g = 2.5 times (inference-mode ResNet is per-frame independent, so the
result is identical; pass ``dedup=False`` for the literal schedule).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from .. import _lib as L
from .. import assets
from ..engine import DEFAULT_DTYPE, HmmrEngine
from ..models import (batch_pred_omega, get_hallucinator_model, get_image_encoder,
                      get_temporal_encoder)
from ..omega import OmegasPred
from ..tf_smpl.batch_smpl import SMPL, load_smpl_constants

OUTPUT_KEYS = ("cams", "joints", "kps", "poses", "shapes", "verts", "omegas")


def mean_theta_from_file(path):
    """The mean theta the reference initialises `mean_param` with (tester.py:118-135): from `neutral_smpl_meanwjoints.h5` itself (round 6:
    human_dynamics_amd/hdf5_lite.py reads the deepdish / PyTables file -- HDF5 1.8 structures, Blosc-filtered arrays -- without h5py or blosc),
    or from an .npy / .npz: either the assembled [85] / [1,85] mean theta, or the h5's own fields `pose` [72] and `shape` [10].  To the fields
    the reference's assembly is applied: cam = [0.9, 0, 0], pose[:3] = [pi, 0, 0]."""
    if not os.path.exists(path):
        raise FileNotFoundError("{} doesnt exist..".format(path))
    if str(path).endswith((".h5", ".hdf5")):
        # the file the reference reads (tester.py:120-123): deepdish / PyTables on HDF5, Blosc-filtered arrays, decoded here without
        # either library (human_dynamics_amd/hdf5_lite.py)
        from .. import hdf5_lite
        v = hdf5_lite.load(path)
        if "pose" not in v or "shape" not in v:
            raise ValueError("%s has no 'pose' / 'shape' datasets (found %s)" % (path, sorted(v)))
    else:
        v = np.load(path)
    if isinstance(v, np.ndarray):
        if v.size != 85:
            raise ValueError("%s holds %d values, the mean theta has 85" % (path, v.size))
        return np.asarray(v, np.float32).reshape(1, 85)
    pose = np.array(v["pose"], np.float64).reshape(72)
    pose[:3] = 0.0
    pose[0] = np.pi
    shape = np.asarray(v["shape"], np.float64).reshape(10)
    return np.hstack(([0.9, 0.0, 0.0], pose, shape))[None].astype(np.float32)


def load_weights(load_path, resnet_path="", mean_param_path=""):
    """Checkpoint variables by name: a TensorFlow checkpoint-V2 prefix such as
    'models/hmmr_model.ckpt-1119816' (read natively, human_dynamics_amd/tf_checkpoint.py), an
    .npz with the same names, or 'synthetic[:seed]'.  ResNet variables come from `resnet_path`
    when given (tester.py:99-112).  mean_param_path: .npy / .npz initialiser of `mean_param` (mean_theta_from_file),
    used only when the checkpoint does not carry the variable -- Saver.restore overwrites it otherwise
    (tester.py:114-116).  An .h5 is the reference's own file (hdf5_lite)."""
    from .. import tf_checkpoint

    def one(path):
        if str(path).startswith("synthetic"):
            seed = int(str(path).split(":")[1]) if ":" in str(path) else 0
            return assets.make_synthetic_weights(seed)
        if os.path.exists(path) and str(path).endswith(".npz"):
            return {k: v for k, v in np.load(path).items()}
        if tf_checkpoint.is_checkpoint(path):
            return tf_checkpoint.read_checkpoint(path)
        raise FileNotFoundError("{} doesnt exist..".format(path))
    w = dict(one(load_path))
    if resnet_path:
        for k, v in one(resnet_path).items():
            if "resnet" in k:
                w[k] = v
    if mean_param_path:
        mean = mean_theta_from_file(mean_param_path)
        w.setdefault("mean_param", mean)
    return w


def window_plan(n_frames, batch_size, sequence_length, fov):
    """Index arithmetic of predict_all_images (tester.py:281-289)."""
    margin = (fov - 1) // 2
    g = sequence_length - 2 * margin
    count = int(np.ceil(n_frames / float(g * batch_size)))
    num_fill = count * batch_size * g + sequence_length - n_frames
    return margin, g, count, num_fill


class Tester(object):
    MAX_DEVICE_FRAMES = 1024      # frames per ResNet pass when the video is already in HBM
    MAX_TAIL_WINDOWS = 128        # windows per f_movie / IEF / SMPL pass of predict_windows_device

    def __init__(self, config, pretrained_resnet_path="", sequence_length=None,
                 weights=None, smpl=None, dtype=None, device=None, dedup=True, use_containers=False):
        self.config = config
        self.load_path = config.load_path
        if not config.load_path and weights is None:
            raise Exception("[!] You need to specify `load_path` to load a pretrained model")
        # Model parameters (tester.py:41-50).
        self.batch_size = config.batch_size
        self.sequence_length = sequence_length if sequence_length else config.sequence_length
        self.pred_mode = config.pred_mode
        self.num_conv_layers = config.num_conv_layers
        self.fov = self.num_conv_layers * 4 + 1
        self.delta_t_values = [int(dt) for dt in config.delta_t_values]
        self.img_size = 224
        self.num_output = 85
        self.smpl_model_path = config.smpl_model_path
        self.dedup = dedup
        self.use_containers = use_containers
        if self.pred_mode not in ("pred", "hal"):
            raise Exception("Pred mode {} not recognized".format(self.pred_mode))

        if weights is None:
            weights = load_weights(config.load_path, pretrained_resnet_path,
                                   getattr(config, "mean_param_path", "") or self._default_mean_path(config))
        if smpl is None:
            smpl = load_smpl_constants(self.smpl_model_path, checkpoint_vars=weights)
        # operand type of the GEMM stages: the reference graph is fp32 throughout (tester.py:64-66).  Default "auto": the
        # fastest rung of precision.LADDER (f16x3 -> f32 ResNet -> all f32) whose vertices / joints stay within 0.3 x
        # the 1e-4 tolerance of the exact-fp32 mode on a probe batch run here, on the device, with THESE weights
        # (precision.py; the report is self.precision).  'f16x3' / 'bf16' / 'f32' are explicit choices, not probed.
        dtype = dtype or getattr(config, "dtype", None) or "auto"
        device = device or getattr(config, "device", "cuda:0")
        kw = dict(num_conv_layers=self.num_conv_layers, delta_t_values=self.delta_t_values,
                  resnet_chunk=getattr(config, "resnet_chunk", 0))
        self.precision = None
        if dtype == "auto":
            from .. import precision
            self.engine, self.precision = precision.choose_engine(weights, smpl, device, pred_mode=self.pred_mode, **kw)
        else:
            self.engine = HmmrEngine(weights, smpl, dtype=dtype, device=device, **kw)
        if self.precision is None:
            from .. import precision
            self.precision = {"operands": precision.describe(self.engine), "rungs": []}
        self.precision["saturated"] = False
        self._rebuild = (weights, smpl, device, kw)          # what the f32 rung is built from if a call saturates the split format
        self.smpl = SMPL(smpl, engine=self.engine)
        if self.engine.num_kps != config.num_kps:
            raise ValueError("config.num_kps=%d but the SMPL regressor has %d keypoints"
                             % (config.num_kps, self.engine.num_kps))
        self.f_hal = get_hallucinator_model()
        self.f_image_enc = get_image_encoder()
        self.f_temporal_enc = get_temporal_encoder()
        if "mean_param" not in weights:
            # the reference initialises this variable from neutral_smpl_meanwjoints.h5 (tester.py:118-141) and then restores
            # it from the checkpoint (tester.py:114-116); load_weights() reads the .h5 next to the SMPL model when the checkpoint lacks it
            raise KeyError("the loaded weights have no 'mean_param' variable (the 1 x 85 mean theta both published "
                           "checkpoints carry) and no neutral_smpl_meanwjoints.h5 sits next to the SMPL model; add the variable to "
                           "the .npz, load a checkpoint that was saved by the reference, or set config.mean_param_path to the .h5 "
                           "(or an .npy / .npz of its pose / shape fields)")
        self.theta_mean = np.asarray(weights["mean_param"], np.float32).reshape(1, 85)
        self._streamer = None

    @staticmethod
    def _default_mean_path(config):
        """neutral_smpl_meanwjoints.{h5,npz,npy} next to the SMPL model, where the reference looks for the .h5
        (tester.py:120-121); '' when there is none."""
        d = os.path.dirname(str(getattr(config, "smpl_model_path", "") or ""))
        for ext in (".h5", ".npz", ".npy"):
            p = os.path.join(d, "neutral_smpl_meanwjoints" + ext)
            if d and os.path.exists(p):
                return p
        return ""

    # ------------------------------------------------------------------------
    def make_omega_pred(self, registry, use_optcam=False, batch_size=None):
        return OmegasPred(config=self.config, smpl_engine=self.engine, use_optcam=use_optcam,
                          vis_max_batch=batch_size or self.batch_size, batch_size=batch_size,
                          is_training=False, registry=registry)

    def _regress(self, movie_strips, B, T):
        """movie strips [B,T,2048] -> {0, -5, +5: OmegasPred} with SMPL computed
        (the tail of build_test_model, tester.py:196-214)."""
        registry = []
        omegas_pred = {0: self.make_omega_pred(registry, use_optcam=False, batch_size=B)}
        for dt in self.delta_t_values:
            omegas_pred[dt] = self.make_omega_pred(registry, use_optcam=True, batch_size=B)
        omegas_raw, deltas_pred = batch_pred_omega(
            input_features=movie_strips, batch_size=B, sequence_length=T,
            num_output=self.num_output, is_training=False, omega_mean=None,
            scope="single_view_ief", engine=self.engine,
            predict_delta_keys=omegas_pred.keys(), use_optcam=True, use_delta_from_pred=True)
        omegas_pred[0].append_batched(omegas_raw)
        for k in deltas_pred.keys():
            omegas_pred[k].append_batched(deltas_pred[k])
            omegas_pred[k].set_cams(omegas_pred[0].get_cams())
        OmegasPred.compute_all_smpl(registry)
        return omegas_pred

    @staticmethod
    def make_fetch_dict(omegas, suffix=""):
        return {
            "cams" + suffix: omegas.get_cams(),
            "joints" + suffix: omegas.get_joints(),
            "kps" + suffix: omegas.get_kps(),
            "poses" + suffix: omegas.get_poses_rot(),
            "shapes" + suffix: omegas.get_shapes(),
            "verts" + suffix: omegas.get_verts(),
            "omegas" + suffix: omegas.get_raw(),
        }

    def _fetch(self, omegas_pred, to_numpy=True):
        fetch = self.make_fetch_dict(omegas_pred[0])
        deltas = {}
        for delta_t, omega_delta in sorted(omegas_pred.items()):
            if delta_t == 0:
                continue
            for k, v in self.make_fetch_dict(omega_delta, suffix="_delta").items():
                deltas.setdefault(k, []).append(v)
        for k, v in deltas.items():                  # DxBxTx... --> BxTxDx...
            fetch[k] = torch.stack(v, dim=2)
        if not to_numpy:
            return fetch
        torch.cuda.synchronize(self.engine.device)
        return {k: v.float().cpu().numpy() for k, v in fetch.items()}

    # ------------------------------------------------------------------------
    def record_layout(self):
        from ..dist import record_layout
        K = self.engine.num_kps
        V = self.engine.num_verts
        fields = (("cams", (3,)), ("joints", (K, 3)), ("kps", (K, 2)), ("poses", (24, 3, 3)),
                  ("shapes", (10,)), ("verts", (V, 3)), ("omegas", (85,)))
        return record_layout(len(self.delta_t_values), fields)

    def predict_records(self, strips, out=None):
        """movie strips [n,2048] -> packed per-frame records [n, rec_len] (device):
        IEF for the present and the delta regressors, then one SMPL evaluation per
        container that writes verts/joints/kps/poses straight into the record
        (the tail of build_test_model + make_fetch_dict, tester.py:196-227, with the
        containers' cams = omega_0's cams, tester.py:211-213)."""
        return self.records_from_omegas(self.engine.ief(strips), out)   # [R, n, 85], deltas in sorted order

    def records_from_omegas(self, om, out=None):
        """omegas [R, n, 85] (present, then the deltas in sorted order) -> packed per-frame records: one SMPL
        evaluation per container, written in place.  Per-frame independent, so a rank may evaluate it for
        frames whose omegas another rank regressed (dist.ShardedPredictor, gather_mode='theta')."""
        eng = self.engine
        n = om.shape[1]
        layout, rec_len = self.record_layout()
        off = {k: (o, sz) for k, shp, o, sz in layout}
        if out is None:
            out = torch.empty((n, rec_len), dtype=torch.float32, device=eng.device)
        if n == 0:
            return out
        rows = []
        for r, key in enumerate(eng.reg_keys):
            if key == 0:
                rows.append([off[k][0] for k in OUTPUT_KEYS])
            else:
                d = r - 1
                rows.append([off[k + "_delta"][0] + d * (off[k + "_delta"][1] // len(self.delta_t_values)) for k in OUTPUT_KEYS])
        # one launch set for every container; cams / shapes / omegas are written by the keypoint kernel (no copies)
        eng.smpl_records(om.contiguous(), out, rows)
        return out

    def _movie_strips(self, img_feat_full):
        """tester.py:183-194: temporal encoder ('pred') or hallucinator ('hal')."""
        if self.pred_mode == "pred":
            return self.f_temporal_enc(is_training=False, net=img_feat_full,
                                       num_conv_layers=self.num_conv_layers, engine=self.engine)
        return self.f_hal(img_feat_full, engine=self.engine)

    def predict_device(self, images):
        """Forward pass on device tensors: images [B,T,224,224,3] -> dict of device tensors."""
        from ..dist import unpack_outputs
        B, T = images.shape[0], images.shape[1]
        I_t = self.engine.to_device(images).reshape(B * T, self.img_size, self.img_size, 3)
        img_feat, _ = self.f_image_enc(I_t, engine=self.engine, is_training=False, reuse=False)
        img_feat_full = img_feat.reshape(B, T, -1)
        movie_strips = self._movie_strips(img_feat_full)
        if self.use_containers:      # the reference's OmegasPred route (same numbers, more copies)
            return self._fetch(self._regress(movie_strips, B, T), to_numpy=False)
        rec = self.predict_records(movie_strips.reshape(B * T, -1))
        layout, _ = self.record_layout()
        return {k: v.reshape((B, T) + v.shape[1:]) for k, v in unpack_outputs(rec, layout).items()}

    def predict(self, images):
        """Runs forward pass of model.  images (BxTxHxWx3) -> dict of float32 ndarrays."""
        def run():
            out = self.predict_device(images)
            torch.cuda.synchronize(self.engine.device)
            return {k: v.float().cpu().numpy() for k, v in out.items()}
        return self._guard_saturation(run)

    # ------------------------------------------------------------------------
    def _uses_split_operands(self):
        e = self.engine
        return L.HMMR_F16X3 in (e.dtype, e.temporal_dtype, e.ief_dtype) or bool(e.sc is not None and e.sc.dirs_split)

    def _guard_saturation(self, run):
        """Run a host-facing call; if any split (f16x3) store clamped a value to the fp16 range while it ran (libhmmr_hip.so's
        sticky hmmr_run_flags: activations beyond +-65504, e.g. un-normalised frames or a checkpoint with very large trunk
        activations), the results are not the network's: warn once, switch this Tester to exact-fp32 operands for good
        (precision["saturated"] = True, precision["operands"] = "f32") and run the call again.  The dtype="auto" probe sees
        synthetic frames only; this is the check on the caller's own data."""
        if not self._uses_split_operands():
            return run()
        with self.engine.flag_scope() as scope:      # only what THIS call raises counts (the word is device-wide and sticky)
            out = run()
        if not (scope.flags & L.FLAG_SATURATED):
            return out
        import warnings
        e = self.engine
        stages = [n for n, d in (("ResNet", e.dtype), ("f_movie", e.temporal_dtype), ("IEF", e.ief_dtype)) if d == L.HMMR_F16X3]
        if e.sc is not None and e.sc.dirs_split:
            stages.append("SMPL blend product")
        what = "a NaN reached" if scope.flags & L.FLAG_NAN else "a value left the fp16 range (clamped to +-65504) in"
        warnings.warn("human_dynamics_amd: %s a split-fp16 store (stages with split operands: %s) -- repeating the call with "
                      "fp32 operands and keeping them for this Tester" % (what, ", ".join(stages)), RuntimeWarning, stacklevel=3)
        weights, smpl, device, kw = self._rebuild
        self.engine = HmmrEngine(weights, smpl, dtype="f32", device=device, **kw)
        self.smpl = SMPL(smpl, engine=self.engine)
        self._streamer = None
        self.precision.update(saturated=True, operands="f32", nan=bool(scope.flags & L.FLAG_NAN))
        return run()

    # ------------------------------------------------------------------------
    def features(self, frames, chunk=256, n_zero=0):
        """frames [N,224,224,3] (host or device) -> phi [N + n_zero,2048] on device."""
        if isinstance(frames, torch.Tensor) and frames.is_cuda:
            N, big = len(frames), self.MAX_DEVICE_FRAMES    # long device-resident videos: bounded ResNet workspace (18 MB/frame)
            if N <= big:
                return self.engine.resnet(frames, n_zero=n_zero)
            phi = torch.empty((N + n_zero, 2048), dtype=torch.float32, device=self.engine.device)
            for i in range(0, N, big):
                last = i + big >= N
                self.engine.resnet(frames[i:i + big], n_zero=n_zero if last else 0,
                                   out=phi[i:min(i + big, N) + (n_zero if last else 0)])
            return phi
        outs = []
        for i in range(0, len(frames), chunk):
            last = i + chunk >= len(frames)
            outs.append(self.engine.resnet(np.asarray(frames[i:i + chunk], np.float32),
                                           n_zero=n_zero if last else 0))
        return torch.cat(outs, dim=0)

    def predict_strips_records(self, windows, n_keep, out=None):
        """windows [W,T,2048] of frame features (padding slots hold the feature of
        the zero image) -> records of the first n_keep frames the windows keep
        (the centre g = T - 2*margin of each, tester.py:305-311)."""
        T = self.sequence_length
        margin = (self.fov - 1) // 2
        strips = self._movie_strips(windows)
        kept = strips[:, margin:T - margin].reshape(-1, strips.shape[-1])[:n_keep]
        return self.predict_records(kept, out)

    def predict_strips_omegas(self, windows, n_keep):
        """As predict_strips_records, stopping at the regressed omegas [R, n_keep, 85]."""
        T = self.sequence_length
        margin = (self.fov - 1) // 2
        strips = self._movie_strips(windows)
        kept = strips[:, margin:T - margin].reshape(-1, strips.shape[-1])[:n_keep]
        return self.engine.ief(kept)

    def predict_strips_device(self, windows, n_keep):
        from ..dist import unpack_outputs
        if n_keep == 0:
            return {}
        layout, _ = self.record_layout()
        return unpack_outputs(self.predict_strips_records(windows, n_keep), layout)

    def predict_windows_device(self, phi, phi_zero):
        """The part of predict_all_images after the ResNet: phi [N,2048] are the
        features of the N real frames, phi_zero [1,2048] the feature of the
        all-zero padding image (tester.py:285-289 pads with zero IMAGES)."""
        B, T = self.batch_size, self.sequence_length
        N = phi.shape[0]
        margin, g, count, num_fill = window_plan(N, B, T, self.fov)
        padded = torch.cat([phi_zero.expand(margin, -1), phi, phi_zero.expand(num_fill, -1)], dim=0)
        idx = (torch.arange(count * B, device=phi.device)[:, None] * g +
               torch.arange(T, device=phi.device)[None, :])          # window i = padded[i*g : i*g+T]
        wmax = self.MAX_TAIL_WINDOWS              # windows per tail pass: bounds the f_movie / IEF / SMPL workspaces
        if idx.shape[0] <= wmax or N == 0:
            return self.predict_strips_device(padded[idx], N)
        from ..dist import unpack_outputs
        layout, rec_len = self.record_layout()
        rec = torch.empty((N, rec_len), dtype=torch.float32, device=phi.device)
        for w0 in range(0, idx.shape[0], wmax):
            o0, o1 = w0 * g, min(N, (w0 + wmax) * g)
            if o0 >= N:
                break
            self.predict_strips_records(padded[idx[w0:w0 + wmax]], o1 - o0, out=rec[o0:o1])
        return unpack_outputs(rec, layout)

    def predict_all_images(self, all_images, want=None, stream=True):
        """Wrapper to predict an entire sequence with the sliding-window scheme of
        tester.py:260-312: windows of T frames every g = T - (fov-1) frames over
        the zero-image-padded video, keeping the centre g predictions of each.

        all_images: [N,224,224,3] float32 in [-1,1] (the reference's contract), or uint8 crops (converted on
        the device with the reference's arithmetic; a quarter of the upload).  Host input runs as a chunked
        three-stream pipeline (copy-in / kernels / copy-out, evaluation/streaming.py; byte-identical to the
        one-shot path, `stream=False`).  want: optional subset of the output keys to compute copies for
        (e.g. ("joints", "omegas") skips the 250 KB/frame of vertices on the way back)."""
        if self._uses_split_operands() and not getattr(self, "_in_guard", False):
            self._in_guard = True
            try:
                return self._guard_saturation(lambda: self.predict_all_images(all_images, want, stream))
            finally:
                self._in_guard = False
        N = len(all_images)
        if not self.dedup:
            return self._predict_all_images_literal(all_images)
        host_input = not (isinstance(all_images, torch.Tensor) and all_images.is_cuda)
        if host_input and stream:
            if self._streamer is None:
                from .streaming import HostStreamer
                self._streamer = HostStreamer(self)
            return self._streamer.run(all_images, want)
        if host_input and getattr(all_images, "dtype", None) == np.uint8:
            raise ValueError("uint8 input goes through the streamed path (stream=True)")
        phi = self.features(all_images, n_zero=1)        # last row: feature of the zero padding image
        out = self.predict_windows_device(phi[:N], phi[N:])
        torch.cuda.synchronize(self.engine.device)
        return {k: v.float().cpu().numpy() for k, v in out.items() if want is None or k in want}

    def predict_videos(self, videos, want=None):
        """`predict_all_images` for several videos at once -- how the demo is driven, one call per person track
        (/root/reference/demo_video.py:172, src/evaluation/tester.py:229-312): a list of host arrays [N_i,224,224,3] (float32 in
        [-1,1] or uint8 crops) -> a list of the dicts `predict_all_images` returns, each byte-identical to its own call.  The tracks run
        as ONE pipeline (evaluation/streaming.py: run_many): track k+1 uploads under track k's ResNet, track k's tail and download run
        under track k+1's ResNet -- the sustained rate of the host-in / host-out surface instead of one isolated call's latency."""
        videos = list(videos)
        if self._uses_split_operands() and not getattr(self, "_in_guard", False):
            self._in_guard = True
            try:
                return self._guard_saturation(lambda: self.predict_videos(videos, want))
            finally:
                self._in_guard = False
        if not self.dedup or any(isinstance(v, torch.Tensor) and v.is_cuda for v in videos):
            return [self.predict_all_images(v, want) for v in videos]
        if self._streamer is None:
            from .streaming import HostStreamer
            self._streamer = HostStreamer(self)
        return self._streamer.run_many(videos, want)

    def _predict_all_images_literal(self, all_images):
        B, T = self.batch_size, self.sequence_length
        N = len(all_images)
        H = W = self.img_size
        margin, g, count, num_fill = window_plan(N, B, T, self.fov)
        images_padded = np.concatenate((np.zeros((margin, H, W, 3), np.float32),
                                        np.asarray(all_images, np.float32),
                                        np.zeros((num_fill, H, W, 3), np.float32)), axis=0)
        results = {}
        for c in range(count):
            batch = np.stack([images_padded[i * g:i * g + T] for i in range(c * B, (c + 1) * B)])
            for k, v in self.predict(batch).items():
                results.setdefault(k, []).append(v)
        new_results = {}
        for k, v in results.items():
            v = np.array(v)[:, :, margin:-margin]
            new_results[k] = v.reshape((-1,) + v.shape[3:])[:N]
        return new_results
