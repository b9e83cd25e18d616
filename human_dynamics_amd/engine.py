"""Device engine: owns the packed weights, the HBM workspaces and the four
stage calls of libhmmr_hip.so.  Everything above it (models.py, omega.py,
evaluation/tester.py) mirrors the reference's Python interface and only
shuffles tensors.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import time

import numpy as np
import torch

from . import _lib as L
from . import assets, devflags, packing

DTYPES = {"f32": L.HMMR_F32, "fp32": L.HMMR_F32, "float32": L.HMMR_F32,
          "bf16": L.HMMR_BF16, "bfloat16": L.HMMR_BF16,
          "f16x3": L.HMMR_F16X3, "split": L.HMMR_F16X3}
DTYPE_NAMES = {L.HMMR_F32: "f32", L.HMMR_BF16: "bf16", L.HMMR_F16X3: "f16x3"}
# The explicit fast mode: split-fp16 operands (hi/lo pairs, three fp16 MFMAs per product, fp32 accumulate) --
# the fastest mode whose end-to-end vertices / joints stay within the reference tolerance of 1e-4.
# 'bf16' is the opt-in throughput mode (vertex error ~7e-3), 'f32' the exact-fp32 MFMA mode.
DEFAULT_DTYPE = "f16x3"
TILE_TABLES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tile_tables.json")


def set_debug(stem_route=0, stem_no_conv1=0, gemm_probe=0, smpl_blend_mfma=0, ief_no_group=0, pair_min_pixels=0, pair_two_tile_min=0, pair_form=0):
    """Development switches of libhmmr_hip.so (hmmr_debug_t; process-wide, zeros = product defaults).
    pair_min_pixels: the unit-pair switch of hmmr_resnet50_fwd (0 = 12000 pixels, 1 = always the pair kernel, 2**31 - 1 = never);
    pair_two_tile_min: tiles from which a block-2 unit pair runs two tiles per workgroup (0 = 512, 1 = always, 2**31 - 1 = never);
    pair_form: 0 = the one-wave-per-SIMD unit pair of round 4 (the default), 2 = the wave-specialised form of round 6 (two waves per SIMD; same bits)."""
    d = L.Debug()
    d.stem_route, d.stem_no_conv1, d.gemm_probe = int(stem_route), int(stem_no_conv1), int(gemm_probe)
    d.smpl_blend_mfma, d.ief_no_group, d.pair_min_pixels = int(smpl_blend_mfma), int(ief_no_group), int(pair_min_pixels)
    d.pair_two_tile_min, d.pair_form = int(pair_two_tile_min), int(pair_form)
    L.load().hmmr_set_debug(C.byref(d))


def _debug_from_env():
    """devflags STEM / STEM_C1 / GEMM_PROBE (tools/ A/B scripts) -> hmmr_set_debug; the library itself never reads the
    environment."""
    e = devflags.get("STEM")
    c1 = devflags.get("STEM_C1")
    probe = int(devflags.get("GEMM_PROBE") or 0)
    if e or c1 == "0" or probe or _debug_from_env.was_set:
        set_debug(stem_route={"u": 1, "f": 2}.get(e[:1], 0), stem_no_conv1=int(c1 == "0"), gemm_probe=probe)
        _debug_from_env.was_set = bool(e or c1 == "0" or probe)


_debug_from_env.was_set = False


def _dt(d):
    if isinstance(d, str):
        if d == "bf16x3":              # rounds 1-2 called the split mode by its (then bf16) halves
            import warnings
            warnings.warn("dtype 'bf16x3' is now 'f16x3' (split operands with fp16 halves since round 3)", DeprecationWarning, stacklevel=3)
            d = "f16x3"
        if d not in DTYPES:
            raise ValueError("unknown operand mode %r: use one of %s" % (d, ", ".join(sorted(set(DTYPES)))))
        return DTYPES[d]
    return int(d)


class _Workspace(object):
    """Grow-only HBM scratch buffer."""

    def __init__(self, device):
        self.device, self.buf = device, None

    def get(self, nbytes):
        if self.buf is None or self.buf.numel() < nbytes:
            self.buf = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        return self.buf


class HmmrEngine(object):
    """weights: dict of checkpoint-named arrays (assets.py); smpl: dict in the
    src/tf_smpl layout.  dtype: GEMM operand mode of ResNet / temporal / IEF: 'f16x3' (default: split-fp16 hi/lo
    operands, three fp16 MFMAs per product -- the mode inside the reference tolerance), 'bf16' or 'f32'; SMPL is
    always fp32 at its boundaries (its blend-shape product runs on split-fp16 MFMAs)."""

    def __init__(self, weights, smpl, dtype=DEFAULT_DTYPE, device="cuda:0", num_conv_layers=3,
                 delta_t_values=(-5, 5), joint_type="cocoplus", resnet_chunk=0,
                 temporal_dtype=None, ief_dtype=None, autotune=True, fold_sc=None, fuse_tail=None, patch_3x3=None,
                 unit_pair=None, b1_stream=None, b1_unit=None, stem_conv1=True, stream_1x1=None):
        self.lib = L.load()
        _debug_from_env()
        if not torch.cuda.is_available():
            raise L.HmmrError("HmmrEngine needs a HIP device (torch.cuda.is_available() is False)")
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.dtype = _dt(dtype)
        self.temporal_dtype = self.dtype if temporal_dtype is None else _dt(temporal_dtype)
        self.ief_dtype = self.dtype if ief_dtype is None else _dt(ief_dtype)
        self.resnet_chunk = int(devflags.get("RESNET_CHUNK") or resnet_chunk)
        self.store = packing.DeviceStore(self.device)
        self.num_conv_layers = num_conv_layers
        self.delta_keys = sorted(int(d) for d in delta_t_values)
        # packer choices: constructor arguments, defaults overridable by the development switches of devflags.py
        fuse = tuple(b for b in devflags.get("FUSE_PREACT").split(",") if b)
        tail = {"0": False, "1": True}.get(devflags.get("FUSE_TAIL"), devflags.get("FUSE_TAIL"))
        if fuse_tail is not None:
            tail = fuse_tail
        fsc = {"0": False, "1": True}.get(devflags.get("FUSE_SC"), "all")
        pfirst = devflags.get("PREACT_FIRST") != "0"
        fold = {"0": False, "1": True}.get(devflags.get("FOLD_SC"), None) if fold_sc is None else fold_sc
        # every stage is packed only when its variables exist: a ResNet-only checkpoint (hmr_noS5.ckpt-642561, what
        # FeatureExtractor is given: src/datasets/resnet_extractor.py:31-40) has no AZ_FC_* / single_view_ief* names
        w = weights if weights is not None else {}
        # (round 6) a NaN / inf variable is refused here, by name: the clamp of a split store would turn what it produces into finite
        # numbers, and a pre-activation constant is not behind any of the kernels' range checks (hmmr_run_flags: HMMR_FLAG_NAN)
        for name, v in w.items():
            a_ = np.asarray(v)
            if a_.dtype.kind == "f" and not np.isfinite(a_).all():
                raise ValueError("variable %r holds %d non-finite value(s)" % (name, int((~np.isfinite(a_)).sum())))
        self.rw = (packing.pack_resnet(w, self.dtype, self.store, fuse_preact_blocks=fuse, fuse_tail=tail, fuse_sc=fsc,
                                       fuse_preact_first=pfirst, fold_sc=fold,
                                       patch_3x3=int(devflags.get("PATCH_3X3")) if patch_3x3 is None else patch_3x3,
                                       unit_pair=({"0": False, "1": True}.get(devflags.get("UNIT_PAIR"), devflags.get("UNIT_PAIR"))
                                                  if unit_pair is None else unit_pair),
                                       b1_stream=(devflags.get("B1_STREAM") == "1") if b1_stream is None else b1_stream,
                                       b1_unit=(devflags.get("B1_UNIT") == "1") if b1_unit is None else b1_unit,
                                       stem_conv1=stem_conv1,
                                       stream_1x1=(devflags.get("STREAM_1X1") == "1") if stream_1x1 is None else stream_1x1)
                   if "resnet_v2_50/conv1/weights" in w else None)
        self.tw = (packing.pack_temporal(w, self.temporal_dtype, self.store, num_conv_layers)
                   if assets.temporal_scopes(0)[1] + "/weights" in w else None)
        self.hw = packing.pack_hallucinator(w, self.temporal_dtype, self.store)
        if "single_view_ief/3D_module/fc1/weights" in w and "mean_param" in w:
            self.iw, self.reg_keys = packing.pack_ief(w, self.ief_dtype, self.store, self.delta_keys)
        else:
            self.iw, self.reg_keys = None, [0]
        # an all-fp32 engine is the exact-arithmetic reference (the probe's last rung, the saturation fallback): its SMPL blend product
        # stays on fp32 operands too, so nothing in it can clamp
        all_f32 = (self.dtype, self.temporal_dtype, self.ief_dtype) == (L.HMMR_F32,) * 3
        self.sc = packing.pack_smpl(smpl, self.store, joint_type, split=not all_f32) if smpl is not None else None
        self.num_kps = self.sc.num_kps if self.sc is not None else assets.NUM_KPS
        self.num_verts = self.sc.num_verts if self.sc is not None else assets.NUM_VERTS
        self._ws = {k: _Workspace(self.device) for k in ("resnet", "temporal", "ief", "smpl", "hal")}   # + "resnet<i>" per side stream
        # per-layer conv tiles of the ResNet.  Shipped tables (tile_tables.json: measured on an MI355X for the batch sizes
        # the BASELINE configurations produce, per operand mode) make the first call cost nothing; a batch size with no
        # shipped table within 30 % is tuned on first use (_tune_resnet), autotune="force" tunes every size and ignores
        # the shipped tables, autotune=False never tunes.  The tile never changes a result bit.
        at = devflags.get("AUTOTUNE")
        self.autotune = False if (not autotune or at == "0") else ("force" if (autotune == "force" or at == "force") else True)
        self._tiles, self._shipped = {}, set()
        self.tune_log = []       # [(frames, ms)] of every tuning pass run by this engine: never silent
        # optional on-disk copy of the tuned tables (devflags TILE_CACHE): profiling runs load it so that no tuning
        # pass ends up inside the rocprofv3 trace; tools/make_tile_tables.py writes the shipped file through it
        self._tile_cache = devflags.get("TILE_CACHE")
        sources = []
        if self.autotune != "force" and devflags.get("TILE_TABLE") != "0" and os.path.exists(TILE_TABLES):
            sources.append((TILE_TABLES, True))
        if self._tile_cache and os.path.exists(self._tile_cache):
            sources.append((self._tile_cache, False))
        for path, shipped in sources:
            import json
            for key, tab in json.load(open(path)).items():
                if key.startswith("_"):
                    continue
                dt, nt = key.split(":")
                if int(dt) == self.dtype:
                    self._tiles[int(nt)] = {(int(k.split(":")[0]), k.split(":")[1]): int(v) for k, v in tab.items()}
                    if shipped:
                        self._shipped.add(int(nt))
        # concurrent half-batches (see resnet()); env: dev A/B switch
        self.resnet_streams = int(devflags.get("RESNET_STREAMS"))
        self._side_streams = []

    # -- helpers ---------------------------------------------------------------
    @staticmethod
    def _need(packed, names):
        if packed is None:
            raise L.HmmrError("the loaded weights have no %s variables: this stage was not packed" % names)

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    # flags that a scoped reader (flag_scope: Tester's saturation guard, the operand-mode probe) found raised BEFORE its own work and
    # took out of the device word so that it would not mistake them for its own: run_flags() keeps reporting them, per device
    _carried_flags = {}

    def run_flags(self, clear=False):
        """Sticky run flags of this engine's device (hmmr_run_flags; _lib.FLAG_SATURATED: a split store clamped a value to the
        fp16 range since the last clear).  Synchronises with the device."""
        key = (self.device.type, self.device.index)
        v = self._device_flags(clear) | HmmrEngine._carried_flags.get(key, 0)
        if clear:
            HmmrEngine._carried_flags.pop(key, None)
        return v

    def _device_flags(self, clear):
        torch.cuda.synchronize(self.device)
        with torch.cuda.device(self.device):
            v = C.c_uint(0)
            L.check(self.lib.hmmr_run_flags(C.byref(v), int(bool(clear))), "hmmr_run_flags")
        return int(v.value)

    def flag_scope(self):
        """Context manager around a piece of work whose OWN flags are wanted: on entry whatever the (device-wide, sticky) word already
        holds is moved aside -- it stays visible to run_flags() -- and `.flags` after the block holds what was raised inside it (the word
        is left clear).  The flags are per device, not per engine: earlier work of any engine (the operand-mode probe's rejected rungs,
        raw engine calls, another Tester) must neither demote this caller nor be erased by it."""
        return _FlagScope(self)

    def to_device(self, a, dtype=torch.float32):
        if isinstance(a, torch.Tensor):
            return a.to(self.device, dtype).contiguous()
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.device).to(dtype).contiguous()

    # -- ResNet launch tuning ----------------------------------------------------
    _TUNE_TILES = tuple(int(t) for t in devflags.get("TUNE_TILES").split(","))   # hmmr_conv_desc_t.tile candidates (8-wave 128x128 / 128x64, 4-wave 64x64 / 128x128 / 128x64, 8-wave ping-pong 256x128 / 128x256)
    _TUNE_MIN_FRAMES = 32
    # two contiguous parts on concurrent streams from this many frames on (tools/resnet_parts_small.py, profiles/r06t_resnet_parts_*.log, f16x3: one stream /
    # two parts 3.35 / 3.44 ms at 128 frames, 4.01 / 4.17 at 160 -- the reference's default Tester.predict, B = 8 x T = 20 -- 5.21 / 4.80 at 192, 6.47 / 6.19 at 257)
    _SPLIT_MIN_FRAMES = 176

    def _resnet_layers(self):
        """[(profile slot, unit index, layer name)] in launch order (csrc/resnet.hip)."""
        out, slot = [], 3
        for u in range(L.RESNET_UNITS):
            names = (["shortcut"] if self.rw.unit[u].shortcut.w else []) + ["conv1", "conv2", "conv3"]
            for nm in names:
                out.append((slot, u, nm))
                slot += 1
        return out

    def _layer_of(self, u, nm):
        """the hmmr_layer_t whose .tile the launch of (unit, layer name) reads (csrc/resnet.hip)"""
        U = self.rw.unit[u]
        if nm == "conv3" and U.c3sc.w:
            return U.c3sc
        return getattr(U, nm)

    @staticmethod
    def _tile_for(lay, cand, cout, dtype=None, nm="conv2"):
        """hmmr_layer_t.tile for candidate `cand` on a layer with `cout` output columns: 0 (the library's choice) where
        the tile does not fit; a layer packed chunk-major (k_order 1, the 3x3 patch kernels) runs tiles 9 / 10, the patch
        forms of 7 / 8, or 11, the 256x128 tile without a load segment; a k_order 2 layer (csrc/conv3x3_stream.hip) takes the
        tuner's candidates as its own tile shapes 13 .. 18; a k_order 2 layer with a 1x1 filter (any layer name but conv2; csrc/conv1x1_stream.hip)
        tiles 22 .. 26, its conv3 form 24 .. 26."""
        if lay.k_order == 2 and nm != "conv2":
            t = {5: 22, 6: 23, 3: 24, 1: 25, 2: 26, 4: 29}.get(cand, cand if (22 <= cand <= 26 or cand == 29) else 0)
            return 0 if (nm == "conv3" and t in (22, 23, 29)) else t
        if lay.k_order == 2:                                 # the stream kernel's tiles: 12 .. 18 and 21 (128-channel tiles), 19 / 20 (64 channels)
            if cout == 64:
                return {5: 19, 6: 20}.get(cand, cand if cand in (19, 20) else 0)
            t = {5: 13, 6: 14, 3: 15, 1: 16, 2: 17, 7: 18, 8: 12, 11: 21, 4: 27, 9: 28}.get(cand, cand if (12 <= cand <= 18 or cand in (21, 27, 28)) else 0)
            return 0 if (t in (21, 27, 28) and dtype == L.HMMR_BF16) else t       # (the 7 x 1, 2 x 2 and 3 x 2 wave tiles are built for split tensors)
        if lay.k_order:
            cand = {7: 9, 8: 10}.get(cand, cand)
            if cand == 11 and dtype == L.HMMR_BF16:          # the tile without a load segment is written for split operands
                return 0
            return cand if (cand in (9, 11) and cout % 128 == 0) or (cand == 10 and cout % 256 == 0) else 0
        # the generic kernel's tiles are 1, 2, 3, 5 .. 8: anything else (the tuner's candidate 4 is a stream-kernel shape only; a cached table written under another packing configuration, e.g. 22 .. 26
        # of a 1x1 stream layer read back with STREAM_1X1 off) becomes the library's choice instead of a 'bad tile' error in the pass
        if cand not in (1, 2, 3, 5, 6, 7, 8) or (cand in (1, 5, 7) and cout % 128) or (cand == 8 and cout % 256):
            return 0
        return cand

    def _layer_cout(self, u, nm):
        U = self.rw.unit[u]
        if nm == "shortcut" and U.sc_c1.w:          # shortcut + conv1 as one column-split GEMM: depth + base columns
            return U.depth + U.base
        return U.base if nm in ("conv1", "conv2") else U.depth

    def _set_tiles(self, table):
        for (u, nm), t in table.items():
            self._layer_of(u, nm).tile = self._tile_for(self._layer_of(u, nm), int(t), self._layer_cout(u, nm), self.dtype, nm)

    def _needs_tuning(self, nt):
        """A tuning pass is due for batch size nt: tuning is on, the size is worth it, it has no table of its own and (unless
        forced) no shipped or tuned table within 30 % of it; at most 8 passes per engine."""
        if not self.autotune or nt < self._TUNE_MIN_FRAMES or nt in self._tiles or torch.cuda.is_current_stream_capturing():
            return False
        if len(self.tune_log) >= 8:
            return False
        if self.autotune == "force":
            return True
        return not any(max(k, nt) <= 1.3 * min(k, nt) for k in self._tiles)

    def _tune_resnet(self, images, n, n_zero, ws_key="resnet", reps=3):
        """Pick hmmr_layer_t.tile for every ResNet conv at this batch size: `reps` instrumented passes per
        candidate tile (the first is a warm-up; every layer timed in place, behind its real producer), fastest wins per layer.
        The tile never changes a result bit (each output element is one fixed-order K reduction), it
        only moves the balance between tile-count quantisation, occupancy and operand reuse, which
        flips between layers as the batch grows.  ~90 ms once per batch size."""
        layers = self._resnet_layers()
        t_start = time.perf_counter()
        nt = n + n_zero
        nbytes = self.lib.hmmr_resnet50_workspace_bytes(nt, self.dtype)
        ws = self._ws.setdefault(ws_key, _Workspace(self.device)).get(nbytes)     # the workspace of the pass being tuned
        phi = torch.empty((nt, 2048), dtype=torch.float32, device=self.device)
        src = images.data_ptr() if n else None
        best = {}
        for cand in (0,) + self._TUNE_TILES:
            for slot, u, nm in layers:
                lay = self._layer_of(u, nm)
                lay.tile = self._tile_for(lay, cand, self._layer_cout(u, nm), self.dtype, nm)
            t = None
            for rep in range(reps):
                pm = (C.c_float * L.RESNET_PROF_SLOTS)()
                L.check(self.lib.hmmr_resnet50_fwd(C.byref(self.rw), src, n, n_zero, phi.data_ptr(), ws.data_ptr(),
                                                   nbytes, self._stream(), pm), "hmmr_resnet50_fwd")
                cur = np.frombuffer(pm, dtype=np.float32).copy()
                t = cur if (t is None or rep == 0) else np.minimum(t, cur)      # rep 0 is the warm-up
            for slot, u, nm in layers:
                tile = self._layer_of(u, nm).tile
                if (u, nm) not in best or t[slot] < best[(u, nm)][0] * 0.98:    # 2 % hysteresis towards the heuristic
                    best[(u, nm)] = (float(t[slot]), tile)
        table = {k: v[1] for k, v in best.items()}
        self.tune_log.append((nt, round((time.perf_counter() - t_start) * 1e3, 1)))     # (frames, ms): never silent
        if self._tile_cache:
            import json
            old = json.load(open(self._tile_cache)) if os.path.exists(self._tile_cache) else {}
            old["%d:%d" % (self.dtype, nt)] = {"%d:%s" % k: v for k, v in table.items()}
            json.dump(old, open(self._tile_cache, "w"))
        return table

    # -- stages ----------------------------------------------------------------
    def _resnet_pass(self, images, n, n_zero, phi, ws_key, prof=False, tune=True):
        """One hmmr_resnet50_fwd launch sequence on the current stream: images[:n] (+ n_zero zero
        images) -> phi[:n + n_zero]."""
        nt = n + n_zero
        table = None
        if tune and nt >= self._TUNE_MIN_FRAMES:
            if self._needs_tuning(nt):
                self._tiles[nt] = self._tune_resnet(images, n, n_zero, ws_key)
            if self._tiles:                                  # an untuned size borrows the nearest tuned one
                table = self._tiles[min(self._tiles, key=lambda k: abs(k - nt))]
        if table is not None:
            self._set_tiles(table)
        elif self._tiles:
            self._set_tiles({k: 0 for k in next(iter(self._tiles.values()))})
        nbytes = self.lib.hmmr_resnet50_workspace_bytes(nt, self.dtype)
        ws = self._ws.setdefault(ws_key, _Workspace(self.device)).get(nbytes)
        pm = (C.c_float * L.RESNET_PROF_SLOTS)() if prof else None
        L.check(self.lib.hmmr_resnet50_fwd(C.byref(self.rw), images.data_ptr() if n else None, n, n_zero,
                                           phi.data_ptr(), ws.data_ptr(), nbytes, self._stream(), pm),
                "hmmr_resnet50_fwd")
        return np.frombuffer(pm, dtype=np.float32).astype(np.float64) if prof else None

    def resnet_cuts(self, n, parts=None):
        """Frame boundaries of the contiguous parts `resnet()` runs n frames as ([0, n]: one part).  A caller that
        feeds the parts itself (evaluation/streaming.py: part i starts when ITS upload has landed) runs part i as
        `resnet(images[a:b], parts=1, ws_key="resnet%d" % i)` on `side_stream(i)`."""
        parts = self.resnet_streams if parts is None else int(parts)
        if parts < 2 or n < self._SPLIT_MIN_FRAMES or self.resnet_chunk > 0:
            return [0, n]
        return [(i * n) // parts for i in range(parts + 1)]

    def resnet(self, images, prof=False, n_zero=0, out=None, ws_key="resnet", parts=None):
        """images [n,224,224,3] fp32 (device) -> phi [n + n_zero,2048] fp32; the last
        n_zero rows are the features of all-zero images (the padding frames of
        predict_all_images), encoded in the same pass.  encoder_resnet, src/models.py:50-77.

        Large batches run as `resnet_streams` (default 2) contiguous parts on concurrent HIP streams:
        every layer launch ends in a partial round of workgroups (tile-count quantisation, worst in
        blocks 3-4 where a 256-frame batch is only 1.5-3 rounds), and a second, independent launch
        sequence fills those tails.  Per-frame independent => bit-identical; measured -4 %.
        parts: overrides `resnet_streams` for this call (1 = one launch sequence on the current stream, whose
        workspace is `ws_key`: a caller that keeps several passes in flight gives each its own)."""
        self._need(self.rw, "resnet_v2_50/*")
        images = self.to_device(images)
        n = images.shape[0]
        assert n == 0 or tuple(images.shape[1:]) == (224, 224, 3), images.shape
        nt = n + n_zero
        if out is not None:                                  # caller-owned feature rows (a slice of a longer video's phi)
            assert out.is_cuda and out.dtype == torch.float32 and out.shape == (nt, 2048) and out.is_contiguous()
        phi = out if out is not None else torch.empty((nt, 2048), dtype=torch.float32, device=self.device)
        if self.resnet_chunk > 0:                            # dev switch: sequential chunks, heuristic tiles
            chunk, i = self.resnet_chunk, 0
            prof_tot = np.zeros(L.RESNET_PROF_SLOTS, np.float64) if prof else None
            while i < nt:
                c_real = max(0, min(chunk, n - i))
                c_zero = min(chunk - c_real, nt - i - c_real) if i + c_real >= n else 0
                pm = self._resnet_pass(images[i:i + c_real], c_real, c_zero, phi[i:i + c_real + c_zero], "resnet",
                                       prof, tune=False)
                if prof:
                    prof_tot += pm
                i += c_real + c_zero
            return (phi, prof_tot) if prof else phi
        parts = self.resnet_streams if parts is None else int(parts)
        if prof or parts < 2 or n < self._SPLIT_MIN_FRAMES or torch.cuda.is_current_stream_capturing():
            pm = self._resnet_pass(images, n, n_zero, phi, ws_key, prof)
            return (phi, pm) if prof else phi
        cur = torch.cuda.current_stream(self.device)
        self.side_stream(parts - 1)
        cuts = [(i * n) // parts for i in range(parts + 1)]
        if self.autotune:                                    # tune every part size before anything overlaps
            for i in range(parts):
                ni, nz = cuts[i + 1] - cuts[i], (n_zero if i == parts - 1 else 0)
                if self._needs_tuning(ni + nz):
                    self._tiles[ni + nz] = self._tune_resnet(images[cuts[i]:cuts[i + 1]], ni, nz, "resnet%d" % i)
        for i in range(parts):
            side = self._side_streams[i]
            side.wait_stream(cur)                            # the frames are ready on the caller's stream
            with torch.cuda.stream(side):
                nz = n_zero if i == parts - 1 else 0
                self._resnet_pass(images[cuts[i]:cuts[i + 1]], cuts[i + 1] - cuts[i], nz,
                                  phi[cuts[i]:cuts[i + 1] + nz], "resnet%d" % i)
        for i in range(parts):
            cur.wait_stream(self._side_streams[i])
        return phi

    def side_stream(self, i):
        """The i-th of the high-priority streams the ResNet parts run on (created on first use)."""
        while len(self._side_streams) <= i:
            self._side_streams.append(torch.cuda.Stream(device=self.device, priority=int(devflags.get("RESNET_PRIORITY"))))
        return self._side_streams[i]

    def temporal(self, phi):
        """phi [b,t,2048] fp32 -> movie strips [b,t,2048].  az_fc2_groupnorm, src/models.py:121-141."""
        self._need(self.tw, "AZ_FC_block*")
        phi = self.to_device(phi)
        b, t, c = phi.shape
        assert c == 2048
        out = torch.empty_like(phi)
        nbytes = self.lib.hmmr_temporal_workspace_bytes(b, t, self.temporal_dtype)
        ws = self._ws["temporal"].get(nbytes)
        L.check(self.lib.hmmr_temporal_fwd(C.byref(self.tw), phi.data_ptr(), b, t, out.data_ptr(),
                                           ws.data_ptr(), nbytes, self._stream()), "hmmr_temporal_fwd")
        return out

    def hallucinate(self, phi):
        """phi [..., 2048] -> hallucinated movie strips, same shape.  fc2_res, src/models.py:270-296."""
        if self.hw is None:
            raise L.HmmrError("the loaded weights have no fc2_res/* variables (pred_mode 'hal')")
        phi = self.to_device(phi)
        flat = phi.reshape(-1, 2048)
        m = flat.shape[0]
        out = torch.empty_like(flat)
        nbytes = self.lib.hmmr_hallucinator_workspace_bytes(m, self.temporal_dtype)
        ws = self._ws["hal"].get(nbytes)
        L.check(self.lib.hmmr_hallucinator_fwd(C.byref(self.hw), flat.data_ptr(), m, out.data_ptr(),
                                               ws.data_ptr(), nbytes, self._stream()), "hmmr_hallucinator_fwd")
        return out.reshape(phi.shape)

    def ief(self, strips, omega_start=None, use_delta_from_pred=True):
        """strips [m,2048] -> omegas [R,m,85]; R = 1 + len(delta_t_values), deltas in sorted order.
        batch_pred_omega / call_hmr_ief, src/models.py:233-267, 299-377.  omega_start: [m,85] starting point of the IEF
        (`omega_mean`; None = the checkpoint's mean theta in every row, tester.py:181).  use_optcam is a property of the
        packed delta regressors (72- or 75-wide, `self.use_optcam`)."""
        self._need(self.iw, "single_view_ief*/3D_module/* and mean_param")
        strips = self.to_device(strips)
        m = strips.shape[0]
        R = self.iw.num_regressors
        out = torch.empty((R, m, 85), dtype=torch.float32, device=self.device)
        nbytes = self.lib.hmmr_ief_workspace_bytes(m, R, self.ief_dtype)
        ws = self._ws["ief"].get(nbytes)
        start = None
        if omega_start is not None:
            start = self.to_device(omega_start).reshape(-1, 85)
            if start.shape[0] == 1:
                start = start.expand(m, 85).contiguous()
            assert start.shape == (m, 85), start.shape
        iw = L.IefWeights.from_buffer_copy(self.iw)          # a per-call copy of the (host) struct: the shared one is never toggled
        iw.delta_from_start = int(not use_delta_from_pred)
        L.check(self.lib.hmmr_ief_fwd_from(C.byref(iw), strips.data_ptr(), L.ptr(start), m, out.data_ptr(),
                                           ws.data_ptr(), nbytes, self._stream()), "hmmr_ief_fwd_from")
        return out

    @property
    def use_optcam(self):
        """True when the packed delta regressors predict 72 values and get the fixed camera [1, 0, 0] (models.py:333-371)."""
        return self.iw is None or not self.iw.no_optcam

    def smpl(self, theta, beta, cams=None, want_rs=True):
        """theta [m,72], beta [m,10], cams [m,3] or None (row-major views with a
        unit inner stride are used in place).  Returns verts [m,V,3], joints
        [m,K,3], kps [m,K,2] or None, Rs [m,24,3,3] or None.
        SMPL.__call__, src/tf_smpl/batch_smpl.py:89-162."""
        def prep(x, width):
            if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32
                    and x.dim() == 2 and x.stride(1) == 1 and x.shape[1] == width):
                x = self.to_device(x).reshape(-1, width)
            return x
        theta, beta = prep(theta, 72), prep(beta, 10)
        m = theta.shape[0]
        cams = prep(cams, 3) if cams is not None else None
        V, K = self.num_verts, self.num_kps
        verts = torch.empty((m, V, 3), dtype=torch.float32, device=self.device)
        joints = torch.empty((m, K, 3), dtype=torch.float32, device=self.device)
        kps = torch.empty((m, K, 2), dtype=torch.float32, device=self.device) if cams is not None else None
        rs = torch.empty((m, 24, 3, 3), dtype=torch.float32, device=self.device) if want_rs else None
        nbytes = self.lib.hmmr_smpl_workspace_bytes(m)
        ws = self._ws["smpl"].get(nbytes)
        L.check(self.lib.hmmr_smpl_fwd(C.byref(self.sc), theta.data_ptr(), theta.stride(0),
                                       beta.data_ptr(), beta.stride(0),
                                       L.ptr(cams), cams.stride(0) if cams is not None else 0, m,
                                       verts.data_ptr(), joints.data_ptr(), L.ptr(kps), L.ptr(rs),
                                       ws.data_ptr(), nbytes, self._stream()), "hmmr_smpl_fwd")
        return verts, joints, kps, rs

    def smpl_into(self, theta, beta, cams, rec, off_verts, off_joints, off_kps, off_rs):
        """SMPL forward that writes instance i's verts/joints/kps/Rs straight into row i of
        the packed record tensor `rec` [m, rec_len] at the given float offsets
        (hmmr_smpl_fwd_strided).  theta/beta/cams: fp32 device views with unit inner stride."""
        m = theta.shape[0]
        assert rec.dtype == torch.float32 and rec.stride(1) == 1 and rec.shape[0] >= m
        for x, wdt in ((theta, 72), (beta, 10), (cams, 3)):
            assert x.is_cuda and x.dtype == torch.float32 and x.stride(1) == 1 and x.shape == (m, wdt)
        nbytes = self.lib.hmmr_smpl_workspace_bytes(m)
        ws = self._ws["smpl"].get(nbytes)
        base = rec.data_ptr()
        L.check(self.lib.hmmr_smpl_fwd_strided(
            C.byref(self.sc), theta.data_ptr(), theta.stride(0), beta.data_ptr(), beta.stride(0),
            cams.data_ptr(), cams.stride(0), m, base + 4 * off_verts, base + 4 * off_joints,
            base + 4 * off_kps, base + 4 * off_rs, rec.stride(0), ws.data_ptr(), nbytes, self._stream()),
            "hmmr_smpl_fwd_strided")

    def smpl_records(self, om, rec, field_offsets):
        """All containers in one launch set (hmmr_smpl_fwd_records): om [R,n,85] contiguous fp32 (present first), rec
        [>= n, rec_len] fp32 with unit inner stride, field_offsets [R][7] = float offsets of (cams, joints, kps, poses,
        shapes, verts, omegas) of each container inside a record."""
        R, n = om.shape[0], om.shape[1]
        assert om.is_cuda and om.dtype == torch.float32 and om.is_contiguous() and om.shape[2] == 85
        assert rec.dtype == torch.float32 and rec.stride(1) == 1 and rec.shape[0] >= n
        offs = (C.c_int32 * (R * 7))(*[int(o) for row in field_offsets for o in row])
        nbytes = self.lib.hmmr_smpl_workspace_bytes(R * n)
        ws = self._ws["smpl"].get(nbytes)
        L.check(self.lib.hmmr_smpl_fwd_records(C.byref(self.sc), om.data_ptr(), R, n, rec.data_ptr(), rec.stride(0), offs,
                                               ws.data_ptr(), nbytes, self._stream()), "hmmr_smpl_fwd_records")

    def groupnorm_relu(self, x, gamma, beta, groups=32, out_dtype=L.HMMR_F32):
        x = self.to_device(x)
        b, t, c = x.shape
        g, be = self.to_device(gamma), self.to_device(beta)
        out = packing.empty_act((b, t, c), out_dtype, self.device)
        L.check(self.lib.hmmr_groupnorm_relu(x.data_ptr(), g.data_ptr(), be.data_ptr(), b, t, c, groups,
                                             out.data_ptr(), out_dtype, self._stream()), "hmmr_groupnorm_relu")
        return out


class _FlagScope(object):
    """See HmmrEngine.flag_scope.  The flag word is per DEVICE: two scopes on one device that overlapped would read (and clear) each
    other's flags -- scope A's entry would move what B's in-flight call raised to 'carried', and B's exit would then read 0 and return
    clamped results without falling back.  So scopes of one device are serialised: a per-device re-entrant lock is held from entry
    to exit (two Testers on one device in two threads run their guarded calls one after the other; different devices do not wait for
    each other)."""
    _locks = {}
    _locks_guard = threading.Lock()

    def __init__(self, engine):
        self.engine, self.flags = engine, 0
        key = (engine.device.type, engine.device.index)
        with _FlagScope._locks_guard:
            self._lock = _FlagScope._locks.setdefault(key, threading.RLock())

    def __enter__(self):
        self._lock.acquire()
        try:
            e = self.engine
            stale = e._device_flags(True)
            if stale:
                key = (e.device.type, e.device.index)
                HmmrEngine._carried_flags[key] = HmmrEngine._carried_flags.get(key, 0) | stale
        except BaseException:
            self._lock.release()
            raise
        return self

    def __exit__(self, *exc):
        try:
            self.flags = self.engine._device_flags(True)
        finally:
            self._lock.release()
        return False


def conv_gemm(x, w_hwio, stride=1, pad=0, scale=None, shift=None, res=None, relu=False,
              scale2=None, shift2=None, in_dtype=L.HMMR_F32, out_dtype=L.HMMR_F32, tile=0,
              device="cuda:0", res_stride=1, split_k=0, pro=None, raw=False, second=None, k_order=0, n_split=0, relu_b=False):
    """Test/utility entry: run one NHWC convolution through hmmr_conv_gemm.
    x [n,h,w,cin] (numpy/torch), w_hwio [kh,kw,cin,cout].  Returns (out, out2) as float32 arrays, or with
    raw=True the device tensors in their storage type; x may itself be such a device tensor.  n_split > 0: the column split
    (hmmr_conv_desc_t.out_b) -- returns (out [.., n_split], out_b [.., cout - n_split]) with ReLU flags relu / relu_b."""
    lib = L.load()
    dev = torch.device(device)
    store = packing.DeviceStore(dev)
    if isinstance(x, torch.Tensor) and x.is_cuda:
        xt = x.contiguous()
    else:
        xt = store.put(np.asarray(x, np.float32), packing.TORCH_DT[in_dtype])
    n, h, w_, cin = xt.shape
    kh, kw, _, cout = w_hwio.shape
    py, px = (pad, pad) if isinstance(pad, int) else pad
    ho = (h + 2 * py - kh) // stride + 1
    wo = (w_ + 2 * px - kw) // stride + 1
    x2 = None
    if second is not None:          # (x2 [n,h,w,cin2], w2 [1,1,cin2,cout]): a second 1x1 source appended along K (hmmr_conv_desc_t.in2)
        x2 = store.put(np.asarray(second[0], np.float32), packing.TORCH_DT[in_dtype])
        w_hwio = np.concatenate([np.asarray(w_hwio, np.float32), np.asarray(second[1], np.float32)], axis=2)
    wp = packing.pack_conv_weight(np.asarray(w_hwio, np.float32), k_order if k_order != 2 else 0, chunk=64 if in_dtype == L.HMMR_BF16 else 32)
    if k_order == 2 and in_dtype == L.HMMR_BF16:        # the filter stream of csrc/conv3x3_stream.hip, bf16 form
        scale = np.ones(cout, np.float32) if scale is None else scale
        shift = np.zeros(cout, np.float32) if shift is None else shift
        wt = store.put_tensor(packing.pack_conv3x3_stream(np.asarray(w_hwio, np.float32), bf16=True))
    elif k_order == 2:                                  # ... split form (as packing._layer_stream3x3)
        k = packing.row_pow2(wp[:cout])
        sc = np.ones(cout, np.float64) if scale is None else np.asarray(scale, np.float64)
        scale = (sc * np.exp2(-k.astype(np.float64))).astype(np.float32)
        shift = np.zeros(cout, np.float32) if shift is None else shift
        wt = store.put_tensor(packing.pack_conv3x3_stream(np.asarray(w_hwio, np.float32), k))      # (a 1x1 filter: pack_conv1x1_stream, the same layout)
    elif packing.TORCH_DT[in_dtype] is packing.SPLIT:   # as packing._layer: rows scaled by a power of two, undone by `scale`
        k = packing.row_pow2(wp)
        wp = packing.scale_rows(wp, k)
        sc = np.ones(wp.shape[0], np.float64)
        if scale is not None:
            sc[:len(scale)] = np.asarray(scale, np.float64)
        scale = (sc * np.exp2(-k.astype(np.float64))).astype(np.float32)
    if k_order != 2:
        wt = store.put(wp, packing.TORCH_DT[in_dtype])
    ldo = (cout + 7) // 8 * 8
    out = packing.empty_act((n, ho, wo, ldo), out_dtype, dev, zero=True)
    d = L.ConvDesc()
    d.in_, d.w, d.out = xt.data_ptr(), wt.data_ptr(), out.data_ptr()
    if x2 is not None:
        d.in2, d.cin2 = x2.data_ptr(), x2.shape[-1]
    d.scale = store.vec(scale).data_ptr() if scale is not None else None
    d.shift = store.vec(shift).data_ptr() if shift is not None else None
    out2 = None
    if scale2 is not None:
        out2 = torch.zeros_like(out)
        d.out2, d.scale2, d.shift2 = out2.data_ptr(), store.vec(scale2).data_ptr(), store.vec(shift2).data_ptr()
    if res is not None:
        rt = store.put(np.asarray(res, np.float32), packing.TORCH_DT[out_dtype])
        d.res = rt.data_ptr()
        if res_stride == 1:
            assert tuple(rt.shape) == (n, ho, wo, cout) and cout % 8 == 0
            d.ldr = cout
        else:
            d.res_strided = 1
            d.res_img_stride = rt.shape[1] * rt.shape[2] * cout
            d.res_row_stride = res_stride * rt.shape[2] * cout
            d.res_px_stride = res_stride * cout
    d.in_dtype, d.out_dtype = in_dtype, out_dtype
    d.n_img, d.hin, d.win, d.cin = n, h, w_, cin
    d.in_img_stride, d.in_row_stride, d.in_px_stride = h * w_ * cin, w_ * cin, cin
    d.kh, d.kw, d.sy, d.sx, d.py, d.px = kh, kw, stride, stride, py, px
    d.ho, d.wo, d.cout, d.ldo = ho, wo, cout, ldo
    d.relu, d.tile, d.k_order = int(relu), tile, k_order
    out_b = None
    if n_split:
        out_b = packing.empty_act((n, ho, wo, cout - n_split), out_dtype, dev, zero=True)
        d.out_b, d.ldo_b, d.n_split, d.relu_b = out_b.data_ptr(), cout - n_split, n_split, int(relu_b)
    if pro is not None:
        d.pro_scale, d.pro_shift = store.put(pro[0]).data_ptr(), store.put(pro[1]).data_ptr()
    if split_k > 1:
        nb = lib.hmmr_conv_splitk_workspace_bytes(n * ho * wo, cout, split_k)
        skws = torch.empty(int(nb), dtype=torch.uint8, device=dev)
        d.split_k, d.ws, d.ws_bytes = split_k, skws.data_ptr(), nb
    L.check(lib.hmmr_conv_gemm(C.byref(d), torch.cuda.current_stream(dev).cuda_stream), "hmmr_conv_gemm")
    torch.cuda.synchronize(dev)
    if raw:
        return out, (out_b if n_split else out2)
    if n_split:
        return (packing.act_to_f32(out, out_dtype)[..., :n_split].cpu().numpy(), packing.act_to_f32(out_b, out_dtype).cpu().numpy())
    o = packing.act_to_f32(out, out_dtype)[..., :cout].cpu().numpy()
    o2 = packing.act_to_f32(out2, out_dtype)[..., :cout].cpu().numpy() if out2 is not None else None
    return o, o2


def bottleneck_tail(h2, w3_hwio, bias3, res, pre, w1_hwio, bn1, res_stride=1, device="cuda:0", conv2=None, shortcut=None):
    """Test/utility entry for hmmr_bottleneck_tail (bf16): h2 [n,h,w,64], w3 [1,1,64,256], res
    [n,h*s,w*s,256] (s = res_stride), pre = (scale, shift) [256], w1 [1,1,256,64], bn1 = (scale, shift) [64].
    With conv2 = (w2 [3,3,64,64], scale2, shift2) the first argument is h1 and the 3x3 conv runs inside the launch.
    Returns (trunk [n,h,w,256], h1 [n,h,w,64]) as float32 arrays of the bf16 results."""
    lib = L.load()
    dev = torch.device(device)
    store = packing.DeviceStore(dev)
    bf = torch.bfloat16
    x = store.put(np.asarray(h2, np.float32), bf)
    n, h, w_, cm = x.shape
    depth, n2 = w3_hwio.shape[3], w1_hwio.shape[3]
    w3 = store.put(packing.pack_conv_weight(np.asarray(w3_hwio, np.float32)), bf)
    w1 = store.put(packing.pack_conv_weight(np.asarray(w1_hwio, np.float32)), bf)
    rt = store.put(np.asarray(res, np.float32), bf) if res is not None else None
    out = torch.zeros((n, h, w_, depth), dtype=bf, device=dev)
    h1 = torch.zeros((n, h, w_, n2), dtype=bf, device=dev)
    d = L.TailDesc()
    d.dtype, d.m, d.c_mid, d.depth = L.HMMR_BF16, n * h * w_, cm, depth
    d.ho, d.wo = h, w_
    if conv2 is None:
        d.h2 = x.data_ptr()
    else:                                  # h2 argument is h1; conv2 = (w2_hwio [3,3,64,64], scale2, shift2)
        w2 = store.put(packing.pack_conv_weight(np.asarray(conv2[0], np.float32)), bf)
        d.h1, d.hin, d.win, d.w2 = x.data_ptr(), h, w_, w2.data_ptr()
        d.scale2, d.shift2 = store.vec(conv2[1]).data_ptr(), store.vec(conv2[2]).data_ptr()
    d.w3, d.shift3 = w3.data_ptr(), store.vec(bias3).data_ptr()
    if shortcut is not None:               # (xp [n,h,w,64], wsc [1,1,64,256], bias): the shortcut conv runs in the launch
        xp = store.put(np.asarray(shortcut[0], np.float32), bf)
        wsc = store.put(packing.pack_conv_weight(np.asarray(shortcut[1], np.float32)), bf)
        d.xp, d.wsc, d.shift_sc = xp.data_ptr(), wsc.data_ptr(), store.vec(shortcut[2]).data_ptr()
    else:
        d.res = rt.data_ptr()
    if shortcut is not None:
        pass
    elif res_stride == 1:
        d.ldr = depth
    else:
        d.res_strided, d.ho, d.wo = 1, h, w_
        d.res_img_stride = rt.shape[1] * rt.shape[2] * depth
        d.res_row_stride, d.res_px_stride = res_stride * rt.shape[2] * depth, res_stride * depth
    d.out, d.out_h1 = out.data_ptr(), h1.data_ptr()
    d.pre_scale, d.pre_shift = store.vec(pre[0]).data_ptr(), store.vec(pre[1]).data_ptr()
    d.w1, d.scale1, d.shift1, d.relu1, d.n2 = w1.data_ptr(), store.vec(bn1[0]).data_ptr(), store.vec(bn1[1]).data_ptr(), 1, n2
    L.check(lib.hmmr_bottleneck_tail(C.byref(d), torch.cuda.current_stream(dev).cuda_stream), "hmmr_bottleneck_tail")
    torch.cuda.synchronize(dev)
    return out.float().cpu().numpy(), h1.float().cpu().numpy()


def bottleneck_tail_single(h1, conv2, stride, w3_hwio, bias3, res, pre, want_raw=True, want_pre=True, device="cuda:0"):
    """Test/utility entry for the single-phase hmmr_bottleneck_tail (a block's stride-2 last unit):
    h1 [n,h,w,cm], conv2 = (w2 [3,3,cm,cm], scale2, shift2) with `stride`, w3 [1,1,cm,depth], res [n,h,w,depth]
    (sub-sampled by `stride` as the identity shortcut), pre = (scale, shift).  Returns (trunk, preact) float32."""
    lib = L.load()
    dev = torch.device(device)
    store = packing.DeviceStore(dev)
    bf = torch.bfloat16
    x = store.put(np.asarray(h1, np.float32), bf)
    n, h, w_, cm = x.shape
    ho, wo = (h + stride - 1) // stride, (w_ + stride - 1) // stride
    depth = w3_hwio.shape[3]
    w2 = store.put(packing.pack_conv_weight(np.asarray(conv2[0], np.float32)), bf)
    w3 = store.put(packing.pack_conv_weight(np.asarray(w3_hwio, np.float32)), bf)
    rt = store.put(np.asarray(res, np.float32), bf)
    out = torch.zeros((n, ho, wo, depth), dtype=bf, device=dev)
    outp = torch.zeros((n, ho, wo, depth), dtype=bf, device=dev)
    d = L.TailDesc()
    d.dtype, d.m, d.c_mid, d.depth = L.HMMR_BF16, n * ho * wo, cm, depth
    d.h1, d.hin, d.win, d.ho, d.wo, d.conv2_stride = x.data_ptr(), h, w_, ho, wo, stride
    d.w2, d.scale2, d.shift2 = w2.data_ptr(), store.vec(conv2[1]).data_ptr(), store.vec(conv2[2]).data_ptr()
    d.w3, d.shift3 = w3.data_ptr(), store.vec(bias3).data_ptr()
    d.res = rt.data_ptr()
    if stride == 1:
        d.ldr = depth
    else:
        d.res_strided = 1
        d.res_img_stride = rt.shape[1] * rt.shape[2] * depth
        d.res_row_stride, d.res_px_stride = stride * rt.shape[2] * depth, stride * depth
    if want_raw:
        d.out = out.data_ptr()
    if want_pre:
        d.out_pre, d.pre_scale, d.pre_shift = outp.data_ptr(), store.vec(pre[0]).data_ptr(), store.vec(pre[1]).data_ptr()
    L.check(lib.hmmr_bottleneck_tail(C.byref(d), torch.cuda.current_stream(dev).cuda_stream), "hmmr_bottleneck_tail")
    torch.cuda.synchronize(dev)
    return (out.float().cpu().numpy() if want_raw else None), (outp.float().cpu().numpy() if want_pre else None)


def b1_unit(h1, conv2, w3_hwio, bias3, pre, w1_hwio, bn1, res=None, shortcut=None, device="cuda:0"):
    """Test/utility entry for the whole-unit kernel of block 1 (hmmr_tail_desc_t.unit_stream, csrc/b1_unit.hip; f16x3):
    h1 [n,h,w,64] (a split device tensor or a float array), conv2 = (w2 [3,3,64,64], scale2, shift2), w3 [1,1,64,256], bias3 [256],
    pre = (scale, shift) [256], w1 [1,1,256,64], bn1 = (scale, shift) [64]; either res [n,h,w,256] (a float array: the identity
    shortcut) or shortcut = (xp [n,h,w,64], wsc [1,1,64,256], bias [256]) folded into conv3.  Returns the split device tensors
    (trunk [n,h,w,256], h1' [n,h,w,64])."""
    lib = L.load()
    dev = torch.device(device)
    store = packing.DeviceStore(dev)
    X3 = L.HMMR_F16X3
    x = h1.contiguous() if isinstance(h1, torch.Tensor) and h1.is_cuda else store.put(np.asarray(h1, np.float32), packing.SPLIT)
    n, h, w_, cm = x.shape
    w2 = np.asarray(conv2[0], np.float32)
    k2 = packing.row_pow2(packing.pack_conv_weight(w2)[:64])
    w3 = np.asarray(w3_hwio, np.float32)[0, 0]                                   # [K][256]
    b3 = np.asarray(bias3, np.float64)
    d = L.TailDesc()
    if shortcut is not None:
        xp = store.put(np.asarray(shortcut[0], np.float32), packing.SPLIT)
        w3 = np.concatenate([w3, np.asarray(shortcut[1], np.float32)[0, 0]], axis=0)
        b3 = b3 + np.asarray(shortcut[2], np.float64)
        d.xp, d.c_xp = xp.data_ptr(), 64
    else:
        rt = store.put(np.asarray(res, np.float32), packing.SPLIT)
        d.res, d.ldr = rt.data_ptr(), 256
    w1 = np.asarray(w1_hwio, np.float32)[0, 0]                                   # [256][64]
    stream = store.put_tensor(packing.pack_b1_unit_stream(w2, k2, w3.T, w1.T))
    assert stream.numel() * 2 == lib.hmmr_b1_unit_stream_bytes(0 if shortcut is None else 64)
    k3, k1 = packing.row_pow2(w3.T), packing.row_pow2(w1.T)
    out = packing.empty_act((n, h, w_, 256), X3, dev, zero=True)
    h1n = packing.empty_act((n, h, w_, 64), X3, dev, zero=True)
    d.dtype, d.m, d.c_mid, d.depth, d.n2 = X3, n * h * w_, cm, 256, 64
    d.h1, d.hin, d.win, d.ho, d.wo = x.data_ptr(), h, w_, h, w_
    d.unit_stream = stream.data_ptr()
    d.scale2 = store.vec((np.asarray(conv2[1], np.float64) * np.exp2(-k2.astype(np.float64))).astype(np.float32)).data_ptr()
    d.shift2 = store.vec(conv2[2]).data_ptr()
    d.scale3 = store.vec(np.exp2(-k3.astype(np.float64)).astype(np.float32)).data_ptr()
    d.shift3 = store.vec(b3.astype(np.float32)).data_ptr()
    d.pre_scale, d.pre_shift = store.vec(pre[0]).data_ptr(), store.vec(pre[1]).data_ptr()
    d.scale1 = store.vec((np.asarray(bn1[0], np.float64) * np.exp2(-k1.astype(np.float64))).astype(np.float32)).data_ptr()
    d.shift1, d.relu1 = store.vec(bn1[1]).data_ptr(), 1
    d.out, d.out_h1 = out.data_ptr(), h1n.data_ptr()
    L.check(lib.hmmr_bottleneck_tail(C.byref(d), torch.cuda.current_stream(dev).cuda_stream), "hmmr_bottleneck_tail")
    torch.cuda.synchronize(dev)
    return out, h1n
