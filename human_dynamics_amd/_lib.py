"""ctypes binding of libhmmr_hip.so (include/hmmr_hip.h).

There is no CPU fallback: if the shared library has not been built
(`python -m human_dynamics_amd.build`) `load()` raises, and every compute
entry point of the package goes through `load()`.
"""
from __future__ import annotations

import ctypes as C
import os

from . import devflags

HERE = os.path.dirname(os.path.abspath(__file__))
# devflags LIB_PATH (HMMR_LIB_PATH) lets a development run A/B two builds of the library; the default is the in-tree build
LIB_PATH = devflags.get("LIB_PATH") or os.path.join(HERE, "libhmmr_hip.so")

HMMR_F32, HMMR_BF16, HMMR_F16X3 = 0, 1, 2
FLAG_SATURATED = 1
FLAG_NAN = 2          # with FLAG_SATURATED: the clamped value was a NaN (include/hmmr_hip.h)
ABI_VERSION = 19
RESNET_UNITS = 16
RESNET_PROF_SLOTS = 64
MAX_TEMPORAL_BLOCKS = 8
MAX_REGRESSORS = 8

_vp, _fp, _ip = C.c_void_p, C.c_void_p, C.c_void_p   # all device pointers travel as void*


class ConvDesc(C.Structure):
    _fields_ = [
        ("in_", _vp), ("w", _vp), ("scale", _fp), ("shift", _fp), ("res", _vp), ("out", _vp),
        ("out2", _vp), ("scale2", _fp), ("shift2", _fp),
        ("in_dtype", C.c_int), ("out_dtype", C.c_int),
        ("n_img", C.c_int), ("hin", C.c_int), ("win", C.c_int), ("cin", C.c_int),
        ("in_img_stride", C.c_int64), ("in_row_stride", C.c_int), ("in_px_stride", C.c_int),
        ("kh", C.c_int), ("kw", C.c_int), ("sy", C.c_int), ("sx", C.c_int), ("py", C.c_int), ("px", C.c_int),
        ("ho", C.c_int), ("wo", C.c_int), ("cout", C.c_int), ("ldo", C.c_int),
        ("ldr", C.c_int), ("res_strided", C.c_int), ("res_img_stride", C.c_int64),
        ("res_row_stride", C.c_int), ("res_px_stride", C.c_int),
        ("relu", C.c_int), ("tile", C.c_int),
        ("split_k", C.c_int), ("ws", _vp), ("ws_bytes", C.c_size_t),
        ("pro_scale", _fp), ("pro_shift", _fp),
        ("out_b", _vp), ("ldo_b", C.c_int), ("n_split", C.c_int), ("relu_b", C.c_int),
        ("in2", _vp), ("cin2", C.c_int), ("k_order", C.c_int),
        ("batch", C.c_int), ("batch_in_bytes", C.c_int64), ("batch_w_bytes", C.c_int64), ("batch_out_bytes", C.c_int64),
        ("batch_res_bytes", C.c_int64), ("batch_scale_bytes", C.c_int64), ("batch_shift_bytes", C.c_int64),
    ]


class TailDesc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int), ("h2", _vp), ("m", C.c_int), ("c_mid", C.c_int), ("depth", C.c_int),
        ("w3", _vp), ("scale3", _fp), ("shift3", _fp),
        ("res", _vp), ("ldr", C.c_int), ("res_strided", C.c_int), ("res_img_stride", C.c_int64),
        ("res_row_stride", C.c_int), ("res_px_stride", C.c_int), ("ho", C.c_int), ("wo", C.c_int),
        ("out", _vp), ("pre_scale", _fp), ("pre_shift", _fp),
        ("w1", _vp), ("scale1", _fp), ("shift1", _fp), ("relu1", C.c_int), ("n2", C.c_int),
        ("out_h1", _vp),
        ("h1", _vp), ("hin", C.c_int), ("win", C.c_int), ("w2", _vp), ("scale2", _fp), ("shift2", _fp),
        ("xp", _vp), ("wsc", _vp), ("shift_sc", _fp),
        ("conv2_stride", C.c_int), ("out_pre", _vp),
        ("pair_stream", _vp), ("c_xp", C.c_int), ("unit_stream", _vp),
    ]


class Debug(C.Structure):
    """hmmr_debug_t: development switches, all zero = product defaults."""
    _fields_ = [("stem_route", C.c_int), ("stem_no_conv1", C.c_int), ("gemm_probe", C.c_int), ("smpl_blend_mfma", C.c_int), ("ief_no_group", C.c_int), ("reserved", C.c_int * 3),
                ("pair_min_pixels", C.c_int), ("pair_two_tile_min", C.c_int), ("pair_form", C.c_int)]


class LaunchCounts(C.Structure):
    """hmmr_launch_counts_t: how often each fused-unit kernel was launched since the last clear."""
    _fields_ = [("unit_pair", C.c_ulonglong), ("b1_unit", C.c_ulonglong), ("tail_split", C.c_ulonglong), ("conv3x3_stream", C.c_ulonglong),
                ("conv1x1_stream", C.c_ulonglong)]


class Layer(C.Structure):
    _fields_ = [("w", _vp), ("scale", _fp), ("shift", _fp), ("tile", C.c_int), ("k_order", C.c_int)]


class ResnetUnit(C.Structure):
    _fields_ = [("conv1", Layer), ("conv2", Layer), ("conv3", Layer), ("shortcut", Layer), ("c3sc", Layer), ("sc_c1", Layer),
                ("w3_frag", _vp), ("w1n_frag", _vp), ("pair_stream", _vp), ("conv1_frag", _vp), ("unit_stream", _vp), ("pre_scale", _fp), ("pre_shift", _fp),
                ("c_in", C.c_int), ("base", C.c_int), ("depth", C.c_int), ("stride", C.c_int),
                ("fuse_preact", C.c_int), ("fuse_tail", C.c_int)]


class ResnetWeights(C.Structure):
    _fields_ = [("dtype", C.c_int), ("stem", Layer),
                ("unit", ResnetUnit * RESNET_UNITS), ("post_scale", _fp), ("post_shift", _fp)]


class TemporalBlock(C.Structure):
    _fields_ = [("gn1_gamma", _fp), ("gn1_beta", _fp), ("conv1", Layer),
                ("gn2_gamma", _fp), ("gn2_beta", _fp), ("conv2", Layer)]


class TemporalWeights(C.Structure):
    _fields_ = [("dtype", C.c_int), ("num_blocks", C.c_int), ("block", TemporalBlock * MAX_TEMPORAL_BLOCKS)]


class HallucinatorWeights(C.Structure):
    _fields_ = [("dtype", C.c_int), ("fc1", Layer), ("fc2", Layer), ("fc3", Layer)]


class IefRegressor(C.Structure):
    _fields_ = [("nd", C.c_int), ("fc1_phi", Layer), ("fc1_theta", Layer), ("fc2", Layer), ("fc3", Layer)]


class IefWeights(C.Structure):
    _fields_ = [("dtype", C.c_int), ("num_regressors", C.c_int), ("num_stages", C.c_int),
                ("reg", IefRegressor * MAX_REGRESSORS), ("mean_theta", _fp),
                ("no_optcam", C.c_int), ("delta_from_start", C.c_int)]


class SmplConsts(C.Structure):
    _fields_ = [("num_verts", C.c_int), ("num_kps", C.c_int), ("lbs_nnz", C.c_int), ("vpad", C.c_int),
                ("dirs", _fp), ("j_template", _fp), ("j_shapedirs", _fp), ("parents", _ip),
                ("lbs_idx", _ip), ("lbs_w", _fp), ("kreg_ptr", _ip), ("kreg_idx", _ip), ("kreg_val", _fp),
                ("dirs_split", _vp)]


class Var(C.Structure):
    """hmmr_var_t: one checkpoint variable for the C-side packers (csrc/pack.cpp)"""
    _fields_ = [("name", C.c_char_p), ("data", _fp), ("numel", C.c_int64)]


class SmplSource(C.Structure):
    """hmmr_smpl_source_t: the body model in the src/tf_smpl layout"""
    _fields_ = [("num_verts", C.c_int), ("num_kps", C.c_int), ("v_template", _fp), ("shapedirs", _fp), ("posedirs", _fp),
                ("J_regressor", _fp), ("lbs_weights", _fp), ("kp_regressor", _fp), ("parents", _ip)]


# name -> (restype, argtypes); mirrors include/hmmr_hip.h one to one
SIGNATURES = {
    "hmmr_abi_version": (C.c_int, []),
    "hmmr_pack_resnet_bytes": (C.c_size_t, [C.POINTER(Var), C.c_int, C.c_int]),
    "hmmr_pack_resnet": (C.c_int, [C.POINTER(Var), C.c_int, C.c_int, _vp, C.c_size_t, _vp, C.POINTER(ResnetWeights)]),
    "hmmr_pack_temporal_bytes": (C.c_size_t, [C.POINTER(Var), C.c_int, C.c_int, C.c_int]),
    "hmmr_pack_temporal": (C.c_int, [C.POINTER(Var), C.c_int, C.c_int, C.c_int, _vp, C.c_size_t, _vp, C.POINTER(TemporalWeights)]),
    "hmmr_pack_hallucinator_bytes": (C.c_size_t, [C.POINTER(Var), C.c_int, C.c_int]),
    "hmmr_pack_hallucinator": (C.c_int, [C.POINTER(Var), C.c_int, C.c_int, _vp, C.c_size_t, _vp, C.POINTER(HallucinatorWeights)]),
    "hmmr_pack_ief_bytes": (C.c_size_t, [C.POINTER(Var), C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]),
    "hmmr_pack_ief": (C.c_int, [C.POINTER(Var), C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, _vp, C.c_size_t, _vp, C.POINTER(IefWeights)]),
    "hmmr_pack_smpl_bytes": (C.c_size_t, [C.POINTER(SmplSource), C.c_int, C.c_int]),
    "hmmr_pack_smpl": (C.c_int, [C.POINTER(SmplSource), C.c_int, C.c_int, _vp, C.c_size_t, _vp, C.POINTER(SmplConsts)]),
    "hmmr_last_error": (C.c_char_p, []),
    "hmmr_run_flags": (C.c_int, [C.POINTER(C.c_uint), C.c_int]),
    "hmmr_set_debug": (None, [C.POINTER(Debug)]),
    "hmmr_get_debug": (None, [C.POINTER(Debug)]),
    "hmmr_launch_counts": (None, [C.POINTER(LaunchCounts), C.c_int]),
    "hmmr_conv_gemm": (C.c_int, [C.POINTER(ConvDesc), _vp]),
    "hmmr_conv_splitk_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "hmmr_resnet50_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "hmmr_resnet50_fwd": (C.c_int, [C.POINTER(ResnetWeights), _fp, C.c_int, C.c_int, _fp, _vp, C.c_size_t, _vp,
                                    C.POINTER(C.c_float)]),
    "hmmr_temporal_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "hmmr_temporal_fwd": (C.c_int, [C.POINTER(TemporalWeights), _fp, C.c_int, C.c_int, _fp, _vp, C.c_size_t, _vp]),
    "hmmr_hallucinator_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "hmmr_hallucinator_fwd": (C.c_int, [C.POINTER(HallucinatorWeights), _fp, C.c_int, _fp, _vp, C.c_size_t, _vp]),
    "hmmr_groupnorm_relu": (C.c_int, [_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp]),
    "hmmr_ief_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "hmmr_ief_fwd": (C.c_int, [C.POINTER(IefWeights), _fp, C.c_int, _fp, _vp, C.c_size_t, _vp]),
    "hmmr_ief_fwd_from": (C.c_int, [C.POINTER(IefWeights), _fp, _fp, C.c_int, _fp, _vp, C.c_size_t, _vp]),
    "hmmr_smpl_workspace_bytes": (C.c_size_t, [C.c_int]),
    "hmmr_smpl_fwd": (C.c_int, [C.POINTER(SmplConsts), _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, C.c_int,
                                _fp, _fp, _fp, _fp, _vp, C.c_size_t, _vp]),
    "hmmr_crop_frames": (C.c_int, [_vp, _ip, C.c_int, C.c_int, C.c_int, _fp, _vp]),
    "hmmr_bottleneck_tail": (C.c_int, [C.POINTER(TailDesc), _vp]),
    "hmmr_pair_stream_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "hmmr_b1_unit_stream_bytes": (C.c_size_t, [C.c_int]),
    "hmmr_conv3x3_stream_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "hmmr_conv1x1_stream_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "hmmr_mfma_rate_probe": (C.c_int, [C.c_int, C.c_int, _fp, _vp]),
    "hmmr_clock_probe": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp]),
    "hmmr_render_handoff": (C.c_int, [_fp, C.c_int64, _fp, C.c_int64, _fp, C.c_int64, _fp, C.c_int, C.c_int, C.c_int,
                                      _fp, _fp, _fp, _vp]),
    "hmmr_eval_joints": (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _vp]),
    "hmmr_eval_verts": (C.c_int, [_fp, C.c_int64, _fp, C.c_int64, C.c_int, C.c_int, _fp, _vp]),
    "hmmr_global_rigid_transformation": (C.c_int, [_fp, _fp, _ip, C.c_int, _fp, _fp, C.c_int, _vp]),
    "hmmr_smpl_fwd_records": (C.c_int, [C.POINTER(SmplConsts), _fp, C.c_int, C.c_int, _fp, C.c_int64, C.POINTER(C.c_int32),
                                        _vp, C.c_size_t, _vp]),
    "hmmr_smpl_fwd_strided": (C.c_int, [C.POINTER(SmplConsts), _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, C.c_int,
                                        _fp, _fp, _fp, _fp, C.c_int64, _vp, C.c_size_t, _vp]),
}

_lib = None


class HmmrError(RuntimeError):
    pass


def load():
    """Load libhmmr_hip.so and bind every symbol of the header.  Raises if the
    library is missing -- there is deliberately no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HmmrError("%s not found: build it with `python -m human_dynamics_amd.build` "
                        "(the HIP library is mandatory; there is no CPU fallback)" % LIB_PATH)
    # PyTorch-ROCm ships its own HIP runtime (torch/lib/libamdhip64.so).  Import
    # torch FIRST so that libhmmr_hip.so binds to that already-loaded runtime
    # (same soname) -- two HIP runtimes in one process cannot see each other's
    # streams, allocations or code objects.
    import torch  # noqa: F401
    if torch.cuda.is_available():
        torch.cuda.init()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the .so does not export it
        fn.restype, fn.argtypes = res, args
    if lib.hmmr_abi_version() != ABI_VERSION:
        raise HmmrError("libhmmr_hip.so ABI version mismatch")
    _lib = lib
    return lib


def launch_counts(clear=False):
    """{kernel family: launches since the last clear} (hmmr_launch_counts)."""
    c = LaunchCounts()
    load().hmmr_launch_counts(C.byref(c), int(bool(clear)))
    return {k: int(getattr(c, k)) for k, _ in LaunchCounts._fields_}


def check(rc, what=""):
    if rc != 0:
        msg = load().hmmr_last_error()
        raise HmmrError("%s failed (%d): %s" % (what or "hmmr call", rc, msg.decode() if msg else "?"))


def ptr(t):
    """Device pointer of a torch tensor (or None) as a ctypes-compatible int."""
    return None if t is None else t.data_ptr()
