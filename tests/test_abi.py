"""The C-ABI library builds for gfx950, loads, and exports every symbol
include/hmmr_hip.h declares (no compute calls: there is no GPU here)."""
import os
import re

import pytest

from human_dynamics_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "hmmr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hmmr_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    build.build(verbose=False)
    lib = _lib.load()
    names = _declared_functions()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), "libhmmr_hip.so does not export %s" % n
    assert sorted(_lib.SIGNATURES) == names        # the ctypes table covers the header one to one
    assert lib.hmmr_abi_version() == _lib.ABI_VERSION == 19


def test_workspace_queries_need_no_gpu():
    lib = _lib.load()
    assert lib.hmmr_resnet50_workspace_bytes(0, _lib.HMMR_BF16) == 0
    b16 = lib.hmmr_resnet50_workspace_bytes(64, _lib.HMMR_BF16)
    f32 = lib.hmmr_resnet50_workspace_bytes(64, _lib.HMMR_F32)
    assert 0 < b16 < f32 <= 2 * b16 + 4096
    assert lib.hmmr_smpl_workspace_bytes(256) >= 256 * (224 + 288) * 4
    assert lib.hmmr_temporal_workspace_bytes(8, 20, _lib.HMMR_F32) >= 4 * 160 * 2048 * 4
    assert lib.hmmr_ief_workspace_bytes(160, 3, _lib.HMMR_F32) > 0
    # split (f16x3) tensors are 4 bytes per element, like fp32
    assert lib.hmmr_resnet50_workspace_bytes(64, _lib.HMMR_F16X3) == f32


def test_debug_switches_round_trip():
    import ctypes as C
    lib = _lib.load()
    d = _lib.Debug()
    lib.hmmr_get_debug(C.byref(d))
    assert (d.stem_route, d.stem_no_conv1) == (0, 0)          # product defaults
    d.stem_route, d.stem_no_conv1 = 1, 1
    lib.hmmr_set_debug(C.byref(d))
    e = _lib.Debug()
    lib.hmmr_get_debug(C.byref(e))
    assert (e.stem_route, e.stem_no_conv1) == (1, 1)
    lib.hmmr_set_debug(None)
    lib.hmmr_get_debug(C.byref(e))
    assert (e.stem_route, e.stem_no_conv1) == (0, 0)


def test_argument_validation_reports_errors():
    lib = _lib.load()
    d = _lib.ConvDesc()
    rc = lib.hmmr_conv_gemm(d, None)
    assert rc != 0 and b"null operand" in lib.hmmr_last_error()
    with pytest.raises(_lib.HmmrError):
        _lib.check(rc, "hmmr_conv_gemm")


def _dense_1x1(m=256, cin=64, cout=256, dtype=_lib.HMMR_F16X3):
    """a syntactically valid 1x1 descriptor with dummy (never dereferenced) pointers: validation runs before any launch"""
    d = _lib.ConvDesc()
    d.in_, d.w, d.out = 0x1000, 0x2000, 0x3000
    d.in_dtype = d.out_dtype = dtype
    d.n_img, d.hin, d.win, d.cin = 1, m, 1, cin
    d.in_img_stride, d.in_row_stride, d.in_px_stride = m * cin, cin, cin
    d.kh = d.kw = d.sy = d.sx = 1
    d.ho, d.wo, d.cout, d.ldo = m, 1, cout, cout
    return d


def test_conv_desc_validation_of_the_round_2_fields():
    """hmmr_conv_desc_t.in2 (a second operand source appended along K) and the 128x256 tile are refused with a message
    where the kernel could not honour them -- before anything is launched, so this runs without a GPU."""
    lib = _lib.load()
    d = _dense_1x1()
    d.in2, d.cin2 = 0x4000, 48                       # not a multiple of the 128-byte K step (32 split elements)
    assert lib.hmmr_conv_gemm(d, None) != 0 and b"multiple of 32" in lib.hmmr_last_error()
    d = _dense_1x1()
    d.in2, d.cin2, d.sy, d.sx, d.ho = 0x4000, 64, 2, 2, 128       # strided: the two sources would not share a pixel grid
    assert lib.hmmr_conv_gemm(d, None) != 0 and b"second operand source" in lib.hmmr_last_error()
    d = _dense_1x1()
    d.in2, d.cin2 = 0x4000, 64
    d.pro_scale = d.pro_shift = 0x5000               # the fused pre-activation takes the register route: no second source there
    assert lib.hmmr_conv_gemm(d, None) != 0
    d = _dense_1x1(cout=128)
    d.tile = 8
    assert lib.hmmr_conv_gemm(d, None) != 0 and b"tile 8" in lib.hmmr_last_error()
    t = _lib.TailDesc()
    t.dtype, t.h2, t.w3, t.res, t.m, t.c_mid, t.depth, t.n2 = _lib.HMMR_F16X3, 0x1000, 0x2000, 0x3000, 64, 256, 1024, 256
    t.w1 = t.out = t.out_h1 = t.pre_scale = t.pre_shift = t.scale1 = t.shift1 = 0x4000
    t.ldr = 1024
    assert lib.hmmr_bottleneck_tail(t, None) != 0 and b"supported shapes" in lib.hmmr_last_error()


def test_fragment_major_packing_layout():
    """packing.pack_frag_major: [n][K] -> [n / 32][K / 16][64 lanes][hi, lo][8]; lane = 32 * (k half) + row, i.e. the 16 bytes a
    lane feeds v_mfma_f32_32x32x16_f16 as its A operand (csrc/bottleneck_split.hip reads them straight from L2); fp16 halves
    of the rows scaled by packing.row_pow2."""
    import numpy as np
    import torch
    from human_dynamics_amd import packing
    rng = np.random.default_rng(0)
    w = rng.normal(size=(64, 48)).astype(np.float32)
    f = packing.pack_frag_major(w)
    assert tuple(f.shape) == (2, 3, 64, 2, 8) and f.dtype == torch.float16
    k = packing.row_pow2(w)
    ws = torch.from_numpy(packing.scale_rows(w, k))
    assert float(ws.abs().max()) < 2.0 ** 14 and float(ws.abs().max(dim=1).values.min()) >= 2.0 ** 13
    hi = ws.to(torch.float16)
    lo = (ws - hi.float()).to(torch.float16)
    for rb, kc, lane in ((0, 0, 0), (1, 2, 63), (0, 1, 37), (1, 0, 31)):
        row, half = rb * 32 + lane % 32, lane // 32
        k0 = kc * 16 + 8 * half
        assert torch.equal(f[rb, kc, lane, 0], hi[row, k0:k0 + 8]) and torch.equal(f[rb, kc, lane, 1], lo[row, k0:k0 + 8])
    back = (f[:, :, :, 0].float() + f[:, :, :, 1].float()).reshape(2, 3, 2, 32, 8).permute(0, 3, 1, 2, 4).reshape(64, 48)
    unscaled = back.double() / torch.from_numpy(np.exp2(k.astype(np.float64)))[:, None]
    assert float(((unscaled - torch.from_numpy(w).double()).abs() / torch.from_numpy(w).double().abs()).max()) < 2.0 ** -20


def test_engine_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from human_dynamics_amd.engine import HmmrEngine
    with pytest.raises(_lib.HmmrError):
        HmmrEngine(None, None)


def test_integration_doc_stub_matches_the_struct():
    """INTEGRATION.md shows a ctypes stub of hmmr_smpl_consts_t for a maintainer to paste: its fields are the
    library's, in order (the doc had drifted once, when `vpad` was added)."""
    import ctypes as C
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(root, "INTEGRATION.md")).read()
    block = txt[txt.index("class SmplConsts(C.Structure)"):]
    block = block[:block.index("def smpl_forward")]
    doc_fields = re.findall(r"\('(\w+)',\s*C\.(\w+)\)", block)
    lib_fields = [(n, "c_int" if t is C.c_int else "c_void_p") for n, t in _lib.SmplConsts._fields_]
    assert doc_fields == lib_fields
