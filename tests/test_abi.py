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
    assert lib.hmmr_abi_version() == _lib.ABI_VERSION == 9


def test_workspace_queries_need_no_gpu():
    lib = _lib.load()
    assert lib.hmmr_resnet50_workspace_bytes(0, _lib.HMMR_BF16) == 0
    b16 = lib.hmmr_resnet50_workspace_bytes(64, _lib.HMMR_BF16)
    f32 = lib.hmmr_resnet50_workspace_bytes(64, _lib.HMMR_F32)
    assert 0 < b16 < f32 <= 2 * b16 + 4096
    assert lib.hmmr_smpl_workspace_bytes(256) >= 256 * (224 + 288) * 4
    assert lib.hmmr_temporal_workspace_bytes(8, 20, _lib.HMMR_F32) >= 4 * 160 * 2048 * 4
    assert lib.hmmr_ief_workspace_bytes(160, 3, _lib.HMMR_F32) > 0
    # split (bf16x3) tensors are 4 bytes per element, like fp32
    assert lib.hmmr_resnet50_workspace_bytes(64, _lib.HMMR_BF16X3) == f32


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


def test_engine_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from human_dynamics_amd.engine import HmmrEngine
    with pytest.raises(_lib.HmmrError):
        HmmrEngine(None, None)
