"""Host-side packing / folding / windowing logic, checked on CPU against the
unfused forms (every fold re-associates fp32 arithmetic)."""
import ctypes as C

import numpy as np
import torch

from human_dynamics_amd import _lib, assets, packing
from human_dynamics_amd.evaluation import tester as T
from oracle import hmmr_oracle as O


class _HostStore(packing.DeviceStore):
    """DeviceStore that keeps everything on the CPU so the packers run without a GPU."""
    def __init__(self):
        super().__init__("cpu")


def _tensor_at(store, ptr):
    for t in store.tensors:
        if t.data_ptr() == ptr:
            return t
    raise KeyError(ptr)


def test_bn_fold_matches_unfused(weights):
    scope = "resnet_v2_50/block2/unit_1/bottleneck_v2/preact"
    s, b = packing.fold_bn(weights, scope)
    x = np.random.default_rng(0).normal(size=(7, 256))
    ref = (weights[scope + "/gamma"] * (x - weights[scope + "/moving_mean"])
           / np.sqrt(weights[scope + "/moving_variance"] + 1e-5) + weights[scope + "/beta"])
    assert np.allclose(x * s + b, ref, atol=1e-5)


def test_conv_weight_pack_is_k_contiguous_per_output_channel():
    w = np.random.default_rng(0).normal(size=(3, 3, 64, 96)).astype(np.float32)
    p = packing.pack_conv_weight(w)
    assert p.shape == (128, 576) and not p[96:].any()
    # k = (ky*kw + kx)*cin + ci
    assert p[5, (1 * 3 + 2) * 64 + 17] == w[1, 2, 17, 5]


def test_stem_pack_equals_7x7_conv_on_padded_rgbx(weights):
    """The 8-tap x 32-element GEMM on a zero-padded RGBX image is the slim stem conv."""
    w = weights["resnet_v2_50/conv1/weights"]
    img = assets.make_synthetic_frames(1, seed=3)[0, :40, :40]          # small crop, same algebra
    ref = O._conv(torch.tensor(img[None]).permute(0, 3, 1, 2).double(), w, torch.float64, stride=2, pad=3)[0]
    pk = packing.pack_stem_weight(w).astype(np.float64)                  # [128][256]
    H = img.shape[0]
    xp = np.zeros((H + 6 + 2, H + 6 + 8, 4))
    xp[3:3 + H, 3:3 + H, :3] = img
    Ho = ref.shape[1]
    out = np.zeros((64, Ho, Ho))
    for oy in range(Ho):
        for ox in range(Ho):
            a = np.concatenate([xp[2 * oy + ky, 2 * ox:2 * ox + 8].reshape(32) for ky in range(8)])
            out[:, oy, ox] = pk[:64] @ a
    assert np.abs(out - ref.numpy()).max() < 1e-9


def test_smpl_pack_joint_fold_and_sparse_forms(smpl_consts):
    st = _HostStore()
    sc = packing.pack_smpl(smpl_consts, st, impl="py")      # (one tensor per field to look at; the C packer's blob holds the same bytes: tests/test_packers.py)
    rng = np.random.default_rng(0)
    beta = rng.normal(size=10)
    v_shaped = (beta @ smpl_consts["shapedirs"].astype(np.float64)).reshape(-1, 3) + smpl_consts["v_template"]
    J_ref = smpl_consts["J_regressor"].astype(np.float64).T @ v_shaped
    jt = _tensor_at(st, sc.j_template).numpy().astype(np.float64)
    js = _tensor_at(st, sc.j_shapedirs).numpy().astype(np.float64)
    assert np.abs((jt + beta @ js).reshape(24, 3) - J_ref).max() < 1e-6
    # planar basis: dirs[k][c][v] = basis[k][3v + c]
    dirs = _tensor_at(st, sc.dirs).numpy()
    assert dirs.shape == (224, 3, 6912) and not dirs[218:].any()
    assert dirs[0, 1, 100] == smpl_consts["v_template"][100, 1]
    assert dirs[1 + 4, 2, 77] == smpl_consts["shapedirs"][4, 3 * 77 + 2]
    assert dirs[11 + 200, 0, 6889] == smpl_consts["posedirs"][200, 3 * 6889]
    # ELL skinning weights reproduce the dense matrix
    idx = _tensor_at(st, sc.lbs_idx).numpy(); val = _tensor_at(st, sc.lbs_w).numpy()
    dense = np.zeros((6890, 24), np.float32)
    for z in range(sc.lbs_nnz):
        np.add.at(dense, (np.arange(6890), idx[:, z]), val[:, z])
    assert np.array_equal(dense, smpl_consts["lbs_weights"])
    # CSR keypoint regressor reproduces the dense matrix
    kp = _tensor_at(st, sc.kreg_ptr).numpy(); ki = _tensor_at(st, sc.kreg_idx).numpy(); kv = _tensor_at(st, sc.kreg_val).numpy()
    dense = np.zeros((6890, 25), np.float32)
    for k in range(25):
        dense[ki[kp[k]:kp[k + 1]], k] = kv[kp[k]:kp[k + 1]]
    assert np.array_equal(dense, smpl_consts["cocoplus_regressor"])


def test_ief_pack_splits_fc1_and_orders_regressors(weights):
    st = _HostStore()
    iw, keys = packing.pack_ief(weights, _lib.HMMR_F32, st, (5, -5), impl="py")
    assert keys == [0, -5, 5] and iw.num_regressors == 3 and iw.num_stages == 3
    assert [iw.reg[i].nd for i in range(3)] == [85, 72, 72]
    W1 = weights["single_view_ief_past5/3D_module/fc1/weights"]
    wphi = _tensor_at(st, iw.reg[1].fc1_phi.w).numpy()
    wth = _tensor_at(st, iw.reg[1].fc1_theta.w).numpy()
    assert wphi.shape == (1024, 2048) and wth.shape == (1024, 128)
    assert np.array_equal(wphi, W1[:2048].T) and np.array_equal(wth[:, :72], W1[2048:].T) and not wth[:, 72:].any()
    w3 = _tensor_at(st, iw.reg[0].fc3.w).numpy()
    assert w3.shape == (128, 1024) and not w3[85:].any()


def test_resnet_pack_unit_table(weights):
    st = _HostStore()
    rw = packing.pack_resnet(weights, _lib.HMMR_F32, st, impl="py")
    strides = [rw.unit[i].stride for i in range(16)]
    assert strides == [1, 1, 2, 1, 1, 1, 2, 1, 1, 1, 1, 1, 2, 1, 1, 1]      # stride on the LAST unit of blocks 1-3
    assert [bool(rw.unit[i].shortcut.w) for i in range(16)] == [i in (0, 3, 7, 13) for i in range(16)]
    assert all(rw.unit[i].pre_scale and rw.unit[i].pre_shift for i in range(16))
    assert (rw.unit[0].c_in, rw.unit[15].depth) == (64, 2048)


def test_window_plan_matches_oracle():
    for n in (1, 8, 24, 64, 65, 256, 4096):
        for b in (1, 2, 8):
            assert T.window_plan(n, b, 20, 13) == O.window_plan(n, b, 20, 13)


def test_ctypes_struct_sizes_are_plausible():
    # catches accidental field drift between include/hmmr_hip.h and _lib.py
    assert C.sizeof(_lib.Layer) == 32
    assert C.sizeof(_lib.ResnetUnit) == 6 * 32 + 40 + 16 + 16 + 8
    assert C.sizeof(_lib.Debug) == 11 * 4 and C.sizeof(_lib.LaunchCounts) == 40      # (+ pair_form: ABI 18)
    assert C.sizeof(_lib.Var) == 24 and C.sizeof(_lib.SmplSource) == 8 + 7 * 8
    assert C.sizeof(_lib.ConvDesc) % 8 == 0


def test_packer_marks_the_fused_launches(weights):
    """Which units the packer folds into one launch (host-side decision, no GPU needed): the stride-1
    units of blocks 1-2 end in a fused tail with their conv2 inside, the first units of blocks 3-4 run
    shortcut + conv1 as one column-split GEMM; fp32 keeps every layer separate except the column split."""
    from human_dynamics_amd import packing
    rw = packing.pack_resnet(weights, _lib.HMMR_BF16, packing.DeviceStore("cpu"))
    assert [rw.unit[i].fuse_tail for i in range(16)] == [3, 2, 4, 2, 2, 2, 4] + [0] * 9
    assert [i for i in range(16) if rw.unit[i].sc_c1.w] == [7, 13]
    assert [i for i in range(16) if not rw.unit[i].fuse_preact] == [0, 3, 7, 13]      # the blocks' first units
    assert not any(rw.unit[i].c3sc.w for i in range(16))
    # f16x3: every conv shortcut is folded into its unit's conv3 (one GEMM over {h2, preact}); no column-split GEMMs then
    rwx = packing.pack_resnet(weights, _lib.HMMR_F16X3, packing.DeviceStore("cpu"))
    # (block3/unit_1 excepted: it heads a chain of register-resident unit pairs, csrc/unit_pair.hip, and runs shortcut + conv1 as one
    #  column-split GEMM)
    assert [i for i in range(16) if rwx.unit[i].c3sc.w] == [0, 3, 13] and [i for i in range(16) if rwx.unit[i].sc_c1.w] == [7]
    # ... and conv3 + add + the next conv1 run as one launch for the stride-1 units of blocks 1-3 with an identity successor
    assert [rwx.unit[i].fuse_tail for i in range(16)] == [2, 2, 0, 1, 1, 1, 0, 1, 1, 1, 1, 1, 0, 0, 0, 0]      # block 1: conv2 inside as well
    # (b1_stream: block 1's conv2 as a launch of the 3x3 stream kernel, csrc/conv3x3_stream.hip, the tails start at conv3)
    b1s = packing.pack_resnet(weights, _lib.HMMR_F16X3, packing.DeviceStore("cpu"), b1_stream=True)
    assert [b1s.unit[i].fuse_tail for i in range(3)] == [1, 1, 0] and [b1s.unit[i].conv2.k_order for i in range(3)] == [2, 2, 0]
    # round 5: block1/unit_1 and unit_2 are whole-unit launches (csrc/b1_unit.hip): one filter stream each, conv2 chunk-major (k_order 2)
    assert [bool(rwx.unit[i].unit_stream) for i in range(16)] == [True, True] + [False] * 14 and not any(rwx.unit[i].w3_frag for i in range(16))
    assert [rwx.unit[i].conv2.k_order for i in range(3)] == [2, 2, 0] and not any(b1s.unit[i].unit_stream for i in range(16))
    r3 = packing.pack_resnet(weights, _lib.HMMR_F16X3, packing.DeviceStore("cpu"), b1_unit=False)       # the round-3 / 4 block 1
    assert [bool(r3.unit[i].w3_frag) for i in range(16)] == [True, True] + [False] * 14              # LDS-panel tails, fragment-major filters
    assert [r3.unit[i].fuse_tail for i in range(3)] == [2, 2, 0] and [r3.unit[i].conv2.k_order for i in range(3)] == [0, 0, 0]
    assert not any(r3.unit[i].unit_stream for i in range(16))
    assert [bool(rwx.unit[i].pair_stream) for i in range(16)] == [False] * 3 + [True] * 3 + [False] + [True] * 5 + [False] * 4
    old = packing.pack_resnet(weights, _lib.HMMR_F16X3, packing.DeviceStore("cpu"), unit_pair=False)      # the round-3 schedule
    assert [i for i in range(16) if old.unit[i].c3sc.w] == [0, 3, 7, 13] and not any(old.unit[i].sc_c1.w for i in range(16))
    assert [old.unit[i].fuse_tail for i in range(16)] == [2, 2, 0, 0, 1, 1, 0] + [0] * 9 and not any(old.unit[i].pair_stream for i in range(16))
    rw32 = packing.pack_resnet(weights, _lib.HMMR_F32, packing.DeviceStore("cpu"))
    assert sum(rw32.unit[i].fuse_tail for i in range(16)) == 0
    assert [i for i in range(16) if rw32.unit[i].sc_c1.w] == [7, 13]
    off = packing.pack_resnet(weights, _lib.HMMR_BF16, packing.DeviceStore("cpu"), fuse_tail=False, fuse_sc=False)
    assert sum(off.unit[i].fuse_tail for i in range(16)) == 0 and not any(off.unit[i].sc_c1.w for i in range(16))


def test_resnet_part_boundaries():
    """engine.resnet_cuts: the contiguous parts a ResNet call runs as (what the host streamer cuts its uploads by);
    host arithmetic only -- the object is built without a device."""
    from human_dynamics_amd.engine import HmmrEngine
    e = HmmrEngine.__new__(HmmrEngine)
    e.resnet_streams, e.resnet_chunk = 2, 0
    assert e.resnet_cuts(256) == [0, 128, 256] and e.resnet_cuts(257) == [0, 128, 257]
    assert e.resnet_cuts(175) == [0, 175] and e.resnet_cuts(160) == [0, 160] and e.resnet_cuts(176) == [0, 88, 176] and e.resnet_cuts(0) == [0, 0]
    assert e.resnet_cuts(300, parts=3) == [0, 100, 200, 300] and e.resnet_cuts(300, parts=1) == [0, 300]
    e.resnet_chunk = 64                     # dev switch: sequential chunks, one stream
    assert e.resnet_cuts(256) == [0, 256]


def test_filter_pack_order():
    """pack_conv_weight: row co of the filter bank is K-contiguous with k = (ky*kw + kx)*cin + ci -- the order
    hmmr_conv_gemm gathers the A operand in (csrc/gemm_conv.hip tap_of); rows padded to 128."""
    w = np.random.default_rng(0).normal(size=(3, 3, 128, 40)).astype(np.float32)
    p0 = packing.pack_conv_weight(w)
    assert p0.shape == (128, 1152) and not p0[40:].any()
    for ky, kx, ci, co in ((0, 0, 0, 0), (1, 2, 37, 5), (2, 2, 127, 39), (0, 1, 64, 3)):
        assert p0[co, (ky * 3 + kx) * 128 + ci] == w[ky, kx, ci, co]


def test_chunk_major_filter_pack():
    """pack_conv_weight(k_order=1): k' = ((ci // 32)*9 + tap)*32 + ci % 32 -- the K order of the 3x3 patch kernel
    (hmmr_conv_desc_t.k_order = 1: K step kt = chunk kt // 9, tap kt % 9).  pack_resnet gives it to the stride-1 conv2 of
    blocks 2-4 in the f16x3 mode only: block 1's fused tails sum tap-major, the stride-2 units keep the im2col gather."""
    w = np.random.default_rng(0).normal(size=(3, 3, 128, 40)).astype(np.float32)
    p0, p1 = packing.pack_conv_weight(w), packing.pack_conv_weight(w, 1)
    assert p0.shape == p1.shape == (128, 1152) and np.array_equal(np.sort(p0, axis=1), np.sort(p1, axis=1))
    for ky, kx, ci, co in ((0, 0, 0, 0), (1, 2, 37, 5), (2, 2, 127, 39), (0, 1, 64, 3)):
        tap = ky * 3 + kx
        assert p0[co, tap * 128 + ci] == w[ky, kx, ci, co]
        assert p1[co, ((ci // 32) * 9 + tap) * 32 + ci % 32] == w[ky, kx, ci, co]
    ws = assets.make_synthetic_weights(0)
    on = packing.pack_resnet(ws, _lib.HMMR_F16X3, packing.DeviceStore("cpu"), patch_3x3=1)
    assert [on.unit[i].conv2.k_order for i in range(16)] == [0, 0, 0, 1, 1, 1, 0, 1, 1, 1, 1, 1, 0, 1, 1, 1]
    # the default: the same layers as the filter stream of the one-wave-per-SIMD kernel (k_order 2, csrc/conv3x3_stream.hip)
    st = packing.pack_resnet(ws, _lib.HMMR_F16X3, packing.DeviceStore("cpu"))
    assert [st.unit[i].conv2.k_order for i in range(16)] == [2, 2, 0, 2, 2, 2, 0, 2, 2, 2, 2, 2, 0, 2, 2, 2]
    off = packing.pack_resnet(ws, _lib.HMMR_F16X3, packing.DeviceStore("cpu"), patch_3x3=False)
    assert not any(off.unit[i].conv2.k_order for i in range(16))
    # bf16 (round 4): blocks 3-4 only -- the conv2 of blocks 1-2 runs inside the fused bf16 units, which read the tap-major order
    b16 = packing.pack_resnet(ws, _lib.HMMR_BF16, packing.DeviceStore("cpu"), patch_3x3=1)
    assert [b16.unit[i].conv2.k_order for i in range(16)] == [0] * 7 + [1, 1, 1, 1, 1, 0, 1, 1, 1]
    b16s = packing.pack_resnet(ws, _lib.HMMR_BF16, packing.DeviceStore("cpu"))              # default: the stream kernel's bf16 form
    assert [b16s.unit[i].conv2.k_order for i in range(16)] == [0] * 7 + [2, 2, 2, 2, 2, 0, 2, 2, 2]
    assert not any(packing.pack_resnet(ws, _lib.HMMR_F32, packing.DeviceStore("cpu")).unit[i].conv2.k_order for i in range(16))
    w64 = np.random.default_rng(1).normal(size=(3, 3, 128, 64)).astype(np.float32)
    p64 = packing.pack_conv_weight(w64, 1, chunk=64)          # bf16: 64 elements per 128-byte K step
    assert p64[5, ((70 // 64) * 9 + 4) * 64 + 70 % 64] == w64[1, 1, 70, 5]


def test_conv3x3_stream_pack():
    """packing.pack_conv3x3_stream: K step kt = (ci // 16) * 9 + tap of a 128-channel tile = 4 row blocks x (hi | lo plane) of MFMA
    A-operand fragments, lane = 32 * (k half) + row (hmmr_conv_desc_t.k_order = 2); hi + lo reproduce the scaled filter to 2^-22."""
    w = np.random.default_rng(2).normal(size=(3, 3, 64, 256)).astype(np.float32)
    k = packing.row_pow2(packing.pack_conv_weight(w))
    st = packing.pack_conv3x3_stream(w, k)
    assert tuple(st.shape) == (2, 36, 4, 2, 64, 8) and st.dtype == torch.float16
    for ky, kx, ci, co in ((0, 0, 0, 0), (1, 2, 37, 5), (2, 2, 63, 255), (0, 1, 17, 130)):
        tile, rb, row = co // 128, (co % 128) // 32, co % 32
        kt, half, e = (ci // 16) * 9 + ky * 3 + kx, (ci % 16) // 8, ci % 8
        got = float(st[tile, kt, rb, 0, 32 * half + row, e]) + float(st[tile, kt, rb, 1, 32 * half + row, e])
        want = float(w[ky, kx, ci, co]) * 2.0 ** int(k[co])
        assert abs(got - want) <= abs(want) * 2.0 ** -21
    # the size the library expects (hmmr_conv3x3_stream_bytes: host arithmetic, no device needed); cout = 64: one tile of two row blocks
    lib = _lib.load()
    assert st.numel() * 2 == lib.hmmr_conv3x3_stream_bytes(64, 256)
    w64 = np.random.default_rng(3).normal(size=(3, 3, 64, 64)).astype(np.float32)
    s64 = packing.pack_conv3x3_stream(w64)
    assert tuple(s64.shape) == (1, 36, 2, 2, 64, 8) and s64.numel() * 2 == lib.hmmr_conv3x3_stream_bytes(64, 64)
    # bf16 tensors: K steps of 32 channels, the planes = the two 16-wide MFMA chunks, half the bytes
    sb = packing.pack_conv3x3_stream(w, bf16=True)
    assert tuple(sb.shape) == (2, 18, 4, 2, 64, 8) and sb.dtype == torch.bfloat16 and sb.numel() * 2 * 2 == lib.hmmr_conv3x3_stream_bytes(64, 256)
    for ky, kx, ci, co in ((0, 0, 0, 0), (1, 2, 37, 5), (2, 2, 63, 255), (0, 1, 17, 130)):
        tile, rb, row = co // 128, (co % 128) // 32, co % 32
        kt, plane, half, e = (ci // 32) * 9 + ky * 3 + kx, (ci % 32) // 16, (ci % 16) // 8, ci % 8
        assert float(sb[tile, kt, rb, plane, 32 * half + row, e]) == float(torch.tensor(w[ky, kx, ci, co]).to(torch.bfloat16))


def test_b1_unit_stream_pack():
    """packing.pack_b1_unit_stream (hmmr_tail_desc_t.unit_stream, csrc/b1_unit.hip): conv2's k_order 2 stream, then the tail as
    A(0) | A(1) B(0) | ... | A(7) B(6) | B(7) (A(c): the K3 / 16 conv3 fragments of output chunk c, B(c): the four conv1' fragments of K chunks 2 c, 2 c + 1); a fragment = [hi | lo plane][lane =
    32 * (k half) + row][8]; hi + lo reproduce the scaled filter rows."""
    rng = np.random.default_rng(5)
    w2 = rng.normal(size=(3, 3, 64, 64)).astype(np.float32)
    k2 = packing.row_pow2(packing.pack_conv_weight(w2)[:64])
    lib = _lib.load()
    for K3 in (64, 128):
        w3 = rng.normal(size=(256, K3)).astype(np.float32)
        w1 = rng.normal(size=(64, 256)).astype(np.float32)
        st = packing.pack_b1_unit_stream(w2, k2, w3, w1)
        nf = K3 // 16 + 4
        assert tuple(st.shape) == (72 + 8 * nf, 2, 64, 8) and st.dtype == torch.float16
        assert st.numel() * 2 == lib.hmmr_b1_unit_stream_bytes(K3 - 64)
        assert torch.equal(st[:72].reshape(-1), packing.pack_conv3x3_stream(w2, k2).reshape(-1))
        k3, k1 = packing.row_pow2(w3), packing.row_pow2(w1)
        val = lambda f, lane, e: float(st[f, 0, lane, e]) + float(st[f, 1, lane, e])
        na = K3 // 16
        pos_a = lambda c: 72 if c == 0 else 72 + na + (c - 1) * nf              # first fragment of A(c) / B(c) in the pipelined order
        pos_b = lambda c: 72 + 2 * na + c * nf if c < 7 else 72 + 8 * nf - 4
        for co, ci in ((0, 0), (37, 5), (255, K3 - 1), (130, 17)):              # conv3: W3[co][ci]
            f = pos_a(co // 32) + ci // 16
            want = float(w3[co, ci]) * 2.0 ** int(k3[co])
            assert abs(val(f, 32 * ((ci % 16) // 8) + co % 32, ci % 8) - want) <= abs(want) * 2.0 ** -21
        for n2, ci in ((0, 0), (33, 47), (63, 255), (5, 144)):                  # conv1': W1[n2][ci], K chunk ci // 16 = 2 c + kcl
            c, kcl = ci // 32, (ci // 16) % 2
            f = pos_b(c) + kcl * 2 + n2 // 32
            want = float(w1[n2, ci]) * 2.0 ** int(k1[n2])
            assert abs(val(f, 32 * ((ci % 16) // 8) + n2 % 32, ci % 8) - want) <= abs(want) * 2.0 ** -21


def test_tuner_candidates_map_onto_the_stream_kernels_tiles():
    """HmmrEngine._tile_for: a k_order 2 layer (csrc/conv3x3_stream.hip) takes the tuner's candidate ids as its own tile shapes; the 7 x 1
    wave tile (21) is split-only, the 64-channel tiles (19 / 20) belong to 64-channel layers; what does not fit maps to 0 (the library's
    choice: fewest rounds of 256 workgroups)."""
    from human_dynamics_amd import engine as E
    lay = _lib.Layer()
    lay.k_order = 2
    f = E.HmmrEngine._tile_for
    assert [f(lay, c, 256, _lib.HMMR_F16X3) for c in (0, 5, 6, 3, 1, 2, 7, 8, 11)] == [0, 13, 14, 15, 16, 17, 18, 12, 21]
    assert f(lay, 21, 256, _lib.HMMR_F16X3) == 21 and f(lay, 21, 256, _lib.HMMR_BF16) == 0 and f(lay, 11, 512, _lib.HMMR_BF16) == 0
    assert f(lay, 12, 256, _lib.HMMR_BF16) == 12 and f(lay, 10, 256, _lib.HMMR_F16X3) == 0
    # round 6: the 128- / 192-pixel tiles for short launches (27 / 28: candidates 4 / 9), split-only like 21
    assert [f(lay, c, 256, _lib.HMMR_F16X3) for c in (4, 9, 27, 28)] == [27, 28, 27, 28] and f(lay, 4, 256, _lib.HMMR_BF16) == 0
    # ... and of the 1x1 stream kernel (29: candidate 4), conv1 form only
    assert f(lay, 4, 512, _lib.HMMR_F16X3, "conv1") == 29 and f(lay, 29, 2048, _lib.HMMR_F16X3, "conv3") == 0
    plain = _lib.Layer()
    assert f(plain, 4, 256, _lib.HMMR_F16X3, "conv3") == 0 and f(plain, 26, 256, _lib.HMMR_F16X3, "conv1") == 0      # (the generic kernel has no tile 4 / 26)
    assert [f(lay, c, 64, _lib.HMMR_F16X3) for c in (0, 5, 6, 19, 20, 12)] == [0, 19, 20, 19, 20, 0]


def test_shipped_tile_tables_fit_their_layers():
    """human_dynamics_amd/tile_tables.json: every entry names a layer of the ResNet and a tile that layer's launch accepts
    (csrc/gemm_conv.hip: 128- / 256-column tiles need cout % 128 / 256 == 0; a layer packed chunk-major runs the patch tiles
    9 / 10 / 11, every other layer the tiles 1 ... 8), for the three operand modes and the batch sizes the BASELINE
    configurations produce.  HmmrEngine._tile_for is the same rule at run time (it maps what does not fit to 0)."""
    import json
    import os
    from human_dynamics_amd import engine as E
    tabs = json.load(open(os.path.join(os.path.dirname(E.__file__), "tile_tables.json")))
    ws = assets.make_synthetic_weights(0)
    sizes = set()
    for dt in (_lib.HMMR_F32, _lib.HMMR_BF16, _lib.HMMR_F16X3):
        rw = packing.pack_resnet(ws, dt, packing.DeviceStore("cpu"))
        for key, tab in tabs.items():
            if key.startswith("_") or int(key.split(":")[0]) != dt:
                continue
            sizes.add(int(key.split(":")[1]))
            for lk, tile in tab.items():
                u, nm = int(lk.split(":")[0]), lk.split(":")[1]
                U = rw.unit[u]
                assert 0 <= u < 16 and nm in ("conv1", "conv2", "conv3", "shortcut")
                lay = U.c3sc if (nm == "conv3" and U.c3sc.w) else getattr(U, nm)
                cout = U.depth + U.base if (nm == "shortcut" and U.sc_c1.w) else (U.base if nm in ("conv1", "conv2") else U.depth)
                assert E.HmmrEngine._tile_for(lay, tile, cout, dt, nm) == tile, (key, lk, tile)
                ok = {0: (0, 1, 2, 3, 5, 6, 7, 8), 1: (0, 9, 10, 11), 2: (0, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 27, 28)}[lay.k_order]
                if lay.k_order == 2 and nm != "conv2":          # a 1x1 layer of csrc/conv1x1_stream.hip
                    ok = (0, 24, 25, 26) if nm == "conv3" else (0, 22, 23, 24, 25, 26, 29)
                assert tile in ok, (key, lk, tile)
    assert {40, 64, 65, 128, 129, 256, 257, 512, 513, 1024} <= sizes


def test_b1_unit_instruction_stream_matches_its_wait_table():
    """csrc/b1_unit.hip waits on its filter ring / patch / shortcut chunks with COUNTED s_waitcnt vmcnt(N): N comes from a constexpr table
    of the vector-memory instructions each wave issues per slab step, so the table and the instruction stream hipcc emits have to agree --
    a reordered store or a request the table does not know would make a wave read a slab before it has landed.  This test cross-compiles
    the file (no GPU needed), walks both kernels' ISA and checks, step by step, the global_load_lds / global_store counts between the
    barriers and every vmcnt against a Python restatement of the table."""
    import os
    import subprocess
    import tempfile
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(here, "human_dynamics_amd", "csrc")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "b1.s")
        from human_dynamics_amd import build as B
        cmd = [B.HIPCC] + [f for f in B.FLAGS if f != "-fPIC"] + ["-x", "hip", "--cuda-device-only", "-S", os.path.join(csrc, "b1_unit.hip"), "-o", out]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout[-2000:]
        asm = open(out).read()
    NCH, CONV2, NS = 8, 18, 5

    def table(kc3, res):
        na = kc3 // 4
        total = CONV2 + NCH * (na + 1)
        step_b = lambda c: CONV2 + 2 * na + c * (na + 1) if c + 1 < NCH else total - 1
        chunk_of_b = {step_b(c): c for c in range(NCH)}
        ring = lambda t: 2 if t + NS - 1 < total else 0
        first = lambda t: 4 if (t == CONV2 and res) else 0

        def last(t):
            if t in (5, 9):
                return 4
            if t == 14:
                return 4 if res else 0
            c = chunk_of_b.get(t)
            return 0 if c is None else 4 + (4 if (res and c + 2 < NCH) else 0)

        def cum(t, phase):
            n = 0
            for u in range(t + 1):
                n += ring(u)
                if u < t or phase >= 1:
                    n += first(u)
                if u < t or phase >= 2:
                    n += last(u)
            return n

        def wait_n(s):
            cover = cum(s - 4, 2)
            c = chunk_of_b.get(s)
            if res and c == 1:
                cover = max(cover, cum(CONV2, 1))
            if res and c is not None and c >= 2:
                cover = max(cover, cum(step_b(c - 2), 2))
            return cum(s - 1, 2) - cover
        return total, ring, first, last, wait_n

    for sym, kc3, res in (("ILi0ELb1EEE", 4, True), ("ILi4ELb0EEE", 8, False)):
        start = asm.index("_ZN12_GLOBAL__N_114b1_unit_kernel%svNS_6B1ArgsE:" % sym)
        body = asm[start:asm.index("s_endpgm", start)]
        ev = []
        for line in body.splitlines():
            t = line.strip().split()
            if not t:
                continue
            if t[0].startswith("global_load_lds"):
                ev.append("D")
            elif t[0].startswith("global_store"):
                ev.append("S")
            elif t[0] == "s_barrier":
                ev.append("|")
            elif t[0] == "s_waitcnt" and "vmcnt" in line:
                ev.append(int(line.split("vmcnt(")[1].split(")")[0]))
        total, ring, first, last, wait_n = table(kc3, res)
        # skip the prologue: everything up to and including its barrier (the first one)
        i = ev.index("|") + 1
        for s in range(total):
            waits = []
            while ev[i] != "|":                                  # what sits in front of step s's barrier: its counted wait (from step 4 on)
                assert isinstance(ev[i], int) or s == 0, (sym, s, ev[i])
                if isinstance(ev[i], int):
                    waits.append(ev[i])
                i += 1
            i += 1
            if s >= NS - 1:
                assert waits and waits[-1] == wait_n(s), (sym, "step", s, "vmcnt", waits, "table", wait_n(s))
            j, ops = i, []
            while j < len(ev) and ev[j] in ("D", "S"):
                ops.append(ev[j])
                j += 1
            want_d_first = ring(s) + first(s)
            if s == total - 1:
                ops = ops[:last(s)]                              # (the last step's stores are followed by the eight h1' stores)
            lst = last(s)
            stores = 4 if (lst >= 4 and s not in (5, 9, 14)) else 0
            assert ops == ["D"] * want_d_first + ["S"] * stores + ["D"] * (lst - stores), (sym, "step", s, "".join(ops), want_d_first, stores, lst)
            i = j if s < total - 1 else i


def test_conv1x1_stream_instruction_stream_matches_its_counted_waits():
    """csrc/conv1x1_stream.hip waits on its two operand rings with COUNTED s_waitcnt vmcnt(N): a K step issues P = 2 + BM / 64 requests per
    wave for stage kt + D and ends on vmcnt((D - 2) P); the conv3 form's epilogue counts shortcut requests and stores.  This test
    cross-compiles the file (no GPU needed) and walks EVERY instantiation's ISA: D P requests in the prologue, then per unrolled step exactly
    P requests, one counted wait and one barrier (nothing the compiler added in between), the drain in front of the epilogue, and for the
    one-workgroup-per-CU conv3 forms the epilogue's wait sequence 4 x (blocks ahead) + stores x (blocks behind)."""
    import os
    import re
    import subprocess
    import tempfile
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(here, "human_dynamics_amd", "csrc")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "s1.s")
        from human_dynamics_amd import build as B
        cmd = [B.HIPCC] + [f for f in B.FLAGS if f != "-fPIC"] + ["-x", "hip", "--cuda-device-only", "-S", os.path.join(csrc, "conv1x1_stream.hip"), "-o", out]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout[-2000:]
        asm = open(out).read()
    syms = re.findall(r"^(_ZN12_GLOBAL__N_121conv1x1_stream_kernelI(\w+?)EEvNS_6S1ArgsE):", asm, re.M)
    assert len(syms) >= 14
    seen = set()
    for sym, targs in syms:
        vals = [int(v) for v in re.findall(r"L[ib](\d+)E", targs)]
        FM, FN, WGM, WGN, D, EPI, IN2, RES, OUT2, OCC = vals
        seen.add((FM, FN, D, EPI, OCC))
        BM = 32 * WGM * FM
        P = 2 + (BM + 63) // 64
        U = 2 * D if D % 2 else D
        start = asm.index(sym + ":")
        body = asm[start:asm.index("s_endpgm", start)]
        ev = []
        for line in body.splitlines():
            t = line.split(";")[0].strip().split()
            if not t:
                continue
            if t[0].startswith("global_load_lds"):
                ev.append("D")
            elif t[0].startswith("global_store"):
                ev.append("S")
            elif t[0] == "s_barrier":
                ev.append("|")
            elif t[0] == "s_waitcnt" and "vmcnt" in line.split(";")[0]:
                ev.append(int(line.split("vmcnt(")[1].split(")")[0]))
        bars = [i for i, e in enumerate(ev) if e == "|"]
        assert len(bars) == 2 + U + 1, (targs, len(bars))
        # prologue: D stages of P requests, the counted waits for stage 0 and stage 1 in front of the first two barriers
        assert ev[:bars[0]].count("D") == D * P and ev[bars[0] - 1] == (D - 1) * P, (targs, ev[:bars[0]])
        assert ev[bars[0] + 1:bars[1]] == [(D - 2) * P], (targs, ev[bars[0] + 1:bars[1]])
        # the unrolled steps: P requests, ONE wait, the barrier
        for s in range(U):
            seg = ev[bars[1 + s] + 1:bars[2 + s]]
            assert seg == ["D"] * P + [(D - 2) * P], (targs, "step", s, seg)
        # the drain in front of the epilogue (the repeats of the last stage land in rings the epilogue reuses)
        assert ev[bars[1 + U] + 1:bars[2 + U]] == [0], (targs, ev[bars[1 + U] + 1:bars[2 + U]])
        tail = ev[bars[2 + U] + 1:]
        nblk = FM * FN
        if EPI == 0:
            assert tail.count("S") == 4 * nblk and "D" not in tail, (targs, tail)
        else:
            nst = 8 if OUT2 else 4
            assert tail.count("S") == nst * nblk and tail.count("D") == (4 * nblk if RES else 0), (targs, tail.count("S"), tail.count("D"))
            if RES and OCC == 1:
                pd = 4
                want = [4 * min(nblk - 1 - b, pd) + nst * min(b, pd) for b in range(nblk)]
                assert [e for e in tail if isinstance(e, int)] == want, (targs, [e for e in tail if isinstance(e, int)], want)
    assert {(7, 1, 6, 0, 1), (8, 1, 6, 0, 1), (7, 2, 4, 0, 1), (4, 2, 6, 0, 1), (4, 2, 3, 0, 2), (7, 2, 4, 1, 1), (4, 2, 3, 1, 2)} <= seen


def test_conv1x1_stream_pack_and_packer_marks():
    """packing.pack_conv1x1_stream (hmmr_conv_desc_t.k_order = 2 on a 1x1 filter, csrc/conv1x1_stream.hip): [cout / 128][cin / 16][4 row
    blocks][hi | lo plane][lane = 32 * (k half) + row][8]; hi + lo reproduce the scaled filter rows; the size the library expects.
    pack_resnet marks block 4's conv1 / conv3, block2/unit_1's conv1 and block3/unit_1's shortcut + conv1 bank for it in the split mode
    only, and those units read a materialised pre-activation (fuse_preact = 0); the tuner's candidates map onto tiles 22 .. 26."""
    from human_dynamics_amd import engine as E
    rng = np.random.default_rng(9)
    lib = _lib.load()
    for cin, cout in ((256, 128), (2048, 512), (512 + 1024, 2048)):
        w = rng.normal(size=(1, 1, cin, cout)).astype(np.float32)
        st = packing.pack_conv1x1_stream(w)
        assert tuple(st.shape) == (cout // 128, cin // 16, 4, 2, 64, 8) and st.dtype == torch.float16
        assert st.numel() * 2 == lib.hmmr_conv1x1_stream_bytes(cin, cout)
        k = packing.row_pow2(w[0, 0].T)
        for co, ci in ((0, 0), (cout - 1, cin - 1), (77, 130), (cout // 2 + 3, 21)):
            t, rb, row = co // 128, (co % 128) // 32, co % 32
            lane, e = 32 * ((ci % 16) // 8) + row, ci % 8
            got = float(st[t, ci // 16, rb, 0, lane, e]) + float(st[t, ci // 16, rb, 1, lane, e])
            want = float(w[0, 0, ci, co]) * 2.0 ** int(k[co])
            assert abs(got - want) <= abs(want) * 2.0 ** -21
    ws = assets.make_synthetic_weights(0)
    rw = packing.pack_resnet(ws, _lib.HMMR_F16X3, packing.DeviceStore("cpu"))
    c1 = [i for i in range(16) if rw.unit[i].conv1.k_order == 2]
    assert c1 == [3, 7, 13, 14, 15] and [i for i in range(16) if rw.unit[i].conv3.k_order == 2] == [13, 14, 15]
    assert rw.unit[7].sc_c1.k_order == 2 and rw.unit[7].shortcut.k_order == 2 and rw.unit[13].c3sc.k_order == 2
    assert all(rw.unit[i].fuse_preact == 0 for i in c1) and rw.unit[12].fuse_preact == 1
    for dt, kw in ((_lib.HMMR_BF16, {}), (_lib.HMMR_F32, {}), (_lib.HMMR_F16X3, dict(stream_1x1=False))):
        other = packing.pack_resnet(ws, dt, packing.DeviceStore("cpu"), **kw)
        assert all(other.unit[i].conv1.k_order == 0 and other.unit[i].conv3.k_order == 0 and other.unit[i].sc_c1.k_order == 0 for i in range(16))
    f = E.HmmrEngine._tile_for
    lay = rw.unit[14].conv1
    assert [f(lay, c, 512, _lib.HMMR_F16X3, "conv1") for c in (0, 5, 6, 3, 1, 2, 7, 8, 11, 24, 26)] == [0, 22, 23, 24, 25, 26, 0, 0, 0, 24, 26]
    assert [f(rw.unit[14].conv3, c, 2048, _lib.HMMR_F16X3, "conv3") for c in (0, 5, 6, 3, 1, 2, 22, 26)] == [0, 0, 0, 24, 25, 26, 0, 26]


def test_bench_smi_sampler_without_a_gpu_reports_nothing():
    """bench.SmiSampler is a reporting aid: where amdsmi cannot be initialised (this container) start / stop do nothing and return None."""
    import bench
    s = bench.SmiSampler(0)
    s.start()
    r = s.stop()
    assert r is None or isinstance(r, dict)
