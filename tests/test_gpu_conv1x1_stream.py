"""csrc/conv1x1_stream.hip on the GPU: the 1x1 layers of the late ResNet blocks (slim resnet_v2.bottleneck conv1 / shortcut, src/models.py:65-75)
as one MFMA stream per SIMD with both operands in LDS rings (hmmr_conv_desc_t.k_order = 2 with a 1x1 filter, tiles 22 .. 25)."""
import zlib

import numpy as np
import pytest

from human_dynamics_amd import _lib as L
from test_gpu_f16x3 import _split_round, _split_round_w
from test_gpu_kernels import _ref_conv

pytestmark = pytest.mark.gpu
X3 = L.HMMR_F16X3

CASES = [
    # name, n, h, w, cin, cout, n_split
    ("b4_conv1_1024", 3, 7, 7, 1024, 512, 0),          # 147 pixels: one ragged tile
    ("b4_conv1_2048", 11, 7, 7, 2048, 512, 0),         # 539 pixels: 3 tiles of 224, K = 128 steps
    ("b31_sc_c1", 2, 14, 14, 512, 1280, 1024),         # shortcut + conv1 of block3/unit_1 as one launch: 10 N tiles, split after 8
    ("short_k", 1, 5, 7, 128, 128, 0),                 # 8 K steps (cin is a power of two: 6 + 2 for the 6-deep rings), 35 pixels
    ("k_not_a_group", 5, 14, 14, 256, 256, 128),       # 16 K steps: 6 + 6 + 4, the last group of the unrolled loop is cut short
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv1x1_stream_kernel(case, gpu_device):
    """Against a float64 convolution of the same 16-bit operands (2e-5 of the largest value, the bound of every split kernel's test) and
    against hmmr_conv_gemm's 8-wave tiles (another accumulation order); the four tile shapes agree bit for bit, with and without ReLU, and
    the column split writes each range with its own ReLU flag."""
    from human_dynamics_amd.engine import conv_gemm
    name, n, h, w_, cin, cout, n_split = case
    rng = np.random.default_rng(zlib.crc32(name.encode()) % 1000)
    x = rng.normal(size=(n, h, w_, cin)).astype(np.float32)
    w = (rng.normal(size=(1, 1, cin, cout)) / np.sqrt(cin)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.normal(size=cout).astype(np.float32)
    for relu in (True, False):
        kw = dict(stride=1, pad=0, scale=scale, shift=shift, relu=relu, in_dtype=X3, out_dtype=X3, device=gpu_device)
        if n_split:
            kw.update(n_split=n_split, relu_b=not relu)
        outs = {tile: conv_gemm(x, w, tile=tile, k_order=2, **kw) for tile in ((0, 22, 23, 24, 25, 26, 29) if relu else (0, 24, 26, 29))}
        ref, _ = _ref_conv(_split_round(x), _split_round_w(w), 1, 0, scale, shift, None, False, None, None, 1)
        mag = max(1.0, np.abs(ref).max())
        if n_split:
            refs = (np.maximum(ref[..., :n_split], 0) if relu else ref[..., :n_split],
                    np.maximum(ref[..., n_split:], 0) if not relu else ref[..., n_split:])
        else:
            refs = (np.maximum(ref, 0) if relu else ref,)
        for tile, out in outs.items():
            for part, (o, r) in enumerate(zip(out, refs)):
                assert o.shape == r.shape
                assert np.abs(o - r).max() < 2e-5 * mag, "%s tile %d part %d" % (name, tile, part)
                assert np.array_equal(o, outs[0][part]), "%s: tile %d differs from the library's choice" % (name, tile)
        other = conv_gemm(x, w, tile=0, k_order=0, **kw)
        for o, r in zip(outs[0], other):
            if r is not None:
                assert np.abs(o - r).max() < 2e-5 * mag


def test_conv1x1_stream_kernel_saturation_flag(gpu_device):
    """Values beyond the fp16 range are clamped where they are split and raise the sticky run flag, as in every split kernel."""
    from human_dynamics_amd.engine import conv_gemm
    lib = L.load()
    rng = np.random.default_rng(3)
    x = rng.normal(size=(1, 7, 7, 128)).astype(np.float32)
    w = (rng.normal(size=(1, 1, 128, 128)) / np.sqrt(128)).astype(np.float32)
    import ctypes as C
    fl = C.c_uint(0)
    L.check(lib.hmmr_run_flags(C.byref(fl), 1), "hmmr_run_flags")
    kw = dict(stride=1, pad=0, shift=np.zeros(128, np.float32), in_dtype=X3, out_dtype=X3, device=gpu_device, k_order=2)
    out, _ = conv_gemm(x, w, scale=np.ones(128, np.float32), **kw)
    L.check(lib.hmmr_run_flags(C.byref(fl), 1), "hmmr_run_flags")
    assert fl.value == 0 and np.abs(out).max() < 100
    out, _ = conv_gemm(x, w, scale=np.full(128, 1e6, np.float32), **kw)
    L.check(lib.hmmr_run_flags(C.byref(fl), 1), "hmmr_run_flags")
    assert fl.value & 1 and np.abs(out).max() == 65504.0
    # round 6: a NaN (in a constant, or in an operand) is not a number the clamp may hide: FLAG_NAN | FLAG_SATURATED, in both epilogue forms
    sc = np.ones(128, np.float32); sc[5] = np.nan
    out, _ = conv_gemm(x, w, scale=sc, **kw)
    L.check(lib.hmmr_run_flags(C.byref(fl), 1), "hmmr_run_flags")
    assert fl.value == (L.FLAG_NAN | L.FLAG_SATURATED), fl.value
    xn = x.copy(); xn[0, 3, 3, 17] = np.nan
    for extra in (dict(), dict(res=np.zeros((1, 7, 7, 128), np.float32), relu=False)):
        k2 = dict(kw); k2.update(extra)
        out, _ = conv_gemm(xn, w, scale=np.ones(128, np.float32), **k2)
        L.check(lib.hmmr_run_flags(C.byref(fl), 1), "hmmr_run_flags")
        assert fl.value == (L.FLAG_NAN | L.FLAG_SATURATED), (extra.keys(), fl.value)
    out, _ = conv_gemm(x, w, scale=np.ones(128, np.float32), **kw)
    L.check(lib.hmmr_run_flags(C.byref(fl), 1), "hmmr_run_flags")
    assert fl.value == 0


def test_conv1x1_stream_kernel_refuses_what_it_is_not_built_for(gpu_device):
    from human_dynamics_amd.engine import conv_gemm
    rng = np.random.default_rng(0)
    x = rng.normal(size=(2, 7, 7, 256)).astype(np.float32)
    w = rng.normal(size=(1, 1, 256, 256)).astype(np.float32)
    ok = dict(stride=1, pad=0, shift=np.zeros(256, np.float32), in_dtype=X3, out_dtype=X3, device=gpu_device, k_order=2)
    conv_gemm(x, w, **ok)
    for bad in (dict(stride=2), dict(tile=5), dict(tile=12), dict(pro=(np.ones(256, np.float32), np.zeros(256, np.float32))), dict(n_split=64),
                dict(res=np.zeros((2, 7, 7, 256), np.float32), tile=23), dict(res=np.zeros((2, 7, 7, 256), np.float32), n_split=128)):
        with pytest.raises(L.HmmrError):
            conv_gemm(x, w, **dict(ok, **bad))
    conv_gemm(x[..., :64], w[:, :, :64], **ok)                   # 4 K steps: the library's choice is the tile with rings 4 deep
    with pytest.raises(L.HmmrError, match="at least"):          # ... which a tile with 6-deep rings refuses
        conv_gemm(x[..., :64], w[:, :, :64], **dict(ok, tile=22))
    with pytest.raises(L.HmmrError, match="at least"):          # fewer K steps than any ring is deep
        conv_gemm(x[..., :32], w[:, :, :32], **ok)


C3_CASES = [
    # name, n, h, w, cin, cout, cin2, res, out2
    ("b4_conv3_res_out2", 11, 7, 7, 512, 2048, 0, True, True),       # block4/unit_2: shortcut add + the next unit's pre-activation
    ("b4_conv3_res", 5, 7, 7, 512, 2048, 0, True, False),            # block4/unit_3 (the last unit: no second output)
    ("b4_conv3_in2_out2", 6, 7, 7, 512, 2048, 1024, False, True),    # block4/unit_1: the shortcut folded into K ({h2, preact})
    ("conv3_out2_only", 3, 14, 14, 256, 256, 0, False, True),
    ("conv3_in2_only_ragged", 1, 5, 7, 128, 128, 64, False, False),
]


@pytest.mark.parametrize("case", C3_CASES, ids=[c[0] for c in C3_CASES])
def test_conv1x1_stream_kernel_conv3_form(case, gpu_device):
    """The conv3 form (hmmr_conv_desc_t.res / out2 / in2 with k_order 2 on a 1x1 filter): bias, + shortcut, split -> out; ReLU(BN(stored
    value)) -> out2; a second operand source along K.  Against float64 on the same 16-bit operands and against the 8-wave tiles; the two
    tile shapes agree bit for bit."""
    from human_dynamics_amd.engine import conv_gemm
    name, n, h, w_, cin, cout, cin2, has_res, has_out2 = case
    rng = np.random.default_rng(zlib.crc32(name.encode()) % 1000)
    x = rng.normal(size=(n, h, w_, cin)).astype(np.float32)
    w = (rng.normal(size=(1, 1, cin, cout)) / np.sqrt(cin + cin2)).astype(np.float32)
    shift = rng.normal(size=cout).astype(np.float32)
    kw = dict(stride=1, pad=0, shift=shift, in_dtype=X3, out_dtype=X3, device=gpu_device)
    xr, wr = _split_round(x), w
    if cin2:
        x2 = rng.normal(size=(n, h, w_, cin2)).astype(np.float32)
        w2 = (rng.normal(size=(1, 1, cin2, cout)) / np.sqrt(cin + cin2)).astype(np.float32)
        kw["second"] = (x2, w2)
        xr, wr = np.concatenate([xr, _split_round(x2)], axis=3), np.concatenate([w, w2], axis=2)
    res = None
    if has_res:
        res = (3.0 * rng.normal(size=(n, h, w_, cout))).astype(np.float32)
        kw["res"] = res
    s2 = b2 = None
    if has_out2:
        s2, b2 = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.normal(size=cout).astype(np.float32)
        kw.update(scale2=s2, shift2=b2)
    outs = {tile: conv_gemm(x, w, tile=tile, k_order=2, **kw) for tile in (0, 24, 25, 26)}
    ref, ref2 = _ref_conv(xr, _split_round_w(wr), 1, 0, None, shift, None if res is None else _split_round(res), False, s2, b2, 1)
    mag = max(1.0, np.abs(ref).max())
    other = conv_gemm(x, w, tile=0, k_order=0, **kw)
    for tile, (o, o2) in outs.items():
        assert np.abs(o - ref).max() < 2e-5 * mag, "%s tile %d" % (name, tile)
        assert np.array_equal(o, outs[0][0])
        assert np.abs(o - other[0]).max() < 2e-5 * mag
        if has_out2:
            assert np.abs(o2 - ref2).max() < 4e-5 * mag, "%s tile %d out2" % (name, tile)
            assert np.array_equal(o2, outs[0][1])
            assert np.abs(o2 - other[1]).max() < 4e-5 * mag
        else:
            assert o2 is None
    for bad in (dict(tile=22), dict(n_split=64), dict(res_stride=2, res=np.zeros((n, 2 * h, 2 * w_, cout), np.float32))):
        with pytest.raises(L.HmmrError):
            conv_gemm(x, w, k_order=2, **dict(kw, **bad))


def test_conv1x1_stream_kernel_above_4_gb(gpu_device):
    """Offsets inside the kernel are relative to a tile's first pixel: a tensor above 4 GB (540 000 pixels x 2048 channels x 4 bytes) gives
    its last pixels the bits a launch over those pixels alone gives them (conv1 form and conv3 form)."""
    import torch
    from human_dynamics_amd import packing
    from human_dynamics_amd.engine import conv_gemm
    g = torch.Generator(device=gpu_device).manual_seed(5)
    h, w_, cin, cout, tail = 540, 1000, 2048, 128, 700
    x = packing.to_split(torch.randn((1, h, w_, cin), device=gpu_device, generator=g))
    assert x.numel() * 4 > (1 << 32)
    rng = np.random.default_rng(5)
    w = (rng.normal(size=(1, 1, cin, cout)) / np.sqrt(cin)).astype(np.float32)
    shift = rng.normal(size=cout).astype(np.float32)
    xt = x.reshape(1, 1, h * w_, cin)[:, :, -tail:].contiguous()
    for kw in (dict(relu=True), dict(scale2=np.ones(cout, np.float32), shift2=shift)):
        kw = dict(kw, stride=1, pad=0, shift=shift, in_dtype=X3, out_dtype=X3, device=gpu_device, k_order=2, raw=True)
        big, big2 = conv_gemm(x, w, **kw)
        small, small2 = conv_gemm(xt, w, **kw)
        assert torch.equal(big.reshape(-1, cout)[-tail:], small.reshape(-1, cout))
        if big2 is not None:
            assert torch.equal(big2.reshape(-1, cout)[-tail:], small2.reshape(-1, cout))
    del x
    torch.cuda.empty_cache()


def test_resnet_with_the_1x1_stream_kernel(gpu_device):
    """The default f16x3 schedule runs 8 launches per pass through csrc/conv1x1_stream.hip (block 4's conv1 and conv3, block2/unit_1's conv1,
    block3/unit_1's shortcut + conv1: counted); stream_1x1=False keeps them on the 8-wave tiles.  The two differ by fp32 accumulation
    rounding (another K-step width) and nothing else; the fused schedule equals the layer-per-launch schedule of the same packing bit for
    bit; batch composition does not move a frame's bits."""
    import torch
    from human_dynamics_amd import assets
    from human_dynamics_amd.engine import HmmrEngine
    weights = assets.make_synthetic_weights(0)
    frames = assets.make_synthetic_frames(9, seed=41)
    on = HmmrEngine(weights, None, dtype="f16x3", device=gpu_device, autotune=False)
    off = HmmrEngine(weights, None, dtype="f16x3", device=gpu_device, autotune=False, stream_1x1=False)
    assert [int(on.rw.unit[i].conv1.k_order) for i in (3, 13, 14, 15)] == [2, 2, 2, 2] and int(on.rw.unit[7].sc_c1.k_order) == 2
    assert [int(on.rw.unit[i].conv3.k_order) for i in (13, 14, 15)] == [2, 2, 2] and int(on.rw.unit[13].c3sc.k_order) == 2
    assert all(int(off.rw.unit[i].conv1.k_order) == 0 and int(off.rw.unit[i].conv3.k_order) == 0 for i in range(16))
    L.launch_counts(clear=True)
    a = on.resnet(frames, n_zero=1)
    torch.cuda.synchronize()
    assert L.launch_counts(clear=True)["conv1x1_stream"] == 8
    b = off.resnet(frames, n_zero=1)
    torch.cuda.synchronize()
    assert L.launch_counts(clear=True)["conv1x1_stream"] == 0
    assert float(a.abs().max()) > 0.1 and float((a - b).abs().max()) < 2e-4 * float(a.abs().max())
    plain = HmmrEngine(weights, None, dtype="f16x3", device=gpu_device, autotune=False, fuse_tail=False)
    assert torch.equal(a, plain.resnet(frames, n_zero=1))
    # a frame's features do not depend on what else is in the batch (tiles are cut by pixel count, K order is fixed per output channel)
    alone = on.resnet(frames[3:5])
    assert torch.equal(a[3:5], alone)
