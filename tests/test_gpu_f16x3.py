"""GPU parity of the split-fp16 ("f16x3", HMMR_F16X3) mode: the throughput mode that has to stay
inside the reference tolerance (vertices / joints within 1e-4 of the fp32 TF graph,
BASELINE.json north_star; the reference computes in fp32 throughout, tester.py:64-66).

Operands are fp16 hi/lo pairs (x ~ hi + lo, 22 mantissa bits; filters scaled per output channel), every product is three fp16 MFMAs
(hi*hi + hi*lo + lo*hi) accumulated in fp32, activations between layers stay hi/lo pairs.
"""
import numpy as np
import pytest
import torch

from conftest import Config
from human_dynamics_amd import _lib as L
from human_dynamics_amd import assets
from test_gpu_kernels import _ref_conv

pytestmark = pytest.mark.gpu
F64 = torch.float64
X3 = L.HMMR_F16X3


def _split_round(a):
    """The value a split tensor holds for fp32 input a (hi + lo, fp16 halves, clamped to the fp16 range)."""
    t = torch.tensor(np.asarray(a, np.float32)).clamp(-65504.0, 65504.0)
    hi = t.to(torch.float16).to(torch.float32)
    return (hi.to(F64) + (t - hi).to(torch.float16).to(F64)).numpy()


def _split_round_w(w_hwio):
    """... and a filter bank [kh,kw,cin,cout]: every output channel scaled by its power of two before the split
    (packing.row_pow2) and unscaled after it."""
    from human_dynamics_amd import packing
    w = np.asarray(w_hwio, np.float32)
    rows = w.reshape(-1, w.shape[-1]).T                       # [cout][K]
    f = np.exp2(packing.row_pow2(rows).astype(np.float64))
    return (_split_round((rows * f[:, None]).astype(np.float32)) / f[:, None]).T.reshape(w.shape)


def test_split_layout_round_trip():
    """to_split / from_split: 22 mantissa bits (values whose lo half is a normal fp16; an absolute 6e-8 below that),
    32-byte groups [hi x8][lo x8], values beyond the fp16 range clamped."""
    from human_dynamics_amd.packing import from_split, to_split
    x = torch.randn(5, 7, 64, generator=torch.Generator().manual_seed(0)) * 3
    x[0, 0, :4] = torch.tensor([1e5, -3e6, 65504.0, 1e-7])
    s = to_split(x)
    assert s.dtype == torch.int32 and s.shape == x.shape
    y = from_split(s)
    xc = x.clamp(-65504.0, 65504.0)
    assert bool(((xc - y).abs() <= torch.maximum(xc.abs() * 2.0 ** -21, torch.tensor(6.0e-8))).all())
    raw = s.view(torch.float16).reshape(5, 7, 8, 2, 8)
    assert torch.equal(raw[..., 0, :].reshape(5, 7, 64), xc.to(torch.float16))


CASES = [
    # name, n, h, w, cin, cout, k, stride, pad, flags
    ("1x1_64_64", 2, 12, 12, 64, 64, 1, 1, 0, ""),
    ("1x1_256_64_bnrelu", 1, 28, 28, 256, 64, 1, 1, 0, "sbr"),
    ("1x1_64_256_res_out2", 2, 14, 14, 64, 256, 1, 1, 0, "bR2"),
    ("3x3_s1", 2, 14, 14, 64, 64, 3, 1, 1, "sbr"),
    ("3x3_s2", 2, 14, 14, 128, 128, 3, 2, 1, "sbr"),
    ("3x3_s1_odd", 1, 7, 7, 512, 512, 3, 1, 1, "sbr"),
    ("4x4_cin16", 2, 9, 9, 16, 64, 4, 1, 1, "b"),          # two taps per 128-byte K step
    ("fc_ragged_85", 37, 1, 1, 1024, 85, 1, 1, 0, "b"),
    ("fc_2048_1024", 37, 1, 1, 2048, 1024, 1, 1, 0, "br"),
    ("1x1_strided_res", 1, 14, 14, 64, 256, 1, 1, 0, "bS"),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 5, 6, 7, 8])
@pytest.mark.parametrize("out_dt", ["f32", "x3"])
def test_conv_gemm_split(case, tile, out_dt, gpu_device):
    """Against a float64 convolution of the SAME 16-bit operands: what is left is the dropped lo*lo
    term (2^-18 per product) and fp32 accumulation; and against the unrounded fp32 operands with the
    tolerance the mode promises."""
    from human_dynamics_amd.engine import conv_gemm
    name, n, h, w_, cin, cout, k, stride, pad, flags = case
    if (tile in (1, 5, 7) and cout % 128) or (tile == 8 and cout % 256):
        pytest.skip("128- / 256-wide tiles are only selected for cout % 128 / 256 == 0")
    if out_dt == "x3" and cout % 8:
        pytest.skip("split outputs are whole 8-channel groups")
    import zlib
    rng = np.random.default_rng(zlib.crc32(name.encode()) % 1000)
    x = rng.normal(size=(n, h, w_, cin)).astype(np.float32)
    w = (rng.normal(size=(k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
    ho = (h + 2 * pad - k) // stride + 1
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32) if "s" in flags else None
    shift = rng.normal(size=cout).astype(np.float32) if "b" in flags else None
    res, res_stride = None, 1
    if "R" in flags:
        res = rng.normal(size=(n, ho, ho, cout)).astype(np.float32)
    if "S" in flags:
        res = rng.normal(size=(n, 2 * ho, 2 * ho, cout)).astype(np.float32)
        res_stride = 2
    s2 = rng.uniform(0.5, 1.5, cout).astype(np.float32) if "2" in flags else None
    b2 = rng.normal(size=cout).astype(np.float32) if "2" in flags else None
    odt = X3 if out_dt == "x3" else L.HMMR_F32
    out, out2 = conv_gemm(x, w, stride, pad, scale, shift, res, "r" in flags, s2, b2, in_dtype=X3, out_dtype=odt,
                          tile=tile, device=gpu_device, res_stride=res_stride)
    res_r = res if (res is None or out_dt == "f32") else _split_round(res)     # the residual is read in the output type
    ref, ref2 = _ref_conv(_split_round(x), _split_round_w(w), stride, pad, scale, shift, res_r, "r" in flags, s2, b2,
                          res_stride)
    mag = max(1.0, np.abs(ref).max())
    err = np.abs(out - ref).max()
    assert err < 2e-5 * mag, "%s tile %d: max abs err %.3e vs same-operand f64" % (name, tile, err)
    if ref2 is not None:
        assert np.abs(out2 - ref2).max() < 3e-5 * max(1.0, np.abs(ref2).max())
    exact, _ = _ref_conv(x, w, stride, pad, scale, shift, res, "r" in flags, None, None, res_stride)
    assert np.abs(out - exact).max() < 6e-5 * mag          # vs the unrounded operands: 2^-17-sized inputs


PATCH_CASES = [
    # name, n, h, w, cin, cout: 3x3 / stride 1 / SAME through the patch kernel (hmmr_conv_desc_t.k_order = 1)
    ("b3_14x14", 5, 14, 14, 256, 256),          # a 128-pixel tile straddles two images
    ("b4_7x7", 9, 7, 7, 512, 512),              # ... four images
    ("b2_28x28", 3, 28, 28, 128, 128),          # 128 output columns: tile 9 only
    ("odd_5x9", 4, 5, 9, 64, 256),              # non-square image, two channel chunks
    ("one_image", 1, 14, 14, 256, 256),         # M tail inside the second tile
    ("tiny_3x3", 2, 3, 3, 32, 256),             # every pixel is a border pixel, one chunk
    ("ragged_384", 7, 7, 7, 128, 384),          # three column tiles of 128
    ("wide_28x28", 2, 28, 28, 128, 256),
]


@pytest.mark.parametrize("case", PATCH_CASES, ids=[c[0] for c in PATCH_CASES])
def test_conv3x3_patch_kernel(case, gpu_device):
    """The 3x3 patch kernel (tiles 9 / 10: the A operand out of an LDS-resident input patch, K chunk-major, out-of-image taps read a zero row)
    against a float64 convolution of the same 16-bit operands and against the im2col ring tiles: the same products,
    another accumulation order.  Tiles 9 and 10 agree bit for bit (one fixed-order K reduction per output element)."""
    from human_dynamics_amd.engine import conv_gemm
    name, n, h, w_, cin, cout = case
    import zlib
    rng = np.random.default_rng(zlib.crc32(name.encode()) % 1000)
    x = rng.normal(size=(n, h, w_, cin)).astype(np.float32)
    w = (rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.normal(size=cout).astype(np.float32)
    kw = dict(stride=1, pad=1, scale=scale, shift=shift, relu=True, in_dtype=X3, out_dtype=X3, device=gpu_device)
    outs = {}
    for tile in (0, 9, 10, 11):
        if tile == 10 and cout % 256:
            continue
        outs[tile], _ = conv_gemm(x, w, tile=tile, k_order=1, **kw)
    ref, _ = _ref_conv(_split_round(x), _split_round_w(w), 1, 1, scale, shift, None, True, None, None, 1)
    mag = max(1.0, np.abs(ref).max())
    for tile, out in outs.items():
        assert np.abs(out - ref).max() < 2e-5 * mag, "%s tile %d" % (name, tile)
        assert np.array_equal(out, outs[0]), "%s: tile %d differs from the library's choice" % (name, tile)
    ring, _ = conv_gemm(x, w, tile=0, **kw)
    assert np.abs(outs[0] - ring).max() < 2e-5 * mag


STREAM_CASES = PATCH_CASES + [
    ("b1_56x56", 2, 56, 56, 64, 64),            # 64-channel tiles (19 / 20), the widest image
    ("b1_one_5x7", 1, 5, 7, 64, 64),
]


@pytest.mark.parametrize("case", STREAM_CASES, ids=[c[0] for c in STREAM_CASES])
def test_conv3x3_stream_kernel(case, gpu_device):
    """The one-wave-per-SIMD 3x3 kernel (hmmr_conv_desc_t.k_order = 2, csrc/conv3x3_stream.hip: filters as a fragment stream, pixels out
    of an LDS patch in 16-channel chunks, SAME padding as an address select) against a float64 convolution of the same 16-bit operands
    and against the patch kernel (another accumulation order).  Its seven tile shapes agree bit for bit; without ReLU too."""
    from human_dynamics_amd.engine import conv_gemm
    name, n, h, w_, cin, cout = case
    import zlib
    rng = np.random.default_rng(zlib.crc32(name.encode()) % 1000)
    x = rng.normal(size=(n, h, w_, cin)).astype(np.float32)
    w = (rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.normal(size=cout).astype(np.float32)
    for relu in (True, False):
        kw = dict(stride=1, pad=1, scale=scale, shift=shift, relu=relu, in_dtype=X3, out_dtype=X3, device=gpu_device)
        tiles = ((0, 19, 20) if relu else (0, 20)) if cout == 64 else ((0, 12, 13, 14, 15, 16, 17, 18, 21, 27, 28) if relu else (0, 13, 27))
        outs = {tile: conv_gemm(x, w, tile=tile, k_order=2, **kw)[0] for tile in tiles}
        ref, _ = _ref_conv(_split_round(x), _split_round_w(w), 1, 1, scale, shift, None, relu, None, None, 1)
        mag = max(1.0, np.abs(ref).max())
        for tile, out in outs.items():
            assert np.abs(out - ref).max() < 2e-5 * mag, "%s tile %d" % (name, tile)
            assert np.array_equal(out, outs[0]), "%s: tile %d differs from the library's choice" % (name, tile)
        other, _ = conv_gemm(x, w, tile=0, k_order=0 if cout == 64 else 1, **kw)
        assert np.abs(outs[0] - other).max() < 2e-5 * mag


def test_conv3x3_stream_kernel_refuses_what_it_is_not_built_for(gpu_device):
    from human_dynamics_amd.engine import conv_gemm
    rng = np.random.default_rng(0)
    x = rng.normal(size=(2, 14, 14, 64)).astype(np.float32)
    w = rng.normal(size=(3, 3, 64, 256)).astype(np.float32)
    for kw in (dict(stride=2, pad=1), dict(stride=1, pad=0), dict(stride=1, pad=1, tile=9), dict(stride=1, pad=1, tile=5),
               dict(stride=1, pad=1, res=np.zeros((2, 14, 14, 256), np.float32))):
        with pytest.raises(L.HmmrError):
            conv_gemm(x, w, in_dtype=X3, out_dtype=X3, device=gpu_device, k_order=2, **kw)
    with pytest.raises(L.HmmrError, match="28 pixels"):        # 128-channel tiles: images up to 28 pixels wide
        conv_gemm(rng.normal(size=(1, 4, 56, 64)).astype(np.float32), w, stride=1, pad=1, in_dtype=X3, out_dtype=X3, device=gpu_device, k_order=2)
    with pytest.raises(L.HmmrError):                           # a 64-channel layer on a 128-channel tile
        conv_gemm(x, w[..., :64], stride=1, pad=1, in_dtype=X3, out_dtype=X3, device=gpu_device, k_order=2, tile=12)


def test_conv3x3_patch_kernel_refuses_what_it_is_not_built_for(gpu_device):
    from human_dynamics_amd.engine import conv_gemm
    rng = np.random.default_rng(0)
    x = rng.normal(size=(2, 14, 14, 64)).astype(np.float32)
    w = rng.normal(size=(3, 3, 64, 256)).astype(np.float32)
    for kw in (dict(stride=2, pad=1), dict(stride=1, pad=0), dict(stride=1, pad=1, tile=8),
               dict(stride=1, pad=1, res=np.zeros((2, 14, 14, 256), np.float32))):
        with pytest.raises(L.HmmrError):
            conv_gemm(x, w, in_dtype=X3, out_dtype=X3, device=gpu_device, k_order=1, **kw)
    with pytest.raises(L.HmmrError, match="more than LDS holds"):     # the patch = the tile's pixels + a halo of W + 1 on either side
        conv_gemm(rng.normal(size=(1, 4, 112, 32)).astype(np.float32), rng.normal(size=(3, 3, 32, 256)).astype(np.float32),
                  stride=1, pad=1, in_dtype=X3, out_dtype=X3, device=gpu_device, k_order=1)
    with pytest.raises(L.HmmrError):           # a tile of the other K order
        conv_gemm(x, w, stride=1, pad=1, in_dtype=X3, out_dtype=X3, device=gpu_device, tile=9)
    with pytest.raises(L.HmmrError):           # built for split and bf16 tensors
        conv_gemm(x, w, stride=1, pad=1, in_dtype=L.HMMR_F32, out_dtype=L.HMMR_F32, device=gpu_device, k_order=1)
    with pytest.raises(L.HmmrError, match="tile 11"):          # the tile without a load segment: split operands only
        conv_gemm(x, w, stride=1, pad=1, in_dtype=L.HMMR_BF16, out_dtype=L.HMMR_BF16, device=gpu_device, k_order=1, tile=11)


@pytest.mark.parametrize("case", [c for c in PATCH_CASES if c[4] % 64 == 0], ids=[c[0] for c in PATCH_CASES if c[4] % 64 == 0])
def test_conv3x3_patch_kernel_bf16(case, gpu_device):
    """Round 4: the patch kernel for bf16 operands (64 channels per 128-byte chunk; blocks 3-4 of the bf16 mode): against a
    float64 convolution of the same bf16-rounded operands, tiles 9 / 10 bit-identical, the im2col ring tiles within the rounding
    of another accumulation order and of the bf16 output."""
    from human_dynamics_amd.engine import conv_gemm
    name, n, h, w_, cin, cout = case
    import zlib
    rng = np.random.default_rng(zlib.crc32(name.encode()) % 1000 + 1)
    bf = lambda a: torch.from_numpy(np.asarray(a, np.float32)).to(torch.bfloat16).float().numpy()
    x = bf(rng.normal(size=(n, h, w_, cin)))
    w = bf(rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin))
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.normal(size=cout).astype(np.float32)
    kw = dict(stride=1, pad=1, scale=scale, shift=shift, relu=True, in_dtype=L.HMMR_BF16, out_dtype=L.HMMR_F32 if False else L.HMMR_BF16, device=gpu_device)
    outs = {}
    for tile in (0, 9, 10):
        if tile == 10 and cout % 256:
            continue
        outs[tile], _ = conv_gemm(x, w, tile=tile, k_order=1, **kw)
    ref, _ = _ref_conv(x, w, 1, 1, scale, shift, None, True, None, None, 1)
    mag = max(1.0, np.abs(ref).max())
    for tile, out in outs.items():
        assert np.abs(out - ref).max() < 6e-3 * mag, "%s tile %d" % (name, tile)          # (the output is stored as bf16)
        assert np.array_equal(out, outs[0]), "%s: tile %d differs from the library's choice" % (name, tile)
    ring, _ = conv_gemm(x, w, tile=0, **kw)
    assert np.abs(outs[0] - ring).max() < 8e-3 * mag


@pytest.mark.parametrize("case", [c for c in PATCH_CASES if c[4] % 64 == 0], ids=[c[0] for c in PATCH_CASES if c[4] % 64 == 0])
def test_conv3x3_stream_kernel_bf16(case, gpu_device):
    """The stream kernel's bf16 instantiation (k_order 2 with bf16 tensors: K steps of 32 channels, two MFMAs per accumulator and step):
    against a float64 convolution of the same bf16-rounded operands; its tiles agree bit for bit; the patch kernel (another accumulation
    order) within the rounding of the bf16 output."""
    from human_dynamics_amd.engine import conv_gemm
    name, n, h, w_, cin, cout = case
    import zlib
    rng = np.random.default_rng(zlib.crc32(name.encode()) % 1000 + 2)
    bf = lambda a: torch.from_numpy(np.asarray(a, np.float32)).to(torch.bfloat16).float().numpy()
    x = bf(rng.normal(size=(n, h, w_, cin)))
    w = bf(rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin))
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.normal(size=cout).astype(np.float32)
    for relu in (True, False):
        kw = dict(stride=1, pad=1, scale=scale, shift=shift, relu=relu, in_dtype=L.HMMR_BF16, out_dtype=L.HMMR_BF16, device=gpu_device)
        outs = {tile: conv_gemm(x, w, tile=tile, k_order=2, **kw)[0] for tile in ((0, 12, 13, 14, 15, 16, 17, 18) if relu else (0, 13))}
        ref, _ = _ref_conv(x, w, 1, 1, scale, shift, None, relu, None, None, 1)
        mag = max(1.0, np.abs(ref).max())
        for tile, out in outs.items():
            assert np.abs(out - ref).max() < 6e-3 * mag, "%s tile %d" % (name, tile)          # (the output is stored as bf16)
            assert np.array_equal(out, outs[0]), "%s: tile %d differs from the library's choice" % (name, tile)
        patch, _ = conv_gemm(x, w, tile=0, k_order=1, **kw)
        assert np.abs(outs[0] - patch).max() < 8e-3 * mag
    with pytest.raises(L.HmmrError, match="tile 21"):          # the 7 x 1 wave tile: split tensors only
        conv_gemm(x, w, tile=21, k_order=2, stride=1, pad=1, in_dtype=L.HMMR_BF16, out_dtype=L.HMMR_BF16, device=gpu_device)


@pytest.mark.parametrize("tile", [0, 3, 5, 6, 1, 7])
def test_conv_gemm_split_fused_preactivation(tile, gpu_device):
    """A = relu(x*scale[ci] + shift[ci]) applied while staging a split operand (hi and lo halves sit in
    neighbouring lanes): equal, bit for bit, to pre-activating on the producer side (`out2`)."""
    from human_dynamics_amd.engine import conv_gemm
    rng = np.random.default_rng(7)
    x = rng.normal(size=(2, 14, 14, 256)).astype(np.float32)
    w = (rng.normal(size=(1, 1, 256, 128)) / 16).astype(np.float32)
    ps = rng.uniform(0.5, 1.5, 256).astype(np.float32)
    pb = rng.normal(size=256).astype(np.float32)
    s = rng.uniform(0.5, 1.5, 128).astype(np.float32)
    b = rng.normal(size=128).astype(np.float32)
    out, _ = conv_gemm(x, w, 1, 0, s, b, None, True, in_dtype=X3, tile=tile, device=gpu_device, pro=(ps, pb))
    xa = _split_round(np.maximum(_split_round(x) * ps.astype(np.float64) + pb, 0).astype(np.float32))
    ref, _ = _ref_conv(xa, _split_round_w(w), 1, 0, s, b, None, True, None, None)
    assert np.abs(out - ref).max() < 3e-5 * max(1.0, np.abs(ref).max())
    # producer-side route: an identity 1x1 conv writes relu(x*ps+pb) as its second output, the GEMM reads that
    eye = np.eye(256, dtype=np.float32).reshape(1, 1, 256, 256)
    _, pre = conv_gemm(x, eye, 1, 0, None, None, None, False, ps, pb, in_dtype=X3, out_dtype=X3, device=gpu_device,
                       raw=True)               # the split words themselves: re-splitting hi + lo may pick another pair
    out_b, _ = conv_gemm(pre, w, 1, 0, s, b, None, True, in_dtype=X3, tile=tile, device=gpu_device)
    assert np.array_equal(out, out_b)


@pytest.mark.parametrize("split_k", [2, 4])
def test_conv_gemm_split_with_split_k(split_k, gpu_device):
    from human_dynamics_amd.engine import conv_gemm
    rng = np.random.default_rng(split_k)
    x = rng.normal(size=(3, 20, 1, 256)).astype(np.float32)
    w = (rng.normal(size=(3, 1, 256, 200)) / np.sqrt(768)).astype(np.float32)
    b = rng.normal(size=200).astype(np.float32)
    res = rng.normal(size=(3, 20, 1, 200)).astype(np.float32)
    out, _ = conv_gemm(x, w, 1, (1, 0), None, b, res, True, in_dtype=X3, device=gpu_device, split_k=split_k)
    ref, _ = _ref_conv(_split_round(x), _split_round_w(w), 1, (1, 0), None, b, res, True, None, None)
    assert np.abs(out - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())
    x2 = np.concatenate([x, rng.normal(size=(5, 20, 1, 256)).astype(np.float32)])
    res2 = np.concatenate([res, rng.normal(size=(5, 20, 1, 200)).astype(np.float32)])
    out2, _ = conv_gemm(x2, w, 1, (1, 0), None, b, res2, True, in_dtype=X3, device=gpu_device, split_k=split_k)
    assert np.array_equal(out2[:3], out)                    # batch independence, bit for bit


def test_conv_gemm_f32_in_split_out(gpu_device):
    """The IEF theta GEMM: fp32 operand, split output + split residual."""
    from human_dynamics_amd.engine import conv_gemm
    rng = np.random.default_rng(3)
    x = rng.normal(size=(37, 1, 1, 128)).astype(np.float32)
    w = (rng.normal(size=(1, 1, 128, 1024)) / 12).astype(np.float32)
    res = rng.normal(size=(37, 1, 1, 1024)).astype(np.float32)
    out, _ = conv_gemm(x, w, 1, 0, None, None, res, True, in_dtype=L.HMMR_F32, out_dtype=X3, device=gpu_device)
    ref, _ = _ref_conv(x, w, 1, 0, None, None, _split_round(res), True, None, None)
    assert np.abs(out - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())


@pytest.fixture(scope="module")
def eng_x3(weights, smpl_consts, gpu_device):
    from human_dynamics_amd.engine import HmmrEngine
    return HmmrEngine(weights, smpl_consts, dtype="f16x3", device=gpu_device)


def test_groupnorm_relu_split_output(eng_x3):
    from human_dynamics_amd.packing import from_split
    from oracle import hmmr_oracle as O
    rng = np.random.default_rng(0)
    x = (rng.normal(size=(3, 20, 2048)) * 2 + 0.5).astype(np.float32)
    g = rng.uniform(0.5, 1.5, 2048).astype(np.float32)
    b = rng.normal(size=2048).astype(np.float32)
    out = from_split(eng_x3.groupnorm_relu(x, g, b, out_dtype=X3)).cpu().numpy()
    ref = torch.relu(O.group_norm_time(torch.tensor(x, dtype=F64), torch.tensor(g, dtype=F64),
                                       torch.tensor(b, dtype=F64))).numpy()
    assert np.abs(out - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())


def test_resnet_split_matches_oracle(eng_x3, golden_window):
    """BASELINE config-2-shaped check at 3 frames (the batch-64 version is test_gpu_sizes.py)."""
    frames = assets.make_synthetic_frames(3, seed=1)
    phi = eng_x3.resnet(frames).cpu().numpy()
    ref = golden_window["phi"][:3]
    err = np.abs(phi - ref).max()
    rel = np.linalg.norm(phi - ref) / np.linalg.norm(ref)
    print("ResNet f16x3: phi max-abs-err %.3e rel-L2 %.3e (|phi|max %.2f)" % (err, rel, np.abs(ref).max()))
    assert rel < 1e-4 and err < 1e-3


def test_resnet_split_batch_independence(eng_x3):
    frames = assets.make_synthetic_frames(5, seed=9)
    frames[2] = 0.0
    a = eng_x3.resnet(frames).cpu().numpy()
    b = eng_x3.resnet(frames[2:3]).cpu().numpy()
    c = eng_x3.resnet(frames[:2], n_zero=1).cpu().numpy()
    assert np.array_equal(a[2:3], b) and np.array_equal(a[:3], c)


def test_temporal_and_ief_split_match_oracle(eng_x3, golden_window):
    phi = golden_window["phi"].reshape(1, 20, 2048)
    out = eng_x3.temporal(phi).cpu().numpy()
    e1 = np.abs(out[0] - golden_window["strips"]).max()
    om = eng_x3.ief(golden_window["strips"]).cpu().numpy()
    e2 = np.abs(om - golden_window["omegas_all"]).max()
    print("f16x3: strips max-abs-err %.3e, omegas max-abs-err %.3e" % (e1, e2))
    assert e1 < 2e-4 and e2 < 5e-5


def test_predict_split_meets_reference_tolerance(weights, smpl_consts, gpu_device, golden_window):
    """BASELINE config 1 in the throughput mode: vertices and joints within 1e-4 of the float64
    reference-graph oracle (north_star)."""
    from human_dynamics_amd.evaluation.tester import Tester
    from test_gpu_pipeline import _check
    frames = assets.make_synthetic_frames(20, seed=1)
    t = Tester(Config(batch_size=1), weights=weights, smpl=smpl_consts, device=gpu_device)     # the default dtype
    assert t.engine.dtype == X3
    res = t.predict(frames[None])
    errs = _check(res, golden_window, 1e-4)
    print("f16x3 end to end: verts %.3e joints %.3e omegas %.3e" % (errs["verts"], errs["joints"], errs["omegas"]))


def test_predict_all_images_split_matches_golden_video(weights, smpl_consts, gpu_device, golden_video):
    from human_dynamics_amd.evaluation.tester import Tester
    from test_gpu_pipeline import _check
    frames = assets.make_synthetic_frames(24, seed=7)
    t = Tester(Config(batch_size=2), weights=weights, smpl=smpl_consts, dtype="f16x3", device=gpu_device)
    res = t.predict_all_images(frames)
    _check(res, dict(golden_video), 1e-4)
    lit = Tester(Config(batch_size=2), weights=weights, smpl=smpl_consts, dtype="f16x3", device=gpu_device,
                 dedup=False).predict_all_images(frames)
    for k in res:
        assert np.array_equal(res[k], lit[k]), k


@pytest.mark.parametrize("tile", [0, 5, 6, 7, 3])
@pytest.mark.parametrize("dt", ["x3", "bf16", "f32"])
def test_conv_gemm_second_operand_source(tile, dt, gpu_device):
    """hmmr_conv_desc_t.in2: K = cin + cin2 over two tensors of the same pixel grid in ONE accumulator (a unit's conv3
    + its conv shortcut) against the two convolutions evaluated separately in float64 on the same (rounded) operands."""
    from human_dynamics_amd.engine import conv_gemm
    from human_dynamics_amd.packing import from_split, to_split
    rng = np.random.default_rng(11)
    n, h, c1, c2, cout = 3, 13, 64, 256, 256            # 507 pixels: M tail inside the last tile
    x1 = rng.normal(size=(n, h, h, c1)).astype(np.float32)
    x2 = np.maximum(rng.normal(size=(n, h, h, c2)), 0).astype(np.float32)
    w1 = (rng.normal(size=(1, 1, c1, cout)) / 8).astype(np.float32)
    w2 = (rng.normal(size=(1, 1, c2, cout)) / 16).astype(np.float32)
    b = rng.normal(size=cout).astype(np.float32)
    code = {"x3": X3, "bf16": L.HMMR_BF16, "f32": L.HMMR_F32}[dt]
    out, _ = conv_gemm(x1, w1, 1, 0, None, b, None, False, in_dtype=code, out_dtype=L.HMMR_F32, tile=tile,
                       device=gpu_device, second=(x2, w2))
    def rnd(a):
        t = torch.from_numpy(a)
        if dt == "x3":
            return from_split(to_split(t)).double().numpy()
        return (t.to(torch.bfloat16) if dt == "bf16" else t).double().numpy()
    ref = rnd(x1).reshape(-1, c1) @ rnd(w1).reshape(c1, cout) + rnd(x2).reshape(-1, c2) @ rnd(w2).reshape(c2, cout) + b
    err = np.abs(out.reshape(-1, cout) - ref).max()
    assert err < 2e-5 * max(1.0, np.abs(ref).max()), err


def test_resnet_folded_shortcut_is_the_separate_shortcut_up_to_its_rounding(weights, gpu_device):
    """f16x3 default: the conv shortcut of every block's first unit is accumulated inside conv3's GEMM instead of being
    stored (rounded to 16 bits) and added back.  Against the launch-per-layer schedule the features move by that one
    rounding; against the float64 oracle both stay inside the mode's bound."""
    from human_dynamics_amd.engine import HmmrEngine
    from oracle import hmmr_oracle as O
    frames = assets.make_synthetic_frames(5, seed=9)
    folded = HmmrEngine(weights, None, dtype="f16x3", device=gpu_device, autotune=False)
    assert [i for i in range(16) if folded.rw.unit[i].c3sc.w] == [0, 3, 13]       # (block3/unit_1 keeps its shortcut as a launch: its successor pair holds conv3's operand in registers)
    plain = HmmrEngine(weights, None, dtype="f16x3", device=gpu_device, autotune=False, fold_sc=False)
    assert not any(plain.rw.unit[i].c3sc.w for i in range(16))
    a, b = folded.resnet(frames, n_zero=1).cpu().numpy(), plain.resnet(frames, n_zero=1).cpu().numpy()
    ref = O.resnet_v2_50(np.concatenate([frames, np.zeros_like(frames[:1])]), weights, torch.float64).numpy()
    n = np.linalg.norm(ref)
    ea, eb, d = np.linalg.norm(a - ref) / n, np.linalg.norm(b - ref) / n, np.linalg.norm(a - b) / n
    print("folded shortcut: rel-L2 vs f64 %.2e (separate launches %.2e), folded vs separate %.2e" % (ea, eb, d))
    assert ea < 5e-5 and eb < 5e-5 and d < 2e-5
    emu = O.resnet_v2_50_emulated(frames[:2], weights, "f16x3").numpy()
    emu_sep = O.resnet_v2_50_emulated(frames[:2], weights, "f16x3", fold_shortcut=False).numpy()
    assert np.linalg.norm(a[:2] - emu) / np.linalg.norm(emu) < 5e-6
    assert np.linalg.norm(b[:2] - emu_sep) / np.linalg.norm(emu_sep) < 5e-6


def test_resnet_split_fused_tails_equal_layer_per_launch(weights, gpu_device):
    """csrc/bottleneck_split.hip: conv3 (+ folded shortcut / + identity shortcut) + the next unit's preact + conv1 as one
    launch for units 1.1, 1.2 (with their 3x3 conv2 in front as well), 2.2, 2.3 -- the same MFMA order and rounding points as the launches it replaces, so the
    features are bit-identical (6 images: block 2's 4704 pixels end in a half tile)."""
    from human_dynamics_amd.engine import HmmrEngine
    frames = assets.make_synthetic_frames(5, seed=17)
    fused = HmmrEngine(weights, None, dtype="f16x3", device=gpu_device, autotune=False)
    # units 1.1, 1.2: the block-1 tails with conv2 in front; 2.1-2.3 and 3.1-3.5: register-resident unit pairs (csrc/unit_pair.hip)
    assert [int(fused.rw.unit[i].fuse_tail) for i in range(16)] == [2, 2, 0, 1, 1, 1, 0, 1, 1, 1, 1, 1, 0, 0, 0, 0]
    assert [bool(fused.rw.unit[i].pair_stream) for i in range(16)] == [False] * 3 + [True] * 3 + [False] + [True] * 5 + [False] * 4
    plain = HmmrEngine(weights, None, dtype="f16x3", device=gpu_device, autotune=False, fuse_tail=False)
    assert not any(plain.rw.unit[i].fuse_tail for i in range(16))
    a, b = fused.resnet(frames, n_zero=1), plain.resnet(frames, n_zero=1)
    assert float(b.abs().max()) > 0.1
    assert torch.equal(a, b), float((a - b).abs().max())
    for variant in ("block1", "noconv2"):
        eng = HmmrEngine(weights, None, dtype="f16x3", device=gpu_device, autotune=False, fuse_tail=variant)
        assert torch.equal(eng.resnet(frames, n_zero=1), b), variant
    # the round-3 schedule (LDS-panel tails in block 2, layer per launch in block 3, block3/unit_1's shortcut folded into its
    # conv3) and the pairs one block at a time: each against the layer-per-launch schedule of the same folding decisions
    for variant in (False, "block2", "block3"):
        eng = HmmrEngine(weights, None, dtype="f16x3", device=gpu_device, autotune=False, unit_pair=variant)
        ref = HmmrEngine(weights, None, dtype="f16x3", device=gpu_device, autotune=False, unit_pair=variant, fuse_tail=False)
        assert torch.equal(eng.resnet(frames, n_zero=1), ref.resnet(frames, n_zero=1)), variant
