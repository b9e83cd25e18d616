"""GPU parity of the round-5 kernels' schedule switches (f16x3):

* csrc/b1_unit.hip, the whole-unit kernel of block 1 (conv2 + conv3 + add + the next unit's preact + conv1 as one launch), against
  the three hmmr_conv_gemm launches it replaces -- bit for bit, per kernel (both forms: identity shortcut, folded conv shortcut;
  ragged last tile; tiles that straddle images) and through the whole ResNet (slim resnet_v2.bottleneck as invoked at
  /root/reference/src/models.py:65-75);
* csrc/unit_pair.hip against the two launches IT replaces on ONE batch with the switch of hmmr_resnet50_fwd forced either way
  (hmmr_debug_t.pair_min_pixels), with the library's launch counters as the proof of which kernel ran.
"""
import numpy as np
import pytest
import torch

from human_dynamics_amd import _lib as L
from human_dynamics_amd import assets

pytestmark = pytest.mark.gpu
X3 = L.HMMR_F16X3


def _unit_inputs(n, h, w, seed, folded):
    rng = np.random.default_rng(seed)
    f32 = np.float32
    d = {}
    d["h1"] = np.maximum(rng.normal(size=(n, h, w, 64)), 0).astype(f32) * 2.0                  # a ReLU output
    d["w2"] = (rng.normal(size=(3, 3, 64, 64)) * 0.06).astype(f32)
    d["bn2"] = (rng.uniform(0.5, 1.5, 64).astype(f32), (rng.normal(size=64) * 0.3).astype(f32))
    d["w3"] = (rng.normal(size=(1, 1, 64, 256)) * 0.12).astype(f32)
    d["b3"] = (rng.normal(size=256) * 0.2).astype(f32)
    d["pre"] = (rng.uniform(0.5, 1.5, 256).astype(f32), (rng.normal(size=256) * 0.3).astype(f32))
    d["w1"] = (rng.normal(size=(1, 1, 256, 64)) * 0.06).astype(f32)
    d["bn1"] = (rng.uniform(0.5, 1.5, 64).astype(f32), (rng.normal(size=64) * 0.3).astype(f32))
    if folded:
        d["xp"] = np.maximum(rng.normal(size=(n, h, w, 64)), 0).astype(f32)
        d["wsc"] = (rng.normal(size=(1, 1, 64, 256)) * 0.12).astype(f32)
        d["bsc"] = (rng.normal(size=256) * 0.2).astype(f32)
    else:
        d["res"] = (rng.normal(size=(n, h, w, 256)) * 1.5).astype(f32)
    return d


@pytest.mark.parametrize("folded", [False, True], ids=["identity", "folded"])
@pytest.mark.parametrize("shape", [(3, 56, 56), (5, 24, 24), (1, 8, 8)], ids=["3x56x56", "5x24x24", "1x8x8"])
def test_b1_unit_kernel_equals_its_three_launches(shape, folded, gpu_device):
    """56 x 56: block 1's grid, 3 images = 73.5 tiles (the last one half empty); 24 x 24: every tile straddles images; 8 x 8: half a tile."""
    from human_dynamics_amd import engine as E
    from human_dynamics_amd import packing
    n, h, w = shape
    d = _unit_inputs(n, h, w, 11 + n, folded)
    kw = dict(in_dtype=X3, out_dtype=X3, device=gpu_device, raw=True)
    h1 = packing.to_split(torch.from_numpy(d["h1"]).to(gpu_device))
    L.launch_counts(clear=True)
    h2, _ = E.conv_gemm(h1, d["w2"], pad=1, scale=d["bn2"][0], shift=d["bn2"][1], relu=True, k_order=2, **kw)
    if folded:
        bias = (d["b3"].astype(np.float64) + d["bsc"].astype(np.float64)).astype(np.float32)
        trunk, _ = E.conv_gemm(h2, d["w3"], shift=bias, second=(d["xp"], d["wsc"]), **kw)
    else:
        trunk, _ = E.conv_gemm(h2, d["w3"], shift=d["b3"], res=d["res"], **kw)
    h1n, _ = E.conv_gemm(trunk, d["w1"], scale=d["bn1"][0], shift=d["bn1"][1], relu=True, pro=d["pre"], **kw)
    assert L.launch_counts()["conv3x3_stream"] == 1 and L.launch_counts()["b1_unit"] == 0
    out, out_h1 = E.b1_unit(h1, (d["w2"],) + d["bn2"], d["w3"], d["b3"], d["pre"], d["w1"], d["bn1"],
                            res=None if folded else d["res"], shortcut=(d["xp"], d["wsc"], d["bsc"]) if folded else None,
                            device=gpu_device)
    assert L.launch_counts()["b1_unit"] == 1
    assert float(packing.from_split(trunk).abs().max()) > 1.0 and float(packing.from_split(h1n).abs().max()) > 0.1
    bad_t = int((out != trunk).sum()), int((out_h1 != h1n).sum())
    assert torch.equal(out, trunk), ("trunk", bad_t, float((packing.from_split(out) - packing.from_split(trunk)).abs().max()))
    assert torch.equal(out_h1, h1n), ("h1'", bad_t, float((packing.from_split(out_h1) - packing.from_split(h1n)).abs().max()))
    # ... and against float64 on the operands as stored (the conv2 -> conv3 -> conv1 chain rounds h2 and the trunk to 22 bits)
    t64 = packing.from_split(out).double().cpu().numpy()
    x = packing.from_split(h1).double().cpu().numpy()
    xp_ = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
    c2 = sum(xp_[:, ky:ky + h, kx:kx + w, :] @ d["w2"][ky, kx].astype(np.float64) for ky in range(3) for kx in range(3))
    h2r = np.maximum(c2 * d["bn2"][0] + d["bn2"][1], 0)
    ref = h2r @ d["w3"][0, 0].astype(np.float64) + d["b3"]
    ref = ref + (d["xp"].astype(np.float64) @ d["wsc"][0, 0].astype(np.float64) + d["bsc"] if folded else d["res"].astype(np.float64))
    assert np.abs(t64 - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())


def test_b1_unit_refuses_what_it_is_not_built_for(gpu_device):
    from human_dynamics_amd import engine as E
    wide = _unit_inputs(1, 4, 64, 3, False)
    with pytest.raises(L.HmmrError, match="at most 56 pixels wide"):
        E.b1_unit(wide["h1"], (wide["w2"],) + wide["bn2"], wide["w3"], wide["b3"], wide["pre"], wide["w1"], wide["bn1"], res=wide["res"],
                  device=gpu_device)


def test_resnet_block1_unit_kernel_equals_layer_per_launch(weights, gpu_device):
    """The default f16x3 schedule runs block1/unit_1 (folded shortcut) and unit_2 (identity shortcut) as ONE launch each; the features equal
    the layer-per-launch schedule of the same packing bit for bit, and the round-3 block 1 (b1_unit=False: LDS-panel tails, tap-major
    conv2) still equals ITS layer-per-launch schedule."""
    from human_dynamics_amd.engine import HmmrEngine
    frames = assets.make_synthetic_frames(6, seed=23)
    fused = HmmrEngine(weights, None, dtype="f16x3", device=gpu_device, autotune=False)
    assert [bool(fused.rw.unit[i].unit_stream) for i in range(3)] == [True, True, False]
    plain = HmmrEngine(weights, None, dtype="f16x3", device=gpu_device, autotune=False, fuse_tail=False)
    L.launch_counts(clear=True)
    a = fused.resnet(frames, n_zero=1)
    torch.cuda.synchronize()
    c = L.launch_counts(clear=True)
    assert c["b1_unit"] == 2 and c["tail_split"] == 0, c
    b = plain.resnet(frames, n_zero=1)
    torch.cuda.synchronize()
    c = L.launch_counts(clear=True)
    assert c["b1_unit"] == 0 and c["unit_pair"] == 0 and c["conv3x3_stream"] == 13, c      # 11 + block 1's two
    assert float(b.abs().max()) > 0.1
    assert torch.equal(a, b), float((a - b).abs().max())
    old = HmmrEngine(weights, None, dtype="f16x3", device=gpu_device, autotune=False, b1_unit=False)
    old_plain = HmmrEngine(weights, None, dtype="f16x3", device=gpu_device, autotune=False, b1_unit=False, fuse_tail=False)
    o = old.resnet(frames, n_zero=1)
    torch.cuda.synchronize()
    assert L.launch_counts(clear=True)["tail_split"] == 2
    assert torch.equal(o, old_plain.resnet(frames, n_zero=1))
    # the two block-1 forms differ by the summation order of conv2 only
    assert float((o - a).abs().max()) < 2e-4 * float(a.abs().max())


@pytest.mark.parametrize("n", [19, 66], ids=["20_frames", "67_frames"])
def test_unit_pair_on_equals_unit_pair_off_on_one_batch(weights, gpu_device, n):
    """hmmr_resnet50_fwd takes the two launches instead of a unit pair below 12 000 pixels (block 2) / 14 000 (block 3).  Here ONE batch runs with the switch forced on
    (every pair of blocks 2-3 through csrc/unit_pair.hip: 3 + 5 launches, counted) and forced off (none), ragged last tiles included
    (20 images: 122.5 / 30.6 tiles of 128 pixels in blocks 2 / 3; 67: 410.4 / 102.6), and at the default threshold: the same bits."""
    from human_dynamics_amd import engine as E
    frames = assets.make_synthetic_frames(n, seed=31)
    eng = E.HmmrEngine(weights, None, dtype="f16x3", device=gpu_device, autotune=False)
    try:
        E.set_debug(pair_min_pixels=1)
        L.launch_counts(clear=True)
        on = eng.resnet(frames, n_zero=1, parts=1)
        torch.cuda.synchronize()
        assert L.launch_counts(clear=True)["unit_pair"] == 8
        E.set_debug(pair_min_pixels=2 ** 31 - 1)
        off = eng.resnet(frames, n_zero=1, parts=1)
        torch.cuda.synchronize()
        assert L.launch_counts(clear=True)["unit_pair"] == 0
        # the block-2 pairs with TWO tiles per workgroup (the ring left streaming across the tile boundary; the default from 512 tiles of
        # 128 pixels = 84 frames): forced on here at 122.5 / 410.4 tiles -- an odd tile count leaves the last workgroup one tile
        E.set_debug(pair_min_pixels=1, pair_two_tile_min=1)
        two = eng.resnet(frames, n_zero=1, parts=1)
        torch.cuda.synchronize()
        assert L.launch_counts(clear=True)["unit_pair"] == 8
        assert torch.equal(two, on), float((two - on).abs().max())
        E.set_debug(pair_min_pixels=1, pair_two_tile_min=2 ** 31 - 1)
        one = eng.resnet(frames, n_zero=1, parts=1)
        torch.cuda.synchronize()
        L.launch_counts(clear=True)
        assert torch.equal(one, on)
        # round 6: the wave-specialised form of the pairs with a shortcut tensor (hmmr_debug_t.pair_form = 2: two waves per SIMD, conv3 and
        # the trunk epilogue in one, conv1' and all memory traffic in the other; unit_pair_ws_kernel) -- other waves, another ring depth,
        # another epilogue split, the same products in the same order: the same bits, ragged last tiles included
        E.set_debug(pair_min_pixels=1, pair_form=2)
        ws = eng.resnet(frames, n_zero=1, parts=1)
        torch.cuda.synchronize()
        assert L.launch_counts(clear=True)["unit_pair"] == 8
        assert torch.equal(ws, on), float((ws - on).abs().max())
    finally:
        E.set_debug()
    dflt = eng.resnet(frames, n_zero=1, parts=1)
    torch.cuda.synchronize()
    want = {19: 3, 66: 3}[n]            # 20 images: block 2 has 15 680 pixels (pairs on), block 3 3 920 (off); 67: 52 528 (on) and 13 132, below block 3's 14 000 (off)
    assert L.launch_counts(clear=True)["unit_pair"] == want
    assert float(on.abs().max()) > 0.1
    assert torch.equal(on, off), float((on - off).abs().max())
    assert torch.equal(on, dflt)


def test_stem_with_unit1_conv1_inside_equals_the_separate_launch(weights, gpu_device):
    """f16x3, round 5: the fused stem kernel computes block1/unit_1's conv1 on its pooled tile (hmmr_resnet_unit_t.conv1_frag) -- the same
    products in the same order on the tile as stored, so the features equal those of the schedule that launches the layer
    (stem_conv1=False), bit for bit; 7 images: 343 stem tiles, the zero padding image among them."""
    from human_dynamics_amd.engine import HmmrEngine
    frames = assets.make_synthetic_frames(6, seed=41)
    a = HmmrEngine(weights, None, dtype="f16x3", device=gpu_device, autotune=False)
    b = HmmrEngine(weights, None, dtype="f16x3", device=gpu_device, autotune=False, stem_conv1=False)
    assert a.rw.unit[0].conv1_frag and not b.rw.unit[0].conv1_frag
    fa, fb = a.resnet(frames, n_zero=1), b.resnet(frames, n_zero=1)
    assert float(fb.abs().max()) > 0.1
    assert torch.equal(fa, fb), float((fa - fb).abs().max())
