"""Host-side packing of checkpoint-named arrays (assets.py / SURVEY App. B) into
the device layouts libhmmr_hip.so consumes (include/hmmr_hip.h).

All folds happen here, once, in float64:
  * inference BN -> (scale, shift):  scale = gamma*rsqrt(var+1e-5),
    shift = beta - mean*scale                       (slim batch_norm, App. A)
  * HWIO conv filters -> [cout_pad][kh*kw*cin] (K contiguous per output channel)
  * 7x7/2 stem -> 8 taps x 32 elements (8 pixels x RGBX) on a padded image
  * fc1 of the IEF regressors split into its phi rows and theta rows
  * SMPL: planar blend-shape basis, folded joint regressor, ELL skinning
    weights, CSR keypoint regressor
PyTorch is used only as the HBM allocator / H2D copier.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib as L
from . import assets

SPLIT = "split"      # storage marker of HMMR_F16X3 (f16x3) tensors (int32 words holding fp16 hi/lo halves)
SPLIT_HALF = torch.float16
SPLIT_MAX = 65504.0  # values are clamped to the fp16 range where they are split (csrc/common.h split_clamp)
W_TARGET_LOG2 = 14   # filter rows are scaled by a power of two to a maximum in [2^13, 2^14)
TORCH_DT = {L.HMMR_F32: torch.float32, L.HMMR_BF16: torch.bfloat16, L.HMMR_F16X3: SPLIT}


def to_split(t):
    """fp32 tensor [..., C] (C % 8 == 0) -> int32 tensor [..., C] in the split (f16x3) layout: every group of 8
    channels is 32 bytes, [hi x8][lo x8] with hi = fp16(x), lo = fp16(x - hi), x clamped to +-65504 (include/hmmr_hip.h)."""
    t = t.to(torch.float32).clamp(-SPLIT_MAX, SPLIT_MAX)
    C = t.shape[-1]
    assert C % 8 == 0, "split tensors need a channel count that is a multiple of 8"
    hi = t.to(SPLIT_HALF)
    lo = (t - hi.to(torch.float32)).to(SPLIT_HALF)
    g = torch.stack([hi.reshape(t.shape[:-1] + (C // 8, 8)), lo.reshape(t.shape[:-1] + (C // 8, 8))], dim=-2)
    return g.reshape(t.shape[:-1] + (2 * C,)).contiguous().view(torch.int32)


def from_split(t):
    """inverse of to_split: int32 [..., C] -> fp32 [..., C] (hi + lo)."""
    C = t.shape[-1]
    g = t.contiguous().view(SPLIT_HALF).reshape(t.shape[:-1] + (C // 8, 2, 8)).to(torch.float32)
    return (g[..., 0, :] + g[..., 1, :]).reshape(t.shape[:-1] + (C,))


def empty_act(shape, dtype, device, zero=False):
    """Activation tensor of a stage dtype code (split tensors are int32 words)."""
    td = TORCH_DT[dtype]
    td = torch.int32 if td is SPLIT else td
    return (torch.zeros if zero else torch.empty)(shape, dtype=td, device=device)


def act_to_f32(t, dtype):
    return from_split(t) if TORCH_DT[dtype] is SPLIT else t.float()


def _pad_rows(a, mult=128):
    r = (-a.shape[0]) % mult
    if r:
        a = np.concatenate([a, np.zeros((r,) + a.shape[1:], a.dtype)], axis=0)
    return a


def fold_bn(w, prefix, eps=assets.BN_EPS):
    g = w[prefix + "/gamma"].astype(np.float64)
    b = w[prefix + "/beta"].astype(np.float64)
    m = w[prefix + "/moving_mean"].astype(np.float64)
    v = w[prefix + "/moving_variance"].astype(np.float64)
    scale = g / np.sqrt(v + eps)
    return scale.astype(np.float32), (b - m * scale).astype(np.float32)


def pack_conv_weight(w_hwio, k_order=0, chunk=32):
    """[kh,kw,cin,cout] -> [cout_pad][kh*kw*cin] float32.  k_order 0: k = (ky*kw + kx)*cin + ci.  k_order 1
    (hmmr_conv_desc_t.k_order, chunk-major, the 3x3 patch kernel): k = ((ci // chunk)*kh*kw + ky*kw + kx)*chunk +
    ci % chunk, chunk = the elements of one 128-byte K step of the tensor the filter meets (32 for a split tensor)."""
    kh, kw, cin, cout = w_hwio.shape
    if k_order:
        assert cin % chunk == 0, (cin, chunk)
        wk = w_hwio.reshape(kh * kw, cin // chunk, chunk, cout).transpose(1, 0, 2, 3)     # [chunk idx][tap][e][cout]
        return _pad_rows(np.ascontiguousarray(wk.reshape(kh * kw * cin, cout).T))
    return _pad_rows(np.ascontiguousarray(w_hwio.reshape(kh * kw * cin, cout).T))


def row_pow2(w_rows):
    """Per-row exponents k of the power-of-two scaling of a split filter bank [n_out][K]: row r is stored as
    w[r] * 2^k[r] with max |w[r]| * 2^k[r] in [2^13, 2^14), so that the lo half of every filter value (2^-11 of it) is a
    NORMAL fp16 number down to values 2^-14 x the row's largest (an unscaled fp16 split keeps 2-3 bits of a 0.003-sized
    filter's lo half: fp16 subnormals).  The epilogue's per-channel `scale` is multiplied by 2^-k[r]: exact, so the
    arithmetic is that of the unscaled filters with 22-bit operands.  All-zero rows (padding): k = 0.  A pure function
    of the row's values, so every packing of the same rows (K-contiguous, fragment-major) scales them alike."""
    w_rows = np.asarray(w_rows, np.float64)
    m = np.abs(w_rows).max(axis=1)
    k = np.zeros(len(m), np.int64)
    nz = m > 0
    k[nz] = W_TARGET_LOG2 - 1 - np.floor(np.log2(m[nz])).astype(np.int64)
    return k


def scale_rows(w_rows, k):
    return (np.asarray(w_rows, np.float64) * np.exp2(k.astype(np.float64))[:, None]).astype(np.float32)


def pack_frag_major(w_nk):
    """[n_out][K] fp32 -> fp16 [n_out / 32][K / 16][64 lanes][2 (hi, lo)][8]: the MFMA A-operand fragments of a split
    filter bank, one coalesced 2 KB read per (32-row block, 16-wide K chunk); lane = 32 * (k half) + row
    (hmmr_tail_desc_t, csrc/bottleneck_split.hip).  Rows carry the power-of-two scale of row_pow2()."""
    w_nk = np.ascontiguousarray(w_nk, dtype=np.float32)
    t = torch.from_numpy(scale_rows(w_nk, row_pow2(w_nk)))
    n, K = t.shape
    assert n % 32 == 0 and K % 16 == 0, (n, K)
    hi = t.to(SPLIT_HALF)
    lo = (t - hi.to(torch.float32)).to(SPLIT_HALF)

    def frag(x):
        x = x.reshape(n // 32, 32, K // 16, 2, 8)                # rb, row, kc, half, e
        return x.permute(0, 2, 3, 1, 4).reshape(n // 32, K // 16, 64, 8)
    return torch.stack([frag(hi), frag(lo)], dim=3).contiguous()   # [rb, kc, lane, 2, 8]


def pack_conv3x3_stream(w_hwio, k=None, bf16=False):
    """[3,3,cin,cout] -> fp16 [cout / 128][9 cin / 16][4][2 (hi, lo plane)][64 lanes][8]: the filter stream of a k_order 2 layer
    (hmmr_conv_desc_t.k_order, csrc/conv3x3_stream.hip).  K step kt = (ci // 16) * 9 + ky * 3 + kx of a 128-channel tile is 8 KB:
    row blocks 0 .. 3, each the MFMA A operand of 32 rows x 16 K as a hi and a lo plane (lane = 32 * (k half) + row, 8 halves =
    W[ky, kx, 16 (ci // 16) + 8 half .. + 7, 128 tile + 32 block + row]), rows scaled by 2^k (row_pow2 of the rows).  cout = 64: one
    tile of two row blocks ([1][9 cin / 16][2][2][64][8]).
    bf16=True (bf16 tensors): bfloat16 [cout / 128][9 cin / 32][4][2][64][8] -- a K step is (ci // 32) * 9 + tap, its two planes are the
    two 16-wide MFMA chunks of those 32 channels (8 values = W[.., 32 (ci // 32) + 16 plane + 8 half .. + 7, ..]); no row scaling."""
    w = np.asarray(w_hwio, np.float64)
    kh, kw, cin, cout = w.shape
    assert (kh, kw) in ((3, 3), (1, 1)) and cin % 16 == 0 and (cout % 128 == 0 or cout == 64), w.shape      # ((1, 1): pack_conv1x1_stream)
    T = kh * kw
    tw = 128 if cout % 128 == 0 else 64
    if bf16:
        assert cin % 32 == 0 and T == 9, w.shape
        t = torch.from_numpy(w.astype(np.float32)).to(torch.bfloat16)
        x = t.reshape(9, cin // 32, 2, 2, 8, cout // tw, tw // 32, 32)       # tap, c32, plane, half, e, tile, rb, row
        return x.permute(5, 1, 0, 6, 2, 3, 7, 4).reshape(cout // tw, 9 * (cin // 32), tw // 32, 2, 64, 8).contiguous()
    if k is None:
        k = row_pow2(w.reshape(T * cin, cout).T)
    t = torch.from_numpy((w * np.exp2(np.asarray(k, np.float64))).astype(np.float32))
    hi = t.to(SPLIT_HALF)
    lo = (t - hi.to(torch.float32)).to(SPLIT_HALF)

    def frag(x):
        x = x.reshape(T, cin // 16, 2, 8, cout // tw, tw // 32, 32)      # tap, c16, half, e, tile, rb, row
        return x.permute(4, 1, 0, 5, 2, 6, 3).reshape(cout // tw, T * (cin // 16), tw // 32, 64, 8)
    return torch.stack([frag(hi), frag(lo)], dim=3).contiguous()          # [tile, kt, rb, plane, lane, 8]


def pack_conv1x1_stream(w_hwio, k=None):
    """[1,1,cin,cout] -> fp16 [cout / 128][cin / 16][4][2 (hi, lo plane)][64 lanes][8]: the filter stream of a 1x1 layer with k_order 2
    (csrc/conv1x1_stream.hip): K step kt = 16 input channels, the same 8 KB of MFMA A-operand fragments per 128 output channels as one tap
    of pack_conv3x3_stream; rows scaled by 2^k (row_pow2 of the rows)."""
    w = np.asarray(w_hwio)
    assert w.shape[:2] == (1, 1) and w.shape[3] % 128 == 0, w.shape
    return pack_conv3x3_stream(w, k)


def pair_is_a(i, na, ft):
    """fragment i of an iteration's ft = na + nb fragments is a conv3 fragment (csrc/unit_pair.hip: pair_is_a)"""
    return ((i + 1) * na) // ft > (i * na) // ft


def pack_pair_stream(w3_nk, w1_nk):
    """The filter stream of a fused unit pair (hmmr_tail_desc_t.pair_stream, csrc/unit_pair.hip): w3_nk [depth][K3] = this
    unit's conv3 rows ([W3 | Wsc] along K with a folded shortcut), w1_nk [n2][depth] = the next unit's conv1 rows ->
    fp16 [depth / 32 + 2][ft][2 (hi, lo plane)][64 lanes][8]: iteration `it` holds the K3 / 16 fragments of conv3 row block
    `it` (zeros past the last one) and the n2 / 16 fragments of conv1' K step it - 2 (K chunk 2 (it - 2) + kcl, row block of;
    fragment kb = kcl * (n2 / 32) + of; zeros for it < 2), interleaved by pair_is_a.  A fragment = the MFMA A operand of 32
    rows x 16 K: lane = 32 * (k half) + row, 8 halves = W[32 rb + row][16 kc + 8 half .. + 7], rows scaled by row_pow2()."""
    w3_nk = np.ascontiguousarray(w3_nk, dtype=np.float32)
    w1_nk = np.ascontiguousarray(w1_nk, dtype=np.float32)
    depth, K3 = w3_nk.shape
    n2 = w1_nk.shape[0]
    assert w1_nk.shape[1] == depth and depth % 32 == 0 and K3 % 16 == 0 and n2 % 32 == 0, (w3_nk.shape, w1_nk.shape)

    def planar(w_nk):
        t = torch.from_numpy(scale_rows(w_nk, row_pow2(w_nk)))
        n, K = t.shape
        hi = t.to(SPLIT_HALF)
        lo = (t - hi.to(torch.float32)).to(SPLIT_HALF)

        def frag(x):
            x = x.reshape(n // 32, 32, K // 16, 2, 8)                # rb, row, kc, half, e
            return x.permute(0, 2, 3, 1, 4).reshape(n // 32, K // 16, 64, 8)
        return torch.stack([frag(hi), frag(lo)], dim=2)              # [rb, kc, plane, lane, 8]

    f3, f1 = planar(w3_nk), planar(w1_nk)
    nch, na, nf2 = depth // 32, K3 // 16, n2 // 32
    nb = 2 * nf2
    ft = na + nb
    assert ft % 8 == 0, "an iteration must be a whole number of 8-fragment slabs"
    out = torch.zeros((nch + 2, ft, 2, 64, 8), dtype=SPLIT_HALF)
    ia = [i for i in range(ft) if pair_is_a(i, na, ft)]
    ib = [i for i in range(ft) if not pair_is_a(i, na, ft)]
    assert len(ia) == na and [(i * na) // ft for i in ia] == list(range(na))
    out[:nch, ia] = f3                                               # conv3 row block it, K chunks in order
    for kb, i in enumerate(ib):
        kcl, of = divmod(kb, nf2)
        out[2:, i] = f1[of, kcl::2]                                  # K chunk 2 (it - 2) + kcl for it = 2 .. nch + 1
    return out.contiguous()


def pack_b1_unit_stream(w2_hwio, k2, w3_nk, w1_nk):
    """The filter stream of a whole block-1 unit (hmmr_tail_desc_t.unit_stream, csrc/b1_unit.hip): w2_hwio [3,3,64,64] = conv2 with its row
    exponents k2 (as _layer_stream3x3 scales them), w3_nk [256][K3] = conv3's rows ([W3 | Wsc] along K with a folded shortcut, K3 = 64 or
    128), w1_nk [64][256] = the next unit's conv1 rows -> fp16 [fragments][2 (hi, lo plane)][64 lanes][8], 2 KB per fragment:
    conv2's stream exactly as pack_conv3x3_stream lays it out for cout = 64 (36 K steps x two row blocks), then the tail in the order the
    kernel's software pipeline consumes it, A(0) | A(1) B(0) | A(2) B(1) | ... | A(7) B(6) | B(7): A(c) = the K3 / 16 fragments of conv3 row
    block c (K chunks in order), B(c) = the four fragments of conv1' K chunks 2 c and 2 c + 1 (row blocks 0, 1 of each).  Rows of w3 / w1 carry row_pow2() like every split filter bank."""
    w2 = np.asarray(w2_hwio, np.float32)
    assert w2.shape == (3, 3, 64, 64), w2.shape
    w3_nk = np.ascontiguousarray(w3_nk, dtype=np.float32)
    w1_nk = np.ascontiguousarray(w1_nk, dtype=np.float32)
    depth, K3 = w3_nk.shape
    assert depth == 256 and K3 in (64, 128) and w1_nk.shape == (64, 256), (w3_nk.shape, w1_nk.shape)
    parts = [pack_conv3x3_stream(w2, k2).reshape(-1, 2, 64, 8)]          # [36 x 2 fragments][plane][lane][8]

    def planar(w_nk):
        t = torch.from_numpy(scale_rows(w_nk, row_pow2(w_nk)))
        n, K = t.shape
        hi = t.to(SPLIT_HALF)
        lo = (t - hi.to(torch.float32)).to(SPLIT_HALF)

        def frag(x):
            x = x.reshape(n // 32, 32, K // 16, 2, 8)                # rb, row, kc, half, e
            return x.permute(0, 2, 3, 1, 4).reshape(n // 32, K // 16, 64, 8)
        return torch.stack([frag(hi), frag(lo)], dim=2)              # [rb, kc, plane, lane, 8]

    f3, f1 = planar(w3_nk), planar(w1_nk)
    nch = depth // 32
    parts.append(f3[0])                                              # A(0): conv3 row block 0, K3 / 16 fragments
    for c in range(nch):
        if c + 1 < nch:
            parts.append(f3[c + 1])                                  # A(c + 1): issued in front of chunk c's epilogue
        parts.append(torch.stack([f1[j, 2 * c + kcl] for kcl in range(2) for j in range(2)]))     # B(c)
    out = torch.cat(parts).contiguous()
    assert out.numel() * 2 == 18 * 8192 + (depth // 32) * (K3 // 16 + 4) * 2048
    return out


def pack_stem_weight(w_hwio):
    """[7,7,3,64] -> [128][8*32]: k = ky*32 + kx*4 + c (kx = 7, c = 3 and ky = 7 are zero)."""
    out = np.zeros((64, 8, 8, 4), np.float32)
    out[:, :7, :7, :3] = np.transpose(w_hwio, (3, 0, 1, 2))
    return _pad_rows(out.reshape(64, 256))


class DeviceStore(object):
    """Keeps every packed tensor alive and hands out device pointers."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.tensors = []

    def put(self, arr, dtype=torch.float32):
        t = torch.from_numpy(np.ascontiguousarray(arr)).to(self.device)
        if dtype is SPLIT:
            t = to_split(t)
        elif t.dtype != dtype:
            t = t.to(dtype)
        self.tensors.append(t)
        return t

    def put_tensor(self, t):
        """a host tensor that is already in its device layout"""
        t = t.to(self.device)
        self.tensors.append(t)
        return t

    def vec(self, arr, mult=128):
        return self.put(_pad_rows(np.asarray(arr, np.float32), mult))


# --------------------------------------------------------------------------- #
# The C-side packers (csrc/pack.cpp, include/hmmr_hip.h "Packers"): ONE implementation of the shipped layouts, usable
# without Python.  The pack_* functions below call them for the shipped configuration; their Python bodies serve the
# development switches of devflags.py and are pinned to the C bytes by tests/test_packers.py.
# --------------------------------------------------------------------------- #
def _vars_table(w):
    """dict of checkpoint-named arrays -> (hmmr_var_t array, count, the arrays kept alive)"""
    import ctypes as C
    items = [(k, np.ascontiguousarray(v, np.float32)) for k, v in w.items() if np.asarray(v).dtype.kind == "f"]
    arr = (L.Var * len(items))()
    for i, (k, a) in enumerate(items):
        arr[i].name, arr[i].data, arr[i].numel = k.encode(), a.ctypes.data, a.size
    return arr, len(items), items


def _c_pack(store, nbytes, fill):
    """Allocate the stage's blob on the store's device, let `fill(host_ptr, nbytes, device_base)` write the host image and the
    struct, copy it over in one piece."""
    if not nbytes:
        raise L.HmmrError("packer: %s" % (L.load().hmmr_last_error() or b"?").decode())
    dev = torch.empty(int(nbytes), dtype=torch.uint8, device=store.device)
    host = dev if dev.device.type == "cpu" else torch.empty(int(nbytes), dtype=torch.uint8)
    L.check(fill(host.data_ptr(), int(nbytes), dev.data_ptr()), "hmmr_pack_*")
    if host is not dev:
        dev.copy_(host)
    store.tensors.append(dev)
    return dev


def _layer(store, w_packed, dtype, scale=None, shift=None):
    """One hmmr_layer_t.  Split (f16x3) filter banks are scaled row by row (row_pow2) and the epilogue scale takes the
    inverse factor -- a bias-only layer gets a scale vector of pure powers of two for it."""
    lay = L.Layer()
    if TORCH_DT[dtype] is SPLIT:
        w_packed = np.asarray(w_packed, np.float32)
        k = row_pow2(w_packed)
        w_packed = scale_rows(w_packed, k)
        inv = np.exp2(-k.astype(np.float64))
        sc = np.ones(len(k), np.float64)
        if scale is not None:
            sc[:len(scale)] = np.asarray(scale, np.float64)
        scale = (sc * inv).astype(np.float32)          # exact: a power of two times a float32
    lay.w = store.put(w_packed, TORCH_DT[dtype]).data_ptr()
    lay.scale = store.vec(scale).data_ptr() if scale is not None else None
    lay.shift = store.vec(shift).data_ptr() if shift is not None else None
    return lay


def _layer_stream3x3(store, w_hwio, scale, shift, bf16=False):
    """hmmr_layer_t of a k_order 2 conv2: the filter stream instead of a matrix; f16x3: the same row scaling as _layer."""
    lay = L.Layer()
    if bf16:
        lay.w = store.put_tensor(pack_conv3x3_stream(w_hwio, bf16=True)).data_ptr()
        lay.scale = store.vec(scale).data_ptr()
    else:
        rows = pack_conv_weight(w_hwio)[:w_hwio.shape[3]]
        k = row_pow2(rows)
        lay.w = store.put_tensor(pack_conv3x3_stream(w_hwio, k)).data_ptr()
        lay.scale = store.vec((np.asarray(scale, np.float64) * np.exp2(-k.astype(np.float64))).astype(np.float32)).data_ptr()
    lay.shift = store.vec(shift).data_ptr()
    lay.k_order = 2
    return lay


def _layer_stream1x1(store, w_hwio, scale, shift):
    """hmmr_layer_t of a k_order 2 1x1 layer (f16x3): the filter stream instead of a matrix, the same row scaling as _layer."""
    lay = L.Layer()
    w_hwio = np.asarray(w_hwio, np.float32)
    cout = w_hwio.shape[3]
    k = row_pow2(pack_conv_weight(w_hwio)[:cout])
    lay.w = store.put_tensor(pack_conv1x1_stream(w_hwio, k)).data_ptr()
    sc = np.ones(cout, np.float64) if scale is None else np.asarray(scale, np.float64)
    lay.scale = store.vec((sc * np.exp2(-k.astype(np.float64))).astype(np.float32)).data_ptr()
    lay.shift = store.vec(shift).data_ptr()
    lay.k_order = 2
    return lay


def pack_resnet(w, dtype, store, fuse_preact_blocks=("block1", "block2", "block3", "block4"), fuse_tail=True,
                fuse_sc=True, fuse_preact_first=False, fold_sc=None, patch_3x3=2, unit_pair=True, b1_stream=False, b1_unit=True,
                stem_conv1=True, stream_1x1=True, impl=None):
    """The shipped configuration (every argument at its default: what HmmrEngine builds unless a development switch of devflags.py
    says otherwise) is packed by the C-side packer hmmr_pack_resnet (csrc/pack.cpp); any other choice, or impl="py", by the Python
    form below (_pack_resnet_py), whose bytes for the shipped configuration equal the C packer's (tests/test_packers.py)."""
    shipped = (tuple(fuse_preact_blocks) == ("block1", "block2", "block3", "block4") and fuse_tail is True and fuse_sc is True and
               not fuse_preact_first and fold_sc in (None, dtype == L.HMMR_F16X3) and patch_3x3 == 2 and patch_3x3 is not True and
               unit_pair is True and not b1_stream and b1_unit is True and stem_conv1 is True and stream_1x1 is True)
    if impl == "c" and not shipped:
        raise ValueError("the C-side packer builds the shipped configuration only")
    if impl == "py" or not shipped:
        return _pack_resnet_py(w, dtype, store, fuse_preact_blocks, fuse_tail, fuse_sc, fuse_preact_first, fold_sc, patch_3x3, unit_pair,
                               b1_stream, b1_unit, stem_conv1, stream_1x1)
    lib = L.load()
    import ctypes as C
    arr, n, keep = _vars_table({k: v for k, v in w.items() if k.startswith("resnet_v2_50/")})
    rw = L.ResnetWeights()
    _c_pack(store, lib.hmmr_pack_resnet_bytes(arr, n, dtype),
            lambda host, nb, base: lib.hmmr_pack_resnet(arr, n, dtype, host, nb, base, C.byref(rw)))
    del keep
    return rw


def _pack_resnet_py(w, dtype, store, fuse_preact_blocks=("block1", "block2", "block3", "block4"), fuse_tail=True,
                    fuse_sc=True, fuse_preact_first=False, fold_sc=None, patch_3x3=2, unit_pair=True, b1_stream=False, b1_unit=True,
                    stem_conv1=True, stream_1x1=True):
    """fuse_preact_blocks: blocks whose units apply their `preact` BN+ReLU inside the operand
    staging of conv1/shortcut instead of reading a materialised preact tensor (csrc/resnet.hip;
    measured at batch 256: -4.5 % ResNet time with blocks 1-2 fused, neutral for blocks 3-4).
    fuse_tail: mark the units whose conv3 + add runs with the next unit's preact + conv1 as one
    hmmr_bottleneck_tail launch (bf16; the stride-1 units of block1 and block2).
    patch_3x3 (f16x3: blocks 2-4; bf16: blocks 3-4): the stride-1 3x3 conv2 out of an LDS-resident input patch.  2 (default): the
    layers get hmmr_conv_desc_t.k_order = 2, the filter stream of the one-wave-per-SIMD kernel (csrc/conv3x3_stream.hip, tiles 12 .. 18);
    1 / True: k_order = 1, chunk-major rows for the 8-wave patch kernels (csrc/gemm_conv.hip, tiles 9 / 10; 11 for f16x3).
    b1_stream (with patch_3x3 = 2, f16x3; default off): the conv2 of block1/unit_1 and unit_2 is a k_order 2 launch too (64-channel tiles,
    56-pixel images) and their fused tails start at conv3 (hmmr_resnet_unit_t.fuse_tail = 1); False: conv2 runs inside those tails,
    tap-major.  Measured equal (profiles/r04c: 0.188 + 0.434 ms against 0.632 ms per unit), so the form with 0.4 GB less HBM traffic stays.
    The stride-2 units keep the im2col gather.
    b1_unit (with patch_3x3 = 2, f16x3; default on, round 5): block1/unit_1 and unit_2 run as the whole-unit kernel of csrc/b1_unit.hip
    (hmmr_resnet_unit_t.unit_stream, fuse_tail = 2): their conv2 is packed k_order 2 as with b1_stream, so the layer-per-launch schedule
    (fuse_tail=False) of the same configuration runs it through the 3x3 stream kernel and produces the same bits.  False (and b1_stream
    False): the round-3 tails of csrc/bottleneck_split.hip with conv2 inside, tap-major.
    stem_conv1 (f16x3; default on, round 5): block1/unit_1's conv1 is computed by the fused stem kernel on its pooled tile (hmmr_resnet_unit_t.
    conv1_frag); bf16 does that since round 1 (hmmr_debug_t.stem_no_conv1 switches either off at run time).  Same bits as the launch.
    stream_1x1 (f16x3; default on, round 5): the conv1 of block 4's units and of block2/unit_1, and block3/unit_1's shortcut + conv1 launch run the two-ring
    stream kernel of csrc/conv1x1_stream.hip (hmmr_layer_t.k_order = 2 on a 1x1 layer), and so does block 4's conv3 (with its shortcut add, folded
    shortcut and second output); the kernel takes the pre-activated tensor, so
    block4/unit_2 and unit_3 read the one their predecessor's conv3 writes (fuse_preact = 0) instead of applying it while staging.
    unit_pair (f16x3 only; True | "block2" | "block3" | False): the stride-1 units of blocks 2-3 run conv3 + add + the next
    unit's preact + conv1 as the register-resident unit pair of csrc/unit_pair.hip (one filter stream per unit); block3/unit_1
    then keeps its conv shortcut as a launch (shortcut + conv1 as one column-split GEMM) instead of folding it into conv3."""
    if fold_sc is None:
        # fold every conv shortcut into its unit's conv3 (one GEMM over {h2, preact}: hmmr_resnet_unit_t.c3sc).  Default in
        # the f16x3 mode, where it removes the widest tensor of the unit (4 B/element) from HBM; the bf16 mode has its own
        # fused units, and in f32 mode it would move the accumulation order away from the layer-per-launch schedule.
        fold_sc = dtype == L.HMMR_F16X3
    bke = 64 if dtype == L.HMMR_BF16 else 32
    pair_shapes = ()
    if dtype == L.HMMR_F16X3 and unit_pair:
        pair_shapes = {"block2": ((128, 512),), "block3": ((256, 1024),)}.get(unit_pair, ((128, 512), (256, 1024)))
    rw = L.ResnetWeights()
    rw.dtype = dtype
    rw.stem = _layer(store, pack_stem_weight(w["resnet_v2_50/conv1/weights"]), dtype,
                     shift=w["resnet_v2_50/conv1/biases"])
    units = list(assets.resnet_units())
    assert len(units) == L.RESNET_UNITS
    for i, (scope, c_in, base, depth, stride, has_sc) in enumerate(units):
        u = rw.unit[i]
        u.c_in, u.base, u.depth, u.stride = c_in, base, depth, stride
        u.fuse_preact = int(i > 0 and scope.split("/")[1] in fuse_preact_blocks)
        if has_sc and i > 0 and not fuse_preact_first:
            # a block's first unit feeds its preact to a WIDE conv shortcut as well: every N tile of a fused-preact
            # launch repeats the transform, so here the previous unit's conv3 writes the (small, already
            # down-sampled) preact tensor once and both consumers take the plain LDS-DMA operand path
            u.fuse_preact = 0
        s, b = fold_bn(w, scope + "/conv1/BatchNorm")
        s1x1 = bool(stream_1x1) and dtype == L.HMMR_F16X3 and stride == 1
        if s1x1 and i > 0 and base % 128 == 0 and (base == 512 or not u.fuse_preact):
            # block 4's three units, and the first unit of blocks 2-4 (which reads a materialised preact tensor anyway)
            u.fuse_preact = 0
            u.conv1 = _layer_stream1x1(store, w[scope + "/conv1/weights"], s, b)
        else:
            u.conv1 = _layer(store, pack_conv_weight(w[scope + "/conv1/weights"]), dtype, s, b)
        if i == 0 and dtype == L.HMMR_F16X3 and stem_conv1:
            # block1/unit_1's conv1 runs inside the fused stem (csrc/stem.hip): its filters as MFMA A-operand fragments (the same rows and
            # row scaling as the layer above, so the layer's scale / shift apply)
            u.conv1_frag = store.put_tensor(pack_frag_major(np.asarray(w[scope + "/conv1/weights"], np.float32)[0, 0].T)).data_ptr()
        s, b = fold_bn(w, scope + "/conv2/BatchNorm")
        # (bf16, round 4: blocks 3-4 only -- the conv2 of blocks 1-2 runs inside the fused bf16 units, which read the tap-major order)
        stream = dtype in (L.HMMR_F16X3, L.HMMR_BF16) and patch_3x3 == 2 and patch_3x3 is not True
        # (bf16: the conv2 of blocks 1-2 runs inside the fused bf16 units, which read the tap-major order -- unless fuse_tail="conv2b1"
        #  keeps block 2's outside, as launches of the stream kernel)
        b16_min = 128 if (stream and fuse_tail == "conv2b1") else 256
        kord = int(bool(patch_3x3) and stride == 1 and ((dtype == L.HMMR_F16X3 and (base >= 128 or (stream and (b1_stream or b1_unit)))) or
                                                         (dtype == L.HMMR_BF16 and base >= b16_min)))
        # (a chunk-major layer cannot fall back to the im2col gather: its 128-pixel patch -- tile + halo of W + 1 on either side --
        #  must fit the 4 x 64 rows the 128x256 tile keeps in LDS.  ResNet-50 on 224 x 224 crops: W <= 28)
        if kord and not stream and 128 + 2 * (224 // {64: 4, 128: 8, 256: 16, 512: 32}[base]) + 4 > 4 * 64:
            kord = 0
        if kord and stream:
            u.conv2 = _layer_stream3x3(store, np.asarray(w[scope + "/conv2/weights"], np.float32), s, b, bf16=dtype == L.HMMR_BF16)
        else:
            u.conv2 = _layer(store, pack_conv_weight(w[scope + "/conv2/weights"], kord, chunk=bke), dtype, s, b)
            u.conv2.k_order = kord
        s3x = s1x1 and base == 512                   # block 4's conv3 in the conv3 form of the same kernel (shortcut add, the next pre-activation)
        if s3x:
            u.conv3 = _layer_stream1x1(store, w[scope + "/conv3/weights"], None, w[scope + "/conv3/biases"])
        else:
            u.conv3 = _layer(store, pack_conv_weight(w[scope + "/conv3/weights"]), dtype,
                             shift=w[scope + "/conv3/biases"])
        if has_sc:
            u.shortcut = _layer(store, pack_conv_weight(w[scope + "/shortcut/weights"]), dtype,
                                shift=w[scope + "/shortcut/biases"])
            folded = bool(fold_sc) and stride == 1 and not u.fuse_preact and base % bke == 0 and c_in % bke == 0
            if (base, depth) == (256, 1024) and (base, depth) in pair_shapes:
                # the unit pair keeps 32 px x K of conv3's operand in registers: K = 256 + 512 does not fit.  Decided by `unit_pair`
                # alone (not by fuse_tail), so the layer-per-launch schedule of the same configuration rounds the same tensors
                folded = False
            if folded:
                both = np.concatenate([w[scope + "/conv3/weights"], w[scope + "/shortcut/weights"]], axis=2)   # along K
                bias = (np.asarray(w[scope + "/conv3/biases"], np.float64) +
                        np.asarray(w[scope + "/shortcut/biases"], np.float64)).astype(np.float32)
                u.c3sc = _layer_stream1x1(store, both, None, bias) if s3x else _layer(store, pack_conv_weight(both), dtype, shift=bias)
            if not folded and fuse_sc and stride == 1 and (c_in >= 512 or fuse_sc == "all"):   # measured: pays in blocks 3-4 only
                # shortcut and conv1 read the same operand: one [depth + base][c_in] filter bank, conv1's columns
                # after the shortcut's; scale 1 on the shortcut columns (fma(v, 1, b) == v + b exactly)
                both = np.concatenate([w[scope + "/shortcut/weights"], w[scope + "/conv1/weights"]], axis=3)
                s1, b1 = fold_bn(w, scope + "/conv1/BatchNorm")
                sc_s = np.concatenate([np.ones(depth, np.float32), s1])
                sc_b = np.concatenate([np.asarray(w[scope + "/shortcut/biases"], np.float32), b1])
                if s1x1 and not u.fuse_preact and depth % 128 == 0 and base % 128 == 0:
                    u.sc_c1 = _layer_stream1x1(store, both, sc_s, sc_b)
                    u.shortcut.k_order = 2          # (the launch reads its tile from `shortcut`: HmmrEngine._tile_for)
                else:
                    u.sc_c1 = _layer(store, pack_conv_weight(both), dtype, sc_s, sc_b)
        s, b = fold_bn(w, scope + "/preact")
        u.pre_scale, u.pre_shift = store.vec(s).data_ptr(), store.vec(b).data_ptr()
    if dtype == L.HMMR_F16X3 and fuse_tail:
        # f16x3: conv3 + add + the next unit's preact + conv1 as one launch (csrc/bottleneck_split.hip) for the stride-1
        # units of blocks 1-2 whose successor has an identity shortcut; filters fragment-major, a folded shortcut
        # (c3sc) only with a 64-channel unit input (block1/unit_1)
        for i, (scope, c_in, base, depth, stride, has_sc) in enumerate(units[:-1]):
            u, nx = rw.unit[i], rw.unit[i + 1]
            nscope = units[i + 1][0]
            chain = stride == 1 and nx.c_in == depth and nx.base == base and nx.fuse_preact == 1 and not nx.shortcut.w
            pair = (chain and (base, depth) in pair_shapes and fuse_tail not in ("block1",) and
                    (not u.c3sc.w or (base, c_in) == (128, 256)))
            ok = (chain and (base, depth) in ((64, 256), (128, 512)) and (not has_sc or (u.c3sc.w and c_in == 64 and base == 64)))
            if fuse_tail in ("block1",) and base != 64:
                ok = False
            if not (ok or pair):
                continue
            w3 = np.asarray(w[scope + "/conv3/weights"], np.float32)[0, 0]              # [K][depth]
            if u.c3sc.w:
                w3 = np.concatenate([w3, np.asarray(w[scope + "/shortcut/weights"], np.float32)[0, 0]], axis=0)
            w1n = np.asarray(w[nscope + "/conv1/weights"], np.float32)[0, 0]            # [depth][base]
            if pair:
                u.pair_stream = store.put_tensor(pack_pair_stream(w3.T, w1n.T)).data_ptr()
                u.fuse_tail = 1
                continue
            if base == 64 and b1_unit and not b1_stream and fuse_tail != "noconv2" and u.conv2.k_order == 2:
                # the whole unit as one launch (csrc/b1_unit.hip): conv2 (chunk-major, the stream kernel's order) + conv3 + add + next conv1
                w2 = np.asarray(w[scope + "/conv2/weights"], np.float32)
                k2 = row_pow2(pack_conv_weight(w2)[:w2.shape[3]])
                u.unit_stream = store.put_tensor(pack_b1_unit_stream(w2, k2, w3.T, w1n.T)).data_ptr()
                u.fuse_tail = 2
                continue
            u.w3_frag = store.put_tensor(pack_frag_major(w3.T)).data_ptr()
            u.w1n_frag = store.put_tensor(pack_frag_major(w1n.T)).data_ptr()
            u.fuse_tail = 1
            if base == 64 and fuse_tail != "noconv2" and u.conv2.k_order != 2:   # block 1 (56 x 56 = 7 x 7 tiles of 8 x 8): conv2 inside as well
                u.fuse_tail = 2
    for i in range(L.RESNET_UNITS - 1):
        u, nx = rw.unit[i], rw.unit[i + 1]
        shapes = ((64, 256),) if fuse_tail == "block1" else ((64, 256), (128, 512))
        if dtype == L.HMMR_F16X3:
            break
        u.fuse_tail = int(bool(fuse_tail) and dtype == L.HMMR_BF16 and u.stride == 1 and
                          (u.base, u.depth) in shapes and nx.c_in == u.depth and nx.base == u.base and
                          nx.fuse_preact == 1 and not nx.shortcut.w)
        if u.fuse_tail and fuse_tail != "noconv2" and (u.base == 64 or fuse_tail != "conv2b1"):
            u.fuse_tail = 2               # the unit's 3x3 conv2 runs inside the same launch too
            if u.shortcut.w and u.c_in == 64 and not u.fuse_preact and fuse_tail != "nosc":
                u.fuse_tail = 3           # ... and so does its conv shortcut (block1/unit_1)
    for i in range(L.RESNET_UNITS - 1):
        u = rw.unit[i]
        if (fuse_tail and fuse_tail != "nostride2" and dtype == L.HMMR_BF16 and u.stride == 2 and not u.shortcut.w and
                (u.base, u.depth) in ((64, 256), (128, 512))):
            u.fuse_tail = 4               # a block's stride-2 last unit: conv2 + conv3 + add as one launch
    s, b = fold_bn(w, "resnet_v2_50/postnorm")
    rw.post_scale, rw.post_shift = store.vec(s).data_ptr(), store.vec(b).data_ptr()
    return rw


def pack_temporal(w, dtype, store, num_conv_layers=3, impl=None):
    if impl != "py":
        lib = L.load()
        import ctypes as C
        arr, n, keep = _vars_table({k: v for k, v in w.items() if k.startswith("AZ_FC_block")})
        tw = L.TemporalWeights()
        _c_pack(store, lib.hmmr_pack_temporal_bytes(arr, n, dtype, num_conv_layers),
                lambda host, nb, base: lib.hmmr_pack_temporal(arr, n, dtype, num_conv_layers, host, nb, base, C.byref(tw)))
        del keep
        return tw
    tw = L.TemporalWeights()
    tw.dtype, tw.num_blocks = dtype, num_conv_layers
    for i in range(num_conv_layers):
        gn1, c1, gn2, c2 = assets.temporal_scopes(i)
        b = tw.block[i]
        b.gn1_gamma, b.gn1_beta = store.put(w[gn1 + "/gamma"]).data_ptr(), store.put(w[gn1 + "/beta"]).data_ptr()
        b.gn2_gamma, b.gn2_beta = store.put(w[gn2 + "/gamma"]).data_ptr(), store.put(w[gn2 + "/beta"]).data_ptr()
        b.conv1 = _layer(store, pack_conv_weight(w[c1 + "/weights"]), dtype, shift=w[c1 + "/biases"])
        b.conv2 = _layer(store, pack_conv_weight(w[c2 + "/weights"]), dtype, shift=w[c2 + "/biases"])
        if dtype == L.HMMR_F16X3:      # measured (tools/stage_bench.py, 32 windows): the 128x256 ping-pong tile with 5 K slices (csrc/temporal.hip), 0.51 -> 0.43 ms per f_movie pass
            b.conv1.tile = b.conv2.tile = 8
    return tw


def pack_hallucinator(w, dtype, store, impl=None):
    """fc2_res/fc{1,2,3} (src/models.py:283-294); None when the checkpoint has no hallucinator."""
    if "fc2_res/fc1/weights" not in w:
        return None
    if impl != "py":
        lib = L.load()
        import ctypes as C
        arr, n, keep = _vars_table({k: v for k, v in w.items() if k.startswith("fc2_res/")})
        hw = L.HallucinatorWeights()
        _c_pack(store, lib.hmmr_pack_hallucinator_bytes(arr, n, dtype),
                lambda host, nb, base: lib.hmmr_pack_hallucinator(arr, n, dtype, host, nb, base, C.byref(hw)))
        del keep
        return hw
    hw = L.HallucinatorWeights()
    hw.dtype = dtype
    for k in ("fc1", "fc2", "fc3"):
        lay = _layer(store, _pad_rows(np.ascontiguousarray(w["fc2_res/%s/weights" % k].T)), dtype,
                     shift=w["fc2_res/%s/biases" % k])
        setattr(hw, k, lay)
    return hw


def pack_ief(w, dtype, store, delta_t_values=(-5, 5), num_stages=3, impl=None):
    scopes = assets.ief_scopes(delta_t_values)
    keys = [0] + sorted(k for k in scopes if k != 0)      # deltas in sorted order (tester.py:245)
    if impl != "py":
        lib = L.load()
        import ctypes as C
        arr, n, keep = _vars_table({k: v for k, v in w.items() if k.startswith("single_view_ief") or k == "mean_param"})
        dts = (C.c_int * max(1, len(keys) - 1))(*keys[1:])
        iw = L.IefWeights()
        try:
            _c_pack(store, lib.hmmr_pack_ief_bytes(arr, n, dtype, dts, len(keys) - 1),
                    lambda host, nb, base: lib.hmmr_pack_ief(arr, n, dtype, dts, len(keys) - 1, num_stages, host, nb, base, C.byref(iw)))
        except L.HmmrError as e:          # (a mis-shaped checkpoint is the caller's ValueError, as in the Python form)
            raise ValueError(str(e))
        del keep
        return iw, keys
    iw = L.IefWeights()
    iw.dtype, iw.num_regressors, iw.num_stages = dtype, len(keys), num_stages
    for r, key in enumerate(keys):
        scope, nd = scopes[key]
        p = scope + "/3D_module"
        W1 = w[p + "/fc1/weights"]
        if key != 0:
            nd = W1.shape[0] - assets.FEAT_DIM          # 72 (use_optcam) or 75 (models.py:333-336): the checkpoint decides
            if nd not in (72, 75):
                raise ValueError("%s/fc1/weights has %d rows: a delta regressor takes 2048 + 72 or 2048 + 75" % (p, W1.shape[0]))
        assert W1.shape == (assets.FEAT_DIM + nd, 1024) and w[p + "/fc3/weights"].shape == (1024, nd)
        reg = iw.reg[r]
        reg.nd = nd
        reg.fc1_phi = _layer(store, _pad_rows(np.ascontiguousarray(W1[:assets.FEAT_DIM].T)), dtype,
                             shift=w[p + "/fc1/biases"])
        wt = np.zeros((1024, 128), np.float32)
        wt[:, :nd] = W1[assets.FEAT_DIM:].T
        reg.fc1_theta = _layer(store, wt, L.HMMR_F32)            # theta path stays fp32
        reg.fc2 = _layer(store, _pad_rows(np.ascontiguousarray(w[p + "/fc2/weights"].T)), dtype,
                         shift=w[p + "/fc2/biases"])
        reg.fc3 = _layer(store, _pad_rows(np.ascontiguousarray(w[p + "/fc3/weights"].T)), dtype,
                         shift=w[p + "/fc3/biases"])
    iw.mean_theta = store.put(np.asarray(w["mean_param"], np.float32).reshape(85)).data_ptr()
    nds = {iw.reg[r].nd for r in range(1, len(keys))}
    if len(nds) > 1:
        raise ValueError("the delta regressors disagree on use_optcam (theta widths %s)" % sorted(nds))
    iw.no_optcam = int(nds == {75})
    return iw, keys


def pack_smpl(smpl, store, joint_type="cocoplus", split=True, impl=None):
    """tf_smpl-layout constants (src/tf_smpl/batch_smpl.py:35-80) -> SmplConsts.
    split: also pack the blend basis as split-fp16 MFMA fragments (hmmr_smpl_consts_t.dirs_split: the default blend form of
    hmmr_smpl_fwd).  False -- what an all-fp32 engine passes -- or a basis whose entries x 2^13 leave the fp16 range: dirs_split = NULL,
    the library then takes the exact-fp32 vector form."""
    if impl != "py":
        lib = L.load()
        import ctypes as C
        f = lambda k: np.ascontiguousarray(smpl[k], np.float32)
        keep = [f("v_template"), f("shapedirs"), f("posedirs"), f("J_regressor"), f("lbs_weights"), f("cocoplus_regressor"),
                np.ascontiguousarray(np.asarray(smpl["parents"]).astype(np.int32))]
        src = L.SmplSource()
        src.num_verts, src.num_kps = keep[0].shape[0], keep[5].shape[1]
        (src.v_template, src.shapedirs, src.posedirs, src.J_regressor, src.lbs_weights, src.kp_regressor, src.parents) = [a.ctypes.data for a in keep]
        sc = L.SmplConsts()
        lsp = int(joint_type == "lsp")
        _c_pack(store, lib.hmmr_pack_smpl_bytes(C.byref(src), lsp, int(bool(split))),
                lambda host, nb, base: lib.hmmr_pack_smpl(C.byref(src), lsp, int(bool(split)), host, nb, base, C.byref(sc)))
        del keep
        return sc
    nv = smpl["v_template"].shape[0]
    vpad = (nv + 255) // 256 * 256
    v_t = smpl["v_template"].astype(np.float64)
    S = smpl["shapedirs"].astype(np.float64).reshape(10, nv, 3)
    P = smpl["posedirs"].astype(np.float64).reshape(207, nv, 3)
    dirs = np.zeros((224, 3, vpad), np.float32)      # rows >= 218 stay zero (kernel walks k in fours)
    dirs[0, :, :nv] = v_t.T
    dirs[1:11, :, :nv] = np.transpose(S, (0, 2, 1))
    dirs[11:218, :, :nv] = np.transpose(P, (0, 2, 1))
    Jreg = smpl["J_regressor"].astype(np.float64)              # [nv, 24] (stored transposed)
    j_template = (Jreg.T @ v_t).reshape(72)
    j_shapedirs = np.einsum("vj,bvc->bjc", Jreg, S).reshape(10, 72)
    Wl = np.asarray(smpl["lbs_weights"], np.float32)
    nz = Wl != 0
    nnz = max(1, int(nz.sum(axis=1).max()))
    idx = np.zeros((nv, nnz), np.int32)
    val = np.zeros((nv, nnz), np.float32)
    for v in range(nv):
        js = np.nonzero(nz[v])[0]                               # ascending joint order, like the dense sum
        idx[v, :len(js)] = js
        val[v, :len(js)] = Wl[v, js]
    kreg = np.asarray(smpl["cocoplus_regressor"], np.float32)   # [nv, K]
    if joint_type == "lsp":
        kreg = kreg[:, :14]                                     # batch_smpl.py:81-82
    nk = kreg.shape[1]
    kptr, kidx, kval = [0], [], []
    for k in range(nk):
        rows = np.nonzero(kreg[:, k])[0]
        kidx.append(rows.astype(np.int32))
        kval.append(kreg[rows, k])
        kptr.append(kptr[-1] + len(rows))
    kidx = np.concatenate(kidx) if kptr[-1] else np.zeros(1, np.int32)
    kval = np.concatenate(kval) if kptr[-1] else np.zeros(1, np.float32)
    sc = L.SmplConsts()
    sc.num_verts, sc.num_kps, sc.lbs_nnz, sc.vpad = nv, nk, nnz, vpad
    sc.dirs = store.put(dirs).data_ptr()
    # the same basis as split-fp16 MFMA B-operand fragments (hmmr_smpl_consts_t.dirs_split): [K / 16][3][hi, lo][k half][vpad][8]
    # (an entry beyond 65504 / 2^13 = 7.99 m would become inf in torch's fp16 cast: no such body model, but then the vector form it is)
    if split and float(np.abs(dirs).max()) * 8192.0 < 65504.0:
        ds = torch.from_numpy(dirs.astype(np.float64) * 8192.0).to(torch.float32)            # exact: a power of two
        hi = ds.to(SPLIT_HALF)
        lo = (ds - hi.to(torch.float32)).to(SPLIT_HALF)
        frag = lambda x: x.reshape(14, 2, 8, 3, vpad).permute(0, 3, 1, 4, 2)                # kc, c, h, v, e
        sc.dirs_split = store.put_tensor(torch.stack([frag(hi), frag(lo)], dim=2).contiguous()).data_ptr()      # kc, c, plane, h, v, e
    sc.j_template = store.put(j_template.astype(np.float32)).data_ptr()
    sc.j_shapedirs = store.put(j_shapedirs.astype(np.float32)).data_ptr()
    sc.parents = store.put(np.asarray(smpl["parents"]).astype(np.int32), torch.int32).data_ptr()
    sc.lbs_idx = store.put(idx, torch.int32).data_ptr()
    sc.lbs_w = store.put(val).data_ptr()
    sc.kreg_ptr = store.put(np.asarray(kptr, np.int32), torch.int32).data_ptr()
    sc.kreg_idx = store.put(kidx, torch.int32).data_ptr()
    sc.kreg_val = store.put(kval.astype(np.float32)).data_ptr()
    return sc
