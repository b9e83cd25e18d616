"""human_dynamics_amd/hdf5_lite.py -- the reader for `neutral_smpl_meanwjoints.h5` (ref src/evaluation/tester.py:118-135)
-- against files assembled from the published HDF5 / Blosc formats by independent code (tests/hdf5_writer.py), the way the checkpoint and pkl
importers are pinned (tests/test_importers_published_formats.py): neither h5py / PyTables nor the blosc library exist in this image."""
import os
import sys
import zlib

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import hdf5_writer as HW                                   # noqa: E402
from human_dynamics_amd import hdf5_lite as H5              # noqa: E402
from human_dynamics_amd.evaluation import tester as T      # noqa: E402


def test_blosclz_streams_round_trip():
    """the FastLZ-format decoder against the independent greedy encoder: literals only, short and long matches (length extension bytes),
    overlapping matches (runs), distances beyond one byte"""
    rng = np.random.default_rng(0)
    cases = [b"", b"a", bytes(rng.integers(0, 256, 300, dtype=np.uint8)), b"abc" * 200, b"\0" * 1000, b"x" + b"yz" * 700,
             bytes(rng.integers(0, 4, 3000, dtype=np.uint8)), bytes(rng.integers(0, 256, 600, dtype=np.uint8)) * 3]
    for raw in cases:
        c = HW.blosclz_compress(raw)
        assert H5.blosclz_decompress(c, len(raw)) == raw
        if len(raw) > 500 and len(set(raw)) < 10:
            assert len(c) < len(raw) // 2                  # (the encoder does find the matches: the decoder's match path is exercised)
    # hand-assembled: 3 literals 'abc', then a match of length 9 + 4 = 13 at distance 3 (length field 7 + one extension byte)
    s = bytes([2]) + b"abc" + bytes([(7 << 5) | 0, 4, 2])
    assert H5.blosclz_decompress(s, 16) == b"abc" + (b"abc" * 5)[:13]
    with pytest.raises(H5.Hdf5Error):
        H5.blosclz_decompress(bytes([2]) + b"abc" + bytes([(1 << 5) | 0, 9]), 16)      # a match before the start of the output


@pytest.mark.parametrize("mode", ["memcpy", "blosclz", "zlib"])
def test_blosc_frames(mode):
    rng = np.random.default_rng(1)
    smooth = np.cumsum(rng.standard_normal(4096) * 1e-3).astype(np.float64)           # shuffled byte planes compress
    for arr, blocksize in ((smooth, None), (smooth, 8192), (smooth[:1000].astype(np.float32), 1024), (rng.standard_normal(72), None),
                           (np.arange(10, dtype=np.float64), None)):
        raw = arr.tobytes()
        fr = HW.blosc_frame(raw, arr.dtype.itemsize, mode, blocksize)
        assert H5.blosc_decompress(fr) == raw
    with pytest.raises(H5.Hdf5Error):
        H5.blosc_decompress(b"\x02\x01\x21\x08" + b"\0" * 12)                         # lz4: named as unsupported


def test_hdf5_file_of_small_arrays_like_deepdish_writes(tmp_path):
    """dd.io.save({'pose': [72], 'shape': [10], ...}) -> one chunked, Blosc-filtered CArray per array in an old-style root group"""
    rng = np.random.default_rng(2)
    arrays = {"pose": rng.standard_normal(72), "shape": rng.standard_normal(10), "joints": rng.standard_normal((24, 3)).astype(np.float32),
              "count": np.arange(7, dtype=np.int32), "big": np.cumsum(rng.standard_normal((300, 16)), 0)}
    for mode in ("blosclz", "memcpy", "zlib"):
        p = str(tmp_path / ("mean_%s.h5" % mode))
        HW.write_dict(p, arrays, mode=mode, chunk_rows=None if mode != "zlib" else 64)
        got = H5.load(p)
        assert sorted(got) == sorted(arrays)
        for k, v in arrays.items():
            assert got[k].dtype == v.dtype and got[k].shape == v.shape and np.array_equal(got[k], v), (mode, k)
    # the HDF5 library's own filters (shuffle + deflate) instead of Blosc, several chunks, a nested group
    p = str(tmp_path / "gz.h5")
    HW.write_dict(p, {"a": arrays["big"], "sub": {"b": arrays["pose"]}}, blosc=False, extra_filters=("shuffle", "deflate"), chunk_rows=100)
    got = H5.load(p)
    assert np.array_equal(got["a"], arrays["big"]) and np.array_equal(got["sub"]["b"], arrays["pose"])
    with pytest.raises(H5.Hdf5Error):
        open(str(tmp_path / "x.h5"), "wb").write(b"not an hdf5 file" * 10)
        H5.load(str(tmp_path / "x.h5"))


def test_mean_theta_from_the_h5_like_the_reference(tmp_path):
    """tester.py:118-135: cams = [0.9, 0, 0], pose[:3] = [pi, 0, 0], then pose and shape -- from the .h5 itself now, and `load_weights`
    finds `neutral_smpl_meanwjoints.h5` next to the SMPL model where the reference looks for it"""
    rng = np.random.default_rng(3)
    pose, shape = rng.standard_normal(72) * 0.2, rng.standard_normal(10) * 0.5
    p = str(tmp_path / "neutral_smpl_meanwjoints.h5")
    HW.write_dict(p, {"pose": pose, "shape": shape, "joints": rng.standard_normal((24, 3))})
    mean = T.mean_theta_from_file(p)
    ref_pose = pose.copy(); ref_pose[:3] = 0.0; ref_pose[0] = np.pi
    want = np.hstack(([0.9, 0.0, 0.0], ref_pose, shape))[None].astype(np.float32)
    assert mean.shape == (1, 85) and mean.dtype == np.float32 and np.array_equal(mean, want)

    class Cfg(object):
        smpl_model_path = str(tmp_path / "neutral_smpl_with_cocoplus_reg.pkl")
    assert T.Tester._default_mean_path(Cfg()) == p
    w = T.load_weights("synthetic:1", mean_param_path=p)
    assert "mean_param" in w                                                # (the synthetic checkpoint carries the variable: restore wins, tester.py:114-116)
    w.pop("mean_param")
    np.savez(str(tmp_path / "no_mean.npz"), **w)
    w2 = T.load_weights(str(tmp_path / "no_mean.npz"), mean_param_path=p)
    assert np.array_equal(np.asarray(w2["mean_param"]), want)
