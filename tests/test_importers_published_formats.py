"""The asset importers against files assembled from the PUBLISHED formats by code that shares
nothing with the importers or their own writer (tests/published_formats.py): protobuf messages serialised by the real
google.protobuf runtime, Snappy blocks compressed by the real Snappy (pyarrow), a bit-at-a-time CRC-32C, a LevelDB table
builder written from doc/table_format.md, and Python-2 protocol-2 pickle streams with chumpy / scipy.sparse.csc records.
TensorFlow 1.8 and chumpy cannot be installed here, so these are the closest thing to the reference's real
`hmmr_model.ckpt-1119816` / `neutral_smpl_with_cocoplus_reg.pkl` (tester.py:92-116, batch_smpl.py:22-87)."""
import os
import pickle
import struct
import warnings

import numpy as np
import pytest

from human_dynamics_amd import assets, tf_checkpoint as tc
from tests import published_formats as pf


# --------------------------------------------------------------------------------------------- known answers first
def test_crc32c_rfc3720_vectors_both_implementations():
    """RFC 3720 B.4 test vectors: the importer's table-driven CRC and the fixture's bitwise CRC are independent
    implementations and must both reproduce them."""
    vectors = [(b"\x00" * 32, 0x8A9136AA), (b"\xff" * 32, 0x62A8AB43), (bytes(range(32)), 0x46DD794E),
               (bytes(range(31, -1, -1)), 0x113FDB5C), (b"123456789", 0xE3069283)]
    for data, want in vectors:
        assert tc.crc32c(data) == want
        assert pf.crc32c_bitwise(data) == want
    # LevelDB's mask (crc32c.h): the importer's and the fixture's agree with the definition
    for data, _ in vectors:
        c = pf.crc32c_bitwise(data)
        assert tc._mask(c) == (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF == pf.masked_crc(data)


@pytest.mark.parametrize("seed", range(4))
def test_snappy_decoder_against_the_real_compressor(seed):
    """snappy_decompress on streams produced by the real Snappy library (inside pyarrow): literals of every length
    class (1-byte, 2-byte length), copies with 1-byte and 2-byte offsets, overlapping copies (run-length)."""
    rng = np.random.default_rng(seed)
    words = [bytes(rng.integers(97, 123, size=int(n), dtype=np.uint8)) for n in rng.integers(3, 12, size=40)]
    payloads = [
        b"", b"a", b"ab" * 3, b"\x00" * 70000,                                         # run-length: overlapping copies, far beyond 64 KiB
        bytes(rng.integers(0, 256, size=5000, dtype=np.uint8)),                        # incompressible: long literals (2-byte length)
        b" ".join(words[int(i)] for i in rng.integers(0, 40, size=3000)),              # text-like: copy-1 / copy-2 mixes
        np.repeat(rng.normal(size=300).astype(np.float32), 7).tobytes(),               # tensor-like
    ]
    for raw in payloads:
        z = pf.snappy_real(raw)
        assert tc.snappy_decompress(z) == raw


# --------------------------------------------------------------------------------------------- the tensor bundle
def _bundle_tensors(rng):
    t = {
        "global_step": np.array(1119816, np.int64),                                    # 0-d int64
        "mean_param": rng.normal(size=(1, 85)).astype(np.float32),
        "resnet_v2_50/conv1/weights": rng.normal(size=(7, 7, 3, 64)).astype(np.float32),
        "resnet_v2_50/conv1/biases": rng.normal(size=64).astype(np.float32),
        "single_view_ief/3D_module/fc1/weights": rng.normal(size=(2133, 16)).astype(np.float32),
        "AZ_FC_block2_conv1block_0/weights": rng.normal(size=(3, 1, 16, 16)).astype(np.float32),
        "parents_like_int32": np.arange(24, dtype=np.int32),
        "double_stat": rng.normal(size=(3, 2)),                                        # float64
    }
    for b in range(1, 5):                                                              # many keys with long shared prefixes
        for u in range(1, 7):
            for leaf in ("gamma", "beta", "moving_mean", "moving_variance"):
                t["resnet_v2_50/block%d/unit_%d/bottleneck_v2/preact/%s" % (b, u, leaf)] = \
                    rng.normal(size=8 * b).astype(np.float32)
    return t


@pytest.mark.parametrize("num_shards,block_size,compress", [(1, 4096, True), (2, 1024, True), (3, 512, False)])
def test_reader_on_a_bundle_assembled_from_the_published_layout(tmp_path, num_shards, block_size, compress):
    rng = np.random.default_rng(num_shards)
    tensors = _bundle_tensors(rng)
    prefix = str(tmp_path / "hmmr_model.ckpt-1119816")
    n_snappy = pf.write_bundle(prefix, tensors, num_shards=num_shards, block_size=block_size, compress=compress)
    assert (n_snappy > 0) == compress, "the fixture must exercise Snappy-compressed index blocks when asked to"
    assert sorted(os.listdir(tmp_path)) == sorted(
        ["hmmr_model.ckpt-1119816.index"] +
        ["hmmr_model.ckpt-1119816.data-%05d-of-%05d" % (k, num_shards) for k in range(num_shards)])
    assert tc.is_checkpoint(prefix)
    header, entries = tc.read_index(prefix + ".index")
    assert header["num_shards"] == num_shards and sorted(entries) == sorted(tensors)
    assert {e["shard_id"] for e in entries.values()} == set(range(num_shards))
    got = tc.read_checkpoint(prefix, verify_data=True)                                 # per-tensor masked crc32c checked too
    assert sorted(got) == sorted(tensors)
    for k, v in tensors.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape and np.array_equal(got[k], v), k
    assert got["global_step"].shape == () and int(got["global_step"]) == 1119816


def test_reader_rejects_a_damaged_published_bundle(tmp_path):
    rng = np.random.default_rng(0)
    prefix = str(tmp_path / "m.ckpt-1")
    pf.write_bundle(prefix, _bundle_tensors(rng), num_shards=2, block_size=1024)
    raw = bytearray(open(prefix + ".index", "rb").read())
    raw[100] ^= 0x40                                                                   # inside a (compressed) data block
    open(prefix + ".index", "wb").write(bytes(raw))
    with pytest.raises(ValueError):
        tc.read_checkpoint(prefix)
    raw[100] ^= 0x40
    raw[-3] ^= 0x01                                                                    # table magic
    open(prefix + ".index", "wb").write(bytes(raw))
    with pytest.raises(ValueError):
        tc.read_index(prefix + ".index")
    raw[-3] ^= 0x01
    open(prefix + ".index", "wb").write(bytes(raw))
    shard = bytearray(open(prefix + ".data-00001-of-00002", "rb").read())
    shard[17] ^= 0xFF                                                                  # a tensor byte: caught by the entry's crc32c
    open(prefix + ".data-00001-of-00002", "wb").write(bytes(shard))
    with pytest.raises(ValueError):
        tc.read_checkpoint(prefix, verify_data=True)


def test_own_writer_is_readable_by_an_independent_parser(tmp_path):
    """The other direction: what `tf_checkpoint.write_checkpoint` emits is parsed here by google.protobuf + a
    straight-line table walk, so writer and reader cannot share a misreading of the format."""
    rng = np.random.default_rng(1)
    tensors = {"a/b": rng.normal(size=(2, 3)).astype(np.float32), "a/c": np.array(7, np.int64),
               "z": rng.normal(size=5).astype(np.float32)}
    prefix = str(tmp_path / "w")
    tc.write_checkpoint(prefix, tensors)
    Header, Entry = pf.bundle_messages()
    data = open(prefix + ".index", "rb").read()
    assert struct.unpack("<Q", data[-8:])[0] == pf.TABLE_MAGIC

    def rd_varint(buf, pos):
        v = shift = 0
        while True:
            b = buf[pos]; pos += 1
            v |= (b & 0x7F) << shift
            if b < 0x80:
                return v, pos
            shift += 7

    def block(off, size):
        assert data[off + size] == 0                                                   # uncompressed
        assert struct.unpack("<I", data[off + size + 1:off + size + 5])[0] == pf.masked_crc(data[off:off + size + 1])
        blk = data[off:off + size]
        nrest = struct.unpack("<I", blk[-4:])[0]
        end, pos, key, out = len(blk) - 4 - 4 * nrest, 0, b"", []
        while pos < end:
            sh, pos = rd_varint(blk, pos); ns, pos = rd_varint(blk, pos); vl, pos = rd_varint(blk, pos)
            key = key[:sh] + blk[pos:pos + ns]; pos += ns
            out.append((key, blk[pos:pos + vl])); pos += vl
        return out

    foot = data[-48:]
    _, p = rd_varint(foot, 0); _, p = rd_varint(foot, p)
    ioff, p = rd_varint(foot, p); isz, p = rd_varint(foot, p)
    found = {}
    for _, handle in block(ioff, isz):
        boff, q = rd_varint(handle, 0); bsz, _ = rd_varint(handle, q)
        for k, v in block(boff, bsz):
            found[k] = v
    h = Header(); h.ParseFromString(found.pop(b""))
    assert h.num_shards == 1 and h.endianness == 0
    shard = open(prefix + ".data-00000-of-00001", "rb").read()
    assert sorted(found) == sorted(k.encode() for k in tensors)
    for k, v in tensors.items():
        e = Entry(); e.ParseFromString(found[k.encode()])
        raw = shard[e.offset:e.offset + e.size]
        assert [d.size for d in e.shape.dim] == list(v.shape) and e.crc32c == pf.masked_crc(raw)
        assert np.array_equal(np.frombuffer(raw, v.dtype).reshape(v.shape), v)


def test_load_weights_through_a_published_format_checkpoint(tmp_path):
    """Tester's weight loading (evaluation/tester.load_weights) from a two-shard, Snappy-compressed bundle that holds
    every variable of the model (SURVEY App. B names) + the ResNet-only checkpoint override (tester.py:99-112)."""
    from human_dynamics_amd.evaluation.tester import load_weights
    # (tensors under 64 KB only: the fixture's CRC is bit-at-a-time Python)
    w = {k: v for k, v in assets.make_synthetic_weights(3).items() if v.nbytes < 65536}
    assert "mean_param" in w and "resnet_v2_50/conv1/weights" in w and len(w) > 200
    pf.write_bundle(str(tmp_path / "hmmr_model.ckpt-1119816"), w, num_shards=2, block_size=4096)
    r = {"resnet_v2_50/conv1/biases": np.full(64, 2.5, np.float32), "global_step": np.array(642561, np.int64)}
    pf.write_bundle(str(tmp_path / "hmr_noS5.ckpt-642561"), r, num_shards=1)
    got = load_weights(str(tmp_path / "hmmr_model.ckpt-1119816"), str(tmp_path / "hmr_noS5.ckpt-642561"))
    assert np.array_equal(got["mean_param"], w["mean_param"])
    assert np.array_equal(got["resnet_v2_50/conv1/biases"], r["resnet_v2_50/conv1/biases"])
    assert sorted(got) == sorted(w)
    for k in ("single_view_ief/3D_module/fc3/biases", "resnet_v2_50/block1/unit_1/bottleneck_v2/conv1/weights",
              "AZ_FC_block_preact_gn1block_2/gamma"):
        assert np.array_equal(got[k], w[k]), k


# --------------------------------------------------------------------------------------------- mean theta (the .h5's role)
def test_mean_theta_from_npy_when_the_checkpoint_has_none(tmp_path):
    """tester.py:118-152 initialises `mean_param` from neutral_smpl_meanwjoints.h5 (deepdish / blosc: not readable here);
    `load_weights(..., mean_param_path=)` takes the same 85 numbers as .npy / .npz, in the file's layout
    (pose [72], shape [10]) or already assembled [85], and applies the reference's assembly: cam = [0.9, 0, 0],
    pose[:3] = [pi, 0, 0] (load_mean_params, tester.py:123-135)."""
    from human_dynamics_amd.evaluation.tester import load_weights, mean_theta_from_file
    w = {k: v for k, v in assets.make_synthetic_weights(3).items()
         if (k.startswith("single_view_ief/") and v.nbytes < 65536) or k == "mean_param"}
    ck = dict(w); del ck["mean_param"]
    pf.write_bundle(str(tmp_path / "m.ckpt-1"), ck, num_shards=1)
    rng = np.random.default_rng(5)
    pose, shape = rng.normal(size=72).astype(np.float32) * 0.2, rng.normal(size=10).astype(np.float32)
    np.savez(str(tmp_path / "neutral_smpl_meanwjoints.npz"), pose=pose, shape=shape)
    got = load_weights(str(tmp_path / "m.ckpt-1"), mean_param_path=str(tmp_path / "neutral_smpl_meanwjoints.npz"))
    m = got["mean_param"]
    assert m.shape == (1, 85) and m.dtype == np.float32
    assert np.allclose(m[0, :3], [0.9, 0, 0]) and m[0, 3] == np.float32(np.pi) and not m[0, 4:6].any()
    assert np.array_equal(m[0, 6:75], pose[3:]) and np.array_equal(m[0, 75:], shape)
    np.save(str(tmp_path / "mean85.npy"), w["mean_param"])
    assert np.array_equal(mean_theta_from_file(str(tmp_path / "mean85.npy")), np.asarray(w["mean_param"], np.float32).reshape(1, 85))
    # a checkpoint that carries mean_param wins (Saver.restore overwrites the initialiser, tester.py:114-116)
    pf.write_bundle(str(tmp_path / "full.ckpt-2"), w, num_shards=1)
    both = load_weights(str(tmp_path / "full.ckpt-2"), mean_param_path=str(tmp_path / "neutral_smpl_meanwjoints.npz"))
    assert np.array_equal(both["mean_param"], w["mean_param"])
    with pytest.raises(FileNotFoundError):
        load_weights(str(tmp_path / "m.ckpt-1"), mean_param_path=str(tmp_path / "nope.npy"))


# --------------------------------------------------------------------------------------------- the SMPL pickle
def _py2_smpl_pickle(consts):
    import scipy.sparse as sp
    nv = consts["v_template"].shape[0]
    p = pf.Py2Pickle()
    p.dict_({
        "v_template": pf.Chumpy(consts["v_template"].astype(np.float64)),
        "shapedirs": pf.Chumpy(consts["shapedirs"].T.reshape(nv, 3, 10).astype(np.float64)),
        "posedirs": np.ascontiguousarray(consts["posedirs"].T.reshape(nv, 3, 207).astype(np.float64)),   # plain ndarray in the models
        "J_regressor": sp.csc_matrix(consts["J_regressor"].T.astype(np.float64)),
        "cocoplus_regressor": sp.csc_matrix(consts["cocoplus_regressor"].T.astype(np.float64)),
        "weights": pf.Chumpy(consts["lbs_weights"].astype(np.float64)),
        "kintree_table": np.stack([np.where(assets.SMPL_PARENTS < 0, 2 ** 32 - 1, assets.SMPL_PARENTS).astype(np.uint32),
                                   np.arange(24, dtype=np.uint32)]),
        "bs_type": "lrotmin", "bs_style": "lbs",
        "J": pf.Chumpy(np.zeros((24, 3))),
    })
    return p.done()


def test_smpl_pickle_in_the_python2_chumpy_wire_format(tmp_path, smpl_consts):
    stream = _py2_smpl_pickle(smpl_consts)
    # the stream really is the Python-2 flavour: a plain Python-3 load cannot read it (no chumpy module; and without
    # encoding='latin1' the raw array bytes are not even decodable)
    with pytest.raises(Exception):
        pickle.loads(stream)
    import pickletools
    ops = {op.name for op, _, _ in pickletools.genops(stream)}
    assert {"NEWOBJ", "BUILD", "SHORT_BINSTRING", "BINSTRING", "GLOBAL", "REDUCE"} <= ops and "BINUNICODE" not in ops
    path = str(tmp_path / "neutral_smpl_with_cocoplus_reg.pkl")
    open(path, "wb").write(stream)
    from human_dynamics_amd.tf_smpl.batch_smpl import load_smpl_constants
    with warnings.catch_warnings():
        warnings.simplefilter("error")                     # the deprecated scipy.sparse.csc module path must not be imported
        got = load_smpl_constants(path)
    for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "cocoplus_regressor", "lbs_weights"):
        assert got[k].dtype == np.float32 and got[k].shape == smpl_consts[k].shape, k
        assert np.allclose(got[k], smpl_consts[k], atol=1e-7), k
    assert got["parents"].dtype == np.int32 and got["parents"][0] == -1
    assert np.array_equal(got["parents"][1:], assets.SMPL_PARENTS[1:])
