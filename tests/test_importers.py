"""Asset importers: TF checkpoint-V2 reader and chumpy-free SMPL pickle
loader, round-tripped against writers of the same published formats (TensorFlow / chumpy are not
installable here, so real files cannot be part of the fixtures)."""
import os
import pickle
import sys
import types

import numpy as np
import pytest

from human_dynamics_amd import assets, tf_checkpoint as tc


def test_crc32c_and_snappy_known_answers():
    assert tc.crc32c(b"123456789") == 0xE3069283                      # RFC 3720 check value
    assert tc._mask(0) == 0xA282EAD8
    # snappy: literal "abcd" + copy(offset 4, len 8) -> "abcdabcdabcd"
    stream = bytes([12, (3 << 2) | 0]) + b"abcd" + bytes([((8 - 4) << 2) | 1, 4])
    assert tc.snappy_decompress(stream) == b"abcdabcdabcd"


def test_checkpoint_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    tensors = {
        "resnet_v2_50/conv1/weights": rng.normal(size=(7, 7, 3, 64)).astype(np.float32),
        "resnet_v2_50/conv1/biases": rng.normal(size=64).astype(np.float32),
        "mean_param": rng.normal(size=(1, 85)).astype(np.float32),
        "global_step": np.array(1119816, np.int64),
        "AZ_FC_block2_conv1block_0/weights": rng.normal(size=(3, 1, 16, 16)).astype(np.float32),
    }
    for i in range(40):                                                 # several index blocks, shared key prefixes
        tensors["resnet_v2_50/block1/unit_%d/bottleneck_v2/conv1/weights" % i] = rng.normal(size=(1, 1, 4, 4)).astype(np.float32)
    prefix = str(tmp_path / "model.ckpt-7")
    tc.write_checkpoint(prefix, tensors)
    assert tc.is_checkpoint(prefix) and not tc.is_checkpoint(prefix + "x")
    got = tc.read_checkpoint(prefix, verify_data=True)
    assert sorted(got) == sorted(tensors)
    for k, v in tensors.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape and np.array_equal(got[k], v), k
    only = tc.read_checkpoint(prefix, names={"mean_param"})
    assert list(only) == ["mean_param"]


def test_checkpoint_detects_corruption(tmp_path):
    prefix = str(tmp_path / "m")
    tc.write_checkpoint(prefix, {"a": np.arange(6, dtype=np.float32).reshape(2, 3)})
    raw = bytearray(open(prefix + ".index", "rb").read())
    raw[3] ^= 0xFF
    open(prefix + ".index", "wb").write(bytes(raw))
    with pytest.raises(ValueError):
        tc.read_checkpoint(prefix)
    open(prefix + ".index", "wb").write(b"not a table" * 10)
    with pytest.raises(ValueError):
        tc.read_index(prefix + ".index")


def test_load_weights_reads_a_checkpoint_prefix(tmp_path):
    from human_dynamics_amd.evaluation.tester import load_weights
    w = {k: v for k, v in assets.make_synthetic_weights(3).items() if "block4" not in k and "AZ_FC" not in k}
    tc.write_checkpoint(str(tmp_path / "hmmr.ckpt-1"), w)
    r = {"resnet_v2_50/conv1/biases": np.full(64, 2.5, np.float32), "single_view_ief/other": np.zeros(3, np.float32)}
    tc.write_checkpoint(str(tmp_path / "hmr.ckpt-2"), r)
    got = load_weights(str(tmp_path / "hmmr.ckpt-1"), str(tmp_path / "hmr.ckpt-2"))
    assert np.array_equal(got["mean_param"], w["mean_param"])
    assert np.array_equal(got["resnet_v2_50/conv1/biases"], r["resnet_v2_50/conv1/biases"])   # resnet vars overridden
    assert "single_view_ief/other" not in got
    with pytest.raises(FileNotFoundError):
        load_weights(str(tmp_path / "missing.ckpt-3"))


def test_smpl_pickle_loads_without_chumpy(tmp_path, smpl_consts):
    """Build a pickle shaped like the SMPL model files (chumpy leaves + scipy sparse regressors)
    with a throw-away fake `chumpy` module, then load it with NO chumpy importable."""
    import scipy.sparse as sp
    from human_dynamics_amd.tf_smpl.batch_smpl import load_smpl_constants
    mod = types.ModuleType("chumpy"); sub = types.ModuleType("chumpy.ch")

    class Ch(object):
        def __init__(self, x):
            self.x = np.asarray(x)
    Ch.__module__, Ch.__qualname__ = "chumpy.ch", "Ch"
    sub.Ch = Ch; mod.ch = sub
    sys.modules["chumpy"], sys.modules["chumpy.ch"] = mod, sub
    try:
        nv = 6890
        dd = {
            "v_template": Ch(smpl_consts["v_template"].astype(np.float64)),
            "shapedirs": Ch(smpl_consts["shapedirs"].T.reshape(nv, 3, 10).astype(np.float64)),
            "posedirs": Ch(smpl_consts["posedirs"].T.reshape(nv, 3, 207).astype(np.float64)),
            "J_regressor": sp.csc_matrix(smpl_consts["J_regressor"].T.astype(np.float64)),
            "cocoplus_regressor": sp.csc_matrix(smpl_consts["cocoplus_regressor"].T.astype(np.float64)),
            "weights": Ch(smpl_consts["lbs_weights"].astype(np.float64)),
            "kintree_table": np.stack([np.where(assets.SMPL_PARENTS < 0, 2 ** 32 - 1, assets.SMPL_PARENTS).astype(np.uint32),
                                       np.arange(24, dtype=np.uint32)]),
        }
        path = str(tmp_path / "neutral_smpl_with_cocoplus_reg.pkl")
        with open(path, "wb") as f:
            pickle.dump(dd, f, protocol=2)
    finally:
        del sys.modules["chumpy"], sys.modules["chumpy.ch"]
    got = load_smpl_constants(path)
    for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "cocoplus_regressor", "lbs_weights"):
        assert got[k].shape == smpl_consts[k].shape and np.allclose(got[k], smpl_consts[k]), k
    assert np.array_equal(got["parents"][1:], assets.SMPL_PARENTS[1:]) and got["parents"][0] == -1
    # a checkpoint that carries the six SMPL variables overrides the pkl (Saver.restore overwrites the tf.Variables the
    # reference initialised from the pkl): only the kinematic tree still comes from the pkl
    ck = {k: (smpl_consts[k] * 1.5).astype(np.float32) for k in ("v_template", "shapedirs", "J_regressor", "posedirs",
                                                                 "lbs_weights", "cocoplus_regressor")}
    both = load_smpl_constants(path, checkpoint_vars=ck)
    assert np.array_equal(both["v_template"], ck["v_template"]) and np.array_equal(both["posedirs"], ck["posedirs"])
    assert np.array_equal(both["parents"], got["parents"])
    part = load_smpl_constants(path, checkpoint_vars={"v_template": ck["v_template"]})       # incomplete: the pkl stands
    assert np.allclose(part["v_template"], smpl_consts["v_template"])


def test_smpl_constants_fall_back_to_checkpoint_variables(smpl_consts):
    from human_dynamics_amd.tf_smpl.batch_smpl import load_smpl_constants
    ck = {k: smpl_consts[k] for k in ("v_template", "shapedirs", "J_regressor", "posedirs", "lbs_weights",
                                      "cocoplus_regressor")}
    got = load_smpl_constants("/nonexistent/smpl.pkl", checkpoint_vars=ck)
    assert np.array_equal(got["parents"], assets.SMPL_PARENTS) and got["posedirs"].shape == (207, 20670)
    with pytest.raises(FileNotFoundError):
        load_smpl_constants("/nonexistent/smpl.pkl")
