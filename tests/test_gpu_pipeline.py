"""GPU parity of the whole path behind the reference's Tester surface."""
import os

import numpy as np
import pytest

from conftest import Config
from human_dynamics_amd import assets

pytestmark = pytest.mark.gpu
VSUB = 16


def _check(res, gold, tol, keys=None):
    errs = {}
    for k, g in gold.items():
        if k in ("phi", "strips", "omegas_all"):
            continue
        if k.endswith("_sub"):
            base = k[:-4]
            got = res[base][..., ::VSUB, :]
        else:
            got = res[k]
        assert got.shape == g.shape, (k, got.shape, g.shape)
        errs[k] = float(np.abs(got - g).max())
    print("max-abs-err vs golden:", {k: "%.2e" % v for k, v in sorted(errs.items())})
    for k, e in errs.items():
        if keys is None or k in keys:
            assert e < tol, (k, e)
    return errs


@pytest.fixture(scope="module")
def tester_f32(weights, smpl_consts, gpu_device):
    from human_dynamics_amd.evaluation.tester import Tester
    return Tester(Config(batch_size=1), weights=weights, smpl=smpl_consts, dtype="f32", device=gpu_device)


def test_predict_fp32_matches_golden_window(tester_f32, golden_window):
    """BASELINE config 1 (PR1 golden): verts/joints within 1e-4 of the reference-graph oracle."""
    frames = assets.make_synthetic_frames(20, seed=1)
    res = tester_f32.predict(frames[None])
    assert sorted(res) == sorted(k + s for k in ("cams", "joints", "kps", "poses", "shapes", "verts", "omegas")
                                 for s in ("", "_delta"))
    assert res["verts"].shape == (1, 20, 6890, 3) and res["verts_delta"].shape == (1, 20, 2, 6890, 3)
    assert all(v.dtype == np.float32 for v in res.values())
    _check(res, golden_window, 1e-4)


def test_predict_all_images_fp32_matches_golden_video(weights, smpl_consts, gpu_device, golden_video):
    from human_dynamics_amd.evaluation.tester import Tester
    frames = assets.make_synthetic_frames(24, seed=7)
    t = Tester(Config(batch_size=2), weights=weights, smpl=smpl_consts, dtype="f32", device=gpu_device)
    res = t.predict_all_images(frames)
    assert res["verts"].shape == (24, 6890, 3) and res["omegas_delta"].shape == (24, 2, 85)
    gold = dict(golden_video)
    gold["verts_sub"] = gold["verts_sub"]
    _check(res, gold, 1e-4)
    # the literal schedule (every window through the ResNet) gives bit-identical results
    t_lit = Tester(Config(batch_size=2), weights=weights, smpl=smpl_consts, dtype="f32", device=gpu_device,
                   dedup=False)
    lit = t_lit.predict_all_images(frames)
    for k in res:
        assert np.array_equal(res[k], lit[k]), k


def test_predict_bf16_reports_error(weights, smpl_consts, gpu_device, golden_window):
    """bf16-operand mode is the throughput mode: its end-to-end error is reported,
    the 1e-4 bar applies to the fp32 mode and to the SMPL stage."""
    from human_dynamics_amd.evaluation.tester import Tester
    frames = assets.make_synthetic_frames(20, seed=1)
    t = Tester(Config(batch_size=1), weights=weights, smpl=smpl_consts, dtype="bf16", device=gpu_device)
    res = t.predict(frames[None])
    errs = _check(res, golden_window, 0.25, keys=("verts", "joints"))
    # SMPL stage alone, fed the bf16 path's own omega, still meets 1e-4
    from oracle import hmmr_oracle as O
    import torch
    om = res["omegas"][0]
    rv, rj, _ = O.smpl_forward(om[:, 75:], om[:, 3:75], smpl_consts, torch.float64)
    assert np.abs(res["verts"][0] - rv.numpy()).max() < 1e-4
    assert np.abs(res["joints"][0] - rj.numpy()).max() < 1e-4
    assert errs["verts"] < 0.25


def test_feature_extractor_zero_pads_tail(weights, gpu_device, golden_window):
    from human_dynamics_amd.datasets.resnet_extractor import FeatureExtractor
    fe = FeatureExtractor("synthetic:0", batch_size=4, weights=weights, dtype="f32", device=gpu_device)
    frames = assets.make_synthetic_frames(6, seed=1)
    phis = fe.compute_all_phis(frames)
    assert phis.shape == (6, 2048)
    assert np.abs(phis - golden_window["phi"][:6]).max() < 1e-4


def test_tester_rejects_bad_config(weights, smpl_consts, gpu_device):
    from human_dynamics_amd.evaluation.tester import Tester
    with pytest.raises(Exception):
        Tester(Config(load_path=""), smpl=smpl_consts, device=gpu_device)
    with pytest.raises(Exception):
        Tester(Config(pred_mode="bogus"), weights=weights, smpl=smpl_consts, device=gpu_device)
    with pytest.raises(FileNotFoundError):
        Tester(Config(load_path="/nonexistent/model.ckpt-1"), smpl=smpl_consts, device=gpu_device)


def test_container_route_equals_record_route(weights, smpl_consts, gpu_device):
    """The reference-shaped OmegasPred route and the in-place record route are the same numbers."""
    from human_dynamics_amd.evaluation.tester import Tester
    frames = assets.make_synthetic_frames(20, seed=3).reshape(2, 10, 224, 224, 3)
    a = Tester(Config(batch_size=2, sequence_length=10), weights=weights, smpl=smpl_consts, dtype="f32",
               device=gpu_device).predict(frames)
    b = Tester(Config(batch_size=2, sequence_length=10), weights=weights, smpl=smpl_consts, dtype="f32",
               device=gpu_device, use_containers=True).predict(frames)
    assert sorted(a) == sorted(b)
    for k in a:
        assert a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), k


def test_resnet_zero_tail_equals_explicit_zero_images(weights, gpu_device):
    from human_dynamics_amd.engine import HmmrEngine
    eng = HmmrEngine(weights, None, dtype="bf16", device=gpu_device)
    frames = assets.make_synthetic_frames(3, seed=4)
    a = eng.resnet(frames, n_zero=2).cpu().numpy()
    b = eng.resnet(np.concatenate([frames, np.zeros((2, 224, 224, 3), np.float32)])).cpu().numpy()
    assert a.shape == (5, 2048) and np.array_equal(a, b)
    c = eng.resnet(frames[:0], n_zero=1).cpu().numpy()
    assert np.array_equal(c, a[3:4])


def test_hal_mode_matches_oracle(smpl_consts, gpu_device):
    """pred_mode 'hal' (tester.py:189-190): the hallucinator fc2_res replaces the temporal encoder."""
    import torch
    from human_dynamics_amd.evaluation.tester import Tester
    from oracle import hmmr_oracle as O
    w = assets.make_synthetic_weights(0, with_hallucinator=True)
    frames = assets.make_synthetic_frames(8, seed=21).reshape(2, 4, 224, 224, 3)
    t = Tester(Config(batch_size=2, sequence_length=4, pred_mode="hal"), weights=w, smpl=smpl_consts,
               dtype="f32", device=gpu_device)
    got = t.predict(frames)
    ref = O.OracleTester(w, smpl_consts, batch_size=2, sequence_length=4, pred_mode="hal",
                         dtype=torch.float64).predict(frames)
    for k in ("omegas", "verts", "joints", "verts_delta", "kps"):
        assert np.abs(got[k] - ref[k]).max() < 1e-4, k
    # and the stage alone
    phi = np.random.default_rng(0).normal(size=(3, 5, 2048)).astype(np.float32)
    out = t.engine.hallucinate(phi).cpu().numpy()
    ref = O.fc2_res(phi, w, torch.float64).numpy()
    assert np.abs(out - ref).max() < 1e-4


def test_sharded_predictor_graph_replay_equals_eager(weights, smpl_consts, gpu_device):
    """hipGraph replay of the local pass == eager launches, and == Tester.predict_all_images."""
    import torch
    from human_dynamics_amd import dist as hd
    from human_dynamics_amd.evaluation.tester import Tester
    t = Tester(Config(batch_size=2), weights=weights, smpl=smpl_consts, dtype="bf16", device=gpu_device)
    frames = assets.make_synthetic_frames(24, seed=7)
    ref = t.predict_all_images(frames)
    dev = torch.from_numpy(frames).to(gpu_device)
    eager = hd.ShardedPredictor(t, 24, 0, 1).run(dev).clone()
    sp = hd.ShardedPredictor(t, 24, 0, 1, use_graph=True)
    for _ in range(4):                       # 2 eager warm-ups, capture, replay
        out = sp.run(dev)
    assert sp.graph is not None
    torch.cuda.synchronize()
    assert torch.equal(out, eager)
    rec = hd.unpack_outputs(out, sp.layout)
    for k in ("verts", "omegas", "joints_delta"):
        assert np.array_equal(rec[k].cpu().numpy(), ref[k]), k
    # two ranks' shards tile the single-rank result bit-exactly (same kernels, same per-window math)
    parts = []
    for r in range(2):
        p = hd.ShardedPredictor(t, 24, r, 2, gather_mode="records")      # (the local RECORDS of each rank: the default for one video gathers omegas)
        loc = p.run(dev[p.plan.f0:p.plan.f1], gather=False)
        parts.append(loc[:p.plan.o1 - p.plan.o0])
    assert torch.equal(torch.cat(parts, 0), eager)


@pytest.mark.parametrize("dt,step_streams", [("bf16", False), ("bf16", True), ("f16x3", True)])
def test_sharded_predictor_two_stream_pipeline_equals_serial(weights, smpl_consts, gpu_device, dt, step_streams):
    """pipeline=True (the tail of call k on a second stream under the ResNet of call k+1; step_streams: the ResNet passes of
    consecutive calls on alternating streams, two in flight) returns bit-identical records for a stream of different inputs."""
    import torch
    from human_dynamics_amd import dist as hd
    from human_dynamics_amd.evaluation.tester import Tester
    t = Tester(Config(batch_size=2), weights=weights, smpl=smpl_consts, dtype=dt, device=gpu_device)
    clips = [torch.from_numpy(assets.make_synthetic_frames(24, seed=20 + i)).to(gpu_device) for i in range(5)]
    serial = hd.ShardedPredictor(t, 24, 0, 1)
    want = [serial.run(c).clone() for c in clips]
    pipe = hd.ShardedPredictor(t, 24, 0, 1, pipeline=True, step_streams=step_streams)
    assert (pipe.step_streams is not None) == step_streams
    got, prev = [], None
    for c in clips:                          # read result k only after call k+1 has been queued
        cur = pipe.run(c)
        if prev is not None:
            got.append(pipe.ready(prev).clone())
        prev = cur
    got.append(pipe.ready(prev).clone())
    pipe.finish()
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(got, want)):
        assert torch.equal(a, b), i


def test_resnet_concurrent_half_batches_equal_one_pass(weights, gpu_device, monkeypatch):
    """engine.resnet splits large batches over two HIP streams: same bits as one launch sequence."""
    import torch
    from human_dynamics_amd.engine import HmmrEngine
    x = torch.from_numpy(assets.make_synthetic_frames(179, seed=9)).to(gpu_device)
    monkeypatch.setenv("HMMR_RESNET_STREAMS", "1")
    one = HmmrEngine(weights, None, dtype="bf16", device=gpu_device)
    assert one.resnet_streams == 1
    ref = one.resnet(x, n_zero=1)
    monkeypatch.setenv("HMMR_RESNET_STREAMS", "2")
    two = HmmrEngine(weights, None, dtype="bf16", device=gpu_device)
    for _ in range(3):
        got = two.resnet(x, n_zero=1)
        assert torch.equal(got, ref)
    assert len(two._side_streams) == 2 and "resnet1" in two._ws
    three = HmmrEngine(weights, None, dtype="bf16", device=gpu_device)
    three.resnet_streams = 3
    assert torch.equal(three.resnet(x, n_zero=0), ref[:179])


def test_pipelined_predictor_with_asynchronous_gather(weights, smpl_consts, gpu_device, monkeypatch):
    """The N > 1 bench path on one GPU: rank 0 of a 2-rank plan in pipeline mode, with the RCCL all-gather
    replaced by a stand-in that behaves like it (runs on its own stream after the issuing stream, returns a
    Work whose wait() orders the caller's stream behind it).  Checks the stream choreography of
    ShardedPredictor._run_pipelined / ready / finish, which the gloo tests (CPU) cannot reach."""
    import torch
    from human_dynamics_amd import dist as hd
    from human_dynamics_amd.evaluation.tester import Tester
    t = Tester(Config(batch_size=2), weights=weights, smpl=smpl_consts, dtype="bf16", device=gpu_device)
    n = 48
    clips = [torch.from_numpy(assets.make_synthetic_frames(n, seed=40 + i)).to(gpu_device) for i in range(4)]
    whole = hd.ShardedPredictor(t, n, 0, 1)
    want = [whole.run(c).clone() for c in clips]
    comm = torch.cuda.Stream(device=gpu_device)
    calls = []

    class Work(object):
        def __init__(self, ev):
            self.ev = ev

        def wait(self):
            torch.cuda.current_stream().wait_event(self.ev)

    def fake_all_gather(full, local, group=None, async_op=False):
        comm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(comm):
            torch.cuda._sleep(2000000)                     # the gather takes a while
            full[:local.shape[0]].copy_(local)             # rank 0's block; rank 1's block stays as it is
            ev = torch.cuda.Event()
            ev.record(comm)
        calls.append(async_op)
        return Work(ev)
    monkeypatch.setattr(hd.dist, "all_gather_into_tensor", fake_all_gather)
    sp = hd.ShardedPredictor(t, n, 0, 2, pipeline=True, overlap_gather=True)
    p = sp.plan
    got, prev = [], None
    for c in clips:
        cur = sp.run(c[p.f0:p.f1])
        if prev is not None:
            got.append(sp.ready(prev)[:p.o1 - p.o0].clone())
        prev = cur
    sp.finish()
    got.append(prev[:p.o1 - p.o0].clone())
    torch.cuda.synchronize()
    assert calls == [True] * 4
    for i, (a, b) in enumerate(zip(got, want)):
        assert torch.equal(a, b[p.o0:p.o1]), i


def test_resnet_fused_bottleneck_tails_equal_layer_per_launch(weights, gpu_device, monkeypatch):
    """The conv3 -> next conv1 fusion of blocks 1-2 (hmmr_bottleneck_tail) leaves every feature bit unchanged."""
    import torch
    from human_dynamics_amd.engine import HmmrEngine
    x = torch.from_numpy(assets.make_synthetic_frames(9, seed=12)).to(gpu_device)
    monkeypatch.setenv("HMMR_FUSE_TAIL", "0")
    plain = HmmrEngine(weights, None, dtype="bf16", device=gpu_device)
    assert sum(plain.rw.unit[i].fuse_tail for i in range(16)) == 0
    ref = plain.resnet(x, n_zero=1)
    monkeypatch.setenv("HMMR_FUSE_TAIL", "1")
    fused = HmmrEngine(weights, None, dtype="bf16", device=gpu_device)
    assert [fused.rw.unit[i].fuse_tail for i in range(8)] == [3, 2, 4, 2, 2, 2, 4, 0]
    assert torch.equal(fused.resnet(x, n_zero=1), ref)
    f32 = HmmrEngine(weights, None, dtype="f32", device=gpu_device)
    assert sum(f32.rw.unit[i].fuse_tail for i in range(16)) == 0


@pytest.mark.parametrize("dt", ["bf16", "f32"])
def test_resnet_shortcut_and_conv1_as_one_gemm(dt, weights, gpu_device, monkeypatch):
    """The first unit of every block runs its conv shortcut and conv1 as one column-split GEMM: same bits."""
    import torch
    from human_dynamics_amd.engine import HmmrEngine
    x = torch.from_numpy(assets.make_synthetic_frames(7, seed=15)).to(gpu_device)
    monkeypatch.setenv("HMMR_FUSE_SC", "0")
    plain = HmmrEngine(weights, None, dtype=dt, device=gpu_device)
    assert not any(plain.rw.unit[i].sc_c1.w for i in range(16))
    ref = plain.resnet(x, n_zero=1)
    monkeypatch.setenv("HMMR_FUSE_SC", "all")
    fused = HmmrEngine(weights, None, dtype=dt, device=gpu_device)
    assert [bool(fused.rw.unit[i].sc_c1.w) for i in (0, 1, 3, 7, 13)] == [True, False, True, True, True]
    assert torch.equal(fused.resnet(x, n_zero=1), ref)
    monkeypatch.setenv("HMMR_FUSE_SC", "1")            # default: only where it was measured faster (blocks 3-4)
    dflt = HmmrEngine(weights, None, dtype=dt, device=gpu_device)
    assert [bool(dflt.rw.unit[i].sc_c1.w) for i in (0, 3, 7, 13)] == [False, False, True, True]
    assert torch.equal(dflt.resnet(x, n_zero=1), ref)


def test_stem_with_first_conv1_inside(weights, gpu_device):
    """The fused stem also computes block1/unit_1's conv1 on every pooled tile: same features, bit for bit.
    (HMMR_STEM_C1 is read once per process by the library, so the comparison runs in two subprocesses.)"""
    import subprocess
    import sys
    code = (
        "import sys, hashlib, torch; sys.path.insert(0, '.');"
        "from human_dynamics_amd import assets; from human_dynamics_amd.engine import HmmrEngine;"
        "e = HmmrEngine(assets.make_synthetic_weights(0), None, dtype='bf16', device='%s');"
        "x = torch.from_numpy(assets.make_synthetic_frames(5, seed=33)).to('%s');"
        "print('HASH', hashlib.sha1(e.resnet(x, n_zero=1).cpu().numpy().tobytes()).hexdigest())" % (gpu_device, gpu_device))
    out = {}
    for v in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HMMR_STEM_C1=v), capture_output=True, text=True,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        lines = [l for l in r.stdout.splitlines() if l.startswith("HASH")]
        assert lines, r.stderr[-500:]
        out[v] = lines[-1]
    assert out["0"] == out["1"]


@pytest.mark.parametrize("n,chunk", [(24, 256), (300, 128), (257, 256)])
def test_streamed_host_path_is_byte_identical_to_one_shot(weights, smpl_consts, gpu_device, n, chunk):
    """Tester.predict_all_images(host ndarray): the chunked three-stream pipeline (pinned copy-in / kernels /
    copy-out per key) returns exactly the bytes of the synchronous one-shot path, including chunk seams,
    a 1-frame last chunk and the zero-image padding of the first and last windows."""
    from human_dynamics_amd.evaluation.streaming import HostStreamer
    from human_dynamics_amd.evaluation.tester import Tester
    t = Tester(Config(batch_size=8), weights=weights, smpl=smpl_consts, dtype="bf16", device=gpu_device)
    frames = assets.make_synthetic_frames(n, seed=50 + n)
    ref = t.predict_all_images(frames, stream=False)
    t._streamer = HostStreamer(t, chunk=chunk)
    for rep in range(2):                        # second call re-uses the staging buffers
        got = t.predict_all_images(frames)
        assert sorted(got) == sorted(ref)
        for k in ref:
            assert got[k].dtype == np.float32 and got[k].shape == ref[k].shape, k
            assert np.array_equal(got[k], ref[k]), (k, rep)
    sub = t.predict_all_images(frames, want=("joints", "omegas_delta"))
    assert sorted(sub) == ["joints", "omegas_delta"]
    assert np.array_equal(sub["joints"], ref["joints"]) and np.array_equal(sub["omegas_delta"], ref["omegas_delta"])


def test_predict_videos_is_byte_identical_to_one_call_per_video(weights, smpl_consts, gpu_device):
    """Tester.predict_videos: several person tracks as ONE pipeline (track k+1 uploads under track k's ResNet; the reference calls
    predict_all_images once per track, demo_video.py:172, tester.py:229-312).  Every track's result equals its own one-shot call byte for
    byte: a track shorter than a chunk, one that ends in a partial chunk, an empty one, a two-chunk one; float32 and uint8 input."""
    from human_dynamics_amd.evaluation.tester import Tester
    t = Tester(Config(), weights=weights, smpl=smpl_consts, dtype="f16x3", device=gpu_device)
    t._streamer = None
    from human_dynamics_amd.evaluation.streaming import HostStreamer
    t._streamer = HostStreamer(t, chunk=64)
    lens = [37, 64, 0, 100, 8]
    vids = [assets.make_synthetic_frames(n, seed=40 + i) if n else np.zeros((0, 224, 224, 3), np.float32) for i, n in enumerate(lens)]
    got = t.predict_videos(vids)
    assert len(got) == len(vids)
    for v, g in zip(vids, got):
        ref = t.predict_all_images(v, stream=False) if len(v) else None
        for k in g:
            assert g[k].shape[0] == len(v)
            if len(v):
                assert np.array_equal(g[k], ref[k]), k
    sub = t.predict_videos(vids[:2], want=("joints", "omegas"))
    assert sorted(sub[0]) == ["joints", "omegas"] and np.array_equal(sub[1]["joints"], got[1]["joints"])
    u8 = [np.clip(np.rint((v + 1.0) * 127.5), 0, 255).astype(np.uint8) for v in vids[:2]]
    a = t.predict_videos(u8, want=("omegas",))
    for v, g in zip(u8, a):
        assert np.array_equal(g["omegas"], t.predict_all_images(v, want=("omegas",))["omegas"])


def test_streamed_host_path_without_delta_regressors(weights, smpl_consts, gpu_device):
    """config.delta_t_values = []: the record has no *_delta fields (tester.py:245-255 adds them per delta); the
    streamed path derives its keys from the record layout and equals the one-shot path, 640 frames = 3 chunks."""
    from human_dynamics_amd.evaluation.tester import Tester
    t = Tester(Config(batch_size=8, delta_t_values=[]), weights=weights, smpl=smpl_consts, dtype="bf16", device=gpu_device)
    frames = assets.make_synthetic_frames(70, seed=9)
    ref = t.predict_all_images(frames, stream=False)
    got = t.predict_all_images(frames)
    assert sorted(got) == sorted(ref) == sorted(["cams", "joints", "kps", "poses", "shapes", "verts", "omegas"])
    for k in ref:
        assert np.array_equal(got[k], ref[k]), k
    with pytest.raises(KeyError):
        t.predict_all_images(frames, want=("verts_delta",))


def test_streamed_uint8_input_matches_reference_normalisation(weights, smpl_consts, gpu_device):
    """uint8 crops are uploaded as bytes and normalised on the device with the reference's arithmetic
    ((x / 255.0 - 0.5) * 2 in float64, run_video.py:73): identical to feeding the float32 array."""
    from human_dynamics_amd.evaluation.tester import Tester
    t = Tester(Config(batch_size=8), weights=weights, smpl=smpl_consts, dtype="bf16", device=gpu_device)
    u8 = np.random.default_rng(3).integers(0, 256, size=(40, 224, 224, 3), dtype=np.uint8)
    as_float = ((u8 / 255.0 - 0.5) * 2).astype(np.float32)
    a = t.predict_all_images(u8, want=("omegas", "joints"))
    b = t.predict_all_images(as_float, want=("omegas", "joints"))
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def test_device_resident_video_is_chunked_without_changing_a_bit(weights, smpl_consts, gpu_device):
    """A long video that already sits in HBM is encoded MAX_DEVICE_FRAMES at a time and its windows go through
    the tail MAX_TAIL_WINDOWS at a time (bounded workspaces): same bits as one pass."""
    import torch
    from human_dynamics_amd.evaluation.tester import Tester
    t = Tester(Config(batch_size=8), weights=weights, smpl=smpl_consts, dtype="bf16", device=gpu_device)
    dev = torch.from_numpy(assets.make_synthetic_frames(50, seed=77)).to(gpu_device)
    ref = t.predict_all_images(dev)
    t.MAX_DEVICE_FRAMES, t.MAX_TAIL_WINDOWS = 16, 3
    got = t.predict_all_images(dev)
    for k in ref:
        assert np.array_equal(got[k], ref[k]), k


def test_batch_global_rigid_transformation_mirror(smpl_consts, gpu_device):
    """tf_smpl.batch_lbs.batch_global_rigid_transformation (reference :133-194) on the device vs the oracle
    (itself pinned to the reference's own code): absolute joints and the relative 4x4 transforms."""
    import torch
    from human_dynamics_amd.tf_smpl.batch_lbs import batch_global_rigid_transformation
    from oracle import hmmr_oracle as O
    rng = np.random.default_rng(4)
    theta = (rng.normal(size=(7, 72)) * 0.5).astype(np.float32)
    Rs = O.batch_rodrigues(torch.tensor(theta, dtype=torch.float64).reshape(-1, 3)).reshape(7, 24, 3, 3)
    Js = torch.tensor(rng.normal(size=(7, 24, 3)) * 0.3, dtype=torch.float64)
    parents = [int(p) for p in smpl_consts["parents"]]
    ref_j, ref_A = O.batch_global_rigid_transformation(Rs, Js, parents)
    new_j, A = batch_global_rigid_transformation(Rs.float().numpy(), Js.float().numpy(), smpl_consts["parents"])
    assert new_j.shape == (7, 24, 3) and A.shape == (7, 24, 4, 4)
    assert np.abs(new_j.cpu().numpy() - ref_j.numpy()).max() < 2e-6
    assert np.abs(A.cpu().numpy() - ref_A.numpy()).max() < 2e-6


def test_one_smpl_launch_set_for_all_containers_equals_per_container_calls(weights, smpl_consts, gpu_device):
    """hmmr_smpl_fwd_records (all containers in three launches, cams / shapes / omegas written by the keypoint kernel)
    against one hmmr_smpl_fwd_strided per container + the three field copies it replaces: the same bytes in every
    record, including a ragged instance count (the verts kernel works on groups of 16 instances)."""
    import torch
    from human_dynamics_amd.evaluation.tester import Tester, OUTPUT_KEYS
    t = Tester(Config(batch_size=8), weights=weights, smpl=smpl_consts, dtype="f32", device=gpu_device)
    eng = t.engine
    layout, rec_len = t.record_layout()
    off = {k: (o, sz) for k, shp, o, sz in layout}
    for n in (1, 21, 64):
        strips = torch.randn((n, 2048), generator=torch.Generator(device=gpu_device).manual_seed(n), device=gpu_device)
        om = eng.ief(strips)
        got = t.records_from_omegas(om, torch.full((n, rec_len), float("nan"), device=gpu_device))
        ref = torch.full((n, rec_len), float("nan"), device=gpu_device)
        cams0 = om[0][:, :3]
        for r, key in enumerate(eng.reg_keys):
            base = ({k: off[k][0] for k in OUTPUT_KEYS} if key == 0 else
                    {k: off[k + "_delta"][0] + (r - 1) * (off[k + "_delta"][1] // 2) for k in OUTPUT_KEYS})
            eng.smpl_into(om[r][:, 3:75], om[r][:, 75:85], cams0, ref, base["verts"], base["joints"], base["kps"], base["poses"])
            ref[:, base["cams"]:base["cams"] + 3] = cams0
            ref[:, base["shapes"]:base["shapes"] + 10] = om[r][:, 75:85]
            ref[:, base["omegas"]:base["omegas"] + 85] = om[r]
        assert not torch.isnan(ref).any() and not torch.isnan(got).any()          # every float of the record is written
        assert torch.equal(got, ref), n


def test_smpl_blend_on_matrix_cores_agrees_with_the_vector_form(smpl_consts, gpu_device):
    """The dense blend-shape product [m,218] x [218,3 x 6890] in its three forms (hmmr_debug_t.smpl_blend_mfma): 2 = fmaf chains
    on the vector units (smpl_verts_kernel), 1 = exact-fp32 MFMAs (v_mfma_f32_32x32x2_f32: the same exact products, agrees
    with 2 to one ulp, 6e-8 at the vertices' magnitude), 0 = the default since round 4, split-fp16 operands on the matrix cores
    (three v_mfma_f32_32x32x16_f16 per product, 22 operand bits: within 1e-6 of the fp32 forms, two orders inside the path's
    1e-4) -- ragged instance counts included (32-instance MFMA blocks, the last vertex tile reaches past vertex 6889)."""
    import torch
    from human_dynamics_amd.engine import HmmrEngine, set_debug
    eng = HmmrEngine(None, smpl_consts, device=gpu_device)
    assert eng.sc.dirs_split
    eng.run_flags(clear=True)
    rng = np.random.default_rng(8)
    for m in (1, 33, 70):
        theta = (rng.normal(size=(m, 72)) * 0.6).astype(np.float32)
        beta = rng.normal(size=(m, 10)).astype(np.float32)
        cams = rng.normal(size=(m, 3)).astype(np.float32)
        out = {}
        try:
            for form in (0, 1, 2):
                set_debug(smpl_blend_mfma=form)
                out[form] = [t.clone() for t in eng.smpl(theta, beta, cams)]
        finally:
            set_debug()
        for a, b, name in zip(out[1], out[2], ("verts", "joints", "kps", "Rs")):
            assert float((a - b).abs().max()) < 5e-7, (name, m, float((a - b).abs().max()))
        for a, b, name in zip(out[0], out[2], ("verts", "joints", "kps", "Rs")):
            assert float((a - b).abs().max()) < 2e-6, (name, m, float((a - b).abs().max()))
        assert torch.equal(out[1][3], out[2][3]) and torch.equal(out[0][3], out[2][3])      # the rotations do not depend on the blend kernel
    assert eng.run_flags(clear=True) == 0
    # a shape coefficient beyond the fp16 range of the scaled features (|beta| x 2^8 > 65504) raises the saturation flag
    eng.smpl(np.zeros((1, 72), np.float32), np.full((1, 10), 300.0, np.float32), np.ones((1, 3), np.float32))
    assert eng.run_flags(clear=True) & 1
