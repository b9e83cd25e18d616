"""N>1 plumbing on CPU: shard plan, packing and the single all-gather, run with
the gloo backend and world_size 2 (the GPU path uses the same code with RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from human_dynamics_amd import dist as hd


def _fake_outputs(frame_ids, layout):
    """Deterministic per-frame records: value = f(frame id, key, element)."""
    out = {}
    for ki, (k, shp, off, size) in enumerate(layout):
        base = torch.arange(size, dtype=torch.float32).reshape((1,) + shp)
        out[k] = base * 1e-3 + frame_ids.reshape((-1,) + (1,) * len(shp)).float() + 100.0 * ki
    return out


def test_shard_plans_tile_the_video_and_match_the_literal_windows():
    B, T, fov = 8, 20, 13
    for n in (1, 24, 64, 100, 256, 1000):
        margin, g = 6, 8
        count = int(np.ceil(n / (g * B)))
        num_fill = count * B * g + T - n
        padded = np.concatenate([-np.ones(margin), np.arange(n), -np.ones(num_fill)]).astype(np.int64)
        literal = np.stack([padded[i * g:i * g + T] for i in range(count * B)])     # tester.py:293-295
        for world in (1, 2, 3, 8):
            outs, wins = [], []
            for r in range(world):
                p = hd.ShardPlan(n, B, T, fov, world, r)
                idx = p.window_frame_index()
                glob = np.where(idx >= 0, idx + p.f0, -1)
                wins.append(glob)
                assert idx.max(initial=-1) < max(p.f1 - p.f0, 1)
                assert p.f1 - p.f0 <= (p.w1 - p.w0) * g + 2 * margin
                outs.extend(range(p.o0, p.o1))
                assert p.o1 - p.o0 <= p.out_per_rank
            assert outs == list(range(n))
            assert np.array_equal(np.concatenate(wins, 0), literal)


def test_pack_unpack_roundtrip():
    layout, rec_len = hd.record_layout(2)
    assert rec_len == 3 * (3 + 75 + 50 + 216 + 10 + 20670 + 85)          # 253 KB / frame with both deltas
    out = _fake_outputs(torch.arange(5), layout)
    buf = hd.pack_outputs(out, 8, layout, rec_len)
    assert buf.shape == (8, rec_len) and not buf[5:].any()
    back = hd.unpack_outputs(buf[:5], layout)
    for k in out:
        assert torch.equal(back[k], out[k])
    assert back["verts_delta"].shape == (5, 2, 6890, 3)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fields = (("cams", (3,)), ("joints", (25, 3)), ("omegas", (85,)))      # small record for the CPU test
        layout, rec_len = hd.record_layout(2, fields)
        plan = hd.ShardPlan(n_frames, 8, 20, 13, world, rank)
        ids = torch.arange(plan.o0, plan.o1)
        local = hd.pack_outputs(_fake_outputs(ids, layout), plan.out_per_rank, layout, rec_len)
        full = hd.all_gather_outputs(local, plan)
        expect = hd.pack_outputs(_fake_outputs(torch.arange(n_frames), layout), n_frames, layout, rec_len)
        results[rank] = bool(torch.equal(full, expect)) and full.shape[0] == n_frames
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [100, 256])
def test_all_gather_reassembles_the_sequence_world2(n_frames):
    world = 2
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n_frames, results), nprocs=world, join=True)
    assert dict(results) == {0: True, 1: True}


class _FakeEngine(object):
    device = torch.device("cpu")


class _FakeTester(object):
    """Stands in for Tester in the plumbing test: per-frame records are a function of the frame id."""
    batch_size, sequence_length, fov, img_size, delta_t_values = 8, 20, 13, 4, [-5, 5]
    engine = _FakeEngine()

    def record_layout(self):
        return hd.record_layout(2, (("cams", (3,)), ("omegas", (85,))))

    def features(self, frames, n_zero=0):
        return torch.cat([frames[:, 0, 0, :1].repeat(1, 4), torch.zeros((n_zero, 4))], 0)   # id in every column

    def predict_strips_records(self, windows, n_keep, out=None):
        kept = windows[:, 6:14, 0].reshape(-1)[:n_keep]              # the centre 8 slots of each window
        out[:n_keep] = kept[:, None] + torch.arange(out.shape[1])[None, :] * 1e-3
        return out

    def predict_strips_omegas(self, windows, n_keep):
        kept = windows[:, 6:14, 0].reshape(-1)[:n_keep]
        return torch.stack([kept[:, None].repeat(1, 85) + 0.25 * r for r in range(3)])      # [R, n, 85]

    def records_from_omegas(self, om, out=None):
        n = om.shape[1]
        assert om.shape[0] == 3 and torch.equal(om[1], om[0] + 0.25) and torch.equal(om[2], om[0] + 0.5)
        if out is None:
            out = torch.empty((n, self.record_layout()[1]))
        out[:n] = om[0][:, :1] + torch.arange(out.shape[1])[None, :] * 1e-3
        return out


def _expected(n, k, width):
    return ((torch.arange(n) + 1000.0 * k)[:, None] + torch.arange(width)[None, :] * 1e-3).float()


def _overlap_worker(rank, world, port, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 200
        sp = hd.ShardedPredictor(_FakeTester(), n, rank, world, overlap_gather=True)
        ok = sp.overlap and len(sp.locals) == 2
        outs = []
        for step in range(4):                                        # pipelined: gather k overlaps compute k+1
            frames = torch.zeros((sp.plan.f1 - sp.plan.f0, 4, 4, 3))
            frames[:, 0, 0, 0] = torch.arange(sp.plan.f0, sp.plan.f1) + 1000.0 * step
            outs.append(sp.run(frames))                              # issues gather(step) asynchronously
            if step >= 1:                                            # gather(step-1) overlapped this compute
                t = sp.ready(outs[step - 1])
                ok = ok and bool(torch.allclose(t, _expected(n, step - 1, t.shape[1])))
        sp.finish()
        ok = ok and bool(torch.allclose(outs[3], _expected(n, 3, outs[3].shape[1])))
        results[rank] = ok and all(w is None for w in sp.pending)
    finally:
        dist.destroy_process_group()


def test_overlapped_gather_pipeline_world2():
    """ShardedPredictor(overlap_gather=True): double-buffered records, async all-gather per call."""
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_overlap_worker, args=(2, _free_port(), results), nprocs=2, join=True)
    assert dict(results) == {0: True, 1: True}


def _theta_worker(rank, world, port, n, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = True
        for mode in ("records", "theta"):
            sp = hd.ShardedPredictor(_FakeTester(), n, rank, world, gather_mode=mode)
            frames = torch.zeros((sp.plan.f1 - sp.plan.f0, 4, 4, 3))
            frames[:, 0, 0, 0] = torch.arange(sp.plan.f0, sp.plan.f1).float()
            full = sp.run(frames)
            ok = ok and full.shape[0] == n and bool(torch.allclose(full, _expected(n, 0, full.shape[1])))
            ok = ok and (sp.theta == (mode == "theta")) and sp.locals[0].shape[1] == (255 if mode == "theta" else full.shape[1])
        results[rank] = ok
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [200, 4096])
def test_theta_gather_equals_record_gather_world2(n):
    """gather_mode='theta': only the omegas cross the wire, every rank rebuilds all records; and the
    BASELINE configs[4] video (4096 frames, 512 windows) shards over two ranks."""
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_theta_worker, args=(2, _free_port(), n, results), nprocs=2, join=True)
    assert dict(results) == {0: True, 1: True}


def test_config5_plan_4096_frames():
    """BASELINE configs[4]: 4096 frames, B=8, T=20 -> 512 windows; 1/2/4/8 ranks own 4096/N output frames and
    encode at most 12 halo frames more."""
    for world in (1, 2, 4, 8):
        tot = 0
        for r in range(world):
            p = hd.ShardPlan(4096, 8, 20, 13, world, r)
            assert p.n_windows == 512 and p.o1 - p.o0 == 4096 // world
            assert (p.f1 - p.f0) - (p.o1 - p.o0) <= 12
            tot += p.o1 - p.o0
        assert tot == 4096


def test_bench_self_launch_and_default_workload(tmp_path):
    """`python bench.py --gpus N` without a launcher re-runs itself under torch.distributed.run (one rank per GPU; bench.self_launch):
    here with a stand-in script on 2 gloo ranks.  And the N > 1 default is BASELINE configs[4] (one 4096-frame video, strong
    scaling), N = 1 a 256-frame shard (bench.resolve_workload)."""
    import argparse
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    stub = tmp_path / "stub.py"
    stub.write_text(
        "import os, sys, torch, torch.distributed as dist\n"
        "dist.init_process_group('gloo')\n"
        "t = torch.tensor([float(dist.get_rank() + 1)])\n"
        "dist.all_reduce(t)\n"
        "open(os.path.join(sys.argv[1], 'rank%d' % dist.get_rank()), 'w').write('%d %d %s' % (dist.get_world_size(), int(t.item()), ' '.join(sys.argv[2:])))\n"
        "dist.destroy_process_group()\n")
    rc = bench.self_launch(2, [str(tmp_path), "--gpus", "2", "--steps", "3"], script=str(stub))
    assert rc == 0
    for r in (0, 1):
        assert (tmp_path / ("rank%d" % r)).read_text() == "2 3 --gpus 2 --steps 3"
    ns = lambda **kw: argparse.Namespace(**dict(dict(frames=256, video_frames=0, weak=False), **kw))
    assert bench.resolve_workload(ns(), 1) == (False, 256)
    assert bench.resolve_workload(ns(), 8) == (True, 4096)
    assert bench.resolve_workload(ns(weak=True), 8) == (False, 2048)
    assert bench.resolve_workload(ns(video_frames=1024), 2) == (True, 1024)


def _mg_worker(rank, world, port, n, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        t = _FakeTester()
        plan = hd.ShardPlan(n, 8, 20, 13, world, rank)
        frames = torch.zeros((plan.f1 - plan.f0, 4, 4, 3))
        frames[:, 0, 0, 0] = torch.arange(plan.f0, plan.f1).float()
        results[rank] = bench.multi_gpu_fields(t, n, frames, world, rank, torch.device("cpu"), reps=2)
    finally:
        dist.destroy_process_group()


def test_bench_multi_gpu_fields_world2():
    """Round 6: what the N > 1 bench line says about the path's one collective -- `rccl_ranks` read after an actual all-gather, both
    gather modes' isolated `all_gather_ms` / bytes and `single_video_ms` (one un-overlapped step of the whole video), and the mode the
    measurement prefers (bench.py --gather auto takes it for a strong-scaling run) -- produced by bench.multi_gpu_fields on two gloo
    ranks with the stand-in Tester for a 512-frame video (`bench.py --gpus 2 --video-frames 512`; the real thing needs two GPUs)."""
    n = 512
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_mg_worker, args=(2, _free_port(), n, results), nprocs=2, join=True)
    r0, r1 = results[0], results[1]
    rec_len = _FakeTester().record_layout()[1]
    for r in (r0, r1):
        assert r["rccl_ranks"] == 2
        assert set(r["all_gather_ms"]) == set(r["single_video_ms"]) == set(r["all_gather_bytes"]) == {"records", "theta"}
        assert r["all_gather_bytes"] == {"records": n * rec_len * 4, "theta": n * 255 * 4}
        assert all(v > 0 for v in r["all_gather_ms"].values()) and all(v > 0 for v in r["single_video_ms"].values())
        assert r["gather_by_measurement"] == min(r["single_video_ms"], key=r["single_video_ms"].get)
    # max-over-ranks figures: every rank holds the same numbers, so every rank picks the same mode
    assert r0["single_video_ms"] == r1["single_video_ms"] and r0["all_gather_ms"] == r1["all_gather_ms"]


def _choose_worker(rank, world, port, results):
    """precision.choose_engine's collective with DIFFERENT per-process caches: rank 0 already holds a decision for the weight set (a
    Tester built before init_process_group), rank 1 does not -- every rank must still take part in the one broadcast, and all end on
    rank 0's rung.  The engine and the probe are stand-ins (no device here): what is under test is who calls the collective."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from human_dynamics_amd import precision as P
    built = []
    P.HmmrEngine = lambda w, s, dtype, temporal_dtype, ief_dtype, device, **kw: built.append((dtype, temporal_dtype, ief_dtype)) or built[-1]
    probes = []

    def fake_probe(weights, smpl, device, pred_mode, make):
        probes.append(rank)
        idx = 1 if rank == 0 else 0                      # a borderline weight set: the ranks would decide differently on their own
        return make(P.LADDER[idx]), {"operands": "x", "rungs": []}, idx
    P._probe = fake_probe
    w = {"a": np.arange(6, dtype=np.float32).reshape(2, 3)}
    smpl_a, smpl_b = {"v_template": np.ones((4, 3), np.float32)}, {"v_template": np.full((4, 3), 2.0, np.float32)}
    if rank == 0:
        P.choose_engine(w, smpl_a, "cpu")                # before the process group exists: rank 0's cache is filled, rank 1's is not
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        e1, r1 = P.choose_engine(w, smpl_a, "cpu")       # rank 0: cached; rank 1: nothing cached -- one broadcast either way
        e2, r2 = P.choose_engine(w, smpl_b, "cpu")       # another body model is another decision: rank 0 probes, rank 1 adopts
        e3, r3 = P.choose_engine(w, smpl_a, "cpu", unit_pair=False)      # ... and so is another engine configuration
        results[rank] = dict(rungs=[e1, e2, e3], probes=list(probes), cached=[bool(r.get("cached")) for r in (r1, r2, r3)])
    finally:
        dist.destroy_process_group()


def test_choose_engine_collective_with_different_caches():
    from human_dynamics_amd import precision as P
    assert (P.weights_fingerprint({"a": np.ones(3)}, "pred", {"v": np.ones(2)}) !=
            P.weights_fingerprint({"a": np.ones(3)}, "pred", {"v": np.zeros(2)}))
    assert P.weights_fingerprint({"a": np.ones(3)}, "pred", None, {"unit_pair": False}) != P.weights_fingerprint({"a": np.ones(3)}, "pred")
    world, port = 2, _free_port()
    with mp.Manager() as man:
        results = man.dict()
        mp.spawn(_choose_worker, args=(world, port, results), nprocs=world, join=True)
        r0, r1 = results[0], results[1]
    f16, mixed = ("f16x3",) * 3, ("f32", "f16x3", "f16x3")
    assert r0["rungs"] == r1["rungs"] == [mixed, mixed, mixed], (r0, r1)        # everybody runs rank 0's rung (index 1)
    assert r0["probes"] == [0, 0, 0] and r1["probes"] == []                     # only rank 0 ever probes
    assert r1["cached"] == [True, True, True] and r0["cached"] == [True, False, False]
    assert f16 != mixed
