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
