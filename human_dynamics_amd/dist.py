"""Multi-GPU sharding of ``predict_all_images`` (one process per GPU, RCCL).

The reference is single-GPU.  Its sliding windows (tester.py:281-295) are
independent of each other, so a video shards by contiguous WINDOW ranges with
no data-path communication: rank r recomputes the <= 2*margin halo frames its
edge windows need, and a single all-gather re-assembles the per-frame outputs.
Sharding never changes any reduction order, so the N-GPU result is
bit-identical to the 1-GPU result.

Everything here is device-agnostic (works with the gloo backend on CPU
tensors), which is how the N>1 plumbing is tested without GPUs.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from . import devflags

# per-frame output record of one container, in make_fetch_dict order (tester.py:217-227)
FIELDS = (("cams", (3,)), ("joints", (25, 3)), ("kps", (25, 2)), ("poses", (24, 3, 3)),
          ("shapes", (10,)), ("verts", (6890, 3)), ("omegas", (85,)))


def record_layout(num_deltas=2, fields=FIELDS):
    """[(key, shape, offset, size)] of the packed per-frame record and its length."""
    out, off = [], 0
    for suffix, rep in (("", 0), ("_delta", num_deltas)):
        if suffix and rep == 0:
            continue
        for k, shp in fields:
            full = ((rep,) if suffix else ()) + tuple(shp)
            size = int(np.prod(full))
            out.append((k + suffix, full, off, size))
            off += size
    return out, off


class ShardPlan(object):
    """Window range, frame range and output range of one rank."""

    def __init__(self, n_frames, batch_size, sequence_length, fov, world_size, rank):
        self.n_frames = n_frames
        self.margin = (fov - 1) // 2
        self.g = sequence_length - 2 * self.margin
        self.T = sequence_length
        count = int(np.ceil(n_frames / float(self.g * batch_size)))
        self.n_windows = count * batch_size
        self.world_size, self.rank = world_size, rank
        # equal contiguous window ranges (the last ranks may own empty / padding windows)
        self.win_per_rank = int(np.ceil(self.n_windows / float(world_size)))
        self.w0 = min(rank * self.win_per_rank, self.n_windows)
        self.w1 = min(self.w0 + self.win_per_rank, self.n_windows)
        # real frames this rank must encode (window span minus the zero-image padding)
        lo = self.w0 * self.g - self.margin
        hi = (self.w1 - 1) * self.g + self.T - self.margin if self.w1 > self.w0 else lo
        self.f0, self.f1 = max(0, min(lo, n_frames)), max(0, min(hi, n_frames))
        # output frames this rank produces
        self.o0 = min(self.w0 * self.g, n_frames)
        self.o1 = min(self.w1 * self.g, n_frames)
        self.out_per_rank = self.win_per_rank * self.g       # padded, equal on every rank

    def window_frame_index(self):
        """[W, T] int64: index of every window slot into the rank's local frame
        array, -1 for zero-image padding slots (front margin / back fill)."""
        w = np.arange(self.w0, self.w1)[:, None] * self.g + np.arange(self.T)[None, :] - self.margin
        idx = np.where((w >= 0) & (w < self.n_frames), w - self.f0, -1)
        return idx.astype(np.int64)


def pack_outputs(out, n_rows, layout, rec_len, device=None, dtype=torch.float32):
    """dict of [n, ...] tensors -> [n_rows, rec_len] (rows >= n are zero)."""
    n = next(iter(out.values())).shape[0] if out else 0
    dev = device if device is not None else (next(iter(out.values())).device if out else "cpu")
    buf = torch.zeros((n_rows, rec_len), dtype=dtype, device=dev)
    for k, shp, off, size in layout:
        if n:
            buf[:n, off:off + size] = out[k].reshape(n, size)
    return buf


def unpack_outputs(buf, layout):
    n = buf.shape[0]
    return {k: buf[:, off:off + size].reshape((n,) + shp) for k, shp, off, size in layout}


def all_gather_outputs(local_buf, plan, group=None):
    """ONE all-gather (RCCL over xGMI on GPUs) of the packed per-frame records;
    returns the [n_frames, rec_len] result on every rank."""
    world = plan.world_size
    if world == 1:
        return local_buf[:plan.n_frames]
    full = torch.empty((world * plan.out_per_rank, local_buf.shape[1]), dtype=local_buf.dtype,
                       device=local_buf.device)
    dist.all_gather_into_tensor(full, local_buf.contiguous(), group=group)
    return full[:plan.n_frames]


class ShardedPredictor(object):
    """One rank's share of ``Tester.predict_all_images`` for a video of `n_frames` frames,
    re-usable across calls (the shard plan and the window index live on the device).

    overlap_gather=True double-buffers the record tensors and issues the all-gather of call k
    asynchronously (RCCL runs it on its own stream), so it overlaps the compute of call k+1; call
    `ready(t)` (or `finish()`) before reading a returned tensor, and read it before the second-next
    `run()`, which re-uses its buffer.
    With 8 ranks the gather moves 0.5 GB per call -- about a quarter of a step if left serial.

    pipeline=True runs the per-window tail (f_movie, IEF, 3x SMPL: ~150 small, latency-bound
    launches that leave most CUs idle) on a second HIP stream, so the tail of call k executes
    underneath the ResNet of call k+1, which has the CUs busy but is not short of queue slots.
    Results are bit-identical (same kernels, same order per stream); the hand-off is one event,
    the feature tensor is pinned to the tail stream with `record_stream`.  Same contract as
    overlap_gather: `ready(t)` / `finish()` before reading, buffers are re-used two calls later.

    use_graph=True captures the local pass -- every launch from the ResNet to the three SMPL
    evaluations, ~150 kernels -- into ONE hipGraph the first time it runs on a device-resident
    input and replays it afterwards.  (Measured equal to eager launches: the step is GPU-bound.)"""

    def __init__(self, tester, n_frames, rank=None, world_size=None, group=None, use_graph=False,
                 overlap_gather=False, pipeline=False, gather_mode=None, step_streams=True):
        if world_size is None:
            world_size = dist.get_world_size() if dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank() if dist.is_initialized() else 0
        self.tester, self.group = tester, group
        self.plan = ShardPlan(n_frames, tester.batch_size, tester.sequence_length, tester.fov, world_size, rank)
        # gather_mode 'theta' (world_size > 1): the ranks exchange only the regressed omegas (R x 85 floats per
        # frame instead of the 63 K floats of a full record) and every rank evaluates SMPL for the WHOLE video.
        # Same kernels on the same per-frame operands => the same bits as 'records'; it trades 250x less xGMI
        # traffic for world_size x the SMPL work, and pays when the gather is not hidden behind the next step.
        # Default (round 6): ONE call = one video, whose gather nothing hides -- the omegas (4 MB for 4096 frames, + ~2 ms of SMPL for the
        # whole video) instead of the records (1.04 GB, ~7 ms over one xGMI link); a caller that overlaps the gather with the next
        # call (overlap_gather=True: a stream of videos) keeps the records, whose gather then costs nothing on the critical path.
        if gather_mode is None:
            gather_mode = "records" if overlap_gather else "theta"
        assert gather_mode in ("records", "theta")
        self.theta = gather_mode == "theta" and world_size > 1
        if self.theta:
            overlap_gather = False              # the gathered omegas are consumed at once (SMPL), nothing to overlap
        self.overlap = bool(overlap_gather) and world_size > 1
        self.pipeline = bool(pipeline) and tester.engine.device.type == "cuda"
        self.use_graph = bool(use_graph) and not self.overlap and not self.pipeline
        self.layout, self.rec_len = tester.record_layout()
        eng = tester.engine
        p = self.plan
        n_local = p.f1 - p.f0
        if p.w1 > p.w0 and p.o1 > p.o0:
            idx = torch.from_numpy(p.window_frame_index())
            idx = torch.where(idx < 0, torch.full_like(idx, n_local), idx)     # row n_local = zero image
            self.idx = idx.to(eng.device)
        else:
            self.idx = None
        nbuf = max(2, int(devflags.get("STEP_BUFFERS"))) if (self.overlap or self.pipeline) else 1
        self.s_tail = (torch.cuda.Stream(device=eng.device, priority=int(devflags.get("TAIL_PRIORITY")))
                       if self.pipeline else None)
        # pipeline + step_streams: consecutive calls encode on ALTERNATING high-priority streams, each as ONE whole-batch
        # launch sequence with its own workspace, so two ResNet passes (of steps k and k+1) are in flight instead of the two
        # half-batch sequences of one step: twice the tiles per launch against the same gap filling (measured 8.28 -> 7.87 ms
        # per 256-frame step).  Same kernels on the same per-frame operands => the same bits.  The caller must not overwrite
        # a `frames` tensor before `ready()` of that call's result (the encode stream reads it after run() has returned).
        self.step_streams = ([torch.cuda.Stream(device=eng.device, priority=-1) for _ in range(nbuf)]
                             if (self.pipeline and step_streams and devflags.get("STEP_STREAMS") != "0") else None)
        self.done = [None] * nbuf
        self.n_reg = 1 + len(tester.delta_t_values)
        self.locals = [torch.zeros((p.out_per_rank, self.n_reg * 85 if self.theta else self.rec_len),
                                   dtype=torch.float32, device=eng.device) for _ in range(nbuf)]
        self.fulls = [None] * nbuf
        self.pending = [None] * nbuf
        self.calls = 0
        self.graph, self.static_frames, self._warm = None, None, 0

    @property
    def local(self):
        return self.locals[0]

    def _local_pass(self, frames, out):
        """frames [f1-f0,224,224,3] on the device -> out (packed records of this rank)."""
        phi_all = self.tester.features(frames, n_zero=1)            # last row = feature of the zero image
        if self.idx is not None:
            self._tail(phi_all, out)
        return out

    def _tail(self, phi_all, out):
        """Everything after the ResNet for this rank's windows -> `out` (records, or omegas in theta mode)."""
        n = self.plan.o1 - self.plan.o0
        if not self.theta:
            self.tester.predict_strips_records(phi_all[self.idx], n, out=out)
        else:
            om = self.tester.predict_strips_omegas(phi_all[self.idx], n)            # [R, n, 85]
            out[:n] = om.permute(1, 0, 2).reshape(n, -1)

    def _gather_theta(self, local, slot):
        """theta mode: all-gather the omegas, then SMPL + record assembly for every frame of the video."""
        p = self.plan
        full_om = torch.empty((p.world_size * p.out_per_rank, local.shape[1]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(full_om, local.contiguous(), group=self.group)
        om = full_om[:p.n_frames].reshape(p.n_frames, self.n_reg, 85).permute(1, 0, 2).contiguous()
        if self.fulls[slot] is None:
            self.fulls[slot] = torch.empty((p.n_frames, self.rec_len), dtype=torch.float32, device=local.device)
        return self.tester.records_from_omegas(om, out=self.fulls[slot])[:p.n_frames]

    def run_local(self, frames, out=None):
        eng = self.tester.engine
        out = self.locals[0] if out is None else out
        if not (isinstance(frames, torch.Tensor) and frames.device.type == eng.device.type):
            frames = eng.to_device(frames)
        if not self.use_graph:
            return self._local_pass(frames, out)
        if self.graph is not None:
            if frames.data_ptr() != self.static_frames.data_ptr():
                self.static_frames.copy_(frames)
            self.graph.replay()
            return self.locals[0]
        if self._warm < 2:                                     # eager warm-up: sizes the workspaces
            self._warm += 1
            return self._local_pass(frames, out)
        self.static_frames = frames
        torch.cuda.synchronize(eng.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._local_pass(self.static_frames, self.locals[0])
        self.graph = g
        g.replay()
        return self.locals[0]

    def _run_pipelined(self, frames, slot, gather):
        """ResNet on the caller's stream (or on this step's encode stream), everything after it on `s_tail`."""
        eng, p = self.tester.engine, self.plan
        enc = torch.cuda.current_stream(eng.device)
        if not (isinstance(frames, torch.Tensor) and frames.device.type == eng.device.type):
            frames = eng.to_device(frames)
        n_local = frames.shape[0]
        if self.step_streams is not None and n_local <= self.tester.MAX_DEVICE_FRAMES:
            se = self.step_streams[slot]
            se.wait_stream(enc)                               # the frames are ready on the caller's stream
            with torch.cuda.stream(se):
                phi_all = eng.resnet(frames, n_zero=1, ws_key="resnet_step%d" % slot, parts=1)
                handoff = torch.cuda.Event()
                handoff.record(se)
            frames.record_stream(se)
        else:
            phi_all = self.tester.features(frames, n_zero=1)
            handoff = torch.cuda.Event()
            handoff.record(enc)
        out = self.locals[slot]
        with torch.cuda.stream(self.s_tail):
            self.s_tail.wait_event(handoff)
            phi_all.record_stream(self.s_tail)
            if self.pending[slot] is not None:                 # the gather that last read this buffer
                self.pending[slot].wait()
                self.pending[slot] = None
            if self.idx is not None:
                self._tail(phi_all, out)
            if gather and self.theta:
                out = self._gather_theta(out, slot)
            elif gather and p.world_size > 1:
                if self.fulls[slot] is None:
                    self.fulls[slot] = torch.empty((p.world_size * p.out_per_rank, self.rec_len), dtype=out.dtype,
                                                   device=out.device)
                self.pending[slot] = dist.all_gather_into_tensor(self.fulls[slot], out, group=self.group,
                                                                 async_op=True)
                out = self.fulls[slot][:p.n_frames]
            elif gather:
                out = out[:p.n_frames]
            self.done[slot] = torch.cuda.Event()
            self.done[slot].record(self.s_tail)
        return out

    def run(self, frames, gather=True):
        p = self.plan
        slot = self.calls % len(self.locals)
        self.calls += 1
        if self.pipeline:
            return self._run_pipelined(frames, slot, gather)
        if self.pending[slot] is not None:                     # the gather that last read this buffer
            self.pending[slot].wait()
            self.pending[slot] = None
        local = self.run_local(frames, self.locals[slot])
        if not gather:
            return local
        if self.theta:
            return self._gather_theta(local, slot)
        if not self.overlap:
            return all_gather_outputs(local, p, self.group)
        if self.fulls[slot] is None:
            self.fulls[slot] = torch.empty((p.world_size * p.out_per_rank, self.rec_len), dtype=local.dtype,
                                           device=local.device)
        self.pending[slot] = dist.all_gather_into_tensor(self.fulls[slot], local, group=self.group, async_op=True)
        return self.fulls[slot][:p.n_frames]

    def ready(self, result):
        """Block until the asynchronous gather that fills `result` (a tensor returned by `run`) is
        complete.  The tensor stays valid until the second-next `run()`, which re-uses its buffer."""
        for i, full in enumerate(self.fulls):
            mine = (full is not None and result.data_ptr() == full.data_ptr()) or \
                   (self.pipeline and result.data_ptr() == self.locals[i].data_ptr())
            if not mine:
                continue
            if self.pipeline and self.done[i] is not None:
                torch.cuda.current_stream(result.device).wait_event(self.done[i])
            if self.pending[i] is not None:
                self.pending[i].wait()
                self.pending[i] = None
        return result

    def finish(self):
        """Complete every outstanding asynchronous gather."""
        if self.pipeline:
            torch.cuda.current_stream(self.tester.engine.device).wait_stream(self.s_tail)
        for i, w in enumerate(self.pending):
            if w is not None:
                w.wait()
                self.pending[i] = None


def predict_all_images_sharded(tester, frames_fn, n_frames, rank=None, world_size=None, group=None,
                               gather=True):
    """Distributed ``Tester.predict_all_images``.

    frames_fn(f0, f1) -> the real frames [f1-f0,224,224,3] of the video (host or
    device); every rank only ever asks for its own span.  Returns the packed
    [n_frames, rec_len] device tensor (identical on every rank), the layout and the plan."""
    sp = ShardedPredictor(tester, n_frames, rank, world_size, group)
    p = sp.plan
    if p.f1 > p.f0:
        frames = frames_fn(p.f0, p.f1)
    else:
        frames = torch.empty((0, tester.img_size, tester.img_size, 3), dtype=torch.float32,
                             device=tester.engine.device)
    return sp.run(frames, gather), sp.layout, p
