"""Host-in / host-out `predict_all_images` as a three-stream pipeline.

The reference contract is host ndarray in, dict of host ndarrays out (src/evaluation/tester.py:
239-241, 257; demo_video.py:172 hands over the whole cropped video).  Through that surface the
path is bound by PCIe and host copies unless they overlap the kernels, so the video is cut into
chunks of `chunk` frames and three HIP streams run side by side:

    copy-in stream   chunk k+1: the caller's array -> HBM (double-buffered device side; straight from the pageable
                                array, which moves at the pinned rate on this platform, or via pinned staging buffers);
                                uint8 input is converted to float by one kernel on the ResNet's first stream
    compute streams  chunk k  : ResNet -> phi, as the engine's two contiguous parts on its two streams; each part
                                starts when ITS half of the upload has landed
    tail stream      chunk k-1: the per-window tail (f_movie, IEF, 3 x SMPL), whose halo is encoded by now,
                                underneath the ResNet of chunk k+1
    copy-out stream  chunk k-2: record fields -> one pinned host array per output key

Copies are served in the order they were submitted, whatever stream they are on: a download that waits for its
tail blocks every upload queued behind it, and the ResNet of the next chunk with it.  So the upload of chunk k+1
is queued BEFORE the tail of chunk k-1, and a chunk's downloads are queued by a downloader thread at the moment
its records exist (tools/stream_trace.py shows both timelines).

Only the copy-in call occupies the Python thread (2.8 ms per 256 float32 frames), while the GPU works on the
previous chunk.  Every kernel sees exactly the operands it sees in the
one-shot path (per-frame ResNet, per-window tail), so the result is byte-identical to
`Tester.predict_all_images(..., stream=False)`; tested on the GPU.

uint8 input ([N,224,224,3], already cropped): uploaded as bytes (4x less H2D) and converted on
the device by `hmmr_crop_frames` with the identity geometry, i.e. ((x / 255) - 0.5) * 2 evaluated in
float64 like src/evaluation/run_video.py:73.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib as L

OUTPUT_KEYS = ("cams", "joints", "kps", "poses", "shapes", "verts", "omegas")


class HostStreamer(object):
    """Re-usable pinned staging buffers and copy streams of one Tester."""

    def __init__(self, tester, chunk=256, staged="auto"):
        self.t = tester
        # How a chunk of the caller's (pageable) frames reaches HBM.  False: copied straight from the array -- on MI355X /
        # ROCm 7 at the pinned rate (55 GB/s, tools/pcie_probe.py), but the copy does not overlap running kernels; True:
        # through pinned staging buffers (a host memcpy on the private pool, then an async DMA under the kernels).
        # "auto": chunk 0 straight (nothing runs yet), later chunks staged -- best for one-chunk and for long videos.
        self.staged = staged
        self.eng = tester.engine
        self.dev = self.eng.device
        self.margin = (tester.fov - 1) // 2
        self.g = tester.sequence_length - 2 * self.margin
        self.chunk = max(self.g, (int(chunk) // self.g) * self.g)          # whole windows per chunk
        self.s_in = torch.cuda.Stream(device=self.dev)
        self.s_out = torch.cuda.Stream(device=self.dev)
        self.s_tail = torch.cuda.Stream(device=self.dev)       # the ~150 small launches of a chunk's tail run under the next ResNet
        self._pin, self._dev_in, self._geom, self._dev_f32 = {}, {}, None, None
        self._phi_zero = None
        from .. import devflags
        self._poison = devflags.get("STREAM_POISON") == "1"
        self.layout, self.rec_len = tester.record_layout()
        # staging copies (pageable user array -> pinned buffer) run on a small private pool of plain memcpy workers
        # (NumPy releases the GIL): torch's intra-op pool would wake one spinning thread per core for every chunk,
        # which under a container CPU quota stalls the whole process for tens of milliseconds at a time
        from concurrent.futures import ThreadPoolExecutor
        self._copy_workers = 8
        self._pool = ThreadPoolExecutor(max_workers=self._copy_workers)
        # ... and are started one chunk AHEAD by a stager thread (it waits for the staging pair to be free, then fans the
        # memcpy out), so the Python thread keeps enqueueing kernels instead of waiting for 154 MB of memcpy per chunk
        self._stager = ThreadPoolExecutor(max_workers=1)
        self._downloader = ThreadPoolExecutor(max_workers=1)        # queues a chunk's downloads when its records are ready

    def _stage(self, dst_pinned, src):
        """src (ndarray [n,...]) -> the first n rows of the pinned staging tensor."""
        n = len(src)
        dst = dst_pinned.numpy()[:n]
        step = (n + self._copy_workers - 1) // self._copy_workers
        futs = [self._pool.submit(np.copyto, dst[i:i + step], src[i:i + step]) for i in range(0, n, step)]
        for f in futs:
            f.result()

    def _stage_when_free(self, dst_pinned, src, free_event):
        if free_event is not None:
            free_event.synchronize()                               # chunk k-2 has been encoded: its staging pair is free
        self._stage(dst_pinned, src)

    def _staging(self, dtype):
        if dtype not in self._pin:
            shape = (self.chunk, 224, 224, 3)
            self._pin[dtype] = [torch.empty(shape, dtype=dtype, pin_memory=True) for _ in range(2)] if self.staged else [None, None]   # ("auto" is truthy)
            self._dev_in[dtype] = [torch.empty(shape, dtype=dtype, device=self.dev) for _ in range(2)]
        return self._pin[dtype], self._dev_in[dtype]

    def _float_buf(self, slot):
        if self._dev_f32 is None:
            self._dev_f32 = [torch.empty((self.chunk, 224, 224, 3), dtype=torch.float32, device=self.dev) for _ in range(2)]
        return self._dev_f32[slot]

    def _to_float(self, u8, n, out):
        """uint8 crops on the device -> float32 in [-1, 1] (identity geometry of hmmr_crop_frames), on the CURRENT
        stream, into `out` (rows of the slot's float buffer, whose previous reader -- the ResNet of chunk k-2 -- has
        finished: in_free)."""
        if self._geom is None:
            self._geom = torch.tensor([[224, 224, 0, 0]] * self.chunk, dtype=torch.int32, device=self.dev)
        L.check(self.eng.lib.hmmr_crop_frames(u8.data_ptr(), self._geom.data_ptr(), n, 224, 224, out.data_ptr(),
                                              torch.cuda.current_stream(self.dev).cuda_stream), "hmmr_crop_frames")
        return out

    def run(self, all_images, want=None):
        return self.run_many([all_images], want)[0]

    def run_many(self, videos, want=None):
        """Several videos back to back as ONE pipeline (demo_video.py:172 calls predict_all_images once per person track): the chunks of
        all videos form one sequence, so the upload of video v+1's first chunk runs under the ResNet of video v's last one and its tail
        under the next video's ResNet -- what `run` does between the chunks of one long video, across call boundaries.  Every video
        keeps its own padding, windows and output arrays: each result is byte-identical to its own `run` (tested).  Returns a list of
        dicts, one per video."""
        t, eng, dev = self.t, self.eng, self.dev
        keys = [k for k, _, _, _ in self.layout]        # a Tester without delta_t_values has no *_delta fields
        if want is not None:
            unknown = [k for k in want if k not in keys]
            if unknown:
                raise KeyError("unknown output keys %s" % unknown)
            keys = [k for k in keys if k in want]
        fields = {k: (shp, off, size) for k, shp, off, size in self.layout if k in keys}
        C, g, margin, T = self.chunk, self.g, self.margin, t.sequence_length
        results = [None] * len(videos)
        V, G = [], []                   # per-video state; the global chunk sequence [(video, chunk)]
        tdt = None
        for vi, all_images in enumerate(videos):
            N = len(all_images)
            if N == 0:
                results[vi] = {k: np.zeros((0,) + fields[k][0], np.float32) for k in keys}
                V.append(None)
                continue
            src = all_images if isinstance(all_images, np.ndarray) else np.asarray(all_images)
            if src.dtype != np.uint8:
                src = src if src.dtype == np.float32 else src.astype(np.float32)
            assert src.shape[1:] == (224, 224, 3), src.shape
            vdt = torch.uint8 if src.dtype == np.uint8 else torch.float32
            if tdt is not None and vdt != tdt:                 # one pipeline stages one element type: mixed input runs call by call
                return [self.run(v, want) for v in videos]
            tdt = vdt
            if not src.flags.c_contiguous:
                src = np.ascontiguousarray(src)
            n_chunks = (N + C - 1) // C
            V.append(dict(N=N, src=src, n_chunks=n_chunks, first=len(G)))
            G.extend((vi, k) for k in range(n_chunks))
        if not G:
            return results
        pin, dev_in = self._staging(tdt)
        cur = torch.cuda.current_stream(dev)
        for v in V:
            if v is not None:
                v["phi"] = torch.empty((v["N"] + 1, 2048), dtype=torch.float32, device=dev)      # row N: the zero padding image
                v["host"] = {k: torch.empty((v["N"],) + fields[k][0], dtype=torch.float32, pin_memory=True) for k in keys}
        recs = [torch.empty((C, self.rec_len), dtype=torch.float32, device=dev) for _ in range(2)]
        in_free = [None, None]          # compute finished reading dev_in[slot]
        out_free = [None, None]         # copy-out finished reading recs[slot]
        ar_T = torch.arange(T, device=dev)
        import time as _time
        from .. import devflags
        trace = [] if devflags.get("STREAM_TRACE") else None
        t_entry = _time.perf_counter()
        # the zero padding image: its features are a property of the weights (every frame is encoded independently of its batch,
        # tests/test_gpu_f16x3.py::test_resnet_split_batch_independence), so one 1-image pass per streamer serves every call
        if self._phi_zero is None:
            self._phi_zero = torch.empty((1, 2048), dtype=torch.float32, device=dev)
            eng.resnet(torch.empty((0, 224, 224, 3), dtype=torch.float32, device=dev), n_zero=1, out=self._phi_zero)
        for v in V:
            if v is not None:
                v["phi"][v["N"]:v["N"] + 1].copy_(self._phi_zero)
        tr = (lambda tag: trace.append((tag, _time.perf_counter()))) if trace is not None else (lambda tag: None)
        gpu_marks = [] if trace is not None else None          # (tag, event): device-side timeline of the same call

        def mark(tag, stream):
            if gpu_marks is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record(stream)
                gpu_marks.append((tag, e))
        self._mark = mark
        mark("start", cur)
        is_staged = lambda j_: (j_ > 0) if self.staged == "auto" else bool(self.staged)
        n_glob = len(G)

        def span(j_):
            v = V[G[j_][0]]
            lo_ = G[j_][1] * C
            return v, lo_, min(v["N"], lo_ + C)

        def stage_ahead(j_):
            if j_ >= n_glob or not is_staged(j_):
                return None
            v, lo_, hi_ = span(j_)
            return self._stager.submit(self._stage_when_free, pin[j_ % 2], v["src"][lo_:hi_], in_free[j_ % 2])

        enc = [None] * n_glob            # enc[j]: global chunk j is encoded (recorded on the caller's stream)

        def encode(j_, ahead_):
            """Global chunk j_: upload (+ conversion) and ResNet, part by part -- the engine's split of a chunk into contiguous
            parts on concurrent streams, fed here so that part i starts when ITS frames have landed, under the upload of
            part i+1.  Blocks the host until the buffer pair of the slot is free, i.e. until chunk j_-2 is encoded."""
            v, lo_, hi_ = span(j_)
            n_, slot_ = hi_ - lo_, j_ % 2
            staged = is_staged(j_)
            if staged:
                ahead_.result()                                      # pageable -> pinned, done one chunk ahead
                tr(" staged %d" % j_)
            elif in_free[slot_] is not None:
                in_free[slot_].synchronize()                         # the device-side pair is free again (chunk j_-2 is encoded)
            cuts = eng.resnet_cuts(n_)
            one = len(cuts) == 2
            frames = dev_in[slot_][:n_]
            fl = self._float_buf(slot_)[:n_] if tdt == torch.uint8 else frames
            streams = []
            for i_, (a_, b_) in enumerate(zip(cuts[:-1], cuts[1:])):
                with torch.cuda.stream(self.s_in):
                    if staged:
                        frames[a_:b_].copy_(pin[slot_][a_:b_], non_blocking=True)
                    else:
                        # straight from the caller's array; the call returns when the bytes have left it
                        frames[a_:b_].copy_(torch.from_numpy(v["src"][lo_ + a_:lo_ + b_]), non_blocking=True)
                    landed = torch.cuda.Event()
                    landed.record(self.s_in)
                sc = cur if one else eng.side_stream(i_)
                with torch.cuda.stream(sc):
                    if not one:
                        sc.wait_stream(cur)
                    sc.wait_event(landed)
                    if tdt == torch.uint8:                           # a kernel: on the stream of its consumer
                        self._to_float(frames[a_:b_], b_ - a_, fl[a_:b_])
                    mark("upload %d.%d landed" % (j_, i_), sc)
                    eng.resnet(fl[a_:b_], out=v["phi"][lo_ + a_:lo_ + b_], parts=1, ws_key="resnet" if one else "resnet%d" % i_)
                streams.append(sc)
            for sc in streams:
                if sc is not cur:
                    cur.wait_stream(sc)
            tr(" chunk %d queued" % j_)
            mark("resnet %d done" % j_, cur)
            enc[j_] = in_free[slot_] = torch.cuda.Event()
            enc[j_].record(cur)

        # staging runs ONE CHUNK AHEAD of its use: chunk j+2's memcpy is handed to the stager thread as soon as chunk j+1 is
        # queued (it first waits, on its own thread, for the slot's previous user -- chunk j -- to be encoded), so the Python
        # thread finds chunk j+1 already staged when it comes to encode it
        ahead = {0: stage_ahead(0)}
        encode(0, ahead.pop(0))
        ahead[1] = stage_ahead(1)
        for j in range(n_glob + 1):
            tr("chunk %d" % j)
            if j + 1 < n_glob:
                # chunk j+1 goes into the queues before the tail (and the downloads) of chunk j-1 do: queued behind them,
                # the copy of chunk j+1 was seen to wait for those downloads, and its ResNet with it
                encode(j + 1, ahead.pop(j + 1))
                ahead[j + 2] = stage_ahead(j + 2)
            if j >= 1:
                # tail of the output frames of global chunk j-1: their windows reach margin frames into the NEXT chunk of the same
                # video (global chunk j, encoded just above) -- or, for a video's last chunk, into its padding
                vi, k = G[j - 1]
                v = V[vi]
                last = k == v["n_chunks"] - 1
                with torch.cuda.stream(self.s_tail):
                    self.s_tail.wait_event(enc[j - 1 if last else j])
                    # (a one-chunk video's tail as two halves, the first half's download under the second: measured SLOWER, 0.88 ms per
                    #  half against 1.0 ms for the whole -- the tail is ~60 short launches -- profiles/r04e_host_surface.log)
                    self._tail((j - 1) % 2, k * C, min(v["N"], (k + 1) * C), v["N"], v["phi"], recs, out_free, v["host"], fields, keys, ar_T)
                tr(" tail queued")
        self.s_tail.synchronize()
        for f in out_free:
            if f is not None:
                f.result()                                           # the last downloads are queued (re-raises a failure)
        self.s_out.synchronize()
        cur.synchronize()
        tr("done")
        if trace is not None:
            print("entry -> first trace point: %.2f ms" % ((trace[0][1] - t_entry) * 1e3))
            t0 = trace[0][1]
            print("\n".join("%8.2f ms %s" % ((t - t0) * 1e3, tag) for tag, t in trace))
            print("\n".join("%8.2f ms (device) %s" % (gpu_marks[0][1].elapsed_time(e), tag) for tag, e in gpu_marks[1:]))
        for vi, v in enumerate(V):
            if v is not None:
                results[vi] = {k: v["host"][k].numpy() for k in keys}
        return results

    def _tail(self, slot, o0, o1, N, phi, recs, out_free, host, fields, keys, ar_T):
        """output frames [o0, o1) (o0 a multiple of g): windows -> records in recs[slot] -> the downloader thread"""
        t, eng, dev = self.t, self.eng, self.dev
        C, g, margin = self.chunk, self.g, self.margin
        cur = torch.cuda.current_stream(dev)
        j = o0
        w0, w1 = o0 // g, (o1 + g - 1) // g
        f = (torch.arange(w0, w1, device=dev)[:, None] * g - margin) + ar_T[None, :]
        idx = torch.where((f >= 0) & (f < N), f, torch.full_like(f, N))
        if out_free[slot] is not None:
            cur.wait_event(out_free[slot].result())                 # the previous copy-out of this buffer is done
        rec = recs[slot]
        if self._poison:                                            # development switch: a record row nobody wrote shows up as NaN on the host
            rec.fill_(float("nan"))
        t.predict_strips_records(phi[idx], o1 - o0, out=rec)
        ready = torch.cuda.Event(blocking=True)
        ready.record(cur)
        self._mark("tail %d done" % j, cur)

        def download():
            # issued by the downloader thread once the records EXIST: copies queued while their producer is still running
            # were seen to hold back the uploads queued after them (the copy queues serve in submission order)
            ready.synchronize()
            with torch.cuda.stream(self.s_out):
                for k in keys:
                    shp, off, size = fields[k]
                    host[k][o0:o1].copy_(rec[:o1 - o0, off:off + size].reshape((o1 - o0,) + shp), non_blocking=True)
                done = torch.cuda.Event()
                done.record(self.s_out)
                self._mark("download %d done" % j, self.s_out)
            return done
        out_free[slot] = self._downloader.submit(download)
