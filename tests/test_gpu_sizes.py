"""GPU parity at the sizes BASELINE.json quotes (configs 2, 3 and 4), with the tile table the
autotuner picks at those sizes, and the tight versions of the reduced-precision checks:
the HIP path against an oracle that rounds at the same storage points (oracle.quantize).

The float64 oracle is run on a sample of the frames / windows so that the file stays within a
couple of minutes; every sampled element is compared, not a statistic.
"""
import numpy as np
import pytest
import torch

from conftest import Config
from human_dynamics_amd import assets

pytestmark = pytest.mark.gpu
F64 = torch.float64


def _oracle():
    from oracle import hmmr_oracle as O
    return O


@pytest.fixture(scope="module")
def engines(weights, smpl_consts, gpu_device):
    from human_dynamics_amd.engine import HmmrEngine
    return {dt: HmmrEngine(weights, smpl_consts, dtype=dt, device=gpu_device) for dt in ("f16x3", "bf16", "f32")}


# --------------------------------------------------------------------------- config 2
@pytest.mark.parametrize("dt,emulate,tol_rel", [("f16x3", None, 5e-5), ("f16x3", "f16x3", 5e-6), ("f32", None, 5e-6)])
def test_config2_resnet_batch64(engines, weights, dt, emulate, tol_rel):
    """BASELINE config 2: batch = 64 frames through the ResNet; phi against the float64 oracle on every
    8th frame.  emulate = the oracle rounds operands and stored tensors exactly where that mode does."""
    O = _oracle()
    frames = assets.make_synthetic_frames(64, seed=21)
    phi = engines[dt].resnet(frames).cpu().numpy()
    assert phi.shape == (64, 2048) and np.isfinite(phi).all()
    idx = np.arange(0, 64, 8)
    if emulate is None:
        ref = O.resnet_v2_50(frames[idx], weights, F64).numpy()
    else:
        ref = O.resnet_v2_50_emulated(frames[idx], weights, emulate).numpy()
    rel = np.linalg.norm(phi[idx] - ref) / np.linalg.norm(ref)
    err = np.abs(phi[idx] - ref).max()
    print("config 2 [%s vs oracle%s]: phi rel-L2 %.3e max-abs %.3e" % (dt, "/" + emulate if emulate else "", rel, err))
    assert rel < tol_rel


def test_config2_resnet_batch64_bf16_error_is_the_predicted_size(engines, weights):
    """bf16 operands: 53 layers of 8-bit roundings in a ReLU network decorrelate under ANY perturbation (a
    relative 1e-7 nudge of the pre-rounding values moves phi by 1.8e-3 in the emulating oracle itself), so an
    element-wise match with the emulation is not attainable.  What is: the HIP path's error equals, within
    +-40 %, the error the rounding model predicts, and the HIP-vs-emulation distance is below both."""
    O = _oracle()
    frames = assets.make_synthetic_frames(64, seed=21)
    phi = engines["bf16"].resnet(frames).cpu().numpy()
    idx = np.arange(0, 64, 8)
    exact = O.resnet_v2_50(frames[idx], weights, F64).numpy()
    emu = O.resnet_v2_50_emulated(frames[idx], weights, "bf16").numpy()
    nrm = np.linalg.norm(exact)
    e_hip, e_emu, d = (np.linalg.norm(phi[idx] - exact) / nrm, np.linalg.norm(emu - exact) / nrm,
                       np.linalg.norm(phi[idx] - emu) / nrm)
    print("config 2 [bf16]: HIP vs exact %.3e, emulation vs exact %.3e, HIP vs emulation %.3e" % (e_hip, e_emu, d))
    assert 0.6 * e_emu < e_hip < 1.4 * e_emu and e_hip < 8e-3
    assert d < 0.75 * max(e_hip, e_emu)


# --------------------------------------------------------------------------- config 3
def test_config3_64_windows(weights, smpl_consts, gpu_device):
    """BASELINE config 3: 64 windows x T=20 (1280 frames): ResNet + f_movie + IEF in the default
    (f16x3) mode; omega_0 and both delta omegas of three sampled windows against the float64 oracle."""
    from human_dynamics_amd.evaluation.tester import Tester
    O = _oracle()
    frames = assets.make_synthetic_frames(1280, seed=31).reshape(64, 20, 224, 224, 3)
    t = Tester(Config(batch_size=64), weights=weights, smpl=smpl_consts, device=gpu_device)
    out = t.predict_device(torch.from_numpy(frames).to(gpu_device))
    om = out["omegas"].cpu().numpy()
    omd = out["omegas_delta"].cpu().numpy()
    assert om.shape == (64, 20, 85) and omd.shape == (64, 20, 2, 85)
    sel = [0, 29, 63]
    ot = O.OracleTester(weights, smpl_consts, batch_size=len(sel), dtype=F64)
    ref = ot.predict(frames[sel])
    e0 = np.abs(om[sel] - ref["omegas"]).max()
    e1 = np.abs(omd[sel] - ref["omegas_delta"]).max()
    ev = np.abs(out["verts"].cpu().numpy()[sel] - ref["verts"]).max()
    print("config 3 [f16x3]: omegas %.3e omegas_delta %.3e verts %.3e" % (e0, e1, ev))
    assert e0 < 1e-4 and e1 < 1e-4 and ev < 1e-4


# --------------------------------------------------------------------------- config 4
def _windows_of(frames, starts, T=20, margin=6):
    """The reference's padded windows (tester.py:281-295) that keep output frames [s, s+8)."""
    N = len(frames)
    out = np.zeros((len(starts), T) + frames.shape[1:], np.float32)
    for k, s in enumerate(starts):
        for j in range(T):
            f = s - margin + j
            if 0 <= f < N:
                out[k, j] = frames[f]
    return out


C4_STARTS = [0, 96, 168, 248]


@pytest.fixture(scope="module")
def config4_ref(weights, smpl_consts):
    O = _oracle()
    frames = assets.make_synthetic_frames(256, seed=41)
    ot = O.OracleTester(weights, smpl_consts, batch_size=len(C4_STARTS), dtype=F64)
    return frames, ot.predict(_windows_of(frames, C4_STARTS))


@pytest.mark.parametrize("dt,tol", [("f16x3", 1e-4), ("f32", 1e-4)])
def test_config4_256_frame_video(weights, smpl_consts, gpu_device, config4_ref, dt, tol):
    """BASELINE config 4: a 256-frame video through predict_all_images (B=8, T=20 -> 32 windows, the
    autotuned tile table of a 257-frame ResNet pass): vertices / joints of four sampled windows (first,
    last, two inside) within 1e-4 of the float64 oracle run on the reference's own padded windows."""
    from human_dynamics_amd.evaluation.tester import Tester
    frames, ref = config4_ref
    t = Tester(Config(batch_size=8), weights=weights, smpl=smpl_consts, dtype=dt, device=gpu_device)
    res = t.predict_all_images(frames)
    assert res["verts"].shape == (256, 6890, 3) and res["verts_delta"].shape == (256, 2, 6890, 3)
    starts = C4_STARTS
    errs = {}
    for k in ("verts", "joints", "kps", "omegas", "verts_delta", "joints_delta"):
        got = np.stack([res[k][s:s + 8] for s in starts])
        errs[k] = float(np.abs(got - ref[k][:, 6:14]).max())
    print("config 4 [%s]: " % dt + " ".join("%s %.2e" % kv for kv in sorted(errs.items())))
    for k, e in errs.items():
        assert e < tol, (k, e)


def test_bf16_mode_against_its_emulation(weights, smpl_consts, gpu_device):
    """The bf16-operand throughput mode end to end, next to the float64 oracle that rounds at the same
    storage points."""
    from human_dynamics_amd.evaluation.tester import Tester
    O = _oracle()
    frames = assets.make_synthetic_frames(20, seed=1)[None]
    t = Tester(Config(batch_size=1), weights=weights, smpl=smpl_consts, dtype="bf16", device=gpu_device)
    res = t.predict(frames)
    emu = O.OracleTester(weights, smpl_consts, batch_size=1, dtype=F64, emulate="bf16").predict(frames)
    exact = O.OracleTester(weights, smpl_consts, batch_size=1, dtype=F64).predict(frames)
    e_emu = {k: float(np.abs(res[k] - emu[k]).max()) for k in ("verts", "joints", "omegas")}
    e_ref = {k: float(np.abs(res[k] - exact[k]).max()) for k in ("verts", "joints", "omegas")}
    print("bf16 end to end: vs its emulation %s; vs the unrounded graph %s" % (e_emu, e_ref))
    e_pred = {k: float(np.abs(emu[k] - exact[k]).max()) for k in ("verts", "joints", "omegas")}
    print("bf16 end to end: what the rounding model predicts %s" % e_pred)
    # (element-wise agreement with the emulation is not attainable, see the config-2 test above: the gate is the
    #  predicted error SIZE -- 2.5e-2 m instead of round 1's 0.25 m)
    for k in ("verts", "joints"):
        assert e_ref[k] < 3.0 * e_pred[k] and e_ref[k] < 2.5e-2, (k, e_ref[k], e_pred[k])


# --------------------------------------------------------------------------- fused stem
@pytest.mark.parametrize("dt", ["bf16", "f16x3", "f32"])
def test_fused_stem_equals_three_kernel_route(weights, gpu_device, dt):
    """stem.hip (7x7/2 conv + bias + pool1 + unit_1 preact [+ unit_1 conv1] in one launch) against the
    re-pack + implicit-GEMM + pool route, through the whole ResNet: image-edge tiles, interior tiles
    and the n_zero tail all feed phi, bit for bit."""
    from human_dynamics_amd.engine import HmmrEngine, set_debug
    eng = HmmrEngine(weights, None, dtype=dt, device=gpu_device, autotune=False)
    x = torch.from_numpy(assets.make_synthetic_frames(9, seed=13)).to(gpu_device)
    x[3] = 0.0
    x[4, :7, :, :] = 5.0                     # strong top rows / left columns: edge tiles matter
    x[4, :, :7, :] = -5.0
    x[5, -9:, :, :] = 4.0
    x[5, :, -9:, :] = -4.0
    try:
        outs = {}
        for name, route, no_c1 in (("three_kernel", 1, 0), ("fused", 2, 0), ("fused_no_conv1", 2, 1), ("default", 0, 0)):
            set_debug(stem_route=route, stem_no_conv1=no_c1)
            outs[name] = eng.resnet(x, n_zero=2).clone()
    finally:
        set_debug()
    ref = outs["three_kernel"]
    assert float(ref.abs().max()) > 0.1
    for name, o in outs.items():
        assert torch.equal(o, ref), "%s stem route differs from the three-kernel route: max |d| %.3e" % (
            name, float((o - ref).abs().max()))
    assert torch.equal(ref[-1], ref[-2]) and torch.equal(ref[-1], ref[3])    # zero images: tail == explicit


# --------------------------------------------------------------------------- config 5 on one GPU
def test_config5_4096_frame_video_on_one_gpu(weights, smpl_consts, gpu_device):
    """BASELINE config 5's code path (`bench.py --gpus N --video-frames 4096`: ShardPlan of ONE 4096-frame video,
    ShardedPredictor with the pipelined tail, 1024-frame ResNet passes) executed on hardware with N = 1: the packed
    records of one step equal Tester.predict_all_images on the same 4096 device-resident frames bit for bit, and the
    command line itself runs and reports the strong-scaling line."""
    import json
    import os
    import subprocess
    import sys
    from human_dynamics_amd import dist as hd
    from human_dynamics_amd.evaluation.tester import Tester
    n = 4096
    t = Tester(Config(batch_size=8), weights=weights, smpl=smpl_consts, dtype="f16x3", device=gpu_device)
    plan = hd.ShardPlan(n, 8, 20, 13, 1, 0)
    assert (plan.f0, plan.f1, plan.o0, plan.o1, plan.w1 - plan.w0) == (0, n, 0, n, 512)
    gen = torch.Generator(device=gpu_device)
    gen.manual_seed(1234)                                      # bench.py's rank-0 span
    span = torch.rand((n, 224, 224, 3), generator=gen, device=gpu_device) * 2 - 1
    pred = hd.ShardedPredictor(t, n, 0, 1, pipeline=True, overlap_gather=True, step_streams=True)
    rec = pred.run(span)
    pred.finish()
    torch.cuda.synchronize()
    layout, _ = t.record_layout()
    got = hd.unpack_outputs(rec, layout)
    ref = t.predict_all_images(span)
    for k, v in ref.items():
        assert v.shape[0] == n and np.array_equal(got[k].cpu().numpy(), v), k
    del pred, rec, got, ref, span
    torch.cuda.empty_cache()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--video-frames", str(n), "--steps", "1", "--warmup", "0",
                        "--only-main", "--no-cpu-baseline", "--no-pcie", "--no-by-config", "--sustain", "0", "--dtype", "f16x3"], cwd=root, capture_output=True,
                       text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, r.stderr[-800:]
    js = json.loads(lines[-1])
    assert js["scaling"] == "strong" and js["frames_total"] == n and js["n_gpus"] == 1 and js["steps"] == 1
    assert js["config"]["resnet_frames_encoded_per_gpu"] == n + 1 and js["value"] > 2000


@pytest.mark.gpu
def test_bench_two_rank_code_path_on_one_gpu(gpu_device):
    """The N > 1 path of bench.py end to end -- `torch.distributed.run`, the shard plans of two ranks, the omegas gather of a strong-scaling
    run, `multi_gpu_fields` (both gather modes measured after the headline), the scaling-efficiency leg, one JSON line from rank 0 -- on a one-GPU
    box: both ranks on cuda:0 with the gloo backend (test hooks HMMR_BENCH_BACKEND / HMMR_BENCH_ONE_DEVICE; RCCL refuses two ranks on one
    device).  What it cannot show is RCCL's speed; that the line has every field the N > 1 contract names, it can."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    env = dict(os.environ, HMMR_BENCH_BACKEND="gloo", HMMR_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), "bench.py", "--gpus", "2", "--video-frames", "512", "--steps", "2", "--warmup", "1",
                        "--sustain", "0", "--dtype", "f16x3"], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-500:], r.stderr[-1500:])
    js = json.loads(lines[0])
    assert js["n_gpus"] == 2 and js["scaling"] == "strong" and js["frames_total"] == 512 and js["rccl_ranks"] == 2
    assert js["gather"] == "theta" and js["gather_requested"] == "auto" and js["gather_by_measurement"] in ("theta", "records")
    for k in ("single_video_ms_by_gather", "all_gather_ms_by_gather", "all_gather_bytes_by_gather"):
        assert set(js[k]) == {"records", "theta"} and all(v > 0 for v in js[k].values()), k
    assert js["single_video_ms"] == js["single_video_ms_by_gather"]["theta"] and js["all_gather_bytes"] == 512 * 255 * 4
    assert js["value"] > 1000 and js["per_gpu_fps"] * 2 == pytest.approx(js["value"], rel=1e-3) and js["scaling_efficiency"] > 0
    assert js["config"]["frames_per_gpu_per_step"] == 256 and "window-sharded x2" in js["config"]["parallelism"]


@pytest.mark.gpu
def test_bench_power_and_clock_samplers(weights, gpu_device):
    """bench.py's two reporting aids of round 6 on hardware: SmiSampler (socket power + the XCDs' clocks from the driver's metrics table, a host
    thread) around ResNet passes, ClockSampler (csrc/probe.hip hmmr_clock_probe: s_memtime against s_memrealtime in one wave) beside the bare
    MFMA stream in both operand forms -- plausible numbers, and the changing-operand stream is not faster than the constant one."""
    import time
    import bench
    from human_dynamics_amd import _lib as L
    from human_dynamics_amd.engine import HmmrEngine
    eng = HmmrEngine(weights, None, dtype="f16x3", device=gpu_device)
    x = torch.rand((64, 224, 224, 3), device=gpu_device) * 2 - 1
    for _ in range(3):
        eng.resnet(x)
    torch.cuda.synchronize()
    smi = bench.SmiSampler(torch.device(gpu_device).index or 0)
    smi.start()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        for _ in range(8):
            eng.resnet(x)
        torch.cuda.synchronize()
    p = smi.stop()
    if p is None:
        pytest.skip("amdsmi is not usable on this box")
    assert "error" not in p, p
    assert p["samples"] >= 10 and 200 < p["socket_w_mean"] < 2000 and 400 < p["gfxclk_mhz_mean"] < 2600 and p["joules"] > 50, p
    cus = torch.cuda.get_device_properties(gpu_device).multi_processor_count
    st = torch.cuda.current_stream().cuda_stream
    rates = {}
    for n8 in (2500, -2500):
        cs = bench.ClockSampler(eng)
        cs.start()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            L.check(eng.lib.hmmr_mfma_rate_probe(cus, n8, None, st), "hmmr_mfma_rate_probe")
        e1.record()
        c = cs.stop()
        rates[n8] = cus * 4 * 8 * 2500 * 32768.0 * 10 / (e0.elapsed_time(e1) * 1e-3)
        assert c is not None and 800 < c["mhz_median"] < 2600, c
    assert 0.8e15 < rates[-2500] <= rates[2500] * 1.02 and rates[2500] < 2.6e15, rates
