"""The boundary WITHOUT Python: tests/c_abi/pack_and_run.c -- a C program that fills hmmr_resnet_weights_t with the C-side
packer hmmr_pack_resnet from a dump of checkpoint variables, copies the blob with one hipMemcpy and calls hmmr_resnet50_fwd -- compiled with
hipcc against include/hmmr_hip.h + libhmmr_hip.so, run here, and compared with the Python mirror's features bit for bit."""
import os
import struct
import subprocess

import numpy as np
import pytest

from human_dynamics_amd import _lib as L
from human_dynamics_amd import assets

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["f16x3", "f32"])
def test_c_program_packs_and_runs_the_resnet(weights, gpu_device, tmp_path, dt):
    import torch
    from human_dynamics_amd.engine import HmmrEngine
    exe = str(tmp_path / "pack_and_run")
    pkg = os.path.join(ROOT, "human_dynamics_amd")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-x", "hip", os.path.join(ROOT, "tests", "c_abi", "pack_and_run.c"),
                        "-I", os.path.join(ROOT, "include"), "-L", pkg, "-lhmmr_hip", "-Wl,-rpath," + pkg, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    names = sorted(k for k in weights if k.startswith("resnet_v2_50/"))
    with open(str(tmp_path / "vars.bin"), "wb") as f:
        f.write(struct.pack("<i", len(names)))
        for k in names:
            a = np.ascontiguousarray(weights[k], np.float32)
            f.write(struct.pack("<i", len(k)) + k.encode() + struct.pack("<q", a.size) + a.tobytes())
    n = 5
    frames = assets.make_synthetic_frames(n, seed=17)
    frames.astype(np.float32).tofile(str(tmp_path / "frames.bin"))
    code = {"f32": L.HMMR_F32, "bf16": L.HMMR_BF16, "f16x3": L.HMMR_F16X3}[dt]
    r = subprocess.run([exe, str(tmp_path / "vars.bin"), str(tmp_path / "frames.bin"), str(n), str(code), str(tmp_path / "phi.bin")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "run flags 0" in r.stdout
    phi_c = np.fromfile(str(tmp_path / "phi.bin"), np.float32).reshape(n, 2048)
    eng = HmmrEngine(weights, None, dtype=dt, device=gpu_device, autotune=False)
    phi_py = eng.resnet(frames).cpu().numpy()
    assert float(np.abs(phi_py).max()) > 0.1
    assert np.array_equal(phi_c, phi_py)


def test_c_program_compiles_against_the_header(tmp_path):
    """no GPU: the C side of the boundary compiles (hipcc, host code only) and links against libhmmr_hip.so"""
    from human_dynamics_amd import build
    build.build(verbose=False)                              # (incremental: a no-op when the library is up to date)
    pkg = os.path.join(ROOT, "human_dynamics_amd")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-x", "hip", os.path.join(ROOT, "tests", "c_abi", "pack_and_run.c"),
                        "-I", os.path.join(ROOT, "include"), "-L", pkg, "-lhmmr_hip", "-Wl,-rpath," + pkg, "-o", str(tmp_path / "pack_and_run")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
