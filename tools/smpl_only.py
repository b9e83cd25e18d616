"""Run the SMPL stage alone (for rocprofv3 counter passes; development aid)."""
import sys
import torch
sys.path.insert(0, ".")
from human_dynamics_amd import assets
from human_dynamics_amd.engine import HmmrEngine
eng = HmmrEngine(None, assets.make_synthetic_smpl(2), dtype="bf16")
m = int(sys.argv[1]) if len(sys.argv) > 1 else 256
th = torch.randn((m, 72), device="cuda") * 0.3; be = torch.randn((m, 10), device="cuda"); cm = torch.rand((m, 3), device="cuda")
for _ in range(6):
    eng.smpl(th, be, cm)
torch.cuda.synchronize()
