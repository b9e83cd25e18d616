"""On the GPU: the packed-fp32 forms the shipped library holds are sound beside MFMAs; the form the SLP build of smpl_pose_kernel held is what
tools/probes/pk_fma_opsel.hip counts wrong results for (human_dynamics_amd/isa_check.py)."""
import os
import re
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_shipped_packed_fp32_forms_are_sound_beside_mfma(tmp_path):
    exe = str(tmp_path / "pk_fma_opsel")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-w", os.path.join(HERE, "tools", "probes", "pk_fma_opsel.hip"), "-o", exe], check=True,
                   capture_output=True, timeout=600)
    out = subprocess.run([exe, "--quick"], check=True, capture_output=True, text=True, timeout=300).stdout
    wrong = {m.group(1): int(m.group(2)) for m in re.finditer(r"^(\w+), full EXEC\s+noise\s+7: (\d+) wrong", out, re.M)}
    assert set(wrong) == {"fma_plain", "fma_lo1", "fma_swap1"}, out
    # plain, and source 1's low register to both halves (csrc/smpl.hip's hand-written blend: the only packed fp32 the library holds): exact
    assert wrong["fma_plain"] == 0 and wrong["fma_lo1"] == 0, out
    # source 1's halves swapped (what plain -O3 made of the chain step): measured wrong in ~7 % of the wave-instructions on the boxes of round 6;
    # reported, not asserted -- a part that does not show it is not a failure of this library
    print("v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,0,1] beside MFMA + LDS + loads: %d wrong wave-lane results" % wrong["fma_swap1"])
