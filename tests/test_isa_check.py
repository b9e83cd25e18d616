"""The shipped library holds no packed-fp32 instruction of the form gfx950 gets wrong beside MFMAs (human_dynamics_amd/isa_check.py)."""
import os

import pytest

from human_dynamics_amd import isa_check
from human_dynamics_amd.build import LIB, build


def test_unsafe_form_predicate():
    # what tools/probes/pk_fma_opsel.hip measured (profiles/r06s_pk_fma_opsel.log): wrong whenever source 1's op_sel bit is set ...
    assert isa_check.is_unsafe("op_sel:[0,1,0] op_sel_hi:[1,0,1]")          # the SLP build's chain step
    assert isa_check.is_unsafe("op_sel:[0,1,0]")
    assert isa_check.is_unsafe("op_sel:[0,1] op_sel_hi:[1,0]")              # v_pk_mul_f32 / v_pk_add_f32
    assert isa_check.is_unsafe("op_sel:[0,1]")
    # ... and never otherwise: plain, the other sources' selections, source 1's LOW register broadcast (the hand-written SMPL blend)
    for mods in ("", "op_sel_hi:[1,0,1]", "op_sel:[1,0,0]", "op_sel:[1,0,0] op_sel_hi:[0,1,1]", "op_sel:[0,0,1] op_sel_hi:[1,1,0]", "op_sel_hi:[0,1,1]",
                 "op_sel:[1,0]", "op_sel_hi:[1,0]", "op_sel:[1,0] op_sel_hi:[0,1]"):
        assert not isa_check.is_unsafe(mods), mods


@pytest.mark.skipif(not os.path.exists(isa_check.OBJDUMP), reason="no llvm-objdump")
def test_shipped_library_has_no_unsafe_packed_fp32():
    build(verbose=False)
    assert len(isa_check.code_objects(LIB)) >= 16                             # one gfx950 code object per .hip translation unit
    forms, unsafe = isa_check.scan(LIB)
    assert unsafe == {}, unsafe
    # the packed-fp32 form the library does hold: the hand-written blend of csrc/smpl.hip (source 1's low register to both halves)
    assert forms.get(("v_pk_fma_f32", "op_sel_hi:[1,0,1]"), 0) > 0, forms


@pytest.mark.skipif(not os.path.exists(isa_check.OBJDUMP), reason="no llvm-objdump")
def test_scanner_finds_the_chain_step_of_a_plain_O3_build(tmp_path):
    """csrc/smpl.hip WITHOUT -fno-slp-vectorize: the 23 unrolled chain steps of smpl_pose_kernel each hold the instruction
    (`v_pk_fma_f32 ... op_sel:[0,1,0] op_sel_hi:[1,0,1]`) that tools/tail_race_check.py + the asm-level bisection put the wrong frames on."""
    import subprocess
    from human_dynamics_amd.build import CSRC, FLAGS, HIPCC
    obj = str(tmp_path / "smpl_slp.o")
    flags = [f for f in FLAGS if f != "-fno-slp-vectorize"]
    subprocess.run([HIPCC] + flags + ["-x", "hip", "-c", os.path.join(CSRC, "smpl.hip"), "-o", obj], check=True, capture_output=True)
    _, unsafe = isa_check.scan(obj)
    pose = {k: n for k, n in unsafe.items() if "smpl_pose_kernel" in k[0]}
    assert sum(pose.values()) >= 23 and all(op == "v_pk_fma_f32" for _, op, _ in pose), unsafe
