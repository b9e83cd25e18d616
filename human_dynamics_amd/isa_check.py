"""Which packed-fp32 instructions a built library holds, by source-selection form -- and the check that none of them is a form gfx950 gets wrong.

gfx950 reads ZERO for the HIGH register of source 1 when it feeds the LOW result of a packed-fp32 arithmetic instruction (`op_sel` bit 1 set:
`v_pk_fma_f32 D, A, B, C op_sel:[0,1,0] ...`, likewise `v_pk_mul_f32` / `v_pk_add_f32 ... op_sel:[0,1]`) in the last 16 lanes of a wave, now and
then, while another wave of the CU has MFMAs in flight -- never alone (tools/probes/pk_fma_opsel.hip, profiles/r06s_pk_fma_opsel.log; DESIGN.md
section 4.6).  Plain -O3 gets such instructions from the SLP vectoriser (the chain step of csrc/smpl.hip's smpl_pose_kernel: a wrong frame in ~1 %
of the streamed calls, round 5); the build passes -fno-slp-vectorize, and `unsafe_forms()` is the proof, on the ISA of the library that was
actually linked, that no kernel holds one -- whatever compiler pass might make it.  `build.build()` runs it after every link and
tests/test_isa_check.py on the shipped library.
"""
import os
import re
import struct
import subprocess
import tempfile

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
OBJDUMP = os.environ.get("HMMR_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
PK_F32 = re.compile(r"\b(v_pk_(?:fma|mul|add)_f32)\b(.*)")
SEL = re.compile(r"op_sel:\[([01,]+)\]")


def code_objects(path):
    """The gfx950 code objects bundled into a host library / object (one uncompressed clang offload bundle per translation unit)."""
    blob = open(path, "rb").read()
    out, at = [], 0
    while True:
        at = blob.find(MAGIC, at)
        if at < 0:
            return out
        (n,) = struct.unpack_from("<Q", blob, at + len(MAGIC))
        p = at + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx950" in triple and size:
                out.append(blob[at + off:at + off + size])
        at += len(MAGIC)


def is_unsafe(mods):
    """True for the modifier text of a packed-fp32 arithmetic instruction whose LOW result reads the HIGH register of source 1."""
    m = SEL.search(mods)
    sel = [int(v) for v in m.group(1).split(",")] if m else []
    return len(sel) > 1 and sel[1] == 1


def scan(path):
    """({(mnemonic, modifiers): count} of the packed-fp32 arithmetic of every kernel, {(kernel, mnemonic, modifiers): count} of the unsafe ones)."""
    forms, unsafe = {}, {}
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            txt = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", f.name], capture_output=True, text=True, check=True).stdout
        kern = "?"
        for line in txt.splitlines():
            if line.endswith(">:"):
                kern = line.split("<")[-1][:-2]
                continue
            m = PK_F32.search(line)
            if not m:
                continue
            mods = " ".join(re.findall(r"op_sel(?:_hi)?:\[[01,]+\]", m.group(2)))
            key = (m.group(1), mods)
            forms[key] = forms.get(key, 0) + 1
            if is_unsafe(mods):
                unsafe[(kern, m.group(1), mods)] = unsafe.get((kern, m.group(1), mods), 0) + 1
    return forms, unsafe


def unsafe_forms(path):
    """The unsafe packed-fp32 instructions of the library at `path` ({} = none), or None when there is no disassembler to look with."""
    if not os.path.exists(OBJDUMP) or not code_objects(path):      # (no bundle found: a format this parser does not know -- "not checked", never "clean")
        return None
    return scan(path)[1]
