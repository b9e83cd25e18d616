"""Build libhmmr_hip.so (gfx950) in-tree with hipcc.

    python -m human_dynamics_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but
travels with the working tree to the GPU box.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libhmmr_hip.so")
SOURCES = ["api.cpp", "pack.cpp", "gemm_conv.hip", "conv3x3_stream.hip", "conv1x1_stream.hip", "stem.hip", "bottleneck.hip", "bottleneck_split.hip", "unit_pair.hip", "b1_unit.hip", "resnet.hip", "temporal.hip", "ief.hip", "smpl.hip", "eval_metrics.hip", "preprocess.hip", "handoff.hip", "probe.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize (every file, round 5): under plain -O3 the SLP vectoriser packs adjacent scalar fp32 operations into v_pk_*_f32 with
# op_sel shuffles.  In smpl_pose_kernel's kinematic chain that code produced WRONG translations for the last quarter of a wave (lanes
# 48-55: joints 16-23 of every odd instance) in 20-60 % of the launches that ran beside ResNet kernels of another stream, and in none that
# ran alone -- a rare wrong frame in Tester.predict_all_images' streamed path (profiles/r05_smpl_pose_packed_fp32.log; tests/
# test_gpu_stress.py::test_tail_beside_the_resnet_is_deterministic).  Without the flag-made packing: 0 of 600.  The flag changes no
# arithmetic (the same IEEE operations, unpacked); the one-wave-per-SIMD kernels had it already because the packing costs them issue slots.
# Round 6 found the instruction: packed-fp32 arithmetic whose LOW result reads the HIGH register of source 1 (op_sel bit 1) gets 0.0 for it
# in a wave's last 16 lanes while a neighbour's MFMAs are in flight (isa_check.py) -- so the flag is only the usual way such an
# instruction gets made, and build() checks the linked library's ISA for the form itself (`isa_check.unsafe_forms`).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-I", INCLUDE, "-I", CSRC,
         "-Wall", "-Wno-unused-function"]


# per-file flags: the host-side packers fold in double exactly as written (no contraction of a * b + c into one fma: numpy does not)
EXTRA_FLAGS = {"pack.cpp": ["-ffp-contract=off"]}


def _newer(src, dst):
    return (not os.path.exists(dst)) or os.path.getmtime(src) > os.path.getmtime(dst)


def _deps():
    return [os.path.join(CSRC, "common.h"), os.path.join(INCLUDE, "hmmr_hip.h")]


def build(force=False, verbose=True):
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _newer(s, o) or any(_newer(d, o) for d in _deps()):
            lang = ["-x", "hip"] if src.endswith(".hip") else []
            jobs.append([HIPCC] + FLAGS + EXTRA_FLAGS.get(src, []) + lang + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stdout))
        if verbose and r.stdout.strip():
            print(r.stdout)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(run, jobs))
    if jobs or force or not os.path.exists(LIB):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
        from . import isa_check
        unsafe = isa_check.unsafe_forms(LIB)
        if unsafe:
            os.replace(LIB, LIB + ".rejected")
            raise RuntimeError("the linked library holds packed-fp32 instructions gfx950 gets wrong beside MFMAs (isa_check.py, DESIGN 4.6); "
                               "kept as %s.rejected:\n%s" % (LIB, "\n".join("  %d x %s %s in %s" % (n, op, mods, kern)
                                                                             for (kern, op, mods), n in sorted(unsafe.items()))))
        if verbose:
            print("isa_check: %s" % ("no disassembler (%s) or no gfx950 code object found in the library: NOT checked" % isa_check.OBJDUMP if unsafe is None
                                     else "no packed-fp32 instruction reads source 1's high register into its low result"), flush=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
