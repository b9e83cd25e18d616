#!/bin/bash
# Development aid (round 6): a build of the library whose csrc/smpl.hip is compiled WITH the SLP vectoriser (plain -O3) and whose device
# assembly is then EDITED before it is assembled -- the bisection that put the wrong frames of that build on one instruction form.
#   tools/asm_variant_build.sh TAG [SED_EXPRESSION]     -> human_dynamics_amd/libhmmr_hip_slp_TAG.so   (no expression: the unedited SLP build)
# e.g. the variant that repairs joints 1 .. 22 (scalar FMAs in place of the chain step's packed FMA with source 1's halves swapped):
#   tools/asm_variant_build.sh C 's/^\tv_pk_fma_f32 v\[6:7\], v\[64:65\], v\[46:47\], v\[38:39\] op_sel:\[0,1,0\] op_sel_hi:\[1,0,1\]$/\tv_fma_f32 v6, v64, v47, v38\n\tv_fma_f32 v7, v65, v46, v39/'
# then  HMMR_LIB_PATH=human_dynamics_amd/libhmmr_hip_slp_C.so DBG_DUMP=1 DBG_MODE="smpl only beside the resnet" python tools/tail_race_check.py
# How: hipcc -save-temps leaves the device .s; the remaining steps of `hipcc -###` (device assembler, lld, bundler, host compile) are re-run on the edit.
set -e
TAG=$1; EXPR=$2
REPO="$(cd "$(dirname "$0")/.." && pwd)"
C=$REPO/human_dynamics_amd/csrc
W=$(mktemp -d /tmp/hmmr_asm_XXXX)
cd $W
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I $REPO/include -I $C -Wall -Wno-unused-function -x hip -c $C/smpl.hip -o smpl_slp.o -save-temps"
/opt/rocm/bin/hipcc $FL 2>/dev/null
/opt/rocm/bin/hipcc $FL -### 2>&1 | grep '^ "' > cmds.txt
if [ -n "$EXPR" ]; then
    cp smpl-hip-amdgcn-amd-amdhsa-gfx950.s orig.s
    sed "$EXPR" orig.s > smpl-hip-amdgcn-amd-amdhsa-gfx950.s
    echo "edited lines: $(diff orig.s smpl-hip-amdgcn-amd-amdhsa-gfx950.s | grep -c '^<')"
    tail -n +4 cmds.txt > redo.sh       # from the device assembler on (lines 1-3: preprocess, compile to bitcode, bitcode to .s)
    bash -e redo.sh
fi
objs=$(ls $C/*.o | grep -v "/smpl.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $REPO/human_dynamics_amd/libhmmr_hip_slp_$TAG.so smpl_slp.o $objs
ls -la $REPO/human_dynamics_amd/libhmmr_hip_slp_$TAG.so
rm -rf $W
