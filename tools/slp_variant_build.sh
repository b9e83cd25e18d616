#!/bin/bash
# Development aid (round 6): a build of the library whose csrc/smpl.hip is compiled WITH the SLP vectoriser (plain -O3: packed fp32 in
# smpl_pose_kernel), everything else as shipped -- for tools/tail_race_check.py through HMMR_LIB_PATH=human_dynamics_amd/libhmmr_hip_slp.so.
set -e
cd "$(dirname "$0")/.."
C=human_dynamics_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -I $C -Wall -Wno-unused-function "$@" -x hip -c $C/smpl.hip -o /tmp/smpl_slp.o --save-temps=obj 2>/dev/null || \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -I $C -Wall -Wno-unused-function "$@" -x hip -c $C/smpl.hip -o /tmp/smpl_slp.o
objs=$(ls $C/*.o | grep -v "/smpl.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o human_dynamics_amd/libhmmr_hip_slp.so /tmp/smpl_slp.o $objs
ls -la human_dynamics_amd/libhmmr_hip_slp.so
