#!/bin/bash
# Development aid: a second build of the library with csrc/b1_unit.hip compiled under -DB1_PROBE_BITS=<bits> (and any further flags), for
# A/B runs through HMMR_LIB_PATH=human_dynamics_amd/libhmmr_hip_b1p<name>.so.  Never shipped; results are garbage for bits != 0.
#   bash tools/b1_probe_build.sh <name> <flags...>
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
C=human_dynamics_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -I $C -Wall -Wno-unused-function -fno-slp-vectorize "$@" -x hip -c $C/b1_unit.hip -o /tmp/b1_unit_$NAME.o
objs=$(ls $C/*.o | grep -v b1_unit.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o human_dynamics_amd/libhmmr_hip_b1p$NAME.so /tmp/b1_unit_$NAME.o $objs
ls -la human_dynamics_amd/libhmmr_hip_b1p$NAME.so
