#!/bin/bash
# Development aid: a second build of the library with extra -D flags on gemm_conv.hip (e.g. -DHMMR_EPI_NT, -DHMMR_GEMM_PROBE),
# for A/B runs through HMMR_LIB_PATH=human_dynamics_amd/libhmmr_hip_<name>.so.  Never shipped.
#   bash tools/variant_build.sh <name> <flags...>
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
C=human_dynamics_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -I $C -Wall -Wno-unused-function -fno-slp-vectorize "$@" -x hip -c $C/gemm_conv.hip -o /tmp/gemm_conv_$NAME.o
objs=$(ls $C/*.o | grep -v gemm_conv.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o human_dynamics_amd/libhmmr_hip_$NAME.so /tmp/gemm_conv_$NAME.o $objs
ls -la human_dynamics_amd/libhmmr_hip_$NAME.so
