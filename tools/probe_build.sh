#!/bin/bash
# Development aid: a second build of the library whose GEMM K loop can drop MFMAs / loads / barriers
# (HMMR_GEMM_PROBE=1|2|4 at run time, HMMR_LIB_PATH=human_dynamics_amd/libhmmr_hip_probe.so).  Never shipped.
set -e
cd "$(dirname "$0")/.."
C=human_dynamics_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -I $C -Wall -Wno-unused-function -DHMMR_GEMM_PROBE -x hip -c $C/gemm_conv.hip -o /tmp/gemm_conv_probe.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -I $C -Wall -Wno-unused-function -fno-slp-vectorize -DHMMR_GEMM_PROBE -x hip -c $C/unit_pair.hip -o /tmp/unit_pair_probe.o
objs=$(ls $C/*.o | grep -v gemm_conv.o | grep -v unit_pair.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o human_dynamics_amd/libhmmr_hip_probe.so /tmp/gemm_conv_probe.o /tmp/unit_pair_probe.o $objs
ls -la human_dynamics_amd/libhmmr_hip_probe.so
