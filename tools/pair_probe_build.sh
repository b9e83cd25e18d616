#!/bin/bash
# Development aid: builds of the library whose unit-pair kernel (csrc/unit_pair.hip) drops pieces of its loop at COMPILE time
# (bits of HMMR_PAIR_PROBE_BITS, see the source) and stamps s_memtime around its phases.  usage: tools/pair_probe_build.sh 0 1 8 ...
set -e
cd "$(dirname "$0")/.."
C=human_dynamics_amd/csrc
objs=$(ls $C/*.o | grep -v unit_pair.o)
for b in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -I $C -Wall -Wno-unused-function -Wno-unused-variable -fno-slp-vectorize -DHMMR_GEMM_PROBE -DHMMR_PAIR_PROBE_BITS=$b $PAIR_PROBE_EXTRA -x hip -c $C/unit_pair.hip -o /tmp/unit_pair_probe_$b$PAIR_PROBE_TAG.o &
done
wait
for b in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o human_dynamics_amd/libhmmr_hip_probe_$b$PAIR_PROBE_TAG.so /tmp/unit_pair_probe_$b$PAIR_PROBE_TAG.o $objs
done
ls human_dynamics_amd/libhmmr_hip_probe_*.so
