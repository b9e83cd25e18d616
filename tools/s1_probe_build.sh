#!/bin/bash
# Development aid: builds of the library whose 1x1 stream kernel (csrc/conv1x1_stream.hip) drops pieces of its loop at COMPILE time
# (bits of S1_PROBE_BITS, see the source).  usage: tools/s1_probe_build.sh 1 2 4 ...
# run with HMMR_LIB_PATH=human_dynamics_amd/libhmmr_hip_s1probe_<bits>.so python tools/s1_check.py ...
set -e
cd "$(dirname "$0")/.."
C=human_dynamics_amd/csrc
objs=$(ls $C/*.o | grep -v conv1x1_stream.o)
for b in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -I $C -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable -fno-slp-vectorize -DS1_PROBE_BITS=$b -x hip -c $C/conv1x1_stream.hip -o /tmp/s1_probe_$b.o &
done
wait
for b in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o human_dynamics_amd/libhmmr_hip_s1probe_$b.so /tmp/s1_probe_$b.o $objs
done
ls human_dynamics_amd/libhmmr_hip_s1probe_*.so
