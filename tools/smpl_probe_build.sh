#!/bin/bash
# Development aid: builds of the library whose SMPL vertex kernel (csrc/smpl.hip: smpl_verts_split_kernel) drops its stores (1), its skinning
# sum (2) or its matrix-core loop (4) at COMPILE time.  usage: tools/smpl_probe_build.sh 1 2 4 ...;  HMMR_LIB_PATH=...libhmmr_hip_smplprobe_<bits>.so python tools/smpl_bench.py
set -e
cd "$(dirname "$0")/.."
C=human_dynamics_amd/csrc
objs=$(ls $C/*.o | grep -v "/smpl.o")
for b in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -I $C -Wall -Wno-unused-function -Wno-unused-variable -fno-slp-vectorize -DSMPL_PROBE_BITS=$b -x hip -c $C/smpl.hip -o /tmp/smpl_probe_$b.o &
done
wait
for b in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o human_dynamics_amd/libhmmr_hip_smplprobe_$b.so /tmp/smpl_probe_$b.o $objs
done
ls human_dynamics_amd/libhmmr_hip_smplprobe_*.so
