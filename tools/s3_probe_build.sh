#!/bin/bash
# Development aid: builds of the library whose 3x3 stream kernel (csrc/conv3x3_stream.hip) drops pieces of its loop at COMPILE time
# (bits of S3_PROBE_BITS, see the source) and stamps s_memtime around its phases.  usage: tools/s3_probe_build.sh 0 1 2 ...
# run with HMMR_LIB_PATH=human_dynamics_amd/libhmmr_hip_s3probe_<bits>.so python tools/s3_check.py ...
set -e
cd "$(dirname "$0")/.."
C=human_dynamics_amd/csrc
objs=$(ls $C/*.o | grep -v conv3x3_stream.o)
for b in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -I $C -Wall -Wno-unused-function -Wno-unused-variable -fno-slp-vectorize -DHMMR_GEMM_PROBE -DS3_PROBE_BITS=$b -x hip -c $C/conv3x3_stream.hip -o /tmp/s3_probe_$b.o &
done
wait
for b in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o human_dynamics_amd/libhmmr_hip_s3probe_$b.so /tmp/s3_probe_$b.o $objs
done
ls human_dynamics_amd/libhmmr_hip_s3probe_*.so
