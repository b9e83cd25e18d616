#!/bin/bash
# Which part of a conv_gemm launch bounds each 1x1 shape (dev aid, run on the GPU box after tools/probe_build.sh ran HERE):
# full kernel, K loop without MFMAs (1), without operand loads after the first stage (2), without both (3 = prologue + epilogue
# + barriers), and without anything (7).   bash tools/probe_study.sh [frames] [tiles] > gpurun_out/probe_study.log
N=${1:-257}; T=${2:-5,6,8}
export HMMR_LIB_PATH=$PWD/human_dynamics_amd/libhmmr_hip_probe.so
for p in 0 1 2 3 7; do
  echo "== HMMR_GEMM_PROBE=$p"
  HMMR_GEMM_PROBE=$p timeout 120 python tools/conv_bench.py $N f16x3 $T "b1.conv3,b2.conv3,b3.conv1,b3.conv3,b4.conv1,b4.conv3,b3.conv2" 2>&1 | grep "^{"
done
