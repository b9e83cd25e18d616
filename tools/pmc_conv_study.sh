#!/bin/bash
# PMC study of hmmr_conv_gemm per layer shape (development aid): five separate counter passes (the last one: L2 hits / misses and fabric reads) of
# tools/conv_bench.py; summarise with tools/pmc_conv_study.py.  Run on the GPU box from the repo root:
#   bash tools/pmc_conv_study.sh [frames] [tiles] [dtype] [layers]
R=$PWD
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
i=0
for c in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
         "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" \
         "SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY SQ_WAVES SQ_WAIT_ANY" \
         "FETCH_SIZE"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/cs_$i
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/cs_$i -- python $R/tools/conv_bench.py ${1:-257} ${3:-bf16} ${2:-5} $4 > $R/gpurun_out/cs_$i.log 2>&1
done
