bash tools/pmc_conv_study.sh 257 5,7,8 bf16x3
python tools/pmc_conv_study.py gpurun_out > gpurun_out/r02c_conv_pmc_bf16x3.log 2>&1
cat gpurun_out/r02c_conv_pmc_bf16x3.log
find gpurun_out/cs_* -name "*.csv" -size +1M -delete
