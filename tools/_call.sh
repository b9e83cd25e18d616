python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bf16x3.py -m gpu -x -q -k "conv_gemm or tile_choice" 2>&1 | tail -2
python tools/conv_bench.py 257 bf16x3 5,7,8 b2.conv2,b3.conv1,b3.conv2,b4.conv1,b4.conv2,b4.conv3 2>&1 | grep layer
python tools/conv_bench.py 257 bf16 5,7,8 b3.conv2,b4.conv2 2>&1 | grep layer
