python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bf16x3.py -m gpu -x -q -k "conv_gemm or tile_choice or temporal" 2>&1 | tail -3
B="python bench.py --only-main --no-cpu-baseline --no-pcie --steps 10"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['resnet_pass_ms'], d['roofline']['frac'])" "$1"; }
$B | pick default
HMMR_TUNE_TILES=5,6,3,1,2,7 $B | pick no8
$B --serial | pick serial
