python -m pytest tests/test_gpu_bf16x3.py tests/test_gpu_sizes.py tests/test_reference_golden.py -m gpu -x -q -k "second_operand or folded or config2 or config4 or split or resnet" 2>&1 | tail -12
B="python bench.py --only-main --no-cpu-baseline --no-pcie --steps 10"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['resnet_pass_ms'], d['roofline']['frac'])" "$1"; }
$B | pick folded
HMMR_FOLD_SC=0 $B | pick separate
python tools/layer_table.py 257 bf16x3 3 2>&1 | grep -E " sc | c3 .*(64->256|128->512|256->1024|512->2048)|\.1 c1|TOTAL|stem"
