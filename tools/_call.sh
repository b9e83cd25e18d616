python -m pytest tests/test_gpu_bf16x3.py -m gpu -x -q -k "fused_tails" 2>&1 | tail -2
python tools/layer_table.py 257 bf16x3 3 2>&1 | grep -E "^(1|2)\.. c3|TOTAL"
B="python bench.py --only-main --no-cpu-baseline --no-pcie --steps 10"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['resnet_pass_ms'], d['roofline']['frac'])" "$1"; }
$B | pick default
HMMR_FUSE_TAIL=block1 $B | pick b1only
