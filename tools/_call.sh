B="python bench.py --only-main --no-cpu-baseline --no-pcie --steps 10"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['resnet_pass_ms'])" "$1"; }
$B | pick default
HMMR_TUNE_TILES=5,6,3,1,2,7 $B | pick tune+7
HMMR_TUNE_TILES=5,6,3,1,2,7,8,9 $B | pick tune+789
HMMR_RESNET_STREAMS=3 $B | pick streams3
HMMR_RESNET_STREAMS=4 $B | pick streams4
HMMR_RESNET_STREAMS=1 $B | pick streams1
$B --dtype bf16 | pick bf16
HMMR_TUNE_TILES=5,6,3,1,2,7 $B --dtype bf16 | pick bf16+7
