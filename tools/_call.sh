B="python bench.py --only-main --no-cpu-baseline --no-pcie --steps 10"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['resnet_pass_ms'])" "$1"; }
$B | pick default
HMMR_FUSE_SC=all $B | pick fuse_sc_all
HMMR_FUSE_SC=0 $B | pick fuse_sc_0
HMMR_TAIL_PRIORITY=-1 $B | pick tailprio
HMMR_RESNET_PRIORITY=0 $B | pick resprio0
