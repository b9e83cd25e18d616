B="python bench.py --only-main --no-cpu-baseline --no-pcie --steps 20"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['ms_per_step'])" "$1"; }
$B | pick default
HMMR_RESNET_NOJOIN=1 $B | pick nojoin
for o in 2 5 10 20; do HMMR_RESNET_NOJOIN=1 HMMR_RESNET_OFFSET_MCYC=$o $B | pick nojoin_off$o; done
