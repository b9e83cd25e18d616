python bench.py --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
print(json.dumps(d['modes'], indent=0))
print({k:v for k,v in d.items() if 'pcie' in k}); print(d['cpu_baseline']['value'], d['value_meets_tolerance'])"
