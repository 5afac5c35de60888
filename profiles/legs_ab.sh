#!/bin/bash
# Same-box A/B of every bench leg under an environment switch: bash profiles/legs_ab.sh VAR A B
VAR=$1; A=$2; B=$3
for V in $A $B; do
  env $VAR=$V python bench.py --steps 20 --warmup 5 --no-cpu-baseline --leg-budget 200 2>/tmp/err_legs.txt > /tmp/legs_$V.json || tail -5 /tmp/err_legs.txt
  python - <<PY
import json
d=json.load(open('profiles/bench_detail.json')) if False else json.loads(open('/tmp/legs_$V.json').read().strip().splitlines()[-1])
print('$VAR=$V', 'headline', round(d['value']), {k: (round(v['value'],1) if isinstance(v, dict) and 'value' in v else v) for k, v in d.get('legs', {}).items()})
PY
done
