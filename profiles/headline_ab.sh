#!/bin/bash
# Same-box A/B of the headline under an environment switch: bash profiles/headline_ab.sh VAR A B [extra bench flags]
# (alternating runs: boxes and runs differ by a few per cent)
VAR=$1; A=$2; B=$3; shift 3
F="--steps 20 --warmup 5 --no-cpu-baseline --no-micro --no-airfoil-leg --leg-budget 0 $*"
for V in $A $B $A $B; do
  env $VAR=$V python bench.py $F 2>/tmp/err_ab.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$VAR=$V', 'value', round(d['value'],1), 'spread', round(d['value_spread']['min']), round(d['value_spread']['max']), 'ms', round(d['ms_per_step'],3), 'lanes', d['config'].get('lanes_per_gpu'))" || tail -5 /tmp/err_ab.txt
done
