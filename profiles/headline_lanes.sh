#!/bin/bash
# The headline (64 envs on one GPU) as 1, 2 and 4 lanes (ParallelFluidEnv(lanes=L): sub-batches stepped concurrently on their own HIP
# streams), same box, same command otherwise.  bash profiles/headline_lanes.sh
F="--steps 20 --warmup 5 --no-cpu-baseline --no-micro --no-airfoil-leg --leg-budget 0"
for L in 1 2 4 2 1; do
  python bench.py $F --lanes $L 2>/tmp/err_$L.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('lanes', $L, 'value', d['value'], 'spread', d['value_spread']['min'], d['value_spread']['max'], 'ms', d['ms_per_step'], 'roof', r['kernel'][:18], r['frac'], r.get('avg_launch_us'))" || tail -5 /tmp/err_$L.txt
done
