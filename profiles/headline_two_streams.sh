#!/bin/bash
# Is the headline step's host-bound idle time (GPU busy 78 %) recoverable by overlapping two half batches?  Two PROCESSES of 32 envs
# on the one GPU at the same time, against one of 32 and one of 64.  bash profiles/headline_two_streams.sh
F="--steps 20 --warmup 5 --no-cpu-baseline --no-micro --no-airfoil-leg --leg-budget 0"
val() { python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['ms_per_step'])" $1; }
python bench.py $F --envs-per-gpu 64 > /tmp/b64.json 2>/dev/null; val /tmp/b64.json
python bench.py $F --envs-per-gpu 32 > /tmp/b32.json 2>/dev/null; val /tmp/b32.json
python bench.py $F --envs-per-gpu 32 > /tmp/b32a.json 2>/dev/null &
P1=$!
python bench.py $F --envs-per-gpu 32 > /tmp/b32b.json 2>/dev/null &
P2=$!
wait $P1 $P2
val /tmp/b32a.json; val /tmp/b32b.json
