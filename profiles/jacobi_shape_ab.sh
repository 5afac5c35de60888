run() { python bench.py --no-cpu-baseline --no-micro --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['roofline']['avg_launch_us'], d['config']['iters_per_solve[mean,max]']['velocity'])"; }
for i in 1 2; do
FG_JAC_SHAPE=1 run rows_S6
FG_JAC_SHAPE=2 run bands_S8
FG_JAC_SHAPE=2 FG_JAC_SWEEPS=6 run bands_S6
FG_JAC_SHAPE=2 FG_JAC_SWEEPS=12 run bands_S12
FG_JAC_SHAPE=1 FG_JAC_SWEEPS=4 run rows_S4
done
for sh in 1 2; do for sw in 4 6 8; do FG_JAC_SHAPE=$sh FG_JAC_SWEEPS=$sw python profiles/leg_run.py ChannelJet2D-large-v0 64 3 1 2.0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('large shape $sh sweeps $sw', d['env_steps_per_s'], d['iters[mean,max]']['velocity'])"; done; done
