#!/bin/bash
# rocprofv3 passes behind profiles/r06_*: per-leg kernel statistics (every bench leg, VERDICT r4 "missing 2": PMC for every leg) and the FETCH_SIZE /
# WRITE_SIZE PMC passes (separate runs, no tracing domains mixed in) of the headline command, the 256^3 micro-benchmark and the TCF leg.
#   bash profiles/run_r06_profiles.sh [all | headline | legs | pmc | rbc | refresh | cyl]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
W=${1:-all}
stats() {   # name, command...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats -d $O/p_$name -o t -- "$@" > $O/p_$name.log 2>&1
  python3 $R/profiles/summarize_rocpd.py "$(find $O/p_$name -name '*.db' | head -1)" $O/r06_${name}_kernel_stats.csv > /dev/null
  rm -rf $O/p_$name; tail -n 1 $O/p_$name.log | cut -c1-400
}
pmc() {     # name, counter, command...
  local name=$1 ctr=$2; shift 2
  rocprofv3 --pmc $ctr -d $O/q_$name -o t -- "$@" > $O/q_$name.log 2>&1
  python3 $R/profiles/summarize_pmc.py "$(find $O/q_$name -name '*.db' | head -1)" > $O/r06_${name}_pmc_$(echo $ctr | tr 'A-Z' 'a-z' | sed 's/_size//').csv
  rm -rf $O/q_$name
}
BENCH="python3 $R/bench.py --no-cpu-baseline --no-micro --steps 10 --warmup 3"
TCF="python3 $R/profiles/leg_run.py TCF3D-baseline-v0 8 5 1"
if [ $W = all ] || [ $W = headline ]; then
  stats a_bench $BENCH
  pmc a_bench FETCH_SIZE $BENCH
  pmc a_bench WRITE_SIZE $BENCH
fi
if [ $W = all ] || [ $W = legs ]; then
  stats b_poisson256 python3 $R/profiles/micro_poisson.py
  stats c_cylinder python3 $R/profiles/leg_run.py CylinderJet2D-easy-v0 64 4 1 0 initial_domain_steps=100 randomize_initial_state=false
  stats d_cylinder_medium python3 $R/profiles/leg_run.py CylinderJet2D-medium-v0 64 3 1 0 initial_domain_steps=100 randomize_initial_state=false
  stats e_airfoil64 python3 $R/profiles/airfoil_bench.py 64 2 40
  stats f_tcf $TCF
  stats g_rbc python3 $R/profiles/leg_run.py RBC2D-baseline-v0 32 5 1
  stats h_large python3 $R/profiles/leg_run.py ChannelJet2D-large-v0 64 4 1 2.0
fi
RBC="python3 $R/profiles/leg_run.py RBC2D-baseline-v0 32 5 1"
LARGE="python3 $R/profiles/leg_run.py ChannelJet2D-large-v0 64 4 1 2.0"
CYL="python3 $R/profiles/leg_run.py CylinderJet2D-easy-v0 64 4 1 0 initial_domain_steps=100 randomize_initial_state=false"
AIR="python3 $R/profiles/airfoil_bench.py 64 1 40"
if [ $W = cyl ]; then
  stats c_cylinder $CYL
  stats d_cylinder_medium python3 $R/profiles/leg_run.py CylinderJet2D-medium-v0 64 3 1 0 initial_domain_steps=100 randomize_initial_state=false
fi
if [ $W = rbc ]; then
  stats g_rbc $RBC
fi
if [ $W = refresh ]; then      # the legs whose kernels changed late in the round
  stats a_bench $BENCH
  pmc a_bench FETCH_SIZE $BENCH
  pmc a_bench WRITE_SIZE $BENCH
  stats g_rbc $RBC
  pmc g_rbc FETCH_SIZE $RBC
  pmc g_rbc WRITE_SIZE $RBC
  stats h_large $LARGE
  pmc h_large FETCH_SIZE $LARGE
  pmc h_large WRITE_SIZE $LARGE
  stats f_tcf $TCF
fi
if [ $W = all ] || [ $W = pmc ]; then
  pmc b_poisson256 FETCH_SIZE python3 $R/profiles/micro_poisson.py
  pmc b_poisson256 WRITE_SIZE python3 $R/profiles/micro_poisson.py
  pmc f_tcf FETCH_SIZE $TCF
  pmc f_tcf WRITE_SIZE $TCF
  pmc g_rbc FETCH_SIZE $RBC
  pmc g_rbc WRITE_SIZE $RBC
  pmc h_large FETCH_SIZE $LARGE
  pmc h_large WRITE_SIZE $LARGE
  pmc c_cylinder FETCH_SIZE $CYL
  pmc c_cylinder WRITE_SIZE $CYL
  pmc d_cylinder_medium FETCH_SIZE python3 $R/profiles/leg_run.py CylinderJet2D-medium-v0 64 3 1 0 initial_domain_steps=100 randomize_initial_state=false
  pmc d_cylinder_medium WRITE_SIZE python3 $R/profiles/leg_run.py CylinderJet2D-medium-v0 64 3 1 0 initial_domain_steps=100 randomize_initial_state=false
  pmc e_airfoil64 FETCH_SIZE $AIR
  pmc e_airfoil64 WRITE_SIZE $AIR
fi
ls -la $O/r06_* | head -40
