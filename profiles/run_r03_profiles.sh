#!/bin/bash
# rocprofv3 passes behind profiles/r03_*: kernel statistics and the two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, no
# tracing domains mixed in) of the headline bench command, plus kernel statistics (and the PMC passes) of the 256^3 micro-benchmark and kernel statistics of the cylinder and airfoil legs.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
BENCH="python3 $R/bench.py --no-cpu-baseline --no-micro --steps 10 --warmup 3"
if [ "$1" = "mb" ]; then   # only the kernel statistics of the cylinder and airfoil legs
  rocprofv3 --kernel-trace --stats -d $O/p_cyl -o cyl -- python3 $R/profiles/cylinder_modes.py 64 2 1-0-1 > $O/p_cyl.log 2>&1
  python3 $R/profiles/summarize_rocpd.py "$(find $O/p_cyl -name '*.db' | head -1)" $O/r03_c_cylinder_kernel_stats.csv > /dev/null
  rocprofv3 --kernel-trace --stats -d $O/p_air -o air -- python3 $R/profiles/airfoil_bench.py 16 1 40 > $O/p_air.log 2>&1
  python3 $R/profiles/summarize_rocpd.py "$(find $O/p_air -name '*.db' | head -1)" $O/r03_e_airfoil_trial_kernel_stats.csv > /dev/null   # (default policy = with the trial: the r03_e file)
  rm -rf $O/p_cyl $O/p_air; tail -n 2 $O/p_cyl.log; tail -n 2 $O/p_air.log; ls -la $O/r03_c_* $O/r03_e_*; exit 0
fi
rocprofv3 --kernel-trace --stats -d $O/p_stats -o bench -- $BENCH > $O/p_stats.log 2>&1
python3 $R/profiles/summarize_rocpd.py "$(find $O/p_stats -name '*.db' | head -1)" $O/r03_a_bench_kernel_stats.csv > /dev/null
rocprofv3 --pmc FETCH_SIZE -d $O/p_fetch -o bench -- $BENCH > $O/p_fetch.log 2>&1
python3 $R/profiles/summarize_pmc.py "$(find $O/p_fetch -name '*.db' | head -1)" > $O/r03_a_bench_pmc_fetch.csv
rocprofv3 --pmc WRITE_SIZE -d $O/p_write -o bench -- $BENCH > $O/p_write.log 2>&1
python3 $R/profiles/summarize_pmc.py "$(find $O/p_write -name '*.db' | head -1)" > $O/r03_a_bench_pmc_write.csv
if [ "$1" = "headline" ]; then rm -rf $O/p_stats $O/p_fetch $O/p_write; ls -la $O/r03_*; exit 0; fi   # only the headline passes
rocprofv3 --kernel-trace --stats -d $O/p_p256 -o p256 -- python3 $R/profiles/micro_poisson.py > $O/p_p256.log 2>&1
python3 $R/profiles/summarize_rocpd.py "$(find $O/p_p256 -name '*.db' | head -1)" $O/r03_b_poisson256_kernel_stats.csv > /dev/null
rocprofv3 --kernel-trace --stats -d $O/p_cyl -o cyl -- python3 $R/profiles/cylinder_modes.py 64 2 1-0-1 > $O/p_cyl.log 2>&1
python3 $R/profiles/summarize_rocpd.py "$(find $O/p_cyl -name '*.db' | head -1)" $O/r03_c_cylinder_kernel_stats.csv > /dev/null
rocprofv3 --pmc FETCH_SIZE -d $O/p_p256f -o p256 -- python3 $R/profiles/micro_poisson.py > $O/p_p256f.log 2>&1
python3 $R/profiles/summarize_pmc.py "$(find $O/p_p256f -name '*.db' | head -1)" > $O/r03_b_poisson256_pmc_fetch.csv
rocprofv3 --pmc WRITE_SIZE -d $O/p_p256w -o p256 -- python3 $R/profiles/micro_poisson.py > $O/p_p256w.log 2>&1
python3 $R/profiles/summarize_pmc.py "$(find $O/p_p256w -name '*.db' | head -1)" > $O/r03_b_poisson256_pmc_write.csv
export FLUIDGYM_AMD_PRESSURE_MULTILEVEL_BICGSTAB=0
rocprofv3 --kernel-trace --stats -d $O/p_air -o air -- python3 $R/profiles/airfoil_bench.py 16 1 40 > $O/p_air.log 2>&1
python3 $R/profiles/summarize_rocpd.py "$(find $O/p_air -name '*.db' | head -1)" $O/r03_d_airfoil_kernel_stats.csv > /dev/null
export FLUIDGYM_AMD_PRESSURE_MULTILEVEL_BICGSTAB=1
# (r03_d: the plain refined recurrence, policy off; r03_e: the default since round 3 = with the multilevel trial)
rocprofv3 --kernel-trace --stats -d $O/p_air2 -o air -- python3 $R/profiles/airfoil_bench.py 16 1 40 > $O/p_air2.log 2>&1
unset FLUIDGYM_AMD_PRESSURE_MULTILEVEL_BICGSTAB
python3 $R/profiles/summarize_rocpd.py "$(find $O/p_air2 -name '*.db' | head -1)" $O/r03_e_airfoil_trial_kernel_stats.csv > /dev/null
rm -rf $O/p_stats $O/p_fetch $O/p_write $O/p_p256 $O/p_cyl $O/p_p256f $O/p_p256w $O/p_air $O/p_air2
ls -la $O/r03_*
