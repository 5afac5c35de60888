#!/bin/bash
# profiles/bicg_stress.py under every access mode of the recurrence words (build switches FG_ACC_ACCESS / FG_FLAG_ACCESS of
# csrc/fg_internal.h: bit 0 atomic loads, bit 1 atomic stores).  Usage: bash profiles/bicg_stress.sh [seconds per dump] [modes...]
cd "$(dirname "$0")/.."
secs=${1:-60}; shift
modes=${@:-"0:0 3:3 0:3 3:0 2:0"}
out=gpurun_out/bicg_stress.log
: > $out
base="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast"
for m in $modes; do
  acc=${m%%:*}; flg=${m##*:}
  ( cd fluidgym_amd/csrc && rm -f *.o && make -j8 CXXFLAGS="$base -DFG_ACC_ACCESS=$acc -DFG_FLAG_ACCESS=$flg" >/dev/null 2>&1 )
  echo "acc_access=$acc flag_access=$flg" >> $out
  FG_MB_TRACE_FAIL=1 python profiles/bicg_stress.py $secs >> $out 2>&1
done
( cd fluidgym_amd/csrc && rm -f *.o && make -j8 >/dev/null 2>&1 )
