#!/bin/bash
# 256^3 z-march kernel sweep on the GPU box: register budget (FG_ZMARCH_OCC build switch) x planes per barrier pair x planes per
# z-chunk.  One JSON line of bench.poisson_micro per combination in gpurun_out/zmarch_sweep.log.
cd "$(dirname "$0")/.."
out=gpurun_out/zmarch_sweep.log
: > $out
for occ in 1 0; do
  ( cd fluidgym_amd/csrc && rm -f fg_poisson3d.o && make CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast -DFG_ZMARCH_OCC=$occ" >/dev/null 2>&1 )
  for ppb in 1 2; do
    for zc in 8 16 32; do
      echo "occ=$occ ppb=$ppb zc=$zc" >> $out
      FG_ZMARCH_PPB=$ppb FG_FORCE_ZMARCH=$zc python profiles/micro_poisson.py 256 >> $out 2>&1
    done
  done
done
( cd fluidgym_amd/csrc && rm -f fg_poisson3d.o && make >/dev/null 2>&1 )
