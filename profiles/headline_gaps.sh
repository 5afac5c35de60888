#!/bin/bash
# Kernel trace of the headline loop with one and two lanes: per-kernel statistics, GPU busy fraction and the idle time behind each kernel
# (profiles/gaps_rocpd.py), concurrency of the two lanes' streams (profiles/overlap_rocpd.py).  bash profiles/headline_gaps.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for L in 1 2; do
  rocprofv3 --kernel-trace --stats -d $O/hg_$L -o t -- python3 $R/bench.py --no-cpu-baseline --no-micro --steps 20 --warmup 5 --repeats 3 --lanes $L > $O/hg_$L.log 2>&1
  DB=$(find $O/hg_$L -name '*.db' | head -1)
  python3 $R/profiles/summarize_rocpd.py $DB $O/r06_headline_lanes${L}_kernel_stats.csv > /dev/null
  echo "== lanes $L"; python3 $R/profiles/gaps_rocpd.py $DB 0.5 3 | head -12
  python3 $R/profiles/overlap_rocpd.py $DB | head -3
  head -18 $O/r06_headline_lanes${L}_kernel_stats.csv | cut -c1-120
  rm -rf $O/hg_$L
done
