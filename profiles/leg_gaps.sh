#!/bin/bash
# GPU busy fraction and the idle time behind each kernel for one bench leg (one lane): bash profiles/leg_gaps.sh NAME <leg_run.py args...>
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
NAME=$1; shift
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/lg_$NAME -o t -- python3 $R/profiles/leg_run.py "$@" > $O/lg_$NAME.log 2>&1
DB=$(find $O/lg_$NAME -name '*.db' | head -1)
echo "== $NAME: $(grep '^{' $O/lg_$NAME.log | tail -1 | cut -c1-200)"
python3 $R/profiles/gaps_rocpd.py $DB 0.6 3 | head -12
python3 $R/profiles/summarize_rocpd.py $DB $O/leg_${NAME}_kernel_stats.csv > /dev/null; head -14 $O/leg_${NAME}_kernel_stats.csv | cut -c1-110
rm -rf $O/lg_$NAME
