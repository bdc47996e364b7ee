#!/bin/bash
# Where the device time of the c2 exhaustive search goes, by kernel
# (tests/tools/e2e_search.py, 16 candidates in lock step).
# Usage: gpurun -- 'bash profiles/e2e_prof.sh'
O=gpurun_out/e2e_prof; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export LOCKSTEP=16
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o e2e -- python3 $GRAFT_REPO_ROOT/tests/tools/e2e_search.py 12 > $GRAFT_REPO_ROOT/$O/run.txt 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/trace -name '*kernel_stats.csv' | head -1)
cp "$f" $O/kernel_stats.csv
find $O/trace -name '*kernel_trace.csv' -delete
tail -2 $O/run.txt; head -15 $O/kernel_stats.csv
