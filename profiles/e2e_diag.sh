#!/bin/bash
# Kernel trace of the lock-stepped c2 search + what overlaps what (profiles/e2e_overlap.py).
# Usage: gpurun -- 'bash profiles/e2e_diag.sh [tag] [candidates]'; extra environment is passed through.
TAG=${1:-diag}; NC=${2:-12}
O=$GRAFT_REPO_ROOT/gpurun_out/e2e_$TAG; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export LOCKSTEP=${LOCKSTEP:-16}
python3 $GRAFT_REPO_ROOT/tests/tools/e2e_search.py $NC > $O/plain.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o e2e -- python3 $GRAFT_REPO_ROOT/tests/tools/e2e_search.py $NC > $O/run.txt 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/trace -name '*kernel_stats.csv' | head -1); cp "$f" $O/kernel_stats.csv
t=$(find $O/trace -name '*kernel_trace.csv' | head -1)
python3 profiles/e2e_overlap.py "$t" > $O/overlap.txt 2>&1
rm -rf $O/trace
echo "== un-profiled"; tail -2 $O/plain.txt; echo "== profiled"; tail -2 $O/run.txt; cat $O/overlap.txt; head -8 $O/kernel_stats.csv | cut -c1-200
