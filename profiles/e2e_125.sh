#!/bin/bash
# Kernel trace of the whole program on 125.phy (lock step 32) + profiles/e2e_overlap.py.
# Usage: gpurun -- 'bash profiles/e2e_125.sh <liblbfgsb.so>'
LB=${1:?path to a library exporting setulb}
O=$GRAFT_REPO_ROOT/gpurun_out/e2e_125; rm -rf $O; mkdir -p $O
python3 -c "import lzma; open('/tmp/125.phy','w').write(lzma.open('$GRAFT_REPO_ROOT/tests/golden/data/125.phy.xz','rt').read())"
cd /tmp && export TMPDIR=/tmp
rm -f /tmp/t125.*
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o e2e -- $GRAFT_REPO_ROOT/root_digger_amd/bin/rd_amd --msa /tmp/125.phy \
  --tree $GRAFT_REPO_ROOT/tests/golden/data/125.tree --prefix /tmp/t125 --exhaustive --rate-cats 4 --lbfgsb $LB > $O/run.txt 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/trace -name '*kernel_stats.csv' | head -1); cp "$f" $O/kernel_stats.csv
t=$(find $O/trace -name '*kernel_trace.csv' | head -1)
python3 profiles/e2e_overlap.py "$t" > $O/overlap.txt 2>&1
rm -rf $O/trace
grep -E "Inference took|Final" $O/run.txt; cat $O/overlap.txt; head -8 $O/kernel_stats.csv | cut -c1-200
