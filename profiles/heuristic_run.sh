#!/bin/bash
# The reference's DEFAULT mode (heuristic root search, src/main.cpp:586-635 / src/model.cpp:1008-1137) through the
# native front end on the c2 shape and on 125.phy.  usage: heuristic_run.sh <liblbfgsb.so>
cd /root/repo
LB=${1:?path to a library exporting setulb}
python3 - <<PY
import sys, lzma
sys.path.insert(0, ".")
from root_digger_amd import synth
w = synth.workload(100, 50000, 4, 4, 0xD166E5 + 1)
open("/tmp/c2.nwk", "w").write(w["newick"])
open("/tmp/c2.fasta", "w").write("".join(">%s\n%s\n" % kv for kv in w["seqs"].items()))
open('/tmp/125.phy','w').write(lzma.open('tests/golden/data/125.phy.xz','rt').read())
PY
for mr in 1 8; do
rm -f /tmp/h2.*
echo "== c2, --min-roots $mr"
( time (root_digger_amd/bin/rd_amd --msa /tmp/c2.fasta --tree /tmp/c2.nwk --prefix /tmp/h2 --rate-cats 4 --lbfgsb $LB --min-roots $mr | grep -v "^\[" | cut -c1-120 | tail -3) ) 2>&1 | grep -v "^$\|^user\|^sys"
done
rm -f /tmp/h125.*
echo "== 125.phy"
( time (root_digger_amd/bin/rd_amd --msa /tmp/125.phy --tree tests/golden/data/125.tree --prefix /tmp/h125 --rate-cats 4 --lbfgsb $LB | grep -v "^\[" | cut -c1-120 | tail -3) ) 2>&1 | grep -v "^$\|^user\|^sys"
