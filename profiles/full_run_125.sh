#!/bin/bash
# The whole program on the reference's largest DNA fixture (test/data/dna/125.phy +
# tree/125.tree, committed xz-compressed under tests/golden/data): exhaustive mode with
# parameter optimisation through the native front end.  usage: full_run_125.sh <liblbfgsb.so>
cd /root/repo
LB=${1:?path to a library exporting setulb}
python3 -c "import lzma; open('/tmp/125.phy','w').write(lzma.open('tests/golden/data/125.phy.xz','rt').read())"
rm -f /tmp/r125.*
time (root_digger_amd/bin/rd_amd --msa /tmp/125.phy --tree tests/golden/data/125.tree --prefix /tmp/r125 \
  --exhaustive --rate-cats 4 --lbfgsb $LB | grep -v "^\[" | cut -c1-160 | tail -4)
python3 - <<PY
import sys
sys.path.insert(0, ".")
import root_digger_amd as rd
r = rd.Checkpoint("/tmp/r125").read_results()
print(len(r), "candidates in the log; best", max(r, key=lambda x: x[1])[:3])
PY
