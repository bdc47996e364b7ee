#!/bin/bash
# The reference's largest DNA fixture (125.phy, 247 candidate roots), whole program twice: candidates one after the
# other (--lockstep 0) and 32 in lock step (the default) -- the checkpoint records must be the same, bit for bit.
# usage: lockstep_vs_sequential_125.sh <liblbfgsb.so>
cd /root/repo
LB=${1:?path to a library exporting setulb}
python3 -c "import lzma; open('/tmp/125.phy','w').write(lzma.open('tests/golden/data/125.phy.xz','rt').read())"
for mode in 0 32; do
  rm -f /tmp/ls$mode.*
  echo "== --lockstep $mode"
  root_digger_amd/bin/rd_amd --msa /tmp/125.phy --tree tests/golden/data/125.tree --prefix /tmp/ls$mode --exhaustive \
    --rate-cats 4 --lbfgsb $LB --lockstep $mode 2>&1 | grep -v "^\[" | grep -v "^(" | tail -4
done
python3 - <<PY
import sys
sys.path.insert(0, ".")
import root_digger_amd as rd
a = sorted(rd.Checkpoint("/tmp/ls0").read_results())
b = sorted(rd.Checkpoint("/tmp/ls32").read_results())
same = len(a) == len(b) and all(x == y for x, y in zip(a, b))
print(len(a), "records sequential,", len(b), "in lock step; identical (root id, lnL, alpha) bit for bit:", same)
if not same:
    bad = [(x, y) for x, y in zip(a, b) if x != y]
    print(len(bad), "differ, first:", bad[:2])
    sys.exit(1)
PY
