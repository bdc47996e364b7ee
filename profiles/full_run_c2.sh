#!/bin/bash
# The whole program on the c2 shape: synthetic 100 x 50 000 alignment written to
# FASTA + newick, then the native front end in exhaustive mode with parameter
# optimisation (the caller's L-BFGS-B = the reference's lib/lbfgsb build that
# tests/ keeps under oracle/_ref).  usage: full_run_c2.sh <liblbfgsb.so> [taxa sites]
cd /root/repo
LB=${1:?path to a library exporting setulb}
N=${2:-100}; S=${3:-50000}
python - <<PY
import sys
sys.path.insert(0, ".")
from root_digger_amd import synth
w = synth.workload($N, $S, 4, 4, 0xD166E5 + 1)
open("/tmp/c2.nwk", "w").write(w["newick"])
open("/tmp/c2.fasta", "w").write("".join(">%s\n%s\n" % kv for kv in w["seqs"].items()))
PY
rm -f /tmp/c2run.*
time (root_digger_amd/bin/rd_amd --msa /tmp/c2.fasta --tree /tmp/c2.nwk --prefix /tmp/c2run \
  --exhaustive --rate-cats 4 --lbfgsb $LB | grep -v "^\[" | cut -c1-160 | tail -4)
python - <<PY
import sys
sys.path.insert(0, ".")
import root_digger_amd as rd
r = rd.Checkpoint("/tmp/c2run").read_results()
print(len(r), "candidates in the log; best", max(r, key=lambda x: x[1])[:3])
PY
