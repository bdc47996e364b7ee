#!/bin/bash
# rd_amd on WORLD ranks of ONE device, site groups of G (host reducer), LOCKSTEP candidates in
# flight per rank: usage  multirank_lockstep.sh WORLD G LOCKSTEP [MSA TREE [extra rd_amd options]]
# Every rank's stdout / stderr goes to gpurun_out/mr_<world>_<G>_<lockstep>.rank<r>.log; the
# whole run is limited to $LIMIT seconds (default 150).
WORLD=$1; G=$2; LS=$3; MSA=${4:-tests/golden/data/10.fasta}; TREE=${5:-tests/golden/data/10.tree}
shift 5 2>/dev/null || shift $#
LIMIT=${LIMIT:-150}
REF=oracle/_ref/liblbfgsb_ref.so
mkdir -p gpurun_out
TAG=gpurun_out/mr_${WORLD}_${G}_${LS}
PREFIX=$(mktemp -d)/run
export WORLD_SIZE=$WORLD MASTER_ADDR=127.0.0.1 MASTER_PORT=$((20000 + RANDOM % 20000))
pids=()
for r in $(seq 0 $((WORLD - 1))); do
  RANK=$r LOCAL_RANK=$r timeout $LIMIT root_digger_amd/bin/rd_amd --msa $MSA --tree $TREE --exhaustive --silent \
    --rate-cats 4 --atol 1e-3 --brtol 1e-3 --bfgstol 1e-3 --factor 1e12 --seed 5 --lbfgsb $REF --device 0 \
    --site-shards $G --site-reduce host --stats --prefix $PREFIX --lockstep $LS "$@" > $TAG.rank$r.log 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=$?; done
echo "world $WORLD G $G lockstep $LS: rc $rc"
grep -h "stats:" $TAG.rank*.log | cut -c1-400 | head -$WORLD
grep -h "still waiting" $TAG.rank*.log | tail -4
exit $rc
