#!/bin/bash
# Round-6 bench lines on ONE box, AFTER profiles/summarize.py r6 has been committed (the lines carry the
# counters of the commands they belong to): every BASELINE config, 125.phy, the shard shapes.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6; mkdir -p $O
B="python3 bench.py"
$B > $O/c2_bench.json 2> $O/c2.err
$B --config c3 --steps 5 --warmup 1 > $O/c3_bench.json 2> $O/c3.err
$B --config c4 --steps 4 --warmup 1 --no-cpu-baseline > $O/c4_bench.json 2> $O/c4.err
$B --config c5 --steps 4 --warmup 1 --no-cpu-baseline > $O/c5_bench.json 2> $O/c5.err
$B --config d125 --cpu-seconds 6 > $O/d125_bench.json 2> $O/d125.err
$B --config c4 --sites 62500 --steps 4 --warmup 1 --no-cpu-baseline > $O/c4_shard_bench.json 2> $O/c4_shard.err
$B --config c5 --sites 50000 --steps 4 --warmup 1 --no-cpu-baseline > $O/c5_shard_bench.json 2> $O/c5_shard.err
$B --shard sites --sites 6250 --steps 40 --warmup 5 --no-cpu-baseline > $O/c2_8_sites_pipelined_bench.json 2> $O/c2_8_sites.err
for c in c2 c3 c4 c5 d125 c4_shard c5_shard; do python3 -c "
import json,sys
d=json.load(open('$O/${c}_bench.json')); r=d['roofline']
print('%-9s %10.1f evals/s  kernel %8.4f ms  frac %.4f' % ('$c', d['value'], r['avg_launch_ms'], r['frac']))"; done | tee $O/summary.txt
