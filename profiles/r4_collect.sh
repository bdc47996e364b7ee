#!/bin/bash
# Round-4 bench lines on one box: every BASELINE config, the reference's 125.phy, and the shapes one
# GPU sees under BASELINE's sharding.  (collect.sh r4 + summarize.py r4 give r4_kernel_stats.csv /
# r4_summary.json; e2e_diag.sh, full_run_*.sh, rescale_ab.sh, r4_shard.sh the rest.)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4b; mkdir -p $O
B="python3 bench.py"
$B > $O/c2.json 2> $O/c2.err
$B --config c3 --steps 5 --warmup 1 > $O/c3.json 2> $O/c3.err
$B --config c4 --steps 4 --warmup 1 --no-cpu-baseline > $O/c4.json 2> $O/c4.err
$B --config c5 --steps 4 --warmup 1 --no-cpu-baseline > $O/c5.json 2> $O/c5.err
$B --config d125 --cpu-seconds 6 > $O/d125.json 2> $O/d125.err
$B --config c4 --sites 62500 --steps 4 --warmup 1 --no-cpu-baseline > $O/c4_shard.json 2> $O/c4_shard.err
$B --config c5 --sites 50000 --steps 4 --warmup 1 --no-cpu-baseline > $O/c5_shard.json 2> $O/c5_shard.err
for c in c2 c3 c4 c5 d125 c4_shard c5_shard; do python3 -c "
import json
d=json.load(open('$O/$c.json')); r=d['roofline']; k=d.get('clv_kernel',{})
print('%-9s %10.1f evals/s  kernel %9.3f ms  frac %.4f  clv_kernel frac %s  executed site-CLV/s %.3e' % ('$c', d['value'], r['avg_launch_ms'], r['frac'], k.get('frac'), d['site_clv_updates_per_sec_executed']))"; done
