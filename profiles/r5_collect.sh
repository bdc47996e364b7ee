#!/bin/bash
# Round-5 evidence on ONE box (gpurun -- 'bash profiles/r5_collect.sh'):
#  1. collect.sh r5 + summarize.py r5: rocprofv3 --kernel-trace --stats and the separate PMC passes of the default
#     bench command -> profiles/r5_kernel_stats.csv, r5_summary.json (digest-bound to the kernel sources);
#  2. the bench line of every BASELINE config, the reference's 125.phy and the shapes one GPU sees under BASELINE's
#     sharding -> profiles/r5_<config>_bench.json;
#  3. bench --shard sites at c2 / 8's shape, pipelined (the stream-ordered device batch) with and without a one-rank
#     communicator behind every batch;
#  4. the whole program (rd_amd, exhaustive, the reference's L-BFGS-B) on c2 and 125.phy.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5; mkdir -p $O
LB=$GRAFT_REPO_ROOT/oracle/_ref/liblbfgsb_ref.so
bash profiles/collect.sh r5 > $O/collect.log 2>&1
python3 profiles/summarize.py r5 > $O/summarize.log 2>&1
B="python3 bench.py"
$B > $O/c2_bench.json 2> $O/c2.err
$B --config c3 --steps 5 --warmup 1 > $O/c3_bench.json 2> $O/c3.err
$B --config c4 --steps 4 --warmup 1 --no-cpu-baseline > $O/c4_bench.json 2> $O/c4.err
$B --config c5 --steps 4 --warmup 1 --no-cpu-baseline > $O/c5_bench.json 2> $O/c5.err
$B --config d125 --cpu-seconds 6 > $O/d125_bench.json 2> $O/d125.err
$B --config c4 --sites 62500 --steps 4 --warmup 1 --no-cpu-baseline > $O/c4_shard_bench.json 2> $O/c4_shard.err
$B --config c5 --sites 50000 --steps 4 --warmup 1 --no-cpu-baseline > $O/c5_shard_bench.json 2> $O/c5_shard.err
$B --shard sites --sites 6250 --steps 40 --warmup 5 --no-cpu-baseline > $O/c2_8_sites_pipelined.json 2> $O/c2_8_sites.err
$B --shard sites --sites 6250 --steps 40 --warmup 5 --no-cpu-baseline --one-rank-comm > $O/c2_8_sites_pipelined_comm.json 2> $O/c2_8_sites_comm.err
# the per-rank shapes of the default N = 2 / 4 / 8 line's site_sharded leg (c2's sites / N), one rank, the collective's
# launch included: what profiles/r5_scale_prediction.md extrapolates from
for sites in 25000 12500 6250; do $B --shard sites --sites $sites --steps 40 --warmup 5 --no-cpu-baseline --one-rank-comm 2> /dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('c2 sites/%d = %5d per rank, pipelined + one-rank all-reduce: %9.1f evals/s  %.4f ms per 197-job step  evaluator kernel %.4f ms' % (50000 // $sites, $sites, d['value'], d['ms_per_step'], r['avg_launch_ms']))"; done | tee $O/per_rank_shapes.txt
for c in c2 c3 c4 c5 d125 c4_shard c5_shard; do python3 -c "
import json
d=json.load(open('$O/${c}_bench.json')); r=d['roofline']; k=d.get('clv_kernel',{})
print('%-9s %10.1f evals/s  kernel %9.3f ms  frac %.4f  clv_kernel frac %s  executed site-CLV/s %.3e' % ('$c', d['value'], r['avg_launch_ms'], r['frac'], k.get('frac'), d['site_clv_updates_per_sec_executed']))"; done | tee $O/bench_lines.txt
bash profiles/full_run_c2.sh $LB > $O/full_run_c2.txt 2>&1
bash profiles/full_run_125.sh $LB > $O/full_run_125.txt 2>&1
tail -3 $O/full_run_c2.txt; tail -3 $O/full_run_125.txt
cp profiles/r5_summary.json profiles/r5_kernel_stats.csv $O/ 2>/dev/null
ls $O | head -40
