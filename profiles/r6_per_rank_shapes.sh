#!/bin/bash
# ON THE GPU BOX: what ONE rank of an N-GPU site group does per step on c2 (sites / N, 197 jobs, two batches in flight, the
# library's one-rank communicator queued behind every batch) under both sum modes, and the bare one-rank collective's
# cost in each -- the inputs of profiles/r6_scale_prediction.md
cd $GRAFT_REPO_ROOT
for mode in gather allreduce; do
  for sites in 50000 25000 12500 6250; do
    RDAMD_COMM_SUM=$mode python3 bench.py --shard sites --sites $sites --steps 40 --warmup 5 --no-cpu-baseline --one-rank-comm 2> /dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$mode: c2 sites/%d = %5d per rank: %9.1f evals/s  %.4f ms per 197-job step  evaluator kernel %.4f ms' % (50000 // $sites, $sites, d['value'], d['ms_per_step'], r['avg_launch_ms']))"
  done
done
S=$(python3 -c "import socket; s=socket.socket(); s.bind(('127.0.0.1',0)); print(s.getsockname()[1])")
RDAMD_BENCH_PG=1 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=$S python3 bench.py --no-cpu-baseline --one-rank-shard-legs --sustain-seconds 0 2> /dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); sm=d['site_sharded']['sum_modes']
for m in ('gather','allreduce'): print('one-rank site_sharded leg, %-9s: %9.1f evals/s, %.4f ms per step, bare collective of 198 doubles back to back %.2f us' % (m, sm[m]['value'], sm[m]['ms_per_step'], sm[m]['collective_us_back_to_back']))"
