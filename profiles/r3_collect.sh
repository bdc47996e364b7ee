#!/bin/bash
# Round-3 evidence, one box: kernel trace + PMC passes of the default bench command
# (profiles/collect.sh), the bench lines of every BASELINE config and of the 125.phy fixture,
# the A/B table of the subtree site repeats, and the 20-state traversal's times + trace.
# Outputs under gpurun_out/r3/; profiles/summarize.py r3 turns them into profiles/r3_*.
export TMPDIR=/tmp; R=/root/repo; OUT=$R/gpurun_out/r3; mkdir -p $OUT
bash $R/profiles/collect.sh r3 > $OUT/collect.log 2>&1
cd $R
B="python3 bench.py --allow-stale-profile"
$B > $OUT/bench_c2.json 2> $OUT/bench_c2.err
$B --config c3 --steps 5 --warmup 1 > $OUT/bench_c3.json 2> $OUT/bench_c3.err
$B --config c4 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err
$B --config c5 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_c5.json 2> $OUT/bench_c5.err
$B --config d125 --cpu-seconds 6 > $OUT/bench_d125.json 2> $OUT/bench_d125.err
# subtree site repeats: none / class limit 16 / 64 on three workloads
for c in c2 c5 d125; do for k in 0 16 64; do
  if [ $k = 0 ]; then X="--no-repeats"; else X="--repeat-classes $k"; fi
  $B --config $c --steps 6 --warmup 2 --no-cpu-baseline --sustain-seconds 0 $X 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; s=r['schedule']
print('%-5s limit %2d: %9.1f evaluations/s  kernel %8.3f ms  steps %6.1f of %d  matvecs %6.1f  frac %.3f  algorithmic-equivalent %.3f' % ('$c', $k, d['value'], r['avg_launch_ms'], s['steps_per_evaluation'], s['operations_per_evaluation'], s['matvecs_per_evaluation'], r['frac'], r['algorithmic_equiv']['ratio_to_peak']))"
done; done > $OUT/repeats_ab.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k20trace -- python3 $R/profiles/clv_time.py 200 10000 4 20 20 > $OUT/k20trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3trace -- python3 $R/bench.py --config c3 --steps 5 --warmup 1 --no-cpu-baseline --allow-stale-profile --sustain-seconds 0 > $OUT/c3trace.log 2>&1
cd $R; python3 profiles/clv_time.py 200 10000 4 20 20 > $OUT/k20_time.txt; python3 profiles/clv_time.py 200 12288 4 20 20 >> $OUT/k20_time.txt; python3 profiles/clv_time.py 200 100000 4 6 20 >> $OUT/k20_time.txt
python3 profiles/clv_time.py 100 50000 4 20 4 >> $OUT/k20_time.txt
LOCKSTEP=16 python3 tests/tools/e2e_search.py 32 > $OUT/e2e_lockstep.txt 2>&1
cat $OUT/repeats_ab.txt $OUT/k20_time.txt $OUT/e2e_lockstep.txt
ls $OUT
