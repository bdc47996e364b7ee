cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for l in r3 r4; do
D=$GRAFT_REPO_ROOT; [ $l = r3 ] && D=$GRAFT_REPO_ROOT/profiles/tmp_r3
(cd $D && python3 bench.py --no-cpu-baseline --allow-stale-profile --sustain-seconds 1 2>/dev/null) | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$l c2 %9.1f evals/s kernel %.4f ms sustained %.1f  pmat %.4f ms' % (d['value'], r['avg_launch_ms'], r['sustained']['evals_per_s'], r['pmatrix_ms_per_launch']))"
done; done
for l in r3 r4; do
D=$GRAFT_REPO_ROOT; [ $l = r3 ] && D=$GRAFT_REPO_ROOT/profiles/tmp_r3
(cd $D && python3 bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline --allow-stale-profile --sustain-seconds 0 2>/dev/null) | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$l c5 %9.1f evals/s kernel %.3f ms' % (d['value'], r['avg_launch_ms']))"
done
