cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in HEAD L0 L3; do
python3 profiles/with_ablation.py $PWD/profiles/tmp_libs/$v.so bench.py --no-cpu-baseline --allow-stale-profile --sustain-seconds 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('variant $v c2 %9.1f evals/s kernel %.4f ms' % (d['value'], r['avg_launch_ms']))"
done; done
