cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_repeats.py tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q 2>&1 | tail -4
for rep in 1 2; do for v in HEAD NEW; do
python3 profiles/with_ablation.py $PWD/profiles/tmp_libs/$v.so bench.py --no-cpu-baseline --allow-stale-profile --sustain-seconds 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v c2 %9.1f evals/s kernel %.4f ms' % (d['value'], r['avg_launch_ms']))"
done; done
for v in HEAD NEW; do
python3 profiles/with_ablation.py $PWD/profiles/tmp_libs/$v.so bench.py --config c5 --sites 50000 --steps 3 --warmup 1 --no-cpu-baseline --allow-stale-profile --sustain-seconds 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v c5shard %9.1f evals/s kernel %.4f ms' % (d['value'], r['avg_launch_ms']))"
done
