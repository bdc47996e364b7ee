cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_repeats.py tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_pipeline.py tests/test_gpu_optimizer.py -x -q 2>&1 | tail -4
for cfg in "c2 50000 20 3" "c5 50000 3 1" "c4 62500 3 1" "d125 0 20 3"; do set -- $cfg
S=""; [ $2 != 0 ] && S="--sites $2"
python3 bench.py --config $1 $S --steps $3 --warmup $4 --no-cpu-baseline --allow-stale-profile --sustain-seconds 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$1 %9.1f evals/s kernel %.3f ms frac %.3f' % (d['value'], r['avg_launch_ms'], r['frac']))"
done
