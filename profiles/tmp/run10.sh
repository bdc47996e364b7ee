cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_multirank_preflight.py tests/test_gpu_bench.py tests/test_gpu_pipeline.py tests/test_gpu_repeats.py -x -q 2>&1 | tail -6
python3 bench.py --no-cpu-baseline --allow-stale-profile --sustain-seconds 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('r4 c2 %9.1f evals/s kernel %.4f ms sustained %.1f  pmat %.4f ms' % (d['value'], r['avg_launch_ms'], r['sustained']['evals_per_s'], r['pmatrix_ms_per_launch']))"
