# fused 20-state evaluator: parity subset + c3 bench line (evaluations/s, kernel ms)
[ -n "$PARITY" ] && timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "20_states or fused" 2>&1 | tail -3
for i in 1 2; do python bench.py --config c3 --steps 5 --warmup 1 --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('c3 evals/s %.0f  kernel ms %.3f  frac %.3f  pmatrix ms %.3f' % (d['value'], r['avg_launch_ms'], r['frac'], r['pmatrix_ms_per_launch']))"; done
