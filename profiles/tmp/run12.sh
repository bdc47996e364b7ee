cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_repeats.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -8
for k in 64 256; do for c in c2 c5:50000 d125; do cfg=${c%%:*}; st="--steps 6 --warmup 2"; sites=""; [ $c = c5:50000 ] && { st="--steps 3 --warmup 1"; sites="--sites 50000"; }
python3 bench.py --config $cfg $sites $st --no-cpu-baseline --allow-stale-profile --sustain-seconds 0 --repeat-classes $k 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; s=r['schedule']
print('%-5s limit %3d: %9.1f evaluations/s  kernel %8.3f ms  steps %6.1f of %d  frac %.3f  pmat+clade %.3f ms' % ('$cfg', $k, d['value'], r['avg_launch_ms'], s['steps_per_evaluation'], s['operations_per_evaluation'], r['frac'], r['pmatrix_ms_per_launch']))"
done; done
