mkdir -p gpurun_out/spec
for args in "--repeat-classes 16" "--repeat-classes 17" "--repeat-classes 32" ""; do for m in 1 0; do
  python bench.py $args --rescale-speculation $m --no-cpu-baseline --allow-stale-profile --steps 20 --warmup 5 --sustain-seconds 0 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r=d['roofline']; print('c2 $args spec $m', d['value'], d['ms_per_step'], r.get('avg_launch_ms'), r.get('frac'), r.get('executed'))"
done; done 2>&1 | tee gpurun_out/spec/c2_tr_ab.txt
