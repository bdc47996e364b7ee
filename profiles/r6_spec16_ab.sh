# the 16-row kernels (partitions without site repeats / class limit 16) with and without the speculative pass
mkdir -p gpurun_out/spec
for args in "--no-repeats" "--repeat-classes 16" "--no-repeats --batch 40"; do for m in 0 1 0 1; do
  python bench.py $args --rescale-speculation $m --no-cpu-baseline --allow-stale-profile --steps 20 --warmup 5 --sustain-seconds 0 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('c2 $args spec $m', d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_ms'), d['roofline'].get('frac'))"
done; done 2>&1 | tee gpurun_out/spec/c2_tr16_ab.txt
