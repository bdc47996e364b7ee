mkdir -p gpurun_out/spec
for m in 0 1 0 1; do
  python bench.py --rescale-speculation $m --no-cpu-baseline --allow-stale-profile --steps 20 --warmup 5 --sustain-seconds 0 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('c2 spec $m', d['value'], d['ms_per_step'], d['roofline'].get('kernel_avg_launch_ms'), d['roofline'].get('frac'))"
done 2>&1 | tee gpurun_out/spec/c2_ab.txt
for m in 0 1; do
  python bench.py --config d125 --rescale-speculation $m --no-cpu-baseline --allow-stale-profile --steps 20 --warmup 5 --sustain-seconds 0 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('d125 spec $m', d['value'], d['ms_per_step'], d['roofline'].get('kernel_avg_launch_ms'), d['roofline'].get('frac'))"
done 2>&1 | tee -a gpurun_out/spec/c2_ab.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_repeats.py -x -q 2>&1 | tail -5 | tee gpurun_out/spec/tests.txt
