mkdir -p gpurun_out/spec
one() { python bench.py "$@" --no-cpu-baseline --allow-stale-profile --sustain-seconds 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('   %-44s %9.1f evals/s  kernel %8.4f ms  frac %.4f' % (' '.join(sys.argv[1:]), d['value'], r['avg_launch_ms'], r['frac']))" "$@"; }
for i in 1 2; do
one --steps 20 --warmup 3; one --config d125 --steps 20 --warmup 3; one --repeat-classes 16 --steps 20 --warmup 3; one --no-repeats --steps 20 --warmup 3
one --config c5 --sites 50000 --steps 4 --warmup 1; one --config c4 --sites 62500 --steps 4 --warmup 1; one --shard sites --sites 6250 --steps 40 --warmup 5
done 2>&1 | tee gpurun_out/spec/product_quick.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_repeats.py tests/test_gpu_rescale_speculation.py tests/test_gpu_lockstep_rounds.py -x -q 2>&1 | tail -3
