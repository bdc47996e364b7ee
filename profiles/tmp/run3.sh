cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_optimizer.py tests/test_gpu_repeats.py tests/test_gpu_parity.py -x -q 2>&1 | tail -6
python3 bench.py --no-cpu-baseline --allow-stale-profile --sustain-seconds 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['roofline']['avg_launch_ms'])"
for q in 4 8 16; do echo "== GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q LOCKSTEP=16 GROUPS=0 timeout 300 python3 tests/tools/e2e_search.py 12 2>&1 | tail -2; done
echo "== one group, 8 queues"; GPU_MAX_HW_QUEUES=8 LOCKSTEP=16 GROUPS=1 timeout 300 python3 tests/tools/e2e_search.py 12 2>&1 | tail -2
