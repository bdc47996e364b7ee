cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "20_states" 2>&1 | tail -3
timeout 600 python bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300
