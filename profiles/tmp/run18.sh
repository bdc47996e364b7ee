cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_optimizer.py tests/test_gpu_model.py -x -q 2>&1 | tail -4
for i in 1 2; do LOCKSTEP=32 timeout 600 python3 tests/tools/e2e_search.py 48 2>&1 | tail -2; done
