cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_optimizer.py tests/test_gpu_repeats.py -x -q 2>&1 | tail -5
for cfg in "16 1" "16 0" "32 0" "48 0"; do set -- $cfg; echo "== LOCKSTEP=$1 GROUPS=$2"; LOCKSTEP=$1 GROUPS=$2 timeout 300 python3 tests/tools/e2e_search.py 12 2>&1 | tail -2; done
