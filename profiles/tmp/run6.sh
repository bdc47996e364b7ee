cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_model.py tests/test_gpu_optimizer.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -6
for cfg in "16 0" "32 0"; do set -- $cfg; echo "== LOCKSTEP=$1 GROUPS=$2"; GROUPS=$2 LOCKSTEP=$1 timeout 300 python3 tests/tools/e2e_search.py 12 2>&1 | tail -2; done
LOCKSTEP=16 bash profiles/e2e_diag.sh w1 12 | tail -62 | head -34
