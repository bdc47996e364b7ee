cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_pipeline.py -x -q 2>&1 | tail -8
for i in 1 2 3; do LOCKSTEP=16 GROUPS=0 timeout 300 python3 tests/tools/e2e_search.py 12 2>&1 | tail -2; done
LOCKSTEP=32 GROUPS=0 bash profiles/e2e_diag.sh pipe 12
