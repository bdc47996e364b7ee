cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -4
timeout 300 python profiles/clv_time.py 200 10000 4 10 20
timeout 300 python profiles/clv_time.py 200 100000 4 3 20
