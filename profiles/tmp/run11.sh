cd $GRAFT_REPO_ROOT
echo "== c3 tree (200 taxa, 20 states) at 300 sites, 8 candidates, 8 in lock step"
LOCKSTEP=8 timeout 900 python3 tests/tools/e2e_search.py 50 200 300 20 2>&1 | tail -3
echo "== the same, sequential, 2 candidates"
timeout 900 python3 tests/tools/e2e_search.py 2 200 300 20 2>&1 | tail -4
