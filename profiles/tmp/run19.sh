cd $GRAFT_REPO_ROOT
for i in 1 2; do 
LOCKSTEP=32 timeout 600 python3 tests/tools/e2e_search.py 48 2>&1 | tail -2 | head -1
RDAMD_SIDE=1 LOCKSTEP=32 timeout 600 python3 tests/tools/e2e_search.py 48 2>&1 | tail -2 | head -1
done
