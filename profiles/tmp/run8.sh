cd $GRAFT_REPO_ROOT
for cfg in "16 0" "32 0" "48 0" "32 1"; do set -- $cfg; echo "== LOCKSTEP=$1 GROUPS=$2"; GROUPS=$2 LOCKSTEP=$1 timeout 600 python3 tests/tools/e2e_search.py 48 2>&1 | tail -2; done
