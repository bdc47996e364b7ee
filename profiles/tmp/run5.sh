cd $GRAFT_REPO_ROOT
for cfg in "0 0" "1 0" "1 1"; do set -- $cfg; echo "== PRIO=$1 GROUPS=$2"; PRIO=$1 GROUPS=$2 LOCKSTEP=16 timeout 300 python3 tests/tools/e2e_search.py 12 2>&1 | tail -2; done
PRIO=1 GROUPS=0 LOCKSTEP=16 bash profiles/e2e_diag.sh prio 12 | tail -62 | head -50
