cd $GRAFT_REPO_ROOT
LOCKSTEP=16 bash profiles/e2e_diag.sh w2 12 | tail -40 | head -32
