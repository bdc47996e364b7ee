cd $GRAFT_REPO_ROOT
LOCKSTEP=16 GROUPS=0 bash profiles/e2e_diag.sh pipe2 12 | tail -90
