cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-1800
