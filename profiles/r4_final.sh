#!/bin/bash
# Round 4, the last kernels (root children left by the fused evaluator, root-only steps as few long
# workgroups): counters of the default bench command again (the digest in r4_summary.json follows
# the sources), the whole program on c2 and 125.phy, the trace of the lock-stepped search.
cd $GRAFT_REPO_ROOT
LB=$GRAFT_REPO_ROOT/oracle/_ref/liblbfgsb_ref.so
bash profiles/collect.sh r4 > gpurun_out/r4_collect.log 2>&1
python3 profiles/summarize.py r4 > gpurun_out/r4_summarize.log 2>&1
bash profiles/full_run_c2.sh $LB > gpurun_out/r4_full_run_c2.txt 2>&1
bash profiles/full_run_125.sh $LB > gpurun_out/r4_full_run_125.txt 2>&1
WORKERS=32 LOCKSTEP=32 bash profiles/e2e_diag.sh r4final 50 > gpurun_out/r4_e2e_final.txt 2>&1
timeout 300 python profiles/root_interference.py > gpurun_out/r4_root_interference.txt 2>&1
python3 bench.py > gpurun_out/r4_bench_final.json 2> gpurun_out/r4_bench_final.err
tail -3 gpurun_out/r4_full_run_c2.txt; tail -3 gpurun_out/r4_full_run_125.txt; grep "candidates," gpurun_out/r4_e2e_final.txt
