#!/bin/bash
# Randomised parity soak of the library as built (tests/tools/stress_parity.py), three seeds,
# the last one with two sites per lane forced in the 4-state fused evaluator.
# Usage: gpurun -- 'bash profiles/soak.sh [seconds per seed]'
O=gpurun_out/soak; mkdir -p $O
T=${1:-300}
python tests/tools/stress_parity.py $T 101 2>&1 | tail -3 | tee $O/seed101.txt
python tests/tools/stress_parity.py $T 202 2>&1 | tail -3 | tee $O/seed202.txt
make -s -C root_digger_amd/csrc ablation >/dev/null
RDAMD_FUSED_NS=2 python profiles/with_ablation.py $PWD/root_digger_amd/lib/librdamd_ablation.so tests/tools/stress_parity.py $T 303 2>&1 | tail -3 | tee $O/seed303_ns2.txt
