#!/bin/bash
# one test at a time, each under its own limit, everything logged
export RDAMD_LOCKSTEP_DEBUG=1
mkdir -p gpurun_out
for t in "$@"; do
  echo "=== $t" 
  timeout 240 python -m pytest "$t" -x -q --timeout=200 --timeout-method=thread 2>&1 | tail -40
done
