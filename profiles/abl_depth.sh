#!/bin/bash
# Timing-only A/B behind DESIGN 4.1 "stack levels in the private segment": the fused 4-state
# evaluator with its LDS stack cut to ONE level (results are garbage: deeper entries overwrite
# each other) against the real thing -- i.e. what the resident waves are worth.  Needs the
# ablation library (make -C root_digger_amd/csrc ablation).  Round 3, before the change:
# c5 2 451 -> 2 995, 125.phy 102k -> 124k evaluations/s.
mkdir -p gpurun_out/abl
ABL="profiles/with_ablation.py $PWD/root_digger_amd/lib/librdamd_ablation.so"; export RDAMD_BENCH_TIMING_ONLY=1
for c in d125 c5; do
  st=""; [ $c = c5 ] && st="--steps 3 --warmup 1"
  python $ABL bench.py --config $c $st --allow-stale-profile --no-cpu-baseline > gpurun_out/abl/${c}_base.json 2> gpurun_out/abl/${c}_base.err
  RDAMD_FUSED_DEPTH=1 python $ABL bench.py --config $c $st --allow-stale-profile --no-cpu-baseline > gpurun_out/abl/${c}_d1.json 2> gpurun_out/abl/${c}_d1.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/abl/*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]; print(f, d["value"], r["avg_launch_ms"], r["stack_depth"])
    except Exception as e: print(f,"ERR",e)
PY
