#!/usr/bin/env python3
"""profiles/<tag>_summary.md: ONE table per round -- config x {evals/s, evaluator time, fraction of the FP64 peak on
executed operations and on issued FP64 instructions, steps per evaluation, the materialising leg} from the committed
bench lines profiles/<tag>_<config>_bench.json, plus the per-step instruction mix, issue-slot and L2 figures of the
default command from profiles/<tag>_summary.json (collect.sh + summarize.py).   usage: summary_table.py r5"""
import json
import os
import sys

root = os.path.dirname(os.path.abspath(__file__))
tag = sys.argv[1] if len(sys.argv) > 1 else "r5"
names = [("c2", "c2: 100 x 50 000, 4 states, G4 (the bench line)"), ("c3", "c3: 200 x 10 000, 20 states, G4"),
         ("c4", "c4: 500 x 500 000 (unsharded on one GPU)"), ("c5", "c5: 1000 x 100 000 (unsharded)"),
         ("c4_shard", "c4 / 8: 500 x 62 500 (one GPU's shard)"), ("c5_shard", "c5 4x2: 1000 x 50 000 (one GPU's shard)"),
         ("d125", "125.phy: 125 x 19 436 patterns")]
summ = json.load(open(os.path.join(root, tag + "_summary.json")))
out = ["# Round %s: one table (`profiles/%s_*_bench.json`, `%s_summary.json`; one box; regenerate: `python profiles/summary_table.py %s`)\n"
       % (tag[1:], tag, tag, tag),
       "| config | evals/s | evaluator ms / launch | frac of FP64 peak, operations executed | frac, FP64 instructions issued | "
       "steps of / operations per evaluation | materialising traversal: time, frac of HBM peak in algorithmic bytes |",
       "|---|---|---|---|---|---|---|"]
for key, label in names:
    path = os.path.join(root, "%s_%s_bench.json" % (tag, key))
    if not os.path.exists(path):
        continue
    d = json.load(open(path))
    r, k = d["roofline"], d.get("clv_kernel", {})
    issued = r.get("issued_fp64_tflops")
    sch = r.get("schedule", {})
    out.append("| %s | %.1f | %.3f | %.4f | %s | %s | %s |" % (
        label, d["value"], r["avg_launch_ms"], r["frac"], "%.3f" % (issued / 78.6) if issued else "--",
        "%.1f / %d" % (sch.get("steps_per_evaluation", 0), sch.get("operations_per_evaluation", 0)) if sch else "--",
        ("%.0f us in %d launch%s: %s%s%s" % (
            1e3 * k["avg_launch_ms"], k.get("kernel_launches_per_traversal", 1),
            "" if k.get("kernel_launches_per_traversal", 1) == 1 else "es", k["frac"],
            " (uncapped %.3f)" % k["frac_uncapped"] if "frac_uncapped" in k else "",
            "; %.2f of the peak in bytes that crossed HBM" % k["counter_frac"] if k.get("counter_frac") else ""))
        if k else "--"))
iss = json.load(open(os.path.join(root, tag + "_c2_bench.json")))["roofline"]["issue"]
v = max((v for k, v in summ.items() if k.startswith("fused_dna_eval_kernel") and "SQ_INSTS_SALU" in v),
        key=lambda v: v.get("avg_us", 0))
waves = max(v.get("SQ_WAVES", 0), 1)
steps_per_wave = iss["valu_per_wave"] / iss["valu_per_step"]
per_step = lambda name: v.get(name, 0) / waves / steps_per_wave   # noqa: E731
miss = v.get("TCC_MISS_sum", 0) / max(v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0), 1)
out += ["", "Counters of the c2 command (`%s_summary.json`, the evaluator variant with the longest launches: avg %.1f us, %d calls):"
        % (tag, v["avg_us"], v["calls"]), "",
        "| per step of two sites | VALU | of which FP64 | SALU | SMEM | LDS | VMEM | VALU issue slots busy | L2 miss ratio | HBM bytes / launch |",
        "|---|---|---|---|---|---|---|---|---|---|",
        "| c2 | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f | %.3f | %.3f | %.1f MB |" % (
            iss["valu_per_step"], iss["fp64_valu_per_step"], per_step("SQ_INSTS_SALU"), per_step("SQ_INSTS_SMEM"),
            per_step("SQ_INSTS_LDS"), per_step("SQ_INSTS_VMEM"), iss["valu_issue_slot_util"], miss,
            v.get("hbm_bytes_per_launch", 0) / 1e6)]
extra = os.path.join(root, tag + "_summary_notes.md")
if os.path.exists(extra):
    out += ["", open(extra).read().rstrip()]
open(os.path.join(root, tag + "_summary.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
