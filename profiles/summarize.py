#!/usr/bin/env python3
"""Turn gpurun_out/<tag>/ (profiles/collect.sh) into committed summaries:
profiles/<tag>_kernel_stats.csv (rocprofv3 --stats) and
profiles/<tag>_summary.json (per-kernel average duration + HBM traffic per
launch from the FETCH_SIZE / WRITE_SIZE passes, corrected as
MI355X_MICROARCH.md prescribes: counters are in KiB, and FETCH_SIZE counts half
of the bytes of wide coalesced streaming reads on gfx950 -> doubled)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sources import source_digest   # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
out = {}


def short(name):
    return name.split("(")[0].replace("void ", "").replace("rdamd::", "")


stats = glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(root, "profiles", tag + "_kernel_stats.csv"))
    for r in csv.DictReader(open(stats[0])):
        out.setdefault(short(r["Name"]), {}).update(
            calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3,
            min_us=float(r["MinNs"]) / 1e3, max_us=float(r["MaxNs"]) / 1e3,
            pct=float(r["Percentage"]))


def pmc(sub):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(src, sub, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


for sub in ("fetch", "write", "sq1", "sq2", "sq3", "mem1"):
    for k, cs in pmc(sub).items():
        for c, vals in cs.items():
            out.setdefault(k, {})[c] = sum(vals) / len(vals)
# the 20-state evaluator on ITS command (bench.py --config c3: profiles/r6_collect.sh writes the passes
# into <tag>/c3_*): the same keys, under the kernel's own names; `command` says which bench command
# an entry's counters belong to (bench.py attaches them to that command only)
c3_stats = glob.glob(os.path.join(src, "c3_trace", "*", "*kernel_stats.csv"))
c3_kernels = set()
if c3_stats:
    shutil.copy(c3_stats[0], os.path.join(root, "profiles", tag + "_c3_kernel_stats.csv"))
    for r in csv.DictReader(open(c3_stats[0])):
        k = short(r["Name"])
        if not k.startswith(("fused20", "clv_k20")):
            continue
        c3_kernels.add(k)
        out.setdefault(k, {}).update(calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3,
                                     min_us=float(r["MinNs"]) / 1e3, max_us=float(r["MaxNs"]) / 1e3,
                                     pct=float(r["Percentage"]), command="c3")
for sub in ("c3_fetch", "c3_write", "c3_sq1", "c3_sq2", "c3_mem1"):
    for k, cs in pmc(sub).items():
        if not k.startswith(("fused20", "clv_k20")):
            continue
        for c, vals in cs.items():
            out.setdefault(k, {})[c] = sum(vals) / len(vals)
        out[k]["command"] = "c3"
for k, d in out.items():
    if "FETCH_SIZE" in d or "WRITE_SIZE" in d:
        fetch = d.get("FETCH_SIZE", 0.0) * 1024 * 2      # gfx950: x2 (see docstring)
        write = d.get("WRITE_SIZE", 0.0) * 1024
        d["hbm_read_bytes_per_launch"] = fetch
        d["hbm_write_bytes_per_launch"] = write
        d["hbm_bytes_per_launch"] = fetch + write
# Issue-slot figures of the FP64-bound kernels (bench.py publishes them under
# roofline.issue).  A fused_dna_eval_kernel wave walks (n-1) operations x R rates
# = 396 steps on the default command (c2); FP64 counters count wave instructions.
# (steps per evaluation: from the un-profiled bench line of the same collection -- with subtree
# site repeats a schedule runs fewer than n-1 operations; 99 x 4 without that line)
steps_per_eval = 99.0
try:
    for line in open(os.path.join(src, "bench_plain.log")):
        if line.startswith("{"):
            steps_per_eval = json.loads(line)["roofline"]["schedule"]["steps_per_evaluation"]
except (OSError, KeyError, ValueError):
    pass
STEPS_PER_WAVE = {"fused_dna_eval_kernel": steps_per_eval * 4}
for k, d in out.items():
    if "SQ_INSTS_VALU_FMA_F64" not in d or not d.get("SQ_WAVES") or not d.get("SQ_INSTS_VALU"):
        continue
    fp64 = d["SQ_INSTS_VALU_FMA_F64"] + d["SQ_INSTS_VALU_MUL_F64"] + d["SQ_INSTS_VALU_ADD_F64"]
    flops = (2 * d["SQ_INSTS_VALU_FMA_F64"] + d["SQ_INSTS_VALU_MUL_F64"] + d["SQ_INSTS_VALU_ADD_F64"]) * 64
    der = {"valu_per_wave": round(d["SQ_INSTS_VALU"] / d["SQ_WAVES"], 1),
           "fp64_valu_share": round(fp64 / d["SQ_INSTS_VALU"], 4)}
    if d.get("avg_us"):
        der["executed_fp64_tflops"] = round(flops / (d["avg_us"] * 1e-6) / 1e12, 2)
        der["executed_fp64_frac_of_78.6"] = round(der["executed_fp64_tflops"] / 78.6, 4)
    if d.get("GRBM_GUI_ACTIVE"):   # sum over the 8 XCDs; one VALU instruction holds a SIMD 4 cycles
        cycles = d["GRBM_GUI_ACTIVE"] / 8.0
        der["valu_issue_slot_util"] = round(d["SQ_INSTS_VALU"] * 4.0 / (1024 * cycles), 4)
    for name, steps in STEPS_PER_WAVE.items():
        if k.startswith(name):
            der["valu_per_step"] = round(d["SQ_INSTS_VALU"] / d["SQ_WAVES"] / steps, 2)
            der["fp64_valu_per_step"] = round(fp64 / d["SQ_WAVES"] / steps, 2)
            der["sites_per_lane"] = 2
            der["steps_per_evaluation"] = steps_per_eval
    d["derived"] = der
# the matrix-core kernel: how busy the FP64 matrix pipe and the CU's address unit were
for k, d in out.items():
    if not k.startswith("fused20") or not d.get("GRBM_GUI_ACTIVE"):
        continue
    cycles = d["GRBM_GUI_ACTIVE"] / 8.0            # (the counter sums the 8 XCDs)
    der = {}
    if d.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        der["mfma_busy"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cycles), 4)        # of 1 024 SIMDs
    if d.get("SQ_INSTS_VALU_MFMA_MOPS_F64") and d.get("avg_us"):
        # v_mfma_f64_4x4x4_4b_f64: 4 blocks x 4x4x4 multiply-adds = 512 flops per wave instruction
        tf = d["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512.0 / (d["avg_us"] * 1e-6) / 1e12
        der["executed_mfma_tflops"] = round(tf, 2)
        der["executed_mfma_frac_of_78.6"] = round(tf / 78.6, 4)
    if d.get("TA_TA_BUSY_sum"):
        der["address_unit_busy"] = round(d["TA_TA_BUSY_sum"] / (256 * cycles), 4)             # of 256 CUs
    if d.get("SQ_ACTIVE_INST_LDS") and d.get("SQ_BUSY_CYCLES"):
        der["lds_inst_active_share"] = round(d["SQ_ACTIVE_INST_LDS"] / max(d.get("SQ_ACTIVE_INST_ANY", 0.0), 1.0), 4)
    if d.get("SQ_WAVES") and d.get("SQ_INSTS_VALU"):
        der["valu_per_wave"] = round(d["SQ_INSTS_VALU"] / d["SQ_WAVES"], 1)
        if d.get("SQ_INSTS_MFMA"):
            der["mfma_per_wave"] = round(d["SQ_INSTS_MFMA"] / d["SQ_WAVES"], 1)
        if d.get("SQ_INSTS_SALU"):
            der["salu_per_wave"] = round(d["SQ_INSTS_SALU"] / d["SQ_WAVES"], 1)
    if d.get("SQ_WAVE_CYCLES") and d.get("SQ_WAIT_ANY"):
        der["wave_time_in_waitcnt"] = round(d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], 4)
        der["wave_time_waiting_to_issue"] = round(d.get("SQ_WAIT_INST_ANY", 0.0) / d["SQ_WAVE_CYCLES"], 4)
    d["derived"] = dict(d.get("derived", {}), **der)
for k, d in out.items():   # what these counters describe (bench.py checks it before publishing them)
    if source_digest(k):
        d["source_digest"] = source_digest(k)
json.dump(out, open(os.path.join(root, "profiles", tag + "_summary.json"), "w"), indent=1,
          sort_keys=True)
for k in sorted(out, key=lambda k: -out[k].get("pct", 0)):
    d = out[k]
    print("%-42s avg %9.1f us  calls %4d  hbm/launch %s" % (
        k[:42], d.get("avg_us", 0), d.get("calls", 0),
        ("%.1f MB" % (d["hbm_bytes_per_launch"] / 1e6)) if "hbm_bytes_per_launch" in d else "-"))
