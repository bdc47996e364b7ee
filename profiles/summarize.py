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


for sub in ("fetch", "write", "sq1", "sq2"):
    for k, cs in pmc(sub).items():
        for c, vals in cs.items():
            out.setdefault(k, {})[c] = sum(vals) / len(vals)
for k, d in out.items():
    if "FETCH_SIZE" in d or "WRITE_SIZE" in d:
        fetch = d.get("FETCH_SIZE", 0.0) * 1024 * 2      # gfx950: x2 (see docstring)
        write = d.get("WRITE_SIZE", 0.0) * 1024
        d["hbm_read_bytes_per_launch"] = fetch
        d["hbm_write_bytes_per_launch"] = write
        d["hbm_bytes_per_launch"] = fetch + write
json.dump(out, open(os.path.join(root, "profiles", tag + "_summary.json"), "w"), indent=1,
          sort_keys=True)
for k in sorted(out, key=lambda k: -out[k].get("pct", 0)):
    d = out[k]
    print("%-42s avg %9.1f us  calls %4d  hbm/launch %s" % (
        k[:42], d.get("avg_us", 0), d.get("calls", 0),
        ("%.1f MB" % (d["hbm_bytes_per_launch"] / 1e6)) if "hbm_bytes_per_launch" in d else "-"))
