#!/usr/bin/env python3
"""What the evaluator launches of the lock-stepped search share the device with.

Reads a `rocprofv3 --kernel-trace` CSV of tests/tools/e2e_search.py and prints, per evaluator
launch size class: launches, time per job when nothing else overlapped the launch / when
root-only kernels did, the busy fraction of the device (union of all kernel intervals over the
traced span), the gaps between consecutive evaluator launches, and the root-only kernels'
own durations (queued or not).  Usage: e2e_overlap.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict


def main(path):
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            name = r.get("Kernel_Name") or r.get("Name")
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            gx = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
            gy = int(r.get("Grid_Size_Y", 1) or 1)
            wx = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)
            rows.append((s, e, name, gx, gy, wx))
    rows.sort()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    span = (t1 - t0) / 1e9
    # union of busy intervals
    busy = 0
    cur_s, cur_e = rows[0][0], rows[0][1]
    for s, e, *_ in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print("traced span %.3f s, device busy (union of kernels) %.3f s = %.1f %%" % (span, busy / 1e9, 100 * busy / 1e9 / span))

    def kind(n):
        if "fused_dna_eval_kernel" in n: return "export" if ", true>(" in n else "eval"
        if "root_single" in n or "root_multi" in n: return "root"
        if "clv_dna_traversal" in n: return "trav"
        if "copyBuffer" in n or "fillBuffer" in n: return "copy"
        return "other"

    by_kind = defaultdict(list)
    for r in rows:
        by_kind[kind(r[2])].append(r)
    for k, v in sorted(by_kind.items()):
        tot = sum(e - s for s, e, *_ in v)
        print("  %-6s %7d launches  total %8.3f s  mean %9.1f us" % (k, len(v), tot / 1e9, tot / 1e3 / max(len(v), 1)))

    ev = by_kind["eval"]
    others = sorted(by_kind["root"] + by_kind["trav"] + by_kind["other"] + by_kind["export"])
    # overlap of each evaluator launch with non-evaluator kernels (sum of overlapped ns of each)
    import bisect
    starts = [o[0] for o in others]
    maxlen = max((o[1] - o[0] for o in others), default=0)
    stats = defaultdict(lambda: [0, 0.0, 0.0, 0, 0.0, 0.0])   # n_clean, t_clean, jobs_clean, n_mixed, t_mixed, jobs_mixed
    ovl_frac = []
    for s, e, n, gx, gy, wx in ev:
        jobs = gy
        lo = bisect.bisect_left(starts, s - maxlen)
        hi = bisect.bisect_right(starts, e)
        ov = 0
        cnt = 0
        for o in others[lo:hi]:
            a, b = max(s, o[0]), min(e, o[1])
            if b > a:
                ov += b - a
                cnt += 1
        cls = "<=32" if jobs <= 32 else "<=64" if jobs <= 64 else "<=128" if jobs <= 128 else ">128"
        st = stats[cls]
        if cnt == 0:
            st[0] += 1; st[1] += e - s; st[2] += jobs
        else:
            st[3] += 1; st[4] += e - s; st[5] += jobs
        ovl_frac.append((ov / max(e - s, 1), (e - s) / max(jobs, 1), jobs, cnt))
    print("evaluator launches by jobs (grid.y): alone = no other kernel overlapped; mixed = some did")
    for cls in ("<=32", "<=64", "<=128", ">128"):
        st = stats.get(cls)
        if not st: continue
        print("  jobs %-6s alone %6d launches %8.2f us/job   mixed %6d launches %8.2f us/job" % (
            cls, st[0], st[1] / 1e3 / max(st[2], 1), st[3], st[4] / 1e3 / max(st[5], 1)))
    # us/job against number of overlapping kernels (wide launches only)
    bins = defaultdict(lambda: [0, 0.0])
    for of, upj, jobs, cnt in ovl_frac:
        if jobs < 48: continue
        b = min(cnt, 8)
        bins[b][0] += 1; bins[b][1] += upj
    # who the overlappers of the crowded launches are (>= 48 jobs, >= 5 others), by kernel and by overlapped time
    who = defaultdict(lambda: [0, 0])
    for s, e, n, gx, gy, wx in ev:
        if gy < 48: continue
        lo = bisect.bisect_left(starts, s - maxlen)
        hi = bisect.bisect_right(starts, e)
        ov = [(o, min(e, o[1]) - max(s, o[0])) for o in others[lo:hi] if min(e, o[1]) > max(s, o[0])]
        if len(ov) < 5: continue
        for o, t in ov:
            k = o[2].split("(")[0].replace("void rdamd::", "").replace("rdamd::", "")[:60]
            who[k][0] += 1; who[k][1] += t
    if who:
        print("  overlappers of the launches with >= 5 of them: kernel, launches, overlapped time")
        for k, (c, t) in sorted(who.items(), key=lambda kv: -kv[1][1])[:8]:
            print("    %-60s %7d %8.3f s" % (k, c, t / 1e9))
    print("  launches of >= 48 jobs, us/job against the number of overlapping other kernels:")
    for b in sorted(bins):
        print("    %s%d other kernels: %6d launches  %7.2f us/job" % (">=" if b == 8 else "", b, bins[b][0], bins[b][1] / 1e3 / bins[b][0]))
    # gaps between consecutive evaluator launches
    gaps = [ev[i + 1][0] - ev[i][1] for i in range(len(ev) - 1)]
    gaps.sort()
    if gaps:
        tot_gap = sum(g for g in gaps if g > 0)
        print("gaps between consecutive evaluator launches: total %.3f s, median %.1f us, p90 %.1f us, p99 %.1f us" % (
            tot_gap / 1e9, gaps[len(gaps) // 2] / 1e3, gaps[int(len(gaps) * 0.9)] / 1e3, gaps[int(len(gaps) * 0.99)] / 1e3))
    # a window of the timeline in the middle of the run: evaluator launches and the P-matrix
    # kernels in front of them (does the next batch's front half start before this evaluator ends?)
    mid = ev[len(ev) // 2][0]
    win = [r for r in rows if mid <= r[0] < mid + 12_000_000 and ("fused_dna_eval" in r[2] or "fused_pmatrix" in r[2])]
    print("timeline window (us from its start): kind start end jobs")
    for s_, e_, n_, gx, gy, wx in win[:70]:
        k = "EVAL" if "fused_dna_eval" in n_ else "pmat"
        jobs = gy if k == "EVAL" else -1
        print("   %-4s %9.1f %9.1f  %s" % (k, (s_ - mid) / 1e3, (e_ - mid) / 1e3, jobs if jobs >= 0 else "grid %d" % gx))
    rt = sorted(e - s for s, e, *_ in by_kind["root"])
    if rt:
        print("root-only kernels: median %.1f us, p10 %.1f us, p90 %.1f us" % (
            rt[len(rt) // 2] / 1e3, rt[len(rt) // 10] / 1e3, rt[int(len(rt) * 0.9)] / 1e3))


if __name__ == "__main__":
    main(sys.argv[1])
