#!/usr/bin/env python3
"""(Lives under tests/: it borrows the reference's L-BFGS-B build, oracle/_ref,
as the caller's optimiser -- nothing outside tests/ may touch oracle/.)
End-to-end timing of the exhaustive per-candidate loop (src/model.cpp:1139-1272)
on BASELINE config c2 with the reference's L-BFGS-B (oracle/_ref) driving the
batched GPU objective.  Diagnostic, not the bench line."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import root_digger_amd as rd
from root_digger_amd import synth

ncand = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n, S = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (100, 50000)
K = int(sys.argv[4]) if len(sys.argv) > 4 else 4          # 20: BASELINE c3's kind of data
w = synth.workload(n, S, K, 4, 0xD166E5 + (1 if K == 4 else 2))
tree = rd.Tree.from_newick(w["newick"])
cmap = None
if K != 4:
    cmap = (ctypes.c_uint64 * 256)()
    for i, ch in enumerate(w["alphabet"]):
        cmap[ord(ch)] = 1 << i
m = rd.Model(tree, w["seqs"], states=K, cmap=cmap, rate_cats=4, seed=3)
m.initialize_partitions()
ref = os.path.join(ROOT, "oracle", "_ref", "liblbfgsb_ref.so")
lb = ctypes.CDLL(ref)
m.set_lbfgsb(lb.setulb)
m.compute_lh(tree.root_location(0))
if "CHILDREN" in os.environ:   # 0: the searches' compute_lh stays the full traversal
    m.set_root_children_only(int(os.environ["CHILDREN"]) != 0)
lib = rd.lib
import ctypes as C
workers = int(os.environ.get("WORKERS", "0"))
lockstep = int(os.environ.get("LOCKSTEP", "0"))
if workers or lockstep:
    import numpy as _np
    lib.rdamd_model_assign_by_rank  # noqa
    # first `ncand` candidates in one call, `workers` threads
    m._ok(lib.rdamd_model_assign_by_rank(m._h, 0, max(1, tree.root_count() // ncand)), "assign")
    if "GROUPS" in os.environ:
        m.set_lockstep_groups(int(os.environ["GROUPS"]))
    if "PRIO" in os.environ:
        m.set_lockstep_priority(int(os.environ["PRIO"]))
    t1 = time.time()
    tol = (1e-7, 1e-7, 1e-12, 1e4) if K == 4 else (1e-3, 1e-3, 1e-6, 1e9)   # (380 parameters per candidate)
    res = m.exhaustive_search(*tol, workers=workers, lockstep=lockstep)
    dt = time.time() - t1
    print("%d candidates, %s: %.2fs  (%.3fs per candidate)  sum llh %.6f" % (
        len(res["root_id"]), "%d in lock step" % lockstep if lockstep else "%d workers" % workers,
        dt, dt / len(res["root_id"]), float(sum(res["llh"]))))
    if lockstep:
        st = m.lockstep_stats()
        print("  combined launches: %d objective (%.1f jobs each), %d root-only (%.1f steps each)" % (
            st["objective_launches"], st["objective_jobs"] / max(st["objective_launches"], 1),
            st["root_launches"], st["root_steps"] / max(st["root_launches"], 1)))
    sys.exit(0)
t0 = time.time()
tot = 0
c0 = m.counters()
for rank in range(ncand):
    m.assign_by_rank(rank, tree.root_count())      # one candidate each
    t1 = time.time()
    res = m.exhaustive_search(1e-7, 1e-7, 1e-12, 1e4)
    c1 = m.counters()
    print("candidate %3d  llh %.4f alpha %.4f  %.2fs  %s" % (res["root_id"][0], res["llh"][0], res["alpha"][0], time.time() - t1,
          {k: c1[k] - c0.get(k, 0) for k in c1}), flush=True)
    c0 = c1
print("total %.2fs for %d candidates" % (time.time() - t0, ncand))
