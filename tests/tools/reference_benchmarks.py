#!/usr/bin/env python3
"""(Lives under tests/: it times the CPU oracle beside the GPU path -- nothing
outside tests/ may touch oracle/.)
The reference's own micro-benchmarks (benchmark/src/model.cpp:19-78:
BM_LH_computation, BM_DLH_computation, BM_LH_root_computation, same datasets
and root indices) timed on this implementation: latency of ONE model_t call,
host wall clock, the GPU result synchronously returned each time.  Beside it
the CPU oracle's time for the same full evaluation (1 thread).  The reference
records no results for these, so there is nothing to divide by;
the table is what a user of the reference would compare against their own run.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import root_digger_amd as rd
import util
from oracle_lib import OraclePartition, ORC_MAP_NT

DATA = [("10.fasta", "10.tree", util.read_fasta), ("101.phy", "101.tree", util.read_phylip)]


def timeit(fn, min_time=0.3, min_iters=20):
    fn()
    n, t0 = 0, time.perf_counter()
    while True:
        fn()
        n += 1
        dt = time.perf_counter() - t0
        if dt >= min_time and n >= min_iters:
            return dt / n * 1e6


rows = []
for di, args in (("LH", [(0, 0), (0, 2), (1, 0), (1, 20), (1, 120)]),
                 ("DLH", [(0, 0), (0, 2), (1, 0), (1, 20), (1, 120)]),
                 ("LH_root", [(1, 0), (1, 20), (1, 120)])):
    for d, root in args:
        msa, tre, reader = DATA[d]
        tree = rd.Tree.from_file(os.path.join(util.DATA, tre))
        m = rd.Model.from_file(tree, os.path.join(util.DATA, msa), rate_cats=1, seed=7)
        m.initialize_partitions_uniform_freqs()
        rl = tree.root_location(root)
        m.compute_lh(rl)
        fn = {"LH": lambda: m.compute_lh(rl), "DLH": lambda: m.compute_dlh(rl),
              "LH_root": lambda: m.compute_lh_root(rl)}[di]
        gpu_us = timeit(fn)
        cpu_us = None
        if di == "LH":
            seqs, w = util.compress(reader(os.path.join(util.DATA, msa)))
            o = OraclePartition.for_tree(tree, 4, len(w), 1)
            util.load_tips(o, tree, seqs, ORC_MAP_NT, w)
            o.set_subst_params(0, [1.0] * 12)
            cpu_us = timeit(lambda: util.compute_lh(o, tree, rl))
        rows.append((di, msa, root, m.patterns, gpu_us, cpu_us))
        m.destroy()
print("%-22s %-9s %5s %8s %12s %14s" % ("benchmark", "data", "root", "patterns", "this (us)", "oracle 1T (us)"))
for di, msa, root, pat, g, c in rows:
    print("BM_%-19s %-9s %5d %8d %12.1f %14s" % (di + "_computation", msa, root, pat, g,
                                                "%.1f" % c if c else "-"))
