#!/usr/bin/env python3
"""Where does the fused 4-state evaluator's cost per step start to rise with the tree size?
For n taxa at a fixed site count: operations left per evaluation, LDS stack depth of the programs,
event-timed kernel time per 197-job launch, and ns per (job, step) -- c2 (n = 100) is the reference.
Usage: taxa_sweep.py [sites] [n ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import root_digger_amd as rd          # noqa: E402
from root_digger_amd import synth     # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
ns = [int(x) for x in sys.argv[2:]] or [50, 100, 150, 200, 300, 500, 1000]
nb = 197
for n in ns:
    w = synth.workload(n, S, 4, 4, 0xD166E5 + 1)
    tree = rd.Tree.from_newick(w["newick"])
    part = rd.Partition.for_tree(tree, 4, S, 4, attributes=rd.ATTRIB_SITE_REPEATS)
    for label, seq in w["seqs"].items():
        part.set_tip_states(tree.tip_index(label), rd.MAP_NT, seq)
    freqs = np.asarray(part.empirical_frequencies())
    part.set_frequencies(0, freqs)
    part.set_category_rates(w["rates"])
    rng = np.random.default_rng(5)
    roots = tree.root_count()
    scheds = [part.schedule(*tree.generate_operations(tree.root_location(i % roots))) for i in range(nb)]
    st = [s.stats() for s in scheds]
    steps = float(np.mean([x["steps"] for x in st]))
    depth = max(x["stack_depth"] for x in st)
    sub = np.array([synth.random_params(12, rng) for _ in range(nb)])
    fr = np.tile(freqs, (nb, 1))
    handles = rd.Partition.schedule_handles(scheds)
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.5:          # clocks up, caches warm
        part.evaluate_batch(handles, sub, fr)
    part.profile_enable(True)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.0:
        part.evaluate_batch(handles, sub, fr)
    prof = part.profile_read()
    part.profile_enable(False)
    ms = prof["fused"][0] / max(prof["fused"][1], 1)
    print("n=%5d  steps %6.1f of %4d  LDS stack depth %d  wide tables %5.1f  kernel %8.3f ms  %6.2f ns per (job, step)" % (
        n, steps, n - 1, depth, float(np.mean([x.get("wide_tables", 0) for x in st])) if "wide_tables" in st[0] else -1,
        ms, ms * 1e6 / (nb * steps)))
    part.destroy()
