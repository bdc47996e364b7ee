#!/usr/bin/env python3
"""What does a root-only step cost the objective launch it runs beside?  c2's shape: one partition
evaluates 110-job batches in a loop (the lock-stepped search's launch width) on low stream
priority, a second one -- a "replica" with materialised CLVs -- takes root-only steps of N
positions on high priority from another host thread, as the search's replicas do.  Prints the
batch rate alone / beside the steps, the steps per second, and the device time one step costs
the batches ((t_beside - t_alone) per step).  Usage: root_interference.py [positions ...]"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import root_digger_amd as rd          # noqa: E402
from root_digger_amd import synth     # noqa: E402

S, n, nb = 50000, 100, 110
w = synth.workload(n, S, 4, 4, 0xD166E5 + 1)
tree = rd.Tree.from_newick(w["newick"])


def make(attrs):
    part = rd.Partition.for_tree(tree, 4, S, 4, attributes=attrs)
    for label, seq in w["seqs"].items():
        part.set_tip_states(tree.tip_index(label), rd.MAP_NT, seq)
    part.set_frequencies(0, np.asarray(part.empirical_frequencies()))
    part.set_subst_params(0, w["subst"])
    part.set_category_rates(w["rates"])
    return part


shared = make(rd.ATTRIB_SITE_REPEATS)
shared.set_stream_priority(1)
freqs = np.asarray(shared.empirical_frequencies())
roots = tree.root_count()
scheds = [shared.schedule(*tree.generate_operations(tree.root_location(i % roots))) for i in range(nb)]
rng = np.random.default_rng(5)
sub = np.array([synth.random_params(12, rng) for _ in range(nb)])
fr = np.tile(freqs, (nb, 1))
handles = rd.Partition.schedule_handles(scheds)

replicas = []
for k in range(4):
    r = make(0)
    r.set_stream_priority(-1)
    rl = tree.root_location(7 + k).with_ratio(0.3)
    ops, pmi, brl = tree.generate_operations(rl)
    r.update_prob_matrices(pmi, brl)
    r.update_clvs(ops)
    op, _, _ = tree.generate_derivative_operations(rl)
    replicas.append((r, op, rl))


def batches(seconds):
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < seconds:
        shared.evaluate_batch(handles, sub, fr)
        k += 1
    return k / (time.perf_counter() - t0)


batches(1.5)
alone = batches(2.0)
print("batches of %d jobs alone: %.1f per s (%.2f us per job)" % (nb, alone, 1e6 / alone / nb))
for npos in [int(x) for x in sys.argv[1:]] or [1, 2, 5, 8]:
    for nthreads in (1, 4):
        stop = False
        counts = [0] * nthreads

        def steps(i):
            r, op, rl = replicas[i]
            al = np.linspace(0.1, 0.9, npos)
            l1, l2 = rl.saved_brlen * al, rl.saved_brlen * (1 - al)
            while not stop:
                r.root_loglikelihood_fused(op, l1, l2)
                counts[i] += 1

        th = [threading.Thread(target=steps, args=(i,)) for i in range(nthreads)]
        for t in th:
            t.start()
        t0 = time.perf_counter()
        beside = batches(2.0)
        dt = time.perf_counter() - t0
        stop = True
        for t in th:
            t.join()
        rate = sum(counts) / dt
        lost = (1.0 / beside - 1.0 / alone) * beside / max(rate, 1e-9)   # seconds of batch time per step
        print("%d positions, %d replica threads: batches %.1f per s (%.2f us per job), %.0f steps per s, "
              "%.1f us of batch time per step" % (npos, nthreads, beside, 1e6 / beside / nb, rate, lost * 1e6))
