#!/usr/bin/env python3
"""What does the FRONT HALF of a pipelined batch (parameter upload, P-matrix + tip-table kernel, clade
tables: high-priority stream) cost the evaluator launch it runs beside?  Partition A (c2's tree,
S sites) evaluates 130-job batches in a loop, blocking, on low priority.  Partition B holds the same
tree with 64 sites -- its evaluator is nothing, its front half is that of a full-size batch (the
P-matrix work does not depend on the sites) -- and runs pipelined 130-job batches from another thread.
Usage: front_interference.py [sites of A]"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import root_digger_amd as rd          # noqa: E402
from root_digger_amd import synth     # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
n, nb = 100, 130
w = synth.workload(n, max(S, 64), 4, 4, 0xD166E5 + 1)
tree = rd.Tree.from_newick(w["newick"])
roots = tree.root_count()
rng = np.random.default_rng(5)
sub = np.array([synth.random_params(12, rng) for _ in range(nb)])


def make(sites):
    part = rd.Partition.for_tree(tree, 4, sites, 4, attributes=rd.ATTRIB_SITE_REPEATS)
    for label, seq in w["seqs"].items():
        part.set_tip_states(tree.tip_index(label), rd.MAP_NT, seq[:sites])
    part.set_category_rates(w["rates"])
    freqs = np.asarray(part.empirical_frequencies())
    scheds = [part.schedule(*tree.generate_operations(tree.root_location(i % roots))) for i in range(nb)]
    return part, rd.Partition.schedule_handles(scheds), np.tile(freqs, (nb, 1)), scheds


A = make(S)
A[0].set_stream_priority(1)
B = make(64)
B[0].set_stream_priority(1)


def batches(seconds):
    """batches per second (wall) and the evaluator kernel's mean duration in ms (events on A's stream)"""
    part, handles, fr, _ = A
    part.profile_enable(True)
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < seconds:
        part.evaluate_batch(handles, sub, fr)
        k += 1
    dt = time.perf_counter() - t0
    prof = part.profile_read()
    part.profile_enable(False)
    return k / dt, prof["fused"][0] / max(prof["fused"][1], 1)


batches(1.0)
alone, k_alone = batches(3.0)
print("A: %d sites, batches of %d jobs alone: %.1f per s, evaluator kernel %.1f us (%.2f us per job)"
      % (S, nb, alone, k_alone * 1e3, k_alone * 1e3 / nb))
stop = False
count = [0]


def fronts():
    part, handles, fr, _ = B
    slot = 0
    n0 = part.evaluate_batch_submit(slot, handles, sub, fr)
    while not stop:
        n1 = part.evaluate_batch_submit(1 - slot, handles, sub, fr)
        part.evaluate_batch_wait(slot, n0)
        slot, n0 = 1 - slot, n1
        count[0] += 1
    part.evaluate_batch_wait(slot, n0)


th = threading.Thread(target=fronts)
th.start()
t0 = time.perf_counter()
beside, k_beside = batches(4.0)
dt = time.perf_counter() - t0
stop = True
th.join()
rate = count[0] / dt
print("beside B's front halves: %.1f per s, evaluator kernel %.1f us (%.2f us per job); B ran %.0f front halves per s = "
      "%.2f per evaluator launch of A; %.1f us of evaluator time per front half"
      % (beside, k_beside * 1e3, k_beside * 1e3 / nb, rate, rate * k_beside * 1e-3,
         (k_beside - k_alone) * 1e3 / max(rate * k_beside * 1e-3, 1e-9)))
