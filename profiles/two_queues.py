#!/usr/bin/env python3
"""Two device-filling evaluator launches on two streams at once (a partitioned model's two shared
partitions in a lock-stepped search) against the same launches one after the other: does the
dispatcher's cost per interleaved workgroup (root_interference.py) show up between equals?
c2's sites split in two partitions of 25 000; 110-job batches on each.  Usage: two_queues.py"""
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
roots = tree.root_count()
rng = np.random.default_rng(5)
sub = np.array([synth.random_params(12, rng) for _ in range(nb)])


def make(lo, hi):
    part = rd.Partition.for_tree(tree, 4, hi - lo, 4, attributes=rd.ATTRIB_SITE_REPEATS)
    for label, seq in w["seqs"].items():
        part.set_tip_states(tree.tip_index(label), rd.MAP_NT, seq[lo:hi])
    part.set_category_rates(w["rates"])
    freqs = np.asarray(part.empirical_frequencies())
    scheds = [part.schedule(*tree.generate_operations(tree.root_location(i % roots))) for i in range(nb)]
    return part, rd.Partition.schedule_handles(scheds), np.tile(freqs, (nb, 1)), scheds


parts = [make(0, S // 2), make(S // 2, S)]


def run(p, seconds, out, i):
    part, handles, fr, _ = p
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < seconds:
        part.evaluate_batch(handles, sub, fr)
        k += 1
    out[i] = k / (time.perf_counter() - t0)


out = [0.0, 0.0]
for p in parts:
    run(p, 1.0, out, 0)
run(parts[0], 2.0, out, 0)
alone = out[0]
print("one partition alone: %.1f batches/s" % alone)
# one after the other from one thread
t0 = time.perf_counter()
k = 0
while time.perf_counter() - t0 < 2.0:
    for part, handles, fr, _ in parts:
        part.evaluate_batch(handles, sub, fr)
    k += 1
seq = k / (time.perf_counter() - t0)
print("both, one after the other: %.1f rounds/s (= %.1f batches/s)" % (seq, 2 * seq))
th = [threading.Thread(target=run, args=(parts[i], 2.0, out, i)) for i in range(2)]
for t in th:
    t.start()
for t in th:
    t.join()
print("both at once from two threads: %.1f + %.1f = %.1f batches/s" % (out[0], out[1], out[0] + out[1]))
