"""Wall time of rdamd_evaluate_batch against the batch size (c2 shape): what a
lock-stepped optimiser gains by handing the GPU wider batches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import root_digger_amd as rd
from root_digger_amd import synth

n, S, R = 100, 50000, 4
w = synth.workload(n, S, 4, R, 0xD166E5 + 1)
tree = rd.Tree.from_newick(w["newick"])
part = rd.Partition.for_tree(tree, 4, S, R)
for label, seq in w["seqs"].items():
    part.set_tip_states(tree.tip_index(label), rd.MAP_NT, seq)
freqs = np.asarray(part.empirical_frequencies())
part.set_frequencies(0, freqs)
part.set_category_rates(w["rates"])
rng = np.random.default_rng(5)
scheds = [part.schedule(*tree.generate_operations(tree.root_location(i))) for i in range(tree.root_count())]
for nb in (1, 2, 4, 13, 26, 52, 104, 197, 394):
    sub = np.array([synth.random_params(12, rng) for _ in range(nb)])
    fr = np.tile(freqs, (nb, 1))
    sc = [scheds[i % len(scheds)] for i in range(nb)]
    part.evaluate_batch(sc, sub, fr)
    reps = max(3, 400 // nb)
    t0 = time.perf_counter()
    for _ in range(reps):
        part.evaluate_batch(sc, sub, fr)
    dt = (time.perf_counter() - t0) / reps
    print("batch %4d: %8.3f ms/batch  %7.1f us/job  %8.0f evals/s" % (nb, dt * 1e3, dt / nb * 1e6, nb / dt))
