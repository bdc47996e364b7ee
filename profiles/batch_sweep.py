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
part = rd.Partition.for_tree(tree, 4, S, R, attributes=rd.ATTRIB_SITE_REPEATS)
for label, seq in w["seqs"].items():
    part.set_tip_states(tree.tip_index(label), rd.MAP_NT, seq)
freqs = np.asarray(part.empirical_frequencies())
part.set_frequencies(0, freqs)
part.set_category_rates(w["rates"])
rng = np.random.default_rng(5)
scheds = [part.schedule(*tree.generate_operations(tree.root_location(i))) for i in range(tree.root_count())]
for nb in (1, 2, 4, 13, 26, 52, 78, 104, 130, 156, 197, 296, 394, 788):
    sub = np.array([synth.random_params(12, rng) for _ in range(nb)])
    fr = np.tile(freqs, (nb, 1))
    sc = [scheds[i % len(scheds)] for i in range(nb)]
    part.evaluate_batch(sc, sub, fr)
    reps = max(3, 2000 // nb)
    part.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        part.evaluate_batch(sc, sub, fr)
    dt = (time.perf_counter() - t0) / reps
    prof = part.profile_read()
    part.profile_enable(False)
    k = prof["fused"][0] / max(prof["fused"][1], 1)   # the evaluator kernel alone (events), ms
    print("batch %4d: %8.3f ms/batch  %7.1f us/job  %8.0f evals/s   evaluator kernel %8.1f us = %6.2f us/job"
          % (nb, dt * 1e3, dt / nb * 1e6, nb / dt, k * 1e3, k * 1e3 / nb))
