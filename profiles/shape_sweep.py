"""Fused evaluator throughput across shapes (site-CLV updates per second =
evaluations/s x (n-1) x S): where the kernel leaves the machine idle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import root_digger_amd as rd
from root_digger_amd import synth

shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(10, 1000, 1), (10, 1000, 4), (100, 1000, 4), (100, 5000, 4), (100, 50000, 1),
          (100, 50000, 2), (100, 50000, 4), (100, 50000, 8), (1000, 5000, 4), (30, 200000, 4)]
for n, S, R in shapes:
    w = synth.workload(n, S, 4, R, 11)
    tree = rd.Tree.from_newick(w["newick"])
    part = rd.Partition.for_tree(tree, 4, S, R)
    for label, seq in w["seqs"].items():
        part.set_tip_states(tree.tip_index(label), rd.MAP_NT, seq)
    freqs = np.asarray(part.empirical_frequencies())
    part.set_frequencies(0, freqs)
    part.set_category_rates(w["rates"])
    rng = np.random.default_rng(5)
    roots = tree.root_count()
    scheds = [part.schedule(*tree.generate_operations(tree.root_location(i))) for i in range(min(roots, 64))]
    nb = int(os.environ.get("BATCH", 0)) or int(max(16, min(4096, 4e9 / ((n - 1) * S * R * 60))))    # ~ 4 GF of work per launch
    sub = np.array([synth.random_params(12, rng) for _ in range(nb)])
    fr = np.tile(freqs, (nb, 1))
    sc = [scheds[i % len(scheds)] for i in range(nb)]
    part.evaluate_batch(sc, sub, fr)
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        part.evaluate_batch(sc, sub, fr)
    dt = (time.perf_counter() - t0) / reps
    ups = nb / dt * (n - 1) * S
    print("n=%5d S=%7d R=%d batch=%5d: %9.0f evals/s  %6.1f G site-CLV upd/s  %5.1f TFLOP/s alg" % (
        n, S, R, nb, nb / dt, ups / 1e9, ups * R * 60 / 1e12))
    part.destroy()
