"""Time the materialising traversal kernel (rdamd_update_clvs) on a synthetic
workload, launches queued back to back so the event spans carry no gaps.
usage: python profiles/clv_time.py [n S R reps K]   (K = 4 or 20)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import root_digger_amd as rd
from root_digger_amd import synth

argv = sys.argv[1:] + ["100", "50000", "4", "20", "4"][len(sys.argv) - 1:]
n, S, R, reps, K = (int(a) for a in argv)
w = synth.workload(n, S, K, R, 7)
tree = rd.Tree.from_newick(w["newick"])
part = rd.Partition.for_tree(tree, K, S, R)
cmap = rd.MAP_NT
if K != 4:
    import ctypes
    cmap = (ctypes.c_uint64 * 256)()
    for i, ch in enumerate(w["alphabet"]):
        cmap[ord(ch)] = 1 << i
for label, seq in w["seqs"].items():
    part.set_tip_states(tree.tip_index(label), cmap, seq)
part.set_frequencies(0, part.empirical_frequencies())
part.set_category_rates(w["rates"])
part.set_subst_params(0, synth.random_params(K * K - K, np.random.default_rng(3)))
scheds = [tree.generate_operations(tree.root_location(j)) for j in range(reps)]
for rep in range(2):
    part.profile_enable(rep == 1)
    for ops, pmi, brl in scheds:
        part.update_prob_matrices(pmi, brl)
        part.update_clvs(ops)
    part.sync()
ms, launches = part.profile_read()["clv"]
us = ms / launches * 1e3
W = S * R * K * 8
alg = (2 * n - 3) * W + n * S + (2 * n - 3) * 4 * S     # bench.py clv_kernel_bytes
stores = (n - 1) * (W + 4 * S)
print("n=%d S=%d R=%d K=%d: %.1f us/traversal, algorithmic %.0f GB/s (%.3f of 8 TB/s), stores alone %.0f GB/s"
      % (n, S, R, K, us, alg / us / 1e3, alg / us / 1e3 / 8000, stores / us / 1e3))
