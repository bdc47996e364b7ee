"""All 2n-3 root lnLs at fixed parameters on the c2 shape, three ways:
move_root sweep (the reference's compute_all_root_lh), one fused launch with a
job per root, and the all-directions CLV cache."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import root_digger_amd as rd
from root_digger_amd import synth

n, S = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100, 50000)
w = synth.workload(n, S, 4, 4, 0xD166E5 + 1)
tree = rd.Tree.from_newick(w["newick"])
m = rd.Model(tree, w["seqs"], rate_cats=4, seed=3)
m.initialize_partitions()


def timed(fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    return (time.perf_counter() - t0) / reps * 1e3, out


ta, a = timed(m.compute_all_root_lh, 1)
tb, b = timed(m.compute_all_root_lh_batched)
tc, c = timed(m.compute_all_root_lh_directional)
print("%d roots, %d sites: move_root sweep %.2f ms, fused batch %.2f ms, all-directions cache %.2f ms"
      % (tree.root_count(), S, ta, tb, tc))
print("max rel. difference vs move_root: fused %.2e, directional %.2e" % (
    np.max(np.abs(b - a) / np.abs(a)), np.max(np.abs(c - a) / np.abs(a))))
