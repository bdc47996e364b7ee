#!/usr/bin/env python3
"""Randomised parity soak (run by hand on a GPU box, not collected by pytest):
random shapes (4, 2 and 20 states; one tree in eight perfectly balanced: the deepest
traversal stacks), random valid operation orders, random root
placements, subtree site repeats off / class limit 16 / 64, gaps and ambiguity
codes, the occasional vanishing rate category -- the materialising kernels and
the fused evaluators against the CPU oracle.
usage: stress_parity.py [seconds] [seed]"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
import root_digger_amd as rd
from root_digger_amd import synth
from oracle_lib import OraclePartition, ORC_MAP_NT
import util

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0, rounds, worst, folded, children = time.time(), 0, 0.0, 0, 0
while time.time() - t0 < budget:
    n = int(rng.integers(4, 260))
    R = int(rng.choice([1, 2, 4, 8]))
    S = int(rng.choice([1, 7, 63, 64, 65, 255, 1000, 4097, 20000]))
    if n * S * R > 6e6:
        S = max(1, int(6e6 // (n * R)))
    K = int(rng.choice([4, 4, 4, 2, 20]))
    if K == 20:                  # the oracle is slow there: smaller cases
        n, S = min(n, 90), min(S, 1000)
    w = synth.workload(n, S, K, R, int(rng.integers(1 << 30)))
    if rng.random() < 0.125:     # a balanced tree over the same tips (branch lengths random)
        nodes = ["%s:%.5f" % (name, rng.uniform(0.01, 0.4)) for name in sorted(w["seqs"])]
        while len(nodes) > 3:    # (pair up level by level; the top keeps three children)
            stop = 2 if len(nodes) == 4 else len(nodes)
            nodes = ["(%s,%s):%.5f" % (nodes[i], nodes[i + 1], rng.uniform(0.01, 0.4)) if i + 1 < len(nodes)
                     else nodes[i] for i in range(0, stop, 2)] + nodes[stop:]
        w["newick"] = "(" + ",".join(nodes) + ");"
    tree = rd.Tree.from_newick(w["newick"])
    cmap = rd.MAP_NT if K == 4 else util.make_map(w["alphabet"])
    # subtree site repeats (4 and 2 states): off / class limit 16 / 64; a third of the
    # nucleotide cases get gaps and ambiguity codes (cherries with more than 16 classes),
    # one in ten a rate category small enough to send the launch to the plain programs
    repeats = int(rng.choice([0, 16, 64])) if K != 20 else 0
    if K == 4 and rng.random() < 0.33:
        for k, v in w["seqs"].items():
            v = np.frombuffer(v.encode(), dtype=np.uint8).copy()
            v[rng.random(S) < 0.2] = ord("-")
            v[rng.random(S) < 0.03] = ord(str(rng.choice(list("RYKMSW"))))
            w["seqs"][k] = v.tobytes().decode()
    if K != 20 and rng.random() < 0.1:
        w["rates"] = [1e-42] + list(w["rates"][1:])
    g = rd.Partition.for_tree(tree, K, S, R, attributes=rd.ATTRIB_SITE_REPEATS if repeats else 0)
    if repeats:
        g.set_site_repeats(repeats)
    folded += repeats > 0
    o = OraclePartition.for_tree(tree, K, S, R)
    util.load_tips(g, tree, w["seqs"], cmap)
    util.load_tips(o, tree, w["seqs"], ORC_MAP_NT if K == 4 else cmap)
    freqs = g.empirical_frequencies()
    if min(freqs) < 1e-3:        # a state that never occurs: the reference falls back to
        freqs = [1.0 / K] * K    # uniform frequencies (src/main.cpp:577-581)
    for p in (g, o):
        p.set_subst_params(0, w["subst"])
        p.set_frequencies(0, freqs)
        p.set_category_rates(w["rates"])
    rl = tree.root_location(int(rng.integers(tree.root_count()))).with_ratio(float(rng.uniform(0, 1)))
    ops, pmi, brl = tree.generate_operations(rl)
    ops = [rd.Operation(*op.astuple()) for op in ops]
    # a random valid order
    done, pending, order = set(range(n)), list(ops), []
    while pending:
        ready = [op for op in pending if op.child1_clv_index in done and op.child2_clv_index in done]
        pick = ready[int(rng.integers(len(ready)))]
        order.append(pick)
        done.add(pick.parent_clv_index)
        pending.remove(pick)
    for p in (g, o):
        p.update_prob_matrices(pmi, brl)
        p.update_clvs(order if rng.random() < 0.5 else ops)
    for op in (ops[0], ops[len(ops) // 2], ops[-1]):
        a, b = g.get_clv(op.parent_clv_index), o.get_clv(op.parent_clv_index)
        assert np.allclose(a, b, rtol=1e-12, atol=0.0), (n, S, R, K, op.parent_clv_index)
        if op.parent_scaler_index >= 0:
            assert np.array_equal(g.get_scaler(op.parent_scaler_index), o.get_scaler(op.parent_scaler_index))
    la = g.compute_root_loglikelihood(tree.root_clv_index(), tree.root_scaler_index())
    lb = o.compute_root_loglikelihood(tree.root_clv_index(), tree.root_scaler_index())
    fused = lb
    if K != 20 or R <= 8:        # the shapes the fused evaluators take (20 states: up to eight rate categories since round 6)
        fused = g.evaluate_batch([g.schedule(ops, pmi, brl)], [w["subst"]], [freqs])[0]
    err = max(abs(la - lb), abs(fused - lb)) / abs(lb)
    assert err < 1e-11, (n, S, R, K, la, lb, fused)
    if K != 20 or R <= 8:        # the root's children left behind by the exporting evaluators (4 / 2 / 20 states)
        # (rdamd_evaluate_root_children): the value, the two CLVs up to their scalers, and the
        # root-only evaluation on top of them, with other parameters than the traversal above
        subst2 = [v * float(rng.uniform(0.5, 2.0)) for v in w["subst"]]
        lc = g.evaluate_root_children(ops, pmi, brl, subst2, freqs, w["rates"])
        if rng.random() < 0.3:   # the same on a SPARSE partition (a search replica's): the dense one's bits
            sp = rd.Partition.for_tree(tree, K, S, R, attributes=rd.ATTRIB_SPARSE_CLVS)
            util.load_tips(sp, tree, w["seqs"], cmap)
            assert sp.evaluate_root_children(ops, pmi, brl, subst2, freqs, w["rates"]) == lc, (n, S, R, K, "sparse")
            for clv in (ops[-1].child1_clv_index, ops[-1].child2_clv_index):
                if clv >= n:
                    assert np.array_equal(sp.get_clv(clv), g.get_clv(clv)), (n, S, R, K, clv, "sparse CLV")
            sp.destroy()
        o.set_subst_params(0, subst2)
        g.set_subst_params(0, subst2)
        o.update_prob_matrices(pmi, brl)
        o.update_clvs(ops)
        lo = o.compute_root_loglikelihood(tree.root_clv_index(), tree.root_scaler_index())
        rop, _, _ = tree.generate_derivative_operations(rl)
        lr = g.root_loglikelihood_fused(rop, [rl.saved_brlen * rl.brlen_ratio], [rl.saved_brlen * (1 - rl.brlen_ratio)])[0]
        err2 = max(abs(lc - lo), abs(lr - lo)) / abs(lo)
        assert err2 < 1e-11, (n, S, R, K, lc, lr, lo)
        root = ops[-1]
        for clv, sc in ((root.child1_clv_index, root.child1_scaler_index), (root.child2_clv_index, root.child2_scaler_index)):
            if clv < n:
                continue
            a, b = g.get_clv(clv), o.get_clv(clv)
            sa = g.get_scaler(sc).astype(np.int64)[:, None, None]
            sb = o.get_scaler(sc).astype(np.int64)[:, None, None]
            lo_s = np.minimum(sa, sb)
            fa, fb = np.ldexp(a, -256 * (sa - lo_s)), np.ldexp(b, -256 * (sb - lo_s))
            # (atol: where the per-site rule leaves a rate in or below the denormal range -- a vanishing
            # category: 1.3e-264 here against 0 there, 0 here against 1.9e-309 there in this soak -- its
            # own chain of products has lost the bits; the evaluator's count per (site, rate) has not)
            if not np.allclose(fa, fb, rtol=1e-12, atol=1e-250):
                rel = np.abs(fa - fb) / np.maximum(np.abs(fb), 1e-300)
                i = np.unravel_index(np.argmax(rel), rel.shape)
                print("CLV mismatch", (n, S, R, K, clv, repeats), "worst rel", rel.max(), "at", i, fa[i], fb[i],
                      "scalers", int(sa[i[0], 0, 0]), int(sb[i[0], 0, 0]), "raw", a[i], b[i], "rates", w["rates"][:2])
                raise AssertionError("CLV mismatch")
        err = max(err, err2)
        children += 1
    worst = max(worst, err)
    rounds += 1
    g.destroy()
    o.destroy()
print("%d random cases (%d with subtree site repeats, %d with the root's children from the evaluator) in %.0f s, "
      "worst lnL rel. err %.2e" % (rounds, folded, children, time.time() - t0, worst))
