"""rdamd_evaluate_root_children: one job of the fused evaluator that leaves the root operation's
two children materialised (CLVs + per-site scalers) -- what model_t::exhaustive_search reads
between optimize_params and the root-only steps (src/model.cpp:1154-1229, :415-446).  Checked
against the oracle's full traversal: the lnL, the two CLVs, the root-only evaluations that
follow, and that nothing else of the partition is touched."""
import numpy as np
import pytest

import root_digger_amd as rd
from root_digger_amd import synth
import util
from oracle_lib import OraclePartition, ORC_MAP_NT
from test_gpu_parity import pair, set_model, LNL_TOL

pytestmark = pytest.mark.gpu


def _scaled_equal(a, sa, b, sb, rtol):
    """CLVs with per-site counts of 2^256 rescales: the same numbers?  ([site][rate][state])"""
    sa = sa.astype(np.int64)[:, None, None]
    sb = sb.astype(np.int64)[:, None, None]
    lo = np.minimum(sa, sb)
    fa = np.ldexp(a, (-256 * (sa - lo)).astype(np.int64))
    fb = np.ldexp(b, (-256 * (sb - lo)).astype(np.int64))
    # (atol: entries the per-site rule leaves in the denormal range have lost bits on ITS side)
    return np.allclose(fa, fb, rtol=rtol, atol=1e-250)


def _check_children(g, o, tree, rl, subst, freqs, rates, weights=None, clv_rtol=1e-12):
    ops, pmi, brl = tree.generate_operations(rl)
    got = g.evaluate_root_children(ops, pmi, brl, subst, freqs, rates, weights)
    set_model((o,), subst, freqs, rates, weights)
    want = util.compute_lh(o, tree, rl)
    assert util.rel_err(got, want) < LNL_TOL
    root = ops[len(ops) - 1]
    for clv, sc in ((root.child1_clv_index, root.child1_scaler_index),
                    (root.child2_clv_index, root.child2_scaler_index)):
        if clv < tree.tip_count():
            continue
        assert _scaled_equal(g.get_clv(clv), g.get_scaler(sc), o.get_clv(clv), o.get_scaler(sc), clv_rtol), clv
    # the root-only steps that follow (compute_lh_root / compute_dlh / optimize_alpha)
    set_model((g,), subst, freqs, rates, weights)
    op, _, _ = tree.generate_derivative_operations(rl)
    alphas = [rl.brlen_ratio, 0.0, 1.0, 0.5, 0.123]
    l1 = [rl.saved_brlen * a for a in alphas]
    l2 = [rl.saved_brlen * (1 - a) for a in alphas]
    a = g.root_loglikelihood_fused(op, l1, l2)
    b = o.root_loglikelihood_fused(op, l1, l2)
    for x, y in zip(a, b):
        assert util.rel_err(x, y) < LNL_TOL
    assert util.rel_err(a[0], got) < 1e-12   # full == root-only (test/src/model.cpp:285-286)
    return got


@pytest.mark.parametrize("n,S,R,seed", [(30, 1000, 4, 31), (100, 2000, 4, 41), (37, 777, 1, 42), (12, 64, 2, 43),
                                        (64, 130, 8, 44)])
def test_root_children_vs_oracle(n, S, R, seed):
    w = synth.workload(n, S, 4, R, seed)
    tree = rd.Tree.from_newick(w["newick"])
    g, o = pair(tree, w["seqs"], 4, R)
    freqs = g.empirical_frequencies()
    for i in (0, 3, tree.root_count() // 2, tree.root_count() - 1):
        _check_children(g, o, tree, tree.root_location(i).with_ratio(0.37), w["subst"], freqs, w["rates"])


def test_root_children_on_the_rescaling_caterpillar():
    """per-(site, rate) counts of the evaluator -> the per-site scalers the root kernels read"""
    gd = util.golden("deep_scaling.json")
    tree = rd.Tree.from_newick(gd["newick"])
    g, o = pair(tree, gd["seqs"], 4, 4)
    seen = 0
    for i in (0, 100, 250, tree.root_count() - 1):
        rl = tree.root_location(i).with_ratio(0.31)
        _check_children(g, o, tree, rl, gd["subst"], gd["freqs"], gd["rates"])
        ops, _, _ = tree.generate_operations(rl)
        root = ops[len(ops) - 1]
        for sc in (root.child1_scaler_index, root.child2_scaler_index):
            if sc >= 0:
                seen = max(seen, int(g.get_scaler(sc).max()))
    assert seen >= 1   # the case really rescales


def test_root_children_touch_nothing_else_and_agree_with_the_batch():
    w = synth.workload(40, 900, 4, 4, 77)
    tree = rd.Tree.from_newick(w["newick"])
    g, o = pair(tree, w["seqs"], 4, 4)
    freqs = g.empirical_frequencies()
    set_model((g,), w["subst"], freqs, w["rates"])
    rl0 = tree.root_location(5).with_ratio(0.5)
    util.compute_lh(g, tree, rl0)   # a full traversal: every CLV materialised for rl0
    ops0, _, _ = tree.generate_operations(rl0)
    before = {op.parent_clv_index: g.get_clv(op.parent_clv_index) for op in ops0}
    # other parameters, same root: only the root's children may change
    subst2 = [v * 1.3 for v in w["subst"]]
    ops, pmi, brl = tree.generate_operations(rl0)
    got = g.evaluate_root_children(ops, pmi, brl, subst2, freqs, w["rates"])
    root = ops[len(ops) - 1]
    kids = {root.child1_clv_index, root.child2_clv_index}
    for clv, old in before.items():
        same = np.array_equal(g.get_clv(clv), old)
        assert same != (clv in kids and clv >= tree.tip_count()), clv
    # the value is the batch evaluator's for the same job (1e-12: plain program here, clade tables there)
    sched = g.schedule(ops, pmi, brl)
    assert util.rel_err(got, g.evaluate_batch([sched], [subst2], [freqs], [w["rates"]])[0]) < 1e-12
    # and it is deterministic
    assert got == g.evaluate_root_children(ops, pmi, brl, subst2, freqs, w["rates"])


def test_root_children_binary_and_errors():
    # binary data: the 4-state machinery with two inert states, caller-shaped parameters
    w = synth.workload(40, 1500, 2, 4, 77)
    tree = rd.Tree.from_newick(w["newick"])
    cmap = util.make_map(w["alphabet"], {"-": 3, "?": 3})
    g, o = pair(tree, w["seqs"], 2, 4, cmap, cmap)
    _check_children(g, o, tree, tree.root_location(21).with_ratio(0.35), [0.7, 1.9], g.empirical_frequencies(),
                    w["rates"])
    # a list that is no traversal
    ops, pmi, brl = tree.generate_operations(tree.root_location(0))
    with pytest.raises(rd.RdamdError):
        g.evaluate_root_children([ops[len(ops) - 1]], pmi, brl, [0.7, 1.9], g.empirical_frequencies(), w["rates"])
    # 20 states with more than eight rate categories, and general state counts, are refused (the
    # searches keep the traversal there)
    for K, R in ((20, 9), (5, 2)):
        wk = synth.workload(12, 40, K, R, 6)
        tk = rd.Tree.from_newick(wk["newick"])
        gk = rd.Partition.for_tree(tk, K, 40, R)
        util.load_tips(gk, tk, wk["seqs"], util.make_map(wk["alphabet"]))
        ops, pmi, brl = tk.generate_operations(tk.root_location(0))
        with pytest.raises(rd.RdamdError):
            gk.evaluate_root_children(ops, pmi, brl, wk["subst"], [1.0 / K] * K, wk["rates"])


@pytest.mark.parametrize("n,S,R,seed", [(30, 500, 4, 131), (9, 33, 1, 132), (64, 130, 2, 133), (200, 300, 4, 134),
                                         (30, 200, 8, 135), (17, 97, 6, 136)])   # (five to eight categories: the 512-thread variant)
def test_root_children_20_states_vs_oracle(n, S, R, seed):
    """the exporting variant of fused20_eval_kernel -- the searches' compute_lh in
    front of the root-only steps on protein data (src/model.cpp:415-446) is one job of the fused
    evaluator that leaves the root's two children behind in the partition's operand layout"""
    w = synth.workload(n, S, 20, R, seed)
    tree = rd.Tree.from_newick(w["newick"])
    cmap = util.make_map(w["alphabet"])
    g, o = pair(tree, w["seqs"], 20, R, cmap, cmap)
    freqs = g.empirical_frequencies()
    for i in (0, 3, tree.root_count() // 2, tree.root_count() - 1):
        _check_children(g, o, tree, tree.root_location(i).with_ratio(0.37), w["subst"], freqs, w["rates"])


def test_root_children_20_states_rescaling_and_sparse():
    """a 161-taxon caterpillar of protein data: the per-(site, rate) counts of the evaluator become
    the per-site scalers the root kernels read; and the same on a sparse partition (a replica's)"""
    gd = util.golden("deep_scaling.json")
    tree = rd.Tree.from_newick(gd["newick"])
    rng = np.random.default_rng(5)
    aa = synth.AA
    seqs = {k: "".join(aa[i] for i in rng.integers(0, 20, 96)) for k in gd["seqs"]}
    cmap = util.make_map(aa)
    g, o = pair(tree, seqs, 20, 4, cmap, cmap)
    subst = list(rng.uniform(0.05, 2.0, 380))
    freqs = list(rng.dirichlet(np.ones(20) * 20))
    rates = rd.compute_gamma_cats(0.5, 4)
    seen = 0
    for i in (0, 100, tree.root_count() - 1):
        rl = tree.root_location(i).with_ratio(0.31)
        _check_children(g, o, tree, rl, subst, freqs, rates, clv_rtol=1e-11)
        ops, _, _ = tree.generate_operations(rl)
        root = ops[len(ops) - 1]
        for sc in (root.child1_scaler_index, root.child2_scaler_index):
            if sc >= 0:
                seen = max(seen, int(g.get_scaler(sc).max()))
    assert seen >= 1   # the case really rescales
    sp = rd.Partition.for_tree(tree, 20, 96, 4, rd.ATTRIB_NONREV | rd.ATTRIB_SPARSE_CLVS)
    util.load_tips(sp, tree, seqs, cmap)
    rl = tree.root_location(100).with_ratio(0.31)
    ops, pmi, brl = tree.generate_operations(rl)
    sp.discard_clvs()
    assert sp.evaluate_root_children(ops, pmi, brl, subst, freqs, rates) == \
        g.evaluate_root_children(ops, pmi, brl, subst, freqs, rates)
    root = ops[len(ops) - 1]
    for clv in (root.child1_clv_index, root.child2_clv_index):
        if clv >= tree.tip_count():
            assert np.array_equal(sp.get_clv(clv), g.get_clv(clv))
    assert sp.clv_bytes() < g.clv_bytes() / 20


def test_search_with_root_children_agrees_with_the_traversal_search():
    """exhaustive_search with compute_lh_for_root_steps (the default) against the same search on
    full traversals: the same optimiser on values that differ in their last bits -- the records
    agree to the optimiser's tolerances, lock step == sequential stays bit for bit, and no
    traversal kernel is asked for between the candidates' root-only steps."""
    import ctypes
    import os
    ref = os.path.join(util.ROOT, "oracle", "_ref", "liblbfgsb_ref.so")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/liblbfgsb_ref.so not built (make -C oracle ref)")
    lb = ctypes.CDLL(ref)
    w = synth.workload(12, 4000, 4, 4, 978)
    tree = rd.Tree.from_newick(w["newick"])
    m = rd.Model(tree, w["seqs"], rate_cats=4, seed=5)
    m.initialize_partitions()
    m.set_lbfgsb(lb.setulb)
    m.compute_lh(tree.root_location(0))
    tol = (1e-6, 1e-6, 1e-8, 1e7)
    new = m.exhaustive_search(*tol)
    lock = m.exhaustive_search(*tol, lockstep=6)
    order = np.argsort(new["root_id"])
    assert np.array_equal(lock["llh"], new["llh"][order]) and np.array_equal(lock["alpha"], new["alpha"][order])
    m.set_root_children_only(False)
    old = m.exhaustive_search(*tol)
    m.set_root_children_only(True)
    assert list(old["root_id"]) == list(new["root_id"])
    assert np.allclose(old["llh"], new["llh"], rtol=1e-7, atol=0.0)
    assert np.allclose(old["alpha"], new["alpha"], atol=1e-3)
    assert abs(old["best_llh"] - new["best_llh"]) <= 1e-7 * abs(old["best_llh"])


def test_root_children_with_a_vanishing_rate_category():
    """A category whose rate is ~0 never rescales while the others do: the evaluator's counts per
    (site, rate) drift apart by several 2^256 steps and the per-site scalers written for the
    root kernels put that category back by 2^(-256 d) -- into or below the denormal range, where
    the per-site rule's own chain of products ends up too (entries there are compared by
    magnitude only); the likelihoods agree to the usual tolerance."""
    gd = util.golden("deep_scaling.json")
    tree = rd.Tree.from_newick(gd["newick"])
    g, o = pair(tree, gd["seqs"], 4, 4)
    rates = [1e-42] + list(gd["rates"][1:])
    for i in (0, 100, tree.root_count() - 1):
        _check_children(g, o, tree, tree.root_location(i).with_ratio(0.31), gd["subst"], gd["freqs"], rates)
