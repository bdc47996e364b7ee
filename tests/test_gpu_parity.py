"""Parity tests proper: the HIP path (through the C ABI, librdamd.so) against the
CPU oracle on the same inputs, against the committed golden vectors, and --
at BASELINE.json's full c2 size -- through size-independent properties.

Tolerances: P-matrices 1e-13 absolute; CLVs 1e-12 relative; scalers and every
index bit-exact; lnL <= 1e-11 relative vs the oracle (north_star demands
<= 1e-9 relative) and <= 1e-10 vs the SciPy goldens."""
import math
import os

import numpy as np
import pytest

import root_digger_amd as rd
from root_digger_amd import synth
from oracle_lib import OraclePartition, ORC_MAP_NT
import util

pytestmark = pytest.mark.gpu

LNL_TOL = 1e-11
PARAMS = util.golden("ten_fasta.json")["cases"][0]["subst"], [.34, .42, .24, .74, .16, .88, .75, .54, .20, .06, .08, .41]


def pair(tree, seqs, K, R, cmap_gpu=None, cmap_orc=None, weights=None):
    nsites = len(next(iter(seqs.values())))
    g = rd.Partition.for_tree(tree, K, nsites, R)
    o = OraclePartition.for_tree(tree, K, nsites, R)
    util.load_tips(g, tree, seqs, cmap_gpu or rd.MAP_NT, weights)
    util.load_tips(o, tree, seqs, cmap_orc or ORC_MAP_NT, weights)
    return g, o


def set_model(parts, subst, freqs, rates, weights=None):
    for p in parts:
        p.set_subst_params(0, subst)
        p.set_frequencies(0, freqs)
        p.set_category_rates(rates)
        if weights is not None:
            p.set_category_weights(weights)


def compare_state(g, o, ops, tree, clv_rtol=1e-12):
    """every CLV and scaler an op list produced, GPU vs oracle."""
    for op in ops:
        a, b = g.get_clv(op.parent_clv_index), o.get_clv(op.parent_clv_index)
        assert np.allclose(a, b, rtol=clv_rtol, atol=0.0), op.parent_clv_index
        if op.parent_scaler_index >= 0:
            assert np.array_equal(g.get_scaler(op.parent_scaler_index),
                                  o.get_scaler(op.parent_scaler_index))


def test_library_reports_a_device():
    assert rd.device_count() >= 1


@pytest.mark.parametrize("K,R", [(4, 1), (4, 4), (20, 4), (2, 2), (5, 3)])
def test_prob_matrices(K, R):
    rng = np.random.default_rng(7 + K)
    nm = 37
    g = rd.Partition(5, 8, K, 16, 1, nm, R, 8)
    o = OraclePartition(5, 8, K, 16, 1, nm, R, 8)
    subst = rng.uniform(1e-4, 1, K * K - K)
    freqs = rng.dirichlet(np.ones(K) * 4)
    rates = rd.compute_gamma_cats(0.6, R) if R > 1 else [1.0]
    set_model((g, o), subst, freqs, rates)
    idx = rng.permutation(nm).astype(np.uint32)
    bl = np.concatenate([[0.0, 1e-8, 1e-6, 25.0], rng.exponential(0.3, nm - 4)])
    g.update_prob_matrices(idx, bl)
    o.update_prob_matrices(idx, bl)
    for m in range(nm):
        a, b = g.get_pmatrix(m), o.get_pmatrix(m)
        assert np.max(np.abs(a - b)) < 1e-13
        assert np.all(a >= 0.0)
        assert np.allclose(a.sum(axis=2), 1.0, atol=1e-12)


def test_gamma_cats_match_goldens():
    for gm in util.golden("gamma.json"):
        got = rd.compute_gamma_cats(gm["alpha"], gm["cats"],
                                    rd.GAMMA_RATES_MEAN if gm["mode"] == "mean" else rd.GAMMA_RATES_MEDIAN)
        assert np.allclose(got, gm["rates"], rtol=2e-9, atol=1e-12)


@pytest.mark.parametrize("compressed", [False, True])
def test_c1_ten_fasta_all_roots(compressed):
    """BASELINE config c1 on the reference's own fixture, every root, every
    parameter set of test/src/model.cpp:12-17."""
    tree = rd.Tree.from_file(os.path.join(util.DATA, "10.tree"))
    seqs = util.read_fasta(os.path.join(util.DATA, "10.fasta"))
    weights = None
    if compressed:
        seqs, weights = util.compress(seqs)
    gold = util.golden("ten_fasta.json")
    parts = {}
    for case in gold["cases"]:
        R = case["rate_cats"]
        if R not in parts:
            parts[R] = pair(tree, seqs, 4, R, weights=weights)
            assert np.allclose(parts[R][0].empirical_frequencies(), gold["empirical_freqs"], rtol=1e-12)
        g, o = parts[R]
        set_model((g, o), case["subst"], case["freqs"], case["rates"], [1.0 / R] * R)
        for k, root in enumerate(case["roots"]):
            rl = util.find_root(tree, root["near_tips"], root["far_tips"], root["alpha"])
            a = util.compute_lh(g, tree, rl)
            b = util.compute_lh(o, tree, rl)
            assert util.rel_err(a, b) < LNL_TOL
            assert util.rel_err(a, root["lnl"]) < 1e-10
            assert a == util.compute_lh(g, tree, rl)            # test/src/model.cpp:73
            assert util.rel_err(util.compute_lh_root(g, tree, rl), a) < 1e-13   # :285-286
            if k % 5 == 0:
                ops, _, _ = tree.generate_operations(rl)
                compare_state(g, o, ops, tree)


def test_hundred_one_ambiguity_and_move_root():
    tree = rd.Tree.from_file(os.path.join(util.DATA, "101.tree"))
    seqs, weights = util.compress(util.read_phylip(os.path.join(util.DATA, "101.phy")))
    gold = util.golden("hundred_one.json")
    for case in gold["cases"]:
        R = case["rate_cats"]
        g, o = pair(tree, seqs, 4, R, weights=weights)
        set_model((g, o), case["subst"], case["freqs"], case["rates"])
        for root in case["roots"]:
            rl = util.find_root(tree, root["near_tips"], root["far_tips"], root["alpha"])
            a, b = util.compute_lh(g, tree, rl), util.compute_lh(o, tree, rl)
            assert util.rel_err(a, b) < LNL_TOL
            assert util.rel_err(a, root["lnl"]) < 1e-10
        # compute_all_root_lh (src/model.cpp:1737-1746): move_root sweep
        first = util.compute_lh(g, tree, tree.root_location(0))
        util.compute_lh(o, tree, tree.root_location(0))
        t2 = rd.Tree.from_file(os.path.join(util.DATA, "101.tree"))
        t2.root_by(t2.root_location(0))
        for rl in tree.roots():
            util.move_root(g, tree, rl)
            a = util.compute_lh_root(g, tree, rl)
            util.move_root(o, t2, rl)
            b = util.compute_lh_root(o, t2, rl)
            assert util.rel_err(a, b) < LNL_TOL
            if case["name"] == "jc":     # test/src/model.cpp:367-387
                assert abs(a - first) < 1e-7 * abs(first)
        g.destroy()
        o.destroy()


def test_deep_tree_scalers_bit_exact():
    gd = util.golden("deep_scaling.json")
    tree = rd.Tree.from_newick(gd["newick"])
    g, o = pair(tree, gd["seqs"], 4, 4)
    set_model((g, o), gd["subst"], gd["freqs"], gd["rates"])
    rl = tree.root_location(0)
    ops, pmi, brl = tree.generate_operations(rl)
    for p in (g, o):
        p.update_prob_matrices(pmi, brl)
        p.update_clvs(ops)
    compare_state(g, o, ops, tree)
    assert o.get_scaler(tree.root_scaler_index()).max() >= 1
    a, pa = g.compute_root_loglikelihood(tree.root_clv_index(), tree.root_scaler_index(), persite=True)
    b, pb = o.compute_root_loglikelihood(tree.root_clv_index(), tree.root_scaler_index(), persite=True)
    assert util.rel_err(a, b) < LNL_TOL
    assert np.allclose(pa, pb, rtol=1e-12)
    # golden lnL for its own rooting
    for cand in tree.roots():
        if sorted(tree.side_tips(cand)) == sorted(gd["near_tips"]):
            assert util.rel_err(util.compute_lh(g, tree, cand.with_ratio(gd["alpha"])), gd["lnl"]) < 1e-10


def test_protein20_generic_path():
    gd = util.golden("protein20.json")
    tree = rd.Tree.from_newick(gd["newick"])
    aa = gd["alphabet"]
    extra = {"X": (1 << 20) - 1, "B": (1 << aa.index("N")) | (1 << aa.index("D"))}
    cmap = util.make_map(aa, extra)
    g, o = pair(tree, gd["seqs"], 20, 4, cmap, cmap)
    set_model((g, o), gd["subst"], gd["freqs"], gd["rates"])
    for k, root in enumerate(gd["roots"]):
        rl = util.find_root(tree, root["near_tips"], root["far_tips"], root["alpha"])
        a, b = util.compute_lh(g, tree, rl), util.compute_lh(o, tree, rl)
        assert util.rel_err(a, b) < LNL_TOL
        assert util.rel_err(a, root["lnl"]) < 1e-10
        if k == 0:
            ops, _, _ = tree.generate_operations(rl)
            compare_state(g, o, ops, tree)


@pytest.mark.parametrize("n,S,K,R,seed", [
    (100, 3000, 4, 4, 11),      # c2 shape, oracle-sized
    (60, 1531, 4, 1, 12),       # ragged site count, no rate het
    (33, 777, 4, 2, 13),
    (40, 513, 4, 8, 14),
    (25, 300, 20, 4, 15),       # c3 shape
    (14, 45, 20, 8, 20),        # 20 states, 8 categories: the wide-workgroup variant of the MFMA kernel
    (9, 33, 20, 16, 21),        # 20 states, 16 categories: beyond the MFMA kernel -> generic kernel
    (12, 1, 4, 4, 16),          # a single site
    (9, 65, 3, 3, 17),          # odd K, odd R -> generic kernels
    (30, 2100, 2, 4, 18),       # binary characters: embedded in the 4-state kernels
    (17, 333, 2, 1, 19),
])
def test_synthetic_vs_oracle(n, S, K, R, seed):
    w = synth.workload(n, S, K, R, seed)
    tree = rd.Tree.from_newick(w["newick"])
    cmap = rd.MAP_NT if K == 4 else util.make_map(w["alphabet"])
    g, o = pair(tree, w["seqs"], K, R, cmap, cmap if K != 4 else ORC_MAP_NT)
    freqs = g.empirical_frequencies()
    assert np.allclose(freqs, o.empirical_frequencies(), rtol=1e-13)
    set_model((g, o), w["subst"], freqs, w["rates"])
    rng = np.random.default_rng(seed)
    for i in rng.choice(tree.root_count(), size=min(4, tree.root_count()), replace=False):
        rl = tree.root_location(int(i)).with_ratio(float(rng.uniform(0.05, 0.95)))
        a, b = util.compute_lh(g, tree, rl), util.compute_lh(o, tree, rl)
        assert math.isfinite(a) and a < 0
        assert util.rel_err(a, b) < LNL_TOL, (a, b)
    ops, _, _ = tree.generate_operations(rl)
    compare_state(g, o, ops, tree)


def test_pattern_weights_and_category_weights():
    w = synth.workload(20, 257, 4, 4, 21)
    tree = rd.Tree.from_newick(w["newick"])
    rng = np.random.default_rng(5)
    weights = rng.integers(1, 9, size=257).astype(np.uint32)
    g, o = pair(tree, w["seqs"], 4, 4, weights=weights)
    cw = rng.dirichlet(np.ones(4))
    set_model((g, o), w["subst"], [0.1, 0.2, 0.3, 0.4], w["rates"], cw)
    rl = tree.root_location(3)
    a, pa = util.compute_lh(g, tree, rl), None
    b = util.compute_lh(o, tree, rl)
    assert util.rel_err(a, b) < LNL_TOL
    # linearity in the pattern weights: doubling every weight doubles lnL
    g.set_pattern_weights(weights * 2)
    assert util.rel_err(util.compute_lh(g, tree, rl), 2 * a) < 1e-13


def test_fused_root_evaluation_matches_unfused():
    """compute_dlh (src/model.cpp:481-519) evaluates alpha and alpha+1e-8."""
    w = synth.workload(30, 1000, 4, 4, 31)
    tree = rd.Tree.from_newick(w["newick"])
    g, o = pair(tree, w["seqs"], 4, 4)
    set_model((g, o), w["subst"], g.empirical_frequencies(), w["rates"])
    rl = tree.root_location(7).with_ratio(0.3)
    util.compute_lh(g, tree, rl)
    util.compute_lh(o, tree, rl)
    op, pmi, brl = tree.generate_derivative_operations(rl)
    alphas = [0.3, 0.3 + 1e-8, 0.0, 1.0]
    l1 = [rl.saved_brlen * a for a in alphas]
    l2 = [rl.saved_brlen * (1 - a) for a in alphas]
    got = g.root_loglikelihood_fused(op, l1, l2)
    want = o.root_loglikelihood_fused(op, l1, l2)
    for a, b in zip(got, want):
        assert util.rel_err(a, b) < LNL_TOL
    # the finite difference the reference forms from them is reproduced too
    d_gpu = (got[1] - got[0]) / 1e-8
    d_orc = (want[1] - want[0]) / 1e-8
    assert abs(d_gpu - d_orc) <= 1e-3 * max(1.0, abs(d_orc))


@pytest.mark.parametrize("R", [1, 4, 8])
def test_fused_root_positions_do_not_depend_on_their_launch(R):
    """optimize_alpha's opening (five positions: the root itself and the two finite-difference
    pairs at alpha = 0 and 1, src/model.cpp:679-700) and its scan levels (eight) ride in one
    launch; a position's value must be the one its own launch would give, bit for bit, the state
    left behind that of the LAST position, and full == root-only (test/src/model.cpp:285-286)
    must hold for every one of them."""
    w = synth.workload(24, 1500, 4, R, 77 + R)
    tree = rd.Tree.from_newick(w["newick"])
    g, o = pair(tree, w["seqs"], 4, R)
    set_model((g, o), w["subst"], g.empirical_frequencies(), w["rates"])
    rl = tree.root_location(11).with_ratio(0.42)
    util.compute_lh(g, tree, rl)
    util.compute_lh(o, tree, rl)
    op, pmi, brl = tree.generate_derivative_operations(rl)
    alphas = [0.42, 1e-8, 0.0, 1.0 - 1e-8, 1.0, 0.125, 0.125 + 1e-8, 0.875]
    l1 = [rl.saved_brlen * a for a in alphas]
    l2 = [rl.saved_brlen * (1 - a) for a in alphas]
    single = np.array([g.root_loglikelihood_fused(op, [x], [y])[0] for x, y in zip(l1, l2)])
    for n in (2, 3, 4, 5, 7, 8):
        got = g.root_loglikelihood_fused(op, l1[:n], l2[:n])
        assert np.array_equal(got, single[:n]), n
        # the state contract: root CLV, scaler and matrices of the LAST position
        clv, sc = g.get_clv(op.parent_clv_index), g.get_scaler(op.parent_scaler_index)
        g.root_loglikelihood_fused(op, l1[n - 1:n], l2[n - 1:n])
        assert np.array_equal(clv, g.get_clv(op.parent_clv_index)) and np.array_equal(sc, g.get_scaler(op.parent_scaler_index))
    want = o.root_loglikelihood_fused(op, l1, l2)
    for a, b in zip(single, want):
        assert util.rel_err(a, b) < LNL_TOL
    # the unfused call sequence gives the same bits (compute_lh_root's three calls)
    g.update_prob_matrices(pmi, [l1[5], l2[5]])
    g.update_clvs([op])
    assert g.compute_root_loglikelihood(op.parent_clv_index, op.parent_scaler_index) == single[5]


def test_error_paths():
    tree = rd.Tree.from_file(os.path.join(util.DATA, "single.tree"))
    g = rd.Partition.for_tree(tree, 4, 4, 1)
    with pytest.raises(rd.RdamdError):
        g.set_tip_states(0, rd.MAP_NT, "AC!T")          # unknown character
    with pytest.raises(rd.RdamdError):
        g.set_tip_states(99, rd.MAP_NT, "ACGT")
    with pytest.raises(rd.RdamdError):
        g.update_prob_matrices([99], [0.1])
    with pytest.raises(rd.RdamdError):
        g.update_prob_matrices([0], [-1.0])
    with pytest.raises(rd.RdamdError):
        g.update_prob_matrices([0], [float("nan")])
    with pytest.raises(rd.RdamdError):
        g.update_invariant_sites_proportion(0, 0.2)
    with pytest.raises(rd.RdamdError):
        rd.Partition(4, 6, 100, 4, 1, 6, 1, 6)          # states > 64


@pytest.fixture(scope="module")
def c2_full():
    """BASELINE config c2 at full size: 100 taxa x 50,000 sites, UNREST + G4."""
    w = synth.workload(100, 50000, 4, 4, 0xD166E5 + 1)
    tree = rd.Tree.from_newick(w["newick"])
    g = rd.Partition.for_tree(tree, 4, 50000, 4)
    util.load_tips(g, tree, w["seqs"], rd.MAP_NT)
    g.set_category_rates(w["rates"])
    return w, tree, g


def test_c2_full_size_properties(c2_full):
    w, tree, g = c2_full
    g.set_subst_params(0, w["subst"])
    g.set_frequencies(0, g.empirical_frequencies())
    rls = [tree.root_location(i) for i in (0, 50, 196)]
    for rl in rls:
        a = util.compute_lh(g, tree, rl)
        assert math.isfinite(a) and a < 0
        assert a == util.compute_lh(g, tree, rl)                      # determinism
        assert util.rel_err(util.compute_lh_root(g, tree, rl), a) < 1e-13
    # per-site values sum to the total (checksum of checksums)
    a, ps = (lambda rl: (util.compute_lh(g, tree, rl),
                         g.compute_root_loglikelihood(tree.root_clv_index(),
                                                      tree.root_scaler_index(), persite=True)[1]))(rls[1])
    assert util.rel_err(float(np.sum(ps)), a) < 1e-12
    # pulley principle under a reversible model: all 197 rootings agree
    g.set_subst_params(0, [1.0] * 12)
    g.set_frequencies(0, [0.25] * 4)
    base = util.compute_lh(g, tree, tree.root_location(0))
    for rl in tree.roots():
        util.move_root(g, tree, rl)
        assert abs(util.compute_lh_root(g, tree, rl) - base) < 1e-9 * abs(base)


def test_c2_full_site_slice_additivity(c2_full):
    """lnL is a sum over sites: the first 20k + the remaining 30k sites evaluated
    in separate partitions (the multi-GPU site sharding) add up to the whole."""
    w, tree, g = c2_full
    freqs = [0.22, 0.27, 0.24, 0.27]
    rl = tree.root_location(17).with_ratio(0.4)
    g.set_subst_params(0, w["subst"])
    g.set_frequencies(0, freqs)
    whole = util.compute_lh(g, tree, rl)
    total = 0.0
    for lo, hi in ((0, 20000), (20000, 50000)):
        part = rd.Partition.for_tree(tree, 4, hi - lo, 4)
        util.load_tips(part, tree, {k: v[lo:hi] for k, v in w["seqs"].items()}, rd.MAP_NT)
        set_model((part,), w["subst"], freqs, w["rates"])
        total += util.compute_lh(part, tree, rl)
        part.destroy()
    assert util.rel_err(total, whole) < 1e-12


# ---------------------------------------------------------------------------
# fused batched evaluator (rdamd_evaluate_batch) vs the oracle
# ---------------------------------------------------------------------------
def _oracle_eval(o, tree, rl, subst, freqs, rates, weights=None):
    o.set_subst_params(0, subst)
    o.set_frequencies(0, freqs)
    o.set_category_rates(rates)
    if weights is not None:
        o.set_category_weights(weights)
    return util.compute_lh(o, tree, rl)


@pytest.mark.parametrize("n,S,R,seed", [(100, 2000, 4, 41), (37, 1000, 1, 42),
                                        (64, 333, 2, 43), (16, 65, 8, 44), (5, 7, 3, 45)])
def test_fused_batch_vs_oracle(n, S, R, seed):
    w = synth.workload(n, S, 4, R, seed)
    tree = rd.Tree.from_newick(w["newick"])
    rng = np.random.default_rng(seed)
    weights = rng.integers(1, 4, size=S).astype(np.uint32)
    g, o = pair(tree, w["seqs"], 4, R, weights=weights)
    g.set_category_rates(w["rates"])
    picks = rng.choice(tree.root_count(), size=min(6, tree.root_count()), replace=False)
    rls = [tree.root_location(int(i)).with_ratio(float(rng.uniform(0.02, 0.98))) for i in picks]
    scheds = [g.schedule(*tree.generate_operations(rl)) for rl in rls]
    assert all(1 <= s.stack_depth() <= 12 for s in scheds)
    subst = rng.uniform(1e-4, 1.0, (len(rls), 12))
    freqs = rng.dirichlet(np.ones(4) * 5, len(rls))
    rates = np.array([rd.compute_gamma_cats(a, R) for a in rng.uniform(0.3, 3.0, len(rls))])
    cw = rng.dirichlet(np.ones(R) * 3, len(rls))
    got = g.evaluate_batch(scheds, subst, freqs, rates, cw)
    for j, rl in enumerate(rls):
        want = _oracle_eval(o, tree, rl, subst[j], freqs[j], rates[j], cw[j])
        assert util.rel_err(got[j], want) < LNL_TOL, (j, got[j], want)
    # repeat call: bit-identical (test/src/model.cpp:73)
    assert np.array_equal(got, g.evaluate_batch(scheds, subst, freqs, rates, cw))
    # defaults: partition's own rates/weights
    got2 = g.evaluate_batch(scheds[:2], subst[:2], freqs[:2])
    for j in range(2):
        o.set_category_weights([1.0 / R] * R)
        want = _oracle_eval(o, tree, rls[j], subst[j], freqs[j], w["rates"])
        assert util.rel_err(got2[j], want) < LNL_TOL


@pytest.mark.parametrize("n,S,R,seed", [(40, 300, 4, 46), (25, 77, 2, 47), (12, 16, 1, 48), (70, 130, 3, 49),
                                         (40, 200, 8, 146), (21, 50, 5, 147), (33, 129, 7, 148)])
def test_fused_batch_20_states_vs_oracle(n, S, R, seed):
    """the 20-state fused evaluator (kernels_fused_k20.hip): batched jobs with
    their own parameters, rates and category weights, ragged last tile.  One to four rate
    categories run 256-thread workgroups (a wave per category), five to eight the 512-thread
    instantiation (round 6: `rd --rate-cats N` takes any N, src/main.cpp:256-266)."""
    w = synth.workload(n, S, 20, R, seed)
    tree = rd.Tree.from_newick(w["newick"])
    rng = np.random.default_rng(seed)
    weights = rng.integers(1, 4, size=S).astype(np.uint32)
    cmap = util.make_map(w["alphabet"])
    g, o = pair(tree, w["seqs"], 20, R, cmap, cmap, weights=weights)
    g.set_category_rates(w["rates"])
    picks = rng.choice(tree.root_count(), size=min(5, tree.root_count()), replace=False)
    rls = [tree.root_location(int(i)).with_ratio(float(rng.uniform(0.02, 0.98))) for i in picks]
    scheds = [g.schedule(*tree.generate_operations(rl)) for rl in rls]
    subst = rng.uniform(1e-3, 1.0, (len(rls), 380))
    freqs = rng.dirichlet(np.ones(20) * 5, len(rls))
    rates = np.array([rd.compute_gamma_cats(a, R) for a in rng.uniform(0.3, 3.0, len(rls))])
    cw = rng.dirichlet(np.ones(R) * 3, len(rls))
    got = g.evaluate_batch(scheds, subst, freqs, rates, cw)
    for j, rl in enumerate(rls):
        want = _oracle_eval(o, tree, rl, subst[j], freqs[j], rates[j], cw[j])
        assert util.rel_err(got[j], want) < LNL_TOL, (j, got[j], want)
    assert np.array_equal(got, g.evaluate_batch(scheds, subst, freqs, rates, cw))
    # and it agrees with the materialising path of the same partition
    set_model((g,), subst[0], freqs[0], rates[0], cw[0])
    assert util.rel_err(util.compute_lh(g, tree, rls[0]), got[0]) < 1e-11
    g.destroy()
    o.destroy()


def test_fused_matches_unfused_and_leaves_partition_state_alone():
    w = synth.workload(50, 1500, 4, 4, 51)
    tree = rd.Tree.from_newick(w["newick"])
    g, o = pair(tree, w["seqs"], 4, 4)
    freqs = g.empirical_frequencies()
    set_model((g, o), w["subst"], freqs, w["rates"])
    rl0 = tree.root_location(5)
    base = util.compute_lh(g, tree, rl0)                  # partition state = rooting 5
    rl = tree.root_location(20).with_ratio(0.7)
    sched = g.schedule(*tree.generate_operations(rl))
    other = np.array(w["subst"])[::-1].copy()      # (a common factor would cancel in Q)
    fused = g.evaluate_batch([sched, sched], [w["subst"], other], [freqs, freqs])
    tree.root_by(rl0)
    assert util.compute_lh_root(g, tree, rl0) == base     # state untouched by the batch
    assert util.rel_err(fused[0], util.compute_lh(g, tree, rl)) < 1e-13
    assert fused[0] != fused[1]


def test_fused_deep_tree_per_rate_scaling():
    """161-taxon caterpillar: scaling fires; the per-rate-count form must give
    the same lnL as the per-site rule of the oracle."""
    gd = util.golden("deep_scaling.json")
    tree = rd.Tree.from_newick(gd["newick"])
    g, o = pair(tree, gd["seqs"], 4, 4)
    g.set_category_rates(gd["rates"])
    for i in (0, 100, 250, tree.root_count() - 1):
        rl = tree.root_location(i).with_ratio(0.31)
        sched = g.schedule(*tree.generate_operations(rl))
        got = g.evaluate_batch([sched], [gd["subst"]], [gd["freqs"]])[0]
        want = _oracle_eval(o, tree, rl, gd["subst"], gd["freqs"], gd["rates"])
        assert util.rel_err(got, want) < LNL_TOL
        assert sched.stack_depth() <= 2      # a caterpillar needs (almost) no stack


def test_fused_rate_categories_four_rescales_apart():
    """VERDICT r2 item 9.  The fused evaluators rescale per (site, rate) and align the rate
    terms at the root (kernels_fused.hip), the reference rule rescales per site:
    the two can only differ where one category needs >= 4 more rescales than another -- there
    the aligned term is 2^-1024 of the leading one and is dropped (pow2_neg256), while the
    per-site rule has let the same category underflow to nothing.  Constructed case: the
    161-taxon caterpillar, pi_A = 0.01, rate categories 1e-6 ... 1000.  On a constant column
    of A the slow category keeps its CLV near 1 (never rescaled) while the fast ones shrink by
    ~0.01 per tip (2^-1060 at the root: four rescales); on a column that alternates between
    states the fast categories lead and the slow one pays ~1e-8 per tip.  Both rules must
    lose the same category and agree on lnL."""
    gd = util.golden("deep_scaling.json")
    tree = rd.Tree.from_newick(gd["newick"])
    names = sorted(gd["seqs"])
    rng = np.random.default_rng(9)
    cols = ["A" * len(names), "C" * len(names), "".join("ACGT"[i % 4] for i in range(len(names))),
            "".join("AC"[i % 2] for i in range(len(names)))]
    cols += ["".join(rng.choice(list("ACGT"), p=[0.7, 0.1, 0.1, 0.1]) for _ in names) for _ in range(60)]
    seqs = {k: "".join(c[i] for c in cols) for i, k in enumerate(names)}
    freqs = [0.01, 0.33, 0.33, 0.33]
    rates = [1e-6, 0.01, 10.0, 1000.0]
    g, o = pair(tree, seqs, 4, 4)
    S = len(cols)
    gr = rd.Partition.for_tree(tree, 4, S, 4, attributes=rd.ATTRIB_SITE_REPEATS)
    util.load_tips(gr, tree, seqs, rd.MAP_NT)
    for i in (0, 160, tree.root_count() - 1):
        rl = tree.root_location(i).with_ratio(0.5)
        ops = tree.generate_operations(rl)
        want = _oracle_eval(o, tree, rl, gd["subst"], freqs, rates)
        assert np.isfinite(want)
        for part in (g, gr):
            got = part.evaluate_batch([part.schedule(*ops)], [gd["subst"]], [freqs], [rates])[0]
            assert util.rel_err(got, want) < LNL_TOL, (i, got, want)
        # the per-site rule on the device (materialising path) as well
        set_model((g,), gd["subst"], freqs, rates)
        assert util.rel_err(util.compute_lh(g, tree, rl), want) < LNL_TOL
    # the construction does what it says: on the constant column the categories' scaler
    # needs are >= 4 apart (oracle CLV of the root, per rate: the slow category is O(1),
    # the fast ones have underflowed to zero under the per-site rule)
    rl = tree.root_location(160).with_ratio(0.5)
    _oracle_eval(o, tree, rl, gd["subst"], freqs, rates)
    root = o.get_clv(tree.root_clv_index())
    assert root[0, 0].max() > 1e-3 and root[0, 3].max() < 2.0 ** -1000
    for p in (g, gr, o):
        p.destroy()


def test_fused_schedule_validation():
    w = synth.workload(12, 64, 4, 1, 61)
    tree = rd.Tree.from_newick(w["newick"])
    g = rd.Partition.for_tree(tree, 4, 64, 1)
    ops, pmi, brl = tree.generate_operations(tree.root_location(0))
    with pytest.raises(rd.RdamdError):      # incomplete traversal
        g.schedule([ops[i] for i in range(1, len(ops))], pmi, brl)
    with pytest.raises(rd.RdamdError):
        g.schedule(ops, pmi, -brl)
    p20 = rd.Partition(4, 6, 20, 8, 1, 6, 1, 6)
    with pytest.raises(rd.RdamdError):      # 4-state only
        p20.schedule(ops, pmi, brl)


def test_c2_full_size_fused_properties(c2_full):
    w, tree, g = c2_full
    rng = np.random.default_rng(99)
    freqs = np.array(g.empirical_frequencies())
    rls = [tree.root_location(int(i)) for i in rng.choice(197, 24, replace=False)]
    scheds = [g.schedule(*tree.generate_operations(rl)) for rl in rls]
    subst = rng.uniform(1e-4, 1.0, (24, 12))
    got = g.evaluate_batch(scheds, subst, np.tile(freqs, (24, 1)))
    assert np.all(np.isfinite(got)) and np.all(got < 0)
    assert np.array_equal(got, g.evaluate_batch(scheds, subst, np.tile(freqs, (24, 1))))
    # spot-check three jobs against the unfused HIP path (itself oracle-checked)
    for j in (0, 11, 23):
        g.set_subst_params(0, subst[j])
        g.set_frequencies(0, freqs)
        assert util.rel_err(got[j], util.compute_lh(g, tree, rls[j])) < 1e-12
    # reversible model: all roots agree (pulley principle), one batch of 197 jobs
    all_s = [g.schedule(*tree.generate_operations(rl)) for rl in tree.roots()]
    jc = g.evaluate_batch(all_s, np.ones((197, 12)), np.full((197, 4), 0.25))
    assert np.max(np.abs(jc - jc[0])) < 1e-9 * abs(jc[0])


# ---------------------------------------------------------------------------
# 20-state MFMA kernel (BASELINE config c3 shape)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("R,S", [(1, 47), (3, 33), (4, 130)])
def test_protein_mfma_rescaling_and_ragged_tiles(R, S):
    """Deep 20-state caterpillar: the per-site 2^256 rule must fire and the
    scalers must equal the oracle's bit for bit; S is not a multiple of the
    32-site tile, R covers 1 / odd / 4."""
    rng = np.random.default_rng(200 + R)
    aa = synth.AA
    n = 140
    names = ["q%03d" % i for i in range(n)]
    nw = names[0]
    for i in range(1, n - 2):
        nw = "(%s:0.8,%s:0.6)" % (nw, names[i])
    nw = "(%s:0.8,%s:0.6,%s:0.7);" % (nw, names[n - 2], names[n - 1])
    seqs = {k: "".join(rng.choice(list(aa + "X"), S)) for k in names}
    tree = rd.Tree.from_newick(nw)
    cmap = util.make_map(aa, {"X": (1 << 20) - 1})
    g, o = pair(tree, seqs, 20, R, cmap, cmap)
    subst = rng.uniform(1e-4, 1, 380)
    freqs = rng.dirichlet(np.ones(20) * 6)
    rates = rd.compute_gamma_cats(0.5, R) if R > 1 else [1.0]
    set_model((g, o), subst, freqs, rates)
    rl = tree.root_location(3).with_ratio(0.4)
    ops, pmi, brl = tree.generate_operations(rl)
    for p in (g, o):
        p.update_prob_matrices(pmi, brl)
        p.update_clvs(ops)
    assert o.get_scaler(tree.root_scaler_index()).max() >= 1
    compare_state(g, o, ops, tree)
    a = g.compute_root_loglikelihood(tree.root_clv_index(), tree.root_scaler_index())
    b = o.compute_root_loglikelihood(tree.root_clv_index(), tree.root_scaler_index())
    assert util.rel_err(a, b) < LNL_TOL
    # move_root + root-only path on the MFMA kernel too
    t2 = rd.Tree.from_newick(nw)
    t2.root_by(rl)
    for i in (0, 50, 200):
        r2 = tree.root_location(i)
        util.move_root(g, tree, r2)
        util.move_root(o, t2, r2)
        assert util.rel_err(util.compute_lh_root(g, tree, r2), util.compute_lh_root(o, t2, r2)) < LNL_TOL


# ---------------------------------------------------------------------------
# edge cases: smallest tree, all-gap columns, empty calls, a 1000-taxon tree
# ---------------------------------------------------------------------------
def test_three_taxon_tree_and_all_gap_columns():
    tree = rd.Tree.from_newick("(a:0.1,b:0.2,c:0.3);")
    assert tree.root_count() == 3
    seqs = {"a": "AC-N?", "b": "AG-NX", "c": "TT-N-"}      # columns 2..4 carry no data
    g, o = pair(tree, seqs, 4, 4)
    set_model((g, o), PARAMS[1], [0.1, 0.2, 0.3, 0.4], rd.compute_gamma_cats(0.8, 4))
    for rl in tree.roots():
        rl = rl.with_ratio(0.25)
        a, b = util.compute_lh(g, tree, rl), util.compute_lh(o, tree, rl)
        assert util.rel_err(a, b) < LNL_TOL
        _, ps = g.compute_root_loglikelihood(tree.root_clv_index(), tree.root_scaler_index(),
                                             persite=True)
        assert np.allclose(ps[2:], 0.0, atol=1e-14)         # an all-gap site has likelihood 1
        sched = g.schedule(*tree.generate_operations(rl))
        f = g.evaluate_batch([sched], [PARAMS[1]], [[0.1, 0.2, 0.3, 0.4]])[0]
        assert util.rel_err(f, b) < LNL_TOL


def test_empty_calls_are_no_ops():
    tree = rd.Tree.from_file(os.path.join(util.DATA, "single.tree"))
    g = rd.Partition.for_tree(tree, 4, 3, 1)
    g.update_prob_matrices([], [])
    g.update_clvs([])
    assert len(g.evaluate_batch([], np.zeros((0, 12)), np.zeros((0, 4)))) == 0


def test_thousand_taxon_tree_vs_oracle():
    """BASELINE config c5's tree size (1000 taxa, 1997 candidate roots) at an
    oracle-sized site count: deep stacks in the fused kernel, long operation
    lists in the traversal kernel."""
    w = synth.workload(1000, 1200, 4, 4, 71)
    tree = rd.Tree.from_newick(w["newick"])
    assert tree.root_count() == 1997
    g, o = pair(tree, w["seqs"], 4, 4)
    freqs = g.empirical_frequencies()
    set_model((g, o), w["subst"], freqs, w["rates"])
    rng = np.random.default_rng(71)
    rls = [tree.root_location(int(i)).with_ratio(float(a))
           for i, a in zip(rng.choice(1997, 3, replace=False), rng.uniform(0.1, 0.9, 3))]
    want = [util.compute_lh(o, tree, rl) for rl in rls]
    got = [util.compute_lh(g, tree, rl) for rl in rls]
    scheds = [g.schedule(*tree.generate_operations(rl)) for rl in rls]
    fused = g.evaluate_batch(scheds, [w["subst"]] * 3, [freqs] * 3)
    for a, b, c in zip(got, want, fused):
        assert util.rel_err(a, b) < LNL_TOL
        assert util.rel_err(c, b) < LNL_TOL
    assert max(s.stack_depth() for s in scheds) <= 10
    for st in (s.stats() for s in scheds):   # (no 64-row tables here: the register slot by stack level, the rest on LDS levels)
        assert 50 < st["parks"] and st["parks"] > st["parks_in_registers"] > 0 and st["parks_in_lds_slot"] == 0, st
    ops, _, _ = tree.generate_operations(rls[2])
    compare_state(g, o, [ops[i] for i in (0, 500, 998)], tree)


def _balanced_newick(n_tips, rng):
    nodes = ["t%d:%.5f" % (i, rng.uniform(0.02, 0.3)) for i in range(n_tips)]
    while len(nodes) > 3:
        nodes = ["(%s,%s):%.5f" % (nodes[i], nodes[i + 1], rng.uniform(0.02, 0.3))
                 for i in range(0, len(nodes), 2)]
    return "(" + ",".join(nodes) + ");"


@pytest.mark.parametrize("n_tips,repeats,sites", [(48, 0, 700), (192, 0, 700), (192, 64, 700), (192, 16, 130),
                                                 (768, 0, 300)])
def test_balanced_trees_deep_stacks(n_tips, repeats, sites):
    """Perfectly balanced trees need the deepest stacks a tree of their size can ask for: with
    site repeats (64-row kernels) one register level, one LDS slot and the rest in the waves'
    private segment up to eight levels in all (kernels_fused.hip, SP), two register levels and
    an all-LDS stack beyond and without repeats.
    Random (unrelated) sequences make the rescaling fire on the way up, so parked rescale
    counts travel through every kind of level.  One and two sites per lane (launch size),
    plain and folded programs."""
    rng = np.random.default_rng(n_tips + repeats)
    tree = rd.Tree.from_newick(_balanced_newick(n_tips, rng))
    seqs = {"t%d" % i: "".join(rng.choice(list("ACGT"), sites)) for i in range(n_tips)}
    w = synth.workload(8, 10, 4, 4, 5, simulate_seqs=False)
    g = rd.Partition.for_tree(tree, 4, sites, 4, attributes=rd.ATTRIB_SITE_REPEATS if repeats else 0)
    if repeats:
        g.set_site_repeats(repeats)
    o = OraclePartition.for_tree(tree, 4, sites, 4)
    util.load_tips(g, tree, seqs, rd.MAP_NT)
    util.load_tips(o, tree, seqs, ORC_MAP_NT)
    freqs = [0.22, 0.31, 0.2, 0.27]
    set_model((g, o), w["subst"], freqs, w["rates"])
    picks = [int(i) for i in rng.choice(tree.root_count(), 3, replace=False)]
    rls = [tree.root_location(i).with_ratio(float(rng.uniform(0.1, 0.9))) for i in picks]
    want = [util.compute_lh(o, tree, rl) for rl in rls]
    scheds = [g.schedule(*tree.generate_operations(rl)) for rl in rls]
    depth = max(s.stack_depth() for s in scheds)
    # (in-memory levels: behind one register level with repeats -- the 64-row kernels have
    # private-segment levels --, behind two from four levels on without)
    assert depth >= (3 if n_tips >= 192 and repeats != 16 else 2), depth
    small = g.evaluate_batch(scheds, [w["subst"]] * 3, [freqs] * 3)                 # one site per lane
    big = g.evaluate_batch(scheds * 20, [w["subst"]] * 60, [freqs] * 60)            # two
    for a, b in zip(small, want):
        assert util.rel_err(a, b) < LNL_TOL
    assert np.array_equal(big, np.tile(small, 20))
    g.destroy()
    o.destroy()


def test_binary_data_on_the_four_state_kernels():
    """states == 2 (`rd --states 2`): the partition runs on the 4-state kernels
    with two inert states.  Everything the caller sees keeps its 2-state shape
    and agrees with the oracle's native 2-state arithmetic: P-matrices, CLVs,
    scalers, lnL, the fused root step and the fused batch."""
    w = synth.workload(40, 1500, 2, 4, 77)
    tree = rd.Tree.from_newick(w["newick"])
    cmap = util.make_map(w["alphabet"], {"-": 3, "?": 3})
    seqs = {k: v[:700] + "-" * 5 + v[705:] for k, v in w["seqs"].items()}     # gaps = both states
    g, o = pair(tree, seqs, 2, 4, cmap, cmap)
    assert g.states == 2 and rd.lib.rdamd_partition_states(g.handle) == 2
    freqs = g.empirical_frequencies()
    assert len(freqs) == 2 and np.allclose(freqs, o.empirical_frequencies(), rtol=1e-13)
    subst = [0.7, 1.9]
    set_model((g, o), subst, freqs, w["rates"])
    assert np.allclose(g.subst_params(0), subst)
    rl = tree.root_location(21).with_ratio(0.35)
    ops, pmi, brl = tree.generate_operations(rl)
    for p in (g, o):
        p.update_prob_matrices(pmi, brl)
        p.update_clvs(ops)
    for m in (0, 5, len(pmi) - 1):
        a = g.get_pmatrix(int(pmi[m]))
        assert a.shape == (4, 2, 2) and np.allclose(a, o.get_pmatrix(int(pmi[m])), atol=1e-13)
        assert np.allclose(a.sum(axis=2), 1.0, atol=1e-13)
    compare_state(g, o, ops, tree)
    assert g.get_clv(ops[-1].parent_clv_index).shape == (1500, 4, 2)
    a = g.compute_root_loglikelihood(tree.root_clv_index(), tree.root_scaler_index())
    b = o.compute_root_loglikelihood(tree.root_clv_index(), tree.root_scaler_index())
    assert util.rel_err(a, b) < LNL_TOL
    assert util.rel_err(util.compute_lh_root(g, tree, rl), b) < LNL_TOL
    # fused batch: [n][2] rates and [n][2] frequencies
    rls = [tree.root_location(i).with_ratio(0.5) for i in (0, 33, 76)]
    scheds = [g.schedule(*tree.generate_operations(r)) for r in rls]
    subs = [[0.7, 1.9], [1.0, 1.0], [0.05, 3.0]]
    frs = [freqs, [0.5, 0.5], [0.2, 0.8]]
    got = g.evaluate_batch(scheds, subs, frs)
    for r, s_, f_, val in zip(rls, subs, frs, got):
        o.set_subst_params(0, s_)
        o.set_frequencies(0, f_)
        assert util.rel_err(val, util.compute_lh(o, tree, r)) < LNL_TOL
    g.destroy()
    o.destroy()


def test_binary2_goldens_on_the_gpu():
    """the SciPy golden for 2-state data, through the embedded 4-state kernels:
    materialising path, fused root step and fused batch."""
    gd = util.golden("binary2.json")
    tree = rd.Tree.from_newick(gd["newick"])
    cmap = util.make_map("01", {"-": 3, "?": 3})
    nsites = len(next(iter(gd["seqs"].values())))
    g = rd.Partition.for_tree(tree, 2, nsites, 4)
    util.load_tips(g, tree, gd["seqs"], cmap)
    set_model((g,), gd["subst"], gd["freqs"], gd["rates"])
    rls = [util.find_root(tree, r["near_tips"], r["far_tips"], r["alpha"]) for r in gd["roots"]]
    for rl, root in zip(rls, gd["roots"]):
        assert util.rel_err(util.compute_lh(g, tree, rl), root["lnl"]) < 1e-10
        assert util.rel_err(util.compute_lh_root(g, tree, rl), root["lnl"]) < 1e-10
    scheds = [g.schedule(*tree.generate_operations(rl)) for rl in rls]
    fused = g.evaluate_batch(scheds, [gd["subst"]] * len(rls), [gd["freqs"]] * len(rls))
    for val, root in zip(fused, gd["roots"]):
        assert util.rel_err(val, root["lnl"]) < 1e-10
    g.destroy()


def test_many_sites_few_parking_slots_cross_check():
    """300 taxa x 120 000 sites x 4 rates: 480 k (site, rate) lanes leave the
    traversal kernel ONE LDS parking slot per lane, so most older siblings are
    read back from HBM -- the route small cases never take.  Too big for the
    oracle; the two independent HIP paths (materialised CLVs + root kernel vs
    the fused evaluator) must agree, and so must a site slice."""
    w = synth.workload(300, 120000, 4, 4, 91)
    tree = rd.Tree.from_newick(w["newick"])
    g = rd.Partition.for_tree(tree, 4, 120000, 4)
    util.load_tips(g, tree, w["seqs"], rd.MAP_NT)
    freqs = g.empirical_frequencies()
    set_model((g,), w["subst"], freqs, w["rates"])
    rls = [tree.root_location(i).with_ratio(a) for i, a in ((3, 0.2), (401, 0.7))]
    full = [util.compute_lh(g, tree, rl) for rl in rls]
    scheds = [g.schedule(*tree.generate_operations(rl)) for rl in rls]
    fused = g.evaluate_batch(scheds, [w["subst"]] * 2, [freqs] * 2)
    for a, b in zip(full, fused):
        assert math.isfinite(a) and util.rel_err(a, b) < 1e-12
    # a 3 000-site slice of the same alignment, small enough for every slot
    sub = {k: v[50000:53000] for k, v in w["seqs"].items()}
    h = rd.Partition.for_tree(tree, 4, 3000, 4)
    util.load_tips(h, tree, sub, rd.MAP_NT)
    set_model((h,), w["subst"], freqs, w["rates"])
    _, persite = g.compute_root_loglikelihood(tree.root_clv_index(), tree.root_scaler_index(),
                                              persite=True)      # state left by rls[1]
    part = util.compute_lh(h, tree, rls[1])
    assert util.rel_err(part, float(np.sum(persite[50000:53000]))) < 1e-12
    g.destroy()
    h.destroy()


@pytest.mark.parametrize("n,S,R,seed", [(100, 3000, 4, 81), (400, 1500, 4, 82), (130, 19436, 4, 83), (64, 700, 1, 84),
                                        (90, 50000, 4, 85)])
def test_cut_operation_lists_leave_the_whole_lists_state(n, S, R, seed):
    """A full traversal of a 4-state partition whose one row of blocks leaves the device empty is
    cut into independent subtrees that run side by side, level by level (k20_split.hpp
    list_levels, kernels_clv.hip).  Every CLV and every scaler must be what the oracle's plain
    loop leaves -- for the tree's post-order list (cut: several launches), for a short partial
    list over that state (whole: one launch), and for the list re-run after a root move."""
    w = synth.workload(n, S, 4, R, seed)
    tree = rd.Tree.from_newick(w["newick"])
    g, o = pair(tree, w["seqs"], 4, R)
    set_model((g, o), w["subst"], g.empirical_frequencies(), w["rates"])
    rng = np.random.default_rng(seed)
    for k, i in enumerate(rng.choice(tree.root_count(), size=3 if S < 10000 else 1, replace=False)):
        rl = tree.root_location(int(i)).with_ratio(float(rng.uniform(0.05, 0.95)))
        ops, pmi, brl = tree.generate_operations(rl)
        for p in (g, o):
            p.update_prob_matrices(pmi, brl)
            p.update_clvs(ops)
        assert g.update_clvs_launches() >= 2, (len(ops), g.update_clvs_launches())
        compare_state(g, o, ops, tree)
        root = ops[-1]
        a = g.compute_root_loglikelihood(root.parent_clv_index, root.parent_scaler_index, [0] * R)
        b = o.compute_root_loglikelihood(root.parent_clv_index, root.parent_scaler_index, [0] * R)
        assert util.rel_err(a, b) < LNL_TOL
        if k == 0:   # the last few operations again: too short to cut, children left by the call above
            tail = ops[-5:]
            for p in (g, o):
                p.update_clvs(tail)
            assert g.update_clvs_launches() == 1
            compare_state(g, o, tail, tree)


@pytest.mark.parametrize("n,S,R,seed,K", [(64, 700, 4, 71, 4), (150, 300, 2, 72, 4), (40, 9000, 4, 73, 4),
                                          (48, 333, 4, 74, 20), (30, 100, 3, 75, 20)])
def test_arbitrary_operation_orders(n, S, R, seed, K):
    """rdamd_update_clvs takes ANY valid list (corax_update_clvs contract), not
    only the post-order the tree emits.  The 4-state kernel forwards children
    through registers and LDS parking slots depending on the order, so feed it
    orders that stress each route: level order (every sibling waits long ->
    slots overflow, values evicted to HBM), reversed-sibling order, the same
    CLV as both children, a child read with a different scaler index than it
    was written with, and a partial list over CLVs left by an earlier call.
    The 20-state kernel forwards only the parent of the operation just before
    and asks for the next operation's operands before it stores: the host
    cuts the launch where that would read stale data (case 3)."""
    w = synth.workload(n, S, K, R, seed)
    tree = rd.Tree.from_newick(w["newick"])
    cmap = rd.MAP_NT if K == 4 else util.make_map(w["alphabet"])
    g, o = pair(tree, w["seqs"], K, R, cmap, cmap if K != 4 else ORC_MAP_NT)
    set_model((g, o), w["subst"], g.empirical_frequencies(), w["rates"])
    rl = tree.root_location(seed % tree.root_count())
    ops, pmi, brl = tree.generate_operations(rl)
    ops = [rd.Operation(*op.astuple()) for op in ops]
    for p in (g, o):
        p.update_prob_matrices(pmi, brl)

    def run(op_list):
        for p in (g, o):
            p.update_clvs(op_list)
        compare_state(g, o, op_list, tree)

    # 1. level order: an operation runs as soon as both children exist
    done = set(range(n))
    pending, level_order = list(ops), []
    while pending:
        ready = [op for op in pending if op.child1_clv_index in done and op.child2_clv_index in done]
        assert ready
        level_order += ready
        done |= {op.parent_clv_index for op in ready}
        pending = [op for op in pending if op.parent_clv_index not in done]
    run(level_order)
    a = g.compute_root_loglikelihood(tree.root_clv_index(), tree.root_scaler_index())
    b = o.compute_root_loglikelihood(tree.root_clv_index(), tree.root_scaler_index())
    assert util.rel_err(a, b) < LNL_TOL
    # 2. deepest-first order (reverse of level order within the constraints)
    rng = np.random.default_rng(seed)
    done, pending, rand_order = set(range(n)), list(ops), []
    while pending:
        ready = [op for op in pending if op.child1_clv_index in done and op.child2_clv_index in done]
        pick = ready[int(rng.integers(len(ready)))]
        rand_order.append(pick)
        done.add(pick.parent_clv_index)
        pending.remove(pick)
    run(rand_order)
    # 3. extra operations on spare buffers (for_tree allocates 2n-2 inner CLVs,
    #    the traversal uses n-1): same CLV twice straight after it is written,
    #    and a child consumed with scaler -1 although it was written with one
    spare_clv, spare_sc = 2 * n - 1, n - 1
    last = ops[-1]
    twice = rd.Operation(spare_clv, spare_sc, last.parent_clv_index, last.child1_matrix_index,
                         last.parent_scaler_index, last.parent_clv_index,
                         last.child2_matrix_index, last.parent_scaler_index)
    mismatch = rd.Operation(spare_clv + 1, spare_sc + 1, spare_clv, last.child1_matrix_index, -1,
                            0, last.child2_matrix_index, -1)
    older = rd.Operation(spare_clv + 2, -1, spare_clv + 1, last.child2_matrix_index, spare_sc + 1,
                         ops[0].parent_clv_index, last.child1_matrix_index,
                         ops[0].parent_scaler_index)
    run(list(ops) + [twice, mismatch, older])
    # 4. a partial list reading CLVs a previous call left in HBM
    run([older, twice])
    g.destroy()
    o.destroy()


def test_zero_site_partition():
    tree = rd.Tree.from_file(os.path.join(util.DATA, "single.tree"))
    g = rd.Partition.for_tree(tree, 4, 0, 4)
    rl = tree.root_location(0)
    assert util.compute_lh(g, tree, rl) == 0.0
    sched = g.schedule(*tree.generate_operations(rl))
    assert g.evaluate_batch([sched], [[1.0] * 12], [[0.25] * 4])[0] == 0.0


def test_fused_tip_tip_steps_keep_the_rescale_test_when_tables_are_tiny():
    """The fused 4-state evaluator skips the 2^256 rescale test on tip-tip steps only
    while every non-zero tip-table entry of the job is >= 2^-128 (then a product of two
    rows is 0 or >= 2^-256).  Branches of length 1e-60 make off-diagonal P entries
    ~1e-61 < 2^-128: the job is flagged and must still agree with the oracle; an
    ordinary tree goes through the unflagged path."""
    rng = np.random.default_rng(321)
    names = ["a", "b", "c", "d", "e", "f"]
    seqs = {k: "".join(rng.choice(list("ACGT"), 300)) for k in names}
    for t in (1e-60, 0.03):
        tree = rd.Tree.from_newick("((a:%g,b:%g):0.1,(c:%g,d:%g):0.2,(e:0.05,f:0.07):0.1);" % (t, t, t, t))
        g, o = pair(tree, seqs, 4, 4)
        rates = rd.compute_gamma_cats(0.7, 4)
        g.set_category_rates(rates)
        freqs = [0.2, 0.3, 0.25, 0.25]
        rls = [tree.root_location(i).with_ratio(0.4) for i in range(tree.root_count())]
        scheds = [g.schedule(*tree.generate_operations(rl)) for rl in rls]
        got = g.evaluate_batch(scheds, [PARAMS[1]] * len(rls), [freqs] * len(rls))
        want = [_oracle_eval(o, tree, rl, PARAMS[1], freqs, rates) for rl in rls]
        assert all(math.isfinite(x) for x in want)
        for a, b in zip(got, want):
            assert util.rel_err(a, b) < LNL_TOL
        del scheds
        g.destroy()
        o.destroy()
