"""Pins the CPU oracle (oracle/rd_oracle.c) against the committed golden
vectors (independent SciPy pruning, oracle/gen_golden.py), closed-form JC69
values and the reference's own property tests (test/src/model.cpp:59-75,
:271-288, :367-387).  Runs without a GPU."""
import math
import os

import numpy as np
import pytest

import root_digger_amd as rd
from oracle_lib import OraclePartition, ORC_MAP_NT, orc_expm, orc_gamma_cats
import util

TOL = 1e-10          # oracle vs SciPy goldens, relative on lnL


def test_expm_matches_scipy():
    for case in util.golden("expm.json"):
        k = case["k"]
        q = np.array(case["q"]).reshape(k, k)
        got = orc_expm(q * case["t"])
        want = np.array(case["p"]).reshape(k, k)
        assert np.max(np.abs(got - want)) < 5e-14, (k, case["t"])


def test_qmatrix_convention():
    for case in util.golden("expm.json"):
        k = case["k"]
        p = OraclePartition(3, 4, k, 1, 1, 4, 1, 4)
        p.set_subst_params(0, case["subst"])
        p.set_frequencies(0, case["freqs"])
        q = p.get_qmatrix()
        assert np.allclose(q, np.array(case["q"]).reshape(k, k), rtol=1e-13, atol=1e-15)
        assert np.allclose(q.sum(axis=1), 0.0, atol=1e-13)
        assert abs(-(np.array(case["freqs"]) * np.diag(q)).sum() - 1.0) < 1e-13


def test_gamma_cats_match_scipy():
    for g in util.golden("gamma.json"):
        got = orc_gamma_cats(g["alpha"], g["cats"], 0 if g["mode"] == "mean" else 1)
        assert np.allclose(got, g["rates"], rtol=2e-9, atol=1e-12), g
        assert abs(np.mean(got) - 1.0) < 1e-9


def _jc_setup(part, tree, seqs, cmap=ORC_MAP_NT, weights=None):
    util.load_tips(part, tree, seqs, cmap, weights)
    part.set_subst_params(0, [1.0] * 12)
    part.set_frequencies(0, [0.25] * 4)


def test_single_closed_form_jc69():
    """4 taxa, one site 'A' everywhere: L = sum_x pi_x prod_paths ... closed form
    via the JC transition probabilities along each branch."""
    tree = rd.Tree.from_file(os.path.join(util.DATA, "single.tree"))
    seqs = util.read_phylip(os.path.join(util.DATA, "single.phy"))
    part = OraclePartition.for_tree(tree, 4, 1, 1)
    _jc_setup(part, tree, seqs)

    def same(t):
        return 0.25 + 0.75 * math.exp(-4.0 * t / 3.0)

    def diff(t):
        return 0.25 - 0.25 * math.exp(-4.0 * t / 3.0)

    # unrooted tree: (a:.1,b:.1)n1 --0.55-- n2(c:.1,d:.1); JC is reversible so any
    # root gives the same value; evaluate at n1.
    def cherry(x):   # P(tips = A,A | node = x)
        return (same(0.1) if x == 0 else diff(0.1)) ** 2
    lik = 0.0
    for x in range(4):
        for y in range(4):
            pxy = same(0.55) if x == y else diff(0.55)
            lik += 0.25 * cherry(x) * pxy * cherry(y)
    want = math.log(lik)
    for g in util.golden("single_jc.json"):
        rl = util.find_root(tree, g["near_tips"], g["far_tips"], g["alpha"])
        got = util.compute_lh(part, tree, rl)
        assert util.rel_err(got, want) < 1e-13
        assert util.rel_err(got, g["lnl"]) < 1e-13


@pytest.mark.parametrize("compressed", [False, True])
def test_ten_fasta_all_roots(compressed):
    """BASELINE config c1: the reference's 10.fasta / 10.tree at all 17 roots."""
    tree = rd.Tree.from_file(os.path.join(util.DATA, "10.tree"))
    seqs = util.read_fasta(os.path.join(util.DATA, "10.fasta"))
    weights = None
    if compressed:
        seqs, weights = util.compress(seqs)
        assert weights.sum() == 1000
    nsites = len(next(iter(seqs.values())))
    g = util.golden("ten_fasta.json")
    assert tree.root_count() == 17
    parts = {}
    for case in g["cases"]:
        R = case["rate_cats"]
        if R not in parts:
            parts[R] = OraclePartition.for_tree(tree, 4, nsites, R)
            util.load_tips(parts[R], tree, seqs, ORC_MAP_NT, weights)
            assert np.allclose(parts[R].empirical_frequencies(), g["empirical_freqs"],
                               rtol=1e-12)
        part = parts[R]
        part.set_subst_params(0, case["subst"])
        part.set_frequencies(0, case["freqs"])
        part.set_category_rates(case["rates"])
        part.set_category_weights([1.0 / R] * R)
        assert len(case["roots"]) == 17
        for root in case["roots"]:
            rl = util.find_root(tree, root["near_tips"], root["far_tips"], root["alpha"])
            got = util.compute_lh(part, tree, rl)
            assert util.rel_err(got, root["lnl"]) < TOL, (case["param_set"], R, rl.id)
            # test/src/model.cpp:271-288: full traversal == root-only evaluation
            assert util.rel_err(util.compute_lh_root(part, tree, rl), got) < 1e-13


def test_hundred_one_goldens_and_jc_root_invariance():
    """101.phy has ambiguity codes (M R S W Y X) and zero-length branches;
    test/src/model.cpp:367-387: under JC every rooting gives the same lnL."""
    tree = rd.Tree.from_file(os.path.join(util.DATA, "101.tree"))
    seqs, weights = util.compress(util.read_phylip(os.path.join(util.DATA, "101.phy")))
    nsites = len(next(iter(seqs.values())))
    g = util.golden("hundred_one.json")
    for case in g["cases"]:
        R = case["rate_cats"]
        part = OraclePartition.for_tree(tree, 4, nsites, R)
        util.load_tips(part, tree, seqs, ORC_MAP_NT, weights)
        assert np.allclose(part.empirical_frequencies(), g["empirical_freqs"], rtol=1e-12)
        part.set_subst_params(0, case["subst"])
        part.set_frequencies(0, case["freqs"])
        part.set_category_rates(case["rates"])
        for root in case["roots"]:
            rl = util.find_root(tree, root["near_tips"], root["far_tips"], root["alpha"])
            got = util.compute_lh(part, tree, rl)
            assert util.rel_err(got, root["lnl"]) < TOL
        if case["name"] == "jc":
            # move_root sweep as compute_all_root_lh does (src/model.cpp:1737-1746)
            first = util.compute_lh(part, tree, tree.root_location(0))
            vals = []
            for rl in tree.roots():
                util.move_root(part, tree, rl)
                vals.append(util.compute_lh_root(part, tree, rl))
            assert len(vals) == 199
            assert max(abs(v - first) for v in vals) < 1e-7 * abs(first)
        part.destroy()


def test_deep_tree_scaling_rule():
    """2^256 per-site scaler rule on a 161-taxon caterpillar:
    stored CLV * 2^(-256*scaler) must equal the true conditional likelihood."""
    g = util.golden("deep_scaling.json")
    tree = rd.Tree.from_newick(g["newick"])
    nsites = len(next(iter(g["seqs"].values())))
    part = OraclePartition.for_tree(tree, 4, nsites, 4)
    util.load_tips(part, tree, g["seqs"], ORC_MAP_NT)
    part.set_subst_params(0, g["subst"])
    part.set_frequencies(0, g["freqs"])
    part.set_category_rates(g["rates"])
    # root on the golden's branch
    rl = None
    for cand in tree.roots():
        if sorted(tree.side_tips(cand)) == sorted(g["near_tips"]):
            rl = cand.with_ratio(g["alpha"])
    if rl is None:
        allt = set(tree.label_map())
        for cand in tree.roots():
            if sorted(allt - set(tree.side_tips(cand))) == sorted(g["near_tips"]):
                rl = cand.with_ratio(1 - g["alpha"])
    assert rl is not None
    ops, pmi, brl = tree.generate_operations(rl)
    part.update_prob_matrices(pmi, brl)
    part.update_clvs(ops)
    lnl, persite = part.compute_root_loglikelihood(tree.root_clv_index(),
                                                   tree.root_scaler_index(), persite=True)
    assert util.rel_err(lnl, g["lnl"]) < TOL
    assert np.allclose(persite, g["persite"], rtol=1e-10)
    # at least one site had to be rescaled, and the big nodes are pinned
    assert part.get_scaler(tree.root_scaler_index()).max() >= 1
    checked = 0
    by_tips = {tuple(sorted(n["tips"])): n for n in g["nodes"]}
    label_of = {v: k for k, v in tree.label_map().items()}
    below = {}
    for op in ops:   # tips under each computed CLV
        tips = []
        for c in (op.child1_clv_index, op.child2_clv_index):
            tips += [label_of[c]] if c < tree.tip_count() else below[c]
        below[op.parent_clv_index] = tips
        key = tuple(sorted(tips))
        if key in by_tips and op is not ops[len(ops) - 1]:
            clv = part.get_clv(op.parent_clv_index)
            sc = part.get_scaler(op.parent_scaler_index).astype(np.float64)
            with np.errstate(divide="ignore"):
                got = np.log(clv) - 256.0 * math.log(2.0) * sc[:, None, None]
            want = np.array(by_tips[key]["log_clv"])
            assert np.allclose(got, want, rtol=1e-10, atol=1e-9), len(tips)
            if len(tips) >= 150:
                assert sc.min() >= 1
            checked += 1
    assert checked >= 3


def test_protein20_goldens():
    g = util.golden("protein20.json")
    tree = rd.Tree.from_newick(g["newick"])
    aa = g["alphabet"]
    cmap = util.make_map(aa, {"X": (1 << 20) - 1,
                              "B": (1 << aa.index("N")) | (1 << aa.index("D"))})
    nsites = len(next(iter(g["seqs"].values())))
    part = OraclePartition.for_tree(tree, 20, nsites, 4)
    util.load_tips(part, tree, g["seqs"], cmap)
    part.set_subst_params(0, g["subst"])
    part.set_frequencies(0, g["freqs"])
    part.set_category_rates(g["rates"])
    for root in g["roots"]:
        rl = util.find_root(tree, root["near_tips"], root["far_tips"], root["alpha"])
        got = util.compute_lh(part, tree, rl)
        assert util.rel_err(got, root["lnl"]) < TOL


def test_binary2_goldens():
    """2-state characters with gaps (`rd --states 2`): the oracle's native
    2-state arithmetic against the SciPy pruning, every rooting."""
    g = util.golden("binary2.json")
    tree = rd.Tree.from_newick(g["newick"])
    cmap = util.make_map("01", {"-": 3, "?": 3})
    nsites = len(next(iter(g["seqs"].values())))
    part = OraclePartition.for_tree(tree, 2, nsites, 4)
    util.load_tips(part, tree, g["seqs"], cmap)
    part.set_subst_params(0, g["subst"])
    part.set_frequencies(0, g["freqs"])
    part.set_category_rates(g["rates"])
    assert len(g["roots"]) == tree.root_count()
    for root in g["roots"]:
        rl = util.find_root(tree, root["near_tips"], root["far_tips"], root["alpha"])
        assert util.rel_err(util.compute_lh(part, tree, rl), root["lnl"]) < TOL


def test_determinism_and_negativity():
    """test/src/model.cpp:59-75: finite, negative, bit-identical on repeat."""
    tree = rd.Tree.from_file(os.path.join(util.DATA, "10.tree"))
    seqs = util.read_fasta(os.path.join(util.DATA, "10.fasta"))
    part = OraclePartition.for_tree(tree, 4, 1000, 1)
    util.load_tips(part, tree, seqs, ORC_MAP_NT)
    part.set_subst_params(0, [.34, .42, .24, .74, .16, .88, .75, .54, .20, .06, .08, .41])
    for rl in tree.roots():
        a = util.compute_lh(part, tree, rl)
        b = util.compute_lh(part, tree, rl)
        assert math.isfinite(a) and a < 0.0 and a == b


def test_avx2_clv_loop_is_bit_identical_to_the_scalar_one():
    """bench.py's cpu_baseline times the oracle's 256-bit-vector 4-state loop
    (orc_update_clvs_avx2; the reference selects coraxlib's AVX2 kernel for
    nucleotides, src/model.cpp:145-155).  It must produce the scalar loop's
    bits: every CLV entry, every scaler, the lnL -- on 101.phy (ambiguity codes,
    zero-length branches) and on the deep tree where the rescaling fires."""
    import root_digger_amd as rd
    cases = []
    tree = rd.Tree.from_file(os.path.join(util.DATA, "101.tree"))
    seqs, w = util.compress(util.read_phylip(os.path.join(util.DATA, "101.phy")))
    cases.append((tree, seqs, w, [.34, .42, .24, .74, .16, .88, .75, .54, .20, .06, .08, .41],
                  [0.21, 0.29, 0.24, 0.26], orc_gamma_cats(0.7, 4)))
    gd = util.golden("deep_scaling.json")
    cases.append((rd.Tree.from_newick(gd["newick"]), gd["seqs"], None, gd["subst"], gd["freqs"], gd["rates"]))
    for tree, seqs, w, subst, freqs, rates in cases:
        S = len(next(iter(seqs.values())))
        a = OraclePartition.for_tree(tree, 4, S, 4)
        b = OraclePartition.for_tree(tree, 4, S, 4)
        for p in (a, b):
            util.load_tips(p, tree, seqs, ORC_MAP_NT, w)
            p.set_subst_params(0, subst)
            p.set_frequencies(0, freqs)
            p.set_category_rates(rates)
        rl = tree.root_location(7).with_ratio(0.3)
        ops, pmi, brl = tree.generate_operations(rl)
        for p, vec in ((a, False), (b, True)):
            p.update_prob_matrices(pmi, brl)
            p.update_clvs(ops, avx2=vec)
        for op in ops:
            assert np.array_equal(a.get_clv(op.parent_clv_index), b.get_clv(op.parent_clv_index))
            if op.parent_scaler_index >= 0:
                assert np.array_equal(a.get_scaler(op.parent_scaler_index), b.get_scaler(op.parent_scaler_index))
        la = a.compute_root_loglikelihood(tree.root_clv_index(), tree.root_scaler_index())
        assert la == b.compute_root_loglikelihood(tree.root_clv_index(), tree.root_scaler_index())
    p20 = OraclePartition(4, 6, 20, 8, 1, 6, 1, 6)
    with pytest.raises(RuntimeError):
        p20.update_clvs([], avx2=True)


def test_site_repeats_traversal_is_bit_identical_to_the_plain_one():
    """The reference runs coraxlib WITH subtree site repeats on 4-state data
    (src/model.cpp:145-149); bench.py's cpu_baseline.with_site_repeats times the oracle's
    restatement of that scheme (orc_update_clvs_repeats: a node's CLV once per class of columns
    that agree at all tips below it).  Same arithmetic per class, so the log-likelihood must be
    the plain loop's bit for bit -- scalar and 256-bit loops, pattern weights, ambiguity codes
    (101.phy), the deep caterpillar where the 2^256 rule fires, 20 states -- over several
    rootings of one partition (the class tables are rebuilt per traversal)."""
    import root_digger_amd as rd
    cases = []
    tree = rd.Tree.from_file(os.path.join(util.DATA, "101.tree"))
    seqs, w = util.compress(util.read_phylip(os.path.join(util.DATA, "101.phy")))
    cases.append((tree, seqs, w, 4, ORC_MAP_NT, [.34, .42, .24, .74, .16, .88, .75, .54, .20, .06, .08, .41],
                  [0.21, 0.29, 0.24, 0.26], orc_gamma_cats(0.7, 4)))
    tree10 = rd.Tree.from_file(os.path.join(util.DATA, "10.tree"))
    cases.append((tree10, util.read_fasta(os.path.join(util.DATA, "10.fasta")), None, 4, ORC_MAP_NT,
                  [.3, .4, .2, .7, .1, .8, .7, .5, .2, .1, .1, .4], [0.25, 0.25, 0.25, 0.25], [1.0]))
    gd = util.golden("deep_scaling.json")
    cases.append((rd.Tree.from_newick(gd["newick"]), gd["seqs"], None, 4, ORC_MAP_NT, gd["subst"], gd["freqs"], gd["rates"]))
    gp = util.golden("protein20.json")
    aa = gp["alphabet"]
    cases.append((rd.Tree.from_newick(gp["newick"]), gp["seqs"], None, 20,
                  util.make_map(aa, {"X": (1 << 20) - 1, "B": (1 << aa.index("N")) | (1 << aa.index("D"))}),
                  gp["subst"], gp["freqs"], gp["rates"]))
    for tree, seqs, w, K, cmap, subst, freqs, rates in cases:
        S = len(next(iter(seqs.values())))
        R = len(rates)
        a = OraclePartition.for_tree(tree, K, S, R)
        b = OraclePartition.for_tree(tree, K, S, R)
        for p in (a, b):
            util.load_tips(p, tree, seqs, cmap, w)
            p.set_subst_params(0, subst)
            p.set_frequencies(0, freqs)
            p.set_category_rates(rates)
        for i, vec in ((7, False), (0, True), (tree.root_count() - 1, K == 4)):
            rl = tree.root_location(i).with_ratio(0.3)
            ops, pmi, brl = tree.generate_operations(rl)
            a.update_prob_matrices(pmi, brl)
            a.update_clvs(ops)
            want = a.compute_root_loglikelihood(tree.root_clv_index(), tree.root_scaler_index())
            b.update_prob_matrices(pmi, brl)
            b.update_clvs_repeats(ops, avx2=vec)
            got = b.compute_root_loglikelihood_repeats(tree.root_clv_index(), tree.root_scaler_index())
            assert got == want, (K, i, got, want)
        assert 0.0 < b.repeats_ratio() <= 1.0
        # the bytes the three traversals moved (bench.py: cpu_baseline.one_socket_bandwidth_bound):
        # at most what a plain loop writes and reads, compulsory reads <= uncached reads, and
        # the written CLV bytes follow the class ratio
        written, once, every = b.repeats_bytes()
        n_ops, clv = 3 * (tree.tip_count() - 1), R * K * 8
        assert 0 < once <= every
        assert written <= n_ops * S * (clv + 4 + 8 + 4) and every <= n_ops * S * (8 + 2 * (clv + 4))
        assert abs(written - (b.repeats_ratio() * n_ops * S * (clv + 4 + 8) + 4 * n_ops * S)) <= 4 * b.repeats_ratio() * n_ops * S
        if tree is tree10:       # 1000 columns of 10 taxa: most subtrees see few distinct patterns
            assert b.repeats_ratio() < 0.5
        if K == 4 and w is None and tree is not tree10:
            # the caterpillar's first cherries fold, its deep nodes hold a class per column
            assert b.repeats_ratio() < 1.0


def test_stream_triad_runs_on_the_hosts_cores():
    """bench.py's measured memory bandwidth (cpu_baseline.one_socket_bandwidth_bound.stream_triad)"""
    from oracle_lib import stream_triad
    cpus = sorted(os.sched_getaffinity(0))[:2]
    gbs = stream_triad(cpus, 1 << 20, 0.1)
    assert 0.1 < gbs < 5000.0
