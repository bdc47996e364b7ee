"""Subtree site repeats in the fused evaluator ("clade tables"; the reference
switches coraxlib's site repeats on for every 4-state partition,
/root/reference/src/model.cpp:145-149).  A partition created with
RDAMD_ATTRIB_SITE_REPEATS folds every clade with few tip-pattern classes into a pseudo-tip;
the results must be those of the plain traversal: <= 1e-11 relative against the oracle
(which has no repeats at all), <= 1e-12 against the same library without them."""
import numpy as np
import pytest

import root_digger_amd as rd
from root_digger_amd import synth
from oracle_lib import OraclePartition, ORC_MAP_NT
import util

pytestmark = pytest.mark.gpu

LNL_TOL = 1e-11


def trio(tree, seqs, R, weights=None, classes=16):
    """(partition with repeats, partition without, oracle) on the same data; classes = the
    pseudo-tips' class limit (16: 16-row tables only; 64: 64-row tables through LDS-DMA too)"""
    S = len(next(iter(seqs.values())))
    a = rd.Partition.for_tree(tree, 4, S, R, attributes=rd.ATTRIB_SITE_REPEATS)
    a.set_site_repeats(classes)
    b = rd.Partition.for_tree(tree, 4, S, R)
    o = OraclePartition.for_tree(tree, 4, S, R)
    for p, m in ((a, rd.MAP_NT), (b, rd.MAP_NT), (o, ORC_MAP_NT)):
        util.load_tips(p, tree, seqs, m, weights)
    return a, b, o


def oracle_eval(o, tree, rl, subst, freqs, rates, weights=None):
    o.set_subst_params(0, subst)
    o.set_frequencies(0, freqs)
    o.set_category_rates(rates)
    if weights is not None:
        o.set_category_weights(weights)
    return util.compute_lh(o, tree, rl)


CLASSES = [16, 64]


@pytest.mark.parametrize("classes", CLASSES)
@pytest.mark.parametrize("n,S,R,seed", [(100, 2000, 4, 141), (37, 1000, 1, 142), (64, 333, 2, 143),
                                        (16, 65, 3, 144), (5, 7, 4, 145), (4, 1, 4, 146)])
def test_repeats_vs_oracle_on_random_shapes(n, S, R, seed, classes):
    w = synth.workload(n, S, 4, R, seed)
    tree = rd.Tree.from_newick(w["newick"])
    rng = np.random.default_rng(seed)
    weights = rng.integers(1, 4, size=S).astype(np.uint32)
    if classes > 16:   # gaps and ambiguity codes as in real alignments: cherries with 17-64 classes
        seqs = {}
        for k, v in w["seqs"].items():
            v = np.frombuffer(v.encode(), dtype=np.uint8).copy()
            v[rng.random(S) < 0.15] = ord("-")
            v[rng.random(S) < 0.02] = ord("R")
            seqs[k] = v.tobytes().decode()
        w["seqs"] = seqs
    a, b, o = trio(tree, w["seqs"], R, weights, classes)
    picks = rng.choice(tree.root_count(), size=min(6, tree.root_count()), replace=False)
    rls = [tree.root_location(int(i)).with_ratio(float(rng.uniform(0.02, 0.98))) for i in picks]
    sa = [a.schedule(*tree.generate_operations(rl)) for rl in rls]
    sb = [b.schedule(*tree.generate_operations(rl)) for rl in rls]
    subst = rng.uniform(1e-4, 1.0, (len(rls), 12))
    freqs = rng.dirichlet(np.ones(4) * 5, len(rls))
    rates = np.array([rd.compute_gamma_cats(x, R) for x in rng.uniform(0.3, 3.0, len(rls))])
    cw = rng.dirichlet(np.ones(R) * 3, len(rls))
    got = a.evaluate_batch(sa, subst, freqs, rates, cw)
    ref = b.evaluate_batch(sb, subst, freqs, rates, cw)
    for j, rl in enumerate(rls):
        want = oracle_eval(o, tree, rl, subst[j], freqs[j], rates[j], cw[j])
        assert util.rel_err(got[j], want) < LNL_TOL, (j, got[j], want)
        assert util.rel_err(got[j], ref[j]) < 1e-12
    assert np.array_equal(got, a.evaluate_batch(sa, subst, freqs, rates, cw))   # test/src/model.cpp:73
    # the synthetic tips are unambiguous: every cherry has at most 16 classes and is folded
    for s_a, s_b in zip(sa, sb):
        st, plain = s_a.stats(), s_b.stats()
        assert plain["pseudo_tips"] == 0 and plain["steps"] == plain["operations"] == n - 1
        assert st["operations"] == n - 1 and st["matvecs_plain"] == plain["matvecs"] == n - 2
        if n >= 5:
            assert st["pseudo_tips"] >= 1
        assert st["steps"] == st["operations"] - st["clade_nodes"]
        assert st["matvecs"] == st["matvecs_plain"] - st["clade_nodes"]   # each folded node: one product less
        assert 1 <= st["stack_depth"] <= st["stack_depth_plain"]
    for p in (a, b, o):
        p.destroy()


def test_parks_of_a_deep_tree_find_a_slot():
    """The programs of a partition with 64-row tables run on the kernels with ONE register slot, ONE
    LDS slot and a private-segment stack, and the host places every park on its own
    (traversal_compiler.hpp, place_levels): on a 700-taxon tree all but a few of a traversal's
    parks must find one of the two slots -- two whole stack levels hold far fewer --, and the
    values are the oracle's."""
    n, S, R = 700, 300, 4
    w = synth.workload(n, S, 4, R, 181)
    tree = rd.Tree.from_newick(w["newick"])
    a, b, o = trio(tree, w["seqs"], R, None, 64)
    rng = np.random.default_rng(181)
    rls = [tree.root_location(int(i)).with_ratio(float(rng.uniform(0.05, 0.95)))
           for i in rng.choice(tree.root_count(), size=4, replace=False)]
    sa = [a.schedule(*tree.generate_operations(rl)) for rl in rls]
    subst = rng.uniform(1e-3, 1.0, (len(rls), 12))
    freqs = rng.dirichlet(np.ones(4) * 5, len(rls))
    rates = np.array([rd.compute_gamma_cats(x, R) for x in rng.uniform(0.3, 3.0, len(rls))])
    got = a.evaluate_batch(sa, subst, freqs, rates)
    for j, rl in enumerate(rls):
        want = oracle_eval(o, tree, rl, subst[j], freqs[j], rates[j])
        assert util.rel_err(got[j], want) < LNL_TOL, (j, got[j], want)
    for st in (s.stats() for s in sa):
        in_slots = st["parks_in_registers"] + st["parks_in_lds_slot"]
        assert st["parks"] > 30 and st["parks_in_lds_slot"] > 0
        assert st["parks"] >= in_slots >= 0.9 * st["parks"], st
    for p in (a, b, o):
        p.destroy()


@pytest.mark.parametrize("classes", CLASSES)
def test_repeats_on_the_reference_fixtures(classes):
    """10.fasta (all 17 roots, the four parameter sets of test/src/model.cpp:12-17) and 101.phy
    (ambiguity codes, zero-length branches: most of its cherries have more than 16 classes and
    stay as they are -- the mixture of folded and unfolded clades is the point)."""
    gd = util.golden("ten_fasta.json")
    tree = rd.Tree.from_file(util.DATA + "/10.tree")
    seqs = util.read_fasta(util.DATA + "/10.fasta")
    a, b, o = trio(tree, seqs, 4, None, classes)
    freqs = a.empirical_frequencies()
    rates = rd.compute_gamma_cats(1.0, 4)
    a.set_category_rates(rates)
    rls = [rl.with_ratio(0.5) for rl in tree.roots()]
    scheds = [a.schedule(*tree.generate_operations(rl)) for rl in rls]
    for case in gd["cases"][:4]:
        got = a.evaluate_batch(scheds, np.tile(case["subst"], (len(rls), 1)), np.tile(freqs, (len(rls), 1)))
        for j, rl in enumerate(rls):
            assert util.rel_err(got[j], oracle_eval(o, tree, rl, case["subst"], freqs, rates)) < LNL_TOL
    for p in (a, b, o):
        p.destroy()

    tree = rd.Tree.from_file(util.DATA + "/101.tree")
    seqs, weights = util.compress(util.read_phylip(util.DATA + "/101.phy"))
    a, b, o = trio(tree, seqs, 4, weights, classes)
    freqs = a.empirical_frequencies()
    rng = np.random.default_rng(7)
    picks = [0, 57, 101, 150, tree.root_count() - 1]
    rls = [tree.root_location(i).with_ratio(float(rng.uniform(0.1, 0.9))) for i in picks]
    scheds = [a.schedule(*tree.generate_operations(rl)) for rl in rls]
    folded = [s_.stats()["clade_nodes"] for s_ in scheds]
    assert min(folded) >= (25 if classes > 16 else 0)     # census: ~32 of 100 operations at 64 classes, ~0 at 16
    subst = rng.uniform(1e-3, 1.0, (len(rls), 12))
    got = a.evaluate_batch(scheds, subst, np.tile(freqs, (len(rls), 1)), np.tile(rates, (len(rls), 1)))
    for j, rl in enumerate(rls):
        assert util.rel_err(got[j], oracle_eval(o, tree, rl, subst[j], freqs, rates)) < LNL_TOL
    for p in (a, b, o):
        p.destroy()


@pytest.mark.parametrize("classes", CLASSES)
def test_repeats_on_the_deep_scaling_caterpillar(classes):
    """161-taxon caterpillar whose CLVs are rescaled many times: the folded cherry at the far
    end carries no rescale count, everything above it does."""
    gd = util.golden("deep_scaling.json")
    tree = rd.Tree.from_newick(gd["newick"])
    a, b, o = trio(tree, gd["seqs"], 4, None, classes)
    for i in (0, 100, 250, tree.root_count() - 1):
        rl = tree.root_location(i).with_ratio(0.31)
        sched = a.schedule(*tree.generate_operations(rl))
        got = a.evaluate_batch([sched], [gd["subst"]], [gd["freqs"]], [gd["rates"]])[0]
        assert util.rel_err(got, oracle_eval(o, tree, rl, gd["subst"], gd["freqs"], gd["rates"])) < LNL_TOL
    for p in (a, b, o):
        p.destroy()


@pytest.mark.parametrize("classes", CLASSES)
def test_tiny_table_entries_fall_back_to_the_plain_programs(classes):
    """A pseudo-tip has no rescale count, so the launch must not use them when a clade's class
    could have been rescaled or a table entry is small enough for a tip-tip product to need
    it: with a rate category of 1e-42 the off-diagonal P entries are < 2^-128, the job's flag
    goes up and the evaluator walks its PLAIN program -- the same kernel and program as a
    partition without repeats, hence the same bits."""
    w = synth.workload(30, 700, 4, 4, 151)
    tree = rd.Tree.from_newick(w["newick"])
    a, b, o = trio(tree, w["seqs"], 4, None, classes)
    rng = np.random.default_rng(151)
    rls = [tree.root_location(int(i)).with_ratio(0.4) for i in rng.choice(tree.root_count(), 4, replace=False)]
    sa = [a.schedule(*tree.generate_operations(rl)) for rl in rls]
    sb = [b.schedule(*tree.generate_operations(rl)) for rl in rls]
    assert all(s.stats()["pseudo_tips"] > 0 for s in sa)
    subst = rng.uniform(1e-2, 1.0, (4, 12))
    freqs = rng.dirichlet(np.ones(4) * 5, 4)
    rates = np.tile([1e-42, 0.5, 1.0, 2.5], (4, 1))
    got = a.evaluate_batch(sa, subst, freqs, rates)
    assert np.array_equal(got, b.evaluate_batch(sb, subst, freqs, rates))
    for j, rl in enumerate(rls):
        assert util.rel_err(got[j], oracle_eval(o, tree, rl, subst[j], freqs[j], rates[j])) < LNL_TOL
    # ... per JOB: one such job in a batch runs its plain program, the others keep their folded
    # ones, and no job's value depends on what else shares its launch (the lock-stepped search
    # combines candidates' batches and must reproduce the sequential trajectories bit for bit)
    rates[1:] = rd.compute_gamma_cats(1.0, 4)
    mixed = a.evaluate_batch(sa, subst, freqs, rates)
    plain = b.evaluate_batch(sb, subst, freqs, rates)
    assert mixed[0] == plain[0]
    assert np.max(np.abs(mixed - plain) / np.abs(plain)) < 1e-12
    for j in range(4):
        alone = a.evaluate_batch([sa[j]], subst[j:j + 1], freqs[j:j + 1], rates[j:j + 1])[0]
        assert alone == mixed[j], j
    # ordinary rates again: back on the folded programs, same values to rounding
    rates[0] = rd.compute_gamma_cats(1.0, 4)
    got = a.evaluate_batch(sa, subst, freqs, rates)
    ref = b.evaluate_batch(sb, subst, freqs, rates)
    assert np.max(np.abs(got - ref) / np.abs(ref)) < 1e-12
    assert np.array_equal(got[1:], mixed[1:])
    for p in (a, b, o):
        p.destroy()


def test_repeats_switch_and_stale_schedules():
    w = synth.workload(20, 256, 4, 2, 161)
    tree = rd.Tree.from_newick(w["newick"])
    a = rd.Partition.for_tree(tree, 4, 256, 2, attributes=rd.ATTRIB_SITE_REPEATS)
    util.load_tips(a, tree, w["seqs"], rd.MAP_NT)
    rl = tree.root_location(3)
    ops = tree.generate_operations(rl)
    s1 = a.schedule(*ops)
    assert s1.stats()["pseudo_tips"] > 0
    base = a.evaluate_batch([s1], [w["subst"]], [[0.25] * 4])[0]
    a.set_site_repeats(0)                                  # off: schedules compiled from now on are plain
    s0 = a.schedule(*ops)
    assert s0.stats()["pseudo_tips"] == 0
    assert util.rel_err(a.evaluate_batch([s0], [w["subst"]], [[0.25] * 4])[0], base) < 1e-12
    assert a.evaluate_batch([s1], [w["subst"]], [[0.25] * 4])[0] == base   # the old one still runs
    with pytest.raises(rd.RdamdError):
        a.set_site_repeats(65)
    a.set_site_repeats(16)                                 # 8-bit codes, 16-row tables only
    s16 = a.schedule(*ops)
    assert 0 < s16.stats()["pseudo_tips"] and util.rel_err(
        a.evaluate_batch([s16], [w["subst"]], [[0.25] * 4])[0], base) < 1e-12
    with pytest.raises(rd.RdamdError):                     # 16-row and 64-row schedules in one launch
        a.evaluate_batch([s16, s1], [w["subst"]] * 2, [[0.25] * 4] * 2)
    a.set_site_repeats(64)
    # new characters at a tip: the class codes of s1 describe the old alignment
    label = next(iter(w["seqs"]))
    a.set_tip_states(tree.tip_index(label), rd.MAP_NT, w["seqs"][label][::-1])
    with pytest.raises(rd.RdamdError) as e:
        a.evaluate_batch([s1], [w["subst"]], [[0.25] * 4])
    assert "stale" in str(e.value)
    s2 = a.schedule(*ops)
    b = rd.Partition.for_tree(tree, 4, 256, 2)
    seqs = dict(w["seqs"])
    seqs[label] = seqs[label][::-1]
    util.load_tips(b, tree, seqs, rd.MAP_NT)
    want = b.evaluate_batch([b.schedule(*ops)], [w["subst"]], [[0.25] * 4])[0]
    assert util.rel_err(a.evaluate_batch([s2], [w["subst"]], [[0.25] * 4])[0], want) < 1e-12
    a.destroy()
    b.destroy()


def test_wide_schedule_without_pseudo_tips_survives_new_tip_states():
    """a 64-row schedule in which no clade is small enough to fold (ambiguity codes
    everywhere: a cherry has up to 225 classes) holds no class codes, so it stays valid when a
    tip gets new characters -- but rdamd_set_tip_states drops the 16-bit code arena it reads.
    The next batch must find the arena rebuilt from the new characters."""
    rng = np.random.default_rng(611)
    w = synth.workload(9, 1500, 4, 4, 611)
    tree = rd.Tree.from_newick(w["newick"])
    iupac = np.frombuffer(b"ACGTRYSWKMBDHVN", dtype=np.uint8)
    seqs = {k: iupac[rng.integers(0, 15, 1500)].tobytes().decode() for k in w["seqs"]}
    a = rd.Partition.for_tree(tree, 4, 1500, 4, attributes=rd.ATTRIB_SITE_REPEATS)
    o = OraclePartition.for_tree(tree, 4, 1500, 4)
    util.load_tips(a, tree, seqs, rd.MAP_NT)
    util.load_tips(o, tree, seqs, ORC_MAP_NT)
    rl = tree.root_location(5).with_ratio(0.3)
    s = a.schedule(*tree.generate_operations(rl))
    assert s.stats()["pseudo_tips"] == 0
    freqs = [0.2, 0.3, 0.3, 0.2]
    rates = rd.compute_gamma_cats(0.7, 4)
    got = a.evaluate_batch([s], [w["subst"]], [freqs], [rates])[0]
    assert util.rel_err(got, oracle_eval(o, tree, rl, w["subst"], freqs, rates)) < LNL_TOL
    label = sorted(seqs)[2]
    seqs[label] = seqs[label][::-1]
    for p, m in ((a, rd.MAP_NT), (o, ORC_MAP_NT)):
        p.set_tip_states(tree.tip_index(label), m, seqs[label])
    got2 = a.evaluate_batch([s], [w["subst"]], [freqs], [rates])[0]      # the SAME schedule
    want2 = oracle_eval(o, tree, rl, w["subst"], freqs, rates)
    assert got2 != got and util.rel_err(got2, want2) < LNL_TOL
    a.destroy()


@pytest.mark.parametrize("classes", CLASSES)
def test_shared_matrix_index_is_not_folded(classes):
    """the C ABI (like coraxlib's) lets two branches share a P-matrix index.  A
    pseudo-tip's table lives in the tip-table slot of the branch above it, which is only free
    when that index belongs to this one branch: a list that shares an index is evaluated
    without folding, and still agrees with the oracle."""
    w = synth.workload(12, 900, 4, 4, 623)
    tree = rd.Tree.from_newick(w["newick"])
    a, b, o = trio(tree, w["seqs"], 4, classes=classes)
    rl = tree.root_location(4).with_ratio(0.6)
    ops, pmi, brl = tree.generate_operations(rl)
    ops = list(ops)
    pmi, brl = list(pmi), list(brl)
    # the branch above a folded cherry and one tip branch elsewhere share a matrix (same length)
    plain = a.schedule(ops, pmi, brl)
    assert plain.stats()["pseudo_tips"] > 0
    inner = [i for i, op in enumerate(ops[:-1]) if op.child1_clv_index < tree.tip_count()
             and op.child2_clv_index < tree.tip_count()]
    cherry = ops[inner[0]]
    above = next(op for op in ops if cherry.parent_clv_index in (op.child1_clv_index, op.child2_clv_index))
    m_above = above.child1_matrix_index if above.child1_clv_index == cherry.parent_clv_index else above.child2_matrix_index
    other = ops[inner[1]]
    m_old = other.child1_matrix_index
    other.child1_matrix_index = m_above                    # a tip branch now reads the cherry's branch matrix
    ops[inner[1]] = other
    keep = [k for k, m in enumerate(pmi) if m != m_old]
    pmi2, brl2 = [pmi[k] for k in keep], [brl[k] for k in keep]
    s = a.schedule(ops, pmi2, brl2)
    assert s.stats()["pseudo_tips"] == 0
    freqs = [0.25, 0.25, 0.3, 0.2]
    rates = rd.compute_gamma_cats(1.3, 4)
    got = a.evaluate_batch([s], [w["subst"]], [freqs], [rates])[0]
    for p in (o,):
        p.set_subst_params(0, w["subst"]); p.set_frequencies(0, freqs); p.set_category_rates(rates)
        p.update_prob_matrices(pmi2, brl2)
        p.update_clvs(ops)
        want = p.compute_root_loglikelihood(tree.root_clv_index(), tree.root_scaler_index())
    assert util.rel_err(got, want) < LNL_TOL
    a.destroy(); b.destroy()


def test_binary_data_with_repeats():
    """2-state partitions run on the 4-state kernels and fold their clades too."""
    rng = np.random.default_rng(171)
    w = synth.workload(24, 400, 2, 1, 171)
    tree = rd.Tree.from_newick(w["newick"])
    cmap = util.make_map(w["alphabet"], {"-": 3, "?": 3})
    seqs = {k: v[:100] + "-" * 3 + v[103:] for k, v in w["seqs"].items()}
    S = 400
    a = rd.Partition.for_tree(tree, 2, S, 1, attributes=rd.ATTRIB_SITE_REPEATS)
    o = OraclePartition.for_tree(tree, 2, S, 1)
    util.load_tips(a, tree, seqs, cmap)
    util.load_tips(o, tree, seqs, cmap)
    rls = [tree.root_location(int(i)).with_ratio(0.6) for i in rng.choice(tree.root_count(), 3, replace=False)]
    scheds = [a.schedule(*tree.generate_operations(rl)) for rl in rls]
    assert all(s.stats()["pseudo_tips"] > 0 for s in scheds)
    subst = rng.uniform(0.1, 1.0, (3, 2))
    freqs = rng.dirichlet(np.ones(2) * 5, 3)
    got = a.evaluate_batch(scheds, subst, freqs)
    for j, rl in enumerate(rls):
        assert util.rel_err(got[j], oracle_eval(o, tree, rl, subst[j], freqs[j], [1.0])) < LNL_TOL
    a.destroy()
    o.destroy()


def test_c2_full_size_with_repeats():
    """BASELINE config c2 at full size (100 taxa x 50 000 sites, UNREST+G4): determinism,
    the library without repeats as a second implementation, root invariance under a
    reversible model over all 197 roots in one batch."""
    w = synth.workload(100, 50000, 4, 4, 0xD166E5 + 1)
    tree = rd.Tree.from_newick(w["newick"])
    a = rd.Partition.for_tree(tree, 4, 50000, 4, attributes=rd.ATTRIB_SITE_REPEATS)
    b = rd.Partition.for_tree(tree, 4, 50000, 4)
    for p in (a, b):
        util.load_tips(p, tree, w["seqs"], rd.MAP_NT)
        p.set_category_rates(w["rates"])
    freqs = np.array(a.empirical_frequencies())
    rng = np.random.default_rng(99)
    sa = [a.schedule(*tree.generate_operations(rl)) for rl in tree.roots()]
    st = [s.stats() for s in sa]
    assert min(x["pseudo_tips"] for x in st) >= 20 and max(x["steps"] for x in st) <= 75
    picks = rng.choice(197, 24, replace=False)
    subst = rng.uniform(1e-4, 1.0, (24, 12))
    got = a.evaluate_batch([sa[i] for i in picks], subst, np.tile(freqs, (24, 1)))
    assert np.all(np.isfinite(got)) and np.all(got < 0)
    assert np.array_equal(got, a.evaluate_batch([sa[i] for i in picks], subst, np.tile(freqs, (24, 1))))
    # a job's value does not depend on the launch it rides in: alone (one site per lane, a
    # small launch) or among 24 (two sites per lane) -- the same bits
    for j in (0, 13):
        assert a.evaluate_batch([sa[picks[j]]], subst[j:j + 1], freqs[None, :])[0] == got[j]
    sb = [b.schedule(*tree.generate_operations(tree.root_location(int(i)))) for i in picks]
    ref = b.evaluate_batch(sb, subst, np.tile(freqs, (24, 1)))
    assert np.max(np.abs(got - ref) / np.abs(ref)) < 1e-12
    jc = a.evaluate_batch(sa, np.ones((197, 12)), np.full((197, 4), 0.25))
    assert np.max(np.abs(jc - jc[0])) < 1e-9 * abs(jc[0])
    a.destroy()
    b.destroy()
