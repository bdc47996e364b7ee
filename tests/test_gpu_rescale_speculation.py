"""The 4-state evaluator's speculative first pass (include/root_digger_amd.h,
rdamd_partition_set_rescale_speculation; kernels_fused.hip, SPEC): no rescale test on the way
up, a check of every site's rate sum at the root, and the second pass -- plain program, every
test of the reference's 2^256 rule -- for a job whose smallest sum is
below 2^-900.  What must hold: the oracle's lnL to 1e-11 whichever way a job goes, a job's
value independent of what shares its batch, and no result ever taken from a traversal that
ran out of the FP64 range."""
import ctypes as C

import numpy as np
import pytest

import root_digger_amd as rd
from root_digger_amd import synth
from oracle_lib import OraclePartition, ORC_MAP_NT
import util

pytestmark = pytest.mark.gpu

LNL_TOL = 1e-11


def pair(tree, seqs, R, classes=64):
    """classes: the pseudo-tips' class limit (64: the 64-row kernels, 16: the 16-row ones), 0: a
    partition without site repeats"""
    S = len(next(iter(seqs.values())))
    a = rd.Partition.for_tree(tree, 4, S, R, attributes=rd.ATTRIB_SITE_REPEATS if classes else 0)
    if classes:
        a.set_site_repeats(classes)
    o = OraclePartition.for_tree(tree, 4, S, R)
    util.load_tips(a, tree, seqs, rd.MAP_NT)
    util.load_tips(o, tree, seqs, ORC_MAP_NT)
    return a, o


def oracle_eval(o, ops, pmi, brl, tree, subst, freqs, rates):
    o.set_subst_params(0, subst)
    o.set_frequencies(0, freqs)
    o.set_category_rates(rates)
    o.update_prob_matrices(pmi, brl)
    o.update_clvs(OraclePartition.pack_ops(ops))
    return o.compute_root_loglikelihood(tree.root_clv_index(), tree.root_scaler_index())


def params(rng, J, R):
    return (rng.uniform(1e-2, 1.0, (J, 12)), rng.dirichlet(np.ones(4) * 5, J),
            np.tile(rd.compute_gamma_cats(1.0, R), (J, 1)))


@pytest.mark.parametrize("classes", [64, 16, 0])
@pytest.mark.parametrize("n,S,R,seed", [(100, 3000, 4, 601), (37, 1000, 1, 602), (125, 777, 2, 603), (9, 130, 4, 604)])
def test_the_two_modes_give_the_oracles_values(n, S, R, seed, classes):
    """simulated alignments of ordinary trees: nothing is flagged, the speculative pass is the
    only pass; (site, rate) vectors that the tests would have rescaled on the way (a slow category
    at a fast site: c2 has ~0.3 % of them) come out with the same mantissas -- the factors of the
    rule are exact powers of two -- so the two modes agree to the last bit here"""
    w = synth.workload(n, S, 4, R, seed)
    tree = rd.Tree.from_newick(w["newick"])
    a, o = pair(tree, w["seqs"], R, classes)
    rng = np.random.default_rng(seed)
    J = 5
    subst, freqs, rates = params(rng, J, R)
    lists = [tree.generate_operations(tree.root_location(int(i)).with_ratio(0.37))
             for i in rng.choice(tree.root_count(), J, replace=False)]
    scheds = [a.schedule(*l) for l in lists]
    a.set_rescale_speculation(1)
    spec = a.evaluate_batch(scheds, subst, freqs, rates)
    assert a.second_passes() == 0
    a.set_rescale_speculation(0)
    plain = a.evaluate_batch(scheds, subst, freqs, rates)
    a.set_rescale_speculation(-1)          # up to 256 tips: on
    assert np.array_equal(a.evaluate_batch(scheds, subst, freqs, rates), spec)
    assert np.array_equal(spec, plain)
    for j, (ops, pmi, brl) in enumerate(lists):
        assert util.rel_err(spec[j], oracle_eval(o, ops, pmi, brl, tree, subst[j], freqs[j], rates[j])) < LNL_TOL
    with pytest.raises(rd.RdamdError):
        a.set_rescale_speculation(2)
    a.destroy()
    o.destroy()


def test_sites_between_the_rules_line_and_the_checks_line():
    """200 tips of unrelated sequences: every site's likelihood is ~4^-200, far below the 2^-256
    where the reference's rule rescales (every rate of every site, more than once on the way up)
    and far above the 2^-900 of the root check: the speculative pass keeps every job -- its values
    are the rescaled ones up to the rounding of log(x 2^256) + 256 log 2 against log(x)"""
    w = synth.workload(200, 600, 4, 4, 611, simulate_seqs=False)
    tree = rd.Tree.from_newick(w["newick"])
    a, o = pair(tree, w["seqs"], 4)
    rng = np.random.default_rng(611)
    J = 3
    subst, freqs, rates = params(rng, J, 4)
    lists = [tree.generate_operations(tree.root_location(int(i)).with_ratio(0.5))
             for i in rng.choice(tree.root_count(), J, replace=False)]
    scheds = [a.schedule(*l) for l in lists]
    spec = a.evaluate_batch(scheds, subst, freqs, rates)
    assert a.second_passes() == 0
    a.set_rescale_speculation(0)
    plain = a.evaluate_batch(scheds, subst, freqs, rates)
    assert np.max(np.abs(spec - plain) / np.abs(plain)) < 1e-14
    for j, (ops, pmi, brl) in enumerate(lists):
        ref = oracle_eval(o, ops, pmi, brl, tree, subst[j], freqs[j], rates[j])
        assert util.rel_err(spec[j], ref) < LNL_TOL
        # (the oracle did rescale: per-site likelihoods below 2^-256 on average)
        assert ref / 600 < -256 * np.log(2.0)
    a.destroy()
    o.destroy()


@pytest.mark.parametrize("classes", [64, 16, 0])
def test_a_sum_below_the_line_sends_the_job_to_the_second_pass(classes):
    """branches of 1e-7 under unrelated sequences: every change costs 2^-23, a site of 100 tips
    ends near 2^-1700 -- beyond the FP64 range without the rule.  The speculative pass must not
    answer for such a job: its flag goes up, the second pass walks the plain program with every
    test, the oracle's value comes back.  Per JOB: the ordinary job beside it keeps its value to
    the bit, whether it shares the batch or runs alone."""
    w = synth.workload(100, 500, 4, 4, 621, simulate_seqs=False)
    tree = rd.Tree.from_newick(w["newick"])
    a, o = pair(tree, w["seqs"], 4, classes)
    rng = np.random.default_rng(621)
    subst, freqs, rates = params(rng, 3, 4)
    rl = tree.root_location(11).with_ratio(0.4)
    ops, pmi, brl = tree.generate_operations(rl)
    short = (ops, pmi, np.asarray(brl) * 0 + 1e-7)
    lists = [(ops, pmi, brl), short, (ops, pmi, brl)]
    scheds = [a.schedule(*l) for l in lists]
    got = a.evaluate_batch(scheds, subst, freqs, rates)
    assert a.second_passes() == 1
    refs = [oracle_eval(o, *l, tree, subst[j], freqs[j], rates[j]) for j, l in enumerate(lists)]
    assert refs[1] / 500 < -1022 * np.log(2.0)          # (the short job IS out of range without the rule)
    for j in range(3):
        assert util.rel_err(got[j], refs[j]) < LNL_TOL, j
    for j in (0, 2):
        alone = a.evaluate_batch([scheds[j]], subst[j:j + 1], freqs[j:j + 1], rates[j:j + 1])[0]
        assert alone == got[j]
    assert a.second_passes() == 1
    alone = a.evaluate_batch([scheds[1]], subst[1:2], freqs[1:2], rates[1:2])[0]
    assert alone == got[1] and a.second_passes() == 2
    # with the tests on every step the same job needs no second pass and agrees to rounding
    a.set_rescale_speculation(0)
    plain = a.evaluate_batch(scheds, subst, freqs, rates)
    assert a.second_passes() == 2
    # (not to the bit: unrelated sequences of 100 tips put some sites below 2^-256 in every rate;
    # there the rule's value is log(x 2^256) + 256 log 2^-1 and the speculative pass's log(x))
    assert np.max(np.abs(plain - got) / np.abs(got)) < 1e-13
    a.destroy()
    o.destroy()


def test_the_default_is_off_beyond_256_tips():
    """300 tips on branches of 1e-7: under speculation every job would take both passes; the
    default does not speculate there (and mode 1 does, with the same values to rounding)"""
    w = synth.workload(300, 200, 4, 4, 631, simulate_seqs=False)
    tree = rd.Tree.from_newick(w["newick"])
    a, o = pair(tree, w["seqs"], 4)
    rng = np.random.default_rng(631)
    subst, freqs, rates = params(rng, 2, 4)
    ops, pmi, brl = tree.generate_operations(tree.root_location(5).with_ratio(0.4))
    short = (ops, pmi, np.asarray(brl) * 0 + 1e-7)
    scheds = [a.schedule(*short), a.schedule(ops, pmi, brl)]
    got = a.evaluate_batch(scheds, subst, freqs, rates)
    assert a.second_passes() == 0
    a.set_rescale_speculation(1)
    spec = a.evaluate_batch(scheds, subst, freqs, rates)
    assert a.second_passes() == 1
    assert np.max(np.abs(spec - got) / np.abs(got)) < 1e-13
    assert util.rel_err(got[0], oracle_eval(o, *short, tree, subst[0], freqs[0], rates[0])) < LNL_TOL
    assert util.rel_err(spec[1], oracle_eval(o, ops, pmi, brl, tree, subst[1], freqs[1], rates[1])) < LNL_TOL
    a.destroy()
    o.destroy()


def test_the_flag_travels_with_a_stream_ordered_batch():
    """rdamd_evaluate_batch_submit_device: the speculative pass's flag is the batch's second-pass
    word -- up behind the finishing kernel, the flagged job final after the redo, the others at once"""
    rt = C.CDLL(rd.hip_runtime_path())
    rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    rt.hipFree.argtypes = [C.c_void_p]

    def read(ptr, n):
        out = np.zeros(n)
        assert rt.hipDeviceSynchronize() == 0
        assert rt.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), 8 * n, 2) == 0
        return out

    w = synth.workload(60, 400, 4, 4, 641, simulate_seqs=False)
    tree = rd.Tree.from_newick(w["newick"])
    a, o = pair(tree, w["seqs"], 4)
    rng = np.random.default_rng(641)
    subst, freqs, rates = params(rng, 3, 4)
    ops, pmi, brl = tree.generate_operations(tree.root_location(3).with_ratio(0.4))
    scheds = [a.schedule(ops, pmi, brl), a.schedule(ops, pmi, np.asarray(brl) * 0 + 1e-9), a.schedule(ops, pmi, brl)]
    want = a.evaluate_batch(scheds, subst, freqs, rates)
    assert a.second_passes() == 1
    d = C.c_void_p()
    assert rt.hipMalloc(C.byref(d), 8 * 4) == 0
    n = a.evaluate_batch_submit_device(0, scheds, subst, freqs, d.value, rates)
    got = read(d.value, n + 1)
    assert got[n] == 1.0 and got[0] == want[0] and got[2] == want[2]
    a.evaluate_batch_redo_device(0, d.value)
    got = read(d.value, n + 1)
    assert got[n] == 0.0 and np.array_equal(got[:n], want)
    a.evaluate_batch_finish_device(0)
    assert a.second_passes() == 2
    rt.hipFree(d)
    a.destroy()
    o.destroy()


def test_the_environment_sets_the_mode_of_partitions_created_under_it(monkeypatch):
    """RDAMD_RESCALE_SPECULATION: for the partitions a caller never sees (the model's, its replicas');
    observable through the second pass a 100-tip job on branches of 1e-7 needs under speculation only"""
    w = synth.workload(100, 300, 4, 4, 651, simulate_seqs=False)
    tree = rd.Tree.from_newick(w["newick"])
    rng = np.random.default_rng(651)
    subst, freqs, rates = params(rng, 1, 4)
    ops, pmi, brl = tree.generate_operations(tree.root_location(7).with_ratio(0.4))
    got = {}
    for env, passes in (("0", 0), ("1", 1), (None, 1)):
        if env is None:
            monkeypatch.delenv("RDAMD_RESCALE_SPECULATION", raising=False)
        else:
            monkeypatch.setenv("RDAMD_RESCALE_SPECULATION", env)
        a, o = pair(tree, w["seqs"], 4)
        got[env] = a.evaluate_batch([a.schedule(ops, pmi, np.asarray(brl) * 0 + 1e-7)], subst, freqs, rates)[0]
        assert a.second_passes() == passes, env
        a.destroy()
        o.destroy()
    assert got["1"] == got[None] and abs(got["0"] - got["1"]) <= 1e-13 * abs(got["1"])
