"""Parameter optimisation on the batched objective: the reference's own
L-BFGS-B (lib/lbfgsb, compiled from /root/reference into oracle/_ref/ by
`make -C oracle ref`; test infrastructure) drives model_t::optimize_params
through its reverse-communication interface, exactly as bfgs_params does
(src/model.cpp:1430-1522), with each objective + finite-difference gradient as
one fused launch.  The same procedure is replayed on the CPU oracle."""
import ctypes as C
import math
import os

import numpy as np
import pytest

import root_digger_amd as rd
from oracle_lib import OraclePartition, ORC_MAP_NT, orc_gamma_cats
import util

pytestmark = pytest.mark.gpu

REF = os.path.join(util.ROOT, "oracle", "_ref", "liblbfgsb_ref.so")
PARAMS3 = [.34, .42, .24, .74, .16, .88, .75, .54, .20, .06, .08, .41]


@pytest.fixture(scope="module")
def lbfgsb():
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/liblbfgsb_ref.so not built (needs /root/reference at build time)")
    return C.CDLL(REF)


def bfgs_on(objective, x0, lo, hi, eps, pgtol, factor, setulb):
    """bfgs_params (src/model.cpp:1430-1522) with a Python objective (-lnL)."""
    n = len(x0)
    m = 20
    x = np.array(x0, dtype=np.float64)
    l = np.full(n, lo)
    u = np.full(n, hi)
    nbd = np.full(n, 2, dtype=np.int32)
    g = np.zeros(n)
    wa = np.zeros((2 * m + 5) * n + 12 * m * (m + 1))
    iwa = np.zeros(3 * n, dtype=np.int32)
    task, csave, iprint = C.c_int(1), C.c_int(0), C.c_int(-1)
    lsave = (C.c_int * 4)()
    isave = (C.c_int * 44)()
    dsave = (C.c_double * 29)()
    f = C.c_double(objective(x))
    initial = f.value
    nn, mm = C.c_int(n), C.c_int(m)
    fac, pg = C.c_double(factor), C.c_double(pgtol)
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    for _ in range(500):
        setulb(C.byref(nn), C.byref(mm), P(x, C.c_double), P(l, C.c_double), P(u, C.c_double),
               P(nbd, C.c_int), C.byref(f), P(g, C.c_double), C.byref(fac), C.byref(pg),
               P(wa, C.c_double), P(iwa, C.c_int), C.byref(task), C.byref(iprint), C.byref(csave),
               lsave, isave, dsave)
        f.value = objective(x)
        if 10 <= task.value <= 15:
            for i in range(n):
                h = max(eps * abs(x[i]), eps)
                xi = x.copy()
                xi[i] += h
                g[i] = (objective(xi) - f.value) / h
        elif task.value != 2:
            break
    final = objective(x)
    return (x, final) if initial >= final else (np.array(x0, dtype=np.float64), final)


def test_optimize_params_with_reference_lbfgsb(lbfgsb):
    tree = rd.Tree.from_file(os.path.join(util.DATA, "10.tree"))
    seqs = util.read_fasta(os.path.join(util.DATA, "10.fasta"))
    R = 4
    m = rd.Model(tree, seqs, rate_cats=R, seed=5)
    m.initialize_partitions()
    m.set_lbfgsb(lbfgsb.setulb)
    rl = tree.root_location(3)
    subst0, freqs0 = [1.0 / 12] * 12, [0.25] * 4
    m.set_subst_rates(subst0)
    m.set_freqs(freqs0)
    m.set_gamma_alpha(1.0)
    before = m.compute_lh(rl)
    res = m.optimize_params(rl, subst0, freqs0, 1.0, pgtol=1e-5, factor=1e7)
    m.set_subst_rates(res["subst"])
    m.set_freqs(np.array(res["freqs"]) / np.sum(res["freqs"]))
    m.set_gamma_alpha(res["gamma_alpha"])
    after = m.compute_lh(rl)
    assert after > before + 10.0                      # a real improvement on real data
    assert res["evaluations"] > 3 * res["batches"]    # the gradients ride in the batches

    # the same three optimisations on the CPU oracle, same optimiser
    o = OraclePartition.for_tree(tree, 4, 1000, R)
    util.load_tips(o, tree, seqs, ORC_MAP_NT)
    t2 = rd.Tree.from_file(os.path.join(util.DATA, "10.tree"))
    ops, pmi, brl = t2.generate_operations(rl)
    state = {"subst": np.array(subst0), "freqs": np.array(freqs0), "alpha": 1.0}

    def lnl():
        o.set_subst_params(0, state["subst"])
        o.set_frequencies(0, state["freqs"] / state["freqs"].sum())
        o.set_category_rates(rd.compute_gamma_cats(state["alpha"], R, rd.GAMMA_RATES_MEDIAN))
        o.update_prob_matrices(pmi, brl)
        o.update_clvs(ops)
        return o.compute_root_loglikelihood(t2.root_clv_index(), t2.root_scaler_index())

    def obj(key):
        def f(x):
            old = state[key]
            state[key] = np.array(x) if key != "alpha" else float(x[0])
            v = -lnl()
            state[key] = old
            return v
        return f

    x, _ = bfgs_on(obj("subst"), state["subst"], 1e-4, 1e4, 1e-4, 1e-5, 1e7, lbfgsb.setulb)
    state["subst"] = x
    x, _ = bfgs_on(obj("freqs"), state["freqs"], 1e-4, 1 - 3e-4, 1e-4, 1e-5, 1e7, lbfgsb.setulb)
    state["freqs"] = x
    x, _ = bfgs_on(obj("alpha"), [state["alpha"]], 0.2, 1e4, 1e-4, 1e-5, 1e7, lbfgsb.setulb)
    state["alpha"] = float(x[0])
    cpu_after = lnl()
    # optimiser trajectories are sensitive to the last bits; the optimum is not
    assert abs(after - cpu_after) < 1e-4 * abs(cpu_after)
    assert np.allclose(res["subst"] / np.sum(res["subst"]), state["subst"] / state["subst"].sum(),
                       rtol=5e-2, atol=5e-3)


def test_exhaustive_search_with_parameter_optimisation(lbfgsb):
    """The full per-candidate loop of src/model.cpp:1139-1272 (optimize_params,
    optimize_alpha, stopping rules) on the reference's 10.fasta fixture."""
    tree = rd.Tree.from_file(os.path.join(util.DATA, "10.tree"))
    seqs = util.read_fasta(os.path.join(util.DATA, "10.fasta"))
    m = rd.Model(tree, seqs, rate_cats=1, seed=9, early_stop=True)
    m.initialize_partitions()
    m.compute_lh(tree.root_location(0))
    plain = m.exhaustive_search(1e-3, 1e-3, 1e-3, 1e12)
    m.set_lbfgsb(lbfgsb.setulb)
    opt = m.exhaustive_search(1e-3, 1e-3, 1e-3, 1e12)     # tolerances of test/src/model.cpp:400
    assert sorted(opt["root_id"]) == list(range(17))
    assert np.all(np.isfinite(opt["llh"]))
    # optimised parameters can only improve each candidate's likelihood
    by_id = dict(zip(plain["root_id"], plain["llh"]))
    for rid, llh in zip(opt["root_id"], opt["llh"]):
        assert llh >= by_id[rid] - 1e-6
    assert opt["best_llh"] > plain["best_llh"]


def test_parallel_exhaustive_search_equals_sequential(lbfgsb):
    """worker replicas (one partition + stream each) pull candidates from a shared
    counter; every candidate is computed by the same code on the same data, so the
    results are identical to the sequential loop."""
    tree = rd.Tree.from_file(os.path.join(util.DATA, "10.tree"))
    seqs = util.read_fasta(os.path.join(util.DATA, "10.fasta"))
    m = rd.Model(tree, seqs, rate_cats=4, seed=9)
    m.initialize_partitions()
    m.set_lbfgsb(lbfgsb.setulb)
    m.compute_lh(tree.root_location(0))
    seq = m.exhaustive_search(1e-3, 1e-3, 1e-3, 1e12)
    par = m.exhaustive_search(1e-3, 1e-3, 1e-3, 1e12, workers=4)
    order = np.argsort(seq["root_id"])
    assert par["root_id"] == sorted(seq["root_id"])
    assert np.array_equal(par["llh"], seq["llh"][order])
    assert np.array_equal(par["alpha"], seq["alpha"][order])
    assert par["best_llh"] == seq["best_llh"]
    # lock-stepped: the candidates' L-BFGS-B batches meet in one launch on the
    # shared partition (batch_combiner.hpp).  A job's result does not depend on
    # what else is in its launch, so the trajectories are the sequential ones.
    before = m.counters()["objective_batches"]
    for in_flight in (3, 17):
        lock = m.exhaustive_search(1e-3, 1e-3, 1e-3, 1e12, lockstep=in_flight)
        assert lock["root_id"] == sorted(seq["root_id"])
        assert np.array_equal(lock["llh"], seq["llh"][order])
        assert np.array_equal(lock["alpha"], seq["alpha"][order])
    assert m.counters()["objective_batches"] == before    # the replicas did the asking
    # ... and so do their root-only steps (Brent on the root position, compute_dlh's two
    # positions, optimize_alpha's scans): one launch over the replicas' partitions
    # (rdamd_root_loglikelihood_fused_multi) serves every candidate that is placing its root.
    # Same bits (asserted above).  How many steps share a launch depends on how many
    # candidates are placing their root at the same moment -- a short phase between long
    # parameter optimisations: ~3 of 17 here, 1.3 of 16 on c2.
    st = m.lockstep_stats()
    assert st["root_steps"] > 0 and st["objective_jobs"] > 0
    assert st["root_steps"] >= 2 * st["root_launches"], st


def test_lockstep_records_do_not_depend_on_the_launch_size(lbfgsb):
    """A job's value must not depend on the launch it rides in -- else the lock-stepped search,
    whose launches are as wide as the number of candidates in flight, would walk other
    optimiser trajectories than the sequential one.  Two things could make it: the kernel takes
    two sites per lane once a launch is big enough (here: from 26 jobs on; one candidate's
    batches have 17), and the fall-back to the plain programs is decided by a flag.  Both are
    per job now (one partial sum per 64-site block whatever the sites per lane;
    FusedJob::tt_unsafe per job): same records with 1, 3 and 8 candidates in flight, bit for bit,
    on an alignment long enough for the variants to differ."""
    from root_digger_amd import synth
    w = synth.workload(12, 20000, 4, 4, 977)
    tree = rd.Tree.from_newick(w["newick"])
    m = rd.Model(tree, w["seqs"], rate_cats=4, seed=5)
    m.initialize_partitions()
    m.set_lbfgsb(lbfgsb.setulb)
    m.compute_lh(tree.root_location(0))
    m._ok(rd.lib.rdamd_model_assign_by_rank(m._h, 0, 2), "assign")     # the first half of the 21 roots
    seq = m.exhaustive_search(1e-4, 1e-4, 1e-6, 1e9)
    order = np.argsort(seq["root_id"])
    for in_flight in (3, 8):
        lock = m.exhaustive_search(1e-4, 1e-4, 1e-6, 1e9, lockstep=in_flight)
        assert lock["root_id"] == sorted(seq["root_id"])
        assert np.array_equal(lock["llh"], seq["llh"][order])
        assert np.array_equal(lock["alpha"], seq["alpha"][order])
    st = m.lockstep_stats()
    # launches carried more than one candidate's 17 jobs on average (how many more depends on
    # how the candidates' threads meet: 24-30 measured), so launches of 26+ jobs did occur
    assert st["objective_jobs"] > st["objective_launches"], st


def test_batched_root_sweep_equals_move_root_sweep():
    """all 2n-3 root lnLs at fixed parameters: one fused launch vs the reference's
    move_root sweep (src/model.cpp:865-889, :1737-1746)."""
    tree = rd.Tree.from_file(os.path.join(util.DATA, "101.tree"))
    seqs, w = util.compress(util.read_phylip(os.path.join(util.DATA, "101.phy")))
    m = rd.Model(tree, seqs, rate_cats=4, weights=w, seed=4)
    m.initialize_partitions()
    m.set_gamma_alpha(0.6)
    a = m.compute_all_root_lh_batched()
    b = m.compute_all_root_lh()
    assert len(a) == 199
    assert np.max(np.abs(a - b) / np.abs(b)) < 1e-12
    # the batch leaves model state alone: the sweep state (last root) is still valid
    last = tree.root_location(198)
    assert util.rel_err(m.compute_lh_root(last), b[198]) < 1e-13
    # the all-directions CLV cache: 3(n-2) directed operations +
    # one root operation per branch on a partition of its own; same numbers
    c = m.compute_all_root_lh_directional()
    assert np.max(np.abs(c - b) / np.abs(b)) < 1e-12
    assert util.rel_err(m.compute_lh_root(last), b[198]) < 1e-13      # the model's own state is untouched
    # other root positions on every branch, and new parameters: the cache follows
    ratios = np.linspace(0.05, 0.95, 199)
    m.set_gamma_alpha(1.7)
    d = m.compute_all_root_lh_directional(ratios)
    want = np.array([m.compute_lh(tree.root_location(i).with_ratio(float(ratios[i]))) for i in (0, 57, 198)])
    assert np.max(np.abs(d[[0, 57, 198]] - want) / np.abs(want)) < 1e-12


def test_directional_cache_vs_oracle_all_199_roots():
    """SURVEY 8(f)2, the all-directions CLV cache, against the CPU oracle directly
    (not against this library's own move_root sweep): 101.phy (ambiguity codes,
    zero-length branches), all 199 candidate branches, two root-position vectors
    and two parameter sets; the oracle does a full traversal per root
    (model_t::compute_lh, src/model.cpp:384-413)."""
    tree = rd.Tree.from_file(os.path.join(util.DATA, "101.tree"))
    seqs, w = util.compress(util.read_phylip(os.path.join(util.DATA, "101.phy")))
    S = len(next(iter(seqs.values())))
    m = rd.Model(tree, seqs, rate_cats=4, weights=w, seed=4)
    m.initialize_partitions()
    o = OraclePartition.for_tree(tree, 4, S, 4)
    util.load_tips(o, tree, seqs, ORC_MAP_NT, w)
    freqs = o.empirical_frequencies()
    t2 = rd.Tree.from_file(os.path.join(util.DATA, "101.tree"))
    rng = np.random.default_rng(1012)
    cases = [(PARAMS3, 0.6, np.full(199, 0.5)),
             (list(rng.uniform(1e-4, 1.0, 12)), 1.7, rng.uniform(0.02, 0.98, 199))]
    for subst, alpha, ratios in cases:
        m.set_subst_rates(subst)
        m.set_empirical_freqs()
        m.set_gamma_alpha(alpha)
        o.set_subst_params(0, subst)
        o.set_frequencies(0, freqs)
        o.set_category_rates(orc_gamma_cats(alpha, 4, 1))   # model_t: MEDIAN mode after the first call
        got = m.compute_all_root_lh_directional(ratios)
        assert len(got) == 199
        want = np.array([util.compute_lh(o, t2, t2.root_location(i).with_ratio(float(ratios[i])))
                         for i in range(199)])
        assert np.max(np.abs(got - want) / np.abs(want)) < 1e-11
    o.destroy()


def test_batched_root_reduction_is_bit_identical_to_single_calls():
    """rdamd_compute_root_loglikelihoods: one launch for many root CLVs, each
    value exactly what rdamd_compute_root_loglikelihood returns for that CLV."""
    from root_digger_amd import synth
    # (20 states: CLVs in the matrix-core operand layout, ragged last tile, R = 3 and 4)
    for R, S, K in ((4, 5000, 4), (3, 777, 4), (1, 64, 4), (4, 203, 20), (3, 45, 20)):
        w = synth.workload(20, S, K, R, 31 + R)
        tree = rd.Tree.from_newick(w["newick"])
        d = tree.generate_directional_operations()
        p = rd.Partition(tips=20, clv_buffers=d["clv_buffers"], states=K, sites=S, rate_matrices=1,
                         prob_matrices=d["prob_matrices"], rate_cats=R,
                         scale_buffers=d["scale_buffers"])
        util.load_tips(p, tree, w["seqs"], rd.MAP_NT if K == 4 else util.make_map(w["alphabet"]))
        p.set_subst_params(0, w["subst"])
        p.set_frequencies(0, p.empirical_frequencies())
        p.set_category_rates(w["rates"])
        p.update_prob_matrices(d["matrix_indices"], d["branch_lengths"])
        p.update_clvs(d["ops"])
        many = p.compute_root_loglikelihoods(d["root_clv"], d["root_scaler"])
        for rid in range(tree.root_count()):
            one = p.compute_root_loglikelihood(int(d["root_clv"][rid]), int(d["root_scaler"][rid]))
            assert many[rid] == one
        p.destroy()


def test_heuristic_search(lbfgsb):
    """model_t::search (src/model.cpp:1008-1137) as test/src/model.cpp:310-346 checks
    it: the final lnL reproduces under compute_lh and beats the start."""
    tree = rd.Tree.from_file(os.path.join(util.DATA, "10.tree"))
    seqs = util.read_fasta(os.path.join(util.DATA, "10.fasta"))
    m = rd.Model(tree, seqs, rate_cats=1, seed=21, early_stop=True)
    m.initialize_partitions()
    m.set_lbfgsb(lbfgsb.setulb)
    m.set_subst_rates_uniform()
    m.set_empirical_freqs()
    initial = m.compute_lh(tree.root_location(0))
    m.assign_by_rank(0, 17)                       # one starting root, like min_roots = 1
    best, llh = m.search(1, 0.0, 1e-5, 1e-5, 1e-7, 1e7)
    assert llh >= initial
    assert 0.0 <= best.brlen_ratio <= 1.0
    assert abs(m.compute_lh(best) - llh) < 1e-6 * abs(llh)


def test_cli_exhaustive_with_parameter_optimisation(lbfgsb, tmp_path):
    """`rd --exhaustive` end to end with the reference's L-BFGS-B: the command
    line picks lock step by itself; its results equal the library's sequential
    loop on the same options."""
    from root_digger_amd import cli
    msa, tre = os.path.join(util.DATA, "10.fasta"), os.path.join(util.DATA, "10.tree")
    prefix = str(tmp_path / "opt")
    assert cli.main(["--msa", msa, "--tree", tre, "--prefix", prefix, "--exhaustive", "--silent",
                     "--lbfgsb", REF, "--rate-cats", "4", "--seed", "9", "--atol", "1e-3",
                     "--bfgstol", "1e-3", "--brtol", "1e-3", "--factor", "1e12"]) == 0
    got = {r: (l, a) for r, l, a, _ in rd.Checkpoint(prefix).read_results()}
    tree = rd.Tree.from_file(tre)
    m = rd.Model.from_file(tree, msa, rate_cats=4, seed=9)
    m.initialize_partitions()
    m.set_lbfgsb(lbfgsb.setulb)
    m.compute_lh(tree.root_location(0))
    seq = m.exhaustive_search(1e-3, 1e-3, 1e-3, 1e12)
    assert sorted(got) == sorted(seq["root_id"])
    for rid, llh, alpha in zip(seq["root_id"], seq["llh"], seq["alpha"]):
        assert got[rid] == (llh, alpha)
    lwr = open(prefix + ".lwr.tree").read()
    assert lwr.count("LWR=") == 17


def test_binary_characters_with_parameter_optimisation(lbfgsb, tmp_path):
    """`rd --states 2` with the full per-candidate loop: two substitution rates
    and two frequencies go through the same batched L-BFGS-B objective (the
    partition runs on the 4-state kernels); optimising can only help."""
    from root_digger_amd import synth
    w = synth.workload(12, 1500, 2, 4, 5)
    tree = rd.Tree.from_newick(w["newick"])
    cmap = util.make_map(w["alphabet"])
    m = rd.Model(tree, w["seqs"], states=2, cmap=cmap, rate_cats=4, seed=4, early_stop=True)
    m.initialize_partitions()
    m.compute_lh(tree.root_location(0))
    plain = m.exhaustive_search(1e-3, 1e-3, 1e-3, 1e12)
    m.set_lbfgsb(lbfgsb.setulb)
    opt = m.exhaustive_search(1e-3, 1e-3, 1e-3, 1e12)
    lock = m.exhaustive_search(1e-3, 1e-3, 1e-3, 1e12, lockstep=8)
    by_id = dict(zip(plain["root_id"], plain["llh"]))
    assert sorted(opt["root_id"]) == list(range(tree.root_count()))
    for rid, llh in zip(opt["root_id"], opt["llh"]):
        assert np.isfinite(llh) and llh >= by_id[rid] - 1e-6
    assert opt["best_llh"] > plain["best_llh"]
    order = np.argsort(opt["root_id"])
    assert np.array_equal(lock["llh"], opt["llh"][order])
    # the result log keeps the 2-state shapes
    ck = rd.Checkpoint(str(tmp_path / "bin"))
    ck.save_options({"data_type": "bin"})
    m.set_checkpoint(ck)
    m.assign_by_rank(0, tree.root_count())          # one candidate
    m.exhaustive_search(1e-3, 1e-3, 1e-3, 1e12)
    (rid, llh, alpha, params), = ck.read_results()
    assert len(params[0]["subst_rates"]) == 2 and len(params[0]["freqs"]) == 2


def test_replica_count_is_clamped_to_device_memory():
    """a replica of the parallel / lock-step search whose compute_lh is the full
    traversal is a full model (all CLV buffers); asking for more than fit must be clamped, not
    die in hipMalloc.  A 60-taxon x 400 000-site Γ4 model is 6.1 GB per such replica.  (With the
    children-only compute_lh, the default, a replica holds three CLVs: tests/test_gpu_sparse.py.)"""
    from root_digger_amd import synth
    w = synth.workload(60, 400000, 4, 4, 77, simulate_seqs=False)
    tree = rd.Tree.from_newick(w["newick"])
    m = rd.Model(tree, w["seqs"], rate_cats=4, seed=1)
    free, total = rd.device_memory()
    assert m.max_replicas(32)[0] == 32 and m.max_replicas(32)[1] < 0.4e9
    m.set_root_children_only(False)
    fit, per = m.max_replicas(1000)
    want = rd.lib.rdamd_partition_footprint(60, 118, 4, 400000, 118, 4, 118)
    assert per == want and 5.9e9 < per < 6.6e9
    assert 1 <= fit < 1000 and fit * per <= 0.85 * free + per
    assert m.max_replicas(2) == (2, per)
    # the search itself applies the clamp: 1000 workers requested, it runs (and says so)
    m.initialize_partitions()
    m.assign_by_rank(0, 60)                     # two candidates
    res = m.exhaustive_search(1e-2, 1e-2, 1e-2, 1e12, workers=1000)
    assert len(res["root_id"]) == 2 and np.all(np.isfinite(res["llh"]))
    m.destroy()


def _aa_map(alphabet):
    cmap = (C.c_uint64 * 256)()
    for i, ch in enumerate(alphabet):
        cmap[ord(ch)] = 1 << i
    return cmap


def test_optimize_params_20_states_with_reference_lbfgsb(lbfgsb):
    """the search on 20-state data.  The 381 evaluations of one L-BFGS-B
    iteration over a 20-state rate matrix (the objective + 380 finite differences,
    src/model.cpp:1490-1502) are ONE launch of fused20_eval_kernel; the same optimisation is
    replayed on the CPU oracle with the same setulb."""
    from root_digger_amd import synth
    R = 4
    w = synth.workload(9, 120, 20, R, 515)
    tree = rd.Tree.from_newick(w["newick"])
    cmap = _aa_map(w["alphabet"])
    m = rd.Model(tree, w["seqs"], states=20, cmap=cmap, rate_cats=R, seed=7)
    m.initialize_partitions()
    m.set_lbfgsb(lbfgsb.setulb)
    rl = tree.root_location(4).with_ratio(0.4)
    subst0, freqs0 = [1.0 / 380] * 380, [0.05] * 20
    m.set_subst_rates(subst0)
    m.set_freqs(freqs0)
    m.set_gamma_alpha(1.0)
    before = m.compute_lh(rl)
    res = m.optimize_params(rl, subst0, freqs0, 1.0, pgtol=1e-3, factor=1e9)
    assert res["evaluations"] >= 381 and res["evaluations"] > 100 * res["batches"] / 2   # 381-job batches
    m.set_subst_rates(res["subst"])
    m.set_freqs(np.array(res["freqs"]) / np.sum(res["freqs"]))
    m.set_gamma_alpha(res["gamma_alpha"])
    after = m.compute_lh(rl)
    assert after > before + 1.0

    o = OraclePartition.for_tree(tree, 20, 120, R)
    util.load_tips(o, tree, w["seqs"], cmap)
    t2 = rd.Tree.from_newick(w["newick"])
    ops, pmi, brl = t2.generate_operations(rl)
    state = {"subst": np.array(subst0), "freqs": np.array(freqs0), "alpha": 1.0}

    def lnl():
        o.set_subst_params(0, state["subst"])
        o.set_frequencies(0, state["freqs"] / state["freqs"].sum())
        o.set_category_rates(rd.compute_gamma_cats(state["alpha"], R, rd.GAMMA_RATES_MEDIAN))
        o.update_prob_matrices(pmi, brl)
        o.update_clvs(ops)
        return o.compute_root_loglikelihood(t2.root_clv_index(), t2.root_scaler_index())

    def obj(key):
        def f(x):
            old = state[key]
            state[key] = np.array(x) if key != "alpha" else float(x[0])
            v = -lnl()
            state[key] = old
            return v
        return f

    x, _ = bfgs_on(obj("subst"), state["subst"], 1e-4, 1e4, 1e-4, 1e-3, 1e9, lbfgsb.setulb)
    state["subst"] = x
    x, _ = bfgs_on(obj("freqs"), state["freqs"], 1e-4, 1 - 3e-4, 1e-4, 1e-3, 1e9, lbfgsb.setulb)
    state["freqs"] = x
    x, _ = bfgs_on(obj("alpha"), [state["alpha"]], 0.2, 1e4, 1e-4, 1e-3, 1e9, lbfgsb.setulb)
    state["alpha"] = float(x[0])
    cpu_after = lnl()
    assert abs(after - cpu_after) < 1e-3 * abs(cpu_after)


def test_exhaustive_search_on_20_states_lockstep_equals_sequential(lbfgsb):
    """... and the whole per-candidate loop (src/model.cpp:1139-1272) on a 20-state alignment:
    sequential, and with the candidates' 381-job batches meeting in one launch."""
    from root_digger_amd import synth
    w = synth.workload(7, 300, 20, 4, 516)
    tree = rd.Tree.from_newick(w["newick"])
    m = rd.Model(tree, w["seqs"], states=20, cmap=_aa_map(w["alphabet"]), rate_cats=4, seed=3)
    m.initialize_partitions()
    m.set_lbfgsb(lbfgsb.setulb)
    m.compute_lh(tree.root_location(0))
    m._ok(rd.lib.rdamd_model_assign_by_rank(m._h, 0, 2), "assign")     # half of the 11 roots
    seq = m.exhaustive_search(1e-2, 1e-2, 1e-3, 1e12)
    order = np.argsort(seq["root_id"])
    lock = m.exhaustive_search(1e-2, 1e-2, 1e-3, 1e12, lockstep=4)
    assert lock["root_id"] == sorted(seq["root_id"])
    assert np.array_equal(lock["llh"], seq["llh"][order])
    assert np.array_equal(lock["alpha"], seq["alpha"][order])
    assert np.all(np.isfinite(seq["llh"])) and np.all(seq["llh"] < 0)


@pytest.mark.parametrize("lines", ["UNREST+G4, a = 1-400\nUNREST+G4, b = 401-1000\n",
                                   "UNREST+G4, a = 1-300\nUNREST, b = 301-1000\n"])
def test_partitioned_model_searches_in_lock_step(lbfgsb, tmp_path, lines):
    """a partitioned model (src/main.cpp:512-555; the reference optimises
    its partitions side by side under OpenMP, src/model.cpp:1935) in the lock-stepped search:
    one batch combiner per partition, root-only steps of all partitions of all candidates in
    flight in one launch (per-partition fall-back where the rate categories differ).  Same
    records as the sequential loop, bit for bit."""
    pf = tmp_path / "parts.txt"
    pf.write_text(lines)
    tree = rd.Tree.from_file(os.path.join(util.DATA, "10.tree"))
    m = rd.Model.from_partition_file(tree, os.path.join(util.DATA, "10.fasta"), str(pf), seed=11)
    assert m.partition_count() == 2
    m.initialize_partitions()
    m.set_lbfgsb(lbfgsb.setulb)
    m.compute_lh(tree.root_location(0))
    seq = m.exhaustive_search(1e-3, 1e-3, 1e-3, 1e12)
    order = np.argsort(seq["root_id"])
    for in_flight in (3, 8):
        lock = m.exhaustive_search(1e-3, 1e-3, 1e-3, 1e12, lockstep=in_flight)
        assert lock["root_id"] == sorted(seq["root_id"])
        assert np.array_equal(lock["llh"], seq["llh"][order])
        assert np.array_equal(lock["alpha"], seq["alpha"][order])
    st = m.lockstep_stats()
    assert st["objective_jobs"] > st["objective_launches"] and st["root_steps"] > 0
