"""model_t surface (rows a4-a9, a16, a17 of SURVEY 8a) on the GPU: the
reference's own model tests (test/src/model.cpp) restated, plus parity of the
facade against the oracle-driven call sequences."""
import math
import os

import numpy as np
import pytest

import root_digger_amd as rd
from oracle_lib import OraclePartition, ORC_MAP_NT
import util

pytestmark = pytest.mark.gpu

PARAMS3 = [.34, .42, .24, .74, .16, .88, .75, .54, .20, .06, .08, .41]


def ten():
    tree = rd.Tree.from_file(os.path.join(util.DATA, "10.tree"))
    seqs = util.read_fasta(os.path.join(util.DATA, "10.fasta"))
    return tree, seqs


def test_compute_lh_finite_negative_repeatable():      # test/src/model.cpp:59-75
    tree, seqs = ten()
    m = rd.Model(tree, seqs, rate_cats=1, seed=7)
    m.initialize_partitions_uniform_freqs()
    for rl in tree.roots():
        lh = m.compute_lh(rl)
        assert math.isfinite(lh) and lh < 0.0
        assert lh == m.compute_lh(rl)


def test_compute_dlh_finite():                          # test/src/model.cpp:95-110
    tree, seqs = ten()
    m = rd.Model(tree, seqs, rate_cats=1, seed=7)
    m.initialize_partitions_uniform_freqs()
    for rl in tree.roots():
        m.compute_lh(rl)
        lh, dlh = m.compute_dlh(rl)
        assert math.isfinite(lh) and math.isfinite(dlh)


@pytest.mark.parametrize("start", [0.5, 0.0, 1.0])      # test/src/model.cpp:132-218
def test_optimize_alpha(start):
    tree, seqs = ten()
    m = rd.Model(tree, seqs, rate_cats=1, seed=7)
    m.initialize_partitions_uniform_freqs()
    for rl in tree.roots():
        rl = rl.with_ratio(start)
        m.compute_lh(rl)
        got = m.optimize_alpha(rl, 1e-7)
        assert 0.0 <= got.brlen_ratio <= 1.0
        assert got.edge == rl.edge
        # the optimum is at least as good as the start and as both ends
        best = m.compute_lh_root(got)
        for a in (start, 0.0, 1.0):
            assert best >= m.compute_lh_root(rl.with_ratio(a)) - 1e-6


def test_full_vs_root_only():                           # test/src/model.cpp:271-288
    tree, seqs = ten()
    m = rd.Model(tree, seqs, rate_cats=4, seed=3)
    m.initialize_partitions_uniform_freqs()
    for rl in tree.roots():
        lh1 = m.compute_lh(rl)
        assert abs(m.compute_lh_root(rl) - lh1) < 1e-9 * abs(lh1)


def test_facade_matches_oracle_sequences():
    tree, seqs = ten()
    m = rd.Model(tree, seqs, rate_cats=4, seed=3)
    m.initialize_partitions()
    m.set_subst_rates(PARAMS3)
    o = OraclePartition.for_tree(tree, 4, 1000, 4)
    util.load_tips(o, tree, seqs, ORC_MAP_NT)
    o.set_subst_params(0, PARAMS3)
    o.set_frequencies(0, o.empirical_frequencies())
    o.set_category_rates(rd.compute_gamma_cats(1.0, 4))
    t2 = rd.Tree.from_file(os.path.join(util.DATA, "10.tree"))
    for i in (0, 4, 11, 16):
        rl = tree.root_location(i).with_ratio(0.3)
        assert util.rel_err(m.compute_lh(rl), util.compute_lh(o, t2, rl)) < 1e-11
        # derivative: same one-sided difference as the reference (eps = 1e-8)
        lh, dlh = m.compute_dlh(rl)
        a = util.compute_lh_root(o, t2, rl)
        b = util.compute_lh_root(o, t2, rl.with_ratio(0.3 + 1e-8))
        assert util.rel_err(lh, a) < 1e-11
        assert abs(dlh - (b - a) / 1e-8) <= 2e-3 * max(1.0, abs(dlh))


def test_move_root_jc_invariance():                     # test/src/model.cpp:367-387
    tree = rd.Tree.from_file(os.path.join(util.DATA, "101.tree"))
    seqs, weights = util.compress(util.read_phylip(os.path.join(util.DATA, "101.phy")))
    m = rd.Model(tree, seqs, rate_cats=1, weights=weights, seed=5)
    m.initialize_partitions()
    m.set_subst_rates([1.0] * 12)
    m.set_freqs([0.25] * 4)
    m.compute_lh(tree.root_location(0))
    lhs = m.compute_all_root_lh()
    assert len(lhs) == 199
    assert np.max(np.abs(lhs - lhs[0])) < 1.2e-5 * abs(lhs[0])      # Catch Approx default


def test_compute_lh_batch_matches_single_calls():
    tree, seqs = ten()
    m = rd.Model(tree, seqs, rate_cats=4, seed=3)
    m.initialize_partitions()
    rng = np.random.default_rng(3)
    rls = [tree.root_location(int(i)).with_ratio(float(a))
           for i, a in zip(rng.choice(17, 6, replace=False), rng.uniform(0.1, 0.9, 6))]
    subst = rng.uniform(1e-4, 1, (6, 12))
    freqs = rng.dirichlet(np.ones(4) * 4, 6)
    alphas = rng.uniform(0.3, 3.0, 6)
    got = m.compute_lh_batch(rls, subst, freqs, alphas)
    for j, rl in enumerate(rls):
        m.set_subst_rates(subst[j])
        m.set_freqs(freqs[j])
        m.set_gamma_alpha(alphas[j])
        assert util.rel_err(got[j], m.compute_lh(rl)) < 1e-12


def test_exhaustive_search_runs_and_ranks_roots():       # test/src/model.cpp:389-401
    tree, seqs = ten()
    m = rd.Model(tree, seqs, rate_cats=1, seed=11)
    m.initialize_partitions()
    m.compute_lh(tree.root_location(0))
    res = m.exhaustive_search(1e-3, 1e-3, 1e-3, 1e12)
    assert sorted(res["root_id"]) == list(range(17))
    assert np.all(np.isfinite(res["llh"])) and np.all(res["llh"] < 0)
    assert np.all((res["alpha"] >= 0) & (res["alpha"] <= 1))
    assert res["best_llh"] == res["llh"].max()
    # every reported (root, alpha) reproduces its lnL under the search's model
    m.set_subst_rates_uniform()
    m.set_empirical_freqs()
    for rid, llh, a in zip(res["root_id"], res["llh"], res["alpha"]):
        rl = tree.root_location(rid).with_ratio(float(a))
        assert util.rel_err(m.compute_lh(rl), llh) < 1e-9
    # rank split covers the same roots (src/model.cpp:1867-1911)
    seen = []
    for rank in range(3):
        m.assign_by_rank(rank, 3)
        seen += m.exhaustive_search(1e-3, 1e-3, 1e-3, 1e12)["root_id"]
    assert sorted(seen) == list(range(17))


def test_msa_ingest_and_pattern_compression():
    """msa_t(filename) (src/msa.hpp:23-37): FASTA and PHYLIP fixtures of the
    reference, compressed vs uncompressed lnL identical."""
    tree = rd.Tree.from_file(os.path.join(util.DATA, "10.tree"))
    fasta = os.path.join(util.DATA, "10.fasta")
    mc = rd.Model.from_file(tree, fasta, rate_cats=4, seed=2, compress=True)
    mu = rd.Model.from_file(tree, fasta, rate_cats=4, seed=2, compress=False)
    assert mc.patterns == 991 and mu.patterns == 1000      # 991 patterns
    for m in (mc, mu):
        m.initialize_partitions()
        m.set_subst_rates(PARAMS3)
    for i in (0, 9, 16):
        rl = tree.root_location(i)
        assert util.rel_err(mc.compute_lh(rl), mu.compute_lh(rl)) < 1e-12
    # against the Python-side reader used by the other tests
    mp = rd.Model(tree, util.read_fasta(fasta), rate_cats=4, seed=2)
    mp.initialize_partitions()
    mp.set_subst_rates(PARAMS3)
    assert util.rel_err(mp.compute_lh(tree.root_location(4)), mu.compute_lh(tree.root_location(4))) < 1e-13
    # PHYLIP with ambiguity codes
    t101 = rd.Tree.from_file(os.path.join(util.DATA, "101.tree"))
    m101 = rd.Model.from_file(t101, os.path.join(util.DATA, "101.phy"), rate_cats=1, seed=2)
    seqs, w = util.compress(util.read_phylip(os.path.join(util.DATA, "101.phy")))
    assert m101.patterns <= len(w)          # merging N/X/-/? can only fuse more columns
    m101.initialize_partitions()
    m101.set_subst_rates([1.0] * 12)
    m101.set_freqs([0.25] * 4)
    ref = rd.Model(t101, seqs, rate_cats=1, weights=w, seed=2)
    ref.initialize_partitions()
    ref.set_subst_rates([1.0] * 12)
    ref.set_freqs([0.25] * 4)
    rl = t101.root_location(7)
    assert util.rel_err(m101.compute_lh(rl), ref.compute_lh(rl)) < 1e-12
    single = rd.Tree.from_file(os.path.join(util.DATA, "single.tree"))
    ms = rd.Model.from_file(single, os.path.join(util.DATA, "single.phy"))
    assert ms.patterns == 1
    with pytest.raises(rd.RdamdError):
        rd.Model.from_file(tree, os.path.join(util.DATA, "101.phy"))     # taxa mismatch
    with pytest.raises(rd.RdamdError):
        rd.Model.from_file(tree, "/nonexistent.fasta")


def test_cli_exhaustive_outputs(tmp_path):
    """`rd --exhaustive` outputs (src/main.cpp:636-654): .lwr.tree with LWR / LLH /
    alpha on every branch, .rooted.tree rooted at the best placement."""
    from root_digger_amd import cli
    prefix = str(tmp_path / "ten")
    rc = cli.main(["--msa", os.path.join(util.DATA, "10.fasta"),
                   "--tree", os.path.join(util.DATA, "10.tree"), "--prefix", prefix,
                   "--atol", "1e-3", "--brtol", "1e-3", "--silent", "--exhaustive"])
    assert rc == 0
    lwr = open(prefix + ".lwr.tree").read()
    rooted = open(prefix + ".rooted.tree").read()
    assert lwr.count("LWR=") == 17 and lwr.count("LLH=") == 17 and lwr.count("alpha=") == 17
    import re
    w = [float(x) for x in re.findall(r"LWR=([0-9.]+)", lwr)]
    assert abs(sum(w) - 1.0) < 1e-4                # one annotation per branch, weights sum to 1
    t = rd.Tree.from_newick(rooted)                # a binary-rooted tree parses (and unroots)
    assert t.tip_count() == 10 and "NHX" not in rooted


def test_cli_checkpoint_resume_and_two_ranks(tmp_path):
    """The <prefix>.ckp control flow of the reference's main (src/main.cpp:366-460,
    :612-635): a finished run leaves every candidate in the log; a rerun finds
    nothing left to do and reproduces the trees from the log alone; a log cut
    short is repaired and only the missing candidates are recomputed; two
    processes (RANK/WORLD_SIZE, as torch.distributed.run sets them) split the
    candidates like the reference's MPI ranks and meet in the same file."""
    import subprocess
    import sys
    from root_digger_amd import cli
    msa, tre = os.path.join(util.DATA, "10.fasta"), os.path.join(util.DATA, "10.tree")
    base = ["--msa", msa, "--tree", tre, "--atol", "1e-3", "--brtol", "1e-3", "--silent",
            "--exhaustive"]
    one = str(tmp_path / "one")
    assert cli.main(base + ["--prefix", one]) == 0
    ck = rd.Checkpoint(one)
    first = sorted(ck.read_results())
    assert [r[0] for r in first] == list(range(17))
    assert ck.load_options()["exhaustive"] == 1 and ck.load_options()["msa_filename"] == msa
    lwr = open(one + ".lwr.tree").read()

    os.remove(one + ".lwr.tree")                       # rerun: all work is in the log
    assert cli.main(["--msa", "ignored", "--tree", "ignored", "--prefix", one, "--silent"]) == 0
    assert open(one + ".lwr.tree").read() == lwr
    assert sorted(rd.Checkpoint(one).read_results()) == first

    raw = open(one + ".ckp", "rb").read()              # tear the last record
    open(one + ".ckp", "wb").write(raw[:-9])
    assert rd.Checkpoint(one).needs_cleaning()
    assert cli.main(base + ["--prefix", one]) == 0
    again = rd.Checkpoint(one)
    assert not again.needs_cleaning()
    got = sorted(again.read_results())
    assert [r[0] for r in got] == list(range(17))
    for a, b in zip(got, first):                       # the recomputed one matches the first run
        assert a[0] == b[0] and abs(a[1] - b[1]) < 1e-6 * abs(b[1])

    two = str(tmp_path / "two")                        # two ranks, one GPU, one shared log
    env = dict(os.environ, WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533",
               PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    procs = [subprocess.Popen([sys.executable, "-m", "root_digger_amd.cli"] + base +
                              ["--prefix", two, "--device", "0", "--workers", "0"],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)))
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    both = sorted(rd.Checkpoint(two).read_results())
    assert [r[0] for r in both] == list(range(17))
    for a, b in zip(both, first):
        assert abs(a[1] - b[1]) < 1e-6 * abs(b[1]) and abs(a[2] - b[2]) < 1e-3
    import re
    w = [float(x) for x in re.findall(r"LWR=([0-9.]+)", open(two + ".lwr.tree").read())]
    assert len(w) == 17 and abs(sum(w) - 1.0) < 1e-4


def test_cli_default_mode_is_the_heuristic_search(tmp_path):
    """`rd --msa M --tree T` without --exhaustive runs search() from the starting
    roots of the chosen strategy (src/main.cpp:586-611, src/model.cpp:1809-1865)
    and writes only <prefix>.rooted.tree."""
    from root_digger_amd import cli
    ref = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "_ref",
                       "liblbfgsb_ref.so")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/liblbfgsb_ref.so not built")
    msa, tre = os.path.join(util.DATA, "10.fasta"), os.path.join(util.DATA, "10.tree")
    prefix = str(tmp_path / "s")
    args = ["--msa", msa, "--tree", tre, "--prefix", prefix, "--silent", "--lbfgsb", ref,
            "--atol", "1e-3", "--bfgstol", "1e-3", "--brtol", "1e-3", "--factor", "1e12",
            "--min-roots", "2"]
    assert cli.main(args + ["--initial-root-strategy", "midpoint"]) == 0
    assert not os.path.exists(prefix + ".lwr.tree")
    rooted = open(prefix + ".rooted.tree").read()
    assert rd.Tree.from_newick(rooted).tip_count() == 10
    ck = rd.Checkpoint(prefix)
    done = ck.read_results()
    assert len(done) == 2                                   # one record per starting root
    opts = ck.load_options()
    assert opts["exhaustive"] == 0 and opts["min_roots"] == 2 and opts["initial_root_strategy"] == 1
    assert all(len(p[0]["subst_rates"]) == 12 for _, _, _, p in done)
    # the starting roots were the two best midpoint branches
    tree = rd.Tree.from_file(tre)
    m = rd.Model.from_file(tree, msa, rate_cats=1, seed=1)
    m.initialize_partitions()
    m.assign_by_rank_search(2, 0.01, 0, 1, "midpoint")
    assert m.assigned() == tree.rank_midpoints()[:2]
    m.assign_by_rank_search(2, 0.01, 1, 2, "modified_mad")
    assert m.assigned() == tree.rank_modified_mad()[1:2]
    m.assign_by_rank_search(3, 0.01, 0, 1, "random")
    assert len(set(m.assigned())) == 3


def test_partitioned_model_matches_the_sum_of_its_parts(tmp_path):
    """src/main.cpp:512-555: a partition file cuts the alignment into model
    partitions (own pattern compression, own rate categories); the lnL of the
    partitioned model is the sum over partitions (src/model.cpp:396-411)."""
    from root_digger_amd import cli
    phy, tre = os.path.join(util.DATA, "101.phy"), os.path.join(util.DATA, "101.tree")
    pf = tmp_path / "parts.txt"
    pf.write_text("UNREST+G4, first = 1-300\nUNREST, second = 301-700, 900-1000\n")
    tree = rd.Tree.from_file(tre)
    m = rd.Model.from_partition_file(tree, phy, str(pf), seed=3)
    assert m.partition_count() == 2
    m.initialize_partitions_uniform_freqs()
    m.set_subst_rates_uniform()        # (initialisation draws a random rate set per partition)
    rl = tree.root_location(11).with_ratio(0.3)
    total = m.compute_lh(rl)
    assert abs(m.compute_lh_root(rl) - total) < 1e-9 * abs(total)
    # the same two partitions as separate single-partition models over column slices
    seqs = util.read_phylip(phy)
    parts = [({k: v[0:300] for k, v in seqs.items()}, 4),
             ({k: v[300:700] + v[899:1000] for k, v in seqs.items()}, 1)]
    acc = 0.0
    for sub, cats in parts:
        packed, weights = util.compress(sub)
        t2 = rd.Tree.from_file(tre)
        one = rd.Model(t2, packed, rate_cats=cats, weights=weights, seed=3)
        one.initialize_partitions_uniform_freqs()
        one.set_subst_rates_uniform()
        acc += one.compute_lh(t2.root_location(11).with_ratio(0.3))
    assert abs(total - acc) < 1e-10 * abs(acc)
    # and through the command line (root placement only: no --lbfgsb, exhaustive)
    prefix = str(tmp_path / "p")
    assert cli.main(["--msa", phy, "--tree", tre, "--partition", str(pf), "--prefix", prefix,
                     "--exhaustive", "--atol", "1e-2", "--brtol", "1e-2", "--silent"]) == 0
    ck = rd.Checkpoint(prefix)
    recs = ck.read_results()
    assert len(recs) == 199 and all(len(p) == 2 for _, _, _, p in recs)
    assert ck.load_options()["partition_filename"] == str(pf)


@pytest.mark.parametrize("dummy", [0, 1, 2, 4, 8])
def test_assign_indicies_with_a_checkpoint(tmp_path, dummy):
    """The reference's "assign indicies test" (test/src/model.cpp:448-549): roots
    already in the checkpoint are never assigned again -- search mode with each
    starting-root strategy and exhaustive mode."""
    tree = rd.Tree.from_file(os.path.join(util.DATA, "10.tree"))
    m = rd.Model.from_file(tree, os.path.join(util.DATA, "10.fasta"), rate_cats=1, seed=12345)
    m.initialize_partitions_uniform_freqs()
    ckp = rd.Checkpoint(str(tmp_path / "dummy"))
    ckp.save_options({})
    done = [int(i) for i in np.random.default_rng(dummy).permutation(17)[:dummy]]
    for rid in done:
        ckp.write(rid, 0.0, 0.0, [])
    for strategy in ("random", "midpoint", "modified_mad"):
        for want in (1, 2, 3, 4, 5):
            if want - dummy >= 0:
                m.assign_by_rank_search(want, 0.0, 0, 1, strategy, ckp)
                got = m.assigned()
                assert len(got) == want - dummy and not set(got) & set(done)
        if 1 - dummy < 0:
            with pytest.raises(rd.RdamdError):
                m.assign_by_rank_search(1, 0.0, 0, 1, strategy, ckp)
    m.assign_by_rank(0, 1, ckp)
    got = m.assigned()
    assert len(got) == 17 - dummy and not set(got) & set(done)
    # several ranks: the unfinished roots are split without overlap or loss
    pieces = []
    for rank in range(3):
        m.assign_by_rank(rank, 3, ckp)
        pieces.append(m.assigned())
    assert sorted(sum(pieces, [])) == sorted(set(range(17)) - set(done))


def test_cli_binary_characters(tmp_path):
    """--states 2 (src/main.cpp:484-488): binary alignments go through the
    generic K-state kernels; exhaustive root placement at fixed parameters."""
    from root_digger_amd import cli
    rng = np.random.default_rng(5)
    tree_nwk = open(os.path.join(util.DATA, "10.tree")).read()
    names = sorted(rd.Tree.from_newick(tree_nwk).label_map())
    fa = tmp_path / "bin.fasta"
    fa.write_text("".join(">%s\n%s\n" % (n, "".join(rng.choice(list("01-"), 200, p=[.45, .45, .1])))
                          for n in names))
    tr = tmp_path / "t.nwk"
    tr.write_text(tree_nwk)
    prefix = str(tmp_path / "b")
    assert cli.main(["--msa", str(fa), "--tree", str(tr), "--prefix", prefix, "--states", "2",
                     "--exhaustive", "--atol", "1e-3", "--brtol", "1e-3", "--silent"]) == 0
    recs = rd.Checkpoint(prefix).read_results()
    assert len(recs) == 17 and all(np.isfinite(l) and l < 0 for _, l, _, _ in recs)
    assert all(len(p[0]["subst_rates"]) == 2 and len(p[0]["freqs"]) == 2 for _, _, _, p in recs)


@pytest.mark.parametrize("kind", ["median", "free"])
def test_rate_category_types(kind, tmp_path):
    """`rd --rate-cats 4 --rate-cats-type {median,free}` (src/main.cpp:256-266):
    the rate-heterogeneity option reaches the model; the run completes and the
    likelihood responds to it."""
    from root_digger_amd import cli
    msa, tre = os.path.join(util.DATA, "10.fasta"), os.path.join(util.DATA, "10.tree")
    tree = rd.Tree.from_file(tre)
    base = rd.Model.from_file(tree, msa, rate_cats=4, seed=3)
    base.initialize_partitions_uniform_freqs()
    base.set_subst_rates_uniform()
    other = rd.Model.from_file(tree, msa, rate_cats=4, seed=3, rate_category_type=kind)
    other.initialize_partitions_uniform_freqs()
    other.set_subst_rates_uniform()
    rl = tree.root_location(4)
    a, b = base.compute_lh(rl), other.compute_lh(rl)
    assert np.isfinite(a) and np.isfinite(b)
    prefix = str(tmp_path / kind)
    assert cli.main(["--msa", msa, "--tree", tre, "--prefix", prefix, "--exhaustive", "--silent",
                     "--rate-cats", "4", "--rate-cats-type", kind, "--atol", "1e-3",
                     "--brtol", "1e-3", "--echo"]) == 0
    recs = rd.Checkpoint(prefix).read_results()
    assert len(recs) == 17
    if kind == "free":
        assert all(len(p[0]["gamma_weights"]) == 4 and len(p[0]["gamma_alpha"]) == 4
                   for _, _, _, p in recs)


def test_no_device_memory_leak_over_object_lifetimes():
    """partitions, schedules, models, replicas and checkpoints give back what they
    took: free device memory is where it started after many create/destroy rounds."""
    tree = rd.Tree.from_file(os.path.join(util.DATA, "10.tree"))
    msa = os.path.join(util.DATA, "10.fasta")

    def round_trip():
        p = rd.Partition.for_tree(tree, 4, 20000, 4)
        sc = p.schedule(*tree.generate_operations(tree.root_location(3)))
        p.evaluate_batch([sc] * 8, np.ones((8, 12)), np.full((8, 4), 0.25))
        sc.destroy()
        p.destroy()
        m = rd.Model.from_file(tree, msa, rate_cats=4, seed=2)
        m.initialize_partitions()
        m.compute_lh(tree.root_location(0))
        m.exhaustive_search(1e-2, 1e-2, 1e-2, 1e12, workers=3)
        m.destroy()

    round_trip()                                   # warm allocator pools, code objects
    free0, _ = rd.device_memory()
    for _ in range(12):
        round_trip()
    free1, _ = rd.device_memory()
    assert free0 - free1 < 32 << 20, (free0, free1)
