"""BASELINE.json configs c4 (500 taxa x 500 000 sites, site blocks over 8 GPUs)
and c5 (1000 taxa x 100 000 sites, candidate edges x site blocks) under -m gpu.

Each config is covered twice: at an oracle-sized site count against the CPU
oracle (materialising `rdamd_update_clvs` path: CLVs 1e-12, scalers bit-exact;
fused `rdamd_evaluate_batch` path: lnL 1e-11), and at its FULL size through
size-independent properties -- determinism, the materialising path as a second
implementation, root invariance under a reversible model
(/root/reference/test/src/model.cpp:381-386), and additivity over the site
blocks of the multi-GPU layout (8 blocks for c4, 4 candidate groups x 2 site
blocks for c5) emulated on one device with root_digger_amd.dist's own split."""
import numpy as np
import pytest

import root_digger_amd as rd
from root_digger_amd import synth, dist as rdist
from oracle_lib import OraclePartition, ORC_MAP_NT
import util

pytestmark = pytest.mark.gpu
LNL_TOL = 1e-11
SEED = 0xD166E5          # bench.py: seed = SEED + index of the config


def _pair(tree, seqs, R):
    S = len(next(iter(seqs.values())))
    g = rd.Partition.for_tree(tree, 4, S, R)
    o = OraclePartition.for_tree(tree, 4, S, R)
    util.load_tips(g, tree, seqs, rd.MAP_NT)
    util.load_tips(o, tree, seqs, ORC_MAP_NT)
    return g, o


def _set(parts, subst, freqs, rates):
    for p in parts:
        p.set_subst_params(0, subst)
        p.set_frequencies(0, freqs)
        p.set_category_rates(rates)


@pytest.mark.parametrize("n,S,seed", [(500, 2000, 81), (1000, 1500, 82)])
def test_c4_c5_tree_sizes_vs_oracle(n, S, seed):
    """c4's / c5's tree (500 / 1000 taxa, Γ4) at a site count the oracle finishes
    in seconds: every CLV and scaler of a full traversal, lnL through the
    materialising path and through the fused evaluator with per-job parameters."""
    w = synth.workload(n, S, 4, 4, seed)
    tree = rd.Tree.from_newick(w["newick"])
    assert tree.root_count() == 2 * n - 3
    g, o = _pair(tree, w["seqs"], 4)
    freqs = g.empirical_frequencies()
    g.set_category_rates(w["rates"])      # (the batch takes the partition's category rates)
    rng = np.random.default_rng(seed)
    ids = rng.choice(tree.root_count(), 4, replace=False)
    rls = [tree.root_location(int(i)).with_ratio(float(a)) for i, a in zip(ids, rng.uniform(.05, .95, 4))]
    subst = rng.uniform(1e-4, 1.0, (4, 12))
    scheds = [g.schedule(*tree.generate_operations(rl)) for rl in rls]
    fused = g.evaluate_batch(scheds, subst, [freqs] * 4)
    # ... and with subtree site repeats (clade tables) on a partition of its own
    gr = rd.Partition.for_tree(tree, 4, S, 4, attributes=rd.ATTRIB_SITE_REPEATS)
    util.load_tips(gr, tree, w["seqs"], rd.MAP_NT)
    gr.set_category_rates(w["rates"])
    sr = [gr.schedule(*tree.generate_operations(rl)) for rl in rls]
    assert min(x.stats()["clade_nodes"] for x in sr) > n // 5
    folded = gr.evaluate_batch(sr, subst, [freqs] * 4)
    for j, rl in enumerate(rls):
        _set((g, o), subst[j], freqs, w["rates"])
        want = util.compute_lh(o, tree, rl)
        assert util.rel_err(util.compute_lh(g, tree, rl), want) < LNL_TOL
        assert util.rel_err(fused[j], want) < LNL_TOL
        assert util.rel_err(folded[j], want) < LNL_TOL
    del sr
    gr.destroy()
    ops, _, _ = tree.generate_operations(rls[-1])          # state left by the last job
    for op in ops[::7] + [ops[-1]]:
        a, b = g.get_clv(op.parent_clv_index), o.get_clv(op.parent_clv_index)
        assert np.allclose(a, b, rtol=1e-12, atol=0.0)
        assert np.array_equal(g.get_scaler(op.parent_scaler_index),
                              o.get_scaler(op.parent_scaler_index))
    g.destroy()
    o.destroy()


def _full(config_index, n, S):
    w = synth.workload(n, S, 4, 4, SEED + config_index)
    tree = rd.Tree.from_newick(w["newick"])
    g = rd.Partition.for_tree(tree, 4, S, 4)
    util.load_tips(g, tree, w["seqs"], rd.MAP_NT)
    g.set_category_rates(w["rates"])
    return w, tree, g


@pytest.fixture(scope="module")
def c4_full():
    """BASELINE config c4 at full size (bench.py --config c4's workload): 64 GB
    of CLV buffers for the materialising path, none used by the fused one."""
    w, tree, g = _full(3, 500, 500000)
    yield w, tree, g
    g.destroy()


@pytest.fixture(scope="module")
def c5_full():
    w, tree, g = _full(4, 1000, 100000)
    yield w, tree, g
    g.destroy()


def _slice_partition(w, tree, lo, hi):
    # (the site blocks run WITH subtree site repeats, the reference's configuration for
    # 4-state data -- src/model.cpp:145-149 --, the whole-alignment partition they are summed
    # against without: the additivity checks below are also repeats-vs-plain checks at full size)
    part = rd.Partition.for_tree(tree, 4, hi - lo, 4, attributes=rd.ATTRIB_SITE_REPEATS)
    util.load_tips(part, tree, {k: v[lo:hi] for k, v in w["seqs"].items()}, rd.MAP_NT)
    part.set_category_rates(w["rates"])
    return part


def _properties(w, tree, g, n_jobs, rng):
    S, nroots = g.sites, tree.root_count()
    freqs = np.array(g.empirical_frequencies())
    ids = rng.choice(nroots, n_jobs, replace=False)
    rls = [tree.root_location(int(i)).with_ratio(float(a))
           for i, a in zip(ids, rng.uniform(.05, .95, n_jobs))]
    scheds = [g.schedule(*tree.generate_operations(rl)) for rl in rls]
    subst = rng.uniform(1e-4, 1.0, (n_jobs, 12))
    fb = np.tile(freqs, (n_jobs, 1))
    got = g.evaluate_batch(scheds, subst, fb)
    assert np.all(np.isfinite(got)) and np.all(got < 0) and len(set(got)) == n_jobs
    assert np.array_equal(got, g.evaluate_batch(scheds, subst, fb))          # determinism
    # the materialising drop-in path (oracle-checked at small S) on the same job
    _set((g,), subst[0], freqs, w["rates"])
    full = util.compute_lh(g, tree, rls[0])
    assert util.rel_err(got[0], full) < 1e-12
    assert util.rel_err(util.compute_lh_root(g, tree, rls[0]), full) < 1e-13
    _, ps = g.compute_root_loglikelihood(tree.root_clv_index(), tree.root_scaler_index(), persite=True)
    assert len(ps) == S and util.rel_err(float(np.sum(ps)), full) < 1e-12   # checksum of checksums
    # reversible model: every rooting has the same lnL (pulley principle)
    sample = [tree.root_location(int(i)) for i in rng.choice(nroots, 16, replace=False)]
    jc_s = [g.schedule(*tree.generate_operations(rl)) for rl in sample]
    jc = g.evaluate_batch(jc_s, np.ones((16, 12)), np.full((16, 4), 0.25))
    assert np.max(np.abs(jc - jc[0])) < 1e-9 * abs(jc[0])
    return rls, subst, freqs, got


def test_c4_full_size_properties_and_8_way_site_blocks(c4_full):
    """Full c4.  The 8-GPU layout of BASELINE config c4 -- 8 contiguous site
    blocks, per-block lnLs summed (the RCCL all-reduce) -- is emulated with 8
    partitions on this device: the block lnLs must add up to the one-partition
    value, and the blocks' empirical frequencies must combine to the global
    vector the way dist.global_frequencies combines them."""
    w, tree, g = c4_full
    rng = np.random.default_rng(404)
    rls, subst, freqs, whole = _properties(w, tree, g, 6, rng)
    total = np.zeros(len(rls))
    fsum, wsum = np.zeros(4), 0.0
    for r in range(8):
        lo, hi = rdist.site_block(g.sites, r, 8)
        assert hi - lo == 62500
        part = _slice_partition(w, tree, lo, hi)
        fsum += np.array(part.empirical_frequencies()) * (hi - lo)
        wsum += hi - lo
        sch = [part.schedule(*tree.generate_operations(rl)) for rl in rls]
        total += part.evaluate_batch(sch, subst, np.tile(freqs, (len(rls), 1)))
        del sch
        part.destroy()
    assert np.allclose(fsum / wsum, freqs, rtol=1e-13, atol=0)
    assert np.max(np.abs(total - whole) / np.abs(whole)) < 1e-12
    # The whole alignment WITH subtree site repeats: its code arena (16-bit entries, 500 tips +
    # the pseudo-tips' rows, 1 MB each) passes 512 MB, so these launches run with one wave per
    # rate category (kernels_fused.hip, RW) -- same values as the plain partition's
    gr = rd.Partition.for_tree(tree, 4, g.sites, 4, attributes=rd.ATTRIB_SITE_REPEATS)
    util.load_tips(gr, tree, w["seqs"], rd.MAP_NT)
    gr.set_category_rates(w["rates"])
    sr = [gr.schedule(*tree.generate_operations(rl)) for rl in rls]
    assert min(x.stats()["pseudo_tips"] for x in sr) >= 100
    folded = gr.evaluate_batch(sr, subst, np.tile(freqs, (len(rls), 1)))
    assert np.max(np.abs(folded - whole) / np.abs(whole)) < 1e-12
    assert np.array_equal(folded, gr.evaluate_batch(sr, subst, np.tile(freqs, (len(rls), 1))))
    del sr
    gr.destroy()


def test_c5_full_size_properties_and_2d_grid(c5_full):
    """Full c5 (1997 candidate roots, 100 000 sites).  BASELINE's 2-D layout on 8
    GPUs = 4 candidate groups x 2 site blocks: every candidate belongs to exactly
    one group (dist.assign_candidates), and inside a group the two site blocks'
    lnLs add up to the whole-alignment value."""
    w, tree, g = c5_full
    rng = np.random.default_rng(505)
    rls, subst, freqs, whole = _properties(w, tree, g, 6, rng)
    cg, sg = rdist.grid_2d(8, 2)
    assert (cg, sg) == (4, 2)
    groups = [rdist.assign_candidates(tree.root_count(), c, cg) for c in range(cg)]
    assert sorted(sum(groups, [])) == list(range(1997))
    assert [rdist.rank_coords(r, sg) for r in (0, 1, 6, 7)] == [(0, 0), (0, 1), (3, 0), (3, 1)]
    # one candidate from each group, on both site blocks
    picks = [tree.root_location(grp[len(grp) // 2]).with_ratio(0.35) for grp in groups]
    sub4 = rng.uniform(1e-4, 1.0, (4, 12))
    f4 = np.tile(freqs, (4, 1))
    ref = g.evaluate_batch([g.schedule(*tree.generate_operations(rl)) for rl in picks], sub4, f4)
    total = np.zeros(4)
    for s in range(sg):
        lo, hi = rdist.site_block(g.sites, s, sg)
        part = _slice_partition(w, tree, lo, hi)
        sch = [part.schedule(*tree.generate_operations(rl)) for rl in picks]
        total += part.evaluate_batch(sch, sub4, f4)
        del sch
        part.destroy()
    assert np.max(np.abs(total - ref) / np.abs(ref)) < 1e-12
