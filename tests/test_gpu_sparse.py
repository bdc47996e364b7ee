"""RDAMD_ATTRIB_SPARSE_CLVS: a partition whose CLV / scale buffers get device memory when a call
first names them (the model replicas of a lock-stepped search hold the root's two children and
nothing else).  The kernels see pool slots where a dense partition shows
them the caller's indices -- every result must be the dense partition's, bit for bit."""
import numpy as np
import pytest

import root_digger_amd as rd
from root_digger_amd import synth
import util

pytestmark = pytest.mark.gpu
SUBST = [.34, .42, .24, .74, .16, .88, .75, .54, .20, .06, .08, .41]
FREQS = [.21, .29, .27, .23]


def _pair(tree, seqs, states, R, cmap=rd.MAP_NT):
    S = len(next(iter(seqs.values())))
    out = []
    for attr in (0, rd.ATTRIB_SPARSE_CLVS):
        p = rd.Partition.for_tree(tree, states, S, R, rd.ATTRIB_NONREV | attr)
        util.load_tips(p, tree, seqs, cmap)
        p.set_subst_params(0, SUBST if states == 4 else list(np.linspace(.1, 1.9, states * states - states)))
        p.set_frequencies(0, FREQS if states == 4 else [1.0 / states] * states)
        p.set_category_rates(rd.compute_gamma_cats(0.8, R))
        out.append(p)
    return out


@pytest.mark.parametrize("n,S,R,seed", [(24, 900, 4, 5), (9, 70, 1, 6), (40, 333, 8, 7)])
def test_sparse_partition_equals_dense_bit_for_bit(n, S, R, seed):
    w = synth.workload(n, S, 4, R, seed)
    tree = rd.Tree.from_newick(w["newick"])
    dense, sparse = _pair(tree, w["seqs"], 4, R)
    assert sparse.clv_bytes() < dense.clv_bytes()
    for i in (0, 3, tree.root_count() - 1):
        rl = tree.root_location(i).with_ratio(0.3)
        assert util.compute_lh(sparse, tree, rl) == util.compute_lh(dense, tree, rl)
        # (a traversal names n - 1 of the partition's 2n - 3 buffers: the pool has grown to hold those)
        assert sparse.clv_bytes() <= dense.clv_bytes()
        ops, _, _ = tree.generate_operations(rl)
        for op in (ops[0], ops[len(ops) // 2], ops[len(ops) - 1]):
            assert np.array_equal(sparse.get_clv(op.parent_clv_index), dense.get_clv(op.parent_clv_index))
            assert np.array_equal(sparse.get_scaler(op.parent_scaler_index), dense.get_scaler(op.parent_scaler_index))
        # root-only steps and moves on top of it
        rl2 = tree.root_location((i + 5) % tree.root_count()).with_ratio(0.6)
        ops, pmi, brl = tree.generate_root_update_operations(rl2)   # (moves the TREE: generated once)
        for part in (sparse, dense):
            part.update_prob_matrices(pmi, brl)
            part.update_clvs(ops)
        op, _, _ = tree.generate_derivative_operations(rl2)
        l1 = [rl2.saved_brlen * a for a in (0.6, 0.0, 1.0, 0.25)]
        l2 = [rl2.saved_brlen * (1 - a) for a in (0.6, 0.0, 1.0, 0.25)]
        assert list(sparse.root_loglikelihood_fused(op, l1, l2)) == list(dense.root_loglikelihood_fused(op, l1, l2))
        assert util.compute_lh_root(sparse, tree, rl2) == util.compute_lh_root(dense, tree, rl2)
        assert util.compute_lh_root(dense, tree, rl2) == dense.root_loglikelihood_fused(op, l1, l2)[0]


def test_sparse_replica_keeps_three_buffers():
    """the searches' compute_lh in front of the root-only steps on a sparse partition: discard,
    write the root's two children, evaluate root positions -- four slots of the pool, whatever
    the tree's size, and the dense partition's bits"""
    n, S, R = 200, 5000, 4
    w = synth.workload(n, S, 4, R, 11)
    tree = rd.Tree.from_newick(w["newick"])
    dense, sparse = _pair(tree, w["seqs"], 4, R)
    rates = rd.compute_gamma_cats(0.8, R)
    slot = S * R * 4 * 8 + S * 4
    for i in (1, 77, 200, 396):
        rl = tree.root_location(i).with_ratio(0.4)
        ops, pmi, brl = tree.generate_operations(rl)
        sparse.discard_clvs()
        a = sparse.evaluate_root_children(ops, pmi, brl, SUBST, FREQS, rates)
        b = dense.evaluate_root_children(ops, pmi, brl, SUBST, FREQS, rates)
        assert a == b
        op, _, _ = tree.generate_derivative_operations(rl)
        l1 = [rl.saved_brlen * x for x in (0.4, 0.0, 1.0)]
        l2 = [rl.saved_brlen * (1 - x) for x in (0.4, 0.0, 1.0)]
        assert list(sparse.root_loglikelihood_fused(op, l1, l2)) == list(dense.root_loglikelihood_fused(op, l1, l2))
        assert list(rd.root_loglikelihood_fused_multi([sparse, dense], [op, op], [l1, l1], [l2, l2])[0]) == \
               list(dense.root_loglikelihood_fused(op, l1, l2))
        assert sparse.clv_bytes() == 4 * slot
    assert dense.clv_bytes() == tree.branch_count() * slot
    # a scale buffer nobody wrote reads as zeros after a discard, as on a fresh dense partition
    sparse.discard_clvs()
    assert not np.any(sparse.get_scaler(5))


def test_sparse_20_states_and_generic_kernels():
    for states in (20, 5):
        w = synth.workload(14, 150, states, 2, 21 + states)
        tree = rd.Tree.from_newick(w["newick"])
        cmap = util.make_map(w["alphabet"])
        dense, sparse = _pair(tree, w["seqs"], states, 2, cmap)
        for i in (0, 9):
            rl = tree.root_location(i).with_ratio(0.5)
            assert util.compute_lh(sparse, tree, rl) == util.compute_lh(dense, tree, rl)
            ops, _, _ = tree.generate_operations(rl)
            last = ops[len(ops) - 1]
            assert np.array_equal(sparse.get_clv(last.parent_clv_index), dense.get_clv(last.parent_clv_index))


def test_replicas_of_a_children_only_search_cost_megabytes():
    """c5's per-GPU shard (1000 taxa x 50 000 sites, G4) was 12.8 GB per replica"""
    w = synth.workload(1000, 50000, 4, 4, 77, simulate_seqs=False)
    tree = rd.Tree.from_newick(w["newick"])
    m = rd.Model(tree, w["seqs"], rate_cats=4, seed=1)
    fit, per = m.max_replicas(32)
    assert fit == 32 and per < 0.3e9, (fit, per)
    m.set_root_children_only(False)
    fit_dense, per_dense = m.max_replicas(32)
    assert per_dense > 12e9 and fit_dense < 32
    m.destroy()
