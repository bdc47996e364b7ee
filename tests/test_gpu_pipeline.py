"""Two batches of one partition in flight (rdamd_evaluate_batch_submit / _wait): what the
lock-stepped search's two candidate groups do (batch_combiner.hpp; the loop it serves is
/root/reference/src/model.cpp:1139-1272 with the objective of :1488-1502).  The results must
be those of the blocking call, bit for bit, whatever the two slots' batches look like and
while schedules are compiled and dropped beside them."""
import threading

import numpy as np
import pytest

import root_digger_amd as rd
from root_digger_amd import synth

pytestmark = pytest.mark.gpu


def _setup(n, S, R, seed, gaps=True):
    w = synth.workload(n, S, 4, R, seed)
    tree = rd.Tree.from_newick(w["newick"])
    rng = np.random.default_rng(seed)
    seqs = w["seqs"]
    if gaps:   # cherries with 17-64 classes: 64-row tables, second-pass candidates
        out = {}
        for k, v in seqs.items():
            v = np.frombuffer(v.encode(), dtype=np.uint8).copy()
            v[rng.random(S) < 0.1] = ord("-")
            out[k] = v.tobytes().decode()
        seqs = out
    p = rd.Partition.for_tree(tree, 4, S, R, attributes=rd.ATTRIB_SITE_REPEATS)
    import util
    util.load_tips(p, tree, seqs, rd.MAP_NT)
    return tree, p, rng


@pytest.mark.parametrize("n,S,R,seed", [(40, 6000, 4, 501), (12, 20000, 4, 502), (25, 700, 2, 503)])
def test_pipelined_batches_equal_blocking_batches(n, S, R, seed):
    tree, p, rng = _setup(n, S, R, seed)
    n_sched = 12
    rls = [tree.root_location(int(i)).with_ratio(float(rng.uniform(0.05, 0.95)))
           for i in rng.choice(tree.root_count(), size=n_sched, replace=False)]
    scheds = [p.schedule(*tree.generate_operations(rl)) for rl in rls]
    J = 160   # the job list: (schedule, parameter set) pairs with their blocking results
    which = rng.integers(0, n_sched, J)
    subst = rng.uniform(1e-4, 1.0, (J, 12))
    subst[::7] = rng.uniform(1e-4, 1e4, (len(subst[::7]), 12))   # the corners the optimiser visits
    freqs = rng.dirichlet(np.ones(4) * 5, J)
    freqs[::11] = rng.dirichlet(np.ones(4) * 0.05, len(freqs[::11])) * (1 - 4e-4) + 1e-4
    rates = np.array([rd.compute_gamma_cats(x, R) for x in rng.uniform(0.2, 5.0, J)])
    cw = np.full((J, R), 1.0 / R)
    want = p.evaluate_batch([scheds[i] for i in which], subst, freqs, rates, cw)
    assert np.all(np.isfinite(want))
    errors = []

    def group(slot, rounds, gseed):
        g = np.random.default_rng(gseed)
        try:
            for _ in range(rounds):
                m = int(g.integers(1, 70))
                idx = g.integers(0, J, m)
                cnt = p.evaluate_batch_submit(slot, [scheds[i] for i in which[idx]], subst[idx], freqs[idx],
                                              rates[idx], cw[idx])
                got = p.evaluate_batch_wait(slot, cnt)
                if not np.array_equal(got, want[idx]):
                    bad = np.nonzero(got != want[idx])[0]
                    errors.append((slot, m, int(bad[0]), float(got[bad[0]]), float(want[idx][bad[0]])))
                    return
        except Exception as e:   # noqa: BLE001
            errors.append((slot, repr(e)))

    def churn(rounds, gseed):   # schedules compiled and dropped while batches are in flight
        g = np.random.default_rng(gseed)
        try:
            for _ in range(rounds):
                rl = tree.root_location(int(g.integers(0, tree.root_count()))).with_ratio(float(g.uniform(0.05, 0.95)))
                s = p.schedule(*tree.generate_operations(rl))
                del s
        except Exception as e:   # noqa: BLE001
            errors.append(("churn", repr(e)))

    ts = [threading.Thread(target=group, args=(0, 250, seed + 1)),
          threading.Thread(target=group, args=(1, 250, seed + 2)),
          threading.Thread(target=churn, args=(500, seed + 3))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:3]
    # the blocking call still gives the same values afterwards (slot 0 is its workspace)
    assert np.array_equal(p.evaluate_batch([scheds[i] for i in which], subst, freqs, rates, cw), want)


def test_slot_protocol_errors():
    tree, p, rng = _setup(8, 300, 4, 77, gaps=False)
    s = p.schedule(*tree.generate_operations(tree.root_location(0)))
    subst = rng.uniform(0.1, 1.0, (1, 12))
    freqs = np.full((1, 4), 0.25)
    with pytest.raises(rd.RdamdError):
        p.evaluate_batch_wait(0, 1)             # nothing submitted
    p.evaluate_batch_submit(1, [s], subst, freqs)
    with pytest.raises(rd.RdamdError):
        p.evaluate_batch_submit(1, [s], subst, freqs)   # not waited for yet
    a = p.evaluate_batch_wait(1, 1)
    with pytest.raises(rd.RdamdError):
        p.evaluate_batch_submit(2, [s], subst, freqs)   # no such slot
    assert np.array_equal(a, p.evaluate_batch([s], subst, freqs))
