#!/usr/bin/env python3
"""(Lives under tests/: the caller's optimiser is the reference's L-BFGS-B build, oracle/_ref.)
Randomised check of the searches' central property: the exhaustive search with N candidates in lock
step (pipelined groups, combined root-only launches, root children from the evaluator) leaves the
records of the sequential loop, bit for bit -- random shapes, gaps / ambiguity codes, partitioned
models.  usage: lockstep_soak.py [seconds] [seed]"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import root_digger_amd as rd          # noqa: E402
from root_digger_amd import synth     # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
comm = rd.Comm(rd.Comm.unique_id(), 0, 1)     # a one-rank site group: the device path of the rounds
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
lb = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "liblbfgsb_ref.so"))
t0, rounds, cands, sharded = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    n = int(rng.integers(5, 40))
    S = int(rng.choice([130, 1000, 4097, 20000]))
    R = int(rng.choice([1, 4]))
    w = synth.workload(n, S, 4, R, int(rng.integers(1 << 30)))
    if rng.random() < 0.4:
        for k, v in w["seqs"].items():
            v = np.frombuffer(v.encode(), dtype=np.uint8).copy()
            v[rng.random(S) < 0.15] = ord("-")
            v[rng.random(S) < 0.02] = ord(str(rng.choice(list("RYKMSW"))))
            w["seqs"][k] = v.tobytes().decode()
    tree = rd.Tree.from_newick(w["newick"])
    # (round 6: half of the models stop early -- BASELINE c5's mode, src/model.cpp:1187-1197 -- and the
    # one-rank communicator alternates between its two sum modes: gather + rank-order kernel / ncclAllReduce)
    early = bool(rng.random() < 0.5)
    early_models = globals().get("early_models", 0) + int(early)
    m = rd.Model(tree, w["seqs"], rate_cats=R, seed=int(rng.integers(1 << 20)), early_stop=early)
    m.initialize_partitions()
    m.set_lbfgsb(lb.setulb)
    m.compute_lh(tree.root_location(0))
    take = int(rng.integers(2, 9))
    m._ok(rd.lib.rdamd_model_assign_by_rank(m._h, 0, max(1, tree.root_count() // take)), "assign")
    tol = (1e-4, 1e-4, 1e-6, 1e9)
    seq = m.exhaustive_search(*tol)
    order = np.argsort(seq["root_id"])
    for in_flight in (int(rng.integers(2, 5)), int(rng.integers(5, 12))):
        lock = m.exhaustive_search(*tol, lockstep=in_flight)
        assert lock["root_id"] == sorted(seq["root_id"]), (n, S, R, in_flight)
        assert np.array_equal(lock["llh"], seq["llh"][order]), (n, S, R, in_flight, lock["llh"], seq["llh"][order])
        assert np.array_equal(lock["alpha"], seq["alpha"][order]), (n, S, R, in_flight)
    # round 5: the same search in DETERMINISTIC ROUNDS (lockstep_conductor.hpp), one or two worker groups ...
    m.set_lockstep_rounds(1)
    m.set_lockstep_groups(int(rng.integers(1, 3)))
    lock = m.exhaustive_search(*tol, lockstep=int(rng.integers(2, 12)))
    assert lock["root_id"] == sorted(seq["root_id"]), (n, S, R, "rounds")
    assert np.array_equal(lock["llh"], seq["llh"][order]) and np.array_equal(lock["alpha"], seq["alpha"][order]), (n, S, R, "rounds")
    m.set_lockstep_rounds(-1)
    m.set_lockstep_groups(0)
    if rounds % 3 == 0:   # ... and as a site-sharded model would run it: the RCCL reducer behind every round
        comm.set_sum_mode(rd.COMM_SUM_ALLREDUCE if sharded % 2 else rd.COMM_SUM_GATHER)
        m.set_lnl_reducer(comm.reducer, on_device=True, user=comm.handle)
        lock = m.exhaustive_search(*tol, lockstep=int(rng.integers(2, 12)))
        assert np.array_equal(lock["llh"], seq["llh"][order]) and np.array_equal(lock["alpha"], seq["alpha"][order]), (n, S, R, "rccl rounds")
        m.set_lnl_reducer(None)
        sharded += 1
    rounds += 1
    cands += len(seq["root_id"])
print("%d random models, %d candidates each searched sequentially, twice in lock step (arrival order) and once in "
      "deterministic rounds (%d of the models also with the one-rank RCCL reducer behind every round, both sum modes; %d of "
      "the models with --early-stop): identical records; %.0f s"
      % (rounds, cands, sharded, early_models, time.time() - t0))
