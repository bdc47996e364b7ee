#!/usr/bin/env python3
"""The exhaustive search on the shapes ONE GPU holds under BASELINE's sharding -- c4 / 8 (500 taxa x
62 500 sites) and c5 as 4 candidate groups x 2 site blocks (1000 x 50 000) -- both ways: one
candidate at a time with a collective per request (what --site-shards ran until round 4), and in
lock step in deterministic rounds (csrc/lockstep_conductor.hpp) with one collective per round.

One GPU, one rank: the site group's reducer is the RCCL communicator with ONE rank (the device
path exactly as a multi-GPU group runs it -- batch to device memory, ncclAllReduce queued behind it
on the partition's stream, sums copied back -- minus the wire).  Seconds per candidate, collectives
per candidate, the rounds' phase times.

usage: shard_search.py [c4|c5|c2|TAXAxSITES] [--seq N] [--lock N] [--in-flight N] [--groups 0|1]
"""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import root_digger_amd as rd  # noqa: E402
from root_digger_amd import synth  # noqa: E402

SHAPES = {"c4": (500, 62500), "c5": (1000, 50000), "c2": (100, 50000), "c2s": (100, 6250)}
REF = os.path.join(ROOT, "oracle", "_ref", "liblbfgsb_ref.so")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shape", nargs="?", default="c4")
    ap.add_argument("--seq", type=int, default=3, help="candidates of the sequential run")
    ap.add_argument("--lock", type=int, default=32, help="candidates of the lock-stepped runs")
    ap.add_argument("--in-flight", type=int, default=32)
    ap.add_argument("--groups", type=int, default=0, help="0: two alternating worker groups; 1: one")
    ap.add_argument("--no-reducer", action="store_true", help="an unsharded model (arrival-order lock step too)")
    a = ap.parse_args()
    n, S = SHAPES[a.shape] if a.shape in SHAPES else tuple(int(x) for x in a.shape.split("x"))
    t0 = time.time()
    w = synth.workload(n, S, 4, 4, 0xD166E5 + n)
    tree = rd.Tree.from_newick(w["newick"])
    roots = tree.root_count()
    print("%s: %d taxa x %d sites, %d candidate roots (alignment simulated in %.0f s)" % (a.shape, n, S, roots, time.time() - t0), flush=True)
    setulb = C.CDLL(REF).setulb
    comm = None if a.no_reducer else rd.Comm(rd.Comm.unique_id(), 0, 1)
    tol = (1e-7, 1e-7, 1e-12, 1e4)   # rd's defaults: atol, bfgstol, brtol, factor

    def model():
        m = rd.Model(tree, w["seqs"], rate_cats=4, seed=1)
        if comm is not None:
            m.set_lnl_reducer(comm.reducer, on_device=True, user=comm.handle)
        m.initialize_partitions()
        m.set_lbfgsb(setulb)
        return m

    def run(label, k, **kw):
        m = model()
        m.assign_by_rank(0, max(1, roots // k))
        if kw.get("rounds") is not None:
            m.set_lockstep_rounds(kw["rounds"])
        m.set_lockstep_groups(a.groups)
        t = time.time()
        r = m.exhaustive_search(*tol, lockstep=kw.get("lockstep", 0))
        dt = time.time() - t
        got = len(r["root_id"])
        st, ls, ct = m.round_stats(), m.lockstep_stats(), m.counters()
        coll = st["collectives"] + st["own_collectives"]
        print("%-34s %3d candidates  %8.2f s  %7.3f s/candidate  collectives/candidate %8.1f  "
              "objective launches %6d (%.1f jobs each)" % (
                  label, got, dt, dt / got, coll / got,
                  ls["objective_launches"] or ct["objective_batches"],
                  (ls["objective_jobs"] / max(1, ls["objective_launches"])) if ls["objective_launches"]
                  else ct["objective_evaluations"] / max(1, ct["objective_batches"])), flush=True)
        if st["rounds"]:
            s = st["seconds"]
            print("    rounds %d (%.3f ms each): batch queued %.2f s, root launch %.2f s, sum queued %.2f s, "
                  "waiting %.2f s; redos %d; root launches %d (%.1f steps each)" % (
                      st["rounds"], 1e3 * dt / st["rounds"], s["objective_queued"], s["root_launch"], s["sum_queued"],
                      s["waiting"], st["redos"], ls["root_launches"], ls["root_steps"] / max(1, ls["root_launches"])), flush=True)
        rec = sorted(zip(r["root_id"], r["llh"], r["alpha"]))
        m.destroy()
        return dt / got, rec

    per_seq, rec_seq = run("sequential%s" % ("" if a.no_reducer else ", collective per request"), a.seq)
    per_lock, rec_lock = run("lock step in rounds, %d in flight" % a.in_flight, a.lock, lockstep=a.in_flight, rounds=1)
    same = rec_lock[:len(rec_seq)] == rec_seq
    print("records of the first %d candidates: %s" % (len(rec_seq), "bit-identical" if same else "DIFFERENT"))
    if a.no_reducer:
        per_arr, rec_arr = run("lock step, arrival order (round 4)", a.lock, lockstep=a.in_flight, rounds=0)
        print("arrival-order records %s" % ("bit-identical" if rec_arr == rec_lock else "DIFFERENT"))
    print("sequential / lock step = %.2f x" % (per_seq / per_lock))
    if comm is not None:
        comm.destroy()
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
