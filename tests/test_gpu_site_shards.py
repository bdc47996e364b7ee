"""Site-sharded model_t (north star: "site blocks shard across the
GPUs ... with an RCCL all-reduce of per-block log-likelihoods").

* two ranks, each with the HIP model of ONE column block of the alignment and a
  gloo all-reduce plugged into rdamd_model_set_lnl_reducer (both ranks share
  device 0 of this one-GPU box: RCCL refuses two ranks on one device, the hook
  is the same): every model-level quantity -- empirical frequencies,
  compute_lh, compute_lh_root, compute_dlh, the root sweeps, optimize_params,
  the exhaustive search -- must equal the one-rank whole-alignment run, and the
  two ranks must hold the same bits;
* the RCCL communicator itself (rdamd_comm_*, world size 1 on this box) as the
  device-side reducer behind the optimiser's objective batches.
The candidate split stays the reference's (src/model.cpp:1867-1911)."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

import root_digger_amd as rd
from root_digger_amd import dist as rdist
import util

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(util.ROOT, "oracle", "_ref", "liblbfgsb_ref.so")
SUBST = [.34, .42, .24, .74, .16, .88, .75, .54, .20, .06, .08, .41]
ROOTS = (0, 5, 16)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _measure(m, tree, search, tight=True):
    """the model-level quantities a site-sharded run must reproduce"""
    out = {}
    m.initialize_partitions()                      # empirical frequencies + random rates
    m.set_subst_rates(SUBST)
    m.set_gamma_alpha(0.7)
    rls = [tree.root_location(i).with_ratio(0.3) for i in ROOTS]
    out["lh"] = [m.compute_lh(rl) for rl in rls]
    out["lh_root"] = [m.compute_lh_root(rls[-1].with_ratio(a)) for a in (0.1, 0.9)]
    out["dlh"] = list(m.compute_dlh(rls[-1]))
    out["sweep"] = list(m.compute_all_root_lh())
    out["sweep_batched"] = list(m.compute_all_root_lh_batched())
    out["sweep_directional"] = list(m.compute_all_root_lh_directional())
    if search:
        m.set_lbfgsb(C.CDLL(REF).setulb)
        # (tight settings where two different summation orders are compared: a loosely
        # converged run would amplify last-bit differences)
        pgtol, factor, atol, brtol = (1e-7, 1e4, 1e-7, 1e-9) if tight else (1e-3, 1e12, 1e-3, 1e-3)
        r = m.optimize_params(rls[0], SUBST, [.25] * 4, 1.0, pgtol, factor)
        out["opt"] = list(r["subst"]) + list(r["freqs"]) + [r["gamma_alpha"], r["evaluations"]]
        out["opt_lh"] = m.compute_lh(rls[0])
        m.assign_by_rank(0, 1 if not tight else 3)       # all 17 candidates, or the first 6
        res = m.exhaustive_search(atol, pgtol, brtol, factor)
        out["search"] = (res["root_id"], list(res["llh"]), list(res["alpha"]))
    return out


def _rank(rank, world, port, q, search):
    sys.path.insert(0, HERE)
    import torch
    import torch.distributed as tdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    rd.set_device(0)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tree = rd.Tree.from_file(os.path.join(util.DATA, "10.tree"))
        seqs = util.read_fasta(os.path.join(util.DATA, "10.fasta"))
        lo, hi = rdist.site_block(1000, rank, world)
        block, w = util.compress({k: v[lo:hi] for k, v in seqs.items()})
        m = rd.Model(tree, block, rate_cats=4, weights=w, seed=3)
        calls = [0]

        def reduce(values, n):            # host array in, group sum out
            t = torch.from_numpy(values)
            tdist.all_reduce(t, op=tdist.ReduceOp.SUM)
            calls[0] += 1

        m.set_lnl_reducer(reduce)
        out = _measure(m, tree, search)
        out["calls"] = calls[0]
        q.put((rank, out))
    except Exception as e:      # pragma: no cover
        q.put((rank, {"error": repr(e)}))
        raise
    finally:
        tdist.destroy_process_group()


def _two_ranks(search):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, 2, port, q, search)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert "error" not in got[0] and "error" not in got[1], got
    return got


def _whole(search):
    tree = rd.Tree.from_file(os.path.join(util.DATA, "10.tree"))
    seqs, w = util.compress(util.read_fasta(os.path.join(util.DATA, "10.fasta")))
    return _measure(rd.Model(tree, seqs, rate_cats=4, weights=w, seed=3), tree, search)


def _close(a, b, tol):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)) < tol


def test_two_site_blocks_reproduce_the_whole_alignment_model():
    got = _two_ranks(search=False)
    one = _whole(search=False)
    for key in ("lh", "lh_root", "sweep", "sweep_batched", "sweep_directional"):
        assert got[0][key] == got[1][key], key                 # same bits on both ranks
        assert _close(got[0][key], one[key], 1e-12), key       # = the unsharded model
    assert got[0]["dlh"] == got[1]["dlh"]
    assert _close(got[0]["dlh"][0], one["dlh"][0], 1e-12)
    # the derivative is a difference over eps = 1e-8 (src/model.cpp:481-519):
    # 1e-16 relative noise on lnL ~ 1e4 is up to 1e-4 absolute on it
    assert abs(got[0]["dlh"][1] - one["dlh"][1]) < 1e-3 * max(1.0, abs(one["dlh"][1]))
    assert got[0]["calls"] == got[1]["calls"] > 20


def test_two_site_blocks_search_like_one_rank():
    """optimize_params + the exhaustive candidate loop (src/model.cpp:1139-1272) on a
    2-block model: the ranks agree bit for bit with each other and, to optimiser
    tolerance, with the one-rank run (summation order differs by 1e-16)."""
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/liblbfgsb_ref.so not built (needs /root/reference at build time)")
    got = _two_ranks(search=True)
    one = _whole(search=True)
    assert got[0]["opt"] == got[1]["opt"] and got[0]["search"] == got[1]["search"]
    assert _close(got[0]["opt_lh"], one["opt_lh"], 1e-7)
    ids, llh, alpha = got[0]["search"]
    assert ids == one["search"][0] == list(range(6))
    assert _close(llh, one["search"][1], 2e-6)
    assert np.max(np.abs(np.array(alpha) - np.array(one["search"][2]))) < 2e-2


def test_rccl_communicator_as_device_side_reducer():
    """rdamd_comm_* (RCCL loaded with dlopen) on a one-rank group: ncclAllReduce on
    the partition's stream behind rdamd_evaluate_batch_device, i.e. the code path
    a multi-GPU site group runs; a one-rank sum must change nothing."""
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/liblbfgsb_ref.so not built")
    # (no torch tensor here: librdamd and a PyTorch imported after it may run on two
    # different ROCm runtime instances; rdamd_comm_* loads the RCCL of its own one)
    comm = rd.Comm(rd.Comm.unique_id(), 0, 1)
    tree = rd.Tree.from_file(os.path.join(util.DATA, "10.tree"))
    seqs, w = util.compress(util.read_fasta(os.path.join(util.DATA, "10.fasta")))
    plain = _measure(rd.Model(tree, seqs, rate_cats=4, weights=w, seed=3), tree, True, tight=False)
    m = rd.Model(tree, seqs, rate_cats=4, weights=w, seed=3)
    m.set_lnl_reducer(comm.reducer, on_device=True, user=comm.handle)
    sharded = _measure(m, tree, True, tight=False)
    for key in plain:
        assert sharded[key] == plain[key], key
    with pytest.raises(rd.RdamdError):       # free-running replicas would reorder the collectives
        m.exhaustive_search(1e-3, 1e-3, 1e-3, 1e12, workers=4)
    # (in lock step a site-sharded model advances in rounds: tests/test_gpu_lockstep_rounds.py)
    m.destroy()
    comm.destroy()
