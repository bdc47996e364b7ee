"""Lock step in DETERMINISTIC ROUNDS (csrc/lockstep_conductor.hpp) -- north star: "the outer
root-edge loop batches lnL evaluations so candidate edges and site blocks shard across the 8 GPUs
... with an RCCL all-reduce of per-block log-likelihoods".  The reference's loop is
src/model.cpp:1154-1229 (one candidate at a time), its rank split :1867-1911.

* the stream-ordered device batch the rounds are built on (rdamd_evaluate_batch_submit_device /
  _redo_device / _finish_device) against the blocking call, the second-pass flag included;
* one process: the search in rounds == the sequential search, bit for bit, with one and two worker
  groups, without a reducer and with the RCCL communicator (one rank) as the device-side reducer --
  and with a fraction of the collectives;
* rd_amd --site-shards G --lockstep N on 2 / 4 / 8 ranks of one device (host reducer: RCCL refuses
  two ranks on a GPU; the model hook is the same): the ranks of a group hold the same bits, the
  records are the sequential sharded run's bit for bit and the one-rank run's to tolerance."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import root_digger_amd as rd
from root_digger_amd import synth
import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RD = os.path.join(ROOT, "root_digger_amd", "bin", "rd_amd")
REF = os.path.join(ROOT, "oracle", "_ref", "liblbfgsb_ref.so")
MSA, TREE = os.path.join(util.DATA, "10.fasta"), os.path.join(util.DATA, "10.tree")
# atol, pgtol, brtol, factor: loose -- these tests compare runs that must agree BIT FOR BIT
# (same sums in the same order), which holds at any tolerance; short trajectories keep them cheap
LOOSE = (1e-2, 1e-2, 1e-2, 1e13)


class _Hip:
    """device memory from the HIP runtime librdamd itself runs on (no torch here: a PyTorch
    imported beside the library may bring a second runtime instance, bench.py has the check)"""

    def __init__(self):
        self.rt = C.CDLL(rd.hip_runtime_path())
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipFree.argtypes = [C.c_void_p]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.rt.hipDeviceSynchronize.argtypes = []

    def alloc(self, nbytes):
        p = C.c_void_p()
        assert self.rt.hipMalloc(C.byref(p), nbytes) == 0
        return p.value

    def write(self, ptr, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        assert self.rt.hipMemcpy(C.c_void_p(ptr), arr.ctypes.data_as(C.c_void_p), arr.nbytes, 1) == 0

    def read(self, ptr, n):
        out = np.zeros(n, dtype=np.float64)
        assert self.rt.hipDeviceSynchronize() == 0
        assert self.rt.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), 8 * n, 2) == 0
        return out

    def free(self, ptr):
        self.rt.hipFree(C.c_void_p(ptr))


def test_stream_ordered_device_batch_equals_the_blocking_call():
    hip = _Hip()
    w = synth.workload(30, 700, 4, 4, 151)
    tree = rd.Tree.from_newick(w["newick"])
    p = rd.Partition.for_tree(tree, 4, 700, 4, attributes=rd.ATTRIB_SITE_REPEATS)
    util.load_tips(p, tree, w["seqs"], rd.MAP_NT)
    rng = np.random.default_rng(7)
    rls = [tree.root_location(int(i)).with_ratio(0.4) for i in rng.choice(tree.root_count(), 6, replace=False)]
    scheds = [p.schedule(*tree.generate_operations(rl)) for rl in rls]
    assert all(s.stats()["pseudo_tips"] > 0 for s in scheds)
    J = 6
    subst = rng.uniform(1e-2, 1.0, (J, 12))
    freqs = rng.dirichlet(np.ones(4) * 5, J)
    rates = np.tile(rd.compute_gamma_cats(1.0, 4), (J, 1))
    d = hip.alloc(8 * (J + 1))
    # ordinary parameters: flag down, results final behind the finishing kernel
    want = p.evaluate_batch(scheds, subst, freqs, rates)
    for slot in (0, 1):
        n = p.evaluate_batch_submit_device(slot, scheds, subst, freqs, d, rates)
        got = hip.read(d, n + 1)
        assert got[n] == 0.0 and np.array_equal(got[:n], want)
        p.evaluate_batch_finish_device(slot)
    # two jobs whose tables need the per-site rescaling rule (a rate of 1e-42: P entries below
    # 2^-128): the flag is up, their entries are not final until the redo, the others' are
    rates[1] = rates[4] = [1e-42, 0.5, 1.0, 2.5]
    want = p.evaluate_batch(scheds, subst, freqs, rates)
    n = p.evaluate_batch_submit_device(0, scheds, subst, freqs, d, rates)
    got = hip.read(d, n + 1)
    assert got[n] == 1.0
    keep = [0, 2, 3, 5]
    assert np.array_equal(got[keep], want[keep])
    p.evaluate_batch_redo_device(0, d)
    got = hip.read(d, n + 1)
    assert got[n] == 0.0 and np.array_equal(got[:n], want)
    # a second batch in flight on the other slot while this one is redone
    n1 = p.evaluate_batch_submit_device(1, scheds[:3], subst[:3], freqs[:3], d, rates[:3])
    assert n1 == 3
    p.evaluate_batch_finish_device(0)
    with pytest.raises(rd.RdamdError):
        p.evaluate_batch_finish_device(0)          # nothing in flight on the slot any more
    got = hip.read(d, 4)
    assert got[3] == 1.0                            # (job 1 of this batch is a flagged one)
    p.evaluate_batch_redo_device(1, d)
    assert np.array_equal(hip.read(d, 4)[:3], want[:3])
    p.evaluate_batch_finish_device(1)
    assert np.array_equal(p.evaluate_batch(scheds, subst, freqs, rates), want)
    hip.free(d)


def _model(seed=3, early_stop=False):
    tree = rd.Tree.from_file(TREE)
    seqs, w = util.compress(util.read_fasta(MSA))
    m = rd.Model(tree, seqs, rate_cats=4, weights=w, seed=seed, early_stop=early_stop)
    m.initialize_partitions()
    m.set_lbfgsb(C.CDLL(REF).setulb)
    return m


def _search(m, lockstep, candidates=17):
    m.assign_by_rank(0, 17 // candidates)
    r = m.exhaustive_search(*LOOSE, lockstep=lockstep)
    return list(r["root_id"]), list(r["llh"]), list(r["alpha"])


@pytest.fixture(scope="module")
def sequential_records():
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/liblbfgsb_ref.so not built (needs /root/reference at build time)")
    m = _model()
    out = _search(m, 0)
    m.destroy()
    return out


@pytest.fixture(scope="module")
def sequential_records_early_stop():
    """BASELINE config c5's mode: `--exhaustive --early-stop` (src/model.cpp:1187-1197: a candidate
    whose root position moved by less than brtol leaves its loop at once, with the NEW position)"""
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/liblbfgsb_ref.so not built (needs /root/reference at build time)")
    m = _model(early_stop=True)
    out = _search(m, 0)
    m.destroy()
    return out


def test_early_stop_changes_the_candidates_request_sequences(sequential_records, sequential_records_early_stop):
    """(so that the early-stop cases below test something: with these tolerances the rule fires,
    candidates leave their loops earlier and end elsewhere)"""
    assert sequential_records_early_stop[0] == sequential_records[0]
    assert sequential_records_early_stop[1:] != sequential_records[1:]


@pytest.mark.parametrize("in_flight", [3, 6, 17])
@pytest.mark.parametrize("early_stop", [False, True])
def test_rounds_reproduce_the_sequential_search_on_one_rank(sequential_records, sequential_records_early_stop,
                                                            in_flight, early_stop):
    """no reducer: the rounds' machinery alone (one group below four candidates in flight, two
    alternating groups from four on); a job's value does not depend on its launch.  With and
    without the early-stop rule (it changes when a candidate posts "next candidate")"""
    m = _model(early_stop=early_stop)
    m.set_lockstep_rounds(1)
    assert _search(m, in_flight) == (sequential_records_early_stop if early_stop else sequential_records)
    st, ls = m.round_stats(), m.lockstep_stats()
    assert st["rounds"] > 0 and st["redos"] == 0 and st["own_collectives"] == 0
    assert ls["objective_jobs"] > ls["objective_launches"] > 0 and ls["root_steps"] > ls["root_launches"] > 0
    if in_flight == 17:      # every candidate in flight from the first round on
        assert ls["objective_jobs"] / ls["objective_launches"] > 15     # (two groups of 8 - 9: 13 / 5 / 2 jobs each)
    m.destroy()


def test_rank_order_sum_kernel_is_the_host_reducers_sum():
    """RDAMD_COMM_SUM_GATHER = ncclAllGather + this kernel: ((v0 + v1) + v2) + ... in rank order.
    Bit for bit the loop rd_amd's host reducer (tools/rendezvous.hpp) and dist.allreduce_lnl run
    -- on data whose sum depends on the association"""
    hip = _Hip()
    rng = np.random.default_rng(11)
    for ranks, n in ((2, 1), (8, 449), (4, 12_200), (3, 70_001)):
        g = rng.standard_normal((ranks, n)) * 10.0 ** rng.integers(-18, 18, (ranks, n))
        g[:, 0] = [1e16, 1.0, -1e16, 1.0, 0.5, 0.25, -1.0, 3.0][:ranks]
        want = g[0].copy()
        for r in range(1, ranks):
            want = want + g[r]
        d_g, d_out = hip.alloc(8 * ranks * n), hip.alloc(8 * n)
        hip.write(d_g, g)
        rd.rank_order_sum(d_g, d_out, n, ranks)
        assert np.array_equal(hip.read(d_out, n), want)
        rd.rank_order_sum(d_g, d_g, n, ranks)                 # in place over the first vector
        assert np.array_equal(hip.read(d_g, n), want)
        hip.free(d_g)
        hip.free(d_out)
    # pairwise or reversed association gives other bits on this data: the test would notice
    assert ((1e16 + 1.0) + -1e16) + 1.0 != (1e16 + 1.0) + (-1e16 + 1.0)


@pytest.mark.parametrize("mode", ["gather", "allreduce"])
def test_rounds_with_the_rccl_communicator_as_reducer(sequential_records, mode):
    """the device path of a site group -- batch to device memory, the collective queued behind it
    on the partition's stream (default: ncclAllGather + the rank-order sum kernel; selectable:
    ncclAllReduce), the sums copied back, the event waited for with the communicator's failure
    handling -- on a one-rank group: a one-rank sum changes nothing, so the records are the
    sequential ones, and the collectives are counted"""
    comm = rd.Comm(rd.Comm.unique_id(), 0, 1)
    assert comm.sum_mode == rd.COMM_SUM_GATHER
    if mode == "allreduce":
        comm.set_sum_mode(rd.COMM_SUM_ALLREDUCE)
        assert comm.sum_mode == rd.COMM_SUM_ALLREDUCE
    seq = _model()
    seq.set_lnl_reducer(comm.reducer, on_device=True, user=comm.handle)
    assert _search(seq, 0) == sequential_records
    per_request = seq.round_stats()["own_collectives"]
    seq.destroy()
    m = _model()
    m.set_lnl_reducer(comm.reducer, on_device=True, user=comm.handle)
    assert _search(m, 17) == sequential_records
    st = m.round_stats()
    # (the model's own one: the group's agreement on the number of candidates in flight)
    assert st["own_collectives"] == 1 and 0 < st["collectives"] <= st["rounds"]
    # (two groups of 8 - 9 candidates, each as long as its longest one: 5.6 x fewer on this data)
    assert per_request > 3 * st["collectives"], (per_request, st)
    # a model that refuses rounds refuses the lock-stepped search when it is site-sharded
    m.set_lockstep_rounds(0)
    with pytest.raises(rd.RdamdError):
        _search(m, 4)
    m.destroy()
    comm.destroy()


def test_rounds_repeat_the_collective_when_a_batch_needs_its_second_pass(tmp_path):
    """A branch of length 1e-40 puts P-matrix entries into (0, 2^-128): every job that meets it is
    flagged (FusedJob::tt_unsafe) and must run the evaluator's second pass.  In a round that
    decision travels through the sum: the batch's flag comes back non-zero, the round runs the
    pass, restores this rank's values and repeats its collective (rdamd_evaluate_batch_redo_device)
    -- with and without a reducer, one and two worker groups; the records are the sequential
    search's, whose blocking batches take the pass inside rdamd_evaluate_batch."""
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/liblbfgsb_ref.so not built")
    nw = open(TREE).read()
    mt = re.search(r":[0-9.eE+-]+", nw)
    tree = rd.Tree.from_newick(nw[:mt.start()] + ":1e-40" + nw[mt.end():])
    seqs, w = util.compress(util.read_fasta(MSA))

    def model():
        m = rd.Model(tree, seqs, rate_cats=4, weights=w, seed=3)
        m.initialize_partitions()
        m.set_lbfgsb(C.CDLL(REF).setulb)
        return m

    def search(m, lockstep):
        m.assign_by_rank(0, 3)                     # the first six candidates
        r = m.exhaustive_search(*LOOSE, lockstep=lockstep)
        return list(r["root_id"]), list(r["llh"]), list(r["alpha"])

    seq = model()
    want = search(seq, 0)
    assert np.all(np.isfinite(want[1]))
    seq.destroy()
    comm = rd.Comm(rd.Comm.unique_id(), 0, 1)
    for with_reducer, groups, in_flight in ((False, 1, 3), (False, 2, 6), (True, 1, 6), (True, 2, 4)):
        m = model()
        if with_reducer:
            m.set_lnl_reducer(comm.reducer, on_device=True, user=comm.handle)
        m.set_lockstep_rounds(1)
        m.set_lockstep_groups(groups)
        assert search(m, in_flight) == want, (with_reducer, groups)
        st = m.round_stats()
        assert st["redos"] > 0 and st["collectives"] > st["redos"], st
        m.destroy()
    comm.destroy()


def _run_ranks(args, world, timeout=900):
    s = __import__("socket").socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    # (GPU_MAX_HW_QUEUES: up to eight processes with a dozen HIP streams each share ONE device
    # here; beyond the device's hardware queues the firmware rotates them by the millisecond)
    env = dict(os.environ, WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               GPU_MAX_HW_QUEUES="1")
    procs = [subprocess.Popen(args, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(world)]
    outs = [p.communicate(timeout=timeout) for p in procs]
    assert [p.returncode for p in procs] == [0] * world, outs
    stats = {}
    for _, err in outs:
        for line in err.splitlines():
            mt = re.match(r"\[rank (\d+)\] stats: (.*)", line)
            if mt:
                stats[int(mt.group(1))] = dict(kv.split("=") for kv in mt.group(2).split())
    assert sorted(stats) == list(range(world)), outs
    return stats


@pytest.mark.parametrize("world,shards,early_stop", [(2, 2, False), (4, 2, False), (8, 8, False), (8, 2, False),
                                                     (4, 2, True), (8, 2, True), (2, 2, True)])
def test_site_sharded_lock_step_equals_the_sequential_sharded_search(tmp_path, world, shards, early_stop):
    """c4's layout (site blocks only: 2/2, 8/8) and c5's grid (candidate groups x site blocks:
    4/2, 8/2).  Same G, same reducer: the sums are the same numbers in the same order, so the
    lock-stepped records must be the sequential ones BIT FOR BIT whatever the tolerances; every
    rank of a site group ends with the same bits; one collective per round instead of one per
    request.  early_stop: BASELINE config c5's own mode, `--exhaustive --early-stop`
    (src/model.cpp:1187-1197) -- the break changes a candidate's request sequence, which the rounds
    must reproduce identically on every rank."""
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/liblbfgsb_ref.so not built")
    msa, tree, n_roots = MSA, TREE, 17
    if world == shards == 8:   # (sixteen processes on one device: seven of the ten taxa, 11 candidates)
        keep = "iebdjhg"
        rows = [ln.rstrip("\n") for ln in open(MSA)]
        seqs, name = {}, None
        for ln in rows:
            if ln.startswith(">"):
                name = ln[1:].strip()
            elif name is not None:
                seqs[name] = seqs.get(name, "") + ln.strip()
        msa, tree, n_roots = str(tmp_path / "7.fasta"), str(tmp_path / "7.tree"), 11
        with open(msa, "w") as f:
            for k in keep:
                f.write(">%s\n%s\n" % (k, seqs[k]))
        with open(tree, "w") as f:
            f.write("((((i:0.5697,e:0.3666):0.6028,b:0.4459):0.0993,d:0.6396):0.1579,((j:0.8547,h:0.9835):0.4461,g:0.4874):0.6673);\n")
    common = ["--msa", msa, "--tree", tree, "--exhaustive", "--silent", "--rate-cats", "4",
              "--atol", "0.5", "--brtol", "0.1", "--bfgstol", "0.5", "--factor", "1e15",
              "--seed", "5", "--lbfgsb", REF, "--device", "0", "--site-shards", str(shards),
              "--site-reduce", "host", "--stats"] + (["--early-stop"] if early_stop else [])
    seq, lock = str(tmp_path / "seq"), str(tmp_path / "lock")
    st_seq = _run_ranks([RD] + common + ["--prefix", seq, "--lockstep", "0"], world)
    in_flight = 8 if world < 8 else 4       # (eight processes x eight replicas on one device take minutes)
    # a site-sharded model's default is ONE worker group; the c4-style layouts also run with two
    # alternating groups (the turn protocol between them is what keeps the ranks' collectives in step)
    groups_opt = ["--lockstep-groups", "2"] if shards == world else []
    per_group = in_flight // 2 if groups_opt else in_flight
    st_lock = _run_ranks([RD] + common + ["--prefix", lock, "--lockstep", str(in_flight)] + groups_opt, world)
    ra = sorted(rd.Checkpoint(seq).read_results())
    rb = sorted(rd.Checkpoint(lock).read_results())
    assert [r[0] for r in ra] == list(range(n_roots))
    assert ra == rb                                         # ids, lnL, alpha, parameters: same bits
    groups = world // shards
    for g in range(groups):
        members = range(g * shards, (g + 1) * shards)
        assert len({st_lock[r]["results_digest"] for r in members}) == 1      # the ranks of a group agree
        assert len({st_lock[r]["collectives"] for r in members}) == 1
        assert st_lock[g * shards]["results_digest"] == st_seq[g * shards]["results_digest"]
    per_request = int(st_seq[0]["own_collectives"])
    per_round = int(st_lock[0]["collectives"])
    # (the model's own: the empirical frequencies, model_t::initialize and the group's agreement
    # on the number of candidates in flight, before the search)
    assert int(st_lock[0]["own_collectives"]) <= 3 and per_round > 0
    # a worker group of W candidates: up to W candidates' requests per collective (fewer towards
    # the end of a list; measured 2.7 of 3 with two groups of three)
    assert per_request > 0.45 * per_group * per_round, (per_request, per_round, per_group)
    assert open(seq + ".rooted.tree").read() == open(lock + ".rooted.tree").read()


def test_early_stop_records_differ_from_the_plain_exhaustive_ones(tmp_path):
    """(guards the parametrisation above: at its tolerances the early-stop rule does fire)"""
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/liblbfgsb_ref.so not built")
    common = ["--msa", MSA, "--tree", TREE, "--exhaustive", "--silent", "--rate-cats", "4", "--atol", "0.5", "--brtol", "0.1",
              "--bfgstol", "0.5", "--factor", "1e15", "--seed", "5", "--lbfgsb", REF, "--device", "0", "--lockstep", "8",
              "--lockstep-rounds", "1"]
    a, b = str(tmp_path / "plain"), str(tmp_path / "early")
    for prefix, extra in ((a, []), (b, ["--early-stop"])):
        out = subprocess.run([RD] + common + ["--prefix", prefix] + extra, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout + out.stderr
    ra, rb = sorted(rd.Checkpoint(a).read_results()), sorted(rd.Checkpoint(b).read_results())
    assert [r[0] for r in ra] == [r[0] for r in rb] == list(range(17))
    assert ra != rb


@pytest.mark.parametrize("lockstep", ["8", "0"])
def test_one_ulp_on_one_rank_ends_the_run_at_once_naming_the_round(tmp_path, lockstep):
    """The rounds rest on every rank of a site group receiving the same BITS from the reducer
    (csrc/lockstep_conductor.hpp, header).  Here rank 1's copy of the sums of its 40th reduction
    is off by one unit in the last place (rendezvous.hpp's fault hook) -- what an all-reduce
    without a bit-identity promise may deliver.  Unnoticed, the two ranks' optimisers part ways
    and some later round no longer matches: a hang until the communicator's time limit.  With the
    guard the very next round of that worker group fails on BOTH ranks, by its number, in seconds.
    lockstep 0: the sequential sharded search (a collective per request) -- the model's own
    reductions carry the same two guard words (csrc/model.cpp, guard_check) and name the reduction."""
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/liblbfgsb_ref.so not built")
    import time
    args = [RD, "--msa", MSA, "--tree", TREE, "--exhaustive", "--silent", "--rate-cats", "4", "--atol", "0.5", "--brtol", "0.1",
            "--bfgstol", "0.5", "--factor", "1e15", "--seed", "5", "--lbfgsb", REF, "--device", "0", "--site-shards", "2",
            "--site-reduce", "host", "--lockstep", lockstep, "--prefix", str(tmp_path / "f")]
    s = __import__("socket").socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GPU_MAX_HW_QUEUES="1",
               RDAMD_FAULT_ULP="1:40")
    t0 = time.time()
    procs = [subprocess.Popen(args, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=120) for p in procs]
    took = time.time() - t0
    assert all(p.returncode != 0 for p in procs), outs
    text = "".join(o + e for o, e in outs)
    if lockstep == "0":
        mt = re.search(r"site-group reduction (\d+): the ranks did not receive the same bits from the previous reduction", text)
    else:
        mt = re.search(r"lock-step round (\d+) of worker group \d+ .*has diverged -- the ranks did not receive the same bits", text)
    assert mt, text[-3000:]
    assert took < 60, took                  # (start-up included; the comm time limit is 600 s)


def test_a_site_block_that_lacks_a_state(tmp_path):
    """A replica takes the GROUP's empirical frequencies over from the model it is made from and
    must keep them across its own tip load: here the second block of columns holds
    no T at all -- frequencies computed from that block alone have a zero entry ("One of the state
    frequenices is zero") and would differ per rank.  The lock-stepped search must give the
    sequential sharded search's records, bit for bit."""
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/liblbfgsb_ref.so not built")
    seqs = util.read_fasta(MSA)
    n = len(next(iter(seqs.values())))
    msa = str(tmp_path / "noT.fasta")
    with open(msa, "w") as f:
        for k, v in seqs.items():
            half = v[n // 2:].replace("T", "C").replace("t", "c")
            f.write(">%s\n%s\n" % (k, v[:n // 2] + half))
    common = ["--msa", msa, "--tree", TREE, "--exhaustive", "--silent", "--rate-cats", "4", "--atol", "0.5", "--brtol", "0.1",
              "--bfgstol", "0.5", "--factor", "1e15", "--seed", "5", "--lbfgsb", REF, "--device", "0", "--site-shards", "2",
              "--site-reduce", "host", "--stats"]
    seq, lock = str(tmp_path / "seq"), str(tmp_path / "lock")
    _run_ranks([RD] + common + ["--prefix", seq, "--lockstep", "0"], 2)
    st = _run_ranks([RD] + common + ["--prefix", lock, "--lockstep", "6"], 2)
    assert st[0]["results_digest"] == st[1]["results_digest"]
    assert sorted(rd.Checkpoint(seq).read_results()) == sorted(rd.Checkpoint(lock).read_results())
