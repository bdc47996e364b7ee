"""N > 1 paths on CPU: world-size-2 `gloo` process groups exercising the work
split (candidate roots, site blocks, 2-D grid) and the lnL all-reduce.  The
CPU oracle stands in for the per-rank compute (this is a test of the sharding
and reduction logic, which is identical under RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as tdist
import torch.multiprocessing as mp

import root_digger_amd as rd
from root_digger_amd import dist as rdist
import util

HERE = os.path.dirname(os.path.abspath(__file__))


def test_chunking_matches_reference_formula():
    # src/model.cpp:1899-1907: chunk*rank + min(mod, rank)
    for count in (0, 1, 7, 17, 197, 1997):
        for world in (1, 2, 3, 8):
            seen = []
            for rank in range(world):
                beg, end = rdist.chunk(count, rank, world)
                size, mod = count // world, count % world
                assert beg == size * rank + min(mod, rank)
                assert end - beg in (size, size + 1)
                seen += list(range(beg, end))
            assert seen == list(range(count))


def test_assign_candidates_skips_completed():
    done = [3, 4, 10]
    parts = [rdist.assign_candidates(17, r, 4, done) for r in range(4)]
    flat = sorted(sum(parts, []))
    assert flat == [i for i in range(17) if i not in done]
    assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_grid_2d():
    assert rdist.grid_2d(8, 2) == (4, 2)
    assert [rdist.rank_coords(r, 2) for r in range(4)] == [(0, 0), (0, 1), (1, 0), (1, 1)]
    with pytest.raises(ValueError):
        rdist.grid_2d(8, 3)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, HERE)
    from oracle_lib import OraclePartition, ORC_MAP_NT
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tree = rd.Tree.from_file(os.path.join(util.DATA, "10.tree"))
        seqs = util.read_fasta(os.path.join(util.DATA, "10.fasta"))
        subst = [.34, .42, .24, .74, .16, .88, .75, .54, .20, .06, .08, .41]
        rates = rd.compute_gamma_cats(1.0, 4)

        def lnl_of(sub, roots, freqs=None):
            n = len(next(iter(sub.values())))
            p = OraclePartition.for_tree(tree, 4, n, 4)
            util.load_tips(p, tree, sub, ORC_MAP_NT)
            p.set_subst_params(0, subst)
            p.set_category_rates(rates)
            if freqs is not None:
                p.set_frequencies(0, freqs)
            return [util.compute_lh(p, tree, tree.root_location(i)) for i in roots]

        def empirical(sub):
            n = len(next(iter(sub.values())))
            p = OraclePartition.for_tree(tree, 4, n, 4)
            util.load_tips(p, tree, sub, ORC_MAP_NT)
            return p.empirical_frequencies()

        # --- site-block sharding: every rank evaluates ALL jobs on its slice,
        # one all-reduce of the per-job partial lnLs (BASELINE config c4 pattern)
        jobs = [0, 5, 16]
        lo, hi = rdist.site_block(1000, rank, world)
        part = torch.tensor(lnl_of({k: v[lo:hi] for k, v in seqs.items()}, jobs),
                            dtype=torch.float64)
        rdist.allreduce_lnl(part)
        # the model of a site-sharded run is global: empirical frequencies of the
        # whole alignment from the blocks' own figures, then the same check again
        block = {k: v[lo:hi] for k, v in seqs.items()}
        gfreq = rdist.global_frequencies(empirical(block), hi - lo)
        part_f = torch.tensor(lnl_of(block, jobs, gfreq), dtype=torch.float64)
        rdist.allreduce_lnl(part_f)
        # --- candidate sharding: disjoint roots per rank, gathered at the end
        mine = rdist.assign_candidates(tree.root_count(), rank, world)
        vals = torch.full((tree.root_count(),), 0.0, dtype=torch.float64)
        for i, v in zip(mine, lnl_of(seqs, mine)):
            vals[i] = v
        tdist.all_reduce(vals)           # disjoint support -> a gather
        if rank == 0:
            whole = lnl_of(seqs, jobs)
            every = lnl_of(seqs, range(tree.root_count()))
            wfreq = empirical(seqs)
            out.put(("ok", part.tolist(), whole, vals.tolist(), every,
                     gfreq, wfreq, part_f.tolist(), lnl_of(seqs, jobs, wfreq)))
    except Exception as e:   # pragma: no cover
        if rank == 0:
            out.put(("err", repr(e)))
        raise
    finally:
        tdist.destroy_process_group()


def test_world_size_2_gloo_site_and_candidate_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == "ok", res
    _, sharded, whole, gathered, every, gfreq, wfreq, sharded_f, whole_f = res
    for a, b in zip(sharded, whole):
        assert util.rel_err(a, b) < 1e-13
    for a, b in zip(gathered, every):
        assert a == b
    assert max(abs(a - b) for a, b in zip(gfreq, wfreq)) < 1e-15
    for a, b in zip(sharded_f, whole_f):
        assert util.rel_err(a, b) < 1e-12


def _grid_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cgroups, sgroups = rdist.grid_2d(world, 2)
        cgroup, srank = rdist.rank_coords(rank, sgroups)
        mine = None
        for c in range(cgroups):
            g = tdist.new_group(list(range(c * sgroups, (c + 1) * sgroups)))
            if c == cgroup:
                mine = g
        # per-block "lnLs" of this group's two jobs: group c, shard s contributes (c+1)*10^s
        part = torch.tensor([(cgroup + 1) * 10.0 ** srank, (cgroup + 1) * 2 * 10.0 ** srank],
                            dtype=torch.float64)
        rdist.allreduce_lnl(part, mine)
        freqs = rdist.global_frequencies([0.1 * (srank + 1), 1 - 0.1 * (srank + 1)],
                                         100 * (srank + 1), group=mine)
        gathered = [None] * world
        tdist.all_gather_object(gathered, (cgroup, srank, part.tolist(), freqs))
        if rank == 0:
            out.put(("ok", gathered))
    except Exception as e:   # pragma: no cover
        if rank == 0:
            out.put(("err", repr(e)))
        raise
    finally:
        tdist.destroy_process_group()


def test_world_size_4_grid_of_candidate_groups_and_site_shards():
    """BASELINE config c5's layout: 2 candidate groups x 2 site shards; the
    all-reduce and the global frequencies stay inside a candidate group."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grid_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == "ok", res
    for cgroup, srank, part, freqs in res[1]:
        assert part == [(cgroup + 1) * 11.0, (cgroup + 1) * 22.0]            # 10^0 + 10^1 per job
        want0 = (0.1 * 100 + 0.2 * 200) / 300
        assert abs(freqs[0] - want0) < 1e-15 and abs(sum(freqs) - 1.0) < 1e-15


def _order_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # contributions whose sum depends on the association: rank order gives
        # ((1e16 + 1) - 1e16) + 1 = 1, pairwise (1e16 + 1) + (-1e16 + 1) = 0, and so on
        table = [[1e16, 0.1, 3.0], [1.0, 0.2, 1e-17], [-1e16, 0.3, -3.0], [1.0, 0.4, 1e-17]]
        part = torch.tensor(table[rank], dtype=torch.float64)
        rdist.allreduce_lnl(part)
        gathered = [None] * world
        tdist.all_gather_object(gathered, part.tolist())
        if rank == 0:
            out.put(("ok", gathered, table))
    except Exception as e:   # pragma: no cover
        if rank == 0:
            out.put(("err", repr(e)))
        raise
    finally:
        tdist.destroy_process_group()


def test_site_group_sum_is_the_rank_order_sum_on_every_rank():
    """The optimisers branch on the group's sums, so every rank must hold the same BITS and the
    bits must not depend on the collective's algorithm: allreduce_lnl gathers and adds in rank
    order -- the sum csrc/comm.cpp's RDAMD_COMM_SUM_GATHER makes on the device and rd_amd's host
    reducer makes through its leader (reference split: src/model.cpp:1867-1911)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_order_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == "ok", res
    _, gathered, table = res
    want = []
    for i in range(3):
        acc = table[0][i]
        for r in range(1, 4):
            acc = acc + table[r][i]
        want.append(acc)
    assert want[0] == 1.0
    for got in gathered:
        assert got == want                    # bit for bit, on every rank


def test_bench_parent_spawns_one_child_per_gpu(tmp_path):
    """`python bench.py --gpus 2` with no launcher must start two ranks itself
    (it used to run one rank and print n_gpus 1).  Without a GPU the
    children stop at bench.py's own "needs a HIP device" check: the parent must
    pass that failure on instead of printing a 1-rank line."""
    import subprocess
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2",
                          "--config", "c1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=300, env=env)
    if rd.device_count() > 0:
        pytest.skip("GPU present: covered by tests/test_gpu_bench.py")
    assert out.returncode != 0
    assert '"n_gpus"' not in out.stdout
    assert out.stderr.count("bench.py needs a HIP device") == 2     # one message per child rank


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    import subprocess
    root = os.path.dirname(HERE)
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"],
                         capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode != 0 and "WORLD_SIZE (1) != --gpus (2)" in out.stderr


@pytest.mark.parametrize("world,victim", [(4, 2), (3, 0)])
def test_rd_amd_ranks_notice_a_dead_rank(tmp_path, world, victim):
    """a rank that dies mid-search must not leave the others inside a collective.
    rd_amd's ranks watch their rendezvous connections while they search
    (rendezvous_t::watch); the handler aborts the RCCL communicator and exits.  Here: the
    victim dies silently, every survivor's handler must run within seconds -- through rank 0
    when the victim is not rank 0 (two hops)."""
    import subprocess
    import time
    root = os.path.dirname(HERE)
    exe = str(tmp_path / "rendezvous_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I",
                           os.path.join(root, "root_digger_amd", "csrc", "tools"),
                           os.path.join(HERE, "cpp", "rendezvous_check.cpp"), "-o", exe, "-lpthread"])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    t0 = time.monotonic()
    procs = [subprocess.Popen([exe, str(r), str(world), "watch", str(victim)], env=env,
                              stdout=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=60)[0] for p in procs]
    assert time.monotonic() - t0 < 15
    for r, (p, out) in enumerate(zip(procs, outs)):
        if r == victim:
            assert p.returncode == 9
        else:
            assert p.returncode == 3 and out.strip() == "%d lost" % r, (r, p.returncode, out)


def _rendezvous_exe(tmp_path):
    import subprocess
    root = os.path.dirname(HERE)
    exe = str(tmp_path / "rendezvous_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I",
                           os.path.join(root, "root_digger_amd", "csrc", "tools"),
                           os.path.join(HERE, "cpp", "rendezvous_check.cpp"), "-o", exe, "-lpthread"])
    return exe


@pytest.mark.parametrize("world,odd", [(2, 1), (3, 0), (4, 2)])
def test_host_site_group_refuses_vectors_of_different_length(tmp_path, world, odd):
    """Ranks whose rounds have diverged arrive with vectors of different length: a device
    collective would hang or corrupt there, the host reducer's leader compares the lengths and
    refuses -- and no member is left waiting (the leader's failure closes the group's sockets)."""
    import subprocess
    import time
    exe = _rendezvous_exe(tmp_path)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    t0 = time.monotonic()
    procs = [subprocess.Popen([exe, str(r), str(world), "mismatch", str(odd)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=60) for p in procs]
    assert time.monotonic() - t0 < 15
    for r, (p, (out, err)) in enumerate(zip(procs, outs)):
        assert p.returncode == 5 and out.strip() == "%d refused" % r, (r, p.returncode, out, err)
    assert "differ in length" in outs[0][1]


@pytest.mark.parametrize("world,absent", [(2, 1), (3, 0), (4, 3)])
def test_host_site_group_abort_hook_frees_a_blocked_reducer(tmp_path, world, absent):
    """rdamd_model_set_lnl_reducer_abort's contract on the host reducer: a rank that waits in a
    reduction another rank never joins (that rank's round failed a moment earlier) gets out as
    soon as its own abort hook is called from another thread."""
    import subprocess
    import time
    exe = _rendezvous_exe(tmp_path)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    t0 = time.monotonic()
    procs = [subprocess.Popen([exe, str(r), str(world), "abort", str(absent)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=60) for p in procs]
    assert time.monotonic() - t0 < 15
    for r, (p, (out, err)) in enumerate(zip(procs, outs)):
        if r == absent:
            assert p.returncode == 0
        else:
            assert p.returncode == 6 and out.strip() == "%d aborted" % r, (r, p.returncode, out, err)


@pytest.mark.parametrize("world,group", [(2, 2), (4, 2), (3, 1)])
def test_rd_amd_rendezvous_and_host_site_group_sum(tmp_path, world, group):
    """rd_amd's own channel (csrc/tools/rendezvous.hpp): TCP star allgather + the
    host-side lnL sum inside a site group (`--site-reduce host`), as `world`
    plain processes on CPU.  Sums are formed in rank order by the group leader and
    are identical on every member."""
    import subprocess
    root = os.path.dirname(HERE)
    exe = str(tmp_path / "rendezvous_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I",
                           os.path.join(root, "root_digger_amd", "csrc", "tools"),
                           os.path.join(HERE, "cpp", "rendezvous_check.cpp"), "-o", exe, "-lpthread"])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    procs = [subprocess.Popen([exe, str(r), str(world), str(group)], env=env,
                              stdout=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=120)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs)
    for r, out in enumerate(outs):
        rank, gather, total = out.split()
        assert int(rank) == r
        assert gather == "gather=" + ",".join(str(100 + i) for i in range(world))
        members = range(r - r % group, r - r % group + group)
        want = [sum(1.0 + m for m in members), sum(0.5 * m for m in members), 1e-3 * group]
        got = [float(x) for x in total[len("sum="):].split(",")]
        assert got[:2] == want[:2] and abs(got[2] - want[2]) < 1e-18
        assert total == outs[r - r % group].split()[2]        # same bits as the group leader
