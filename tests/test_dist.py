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
