"""root_digger_amd/bin/rd_amd: the native `rd` front end (csrc/tools/rd_main.cpp,
built on include/root_digger_amd.h alone) against the Python command line on
the same options -- same checkpoint records, same trees."""
import os
import subprocess
import sys

import pytest

import root_digger_amd as rd
import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RD = os.path.join(ROOT, "root_digger_amd", "bin", "rd_amd")
REF = os.path.join(ROOT, "oracle", "_ref", "liblbfgsb_ref.so")
MSA, TREE = os.path.join(util.DATA, "10.fasta"), os.path.join(util.DATA, "10.tree")


def test_native_front_end_matches_the_python_one(tmp_path, capsys):
    """root_digger_amd.cli is a LAUNCHER of bin/rd_amd (argument pass-through + the rank
    environment): the same records, the same trees, the same text on stdout -- and no search
    driver of its own"""
    from root_digger_amd import cli
    assert os.path.exists(RD), "make -C root_digger_amd/csrc builds it"
    src = open(cli.__file__).read()
    assert len(src.splitlines()) < 80 and "exhaustive_search" not in src and "Model" not in src
    common = ["--msa", MSA, "--tree", TREE, "--exhaustive", "--silent", "--rate-cats", "4",
              "--atol", "1e-3", "--brtol", "1e-3", "--bfgstol", "1e-3", "--factor", "1e12",
              "--seed", "5"]
    if os.path.exists(REF):
        common += ["--lbfgsb", REF]
    a, b = str(tmp_path / "native"), str(tmp_path / "python")
    out = subprocess.run([RD] + common + ["--prefix", a], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    capsys.readouterr()
    assert cli.main(common + ["--prefix", b]) == 0
    assert capsys.readouterr().out == out.stdout.replace(a, b)      # (relayed through sys.stdout)
    ra = sorted(rd.Checkpoint(a).read_results())
    rb = sorted(rd.Checkpoint(b).read_results())
    assert [r[:3] for r in ra] == [r[:3] for r in rb] and len(ra) == 17
    assert [r[3] for r in ra] == [r[3] for r in rb]
    assert open(a + ".rooted.tree").read() == open(b + ".rooted.tree").read()
    assert open(a + ".lwr.tree").read() == open(b + ".lwr.tree").read()
    assert out.stdout.strip().splitlines()[-1] == open(a + ".lwr.tree").read().strip()
    # `--workers` is `--threads`; `--gpus 2`: the launcher starts two ranks with the rank environment
    c = str(tmp_path / "two")
    assert cli.main(common + ["--prefix", c, "--gpus", "2", "--device", "0", "--workers", "0", "--lockstep", "0"]) == 0
    rc = sorted(rd.Checkpoint(c).read_results())
    assert [r[0] for r in rc] == list(range(17))
    for x, y in zip(rc, ra):
        assert abs(x[1] - y[1]) <= 1e-6 * abs(y[1])
    # a rerun finds everything in the log
    again = subprocess.run([RD, "--msa", "x", "--tree", "y", "--prefix", a, "--silent"],
                           capture_output=True, text=True, timeout=600)
    assert again.returncode == 0 and len(rd.Checkpoint(a).read_results()) == 17


def test_native_front_end_two_ranks_and_search_mode(tmp_path):
    env = dict(os.environ, WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
    two = str(tmp_path / "two")
    args = [RD, "--msa", MSA, "--tree", TREE, "--exhaustive", "--silent", "--atol", "1e-3",
            "--brtol", "1e-3", "--prefix", two, "--device", "0", "--threads", "0"]
    procs = [subprocess.Popen(args, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=120)[0] for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
    recs = sorted(rd.Checkpoint(two).read_results())
    assert [r[0] for r in recs] == list(range(17))
    assert outs[1].strip() == "" and outs[0].strip().startswith("(")      # only rank 0 prints the tree
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/liblbfgsb_ref.so not built")
    s = str(tmp_path / "search")
    out = subprocess.run([RD, "--msa", MSA, "--tree", TREE, "--prefix", s, "--silent", "--lbfgsb", REF,
                          "--min-roots", "2", "--initial-root-strategy", "midpoint", "--atol", "1e-3",
                          "--bfgstol", "1e-3", "--brtol", "1e-3", "--factor", "1e12"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert len(rd.Checkpoint(s).read_results()) == 2 and not os.path.exists(s + ".lwr.tree")
    assert rd.Tree.from_newick(open(s + ".rooted.tree").read()).tip_count() == 10


def _run_ranks(args, world, timeout=600):
    s = __import__("socket").socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    procs = [subprocess.Popen(args, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = [p.communicate(timeout=timeout)[0] for p in procs]
    assert [p.returncode for p in procs] == [0] * world, outs
    return outs


@pytest.mark.parametrize("world,shards", [(2, 2)])
def test_site_sharded_exhaustive_search_writes_the_one_rank_records(tmp_path, world, shards):
    """`rd_amd --site-shards G` (north star: site blocks + lnL all-reduce inside the
    product): 2 ranks = one site group of two column blocks (BASELINE c4's layout),
    4 ranks = 2 candidate groups x 2 site blocks (c5's 2-D grid).  All ranks share
    device 0 of this box, so the sums go through `--site-reduce host`; the model
    hook is the one RCCL uses.  The checkpoint must hold the one-rank run's
    records: one per candidate, same root ids, lnL/alpha to optimiser tolerance.
    The sharded side runs in LOCK STEP (rd_amd's default with --lbfgsb: deterministic
    rounds, tests/test_gpu_lockstep_rounds.py, which also covers 4/2, 8/8 and 8/2 bit
    for bit against the sequential sharded search)."""
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/liblbfgsb_ref.so not built")
    # (tight optimiser settings: the block sums differ from the one-rank sum in the
    # last bit, and a loosely converged L-BFGS-B run would amplify that)
    common = ["--msa", MSA, "--tree", TREE, "--exhaustive", "--silent", "--rate-cats", "4",
              "--atol", "1e-7", "--brtol", "1e-9", "--bfgstol", "1e-7", "--factor", "1e4",
              "--seed", "5", "--lbfgsb", REF, "--device", "0", "--threads", "0"]
    one, many = str(tmp_path / "one"), str(tmp_path / "many")
    out = subprocess.run([RD] + common + ["--prefix", one, "--lockstep", "0"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    outs = _run_ranks([RD] + common + ["--prefix", many, "--site-shards", str(shards),
                                       "--site-reduce", "host", "--lockstep", "8"], world)
    ra = sorted(rd.Checkpoint(one).read_results())
    rb = sorted(rd.Checkpoint(many).read_results())
    assert [r[0] for r in rb] == [r[0] for r in ra] == list(range(17))      # one record per candidate
    for a, b in zip(ra, rb):
        assert abs(a[1] - b[1]) <= 2e-6 * abs(a[1]), (a, b)                 # lnL
        assert abs(a[2] - b[2]) <= 2e-2, (a, b)                             # alpha
    assert all(o.strip() == "" for o in outs[1:]) and outs[0].strip().startswith("(")
    ta = rd.Tree.from_newick(open(one + ".rooted.tree").read())
    tb = rd.Tree.from_newick(open(many + ".rooted.tree").read())
    assert ta.tip_count() == tb.tip_count() == 10
    best_a, best_b = max(ra, key=lambda r: r[1]), max(rb, key=lambda r: r[1])
    assert best_a[0] == best_b[0]                                           # same best root


def test_site_shards_must_divide_the_world(tmp_path):
    out = subprocess.run([RD, "--msa", MSA, "--tree", TREE, "--exhaustive", "--silent", "--no-checkpoint",
                          "--site-shards", "2"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 1 and "--site-shards must divide" in out.stdout
