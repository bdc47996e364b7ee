"""Multi-GPU pre-flight on a one-GPU box: the 8-rank forms of BASELINE
configs c4 / c5 -- candidate groups x site shards -- run here as 8 processes that share device
0 (gloo / TCP sums instead of RCCL, which refuses two ranks on one device); what must hold is
that the ranks find each other, split the work as designed and reproduce the one-rank numbers.
No scaling curve is claimed: the driver measures that on real hardware."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import root_digger_amd as rd
from root_digger_amd import synth
import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RD = os.path.join(ROOT, "root_digger_amd", "bin", "rd_amd")


def run_bench(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline",
                          "--sustain-seconds", "0", "--warmup", "1"] + list(extra),
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_eight_rank_grid_bench_on_the_c5_tree():
    """bench.py --gpus 8 --shard grid --site-groups 2 (c5's layout: 4 candidate groups x 2 site
    shards, the all-reduce inside a group) on c5's 1000-taxon tree at 512 sites: rank 0's
    checksum -- the all-reduced lnLs of candidate group 0 -- equals what ONE rank computes for
    that group's candidates on the whole alignment."""
    shape = ["--config", "c5", "--sites", "512", "--steps", "2", "--batch", "8"]
    g = run_bench("--gpus", "8", "--dist-backend", "gloo", "--device", "0", "--shard", "grid",
                  "--site-groups", "2", *shape)
    one = run_bench("--as-candidate-group", "0/4", *shape)
    assert g["n_gpus"] == 8 and g["scaling"] == "weak"
    assert g["config"]["sharding"].startswith("4 candidate groups x 2 site shards")
    assert abs(g["lnl_check"] - one["lnl_check"]) <= 1e-9 * abs(one["lnl_check"])
    # c4's layout: 8 site shards, one candidate group
    s = run_bench("--gpus", "8", "--dist-backend", "gloo", "--device", "0", "--shard", "sites", *shape)
    full = run_bench(*shape)
    assert s["n_gpus"] == 8 and s["scaling"] == "strong"
    assert abs(s["lnl_check"] - full["lnl_check"]) <= 1e-9 * abs(full["lnl_check"])


def test_default_eight_rank_line_carries_the_site_sharded_and_grid_legs():
    """what the DRIVER runs -- `bench.py --gpus 8`, no --shard -- must measure
    what north_star names.  Beside the candidate-sharded value (weak scaling, no data-path
    collective) the line carries `site_sharded` (the same workload, every rank a site block, an
    all-reduce of the per-block lnLs per batch) and `grid` (4 candidate groups x 2 site shards);
    their checksums are the one-rank run's.  (gloo here: eight ranks share device 0, which RCCL
    refuses; the driver's run goes through RCCL and adds `rccl_ranks`.)"""
    shape = ["--config", "c5", "--sites", "512", "--steps", "2", "--batch", "8"]
    d = run_bench("--gpus", "8", "--dist-backend", "gloo", "--device", "0", *shape)
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["config"]["sharding"] == "candidate roots"
    full = run_bench(*shape)
    ss, gr = d["site_sharded"], d["grid"]
    assert ss["scaling"] == "strong" and ss["sites_per_rank"] == 64 and ss["value"] > 0
    assert abs(ss["lnl_check"] - full["lnl_check"]) <= 1e-9 * abs(full["lnl_check"])
    one = run_bench("--as-candidate-group", "0/4", *shape)
    assert gr["scaling"] == "weak" and gr["sharding"].startswith("4 candidate groups x 2 site shards")
    assert gr["sites_per_rank"] == 256
    assert abs(gr["lnl_check"] - one["lnl_check"]) <= 1e-9 * abs(one["lnl_check"])
    # rank 0's own candidates (the first eighth) in the headline value's checksum
    mine = run_bench("--as-candidate-group", "0/8", *shape)
    assert abs(d["lnl_check"] - mine["lnl_check"]) <= 1e-9 * abs(mine["lnl_check"])
    # two ranks: no grid (one candidate group would be left), the site-sharded leg only
    d2 = run_bench("--gpus", "2", "--dist-backend", "gloo", "--device", "0", *shape)
    assert "site_sharded" in d2 and "grid" not in d2
    assert abs(d2["site_sharded"]["lnl_check"] - full["lnl_check"]) <= 1e-9 * abs(full["lnl_check"])


def test_eight_rank_site_sharded_rd_amd_on_the_c5_tree(tmp_path):
    """rd_amd --site-shards 2 --site-reduce host with 8 ranks (4 x 2 grid) on c5's tree at 192
    sites: the checkpoint holds the one-rank run's records (root position optimised at every
    one of the 1 997 candidates; no parameter optimiser in the loop, so the values agree to
    the summation order of the two column blocks)."""
    w = synth.workload(1000, 192, 4, 4, 0xD166E5 + 4)
    msa, tree = str(tmp_path / "c5.fasta"), str(tmp_path / "c5.tree")
    with open(msa, "w") as f:
        for k, v in w["seqs"].items():
            f.write(">%s\n%s\n" % (k, v))
    open(tree, "w").write(w["newick"])
    common = [RD, "--msa", msa, "--tree", tree, "--exhaustive", "--silent", "--rate-cats", "4",
              "--atol", "1e-6", "--brtol", "1e-9", "--seed", "5", "--device", "0", "--threads", "0",
              "--lockstep", "0"]
    one, many = str(tmp_path / "one"), str(tmp_path / "many")
    out = subprocess.run(common + ["--prefix", one], capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, WORLD_SIZE="8", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    procs = [subprocess.Popen(common + ["--prefix", many, "--site-shards", "2", "--site-reduce", "host"],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(8)]
    outs = [p.communicate(timeout=1200)[0] for p in procs]
    assert [p.returncode for p in procs] == [0] * 8, outs
    ra = sorted(rd.Checkpoint(one).read_results())
    rb = sorted(rd.Checkpoint(many).read_results())
    assert [r[0] for r in ra] == [r[0] for r in rb] == list(range(1997))
    lnl_a, lnl_b = np.array([r[1] for r in ra]), np.array([r[1] for r in rb])
    assert np.max(np.abs(lnl_a - lnl_b) / np.abs(lnl_a)) < 1e-8
    assert max(abs(a[2] - b[2]) for a, b in zip(ra, rb)) < 1e-3           # alpha (atol 1e-6 on the derivative)
    assert int(np.argmax(lnl_a)) == int(np.argmax(lnl_b))


COMM_CHILD = r"""
import sys, time
sys.path.insert(0, %r)
import root_digger_amd as rd
rd.set_device(0)
rank = int(sys.argv[1])
uid = bytes.fromhex(open(sys.argv[2]).read()) if rank else rd.Comm.unique_id()
if rank == 0:
    open(sys.argv[2] + ".tmp", "w").write(uid.hex())
    import os
    os.replace(sys.argv[2] + ".tmp", sys.argv[2])
t0 = time.time()
try:
    rd.Comm(uid, rank, 2)
    print("CREATED")
except rd.RdamdError as e:
    print("REFUSED after %%.1f s: %%s" %% (time.time() - t0, e))
"""


def test_two_rccl_ranks_on_one_device_fail_loudly(tmp_path):
    """rdamd_comm_create(n_ranks = 2) from two processes that both sit on device 0: RCCL does
    not form such a communicator.  Both calls must come back with an error that says so
    (rdamd_errmsg) -- not hang, not crash."""
    script, idfile = str(tmp_path / "child.py"), str(tmp_path / "id.hex")
    open(script, "w").write(COMM_CHILD % ROOT)
    env = dict(os.environ, NCCL_DEBUG="WARN", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p0 = subprocess.Popen([sys.executable, script, "0", idfile], stdout=subprocess.PIPE,
                          stderr=subprocess.STDOUT, text=True, env=env)
    import time
    for _ in range(600):
        if os.path.exists(idfile):
            break
        time.sleep(0.1)
    assert os.path.exists(idfile), "rank 0 never produced a unique id"
    p1 = subprocess.Popen([sys.executable, script, "1", idfile], stdout=subprocess.PIPE,
                          stderr=subprocess.STDOUT, text=True, env=env)
    try:
        outs = [p.communicate(timeout=180)[0] for p in (p0, p1)]
    except subprocess.TimeoutExpired:
        for p in (p0, p1):
            p.kill()
        pytest.fail("rdamd_comm_create with two ranks on one device hangs")
    for out in outs:
        assert "REFUSED" in out and "ncclCommInitRank" in out, outs
