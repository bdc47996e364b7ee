"""bench.py's contract on a small configuration: one JSON line with the fields
the driver reads, the roofline and cpu_baseline objects, and the site-sharded
variant (which hands rdamd_evaluate_batch_device a torch device pointer)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*extra, config="c1", steps="3", batch="17", env=None):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", config,
                          "--steps", steps, "--warmup", "1", "--batch", batch] + list(extra),
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_line_has_the_contract_fields():
    d = run_bench("--cpu-seconds", "2")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["metric"] == "candidate-root lnL evals/sec" and d["unit"] == "evals/s"
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["dtype"] == "f64" and d["vs_baseline"] is None and d["scaling"] == "weak"
    assert d["value"] > 0 and "workload" in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    # the fused 4-state kernel is bound by FP64 FMA issue, not by HBM: the headline
    # roofline is flops against the FP64 peak, and no fraction may exceed 1
    assert r["kernel"] == "fused_dna_eval_kernel" and r["bound"] == "fp64"
    assert r["unit"] == "TFLOP/s" and r["peak"] == 78.6 and 0 < r["frac"] <= 1.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "frac" not in r["hbm_equiv"]
    k = d["clv_kernel"]
    assert k["bound"] == "hbm" and k["peak"] == 8000.0 and 0 < k["frac"] <= 1.0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0
    assert c["parity_max_rel_err"] < 1e-9
    # the stated baseline is the oracle's AVX2 loop; host sockets x cores are stated
    assert "AVX2" in c["sample"] and c["scalar_loop_1_thread"] > 0
    assert c["host"]["sockets"] >= 1 and c["host"]["cores_per_socket"] >= 1
    # the many-thread leg names itself by what was measured: "one_socket" only when
    # the threads reached >= 20 x the one-thread rate, "host_threads" (with a note) otherwise
    legs = [k for k in ("one_socket", "host_threads") if k in c]
    assert len(legs) <= 1
    for k in legs:
        assert 1 < c[k]["cores"] <= c["host"]["cores_per_socket"]
        assert c[k]["per_thread_evals_per_s"]["min"] <= c[k]["per_thread_evals_per_s"]["max"]
        assert (k == "one_socket") == (c[k]["ratio_to_one_thread"] >= 20.0 and
                                       c[k]["cores"] >= 0.9 * c["host"]["cores_per_socket"])
    assert "loadavg" in c["host"] and "cgroup_cpu_max" in c["host"] and c["host"]["affinity_cpus"] >= 1
    # the comparator bounded by what bounds a CPU: one socket's DRAM bandwidth over the
    # bytes the oracle's site-repeats traversal must move; the speed-up's lower bound takes the smaller of
    # the two socket bounds and says which
    bb = c["one_socket_bandwidth_bound"]
    by = bb["bytes_per_evaluation"]
    assert by["moved"] == by["written"] + by["read_at_least_once"] and by["read_at_least_once"] <= by["read_if_nothing_is_cached"]
    assert bb["value"] > 0 and bb["bandwidth_gbs"] > 0 and bb["stream_triad"]["gbs"] > 0 and bb["stream_triad"]["threads"] >= 1
    assert abs(bb["value"] - bb["bandwidth_gbs"] * 1e9 / by["moved"]) <= 1e-3 * bb["value"]
    low = min(bb["value"], c["one_socket_ideal"]["value"])
    assert abs(c["speedup_lower_bound"] - d["value"] / low) <= 0.01 * c["speedup_lower_bound"] + 0.01
    assert c["speedup_lower_bound_basis"] in ("one_socket_bandwidth_bound", "one_socket_ideal")
    # BASELINE's second metric: algorithmic-equivalent and executed (site repeats fold operations)
    assert 0 < d["site_clv_updates_per_sec_executed"] <= d["site_clv_updates_per_sec"]
    assert r["schedule"]["max_classes"] == 64       # the library's effective limit, not null


@pytest.mark.parametrize("config,extra", [("c2", []), ("c3", []), ("c4", ["--sites", "20000"]),
                                          ("c5", ["--sites", "20000"])])
def test_no_roofline_fraction_exceeds_one(config, extra):
    """every published fraction is a statement about a kernel against a peak: <= 1 on every
    BASELINE shape (c4 / c5 at a site count that keeps the test short; c1 above).  Where the
    CLV kernel's algorithmic-byte rate passes the nominal HBM peak (reads forwarded on chip),
    the cap is stated."""
    d = run_bench("--no-cpu-baseline", "--sustain-seconds", "0", "--allow-stale-profile", *extra,
                  config=config, steps="2", batch="40")
    assert 0 < d["roofline"]["frac"] <= 1.0
    k = d["clv_kernel"]
    assert 0 < k["frac"] <= 1.0
    assert k.get("counter_frac", 0.0) <= 1.0
    # where the algorithmic-byte rate passes the HBM peak the fraction is made of bytes that cross
    # HBM (the command's counters, or the modelled minimum), never a capped 1.0; the algorithmic
    # figure stays beside it
    if k["frac_basis"] == "algorithmic bytes":
        assert abs(k["frac"] - k["achieved"] / k["peak"]) < 1e-3
    else:
        assert k["algorithmic_equiv"]["ratio_to_peak"] > 1.0 and k["frac"] < 1.0
        assert k["frac_basis"].startswith(("hbm counters", "modelled HBM bytes"))


def test_counters_are_attached_to_the_command_they_were_taken_from_only():
    """profiles/r*_summary.json holds counters of ONE command: c2, batch 197, the whole alignment on
    one rank.  A run with --sites / --shard / --as-candidate-group launches other shapes and must
    publish no `traffic`, no `issue` block and no counter-based fraction."""
    for extra in (["--sites", "6250"], ["--shard", "sites"], ["--as-candidate-group", "0/2"]):
        d = run_bench("--no-cpu-baseline", "--sustain-seconds", "0", "--allow-stale-profile", *extra,
                      config="c2", steps="2", batch="197")
        assert d["roofline"]["traffic"] is None and "issue" not in d["roofline"], extra
        assert d["clv_kernel"]["traffic"] is None and "counter_frac" not in d["clv_kernel"], extra


def test_pipelined_leg_is_reported_beside_the_blocking_headline():
    d = run_bench("--no-cpu-baseline", "--sustain-seconds", "0.3")
    pl = d["roofline"]["pipelined"]
    assert pl["batches_in_flight"] == 2 and pl["steps"] >= 2 and pl["evals_per_s"] > 0
    assert d["roofline"]["sustained"]["evals_per_s"] > 0


def test_site_sharded_bench_matches_candidate_sharded_checksum():
    a = run_bench("--no-cpu-baseline", "--shard", "sites")
    b = run_bench("--no-cpu-baseline")
    assert a["scaling"] == "strong" and a["config"]["sharding"].startswith("site blocks")
    # same jobs, same parameters: the device-pointer path returns the same lnLs
    assert abs(a["lnl_check"] - b["lnl_check"]) <= 1e-9 * abs(b["lnl_check"])


def test_protein_config_runs_through_the_20_state_fused_evaluator():
    """c3 (200 taxa x 10 000 sites x 20 states): the batch goes through
    kernels_fused_k20.hip and agrees with the CPU oracle on the sampled jobs."""
    d = run_bench("--cpu-seconds", "3", config="c3", steps="1", batch="6")
    assert d["config"]["path"] == "fused batch" and d["roofline"]["kernel"] == "fused20_eval_kernel"
    assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] <= 1.0
    assert d["value"] > 0
    assert d["cpu_baseline"]["parity_max_rel_err"] < 1e-9


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher: the parent spawns two ranks
    (both on device 0 of this one-GPU box, gloo collectives) and relays rank 0's
    line, which must say n_gpus 2 and carry the two ranks' aggregate rate."""
    d = run_bench("--no-cpu-baseline", "--gpus", "2", "--dist-backend", "gloo", "--device", "0")
    assert d["n_gpus"] == 2 and d["value"] > 0
    s = run_bench("--no-cpu-baseline", "--gpus", "2", "--dist-backend", "gloo", "--device", "0",
                  "--shard", "sites")
    one = run_bench("--no-cpu-baseline", "--shard", "sites")
    assert s["n_gpus"] == 2
    assert abs(s["lnl_check"] - one["lnl_check"]) <= 1e-9 * abs(one["lnl_check"])


def test_site_sharded_bench_on_the_20_state_evaluator():
    """--shard sites is not a 4-state privilege: two ranks (one device, gloo) split c3's
    sites and reproduce the one-rank checksum through fused20_eval_kernel."""
    a = run_bench("--no-cpu-baseline", "--gpus", "2", "--dist-backend", "gloo", "--device", "0",
                  "--shard", "sites", config="c3", steps="1", batch="4")
    b = run_bench("--no-cpu-baseline", "--shard", "sites", config="c3", steps="1", batch="4")
    assert a["n_gpus"] == 2 and a["scaling"] == "strong" and a["roofline"]["kernel"] == "fused20_eval_kernel"
    assert abs(a["lnl_check"] - b["lnl_check"]) <= 1e-9 * abs(b["lnl_check"])


def test_rccl_calls_of_the_multi_gpu_path_with_one_rank():
    """The N > 1 bench path on RCCL itself -- process group on `nccl` with a device id,
    barrier, all-reduce of the per-block lnLs on a device tensor, MAX of the times -- as a
    ONE-rank group (this box has one GPU; RCCL refuses two ranks on a device): the
    all-reduced checksum must equal the plain run's."""
    s = __import__("socket").socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RDAMD_BENCH_PG="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    a = run_bench("--no-cpu-baseline", "--shard", "sites", env=env)
    b = run_bench("--no-cpu-baseline", "--shard", "sites")
    assert a["n_gpus"] == 1 and a["lnl_check"] == b["lnl_check"]
    # ... and with the LIBRARY's communicator beside torch's in one process: its id travels over the
    # process group (broadcast_object_list), its all-reduce of the lnLs + the second-pass flag is
    # queued on the partition's stream behind every stream-ordered batch (what the site_sharded /
    # grid legs of the N > 1 line do on real links)
    a2 = run_bench("--no-cpu-baseline", "--shard", "sites", "--one-rank-comm", env=env)
    assert a2["lnl_check"] == b["lnl_check"]
    # the `site_sharded` leg of the N > 1 line as a one-rank site group: the timed loop under BOTH sum
    # modes of the library's communicator (ncclAllGather + rank-order sum, the default; ncclAllReduce)
    # and the bare collective's latency in each -- what the first hardware SCALE run prints
    leg = run_bench("--no-cpu-baseline", "--one-rank-shard-legs", env=env)["site_sharded"]
    assert "error" not in leg, leg
    assert leg["lnl_check"] == b["lnl_check"]
    sm = leg["sum_modes"]
    assert sm["allreduce"]["lnl_check_equals_gather"] is True
    for mode in ("gather", "allreduce"):
        assert sm[mode]["value"] > 0 and 0 < sm[mode]["collective_us_back_to_back"] < 5000
    c = run_bench("--no-cpu-baseline", env=env)          # candidate sharding: barrier + MAX only
    assert c["value"] > 0
    assert c["rccl_ranks"] == 1 and a["rccl_ranks"] == 1   # a real all-reduce over the communicator
