#!/usr/bin/env python3
"""bench.py -- candidate-root lnL evaluations/sec on BASELINE config c2.

  python bench.py --gpus N --steps K --warmup W

One "step" = one batch of `--batch` full-traversal candidate-root evaluations
(2n-2 P-matrices + n-1 CLV operations + root reduction each;
model_t::compute_lh_partition, /root/reference/src/model.cpp:454-476) with a
different root placement AND a different substitution-parameter set per
evaluation (what the exhaustive search does, src/model.cpp:1154-1229), run as
ONE fused launch (rdamd_evaluate_batch).  Parameters change every step, so
every P-matrix and every CLV is recomputed inside the timed region.  Inputs
(tip codes, pattern weights, compiled schedules) are resident in HBM before
the timed region.  A short second leg times the materialising per-operation
CLV kernel (rdamd_update_clvs) for its HBM roofline.

Multi-GPU (one process per GPU): candidate roots are sharded across ranks the
way the reference shards them across MPI ranks (src/model.cpp:1867-1911) --
independent work, no data-path collective, weak scaling (per-GPU batch fixed);
`--shard sites|grid` splits site blocks instead and all-reduces the per-block
lnLs (RCCL).  `value` is the whole-job rate.  Without a launcher
(`python bench.py --gpus N`, WORLD_SIZE unset) this process starts the N ranks
itself, before anything touches the GPU, and relays rank 0's line.

Rank 0 prints ONE JSON line with `roofline` -- the dominant kernel's binding
resource: FP64 flops against the 78.6 TFLOP/s peak for the fused evaluators,
HIP-event timed on the partition's own stream -- `clv_kernel` (the materialising
kernel against the HBM peak) and `cpu_baseline` (the CPU oracle's AVX2 loop timed
on a bounded sample of the same workload on this box's host cores).
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    "c1": dict(n=10, S=1000, K=4, R=1),
    "c2": dict(n=100, S=50000, K=4, R=4),
    "c3": dict(n=200, S=10000, K=20, R=4),
    "c4": dict(n=500, S=500000, K=4, R=4),
    "c5": dict(n=1000, S=100000, K=4, R=4),
    # not a BASELINE config: the reference's largest DNA fixture (test/data/dna/125.phy +
    # tree/125.tree; 29 149 columns = 19 436 patterns, a third of the characters gaps), for
    # what subtree site repeats do on real, repeat-rich data.  n, S are filled in at load time.
    "d125": dict(n=125, S=19436, K=4, R=4),
}
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
FP64_PEAK_TFLOPS = 78.6     # MI355X FP64 vector = matrix peak


def full_eval_flops(n, S, R, K):
    """2K(2K-1)+K flops per (site, rate) per operation."""
    return (2 * K * (2 * K - 1) + K) * S * R * (n - 1)


def clv_kernel_bytes(n, S, R, K):
    """ALGORITHMIC bytes one full evaluation moves through the CLV kernel
    (DESIGN.md section 4): every computed CLV written once, every computed CLV
    but the root read once, 1-byte tip codes, u32 scalers written/read alike."""
    W = S * R * K * 8
    return (2 * n - 3) * W + n * S + (2 * n - 3) * 4 * S


def full_eval_bytes(n, S, R, K):
    """bytes_full = (2n-2) W + n S + (2n-2) 4 S."""
    W = S * R * K * 8
    return (2 * n - 2) * W + n * S + (2 * n - 2) * 4 * S


class StaleProfile(Exception):
    pass


def profiled_summary(kernel, batch, config, key):
    """The committed counters of `kernel` (profiles/collect.sh + profiles/summarize.py:
    separate rocprofv3 passes of the default command -- c2, batch 197 -- and, for the 20-state
    evaluator, of `--config c3`; other shapes get None).  They describe the kernel as it was when they were taken: the summary carries a
    digest of the kernel's sources (profiles/sources.py), and counters of a kernel that has
    changed since are REFUSED, not published."""
    if config not in ("c2", "c3") or batch != 197:
        return None, None
    import glob
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    from sources import source_digest
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    # (template variants share the prefix: the one that did the work is the slowest)
    # (an entry belongs to ONE bench command: `command`, c2 where the summary does not say)
    hits = [v for k, v in d.items() if k.startswith(kernel) and key in v and v.get("command", "c2") == config]
    if not hits:
        return None, None
    v = max(hits, key=lambda v: v.get("avg_us", 0.0))
    name = os.path.relpath(files[-1], ROOT)
    if v.get("source_digest") != source_digest(kernel):
        raise StaleProfile("%s holds counters of %s taken from sources %s, the sources are now %s: "
                           "re-run profiles/collect.sh + profiles/summarize.py (or pass "
                           "--allow-stale-profile to publish the line without counters)"
                           % (name, kernel, v.get("source_digest"), source_digest(kernel)))
    return v, name


def predicted_for(leg, n_gpus, batch, config, sites_overridden):
    """What this leg of the N-GPU line was PREDICTED to show (profiles/scale_prediction.json: figures
    measured on one GPU + an assumed all-reduce latency; c2, batch 197 only), so that a hardware
    SCALE record carries its own yardstick."""
    if config != "c2" or batch != 197 or sites_overridden or n_gpus < 2:
        return None
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "scale_prediction.json")))
    except (OSError, ValueError):
        return None
    v = d.get(leg, {}).get(str(n_gpus))
    if v is None:
        return None
    return {"value": v, "unit": "evals/s", "assumed_collective_us": d["assumed_collective_us"].get(str(n_gpus)),
            "basis": d["basis"]}


def profiled_traffic(kernel, batch, config):
    """HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes, corrected per
    MI355X_MICROARCH.md."""
    v, name = profiled_summary(kernel, batch, config, "hbm_bytes_per_launch")
    return (int(v["hbm_bytes_per_launch"]), name) if v else (None, None)


def check_one_hip_runtime(rd):
    """Before a torch tensor's device pointer goes to librdamd (rdamd_evaluate_batch_device):
    both must sit on the SAME libamdhip64.  A process can hold two -- PyTorch bundles its own
    ROCm, librdamd links /opt/rocm's; which one librdamd binds to depends on the import order
    (this file imports torch first: one instance) -- and allocations of one are unknown to the
    other (csrc/comm.cpp looks for its RCCL beside its own runtime for the same reason)."""
    mapped = rd.mapped_hip_runtimes()
    if len(mapped) > 1 or (mapped and rd.hip_runtime_path() not in mapped):
        raise SystemExit("bench.py: two HIP runtime instances in this process (%s); librdamd runs on %s. "
                         "Device pointers must not cross between them: import torch before "
                         "root_digger_amd (or build librdamd against the same ROCm)"
                         % (", ".join(mapped), rd.hip_runtime_path()))


def spawn_ranks(n):
    """One child process per GPU (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in the
    environment, rendezvous on 127.0.0.1), started from a parent that has not
    initialised the GPU.  Rank 0's stdout is relayed; the exit code is the
    first non-zero child code."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    children = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        children.append(subprocess.Popen(
            [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
            stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    import threading
    lines = []
    reader = threading.Thread(target=lambda: lines.extend(children[0].stdout), daemon=True)
    reader.start()
    failed = 0
    while any(c.poll() is None for c in children):
        failed = next((c.returncode for c in children if c.poll()), 0)
        if failed:           # one rank died: the others would wait in a collective forever
            # ... unless they are about to fail the same way: a short grace period lets
            # them report their own error before they are stopped
            deadline = time.monotonic() + 5.0
            while time.monotonic() < deadline and any(c.poll() is None for c in children):
                time.sleep(0.05)
            for c in children:
                if c.poll() is None:
                    c.terminate()
            break
        time.sleep(0.2)
    codes = [c.wait() for c in children]
    reader.join(timeout=10)
    sys.stdout.write("".join(lines))
    sys.stdout.flush()
    return failed or next((c for c in codes if c), 0)


def profiled_issue(kernel, batch, config):
    """Issue-slot figures of `kernel` derived by profiles/summarize.py from the
    committed SQ counter passes of the default command (c2, batch 197): VALU
    instructions per (operation, rate) step, the share of them that are the
    algorithm's FP64 instructions, and VALU issue-slot utilisation."""
    v, name = profiled_summary(kernel, batch, config, "derived")
    return {"issue": dict(v["derived"], source=name)} if v else {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=197,
                    help="candidate-root evaluations per step per GPU (default: one sweep "
                         "over the 2n-3 = 197 candidate roots of c2)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0,
                    help="budget for the cpu_baseline leg (rank 0, N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--repeat-classes", type=int, default=None,
                    help="class limit of the pseudo-tips (rdamd_partition_set_site_repeats; "
                         "default: the library's)")
    ap.add_argument("--no-repeats", action="store_true",
                    help="4-state partitions: without RDAMD_ATTRIB_SITE_REPEATS (the reference sets "
                         "CORAX_ATTRIB_SITE_REPEATS for every 4-state run, src/model.cpp:145-149; A/B only)")
    ap.add_argument("--rescale-speculation", type=int, default=-1, choices=(-1, 0, 1),
                    help="rdamd_partition_set_rescale_speculation: -1 the library's default (on up to 256 "
                         "tips), 0 rescale tests on every step, 1 on (A/B only)")
    ap.add_argument("--allow-stale-profile", action="store_true",
                    help="publish the line without `traffic` / `issue` when the committed counter "
                         "summary was taken from other kernel sources (default: fail)")
    ap.add_argument("--sustain-seconds", type=float, default=2.0,
                    help="after the K timed steps: the same step looped for this long, reported as "
                         "`sustained` (0 = skip)")
    ap.add_argument("--sites", type=int, default=None,
                    help="override the config's site count (the multi-rank pre-flight runs c5's tree at "
                         "an oracle-sized alignment)")
    ap.add_argument("--as-candidate-group", default=None, metavar="I/N",
                    help="one rank only: take the candidates (and parameter draws) of candidate group I "
                         "of N, i.e. what rank 0 of that group of an N x G grid evaluates -- the "
                         "reference value of the grid run's lnl_check")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --device lets several ranks share one GPU (how the N>1 code "
                         "path is exercised on a one-GPU box); the driver's runs use nccl (RCCL)")
    ap.add_argument("--device", type=int, default=None, help="HIP device (default: LOCAL_RANK)")
    ap.add_argument("--leg-timeout", type=float, default=240.0,
                    help="N>1: seconds the extra legs (site_sharded / grid / rccl_ranks) may take before the "
                         "line is printed without them")
    ap.add_argument("--one-rank-shard-legs", action="store_true",
                    help="with RDAMD_BENCH_PG=1 on ONE rank: run the `site_sharded` leg anyway (a one-rank site group), so "
                         "that the code the N > 1 line runs on real links -- the library's communicator in both sum "
                         "modes, the bare collective's latency -- is exercised on a one-GPU box (tests/test_gpu_bench.py)")
    ap.add_argument("--no-shard-legs", action="store_true",
                    help="N>1 without --shard: skip the extra `site_sharded` / `grid` objects (the same "
                         "workload site-sharded with the all-reduce of the per-block lnLs)")
    ap.add_argument("--site-groups", type=int, default=2,
                    help="--shard grid: ranks per candidate group, i.e. site shards (BASELINE c5: "
                         "candidate groups x site shards = N)")
    ap.add_argument("--one-rank-comm", action="store_true",
                    help="--shard sites on ONE rank: still queue a (one-rank) ncclAllReduce behind every batch, "
                         "so that the step holds the collective's launch as a multi-GPU site group's does")
    ap.add_argument("--shard", default="candidates", choices=["candidates", "sites", "grid"],
                    help="N>1: split candidate roots (no collective, weak scaling; default) or "
                         "split site blocks and all-reduce the per-block lnLs (strong scaling; "
                         "BASELINE config c4 pattern)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process becomes the
        # launcher.  It starts N ranks as children BEFORE anything here touches
        # HIP or torch, waits, relays rank 0's JSON line and nothing else.
        raise SystemExit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))

    import numpy as np
    import torch
    import root_digger_amd as rd
    from root_digger_amd import synth, dist as rdist

    if not torch.cuda.is_available() or rd.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device; there is no CPU fallback")
    device = local_rank if args.device is None else args.device
    torch.cuda.set_device(device)
    rd.set_device(device)           # (before the rendezvous: see root_digger_amd/cli.py)
    host_collectives = args.dist_backend == "gloo"
    # RDAMD_BENCH_PG=1: form the process group even for one rank, so that the RCCL calls
    # of the N > 1 path (init, barrier, all-reduce on device tensors) can be exercised on
    # a one-GPU box (tests/test_gpu_bench.py)
    use_pg = world > 1 or os.environ.get("RDAMD_BENCH_PG") == "1"
    real_stdout = None
    if use_pg or args.one_rank_comm:
        # RCCL prints a version banner on the process's stdout when a communicator comes
        # up: keep file descriptor 1 for the ONE JSON line, send everything else to stderr
        sys.stdout.flush()
        real_stdout = os.dup(1)
        os.dup2(2, 1)
    if use_pg:
        import torch.distributed as tdist
        if host_collectives:
            tdist.init_process_group("gloo")
        else:
            tdist.init_process_group("nccl", device_id=torch.device("cuda", device))

    cfg = CONFIGS[args.config]
    n, S, K, R = cfg["n"], cfg["S"], cfg["K"], cfg["R"]
    if args.sites:
        S = args.sites
    seed = 0xD166E5 + sorted(CONFIGS).index(args.config)
    data_weights = None
    if args.config == "d125":
        import lzma
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import util
        text = lzma.open(os.path.join(util.DATA, "125.phy.xz"), "rt").read()
        toks = text.split()
        seqs = {toks[2 + 2 * i]: toks[3 + 2 * i] for i in range(int(toks[0]))}
        seqs, data_weights = util.compress(seqs)
        w = {"newick": open(os.path.join(util.DATA, "125.tree")).read(), "seqs": seqs,
             "rates": rd.compute_gamma_cats(1.0, R), "alphabet": "ACGT"}
        n, S = len(seqs), len(next(iter(seqs.values())))
    else:
        w = synth.workload(n, S, K, R, seed)
    # Three ways to use N GPUs: candidate roots only (default), site
    # blocks only, or the 2-D grid of BASELINE config c5: `cgroups` groups of
    # `sgroups` ranks; a group shares its candidates and splits the sites, and the
    # all-reduce of the per-block lnLs stays inside the group (adjacent ranks).
    grid = args.shard == "grid" and world > 1
    cgroups, sgroups = (rdist.grid_2d(world, args.site_groups) if grid else
                        (1, world) if args.shard == "sites" else (world, 1))
    cgroup, srank = rdist.rank_coords(rank, sgroups)
    site_sharded = args.shard in ("sites", "grid")   # (both fused evaluators leave device-side lnLs)
    site_group = None
    if grid:
        for c in range(cgroups):      # every rank creates every group, in the same order
            g = tdist.new_group(list(range(c * sgroups, (c + 1) * sgroups)))
            if c == cgroup:
                site_group = g
    S_total = S
    full_seqs = dict(w["seqs"])
    if site_sharded:   # this rank's contiguous block of patterns (dist.site_block)
        lo, hi = rdist.site_block(S, srank, sgroups)
        w["seqs"] = {k: v[lo:hi] for k, v in w["seqs"].items()}
        S = hi - lo
    tree = rd.Tree.from_newick(w["newick"])
    cmap = rd.MAP_NT
    if K != 4:
        import ctypes
        cmap = (ctypes.c_uint64 * 256)()
        for i, ch in enumerate(w["alphabet"]):
            cmap[ord(ch)] = 1 << i

    repeats = K == 4 and not args.no_repeats
    part = rd.Partition.for_tree(tree, K, S, R, attributes=rd.ATTRIB_SITE_REPEATS if repeats else 0)
    if repeats and args.repeat_classes is not None:
        part.set_site_repeats(args.repeat_classes)
    if K == 4 and args.rescale_speculation != -1:
        part.set_rescale_speculation(args.rescale_speculation)
    for label, seq in w["seqs"].items():
        part.set_tip_states(tree.tip_index(label), cmap, seq)
    if data_weights is not None:
        part.set_pattern_weights(data_weights)
    freqs = part.empirical_frequencies()
    if site_sharded and sgroups > 1:   # the model is global: combine the blocks' counts
        freqs = rdist.global_frequencies(freqs, S, group=site_group,
                                         device="cpu" if host_collectives else "cuda")
    part.set_frequencies(0, freqs)
    part.set_category_rates(w["rates"])

    # this rank's candidate roots (src/model.cpp:1899-1907) and per-candidate
    # parameter sets (random_params, src/model.cpp:87-93)
    if site_sharded:   # a site group sees all candidates of its candidate group, on its own sites
        mine = rdist.assign_candidates(tree.root_count(), cgroup, cgroups)
        rng = np.random.default_rng(seed + 1000 + cgroup)
    elif args.as_candidate_group and world == 1:
        gi, gn = (int(x) for x in args.as_candidate_group.split("/"))
        mine = rdist.assign_candidates(tree.root_count(), gi, gn)
        rng = np.random.default_rng(seed + 1000 + gi)
    else:
        mine = rdist.assign_candidates(tree.root_count(), rank, world)
        rng = np.random.default_rng(seed + 1000 + rank)
    params = [synth.random_params(K * K - K, rng) for _ in range(len(mine))]
    roots = [tree.root_location(i) for i in mine]

    use_fused = K == 4 or (K == 20 and R <= 8)   # the shapes the fused evaluators take
    fused_kernel = "fused_dna_eval_kernel" if K == 4 else "fused20_eval_kernel"
    clv_kernel = ("clv_dna_traversal_kernel" if K == 4 else
                  "clv_k20_traversal_kernel" if K == 20 else "clv_generic_traversal_kernel")
    nb = args.batch   # fixed per GPU whatever N is (weak scaling)
    params = np.array(params)
    freqs_b = np.tile(np.asarray(freqs), (len(mine), 1))

    def evaluate_unfused(j):
        j %= len(mine)
        part.set_subst_params(0, params[j])
        ops, pmi, brl = tree.generate_operations(roots[j])
        part.update_prob_matrices(pmi, brl)
        part.update_clvs(ops)
        return part.compute_root_loglikelihood(tree.root_clv_index(),
                                               tree.root_scaler_index())

    executed = {"steps": 0, "matvecs": 0, "clade_rows": 0, "evals": 0}
    if use_fused:
        scheds = [part.schedule(*tree.generate_operations(rl)) for rl in roots]
        depth = max(sc.stack_depth() for sc in scheds)
        sched_stats = [sc.stats() for sc in scheds]

    prepared = {}

    def step_jobs(s, params_, count):
        """which candidates step s evaluates and with which parameter draws"""
        idx = [(s * nb + b) % count for b in range(nb)]
        # every job of every step is a distinct (root, parameter set) pair
        jitter = 1.0 + 1e-3 * (((s % 97) + 1) + np.arange(nb)[:, None] / (4.0 * nb))
        return idx, np.ascontiguousarray(params_[idx] * jitter * (1.0 + 0.05 * np.sin(np.arange(K * K - K) + s)))

    def site_comm(group, members):
        """The LIBRARY's RCCL communicator of this rank's site group (rdamd_comm_*, what rd_amd
        uses): its all-reduce is queued on the partition's stream right behind the batch --
        torch.distributed's would run on torch's stream, i.e. after a host wait for the batch.
        The 128-byte id travels over the process group that is up already."""
        if host_collectives:
            return None
        if not use_pg:      # one rank, no process group: a one-rank communicator if asked for
            return rd.Comm(rd.Comm.unique_id(), 0, 1) if args.one_rank_comm else None
        ids = [rd.Comm.unique_id() if rank == members[0] else None]
        tdist.broadcast_object_list(ids, src=members[0], group=group)
        return rd.Comm(ids[0], members.index(rank), len(members))

    class Pipeline:
        """Site-sharded steps with TWO batches in flight (rdamd_evaluate_batch_submit_device,
        slots alternating): a batch's finishing kernel leaves the per-block lnLs -- and the
        second-pass flag behind them -- in device memory, the all-reduce is queued behind it on
        the partition's stream, and the next batch's front half runs beside this one's
        evaluator.  No host read between a batch and its collective."""

        def __init__(self, p, comm):
            self.p, self.comm, self.busy = p, comm, [False, False]
            self.stream = C.c_void_p(rd.lib.rdamd_partition_stream(p._h))

        def submit(self, s_, handles, sub, fr, dev):
            slot = s_ & 1
            if self.busy[slot]:
                self.p.evaluate_batch_finish_device(slot)
            n = self.p.evaluate_batch_submit_device(slot, handles, sub, fr, dev.data_ptr())
            if self.comm is not None:
                self.comm.allreduce_sum(C.c_void_p(dev.data_ptr()), n + 1, self.stream)
            self.busy[slot] = True

        def drain(self):
            for slot in (0, 1):
                if self.busy[slot]:
                    self.p.evaluate_batch_finish_device(slot)
                    self.busy[slot] = False

        @staticmethod
        def check(rows):
            """the second-pass flags of the timed batches (summed over the group): the synthetic
            parameter draws never raise one; if they did, the values behind it are not final"""
            if float(rows[:, -1].sum().item()) != 0.0:
                raise SystemExit("a timed batch asked for its second evaluator pass: "
                                 "rdamd_evaluate_batch_redo_device is not part of this loop")

    def prepare(s):
        """the arguments of step s (harness work: which candidates, their parameter draws, the
        handle array): made before the timed region for the steps it times, so that the region
        holds the library's work -- parameter upload, P-matrices, tables, traversal, reduction,
        copy back -- and not numpy's"""
        idx, sub = step_jobs(s, params, len(mine))
        return {"handles": rd.Partition.schedule_handles([scheds[i] for i in idx]), "sub": sub,
                "freqs": np.ascontiguousarray(freqs_b[idx]),
                "steps": sum(sched_stats[i]["steps"] for i in idx),
                "matvecs": sum(sched_stats[i]["matvecs"] for i in idx),
                "clade_rows": sum(sched_stats[i]["clade_rows"] for i in idx)}

    def step(s):
        """one batch: jobs rotate through this rank's candidates; parameters are
        re-drawn around their base values so no step repeats an earlier one."""
        if not use_fused:
            return sum(evaluate_unfused(s * nb + b) for b in range(nb))
        a = prepared.get(s) or prepare(s)
        executed["steps"] += a["steps"]
        executed["matvecs"] += a["matvecs"]
        executed["clade_rows"] += a["clade_rows"]
        executed["evals"] += nb
        handles, sub, fr = a["handles"], a["sub"], a["freqs"]
        if site_sharded:
            row = s - args.warmup
            lnl_dev = lnl_rows[row] if 0 <= row < args.steps else lnl_warm[s & 1]
            # (the tensor's memory comes from torch's HIP runtime, the kernels that write it
            # from librdamd's: check_one_hip_runtime() has made sure they are the same one)
            if pipeline is not None:             # two batches in flight, collective queued behind each
                pipeline.submit(s, handles, sub, fr, lnl_dev)
                return lnl_dev
            part.evaluate_batch_device(handles, sub, fr, lnl_dev.data_ptr())
            if use_pg:                           # gloo test path: through the host
                host = lnl_dev[:nb].cpu()
                rdist.allreduce_lnl(host, site_group)
                lnl_dev[:nb].copy_(host)
            return lnl_dev
        return float(part.evaluate_batch(handles, sub, fr).sum())

    def barrier():
        if pipeline is not None:
            pipeline.drain()
        if use_pg:
            tdist.barrier()
        torch.cuda.synchronize()
        part.sync()

    pipeline = None
    if site_sharded:
        check_one_hip_runtime(rd)
        if use_fused and not host_collectives:
            members = list(range(cgroup * sgroups, (cgroup + 1) * sgroups))
            pipeline = Pipeline(part, site_comm(site_group, members) if sgroups > 1 or args.one_rank_comm else None)
    # site-sharded: every timed step leaves its per-job lnLs in its own row (no
    # torch work inside the timed region; one more column: the batch's second-pass flag);
    # warm-up steps share two scratch rows
    lnl_rows = (torch.zeros((args.steps, nb + 1), dtype=torch.float64, device="cuda")
                if site_sharded else None)
    lnl_warm = torch.zeros((2, nb + 1), dtype=torch.float64, device="cuda") if site_sharded else None
    if use_fused:
        for s in range(args.warmup, args.warmup + args.steps):
            prepared[s] = prepare(s)
    for s in range(args.warmup):
        step(s)
    part.profile_enable(True)
    executed = dict.fromkeys(executed, 0)
    barrier()
    t0 = time.perf_counter()
    check = 0.0
    for s in range(args.steps):
        out = step(args.warmup + s)
        if not site_sharded:
            check += out
    barrier()
    elapsed = time.perf_counter() - t0
    if site_sharded:
        if pipeline is not None:
            Pipeline.check(lnl_rows)
        check = float(lnl_rows[:, :nb].sum().item())
    prof = part.profile_read()
    part.profile_enable(False)
    executed_timed = dict(executed)
    if not np.isfinite(check) and not os.environ.get("RDAMD_BENCH_TIMING_ONLY"):   # (ablation runs: profiles/*_ab.sh)
        raise SystemExit("non-finite lnL in the timed region")

    # The K timed steps last tens of milliseconds: too short for the clocks to settle or for a
    # sampling monitor to see the device busy.  `sustained`: the very same step, looped for
    # --sustain-seconds, with its own rate and its own event-timed kernel figure.
    sustained = None
    if use_fused and not site_sharded and args.sustain_seconds > 0:
        part.profile_enable(True)
        executed = dict.fromkeys(executed, 0)
        barrier()
        t1 = time.perf_counter()
        k = 0
        while time.perf_counter() - t1 < args.sustain_seconds:
            step(args.warmup + args.steps + k)
            k += 1
        barrier()
        dt = time.perf_counter() - t1
        sprof = part.profile_read()
        part.profile_enable(False)
        sustained = {"seconds": round(dt, 3), "steps": k, "evals_per_s": round(k * nb / dt, 2),
                     "kernel_ms": sprof["fused"][0], "launches": sprof["fused"][1],
                     "executed": dict(executed)}

    # `pipelined`: the same steps with TWO batches in flight (rdamd_evaluate_batch_submit / _wait,
    # slots 0 / 1 -- what the lock-stepped search does): the upload, P-matrices and clade
    # tables of step s + 1 and the host's work run beside the evaluator of step s.  Reported beside
    # the headline, which stays the blocking call (one batch at a time, comparable across rounds).
    pipelined = None
    if use_fused and not site_sharded and args.sustain_seconds > 0 and world == 1:
        base = args.warmup + args.steps
        inputs = [prepare(base + 1000 + i) for i in range(8)]
        lnl_sum = 0.0
        barrier()
        t1 = time.perf_counter()
        k = 0
        pending = None
        while True:
            a = inputs[k % len(inputs)]
            part.evaluate_batch_submit(k & 1, a["handles"], a["sub"], a["freqs"])
            if pending is not None:
                lnl_sum += float(part.evaluate_batch_wait(pending, nb).sum())
            pending = k & 1
            k += 1
            if time.perf_counter() - t1 >= args.sustain_seconds:
                break
        lnl_sum += float(part.evaluate_batch_wait(pending, nb).sum())
        barrier()
        dt = time.perf_counter() - t1
        if not np.isfinite(lnl_sum) and not os.environ.get("RDAMD_BENCH_TIMING_ONLY"):
            raise SystemExit("non-finite lnL in the pipelined leg")
        pipelined = {"seconds": round(dt, 3), "steps": k, "evals_per_s": round(k * nb / dt, 2),
                     "ms_per_step": round(dt / k * 1e3, 4), "batches_in_flight": 2,
                     "note": "rdamd_evaluate_batch_submit / _wait, slots alternating: step s + 1 is queued "
                             "before step s is waited for"}

    if use_pg:
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device="cpu" if host_collectives else "cuda")
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        elapsed = float(t.item())

    evals_per_rank = args.steps * nb
    total_evals = evals_per_rank * cgroups      # one set of evaluations per candidate group
    value = total_evals / elapsed

    def sharded_leg(mode):
        """The SAME workload with the sites split over the ranks -- north_star's "site blocks
        shard across the GPUs with an RCCL all-reduce of per-block log-likelihoods" -- as an extra
        object of the default N > 1 line, so that the driver's scaling run measures it without
        asking: mode "sites" = every rank a site block of all candidates (BASELINE c4's layout,
        strong scaling), "grid" = candidate groups x site shards (c5's).  Every batch leaves its
        per-block lnLs on the device (rdamd_evaluate_batch_device), one all-reduce inside the site
        group sums them.  lnl_check = the sum of the all-reduced lnLs rank 0 saw: for "sites"
        the N = 1 line's value, for "grid" that of `--as-candidate-group 0/<groups>`."""
        check_one_hip_runtime(rd)   # (torch tensors' device pointers go to librdamd below)
        cg, sg = (1, world) if mode == "sites" else rdist.grid_2d(world, args.site_groups)
        cgi, sr = rdist.rank_coords(rank, sg)
        group = None
        if mode == "grid":
            for c in range(cg):      # every rank creates every group, in the same order
                g = tdist.new_group(list(range(c * sg, (c + 1) * sg)))
                if c == cgi:
                    group = g
        lo, hi = rdist.site_block(S_total, sr, sg)
        p2 = rd.Partition.for_tree(tree, K, hi - lo, R, attributes=rd.ATTRIB_SITE_REPEATS if repeats else 0)
        if repeats and args.repeat_classes is not None:
            p2.set_site_repeats(args.repeat_classes)
        for label, seq in full_seqs.items():
            p2.set_tip_states(tree.tip_index(label), cmap, seq[lo:hi])
        if data_weights is not None:
            p2.set_pattern_weights(np.ascontiguousarray(data_weights[lo:hi]))
        fr = rdist.global_frequencies(p2.empirical_frequencies(), hi - lo, group=group,
                                      device="cpu" if host_collectives else "cuda")
        p2.set_frequencies(0, fr)
        p2.set_category_rates(w["rates"])
        mine2 = rdist.assign_candidates(tree.root_count(), cgi, cg)
        rng2 = np.random.default_rng(seed + 1000 + cgi)
        params2 = np.array([synth.random_params(K * K - K, rng2) for _ in range(len(mine2))])
        scheds2 = [p2.schedule(*tree.generate_operations(tree.root_location(i))) for i in mine2]
        frb = np.tile(np.asarray(fr), (len(mine2), 1))
        rows = torch.zeros((args.steps, nb + 1), dtype=torch.float64, device="cuda")
        warm = torch.zeros((2, nb + 1), dtype=torch.float64, device="cuda")
        members2 = list(range(cgi * sg, (cgi + 1) * sg))
        pipe2 = Pipeline(p2, site_comm(group, members2)) if not host_collectives else None
        jobs = {}
        for s_ in range(args.warmup + args.steps):
            idx, sub = step_jobs(s_, params2, len(mine2))
            jobs[s_] = (rd.Partition.schedule_handles([scheds2[i] for i in idx]), sub,
                        np.ascontiguousarray(frb[idx]))

        def one(s_):
            row = s_ - args.warmup
            dev = rows[row] if row >= 0 else warm[s_ & 1]
            if pipe2 is not None:
                pipe2.submit(s_, jobs[s_][0], jobs[s_][1], jobs[s_][2], dev)
                return
            p2.evaluate_batch_device(jobs[s_][0], jobs[s_][1], jobs[s_][2], dev.data_ptr())
            host = dev[:nb].cpu()                # gloo test path: through the host
            rdist.allreduce_lnl(host, group)
            dev[:nb].copy_(host)

        def sync():
            if pipe2 is not None:
                pipe2.drain()
            tdist.barrier()
            torch.cuda.synchronize()
            p2.sync()

        for s_ in range(args.warmup):
            one(s_)
        sync()
        t_ = time.perf_counter()
        for s_ in range(args.warmup, args.warmup + args.steps):
            one(s_)
        sync()
        dt = time.perf_counter() - t_
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if host_collectives else "cuda")
        tdist.all_reduce(tt, op=tdist.ReduceOp.MAX)
        dt = float(tt.item())
        lnl_check_default = float(rows[:, :nb].sum().item())
        if pipe2 is not None:
            Pipeline.check(rows)

        def timed_loop():
            sync()
            t0_ = time.perf_counter()
            for s2_ in range(args.warmup, args.warmup + args.steps):
                one(s2_)
            sync()
            t2_ = torch.tensor([time.perf_counter() - t0_], dtype=torch.float64, device="cpu" if host_collectives else "cuda")
            tdist.all_reduce(t2_, op=tdist.ReduceOp.MAX)
            return float(t2_.item())

        # The site group's sum, BOTH ways (csrc/comm.cpp): the default is ncclAllGather + a sum in rank
        # order (identical bits on every rank by construction), the alternative one ncclAllReduce.  The
        # timed loop above ran the default; here the same loop with the all-reduce, and the bare
        # collective of nb + 1 doubles back to back in each mode -- the latency the rounds of a
        # lock-stepped search pay once each.
        modes = {}
        if pipe2 is not None and pipe2.comm is not None:
            first = lnl_check_default
            lat = {}
            probe = torch.zeros(nb + 1, dtype=torch.float64, device="cuda")
            for name, mode_id in (("gather", rd.COMM_SUM_GATHER), ("allreduce", rd.COMM_SUM_ALLREDUCE)):
                pipe2.comm.set_sum_mode(mode_id)
                for rep in range(2):
                    sync()
                    t0_ = time.perf_counter()
                    for _ in range(200 if rep else 20):
                        pipe2.comm.allreduce_sum(C.c_void_p(probe.data_ptr()), nb + 1, pipe2.stream)
                    p2.sync()
                    lat[name] = round((time.perf_counter() - t0_) / 200 * 1e6, 2)
            pipe2.comm.set_sum_mode(rd.COMM_SUM_ALLREDUCE)
            dt_ar = timed_loop()
            modes = {"sum_modes": {
                "gather": {"value": round(args.steps * nb * cg / dt, 2), "ms_per_step": round(dt / args.steps * 1e3, 4),
                           "collective_us_back_to_back": lat["gather"]},
                "allreduce": {"value": round(args.steps * nb * cg / dt_ar, 2), "ms_per_step": round(dt_ar / args.steps * 1e3, 4),
                              "collective_us_back_to_back": lat["allreduce"],
                              "lnl_check_equals_gather": float(rows[:, :nb].sum().item()) == first},
                "note": "gather = ncclAllGather + rank-order sum kernel (the default: same bits on every rank by "
                        "construction); allreduce = one ncclAllReduce"}}
            pipe2.comm.set_sum_mode(rd.COMM_SUM_GATHER)
        out = {"value": round(args.steps * nb * cg / dt, 2), "unit": "evals/s",
               "ms_per_step": round(dt / args.steps * 1e3, 4), **modes,
               "scaling": "strong" if mode == "sites" else "weak",
               "sharding": ("site blocks of all candidates + all-reduce of the per-block lnLs" if mode == "sites"
                            else "%d candidate groups x %d site shards, all-reduce inside a group" % (cg, sg)),
               "sites_per_rank": hi - lo, "batch_per_step": nb,
               "collective": ("gloo (host), blocking batches" if host_collectives else
                              "the library's RCCL communicator: ncclAllGather + rank-order sum of %d values queued on the "
                              "partition's stream behind each batch, two batches in flight" % (nb + 1)),
               "lnl_check": lnl_check_default}
        pred = predicted_for("site_sharded" if mode == "sites" else "grid", world, nb, args.config, args.sites is not None)
        if pred:
            out["predicted"] = pred
        if pipe2 is not None and pipe2.comm is not None:
            pipe2.comm.destroy()
        p2.destroy()
        return out

    def clv_roofline(ms, launches, evals):
        bytes_clv = clv_kernel_bytes(n, S, R, K) * evals
        achieved = bytes_clv / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        # `achieved` counts SURVEY 8d's algorithmic bytes; the kernel forwards most CLV reads
        # through registers / LDS (measured traffic ~0.56 x algorithmic), so on shapes that
        # fill the chip the algorithmic rate can pass the nominal HBM peak -- `frac` then
        # switches its basis (below) instead of being capped at 1
        ratio = achieved / HBM_PEAK_GBS
        out = {"kernel": clv_kernel, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
               "unit": "GB/s", "frac": round(ratio, 4), "frac_basis": "algorithmic bytes", "traffic": None,
               "bytes_per_launch": round(bytes_clv / max(launches, 1)),
               "avg_launch_ms": round(ms / max(launches, 1), 5), "launches": launches}
        # The bytes that MUST cross HBM (every computed CLV and scaler written once, the tip codes read
        # once; the reads the algorithm counts are forwarded on chip) -- stated for every shape, and
        # the basis of `frac` where the algorithmic rate passes the peak and the command has no counters
        W = S * R * K * 8
        modelled = ((n - 1) * (W + 4 * S) + n * S) * evals
        mg = modelled / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        out.update(modelled_gbs=round(mg, 1), modelled_frac=round(mg / HBM_PEAK_GBS, 4))
        if ratio > 1.0:
            # More algorithmic bytes per second than the HBM can move: a fraction of the HBM peak must
            # then be made of bytes that cross HBM -- the command's counters where it has them
            # (`hbm counters`, attached below), this model of the minimum otherwise.
            out.update(frac=round(mg / HBM_PEAK_GBS, 4),
                       frac_basis="modelled HBM bytes: every CLV and scaler written once, tip codes read once "
                                  "(the reads the algorithm counts are forwarded on chip)",
                       algorithmic_equiv={"rate_gbs": round(achieved, 1), "ratio_to_peak": round(ratio, 4),
                                          "note": "SURVEY 8d's algorithmic bytes (every CLV written AND read once) / "
                                                  "time: exceeds the HBM peak because the reads stay on chip"})
        return out

    extra = {}
    if use_fused:
        ms, launches = prof["fused"]
        # Flops (2K(2K-1)+K per (site, rate) and operation), counted two ways:
        #  * `achieved` / `frac`: over the operations the evaluator actually RUNS -- with subtree
        #    site repeats a schedule runs fewer than n-1 per site (clades folded into per-job
        #    tables are evaluated once per pattern class; rdamd_schedule_stats), and work that
        #    was skipped must not count as throughput: frac <= 1 stays a statement about the
        #    kernel;
        #  * `algorithmic_equiv`: SURVEY's formula on all n-1 operations per evaluation, i.e.
        #    what a traversal without repeats would have had to execute for the same answers
        #    (the figure earlier rounds published; it may exceed the peak).
        per_op = (2 * K * (2 * K - 1) + K) * S * R
        flops = per_op * executed_timed["steps"]
        flops_equiv = full_eval_flops(n, S, R, K) * evals_per_rank
        tf = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        tf_equiv = flops_equiv / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        # matrix-vector products and element products the kernel really issues (tip and
        # pseudo-tip children are table look-ups): 2K^2-K per product + K per operation
        fp64_issued = ((2 * K * K - K) * executed_timed["matvecs"] + K * executed_timed["steps"]) * S * R
        hbm_equiv = full_eval_bytes(n, S, R, K) * evals_per_rank / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        # The fused traversal never materialises a CLV: its HBM traffic is tip codes
        # + per-job tables (PMC: `traffic`), far below SURVEY 8d's bytes_full, so the
        # resource that binds it is FP64 issue -- the vector FMA pipe for 4 states,
        # the FP64 matrix core (v_mfma_f64_4x4x4_4b_f64) for 20.  `roofline` is that
        # bound against the 78.6 TFLOP/s FP64 peak (vector = matrix on gfx950), the launch
        # time taken with HIP events on the partition's stream.
        # `hbm_equiv` keeps the algorithmic-byte rate for comparison with the
        # materialising path; it is a ratio to the HBM peak, not a roofline.
        ev = max(executed_timed["evals"], 1)
        roofline = {
            "kernel": fused_kernel, "bound": "fp64" if K == 4 else "mfma",
            "achieved": round(tf, 2), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tf / FP64_PEAK_TFLOPS, 4), "traffic": None,
            "executed_flops_per_launch": flops / max(launches, 1),
            "algorithmic_equiv": {"flops_per_launch": flops_equiv / max(launches, 1),
                                  "rate_tflops": round(tf_equiv, 2),
                                  "ratio_to_peak": round(tf_equiv / FP64_PEAK_TFLOPS, 4),
                                  "note": "SURVEY 8d flops of all n-1 operations per evaluation / launch "
                                          "time: what a traversal without site repeats would execute"},
            "schedule": {"operations_per_evaluation": n - 1,
                         "steps_per_evaluation": round(executed_timed["steps"] / ev, 2),
                         "matvecs_per_evaluation": round(executed_timed["matvecs"] / ev, 2),
                         "clade_table_rows_per_evaluation_and_rate": round(executed_timed["clade_rows"] / ev, 1),
                         "site_repeats": bool(repeats),
                         "max_classes": (part.site_repeats() if repeats else 0),
                         # parks per evaluation (mean over the schedules): all / in the register slot /
                         # in the one LDS slot; the rest wait on the in-memory stack
                         "parks_per_evaluation": [round(float(np.mean([st[k] for st in sched_stats])), 1)
                                                  for k in ("parks", "parks_in_registers", "parks_in_lds_slot")]},
            "issued_fp64_tflops": round(fp64_issued / (ms * 1e-3) / 1e12, 2) if ms > 0 else 0.0,
            "avg_launch_ms": round(ms / max(launches, 1), 4), "launches": launches,
            "share_of_step": round(ms * 1e-3 / elapsed, 3),
            ("measured_dfma_ceiling_tflops" if K == 4 else
             "measured_mfma_4x4x4_ceiling_tflops"): 68.4 if K == 4 else 73.2,
            "stack_depth": depth,
            "pmatrix_ms_per_launch": round(prof["fused_pmatrix"][0] / max(prof["fused_pmatrix"][1], 1), 4),
            "hbm_equiv": {"algorithmic_bytes_per_launch": full_eval_bytes(n, S, R, K) * nb,
                          "rate_gbs": round(hbm_equiv, 1),
                          "ratio_to_hbm_peak": round(hbm_equiv / HBM_PEAK_GBS, 3),
                          "note": "SURVEY 8d bytes_full per evaluation / launch time; CLVs stay in "
                                  "registers/LDS, so this exceeds the HBM peak by design"},
        }
        if pipelined:
            roofline["pipelined"] = pipelined
        if sustained:
            sms, sl = sustained.pop("kernel_ms"), sustained["launches"]
            sx = sustained.pop("executed")
            stf = per_op * sx["steps"] / (sms * 1e-3) / 1e12 if sms > 0 else 0.0
            sustained.update(kernel_avg_launch_ms=round(sms / max(sl, 1), 4),
                             achieved_tflops=round(stf, 2), frac=round(stf / FP64_PEAK_TFLOPS, 4))
            roofline["sustained"] = sustained
        # the committed counters belong to ONE command (profiles/collect.sh: c2, batch 197, the whole
        # alignment on one rank, the library's class limit): any option that changes the shape of the
        # launches -- --sites, --shard, --as-candidate-group, --repeat-classes, --no-repeats -- gets none
        default_cmd = (args.config in ("c2", "c3") and nb == 197 and (repeats or K != 4) and world == 1 and
                       args.sites is None and args.shard == "candidates" and args.as_candidate_group is None and
                       args.repeat_classes is None and not args.one_rank_comm and args.rescale_speculation == -1)
        try:
            if default_cmd:
                roofline.update(profiled_issue(fused_kernel, nb, args.config))
                roofline["traffic"], roofline["traffic_source"] = profiled_traffic(
                    fused_kernel, nb, args.config)
                der = roofline.get("issue", {})
                if K == 20 and "mfma_busy" in der:
                    # north star: "MFMA-busy reported against gfx950 peak" -- the share of SIMD cycles the
                    # FP64 matrix pipe was busy (SQ_VALU_MFMA_BUSY_CYCLES), and the CU's address unit
                    roofline["mfma_busy"] = der["mfma_busy"]
                    roofline["address_unit_busy"] = der.get("address_unit_busy")
        except StaleProfile as e:
            if not args.allow_stale_profile:
                raise SystemExit("bench.py: " + str(e))
            roofline["traffic"], roofline["traffic_stale"] = None, str(e)
        # second leg: the materialising CLV kernel (drop-in rdamd_update_clvs
        # path), HBM-bound.  The traversals are queued back to back without a
        # host sync in between, so the event-timed spans carry no launch gaps.
        scheds_unfused = [tree.generate_operations(roots[j % len(mine)]) for j in range(6)]
        for rep in range(2):
            if rep == 1:
                part.profile_enable(True)
            for j, (ops_j, pmi_j, brl_j) in enumerate(scheds_unfused):
                part.set_subst_params(0, params[j % len(mine)])
                part.update_prob_matrices(pmi_j, brl_j)
                part.update_clvs(ops_j)
        p2 = part.profile_read()
        part.profile_enable(False)
        extra["clv_kernel"] = clv_roofline(p2["clv"][0], p2["clv"][1], 6)
        # (a traversal may be several launches of the kernel -- independent subtrees side by side,
        # level by level: `avg_launch_ms` and `traffic` are per TRAVERSAL, a profiler's per-kernel
        # average times this count)
        lpt = part.update_clvs_launches()
        extra["clv_kernel"]["kernel_launches_per_traversal"] = lpt
        try:
            if default_cmd:
                extra["clv_kernel"]["traffic"], extra["clv_kernel"]["traffic_source"] = profiled_traffic(
                    clv_kernel, nb, args.config)
                if extra["clv_kernel"]["traffic"]:
                    extra["clv_kernel"]["traffic"] *= lpt
                if extra["clv_kernel"]["traffic"]:   # the bytes that really crossed HBM, per second
                    gbs = extra["clv_kernel"]["traffic"] / (extra["clv_kernel"]["avg_launch_ms"] * 1e-3) / 1e9
                    extra["clv_kernel"]["counter_gbs"] = round(gbs, 1)
                    extra["clv_kernel"]["counter_frac"] = round(gbs / HBM_PEAK_GBS, 4)
                    if "algorithmic_equiv" in extra["clv_kernel"]:   # (see clv_roofline)
                        extra["clv_kernel"]["frac"] = extra["clv_kernel"]["counter_frac"]
                        extra["clv_kernel"]["frac_basis"] = "hbm counters"
        except StaleProfile as e:
            if not args.allow_stale_profile:
                raise SystemExit("bench.py: " + str(e))
            extra["clv_kernel"]["traffic_stale"] = str(e)
    else:
        roofline = clv_roofline(prof["clv"][0], prof["clv"][1], evals_per_rank)

    result = {
        "metric": "candidate-root lnL evals/sec", "value": round(value, 2),
        "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "strong" if args.shard == "sites" else "weak",
        "vs_baseline": None,
        "dtype": "f64", "data": ("synthetic" if args.config != "d125" else
                                 "reference fixture 125.phy / 125.tree (pattern-compressed)"),
        "config": {"workload": "%s: %d-taxon %d-site %d-state UNREST+G%d full-traversal "
                               "root lnL" % (args.config, n, S, K, R),
                   "batch_per_gpu": nb,
                   "sharding": ("%d candidate groups x %d site shards, all-reduce inside a group"
                                % (cgroups, sgroups) if grid else
                                "site blocks + RCCL all-reduce" if site_sharded else "candidate roots"),
                   "path": "fused batch" if use_fused else "per-operation"},
        # BASELINE's second metric, two ways: over all n-1 operations of every evaluation (the
        # ALGORITHMIC-EQUIVALENT figure: what a traversal without site repeats would update) and
        # over the operations the schedules really run per site (clades folded into per-class
        # tables are not site-CLV updates)
        "site_clv_updates_per_sec": round(value * (n - 1) * S_total, 1),
        "site_clv_updates_per_sec_note": "algorithmic equivalent: (n-1) operations x sites per evaluation",
        "site_clv_updates_per_sec_executed": round(
            value * (executed_timed["steps"] / max(executed_timed["evals"], 1) if use_fused else n - 1) * S_total, 1),
        # sum of the lnLs rank 0 saw in the timed region: identical across world
        # sizes when site-sharded (each job's lnL is the all-reduced total)
        "lnl_check": check,
        "roofline": roofline,
    }
    result.update(extra)
    if args.shard == "candidates":
        pred = predicted_for("value", world, nb, args.config, args.sites is not None)
        if pred:
            result["predicted"] = pred

    def gpu_eval(j):
        if not use_fused:
            return evaluate_unfused(j)
        return float(part.evaluate_batch([scheds[j]], params[j:j + 1], freqs_b[j:j + 1])[0])

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(w, tree, cmap, freqs, n, S, K, R,
                                              params, roots, args.cpu_seconds,
                                              gpu_eval, data_weights)
        ideal = result["cpu_baseline"].get("one_socket_ideal")
        if ideal and ideal["value"] > 0:
            # north star: ">= 50 x the reference single-socket CPU".  Two UPPER bounds on what one
            # socket running the reference's configuration (site repeats) can do: perfect scaling
            # of the one-thread rate over its physical cores (`one_socket_ideal`), and its DRAM
            # bandwidth over the bytes the traversal must move (`one_socket_bandwidth_bound`).  A
            # real socket is below both, so the speed-up over it is at least GPU / the smaller.
            bound = ideal["value"]
            bw = result["cpu_baseline"].get("one_socket_bandwidth_bound", {}).get("value")
            result["cpu_baseline"]["speedup_vs_one_socket_ideal"] = round(value / ideal["value"], 2)
            if bw and bw > 0:
                result["cpu_baseline"]["speedup_vs_one_socket_bandwidth_bound"] = round(value / bw, 2)
                bound = min(bound, bw)
            result["cpu_baseline"]["speedup_lower_bound"] = round(value / bound, 2)
            result["cpu_baseline"]["speedup_lower_bound_basis"] = (
                "one_socket_bandwidth_bound" if bw and bw < ideal["value"] else "one_socket_ideal")
            result["cpu_baseline"]["speedup_vs_one_thread_with_repeats"] = round(
                value / result["cpu_baseline"]["with_site_repeats"]["value"], 1)

    emitted = threading.Event()

    def emit():
        if emitted.is_set():
            return
        emitted.set()
        if rank == 0:
            line = json.dumps(result) + "\n"
            if real_stdout is not None:
                os.write(real_stdout, line.encode())
            else:
                sys.stdout.write(line)
                sys.stdout.flush()

    # The extra legs of the default N > 1 line (site blocks / grid with the all-reduce of the
    # per-block lnLs, and the all-reduce that counts the RCCL ranks) come AFTER the measurement the
    # line is about, and must never cost it: if a leg raises, its object says so; if the ranks do not
    # get through the legs within --leg-timeout seconds (a collective that never completes on this
    # fabric), every rank prints / leaves on its own -- rank 0 with the line as far as it got.
    want_legs = (use_pg and (world > 1 or args.one_rank_shard_legs) and args.shard == "candidates" and use_fused and
                 not args.no_shard_legs)
    if use_pg and (want_legs or not host_collectives):
        def give_up():
            for key in ("site_sharded", "grid"):
                if want_legs and key not in result and (key == "site_sharded" or
                                                        (world % args.site_groups == 0 and world // args.site_groups > 1)):
                    result[key] = {"error": "not finished within %.0f s" % args.leg_timeout}
            emit()
            # The legs are EXTRAS behind the measurement the line is about (which is complete and
            # printed): a leg that hangs says so in its object and on stderr, and the run still
            # counts -- a non-zero status here would throw the headline away with it.  Status 3
            # only when the line itself could not be completed.
            sys.stderr.write("bench.py: a sharded leg did not finish within %.0f s; its object holds the error\n" % args.leg_timeout)
            os._exit(0 if "value" in result else 3)
        watchdog = threading.Timer(args.leg_timeout, give_up)
        watchdog.daemon = True
        watchdog.start()
        try:
            if want_legs:
                check_one_hip_runtime(rd)
                for key, mode in (("site_sharded", "sites"), ("grid", "grid")):
                    if mode == "grid" and not (world % args.site_groups == 0 and world // args.site_groups > 1):
                        continue
                    try:
                        result[key] = sharded_leg(mode)
                    except Exception as e:   # noqa: BLE001  (the headline must survive a failing leg)
                        result[key] = {"error": "%s: %s" % (type(e).__name__, e)}
                        break                # (the other ranks may be inside a collective: no further legs)
            if not host_collectives and not any("error" in v for v in result.values() if isinstance(v, dict)):
                ones = torch.ones(1, dtype=torch.float64, device="cuda")   # a real all-reduce over the communicator
                tdist.all_reduce(ones)
                result["rccl_ranks"] = int(round(float(ones.item())))
        finally:
            watchdog.cancel()
    emit()
    if use_pg:
        tdist.destroy_process_group()


def host_topology():
    """(sockets, physical cores per socket, logical cpus, {socket: [one logical cpu per
    physical core]}) from /proc/cpuinfo."""
    cores = {}
    cpu = phys = core = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("processor"):
                cpu = int(line.split(":")[1])
            elif line.startswith("physical id"):
                phys = int(line.split(":")[1])
            elif line.startswith("core id"):
                core = int(line.split(":")[1])
            elif not line.strip() and None not in (cpu, phys, core):
                cores.setdefault((phys, core), cpu)     # first hardware thread of the core
                cpu = phys = core = None
    except OSError:
        pass
    logical = os.cpu_count() or 1
    if not cores:
        return 1, logical, logical, {0: sorted(os.sched_getaffinity(0))}
    by_socket = {}
    for (p_, _), c in cores.items():
        by_socket.setdefault(p_, []).append(c)
    sockets = len(by_socket)
    return sockets, max(1, len(cores) // sockets), logical, {k: sorted(v) for k, v in by_socket.items()}


def host_load():
    """What else the host is doing / is allowed to do: the cgroup's CPU quota (cpu.max), its
    memory limit, the load average, and the CPUs this process may run on."""
    def read(path):
        try:
            return open(path).read().strip()
        except OSError:
            return None
    out = {"cgroup_cpu_max": read("/sys/fs/cgroup/cpu.max"),
           "cgroup_memory_max": read("/sys/fs/cgroup/memory.max"),
           "affinity_cpus": len(os.sched_getaffinity(0))}
    try:
        out["loadavg"] = [round(x, 2) for x in os.getloadavg()]
    except OSError:
        out["loadavg"] = None
    if out["cgroup_cpu_max"] is None:   # cgroup v1
        q, per = read("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), read("/sys/fs/cgroup/cpu/cpu.cfs_period_us")
        out["cgroup_cpu_max"] = "%s %s" % (q, per) if q and per else None
    return out


def socket_dram_bandwidth():
    """Nominal DRAM bandwidth of ONE socket of this host in GB/s, from the CPU's model name
    (memory channels x transfer rate x 8 bytes of the platform the model belongs to; the DIMMs
    themselves are only visible to root: dmidecode).  -> (GB/s or None, how it was derived)"""
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    import re
    table = [   # (pattern on the model name, channels, MT/s, platform)
        (r"EPYC 9\d\d5", 12, 6000, "AMD EPYC 9005 (SP5): 12 channels DDR5-6000"),
        (r"EPYC 9\d\d4", 12, 4800, "AMD EPYC 9004 (SP5): 12 channels DDR5-4800"),
        (r"EPYC 8\d\d4", 6, 4800, "AMD EPYC 8004 (SP6): 6 channels DDR5-4800"),
        (r"EPYC 7\d\d[23]", 8, 3200, "AMD EPYC 7002 / 7003 (SP3): 8 channels DDR4-3200"),
        (r"EPYC 7\d\d1", 8, 2666, "AMD EPYC 7001 (SP3): 8 channels DDR4-2666"),
        (r"Xeon.*(Platinum|Gold) [86]5\d\d", 8, 5600, "Intel Xeon 5th gen: 8 channels DDR5-5600"),
        (r"Xeon.*(Platinum|Gold) [86]4\d\d", 8, 4800, "Intel Xeon 4th gen: 8 channels DDR5-4800"),
        (r"Xeon.*(Platinum|Gold) [86]3\d\d", 8, 3200, "Intel Xeon 3rd gen: 8 channels DDR4-3200"),
    ]
    for pat, ch, mts, what in table:
        if re.search(pat, model):
            return ch * mts * 8 / 1000.0, "nominal, %s (model name '%s'; dmidecode needs root)" % (what, model)
    return None, "model name '%s' not in bench.py's table" % model


def site_repeats_class_ratio(tree, seqs, samples=3):
    """Sum over inner nodes of (distinct tip patterns below the node) / (inner nodes x
    patterns), averaged over a few rootings: the fraction of the CLV arithmetic a CPU run
    WITH coraxlib's site repeats (the reference's configuration for 4-state data,
    src/model.cpp:145-149) would still do.  The CPU baseline below has no repeats."""
    import numpy as np
    names = list(seqs)
    S = len(seqs[names[0]])
    tips = {tree.tip_index(k): np.frombuffer(seqs[k].upper().encode(), dtype=np.uint8).astype(np.int64)
            for k in names}
    nroots = tree.root_count()
    ratios = []
    for rid in range(0, nroots, max(1, nroots // samples))[:samples]:
        ops, _, _ = tree.generate_operations(tree.root_location(rid))
        cls, total = dict(tips), 0
        for op in ops:
            a, b = cls[op.child1_clv_index], cls[op.child2_clv_index]
            uniq, inv = np.unique(a * (int(b.max()) + 1) + b, return_inverse=True)
            cls[op.parent_clv_index] = inv.astype(np.int64)
            total += len(uniq)
        ratios.append(total / (len(ops) * S))
    return round(float(np.mean(ratios)), 4)


def cpu_baseline(w, tree, cmap, freqs, n, S, K, R, params, roots, budget, gpu_eval, weights=None):
    """The CPU oracle (oracle/rd_oracle.c) on the SAME workload, bounded to `budget`
    seconds; also the in-run parity gate (<= 1e-9).  4-state data go through the
    oracle's 256-bit-vector CLV loop (bit-identical to its scalar loop; the
    reference selects coraxlib's AVX2 kernel for nucleotides,
    /root/reference/src/model.cpp:145-155).  Two figures: one thread
    -- what one reference rank spends on CLV work with one partition -- and one
    candidate root per thread on the physical cores of ONE socket, what
    `mpirun -np <cores> rd` does (src/model.cpp:1899-1907)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import OraclePartition, ORC_MAP_NT
    ocmap = ORC_MAP_NT if K == 4 else cmap
    vec = K == 4

    def make_partition():
        o = OraclePartition.for_tree(tree, K, S, R)
        for label, seq in w["seqs"].items():
            o.set_tip_states(tree.tip_index(label), ocmap, seq)
        if weights is not None:
            o.set_pattern_weights(weights)
        o.set_frequencies(0, freqs)
        o.set_category_rates(w["rates"])
        return o

    def evaluate(o, j, avx2=vec):
        o.set_subst_params(0, params[j])
        ops, pmi, brl = scheds[j]
        o.update_prob_matrices(pmi, brl)
        o.update_clvs(ops, avx2=avx2)
        return o.compute_root_loglikelihood(tree.root_clv_index(), tree.root_scaler_index())

    scheds = []
    for rl in roots[:64]:
        ops, pmi, brl = tree.generate_operations(rl)
        scheds.append((OraclePartition.pack_ops(ops), pmi, brl))
    o = make_partition()
    done, worst, dt = 0, 0.0, 0.0
    while True:
        j = done % len(scheds)
        t0 = time.perf_counter()
        ref = evaluate(o, j)
        dt += time.perf_counter() - t0             # CPU time only
        done += 1
        got = gpu_eval(j)                          # parity gate, not timed
        worst = max(worst, abs(got - ref) / abs(ref))
        if dt >= 0.5 * budget or done >= 64:
            break
    if worst > 1e-9:
        raise SystemExit("parity gate failed: GPU vs oracle rel.err %.3e" % worst)
    loop = "256-bit-vector CLV loop (AVX2, no FMA), " if vec else "scalar CLV loop, "
    sockets, per_socket, logical, socket_cpus = host_topology()
    out = {"value": round(done / dt, 4), "unit": "evals/s", "cores": 1, "kind": "port",
           "sample": "%d full-traversal evaluations of the same workload (oracle/rd_oracle.c, 1 thread, "
                     "%s-O3 x86-64-v3, no site repeats)" % (done, loop),
           "host": {"sockets": sockets, "cores_per_socket": per_socket, "logical_cpus": logical},
           "parity_max_rel_err": worst}
    if vec:   # the scalar loop of the same oracle, for comparison with earlier rounds
        t0 = time.perf_counter()
        k = 0
        while k < 2 or (time.perf_counter() - t0 < 0.1 * budget and k < 16):
            evaluate(o, k % len(scheds), avx2=False)
            k += 1
        out["scalar_loop_1_thread"] = round(k / (time.perf_counter() - t0), 4)

    if K == 4:
        # The reference's own configuration: coraxlib WITH subtree site repeats
        # (CORAX_ATTRIB_SITE_REPEATS, /root/reference/src/model.cpp:145-149).  The oracle's
        # restatement of that scheme (orc_update_clvs_repeats: libpll-2's pll_update_repeats --
        # class tables rebuilt per traversal, CLVs per class) is bit-identical to the plain
        # loop and is the HONEST comparator: what one `rd` rank executes per evaluation.
        q = make_partition()

        def evaluate_rep(j):
            q.set_subst_params(0, params[j])
            ops, pmi, brl = scheds[j]
            q.update_prob_matrices(pmi, brl)
            q.update_clvs_repeats(ops, avx2=True)
            return q.compute_root_loglikelihood_repeats(tree.root_clv_index(), tree.root_scaler_index())

        if evaluate_rep(0) != evaluate(o, 0):
            raise SystemExit("oracle: the site-repeats traversal differs from the plain one")
        b0 = q.repeats_bytes()
        t0 = time.perf_counter()
        k = 0
        while k < 2 or (time.perf_counter() - t0 < 0.2 * budget and k < 64):
            evaluate_rep(k % len(scheds))
            k += 1
        rep_rate = k / (time.perf_counter() - t0)
        rep_bytes = [(b1 - b0_) / k for b0_, b1 in zip(b0, q.repeats_bytes())]   # per evaluation
        out["with_site_repeats"] = {
            "value": round(rep_rate, 4), "unit": "evals/s", "cores": 1,
            "class_ratio": round(q.repeats_ratio(), 4),
            "ratio_to_no_repeats": round(rep_rate / max(out["value"], 1e-12), 3),
            "sample": "%d full-traversal evaluations, 1 thread, subtree site repeats as coraxlib does them "
                      "(per-node column classes through a lookup table, rebuilt for every traversal; "
                      "CLVs and scalers per class; 256-bit-vector inner loop); lnL bit-identical to the "
                      "plain loop (asserted here and in tests/test_oracle_golden.py)" % k}
        q.destroy()
        # what ONE idealised socket of this host would do: every physical core at the one-thread
        # rate, no memory-system contention -- an UPPER bound on the CPU side, hence a LOWER
        # bound on the speed-up (filled in by main(): GPU evals/s / this)
        out["one_socket_ideal"] = {"value": round(rep_rate * per_socket, 3), "unit": "evals/s",
                                   "cores": per_socket,
                                   "note": "with_site_repeats x the physical cores of one socket of this host"}
        # ... and what bounds a REAL socket: its DRAM bandwidth.  The oracle counts the bytes its
        # repeats traversal moves (orc_repeats_bytes): class CLVs, scalers and class arrays WRITTEN,
        # plus what must be READ at least once (the children's class arrays, every class CLV of an
        # inner child once) -- no write-allocate traffic, no child class read twice: the least a
        # cache hierarchy far smaller than the working set (c2: 99 nodes x up to 6.4 MB per
        # candidate and core) can get away with.  Bandwidth / those bytes is an UPPER bound on the
        # socket's rate however many cores it has.
        moved = rep_bytes[0] + rep_bytes[1]
        nominal, how = socket_dram_bandwidth()
        bb = {"unit": "evals/s", "cores": per_socket,
              "bytes_per_evaluation": {"written": round(rep_bytes[0]), "read_at_least_once": round(rep_bytes[1]),
                                       "moved": round(moved), "read_if_nothing_is_cached": round(rep_bytes[2])},
              "note": "one socket's DRAM bandwidth / the bytes coraxlib's site-repeats traversal cannot avoid moving "
                      "per evaluation (counted by the oracle: orc_repeats_bytes; no write-allocate, every child class "
                      "read once): an upper bound on the socket's rate"}
        if nominal:
            bb.update(value=round(nominal * 1e9 / moved, 3), bandwidth_gbs=round(nominal, 1), bandwidth_source=how)
        out["one_socket_bandwidth_bound"] = bb
        # (the census by another route: distinct tip patterns below every node)
        out["site_repeats_class_ratio"] = site_repeats_class_ratio(tree, w["seqs"])
        out["site_repeats_note"] = ("the reference sets CORAX_ATTRIB_SITE_REPEATS for 4-state data; `value` is the "
                                    "plain loop (every site), `with_site_repeats` the class-compressed one")

    # one socket: one candidate root per thread, one oracle partition per thread (ctypes
    # calls run without the GIL); capped by host memory (a partition holds all 2n-2 CLVs).
    # Every thread is PINNED to its own physical core of socket 0 and builds (first-touches)
    # its partition there, so the figure is one socket's cores on that socket's memory.
    import threading
    try:
        import psutil
        avail = psutil.virtual_memory().available
    except Exception:
        avail = 16 << 30
    per_part = (2 * n - 2) * S * R * K * 8 * 1.15 + n * S * 8
    allowed = os.sched_getaffinity(0)
    socket0 = [c for c in socket_cpus[min(socket_cpus)] if c in allowed]
    # (a cgroup CPU quota below the core count -- a leased slice of a node -- is all the CPU
    # time the threads can get: more threads than that only take turns)
    quota = None
    cm = host_load()["cgroup_cpu_max"]
    if cm and cm.split()[0] not in ("max", "-1"):
        try:
            quota = max(1, int(float(cm.split()[0]) / float(cm.split()[1])))
        except (ValueError, IndexError, ZeroDivisionError):
            quota = None
    threads = int(max(1, min(len(socket0), quota or len(socket0), 0.4 * avail // per_part, len(scheds))))
    if "one_socket_bandwidth_bound" in out:
        # STREAM triad on the cores this process may use (pinned, first touch, 64 MB per array and
        # thread): what the memory system gives THIS slice of the host -- beside the nominal figure,
        # and the bound's bandwidth where the CPU model is not in the table (scaled to the socket's
        # cores: generous, a socket saturates long before its last core)
        from oracle_lib import stream_triad
        tt = int(max(1, min(len(socket0), quota or len(socket0))))
        gbs = stream_triad(socket0[:tt], 1 << 23, min(1.5, 0.1 * budget))
        bb = out["one_socket_bandwidth_bound"]
        bb["stream_triad"] = {"gbs": round(gbs, 1), "threads": tt,
                              "note": "a[i] = b[i] + s c[i], 24 bytes per element, one pinned thread per physical core "
                                      "of socket 0 the cgroup quota allows"}
        moved = bb["bytes_per_evaluation"]["moved"]
        bb["value_at_measured_triad"] = round(gbs * 1e9 / moved, 3)
        if "value" not in bb:
            scaled = gbs * per_socket / tt
            bb.update(value=round(scaled * 1e9 / moved, 3), bandwidth_gbs=round(scaled, 1),
                      bandwidth_source="STREAM triad of %d threads x (%d cores per socket / %d): %s"
                                       % (tt, per_socket, tt, socket_dram_bandwidth()[1]))
    if threads > 1:
        counts = [0] * threads
        window = max(0.4 * budget, 2.0 / max(out["value"], 1e-9))
        ready = threading.Barrier(threads + 1)
        start = threading.Barrier(threads + 1)
        pinned = [False] * threads

        def work(t):
            try:
                os.sched_setaffinity(0, {socket0[t]})      # the calling thread only
                pinned[t] = True
            except OSError:
                pass
            q = make_partition()                            # first touch from this core
            evaluate(q, t % len(scheds))                    # ... of the CLV buffers too
            ready.wait()
            start.wait()
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < window or counts[t] == 0:
                evaluate(q, (t + counts[t] * threads) % len(scheds))
                counts[t] += 1
            q.destroy()

        ths = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
        for th in ths:
            th.start()
        ready.wait()
        start.wait()
        t0 = time.perf_counter()
        for th in ths:
            th.join()
        wall = time.perf_counter() - t0
        rates = sorted(c / wall for c in counts)
        agg = sum(counts) / wall
        # An honest name for what was measured: "one socket" only if the threads
        # really had a socket's cores to themselves -- i.e. the aggregate is at least 20 x the
        # one-thread rate (64 cores streaming their own CLVs reach that on an idle socket).  On
        # a leased slice of a busy node (cgroup CPU quota, neighbours on the memory system) the
        # figure is "what this many host threads got here", and says so.
        leg = "one_socket" if agg >= 20.0 * out["value"] and threads >= 0.9 * per_socket else "host_threads"
        out[leg] = {"value": round(agg, 4), "unit": "evals/s", "cores": threads,
                    "ratio_to_one_thread": round(agg / max(out["value"], 1e-12), 2),
                    "per_thread_evals_per_s": {"min": round(rates[0], 4), "median": round(rates[len(rates) // 2], 4),
                                               "max": round(rates[-1], 4)},
                    "sample": "%d evaluations, one candidate root per thread, %d threads%s, each "
                              "with its own partition first-touched from its core "
                              "(%d socket(s) x %d physical cores on this host%s) for %.1f s"
                              % (sum(counts), threads,
                                 " pinned one per physical core of socket 0" if all(pinned) else " (not pinned)",
                                 sockets, per_socket,
                                 "" if threads == per_socket else "; capped by memory/affinity",
                                 wall)}
        if leg == "host_threads":
            out[leg]["note"] = ("NOT a single-socket figure: %d pinned threads (cgroup CPU quota: %s CPUs) reached "
                                "%.1f x the one-thread rate; the host is a leased, shared slice of a %d x %d-core "
                                "node, see `host`" % (threads, quota if quota else "none", agg / max(out["value"], 1e-12),
                                                      sockets, per_socket))
    out["host"].update(host_load())
    o.destroy()
    return out


if __name__ == "__main__":
    main()
