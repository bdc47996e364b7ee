#!/usr/bin/env python3
"""bench.py -- candidate-root lnL evaluations/sec on BASELINE config c2.

  python bench.py --gpus N --steps K --warmup W

One "step" = one batch of `--batch` full-traversal candidate-root evaluations
(2n-2 P-matrices + n-1 CLV operations + root reduction each; model_t::compute_lh,
/root/reference/src/model.cpp:384-413) with a different root placement AND a
different substitution-parameter set per evaluation (what the exhaustive
search does, src/model.cpp:1154-1229).  Inputs (tip codes, weights) are
resident in HBM before the timed region.

Multi-GPU (one process per GPU, launched by torch.distributed.run): candidate
roots are sharded across ranks the way the reference shards them across MPI
ranks (src/model.cpp:1867-1911) -- independent work, no data-path collective,
weak scaling (per-GPU batch fixed).  `value` is the whole-job rate.

Rank 0 prints ONE JSON line with `roofline` (dominant kernel, HIP-event timed
on the partition's own stream) and `cpu_baseline` (the CPU oracle timed on a
bounded sample of the same workload on this box's host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    "c1": dict(n=10, S=1000, K=4, R=1),
    "c2": dict(n=100, S=50000, K=4, R=4),
    "c3": dict(n=200, S=10000, K=20, R=4),
    "c5": dict(n=1000, S=100000, K=4, R=4),
}
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=8,
                    help="candidate-root evaluations per step per GPU")
    ap.add_argument("--cpu-seconds", type=float, default=15.0,
                    help="budget for the cpu_baseline leg (rank 0, N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))

    import numpy as np
    import torch
    import root_digger_amd as rd
    from root_digger_amd import synth, dist as rdist

    if not torch.cuda.is_available() or rd.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device; there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    rd.set_device(local_rank)
    if world > 1:
        import torch.distributed as tdist
        tdist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    cfg = CONFIGS[args.config]
    n, S, K, R = cfg["n"], cfg["S"], cfg["K"], cfg["R"]
    seed = 0xD166E5 + sorted(CONFIGS).index(args.config)
    w = synth.workload(n, S, K, R, seed)
    tree = rd.Tree.from_newick(w["newick"])
    cmap = rd.MAP_NT
    if K != 4:
        import ctypes
        cmap = (ctypes.c_uint64 * 256)()
        for i, ch in enumerate(w["alphabet"]):
            cmap[ord(ch)] = 1 << i

    part = rd.Partition.for_tree(tree, K, S, R)
    for label, seq in w["seqs"].items():
        part.set_tip_states(tree.tip_index(label), cmap, seq)
    freqs = part.empirical_frequencies()
    part.set_frequencies(0, freqs)
    part.set_category_rates(w["rates"])

    # this rank's candidate roots (src/model.cpp:1899-1907) and per-candidate
    # parameter sets (random_params, src/model.cpp:87-93)
    mine = rdist.assign_candidates(tree.root_count(), rank, world)
    rng = np.random.default_rng(seed + 1000 + rank)
    params = [synth.random_params(K * K - K, rng) for _ in range(len(mine))]
    roots = [tree.root_location(i) for i in mine]

    def evaluate(k):
        j = k % len(mine)
        part.set_subst_params(0, params[j])
        ops, pmi, brl = tree.generate_operations(roots[j])
        part.update_prob_matrices(pmi, brl)
        part.update_clvs(ops)
        return part.compute_root_loglikelihood(tree.root_clv_index(),
                                               tree.root_scaler_index())

    def step(s):
        out = 0.0
        for b in range(args.batch):
            out += evaluate(s * args.batch + b)
        return out

    def barrier():
        if world > 1:
            tdist.barrier()
        torch.cuda.synchronize()
        part.sync()

    for s in range(args.warmup):
        step(s)
    part.profile_enable(True)
    barrier()
    t0 = time.perf_counter()
    check = 0.0
    for s in range(args.steps):
        check += step(args.warmup + s)
    barrier()
    elapsed = time.perf_counter() - t0
    prof = part.profile_read()
    part.profile_enable(False)
    if not np.isfinite(check):
        raise SystemExit("non-finite lnL in the timed region")

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        elapsed = float(t.item())

    evals_per_rank = args.steps * args.batch
    total_evals = evals_per_rank * world
    value = total_evals / elapsed

    clv_ms, clv_launches = prof["clv"]
    bytes_clv = clv_kernel_bytes(n, S, R, K) * evals_per_rank
    avg_launch_ms = clv_ms / max(clv_launches, 1)
    achieved = bytes_clv / (clv_ms * 1e-3) / 1e9 if clv_ms > 0 else 0.0
    roofline = {
        "kernel": "clv_dna_level_kernel" if K == 4 else "clv_generic_level_kernel",
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
        "traffic": None,
        "bytes_per_launch": bytes_clv / max(clv_launches, 1),
        "avg_launch_ms": round(avg_launch_ms, 5), "launches": clv_launches,
        "clv_share_of_step": round(clv_ms * 1e-3 / elapsed, 3),
    }

    result = {
        "metric": "candidate-root lnL evals/sec", "value": round(value, 2),
        "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s: %d-taxon %d-site %d-state UNREST+G%d full-traversal "
                               "root lnL" % (args.config, n, S, K, R),
                   "batch_per_gpu": args.batch, "sharding": "candidate roots"},
        "site_clv_updates_per_sec": round(value * (n - 1) * S, 1),
        "algorithmic_GBps_full_eval": round(value / world * full_eval_bytes(n, S, R, K) / 1e9, 1),
        "roofline": roofline,
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(w, tree, cmap, freqs, n, S, K, R,
                                              params, roots, args.cpu_seconds,
                                              evaluate)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        tdist.destroy_process_group()


def cpu_baseline(w, tree, cmap, freqs, n, S, K, R, params, roots, budget, gpu_eval):
    """The CPU oracle (oracle/rd_oracle.c, single thread: what one reference rank
    spends on CLV work with one partition) on the SAME workload,
    bounded to `budget` seconds; also the in-run parity gate (<= 1e-9)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import OraclePartition, ORC_MAP_NT
    import ctypes
    ocmap = ORC_MAP_NT if K == 4 else cmap
    o = OraclePartition.for_tree(tree, K, S, R)
    for label, seq in w["seqs"].items():
        o.set_tip_states(tree.tip_index(label), ocmap, seq)
    o.set_frequencies(0, freqs)
    o.set_category_rates(w["rates"])
    done, worst, dt = 0, 0.0, 0.0
    while True:
        j = done % len(roots)
        t0 = time.perf_counter()
        o.set_subst_params(0, params[j])
        ops, pmi, brl = tree.generate_operations(roots[j])
        o.update_prob_matrices(pmi, brl)
        o.update_clvs(ops)
        ref = o.compute_root_loglikelihood(tree.root_clv_index(), tree.root_scaler_index())
        dt += time.perf_counter() - t0             # CPU time only
        done += 1
        got = gpu_eval(j)                          # parity gate, not timed
        worst = max(worst, abs(got - ref) / abs(ref))
        if dt >= budget or done >= 64:
            break
    if worst > 1e-9:
        raise SystemExit("parity gate failed: GPU vs oracle rel.err %.3e" % worst)
    return {"value": round(done / dt, 4), "unit": "evals/s", "cores": 1, "kind": "port",
            "sample": "%d full-traversal evaluations of the same workload (oracle/rd_oracle.c, "
                      "1 thread, -O3 x86-64-v3, no site repeats)" % done,
            "host_cores": os.cpu_count(), "parity_max_rel_err": worst}


if __name__ == "__main__":
    main()
