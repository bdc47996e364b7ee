#!/usr/bin/env python3
"""Root placement from the command line: the `rd --msa M --tree T [--exhaustive]`
entry of the reference (/root/reference/src/main.cpp:411-680).

  python -m root_digger_amd.cli --msa aln.fasta --tree t.nwk --prefix out --exhaustive \\
         [--rate-cats 4] [--lbfgsb /path/to/liblbfgsb.so] [--early-stop]

As in the reference, the default mode is the heuristic search (starting roots
picked by --initial-root-strategy, --min-roots and --root-ratio;
src/model.cpp:1008-1137); --exhaustive evaluates every branch
(src/model.cpp:1139-1272) and adds <prefix>.lwr.tree.

Every finished candidate root goes into <prefix>.ckp, the reference's own
checkpoint format (csrc/checkpoint.hpp): an interrupted run resumes from it,
`--clean` repairs it, and the processes of a multi-GPU run -- one per GPU,
started by `python -m torch.distributed.run`, which sets RANK / LOCAL_RANK /
WORLD_SIZE -- take the candidate chunks the reference's MPI ranks take
(src/model.cpp:1867-1911) and meet in that file; rank 0 writes the trees.

Writes <prefix>.lwr.tree (every branch annotated with LWR, LLH and alpha,
src/model.cpp:1237-1268) and <prefix>.rooted.tree (the tree rooted at the best
placement, src/main.cpp:636-654).  Without --lbfgsb the substitution
parameters stay at their start values (uniform rates, empirical frequencies)
and only the root position is optimised; with it, the caller's L-BFGS-B
(`setulb`) optimises them exactly as optimize_params does, on the batched GPU
objective.  Code defaults follow src/util.hpp:159-177."""
import argparse
import ctypes
import math
import os
import sys
import time

from . import MAP_BIN, MAP_NT, Checkpoint, Model, RdamdError, Tree, parse_model_info, set_device

def version_string():
    from . import lib
    lib.rdamd_version.restype = ctypes.c_char_p
    return lib.rdamd_version().decode()


STRATEGIES = ["random", "midpoint", "modified-mad"]   # initial_root_strategy_t, src/util.hpp:74-78


def main(argv=None):
    ap = argparse.ArgumentParser(prog="root_digger_amd.cli")
    ap.add_argument("--msa", required=True)
    ap.add_argument("--tree", required=True)
    ap.add_argument("--prefix", default=None)
    ap.add_argument("--states", type=int, default=4, choices=[2, 4],
                    help="4: nucleotides; 2: binary characters (src/main.cpp:484-488; parameter "
                         "optimisation is available for 4 states)")
    ap.add_argument("--rate-cats", type=int, default=1)
    ap.add_argument("--rate-cats-type", default="mean", choices=["mean", "median", "free"])
    ap.add_argument("--partition", default=None,
                    help="partition file: <MODEL>, <NAME> = <BEGIN>-<END>[, ...] per line; rate "
                         "categories then come from each line's model string")
    ap.add_argument("--model", default=None,
                    help="model string, e.g. UNREST+G4 (only its rate heterogeneity is used, "
                         "src/main.cpp:491-510)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--atol", type=float, default=1e-7)       # abs_tolerance
    ap.add_argument("--bfgstol", type=float, default=1e-7)
    ap.add_argument("--brtol", type=float, default=1e-12)
    ap.add_argument("--factor", type=float, default=1e4)
    ap.add_argument("--early-stop", action="store_true")
    ap.add_argument("--no-early-stop", action="store_true", help="force disable early stop")
    ap.add_argument("--invariant-sites", action="store_true",
                    help="accepted for compatibility; the proportion is pinned to 0 as in the "
                         "reference (src/model.cpp:292-300)")
    ap.add_argument("--verbose", action="count", default=0, help="accepted for compatibility")
    ap.add_argument("--debug", action="store_true", help="accepted for compatibility")
    ap.add_argument("--mpi-debug", action="store_true", help="accepted for compatibility")
    ap.add_argument("--echo", action="store_true", help="print the tree as read")
    ap.add_argument("--version", action="version", version=version_string())
    ap.add_argument("--exhaustive", action="store_true",
                    help="evaluate every branch as a root (default: heuristic search)")
    ap.add_argument("--min-roots", type=int, default=1)
    ap.add_argument("--root-ratio", type=float, default=0.01)
    ap.add_argument("--initial-root-strategy", default="modified-mad",
                    choices=["random", "midpoint", "modified-mad"])
    ap.add_argument("--lbfgsb", default=None,
                    help="shared library exporting the L-BFGS-B entry point `setulb`")
    ap.add_argument("--workers", "--threads", dest="workers", type=int, default=4,
                    help="host threads, each with its own model replica / HIP stream "
                         "(0 = the plain sequential loop)")
    ap.add_argument("--lockstep", type=int, default=-1,
                    help="candidates advanced in lock step, their optimiser batches merged "
                         "into one launch (overrides --workers; 0 = off; default: 16 when "
                         "parameters are optimised on a single partition)")
    ap.add_argument("--device", type=int, default=None,
                    help="HIP device (default: LOCAL_RANK, else 0)")
    ap.add_argument("--silent", action="store_true")
    ap.add_argument("--clean", action="store_true",
                    help="repair the checkpoint file and exit (src/main.cpp:138-146)")
    ap.add_argument("--no-checkpoint", action="store_true",
                    help="keep results in memory only (single process)")
    args = ap.parse_args(argv)
    prefix = args.prefix or args.msa
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and args.no_checkpoint:
        ap.error("--no-checkpoint: the ranks of a multi-process run meet in the checkpoint file")

    def barrier():
        if world > 1:
            import torch.distributed as tdist
            if not tdist.is_initialized():
                tdist.init_process_group("gloo")     # host-side rendezvous only
            tdist.barrier()

    t0 = time.time()
    # the HIP device is claimed BEFORE the rendezvous: a multi-rank gloo group
    # created first leaves this process without visible devices
    device = args.device if args.device is not None else int(os.environ.get("LOCAL_RANK", "0"))
    set_device(device)
    # mpi_create_checkpoint, src/main.cpp:366-409: rank 0 creates / repairs the
    # file and writes the header, the others open it afterwards; an existing
    # header overrides the search options given here (merge_options_checkpoint)
    ckp = None
    if not args.no_checkpoint:
        if rank == 0:
            ckp = Checkpoint(prefix)
            if args.clean:
                ckp.clean()
                return 0
            stored = ckp.load_options()
            if stored is None:
                ckp.save_options({
                    "msa_filename": args.msa, "tree_filename": args.tree, "prefix": prefix,
                    "partition_filename": args.partition or "", "model_string": args.model or "",
                    "data_type": "bin" if args.states == 2 else "nt", "rate_cats": [args.rate_cats], "seed": args.seed,
                    "threads": args.workers, "abs_tolerance": args.atol, "factor": args.factor,
                    "br_tolerance": args.brtol, "bfgs_tol": args.bfgstol,
                    "silent": args.silent, "exhaustive": args.exhaustive,
                    "min_roots": args.min_roots, "root_ratio": args.root_ratio,
                    "initial_root_strategy": STRATEGIES.index(args.initial_root_strategy),
                    "invariant_sites": args.invariant_sites,
                    "early_stop": 2 if args.no_early_stop else 1 if args.early_stop else 0})
            if ckp.needs_cleaning():
                ckp.clean()
        elif args.clean:
            return 0
        barrier()
        if rank != 0:
            ckp = Checkpoint(prefix)
        stored = ckp.load_options()
        if stored is not None and ckp.existing_checkpoint():
            if not args.silent and rank == 0:
                print("Loading options from the checkpoint file. Some cli options are ignored. "
                      "If the program is not working, try deleting the checkpoint file",
                      file=sys.stderr)
            args.msa, args.tree = stored["msa_filename"], stored["tree_filename"]
            args.partition = stored["partition_filename"] or None
            args.model = stored["model_string"] or None
            args.rate_cats = int(stored["rate_cats"][0]["rate_cats"])
            args.seed, args.atol, args.factor = stored["seed"], stored["abs_tolerance"], stored["factor"]
            args.brtol, args.bfgstol = stored["br_tolerance"], stored["bfgs_tol"]
            args.exhaustive = bool(stored["exhaustive"])
            args.min_roots, args.root_ratio = stored["min_roots"], stored["root_ratio"]
            args.initial_root_strategy = STRATEGIES[stored["initial_root_strategy"]]
            # initialized_flag_t: unset means "stop early unless exhaustive" (src/main.cpp:583)
            args.early_stop = (stored["early_stop"] == 1 or
                               (stored["early_stop"] == 0 and not args.exhaustive))
            args.no_early_stop = stored["early_stop"] == 2

    tree = Tree.from_file(args.tree)
    if args.min_roots > tree.root_count():
        raise SystemExit("Min roots is larger than the number of roots on the tree")
    # early_stop.convert_with_default(!exhaustive), src/main.cpp:583
    early_stop = (args.early_stop or not args.exhaustive) and not args.no_early_stop
    cmap = MAP_BIN if args.states == 2 else MAP_NT
    if args.model:
        # (a string without +G / +R means one category here; the reference rejects it
        # with "Rate categories cannot be zero", src/main.cpp:557-560)
        args.rate_cats = parse_model_info(args.model)["ratehet"]["rate_cats"] or 1
    if args.partition:
        model = Model.from_partition_file(tree, args.msa, args.partition, states=args.states,
                                          cmap=cmap, seed=args.seed, early_stop=early_stop)
        if args.lockstep > 0:
            ap.error("--lockstep handles a single partition")
    else:
        model = Model.from_file(tree, args.msa, states=args.states, cmap=cmap,
                                rate_cats=args.rate_cats, seed=args.seed, early_stop=early_stop,
                                rate_category_type=args.rate_cats_type)
    if args.echo:
        print(tree.newick(True))
    try:
        model.initialize_partitions()
    except RdamdError:      # a state that never occurs: uniform frequencies, src/main.cpp:577-581
        model.initialize_partitions_uniform_freqs()
    keep = None
    if args.lbfgsb:
        keep = ctypes.CDLL(args.lbfgsb)
        model.set_lbfgsb(keep.setulb)
    model.compute_lh(tree.root_location(0))                    # model.initialize()
    if ckp is not None:
        model.set_checkpoint(ckp)
    if args.lockstep < 0:   # same results either way; lock step fills the GPU on small alignments
        args.lockstep = 16 if (args.lbfgsb and not args.partition) else 0
    if args.exhaustive:
        model.assign_by_rank(rank, world, ckp)                 # src/main.cpp:612-615
        barrier()
        if not args.silent and rank == 0:
            print("Starting exhaustive search", flush=True)
            model.set_progress(True)
        res = model.exhaustive_search(args.atol, args.bfgstol, args.brtol, args.factor,
                                      workers=args.workers, lockstep=args.lockstep)
    else:
        if not args.lbfgsb:
            ap.error("the heuristic search optimises the model parameters: it needs --lbfgsb")
        model.assign_by_rank_search(args.min_roots, args.root_ratio, rank, world,
                                    args.initial_root_strategy.replace("-", "_"), ckp)
        barrier()
        starts = model.assigned()
        if not args.silent and rank == 0:
            print("Starting root search", flush=True)
            model.set_progress(True)
        best, best_llh = model.search(args.min_roots, args.root_ratio, args.atol, args.bfgstol,
                                      args.brtol, args.factor)
        res = {"root_id": [int(best.id)] if starts else [], "llh": [best_llh], "alpha":
               [best.brlen_ratio], "best": best, "best_llh": best_llh}
    barrier()
    if rank != 0:
        return 0
    if ckp is not None:
        # rank 0 reads everybody's results (and an earlier run's) back from the
        # log, src/model.cpp:1237-1268
        done = ckp.current_progress()
        res = {"root_id": [r for r, _, _ in done], "llh": [l for _, l, _ in done],
               "alpha": [a for _, _, a in done]}
        k = max(range(len(done)), key=lambda i: done[i][1])    # first maximum, as std::max_element
        res["best_llh"] = done[k][1]
        res["best"] = tree.root_location(done[k][0]).with_ratio(done[k][2])
    if not res["root_id"]:
        raise SystemExit("no candidate root was evaluated")

    # likelihood weight ratios, src/model.cpp:1239-1258
    mx = max(res["llh"])
    total = sum(math.exp(l - mx) for l in res["llh"])
    out_tree = Tree.from_file(args.tree)
    for rid, llh, alpha in zip(res["root_id"], res["llh"], res["alpha"]):
        rl = out_tree.root_location(rid).with_ratio(float(alpha))
        out_tree.annotate_branch(rl, "LWR", "%f" % (math.exp(llh - mx) / total))
        out_tree.annotate_branch(rl, "LLH", "%f" % llh)
        out_tree.annotate_branch(rl, "alpha", "%f" % alpha, "%f" % (1 - alpha))
    best = res["best"]
    best_rl = out_tree.root_location(int(best.id)).with_ratio(best.brlen_ratio)
    if not args.exhaustive:
        # search mode writes the rooted tree only (src/main.cpp:600-610)
        out_tree.root_by(best_rl)
        rooted_newick = out_tree.newick(False)
        with open(prefix + ".rooted.tree", "w") as f:
            f.write(rooted_newick)
        if not args.silent:
            print("Final LogLH: %.5f" % res["best_llh"])
        print(rooted_newick)
        if not args.silent:
            print("Inference took: %.3fs" % (time.time() - t0))
        return 0
    # virtual_rooted_tree(final_rl).newick(): rooted there, then unrooted again
    out_tree.root_by(best_rl)
    out_tree.unroot()
    lwr_newick = out_tree.newick(True)
    out_tree.root_by(best_rl)
    rooted_newick = out_tree.newick(False)
    with open(prefix + ".lwr.tree", "w") as f:
        f.write(lwr_newick)
    with open(prefix + ".rooted.tree", "w") as f:
        f.write(rooted_newick)
    if not args.silent:
        print("Final LogLH: %.5f" % res["best_llh"])
    print(lwr_newick)
    if not args.silent:
        print("Inference took: %.3fs" % (time.time() - t0))
    return 0


if __name__ == "__main__":
    sys.exit(main())
