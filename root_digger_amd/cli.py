#!/usr/bin/env python3
"""Exhaustive root search from the command line (the `rd --msa M --tree T
--exhaustive` entry of the reference, /root/reference/src/main.cpp:411-680,
without its MPI / checkpoint control plane):

  python -m root_digger_amd.cli --msa aln.fasta --tree t.nwk --prefix out \\
         [--rate-cats 4] [--lbfgsb /path/to/liblbfgsb.so] [--early-stop]

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
import sys
import time

from . import Model, Tree, set_device


def main(argv=None):
    ap = argparse.ArgumentParser(prog="root_digger_amd.cli")
    ap.add_argument("--msa", required=True)
    ap.add_argument("--tree", required=True)
    ap.add_argument("--prefix", default=None)
    ap.add_argument("--rate-cats", type=int, default=1)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--atol", type=float, default=1e-7)       # abs_tolerance
    ap.add_argument("--bfgstol", type=float, default=1e-7)
    ap.add_argument("--brtol", type=float, default=1e-12)
    ap.add_argument("--factor", type=float, default=1e4)
    ap.add_argument("--early-stop", action="store_true")
    ap.add_argument("--lbfgsb", default=None,
                    help="shared library exporting the L-BFGS-B entry point `setulb`")
    ap.add_argument("--workers", type=int, default=4,
                    help="host threads, each with its own model replica / HIP stream "
                         "(0 = the plain sequential loop)")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--silent", action="store_true")
    args = ap.parse_args(argv)
    prefix = args.prefix or args.msa

    t0 = time.time()
    set_device(args.device)
    tree = Tree.from_file(args.tree)
    model = Model.from_file(tree, args.msa, rate_cats=args.rate_cats, seed=args.seed,
                            early_stop=args.early_stop)
    model.initialize_partitions()
    keep = None
    if args.lbfgsb:
        keep = ctypes.CDLL(args.lbfgsb)
        model.set_lbfgsb(keep.setulb)
    model.compute_lh(tree.root_location(0))                    # model.initialize()
    res = model.exhaustive_search(args.atol, args.bfgstol, args.brtol, args.factor,
                                  workers=args.workers)

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
