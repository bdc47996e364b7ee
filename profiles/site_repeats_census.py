#!/usr/bin/env python3
"""Subtree site-repeat census (the reference turns coraxlib's
CORAX_ATTRIB_SITE_REPEATS on for 4-state data, src/model.cpp:145-149).

For a tree + alignment, rooted at every k-th candidate branch, count for every
inner node the number of DISTINCT site patterns of the tips below it (its
repeat classes): with repeats a CLV operation computes one entry per class
instead of one per site.  Prints, per data set:

  ratio      = sum over inner nodes of classes / (inner nodes x patterns)
               -> the fraction of CLV arithmetic / CLV bytes that survives
  bytes_mat  = algorithmic bytes of one materialising traversal WITH repeats:
               per (node, class) one CLV record written and -- once as a child --
               read (R*K*8 each), plus per (node, site) a 4-byte class index for
               the gather at the parent and at the root reduction
  t_hbm      = bytes_mat / 8 TB/s, next to the measured times of the two
               existing paths on the c2 shape (materialising 175 us, fused 24 us
               per evaluation)

No GPU needed.  Usage:
  python profiles/site_repeats_census.py            # c2 synthetic + the reference's fixtures
The committed output is profiles/r2_site_repeats_census.txt."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import root_digger_amd as rd        # noqa: E402  (tree code only; no device call)
from root_digger_amd import synth   # noqa: E402
import util                         # noqa: E402


def census(tree, seqs, R=4, K=4, root_step=None):
    names = list(seqs)
    S = len(seqs[names[0]])
    n = tree.tip_count()
    tip_codes = {}
    for label in names:
        tip_codes[tree.tip_index(label)] = np.frombuffer(seqs[label].upper().encode(), dtype=np.uint8).astype(np.int64)
    nroots = tree.root_count()
    step = root_step or max(1, nroots // 8)
    ratios, bytes_mat = [], []
    for rid in range(0, nroots, step):
        ops, _, _ = tree.generate_operations(tree.root_location(rid))
        cls = dict(tip_codes)
        total_classes = 0
        for op in ops:
            a, b = cls[op.child1_clv_index], cls[op.child2_clv_index]
            key = a * (int(b.max()) + 1) + b
            uniq, inv = np.unique(key, return_inverse=True)
            cls[op.parent_clv_index] = inv.astype(np.int64)
            total_classes += len(uniq)
        inner = len(ops)
        ratios.append(total_classes / (inner * S))
        rec = R * K * 8
        # write each class record once, read it once as a child (root: by the reduction);
        # 4-byte class index per (node, site) written once and read once; tip codes 1 B
        bytes_mat.append(total_classes * rec * 2 + inner * S * 4 * 2 + n * S)
    return {"patterns": S, "taxa": n, "inner_nodes": inner, "roots_sampled": len(ratios),
            "ratio_mean": float(np.mean(ratios)), "ratio_min": float(np.min(ratios)),
            "ratio_max": float(np.max(ratios)), "bytes_mat": float(np.mean(bytes_mat))}


def report(name, tree, seqs, R=4):
    seqs, w = util.compress(seqs)
    c = census(tree, seqs, R=R)
    S, n = c["patterns"], c["taxa"]
    W = S * R * 4 * 8
    plain = (2 * n - 3) * W + n * S + (2 * n - 3) * 4 * S
    print("%-34s taxa %4d  patterns %6d  class ratio %.3f (min %.3f max %.3f over %d rootings)"
          % (name, n, S, c["ratio_mean"], c["ratio_min"], c["ratio_max"], c["roots_sampled"]))
    print("%-34s materialising traversal: %.1f MB plain -> %.1f MB with repeats (x%.2f); "
          "at 8 TB/s %.1f us -> %.1f us"
          % ("", plain / 1e6, c["bytes_mat"] / 1e6, c["bytes_mat"] / plain,
             plain / 8e12 * 1e6, c["bytes_mat"] / 8e12 * 1e6))
    return c


def main():
    print("# subtree site-repeat census (profiles/site_repeats_census.py)")
    w = synth.workload(100, 50000, 4, 4, 0xD166E5 + 1)
    c2 = report("c2 synthetic (bench.py workload)", rd.Tree.from_newick(w["newick"]), w["seqs"])
    w = synth.workload(100, 50000, 4, 4, 0xD166E5 + 1, simulate_seqs=False)
    report("c2 shape, i.i.d. uniform tips", rd.Tree.from_newick(w["newick"]), w["seqs"])
    ref = "/root/reference/test/data"
    for phy, tr in (("101.phy", "101.tree"), ("125.phy", "125.tree")):
        p = os.path.join(ref, "dna", phy)
        if not os.path.exists(p):
            p = os.path.join(util.DATA, phy)
        t = os.path.join(ref, "tree", tr)
        if not os.path.exists(t):
            t = os.path.join(util.DATA, tr)
        if os.path.exists(p) and os.path.exists(t):
            report("reference fixture " + phy, rd.Tree.from_file(t), util.read_phylip(p))
    print("# measured on MI355X, c2 shape (DESIGN 4.1/4.2): materialising traversal 175 us, "
          "fused evaluator 24 us per evaluation (no CLV traffic at all)")
    sav = c2["bytes_mat"] / 8e12 * 1e6
    print("# => with repeats the materialising path's HBM floor on c2 is %.0f us; the fused "
          "evaluator, which cannot use repeats (it stores no CLV), is already below it." % sav)


if __name__ == "__main__":
    main()
