#!/usr/bin/env python3
"""How many operations would fold into clade tables at a given class limit?  

For a tree + alignment, rooted at a few candidate branches: classes per inner node (distinct
patterns of the tips below it), and the operations that remain per evaluation when every
maximal subtree of nodes with <= L classes becomes a pseudo-tip (the root operation never
folds) -- for L = 16, 64, 256, 1024.  Two questions it answers:
what 256-row tables would buy the 4-state evaluator on c2 / c5 / 125.phy, and whether clade
tables could help the 20-state evaluator on c3 (a protein cherry has up to 400 classes).
No GPU needed (tree code only).  Output committed as profiles/r4_fold_census.txt."""
import lzma
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import root_digger_amd as rd        # noqa: E402
from root_digger_amd import synth   # noqa: E402
import util                         # noqa: E402

LIMITS = (16, 64, 256, 1024)


def fold(tree, seqs, samples=4):
    names = list(seqs)
    tips = {tree.tip_index(k): np.frombuffer(seqs[k].upper().encode(), dtype=np.uint8).astype(np.int64)
            for k in names}
    n = tree.tip_count()
    nroots = tree.root_count()
    left = {L: [] for L in LIMITS}
    cherry_classes = []
    for rid in range(0, nroots, max(1, nroots // samples))[:samples]:
        ops, _, _ = tree.generate_operations(tree.root_location(rid))
        cls, ncls = dict(tips), {}
        for op in ops:
            a, b = cls[op.child1_clv_index], cls[op.child2_clv_index]
            uniq, inv = np.unique(a * (int(b.max()) + 1) + b, return_inverse=True)
            cls[op.parent_clv_index] = inv.astype(np.int64)
            ncls[op.parent_clv_index] = len(uniq)
            if op.child1_clv_index < n and op.child2_clv_index < n:
                cherry_classes.append(len(uniq))
        for L in LIMITS:
            # an operation is dropped when its node is small (classes are monotone up the tree, so
            # small nodes form whole subtrees); the root operation is never folded
            kept = sum(1 for i, op in enumerate(ops) if i + 1 == len(ops) or ncls[op.parent_clv_index] > L)
            left[L].append(kept)
    return len(ops), {L: float(np.mean(v)) for L, v in left.items()}, cherry_classes


def report(name, tree, seqs):
    nops, left, ch = fold(tree, seqs)
    ch = np.array(ch)
    print("%-28s %4d operations; left after folding at class limit " % (name, nops) +
          ", ".join("%d: %.1f" % (L, left[L]) for L in LIMITS))
    print("%-28s cherries: median %d classes, %d %% within 64, %d %% within 256" % (
        "", int(np.median(ch)), round(100 * float(np.mean(ch <= 64))), round(100 * float(np.mean(ch <= 256)))))


if __name__ == "__main__":
    for cfg, (n, S, K, seed) in {"c2 (100 x 50 000, DNA)": (100, 50000, 4, 0xD166E5 + 1),
                                 "c5 shard (1000 x 50 000, DNA)": (1000, 50000, 4, 0xD166E5 + 4),
                                 "c3 (200 x 10 000, protein)": (200, 10000, 20, 0xD166E5 + 2)}.items():
        w = synth.workload(n, S, K, 4, seed)
        report(cfg, rd.Tree.from_newick(w["newick"]), w["seqs"])
    text = lzma.open(os.path.join(util.DATA, "125.phy.xz"), "rt").read().split()
    seqs = {text[2 + 2 * i]: text[3 + 2 * i] for i in range(int(text[0]))}
    seqs, _ = util.compress(seqs)
    report("125.phy (19 436 patterns)", rd.Tree.from_file(os.path.join(util.DATA, "125.tree")), seqs)
