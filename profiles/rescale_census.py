#!/usr/bin/env python3
"""How far down do the CLVs of a bench workload go?  The CPU oracle (reference rule: one rescale count per
site) walks a few jobs of c2 (or of a shard shape's tree) and prints, for the ROOT's vector of every (site, rate)
pair, the smallest magnitude (log2) and how many pairs lie below 2^-255 -- where the fused evaluator's
per-(site, rate) rule has rescaled on the way up -- and how many SITES the reference's per-site rule rescaled.
The input of profiles/r6_speculative_rescale.md.   usage: rescale_census.py c2 [sites] | c5s | c4s   (no GPU)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import root_digger_amd as rd
from root_digger_amd import synth
from oracle_lib import OraclePartition, ORC_MAP_NT
cfgs = {"c2": (100, 50000, 4, 4, 1), "c5s": (1000, 6000, 4, 4, 4), "c4s": (500, 6000, 4, 4, 3)}
name = sys.argv[1]
n, S, K, R, ci = cfgs[name]
if name == "c2": S = int(sys.argv[2]) if len(sys.argv) > 2 else S
seed = 0xD166E5 + ci
w = synth.workload(n, S, K, R, seed)
tree = rd.Tree.from_newick(w["newick"])
o = OraclePartition.for_tree(tree, K, S, R)
for label, seq in w["seqs"].items():
    o.set_tip_states(tree.tip_index(label), ORC_MAP_NT, seq)
o.set_frequencies(0, o.empirical_frequencies())
o.set_category_rates(w["rates"])
rng = np.random.default_rng(seed + 1000)
for j in range(6):
    params = synth.random_params(12, rng)
    rl = tree.root_location(j * 31 % tree.root_count())
    o.set_subst_params(0, params)
    ops, pmi, brl = tree.generate_operations(rl)
    o.update_prob_matrices(pmi, brl)
    o.update_clvs(OraclePartition.pack_ops(ops))
    clv = np.asarray(o.get_clv(tree.root_clv_index())).reshape(S, R, K)
    sc = np.asarray(o.get_scaler(tree.root_scaler_index()))
    m = clv.max(axis=2)   # per (site, rate)
    # true magnitude = m * 2^(-256 sc)
    log2m = np.log2(np.maximum(m, 1e-300)) - 256.0 * sc[:, None]
    print(name, "job", j, "sites with scaler>0:", int((sc > 0).sum()), "min log2 max-entry per (site,rate):", log2m.min(),
          "pairs below 2^-255:", int((log2m < -255).sum()), "of", S * R,
          "| smallest SITE (its best rate):", log2m.max(axis=1).min(), "sites below 2^-900:", int((log2m.max(axis=1) < -900).sum()))
