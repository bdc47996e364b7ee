"""Shared helpers for the parity tests: fixture readers, model set-up and the
compute_lh / compute_lh_root call sequences of the reference
(src/model.cpp:384-452) expressed over the partition API, so the same code
drives the oracle and the HIP library."""
import ctypes as C
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(GOLD, "data")


def golden(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def read_fasta(path):
    seqs, name = {}, None
    for line in open(path):
        line = line.strip()
        if not line:
            continue
        if line.startswith(">"):
            name = line[1:].split()[0]
            seqs[name] = ""
        else:
            seqs[name] += line
    return seqs


def read_phylip(path):
    toks = open(path).read().split()
    n, length = int(toks[0]), int(toks[1])
    seqs, i = {}, 2
    for _ in range(n):
        name = toks[i]
        i += 1
        s = ""
        while len(s) < length:
            s += toks[i]
            i += 1
        seqs[name] = s
    return seqs


def compress(seqs):
    """site-pattern compression -> (compressed seqs, weights); order = first seen."""
    names = list(seqs)
    cols = {}
    order = []
    n = len(seqs[names[0]])
    for s in range(n):
        col = "".join(seqs[k][s] for k in names)
        if col not in cols:
            cols[col] = 0
            order.append(col)
        cols[col] += 1
    out = {k: "".join(col[i] for col in order) for i, k in enumerate(names)}
    return out, np.array([cols[c] for c in order], dtype=np.uint32)


def make_map(alphabet, extra=None):
    m = (C.c_uint64 * 256)()
    for i, ch in enumerate(alphabet):
        m[ord(ch)] = 1 << i
    for ch, v in (extra or {}).items():
        m[ord(ch)] = v
    return m


def load_tips(part, tree, seqs, cmap, weights=None):
    """model_t::set_tip_states (src/model.cpp:302-325)."""
    for label, seq in seqs.items():
        idx = tree.tip_index(label)
        assert idx >= 0, label
        part.set_tip_states(idx, cmap, seq)
    if weights is not None:
        part.set_pattern_weights(weights)


def compute_lh(part, tree, rl):
    """model_t::compute_lh (src/model.cpp:384-413) for one partition."""
    ops, pmi, brl = tree.generate_operations(rl)
    part.update_prob_matrices(pmi, brl)
    part.update_clvs(ops)
    return part.compute_root_loglikelihood(tree.root_clv_index(), tree.root_scaler_index())


def compute_lh_root(part, tree, rl):
    """model_t::compute_lh_root (src/model.cpp:415-452)."""
    op, pmi, brl = tree.generate_derivative_operations(rl)
    part.update_prob_matrices(pmi, brl)
    part.update_clvs([op])
    return part.compute_root_loglikelihood(tree.root_clv_index(), tree.root_scaler_index())


def move_root(part, tree, rl):
    """model_t::move_root (src/model.cpp:823-854)."""
    ops, pmi, brl = tree.generate_root_update_operations(rl)
    if len(ops) == 0:
        return
    part.update_prob_matrices(pmi, brl)
    part.update_clvs(ops)


def find_root(tree, near_tips, far_tips, alpha):
    """Root location whose branch splits the tips as the golden entry says;
    alpha is re-expressed relative to the side rl.edge names."""
    near, far = sorted(near_tips), sorted(far_tips)
    for rl in tree.roots():
        side = sorted(tree.side_tips(rl))
        if side == near:
            return rl.with_ratio(alpha)
        if side == far:
            return rl.with_ratio(1.0 - alpha)
    raise KeyError("no root location matches the golden split")


def rel_err(a, b):
    return abs(a - b) / max(abs(a), abs(b), 1e-300)
