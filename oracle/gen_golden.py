#!/usr/bin/env python3
"""Golden-vector generator (test infrastructure; runs in the build container).

An INDEPENDENT SciPy/NumPy Felsenstein pruning used to pin oracle/rd_oracle.c
and the HIP path.  It deliberately shares no code or technique with either:
  * P(t) comes from scipy.linalg.expm (Pade), not the oracle's Taylor series;
  * underflow is handled by per-node max-normalisation in log space, not by
    the 2^256 per-site scaler rule;
  * root placements are keyed by tip bipartition, not by the reference's
    unode ids, so no index convention is baked in.
The reference (/root/reference) is C++ that cannot be built here (coraxlib
submodule empty), so these vectors come from this script, not from the
reference binary: absolute-lnL parity with coraxlib stays "unpinned".

Inputs: the reference's own fixtures copied to tests/golden/data/
(test/data/dna/{single.phy,10.fasta,101.phy}, test/data/tree/*.tree; MIT).
Parameter vectors: test/src/model.cpp:12-17.

Usage: python oracle/gen_golden.py   (writes tests/golden/*.json)
"""
import json
import os
import sys

import numpy as np
from scipy.linalg import expm
from scipy.special import gammainc, gammaincinv

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "..", "tests", "golden")
DATA = os.path.join(GOLD, "data")

NT = {}
for ch, v in dict(A=1, C=2, G=4, T=8, U=8, R=5, Y=10, S=6, W=9, K=12, M=3, B=14,
                  D=13, H=11, V=7, N=15, O=15, X=15).items():
    NT[ch] = v
    NT[ch.lower()] = v
NT["-"] = 15
NT["?"] = 15


# ----------------------------------------------------------------- parsing
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
    seqs = {}
    i = 2
    for _ in range(n):
        name = toks[i]
        i += 1
        s = ""
        while len(s) < length:
            s += toks[i]
            i += 1
        seqs[name] = s
    return seqs


def parse_newick(text):
    """-> nested (label, length, [children]) tuples."""
    text = text.strip()
    pos = [0]

    def node():
        children = []
        if text[pos[0]] == "(":
            pos[0] += 1
            while True:
                children.append(node())
                if text[pos[0]] == ",":
                    pos[0] += 1
                    while text[pos[0]].isspace():
                        pos[0] += 1
                    continue
                assert text[pos[0]] == ")"
                pos[0] += 1
                break
        j = pos[0]
        while text[j] not in ":,();":
            j += 1
        label = text[pos[0]:j].strip()
        pos[0] = j
        length = 0.0
        if text[pos[0]] == ":":
            j = pos[0] + 1
            while text[j] not in ",();":
                j += 1
            length = float(text[pos[0] + 1:j])
            pos[0] = j
        return (label, length, children)

    return node()


class UTree:
    """Unrooted binary tree as an undirected graph with edge lengths."""

    def __init__(self, newick):
        root = parse_newick(newick)
        self.adj = {}      # node id -> {nbr: length}
        self.label = {}
        self._n = 0

        def add(nd):
            i = self._n
            self._n += 1
            self.adj[i] = {}
            self.label[i] = nd[0]
            for ch in nd[2]:
                c = add(ch)
                self.adj[i][c] = ch[1]
                self.adj[c][i] = ch[1]
            return i

        r = add(root)
        if len(self.adj[r]) == 2:          # binary root -> unroot (sum lengths)
            (a, la), (b, lb) = self.adj[r].items()
            del self.adj[a][r], self.adj[b][r], self.adj[r]
            self.adj[a][b] = la + lb
            self.adj[b][a] = la + lb
        self.tips = [i for i in self.adj if len(self.adj[i]) == 1]

    def edges(self):
        return [(u, v) for u in self.adj for v in self.adj[u] if u < v]

    def side_tips(self, u, v):
        """tip labels reachable from u without crossing edge (u,v)."""
        out, stack, seen = [], [u], {u, v}
        while stack:
            x = stack.pop()
            if len(self.adj[x]) == 1:
                out.append(self.label[x])
            for y in self.adj[x]:
                if y not in seen:
                    seen.add(y)
                    stack.append(y)
        if len(self.adj[u]) == 1 and self.label[u] not in out:
            out.append(self.label[u])
        return sorted(out)


# ------------------------------------------------------------------ model
def build_q(subst, freqs):
    """SURVEY Appendix A1 convention (the one UNVERIFIED choice)."""
    k = len(freqs)
    q = np.zeros((k, k))
    it = iter(subst)
    for i in range(k):
        for j in range(k):
            if i != j:
                q[i, j] = next(it) * freqs[j]
        q[i, i] = -q[i].sum()
    q /= -(np.asarray(freqs) * np.diag(q)).sum()
    return q


def gamma_cats(alpha, cats, mode):
    if cats == 1:
        return [1.0]
    if mode == "median":
        r = np.array([gammaincinv(alpha, (2 * i + 1) / (2.0 * cats)) / alpha
                      for i in range(cats)])
        return list(r * cats / r.sum())
    cut = np.array([gammaincinv(alpha, (i + 1.0) / cats) for i in range(cats - 1)])
    g = gammainc(alpha + 1.0, cut)
    g = np.concatenate([[0.0], g, [1.0]])
    return list((g[1:] - g[:-1]) * cats)


def tip_clv(seq, k, cmap):
    codes = np.array([cmap[c] for c in seq], dtype=np.int64)
    return ((codes[:, None] >> np.arange(k)[None, :]) & 1).astype(float)


def prune(tree, seqs, cmap, q, rates, weights, freqs, u, v, alpha,
          want_nodes=False):
    """lnL with a virtual root on edge (u,v): u at alpha*L, v at (1-alpha)*L."""
    k = q.shape[0]
    L = tree.adj[u][v]
    rates = np.asarray(rates)
    pcache = {}

    def pmats(t):
        if t not in pcache:
            pcache[t] = np.stack([expm(q * r * t) for r in rates])
        return pcache[t]

    sys.setrecursionlimit(10000)
    node_out = {}

    def down(x, parent):
        """-> (clv[S,R,K] normalised, logscale[S])"""
        if len(tree.adj[x]) == 1:
            c = tip_clv(seqs[tree.label[x]], k, cmap)
            res = (np.repeat(c[:, None, :], len(rates), axis=1),
                   np.zeros(c.shape[0]))
        else:
            acc, ls = None, 0.0
            for y, t in tree.adj[x].items():
                if y == parent:
                    continue
                cy, ly = down(y, x)
                term = np.einsum("rij,srj->sri", pmats(t), cy)
                acc = term if acc is None else acc * term
                ls = ls + ly
            m = acc.max(axis=(1, 2))
            res = (acc / m[:, None, None], ls + np.log(m))
        if want_nodes:
            node_out[(x, parent)] = res
        return res

    cu, lu = down(u, v)
    cv, lv = down(v, u)
    root = (np.einsum("rij,srj->sri", pmats(L * alpha), cu)
            * np.einsum("rij,srj->sri", pmats(L * (1 - alpha)), cv))
    site = np.einsum("srk,k,r->s", root, np.asarray(freqs), np.asarray(weights))
    persite = np.log(site) + lu + lv
    if want_nodes:
        return persite, node_out
    return persite


def empirical_freqs(seqs, k, cmap):
    f = np.zeros(k)
    n = 0
    for s in seqs.values():
        c = tip_clv(s, k, cmap)
        f += (c / c.sum(axis=1, keepdims=True)).sum(axis=0)
        n += c.shape[0]
    return list(f / n)


PARAMS = [  # test/src/model.cpp:12-17
    [1, 2.5, 1, 1, 1, 2.5, 2.5, 1, 1, 1, 2.5, 1],
    [1.0] * 12,
    [1.0, 1.2, 1.3, 1.2, 1.0, 1.3, 1.2, 1.1, 1.1, 1.4, 1.0, 1.0],
    [.34, .42, .24, .74, .16, .88, .75, .54, .20, .06, .08, .41],
]


def root_sweep(tree, seqs, q, rates, weights, freqs, alpha):
    out = []
    for (u, v) in tree.edges():
        ps = prune(tree, seqs, NT, q, rates, weights, freqs, u, v, alpha)
        out.append({"near_tips": tree.side_tips(u, v),
                    "far_tips": tree.side_tips(v, u),
                    "alpha": alpha, "lnl": float(ps.sum())})
    return out


def main():
    os.makedirs(GOLD, exist_ok=True)
    rng = np.random.default_rng(20261002)

    # 1. expm vectors ------------------------------------------------------
    cases = []
    for k in (2, 4, 20):
        for _ in range(4 if k < 20 else 1):
            f = rng.dirichlet(np.ones(k) * 5)
            s = rng.uniform(1e-4, 1.0, k * k - k)
            q = build_q(s, f)
            for t in (0.0, 1e-6, 0.01, 0.37, 1.0, 7.5):
                cases.append({"k": k, "subst": list(s), "freqs": list(f), "t": t,
                              "q": q.ravel().tolist(),
                              "p": expm(q * t).ravel().tolist()})
    json.dump(cases, open(os.path.join(GOLD, "expm.json"), "w"))

    # 2. discrete gamma ----------------------------------------------------
    g = []
    for a in (0.2, 0.5, 1.0, 2.0, 10.0, 99.0):
        for cats in (1, 2, 4, 8):
            for mode in ("mean", "median"):
                g.append({"alpha": a, "cats": cats, "mode": mode,
                          "rates": gamma_cats(a, cats, mode)})
    json.dump(g, open(os.path.join(GOLD, "gamma.json"), "w"))

    # 3. single.phy: JC69 at all 5 roots x alphas ---------------------------
    tree = UTree(open(os.path.join(DATA, "single.tree")).read())
    seqs = read_phylip(os.path.join(DATA, "single.phy"))
    q = build_q([1.0] * 12, [0.25] * 4)
    single = []
    for a in (0.0, 0.25, 0.5, 0.75, 1.0):
        single += root_sweep(tree, seqs, q, [1.0], [1.0], [0.25] * 4, a)
    json.dump(single, open(os.path.join(GOLD, "single_jc.json"), "w"))

    # 4. 10.fasta (= BASELINE config c1) ------------------------------------
    tree = UTree(open(os.path.join(DATA, "10.tree")).read())
    seqs = read_fasta(os.path.join(DATA, "10.fasta"))
    emp = empirical_freqs(seqs, 4, NT)
    ten = {"empirical_freqs": emp, "cases": []}
    for pi, sp in enumerate(PARAMS):
        for fname, fr in (("uniform", [0.25] * 4), ("empirical", emp)):
            for cats in (1, 4):
                rates = gamma_cats(1.0, cats, "mean")
                w = [1.0 / cats] * cats
                q = build_q(sp, fr)
                for a in ((0.5,) if cats == 4 else (0.5, 0.1)):
                    ten["cases"].append({
                        "subst": sp, "freqs": fr, "freqs_name": fname,
                        "rate_cats": cats, "rates": rates, "alpha": a,
                        "param_set": pi,
                        "roots": root_sweep(tree, seqs, q, rates, w, fr, a)})
    json.dump(ten, open(os.path.join(GOLD, "ten_fasta.json"), "w"))

    # 5. 101.phy: ambiguity codes + zero-length branches --------------------
    tree = UTree(open(os.path.join(DATA, "101.tree")).read())
    seqs = read_phylip(os.path.join(DATA, "101.phy"))
    emp = empirical_freqs(seqs, 4, NT)
    edges = tree.edges()
    pick = [edges[i] for i in (0, 7, 50, 123, len(edges) - 1)]
    h = {"empirical_freqs": emp, "cases": []}
    for name, sp, fr, cats in (("jc", [1.0] * 12, [0.25] * 4, 1),
                               ("unrest", PARAMS[3], emp, 4)):
        rates = gamma_cats(0.7, cats, "mean") if cats > 1 else [1.0]
        w = [1.0 / cats] * cats
        q = build_q(sp, fr)
        roots = []
        for (u, v) in pick:
            ps = prune(tree, seqs, NT, q, rates, w, fr, u, v, 0.5)
            roots.append({"near_tips": tree.side_tips(u, v),
                          "far_tips": tree.side_tips(v, u),
                          "alpha": 0.5, "lnl": float(ps.sum())})
        h["cases"].append({"name": name, "subst": sp, "freqs": fr,
                           "rate_cats": cats, "rates": rates, "gamma_alpha": 0.7,
                           "roots": roots})
    json.dump(h, open(os.path.join(GOLD, "hundred_one.json"), "w"))

    # 6. deep caterpillar: forces the 2^256 scaler rule ----------------------
    ntips = 160
    names = ["t%03d" % i for i in range(ntips)]
    nw = names[0]
    for i in range(1, ntips - 1):
        nw = "(%s:0.9,%s:0.7)" % (nw, names[i])
    nw = "(%s:0.9,%s:0.7,%s:0.8);" % (nw, names[ntips - 1], "tx")
    names.append("tx")
    nsite = 24
    seqs = {n: "".join(rng.choice(list("ACGT"), nsite)) for n in names}
    tree = UTree(nw)
    sp = PARAMS[3]
    fr = [0.1, 0.2, 0.3, 0.4]
    rates = gamma_cats(0.5, 4, "mean")
    q = build_q(sp, fr)
    u, v = tree.edges()[0]
    persite, nodes = prune(tree, seqs, NT, q, rates, [0.25] * 4, fr, u, v, 0.5,
                           want_nodes=True)
    # log of the TRUE (unscaled) conditional likelihoods for a few deep nodes
    deep = []
    for (x, parent), (clv, ls) in nodes.items():
        ntip = len(tree.side_tips(x, parent))
        if ntip in (2, 40, 100, 150):
            with np.errstate(divide="ignore"):
                deep.append({"tips": tree.side_tips(x, parent),
                             "log_clv": (np.log(clv) + ls[:, None, None]).tolist()})
    json.dump({"newick": nw, "seqs": seqs, "subst": sp, "freqs": fr,
               "rates": rates, "alpha": 0.5,
               "near_tips": tree.side_tips(u, v),
               "lnl": float(persite.sum()), "persite": persite.tolist(),
               "nodes": deep},
              open(os.path.join(GOLD, "deep_scaling.json"), "w"))

    # 7. 20-state synthetic (BASELINE config c3 shape, tiny) -----------------
    k = 20
    ntips = 12
    names = ["p%02d" % i for i in range(ntips)]
    order = list(names)
    sub = order[:3]
    nodes_nw = ["%s:%.4f" % (n, rng.uniform(0.02, 0.6)) for n in sub]
    # random stepwise addition in newick text form
    clades = nodes_nw
    for n in order[3:]:
        i = rng.integers(len(clades))
        clades[i] = "(%s,%s:%.4f):%.4f" % (clades[i], n, rng.uniform(0.02, 0.6),
                                          rng.uniform(0.02, 0.6))
    nw = "(" + ",".join(clades) + ");"
    aa = "ARNDCQEGHILKMFPSTWYV"
    amap = {c: 1 << i for i, c in enumerate(aa)}
    amap["X"] = (1 << 20) - 1
    amap["B"] = amap["N"] | amap["D"]
    nsite = 40
    seqs = {n: "".join(rng.choice(list(aa + "XB"), nsite)) for n in names}
    tree = UTree(nw)
    sp = list(rng.uniform(1e-4, 1.0, k * k - k))
    fr = list(rng.dirichlet(np.ones(k) * 8))
    rates = gamma_cats(1.0, 4, "mean")
    q = build_q(sp, fr)
    roots = []
    for (u, v) in tree.edges():
        ps = prune(tree, seqs, amap, q, rates, [0.25] * 4, fr, u, v, 0.3)
        roots.append({"near_tips": tree.side_tips(u, v),
                      "far_tips": tree.side_tips(v, u), "alpha": 0.3,
                      "lnl": float(ps.sum())})
    json.dump({"newick": nw, "seqs": seqs, "alphabet": aa, "subst": sp,
               "freqs": fr, "rates": rates, "roots": roots},
              open(os.path.join(GOLD, "protein20.json"), "w"))
    # 8. binary characters (`rd --states 2`): 2 substitution rates, 2 frequencies ---
    k = 2
    names = ["b%02d" % i for i in range(14)]
    clades = ["%s:%.4f" % (n, rng.uniform(0.02, 0.6)) for n in names[:3]]
    for n in names[3:]:
        i = rng.integers(len(clades))
        clades[i] = "(%s,%s:%.4f):%.4f" % (clades[i], n, rng.uniform(0.02, 0.6),
                                          rng.uniform(0.02, 0.6))
    nw = "(" + ",".join(clades) + ");"
    bmap = {"0": 1, "1": 2, "-": 3, "?": 3}
    seqs = {n: "".join(rng.choice(list("0011-?"), 60)) for n in names}
    tree = UTree(nw)
    sp = [0.7, 1.9]
    fr = [0.37, 0.63]
    rates = gamma_cats(0.8, 4, "mean")
    q = build_q(sp, fr)
    roots = []
    for (u, v) in tree.edges():
        ps = prune(tree, seqs, bmap, q, rates, [0.25] * 4, fr, u, v, 0.6)
        roots.append({"near_tips": tree.side_tips(u, v), "far_tips": tree.side_tips(v, u),
                      "alpha": 0.6, "lnl": float(ps.sum())})
    json.dump({"newick": nw, "seqs": seqs, "subst": sp, "freqs": fr, "rates": rates,
               "roots": roots}, open(os.path.join(GOLD, "binary2.json"), "w"))
    print("goldens written to", os.path.normpath(GOLD))


if __name__ == "__main__":
    main()
