"""Seeded synthetic workloads: random binary unrooted
trees by sequential random edge insertion, branch lengths Exp(mean 0.1) clamped
to [1e-6, 1], sequences simulated down the tree under the evaluation model.
Bench/test utility only: nothing here is on the likelihood path."""
import numpy as np

DNA = "ACGT"
AA = "ARNDCQEGHILKMFPSTWYV"


class _Node:
    __slots__ = ("name", "children", "length")

    def __init__(self, name=None, length=0.0):
        self.name, self.children, self.length = name, [], length


def random_tree(n_tips, rng, mean=0.1):
    """-> (newick string, root _Node with 3 children)."""
    def brlen():
        return float(min(1.0, max(1e-6, rng.exponential(mean))))

    root = _Node()
    root.children = [_Node("t%04d" % i, brlen()) for i in range(3)]
    edges = [(root, c) for c in root.children]          # (parent, child)
    for i in range(3, n_tips):
        k = int(rng.integers(len(edges)))
        parent, child = edges[k]
        mid = _Node(None, child.length * 0.5)
        child.length *= 0.5
        if child.length < 1e-6:
            child.length = mid.length = 1e-6
        tip = _Node("t%04d" % i, brlen())
        mid.children = [child, tip]
        parent.children[parent.children.index(child)] = mid
        edges[k] = (parent, mid)
        edges.append((mid, child))
        edges.append((mid, tip))

    def nw(node):
        if not node.children:
            return "%s:%.8f" % (node.name, node.length)
        return "(" + ",".join(nw(c) for c in node.children) + "):%.8f" % node.length

    text = "(" + ",".join(nw(c) for c in root.children) + ");"
    return text, root


def build_q(subst, freqs):
    k = len(freqs)
    q = np.zeros((k, k))
    it = iter(subst)
    for i in range(k):
        for j in range(k):
            if i != j:
                q[i, j] = next(it) * freqs[j]
        q[i, i] = -q[i].sum()
    return q / -(np.asarray(freqs) * np.diag(q)).sum()


def _expm(a):
    n = max(0, int(np.ceil(np.log2(max(np.abs(a).sum(axis=0).max(), 1e-300) / 0.25))))
    x = a / (2.0 ** n)
    out = np.eye(a.shape[0])
    term = np.eye(a.shape[0])
    for k in range(1, 20):
        term = term @ x / k
        out = out + term
    for _ in range(n):
        out = out @ out
    return out


def simulate(root, n_sites, subst, freqs, rates, rng, alphabet=DNA):
    """Evolve n_sites down the tree; each site draws one rate category."""
    k = len(freqs)
    q = build_q(subst, freqs)
    cat = rng.integers(len(rates), size=n_sites)
    state0 = rng.choice(k, size=n_sites, p=np.asarray(freqs) / np.sum(freqs))
    seqs = {}
    letters = np.frombuffer(alphabet.encode(), dtype=np.uint8)

    def down(node, states):
        if node is not root:
            # one CDF table per (rate category, parent state); every site looks up
            # its own row and counts the thresholds its uniform draw exceeds
            cdf = np.stack([np.cumsum(_expm(q * rate * node.length), axis=1) for rate in rates])
            cdf[:, :, -1] = 1.0
            u = rng.random(n_sites)
            states = np.minimum((u[:, None] > cdf[cat, states]).sum(axis=1), k - 1)
        if not node.children:
            seqs[node.name] = letters[states].tobytes().decode()
        for c in node.children:
            down(c, states)

    import sys
    sys.setrecursionlimit(max(10000, sys.getrecursionlimit()))
    down(root, state0)
    return seqs


def random_params(size, rng):
    """random_params (src/model.cpp:87-93): U(1e-4, 1)."""
    return rng.uniform(1e-4, 1.0, size)


def workload(n_tips, n_sites, states, rate_cats, seed, gamma_alpha=1.0, simulate_seqs=True):
    """One BASELINE config: newick, sequences, model parameters."""
    from .api import compute_gamma_cats
    rng = np.random.default_rng(seed)
    newick, root = random_tree(n_tips, rng)
    subst = random_params(states * states - states, rng)
    alphabet = DNA if states == 4 else AA[:states]
    rates = compute_gamma_cats(gamma_alpha, rate_cats) if rate_cats > 1 else [1.0]
    sim_freqs = rng.dirichlet(np.ones(states) * 10)
    if simulate_seqs:
        seqs = simulate(root, n_sites, subst, sim_freqs, rates, rng, alphabet)
    else:
        letters = np.frombuffer(alphabet.encode(), dtype=np.uint8)
        seqs = {}

        def tips(node):
            if not node.children:
                seqs[node.name] = letters[rng.integers(states, size=n_sites)].tobytes().decode()
            for c in node.children:
                tips(c)
        tips(root)
    return {"newick": newick, "seqs": seqs, "subst": subst, "rates": rates,
            "alphabet": alphabet, "states": states, "rate_cats": rate_cats}
