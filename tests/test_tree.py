"""Host schedule logic (rooted_tree_t mirror) against the golden vectors the
reference's own tests hold (test/src/tree.cpp).  No GPU needed."""
import os

import pytest

import root_digger_amd as rd
import util

DATASETS = {
    "single": ("single.phy", "single.tree"),
    "101.phy": ("101.phy", "101.tree"),
    "10.fasta": ("10.fasta", "10.tree"),
}


def tree_of(key):
    return rd.Tree.from_file(os.path.join(util.DATA, DATASETS[key][1]))


@pytest.mark.parametrize("key", list(DATASETS))
def test_string_constructor(key):          # test/src/tree.cpp:18-28
    tree = tree_of(key)
    assert tree.root_count() == 2 * tree.tip_count() - 3
    for i, rl in enumerate(tree.roots()):
        assert rl.id == i
        assert rl.saved_brlen >= 0.0
    if key != "101.phy":                   # 101.tree has zero-length branches
        assert all(rl.saved_brlen > 0.0 for rl in tree.roots())


@pytest.mark.parametrize("key", list(DATASETS))
def test_two_constructions_consistent(key):    # test/src/tree.cpp:29-71
    t1, t2 = tree_of(key), tree_of(key)
    assert t1.root_count() == t2.root_count()
    assert t1.newick() == t2.newick()
    for i in range(t1.root_count()):
        a, b = t1.root_location(i), t2.root_location(i)
        assert (a.id, a.saved_brlen, t1.root_label(i)) == (b.id, b.saved_brlen, t2.root_label(i))


@pytest.mark.parametrize("key", list(DATASETS))
def test_generate_operations_all_roots(key):   # test/src/tree.cpp:118-140
    tree = tree_of(key)
    n = tree.tip_count()
    for rl in tree.roots():
        ops, pmi, brl = tree.generate_operations(rl)
        assert len(ops) == n - 1
        assert len(pmi) == len(brl) == 2 * n - 2
        assert sorted(pmi) == list(range(2 * n - 2))
        assert ops[len(ops) - 1].parent_clv_index == 2 * n - 2
        assert ops[len(ops) - 1].parent_scaler_index == n - 2
        # every op's children were produced earlier or are tips
        have = set(range(n))
        for op in ops:
            assert op.child1_clv_index in have and op.child2_clv_index in have
            have.add(op.parent_clv_index)


def test_known_tree_schedule():                # test/src/tree.cpp:142-180
    tree = tree_of("single")
    ops, pmi, brl = tree.generate_operations(tree.root_location("n2"))
    assert len(ops) == 3
    # (parent_clv, parent_sc, c1_clv, c1_mat, c1_sc, c2_clv, c2_mat, c2_sc)
    assert ops[0].astuple() == (4, 0, 0, 0, -1, 1, 1, -1)
    assert ops[1].astuple() == (5, 1, 2, 2, -1, 3, 3, -1)
    assert ops[2].astuple() == (6, 2, 4, 4, 0, 5, 5, 1)


def test_root_operation():                     # test/src/tree.cpp:182-212
    tree = tree_of("single")
    op, pmi, brl = tree.generate_derivative_operations(tree.root_location("n2"))
    assert op.astuple() == (6, 2, 4, 4, 0, 5, 5, 1)
    assert list(pmi) == [4, 5]
    assert list(brl) == [0.275, 0.275]


@pytest.mark.parametrize("key", list(DATASETS))
def test_root_unroot_cycle(key):               # test/src/tree.cpp:214-223
    tree = tree_of(key)
    before = tree.newick()
    for rl in tree.roots():
        tree.root_by(rl)
        assert tree.rooted()
        tree.unroot()
        assert not tree.rooted()
    # lengths are restored by unroot; topology is unchanged
    t2 = tree_of(key)
    assert sorted(r.saved_brlen for r in tree.roots()) == sorted(r.saved_brlen for r in t2.roots())
    assert len(before) == len(tree.newick())


NEWICK_GOLD = [   # test/src/tree.cpp:225-292
    ("b", 0.25, "(b:0.025000,((c:0.100000,d:0.100000)n2:0.550000,a:0.100000)n1:0.075000);"),
    ("b", 0.75, "(b:0.075000,((c:0.100000,d:0.100000)n2:0.550000,a:0.100000)n1:0.025000);"),
    ("a", 0.25, "(a:0.025000,(b:0.100000,(c:0.100000,d:0.100000)n2:0.550000)n1:0.075000);"),
    ("a", 0.75, "(a:0.075000,(b:0.100000,(c:0.100000,d:0.100000)n2:0.550000)n1:0.025000);"),
    ("n2", 0.25, "((c:0.100000,d:0.100000)n2:0.137500,(a:0.100000,b:0.100000)n1:0.412500);"),
    ("n2", 0.75, "((c:0.100000,d:0.100000)n2:0.412500,(a:0.100000,b:0.100000)n1:0.137500);"),
    ("c", 0.25, "(c:0.025000,(d:0.100000,(a:0.100000,b:0.100000)n1:0.550000)n2:0.075000);"),
    ("c", 0.75, "(c:0.075000,(d:0.100000,(a:0.100000,b:0.100000)n1:0.550000)n2:0.025000);"),
    ("d", 0.25, "(d:0.025000,((a:0.100000,b:0.100000)n1:0.550000,c:0.100000)n2:0.075000);"),
    ("d", 0.75, "(d:0.075000,((a:0.100000,b:0.100000)n1:0.550000,c:0.100000)n2:0.025000);"),
]


def test_newick_after_root_by():
    tree = tree_of("single")
    assert tree.root_count() == 5
    for label, ratio, want in NEWICK_GOLD:     # same tree object, in the reference's order
        tree.root_by(tree.root_location(label).with_ratio(ratio))
        assert tree.newick() == want


def test_derivative_vs_regular_ops():          # test/src/tree.cpp:298-334
    tree = tree_of("single")
    for rl in tree.roots():
        ops, _, _ = tree.generate_operations(rl)
        op, _, _ = tree.generate_derivative_operations(rl)
        assert op.astuple() == ops[len(ops) - 1].astuple()


def test_sanity_checks():                      # test/src/tree.cpp:336-345
    def t(name):
        return rd.Tree.from_file(os.path.join(util.DATA, name + ".tree"))
    assert not t("sanity_check1").sanity_check()
    assert not t("sanity_check2").sanity_check()
    assert t("sanity_check3").sanity_check()


ANNOT_BASIC = (   # test/src/tree.cpp:347-365
    "(((j:0.854700[&&NHX:foo=bar:fizz=buzz],((h:0.983500[&&NHX:foo=bar:fizz=buzz],a:0.224900"
    "[&&NHX:foo=bar:fizz=buzz]):0.416200[&&NHX:foo=bar:fizz=buzz],(c:0.540900[&&NHX:foo=bar:"
    "fizz=buzz],f:0.422200[&&NHX:foo=bar:fizz=buzz]):0.785300[&&NHX:foo=bar:fizz=buzz]):0."
    "614100[&&NHX:foo=bar:fizz=buzz]):0.446100[&&NHX:foo=bar:fizz=buzz],g:0.487400[&&NHX:foo="
    "bar:fizz=buzz]):0.825200[&&NHX:foo=bar:fizz=buzz],((i:0.569700[&&NHX:foo=bar:fizz=buzz],"
    "e:0.366600[&&NHX:foo=bar:fizz=buzz]):0.602800[&&NHX:foo=bar:fizz=buzz],b:0.445900[&&NHX:"
    "foo=bar:fizz=buzz]):0.099300[&&NHX:foo=bar:fizz=buzz],d:0.639600[&&NHX:foo=bar:fizz=buzz]);")

ANNOT_ALL_ROOTS = (   # test/src/tree.cpp:388-408
    "(a:0.224900[&&NHX:foo=bar:fizz=buzz],((c:0.540900[&&NHX:foo=bar:fizz=buzz],f:0.422200[&&"
    "NHX:foo=bar:fizz=buzz]):0.785300[&&NHX:foo=bar:fizz=buzz],((g:0.487400[&&NHX:foo=bar:fizz="
    "buzz],(((i:0.569700[&&NHX:foo=bar:fizz=buzz],e:0.366600[&&NHX:foo=bar:fizz=buzz]):0.602800"
    "[&&NHX:foo=bar:fizz=buzz],b:0.445900[&&NHX:foo=bar:fizz=buzz]):0.099300[&&NHX:foo=bar:fizz"
    "=buzz],d:0.639600[&&NHX:foo=bar:fizz=buzz]):0.825200[&&NHX:foo=bar:fizz=buzz]):0.446100[&&"
    "NHX:foo=bar:fizz=buzz],j:0.854700[&&NHX:foo=bar:fizz=buzz]):0.614100[&&NHX:foo=bar:fizz="
    "buzz]):0.416200[&&NHX:foo=bar:fizz=buzz],h:0.983500[&&NHX:foo=bar:fizz=buzz]);")


def test_annotations_basic():
    tree = tree_of("10.fasta")
    for rl in tree.roots():
        tree.annotate_branch(rl, "foo", "bar")
        tree.annotate_branch(rl, "fizz", "buzz")
    assert tree.newick() == ANNOT_BASIC


def test_annotations_all_roots():
    tree = tree_of("10.fasta")
    for rl in tree.roots():
        tree.root_by(rl)
        tree.annotate_branch(rl, "foo", "bar")
        tree.annotate_branch(rl, "fizz", "buzz")
    tree.root_by(tree.root_location("a"))
    tree.unroot()
    assert tree.newick() == ANNOT_ALL_ROOTS


def test_root_update_operations():             # test/src/tree.cpp:410-433
    t1 = tree_of("single")
    t1.root_by(t1.root_location("a"))
    ops, pmi, brl = t1.generate_root_update_operations(t1.root_location("d"))
    assert (len(ops), len(pmi), len(brl)) == (3, 4, 4)
    t2 = tree_of("single")
    t2.root_by(t2.root_location("b"))
    ops, pmi, brl = t2.generate_root_update_operations(t2.root_location("b"))
    assert (len(ops), len(pmi), len(brl)) == (0, 0, 0)


@pytest.mark.parametrize("key", ["10.fasta", "101.phy"])
def test_root_update_is_path_only(key):
    """move-root schedules touch only the old-root -> new-root path."""
    tree = tree_of(key)
    roots = tree.roots()
    tree.root_by(roots[0])
    n = tree.tip_count()
    for rl in roots[1:] + roots[:1]:
        ops, pmi, brl = tree.generate_root_update_operations(rl)
        assert 1 <= len(ops) <= n - 1
        assert ops[len(ops) - 1].parent_clv_index == 2 * n - 2
        assert len(set(pmi)) == len(pmi)


def test_bad_inputs():
    with pytest.raises(rd.RdamdError):
        rd.Tree.from_newick("(a:1,b:1);")
    with pytest.raises(rd.RdamdError):
        rd.Tree.from_newick("((a:1,b:1,c:1,d:1):1,e:1,f:1);")
    with pytest.raises(rd.RdamdError):
        rd.Tree.from_file("/nonexistent/tree.nwk")
    tree = tree_of("single")
    with pytest.raises(rd.RdamdError):
        tree.root_location(99)
    with pytest.raises(rd.RdamdError):
        tree.root_location("nope")


def test_msa_ingest_shapes():                  # test/src/msa.cpp:8-15 + SURVEY 2.1 fixture facts
    assert rd.msa_probe(os.path.join(util.DATA, "10.fasta")) == (10, 991, 1000)
    assert rd.msa_probe(os.path.join(util.DATA, "10.fasta"), compress=False) == (10, 1000, 1000)
    assert rd.msa_probe(os.path.join(util.DATA, "single.phy")) == (4, 1, 1)
    taxa, patterns, total = rd.msa_probe(os.path.join(util.DATA, "101.phy"))
    assert (taxa, total) == (101, 1858) and patterns <= 1858
    with pytest.raises(rd.RdamdError):
        rd.msa_probe(os.path.join(util.DATA, "10.tree"))


def test_midpoint_rooting_golden():
    """test/src/tree.cpp:435-443 of the reference: 10.tree rooted at its midpoint."""
    t = rd.Tree.from_file(os.path.join(util.DATA, "10.tree"))
    t.root_by(t.midpoint())
    assert t.newick(False) == (
        "((j:0.854700,((h:0.983500,a:0.224900):0.416200,(c:0.540900,f:0.422200):"
        "0.785300):0.614100):0.223050,(g:0.487400,(((i:0.569700,e:0.366600):0."
        "602800,b:0.445900):0.099300,d:0.639600):0.825200):0.223050);")
    # the ranking is a property of the unrooted tree: asking again while rooted
    # gives the same order
    assert t.rank_midpoints() == rd.Tree.from_file(os.path.join(util.DATA, "10.tree")).rank_midpoints()


def test_balance_rankings_are_permutations_and_find_the_long_branch():
    t = rd.Tree.from_newick("((a:1.0,b:1.0):0.5,c:1.0,d:9.0);")
    n_roots = t.root_count()
    for ranked in (t.rank_midpoints(), t.rank_modified_mad()):
        assert sorted(ranked) == list(range(n_roots))
    # the path a..d is 11.5 long, so its midpoint lies on d's 9.0 branch
    assert t.side_tips(t.midpoint()) in (["d"], ["a", "b", "c"])
    # brute force of the midpoint score on every branch (src/tree.cpp:864-885)
    import itertools
    best = None
    for rid in range(n_roots):
        rl = t.root_location(rid)
        near, far = t.side_tips(rl), None
        far = sorted(set("abcd") - set(near))
        dist = _tip_distances("((a:1.0,b:1.0):0.5,c:1.0,d:9.0);")
        score = -1.0
        for x, y in itertools.product(near, far):
            # distances from the two ends of the branch = path length minus the branch
            path = dist[(x, y)]
            # split the path at the branch: tip x to its end, tip y to the other end
            lx = _to_branch_end(dist, x, near, far, rl.saved_brlen)
            ly = path - rl.saved_brlen - lx
            a, b = max(lx, ly), min(lx, ly)
            gap = a - b
            if gap < rl.saved_brlen:
                b += gap + (rl.saved_brlen - gap) / 2
                a += (rl.saved_brlen - gap) / 2
            else:
                b += rl.saved_brlen
            span = a + b
            score = max(score, (1 - gap * gap / span) * span)
        if best is None or score > best[0] + 1e-12:
            best = (score, rid)
    assert t.rank_midpoints()[0] == best[1]


def _tip_distances(newick):
    """all tip-to-tip path lengths of the small test tree, by hand."""
    length = {"a": 1.0, "b": 1.0, "c": 1.0, "d": 9.0}
    inner = 0.5      # the (a,b) clade's stem
    d = {}
    for x in "abcd":
        for y in "abcd":
            if x == y:
                d[(x, y)] = 0.0
            elif {x, y} == {"a", "b"}:
                d[(x, y)] = 2.0
            elif x in "ab" or y in "ab":
                d[(x, y)] = length[x] + length[y] + inner
            else:
                d[(x, y)] = length[x] + length[y]
    return d


def _to_branch_end(dist, x, near, far, brlen):
    """distance from tip x to the near end of the root branch: for any far tip y
    and any other near tip x2, tree additivity gives it; for a lone tip it is 0."""
    if len(near) == 1:
        return 0.0
    y = far[0]
    x2 = [t for t in near if t != x][0]
    # d(x, end) = (d(x, y) + d(x, x2) - d(x2, y)) / 2 is the distance to the point
    # where the paths part; with two near tips that point is the near end itself
    return (dist[(x, y)] + dist[(x, x2)] - dist[(x2, y)]) / 2


@pytest.mark.parametrize("key", list(DATASETS))
def test_directional_schedule_shape_and_order(key):
    """generate_directional_operations: 3(n-2) directed operations in dependency
    order, then one root operation per branch; index ranges as documented."""
    t = tree_of(key)
    n, roots = t.tip_count(), t.root_count()
    d = t.generate_directional_operations()
    ops = list(d["ops"])
    assert len(ops) == 3 * (n - 2) + roots
    assert d["clv_buffers"] == 3 * (n - 2) + roots and d["prob_matrices"] == 3 * roots
    assert len(d["matrix_indices"]) == 3 * roots and sorted(d["matrix_indices"]) == list(range(3 * roots))
    ready = set(range(n))
    for op in ops[:3 * (n - 2)]:
        assert op.child1_clv_index in ready and op.child2_clv_index in ready   # dependencies first
        assert op.parent_clv_index not in ready and n <= op.parent_clv_index < n + 3 * (n - 2)
        assert op.parent_scaler_index == op.parent_clv_index - n
        ready.add(op.parent_clv_index)
    for rid, op in enumerate(ops[3 * (n - 2):]):
        assert op.parent_clv_index == d["root_clv"][rid] == n + 3 * (n - 2) + rid
        assert op.child1_clv_index in ready and op.child2_clv_index in ready
        assert (op.child1_matrix_index, op.child2_matrix_index) == (roots + 2 * rid, roots + 2 * rid + 1)
        rl = t.root_location(rid)
        i = list(d["matrix_indices"]).index(roots + 2 * rid)
        assert abs(d["branch_lengths"][i] - rl.brlen()) < 1e-15
        assert abs(d["branch_lengths"][i + 1] - rl.brlen_compliment()) < 1e-15
    # explicit ratios override the stored ones
    quarter = t.generate_directional_operations([0.25] * roots)["branch_lengths"]
    assert quarter[roots] == t.root_location(0).saved_brlen * 0.25
