// CPU check of two pieces of pure host logic added in round 3 (no HIP, no GPU):
//   * clade_classes (csrc/clade_classes.hpp): the pattern classes of a subtree -- what decides
//     which clades the fused evaluator folds into look-up tables -- against a brute-force count;
//   * k20_split (csrc/k20_split.hpp): cutting a post-order operation list into independent
//     subtree pieces -- against the properties the 20-state traversal kernel relies on.
// prints "host logic OK <cases>" on success.
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <set>
#include <vector>

#include "clade_classes.hpp"
#include "k20_split.hpp"

static int fail(const char *what, int a = 0, int b = 0) {
  std::printf("FAILED: %s (%d, %d)\n", what, a, b);
  return 1;
}

// random rooted binary tree with n tips as a post-order operation list (root_digger's index
// conventions: tips 0..n-1, inner CLVs n.., one matrix per child branch)
static std::vector<rdamd_operation_t> random_postorder(unsigned n, std::mt19937 &rng) {
  struct Node { int l = -1, r = -1; };
  std::vector<Node> nodes(n);            // tips
  std::vector<int> roots;
  for (unsigned i = 0; i < n; ++i) roots.push_back((int)i);
  while (roots.size() > 1) {             // join two random subtrees
    const size_t a = rng() % roots.size();
    int x = roots[a];
    roots.erase(roots.begin() + (std::ptrdiff_t)a);
    const size_t b = rng() % roots.size();
    int y = roots[b];
    roots.erase(roots.begin() + (std::ptrdiff_t)b);
    Node j;
    j.l = x; j.r = y;
    nodes.push_back(j);
    roots.push_back((int)nodes.size() - 1);
  }
  std::vector<rdamd_operation_t> ops;
  unsigned next_clv = n, next_mat = 0;
  std::vector<int> clv_of(nodes.size(), -1);
  for (unsigned i = 0; i < n; ++i) clv_of[i] = (int)i;
  // iterative post-order
  struct Frame { int node; int state; };
  std::vector<Frame> st{{roots[0], 0}};
  while (!st.empty()) {
    Frame &f = st.back();
    const Node &nd = nodes[(size_t)f.node];
    if (nd.l < 0) { st.pop_back(); continue; }
    if (f.state == 0) { f.state = 1; st.push_back({nd.l, 0}); continue; }
    if (f.state == 1) { f.state = 2; st.push_back({nd.r, 0}); continue; }
    rdamd_operation_t o;
    o.parent_clv_index = next_clv; o.parent_scaler_index = (int)(next_clv - n);
    o.child1_clv_index = (unsigned)clv_of[(size_t)nd.l]; o.child1_matrix_index = next_mat++;
    o.child1_scaler_index = nd.l < (int)n ? -1 : clv_of[(size_t)nd.l] - (int)n;
    o.child2_clv_index = (unsigned)clv_of[(size_t)nd.r]; o.child2_matrix_index = next_mat++;
    o.child2_scaler_index = nd.r < (int)n ? -1 : clv_of[(size_t)nd.r] - (int)n;
    clv_of[(size_t)f.node] = (int)next_clv++;
    ops.push_back(o);
    st.pop_back();
  }
  return ops;
}

int main() {
  std::mt19937 rng(20240603);
  int cases = 0;
  // ---- clade_classes ------------------------------------------------------------------------
  for (int rep = 0; rep < 400; ++rep) {
    const unsigned na = 1 + rng() % 64, nb = 1 + rng() % 64, limit = rep % 3 == 0 ? 16 : 64;
    const size_t S = rep % 7 == 0 ? 1 : 1 + rng() % 3000;
    const unsigned ua = 1 + rng() % na, ub = 1 + rng() % nb;   // values actually used: keeps many cases under the limit
    std::vector<uint8_t> a(S), b(S), cls, cmap;
    for (size_t s = 0; s < S; ++s) { a[s] = (uint8_t)(rng() % ua); b[s] = (uint8_t)(rng() % ub); }
    const unsigned got = rdamd::clade_classes(a.data(), na, b.data(), nb, S, limit, cls, cmap);
    std::map<std::pair<int, int>, int> first;   // brute force, first appearance order
    for (size_t s = 0; s < S; ++s) first.emplace(std::make_pair(a[s], b[s]), (int)first.size());
    if (first.size() > limit) {
      if (got != 0 || !cls.empty() || !cmap.empty()) return fail("over the limit must return 0", (int)first.size(), (int)got);
    } else {
      if (got != first.size()) return fail("class count", (int)got, (int)first.size());
      if (cmap.size() != 2 * got) return fail("map size");
      std::map<std::pair<int, int>, int> order;
      for (size_t s = 0; s < S; ++s) {
        auto it = order.emplace(std::make_pair(a[s], b[s]), (int)order.size()).first;
        if (cls[s] != it->second) return fail("class id is not the order of first appearance", (int)s);
        if (cmap[2 * cls[s]] != a[s] || cmap[2 * cls[s] + 1] != b[s]) return fail("class map", (int)s);
      }
    }
    ++cases;
  }
  // ---- k20_split ----------------------------------------------------------------------------
  for (int rep = 0; rep < 300; ++rep) {
    const unsigned n = 3 + rng() % 400;
    std::vector<rdamd_operation_t> ops = random_postorder(n, rng);
    const unsigned count = (unsigned)ops.size();
    std::vector<rdamd_operation_t> order;
    std::vector<unsigned> bounds;
    rdamd::k20_split(n, 2 * n, ops.data(), count, 8, order, bounds);
    if (order.empty()) {
      if (!bounds.empty()) return fail("bounds without an order");
      // allowed: short lists, or a tree that does not split into >= 2 pieces
      if (count >= 24) {
        // a list this long only stays whole when the root's children cannot both be pieces,
        // i.e. the root operation has a tip child and so on down (a caterpillar-like top)
        ++cases;
      }
      continue;
    }
    if (order.size() != count) return fail("not a permutation (size)");
    const unsigned np = (unsigned)bounds.size() - 1, top = bounds[np];
    if (np < 2 || np > 8 || bounds[0] != 0) return fail("piece count", (int)np);
    std::set<unsigned> seen;
    std::vector<int> piece_of_clv(3 * n, -2);   // -2: tip / unknown, -1: top, k: piece k
    for (unsigned k = 0; k <= np; ++k) {
      const unsigned lo = bounds[k], hi = k < np ? bounds[k + 1] : count;
      if (k < np && k > 0 && hi - lo > bounds[k] - bounds[k - 1]) return fail("pieces must come longest first", (int)k);
      if (k < np && hi - lo > std::max(12u, count / 4) && np < 8) return fail("an oversized piece was not split", (int)(hi - lo));
      std::set<unsigned> produced;
      for (unsigned i = lo; i < hi; ++i) {
        const rdamd_operation_t &o = order[i];
        if (!seen.insert(o.parent_clv_index).second) return fail("operation twice");
        for (unsigned ch : {o.child1_clv_index, o.child2_clv_index}) {
          if (ch < n) continue;
          if (k < np) {   // a piece reads only what it produced itself, earlier
            if (!produced.count(ch)) return fail("a piece reads from outside itself", (int)k, (int)i);
          } else if (piece_of_clv[ch] == -2) {
            return fail("a top operation reads something not computed yet", (int)i);
          }
        }
        produced.insert(o.parent_clv_index);
        piece_of_clv[o.parent_clv_index] = k < np ? (int)k : -1;
        if (i > lo && rdamd::k20_hazard(n, order.data(), i, lo) && k < np) return fail("hazard inside a piece");
      }
      // a piece is a whole subtree: exactly one of its results is consumed outside it
      if (k < np) {
        unsigned leaving = 0;
        for (unsigned i = top; i < count; ++i)
          for (unsigned ch : {order[i].child1_clv_index, order[i].child2_clv_index})
            if (produced.count(ch)) ++leaving;
        if (leaving != 1) return fail("a piece must hand exactly one CLV to the top list", (int)k, (int)leaving);
      }
    }
    if (order[count - 1].parent_clv_index != ops[count - 1].parent_clv_index) return fail("the root operation must stay last");
    ++cases;
  }
  {   // not a post-order nest: two operations swapped across subtrees -> no split
    std::vector<rdamd_operation_t> ops = random_postorder(200, rng);
    std::swap(ops[3], ops[150]);
    std::vector<rdamd_operation_t> order;
    std::vector<unsigned> bounds;
    rdamd::k20_split(200, 400, ops.data(), (unsigned)ops.size(), 8, order, bounds);
    if (!order.empty() || !bounds.empty()) return fail("a list that is no nest of subtree ranges must not be split");
    ++cases;
  }
  std::printf("host logic OK %d\n", cases);
  return 0;
}
