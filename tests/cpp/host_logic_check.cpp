// CPU check of three pieces of pure host logic added in round 3 (no HIP, no GPU):
//   * clade_classes (csrc/clade_classes.hpp): the pattern classes of a subtree -- what decides
//     which clades the fused evaluator folds into look-up tables -- against a brute-force count;
//   * k20_split (csrc/k20_split.hpp): cutting a post-order operation list into independent
//     subtree pieces -- against the properties the 20-state traversal kernel relies on.
//   * the traversal compiler (csrc/traversal_compiler.hpp): its programs -- step order, which
//     stack level sits in the register slot(s), which in-memory entry in the LDS slot -- replayed
//     symbolically: the running value at the end must be the root's CLV expression, every pop
//     must meet the sibling that was parked for it, for 4-state (one / two register levels,
//     pseudo-tips) and 20-state (parking as a step of its own) programs.
// prints "host logic OK <cases>" on success.
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <set>
#include <vector>

#include "clade_classes.hpp"
#include "k20_split.hpp"
#include "traversal_compiler.hpp"

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

// ---- symbolic replay of a compiled program ---------------------------------------------------
// A CLV expression is a 64-bit hash: tip(row, table) for a table look-up, mv(matrix, x) for a
// matrix-vector product, x * y (commutative) for the element product.
static uint64_t mix(uint64_t a, uint64_t b) {
  a ^= b + 0x9e3779b97f4a7c15ull + (a << 6) + (a >> 2);
  a *= 0xff51afd7ed558ccdull;
  return a ^ (a >> 33);
}
static uint64_t h_tip(unsigned row, unsigned tab) { return mix(mix(1, row), tab); }
static uint64_t h_mv(unsigned mat, uint64_t x) { return mix(mix(2, mat), x); }
static uint64_t h_mul(uint64_t x, uint64_t y) { return mix(3, x < y ? mix(x, y) : mix(y, x)); }

// what the operation list says the root CLV is, in the terms the program uses (byte offsets)
static uint64_t expected(const rdamd::Compiler &c, const std::vector<rdamd_operation_t> &ops, unsigned i) {
  const rdamd_operation_t &o = ops[i];
  auto term = [&](unsigned clv, unsigned mat) -> uint64_t {
    if (!c.is_inner(clv)) {
      const unsigned tab = c.pseudo_wide.count(clv) ? c.wide_base + c.pseudo_wide.at(clv) * c.rate_cats * 512u
                                                    : mat * c.unit;
      return h_tip(c.row_of(clv) * c.tip_stride, tab);
    }
    return h_mv(mat * c.unit, expected(c, ops, c.producer.at(clv)));
  };
  return h_mul(term(o.child1_clv_index, o.child1_matrix_index), term(o.child2_clv_index, o.child2_matrix_index));
}

// runs the program the way the kernels do (kernels_fused.hip / kernels_fused_k20.hip): one or
// two register slots, the other entries on a stack addressed by the count of entries in it
// marked[k] (k = 0, 1): what the running CLV must be behind the step flagged 0x8000 << k -- the root
// operation's child k + 1 as the list defines it (0: that child is no inner node of the program,
// no step may carry the flag)
static int replay(const rdamd::Compiler &c, unsigned lds_pos, uint64_t want, unsigned &mem_depth,
                  const uint64_t marked[2] = nullptr) {
  using namespace rdamd;
  unsigned seen[2] = {0, 0};
  uint64_t run = 0, s0 = 0, s1 = 0, lds = 0;
  bool s0_full = false, s1_full = false, lds_full = false;
  const bool placed = !c.park_class.empty();   // parks with a place of their own: register slot, ONE LDS slot, the rest a stack
  std::vector<uint64_t> mem;
  mem_depth = 0;
  for (size_t i = 0; i < c.out.size(); ++i) {
    const FusedOp &f = c.out[i];
    const unsigned kind = f.flags & 3u;
    auto park = [&](uint64_t v) -> int {
      if (f.flags & 0x200u) { if (s0_full) return 1; s0 = v; s0_full = true; }
      else if (f.flags & 0x800u) { if (s1_full || c.reg_levels < 2) return 1; s1 = v; s1_full = true; }
      else if (placed && (f.flags & 0x20000u)) { if (lds_full) return 1; lds = v; lds_full = true; }
      else { mem.push_back(v); mem_depth = std::max(mem_depth, (unsigned)mem.size()); }
      if (!placed && (f.flags & 0x60000u)) return 1;
      return 0;
    };
    if (kind == kFusedPark) {
      if (!c.split_park) return fail("a park step in a 4-state program", (int)i);
      if (park(h_mv(f.pM, run))) return fail("park into an occupied register slot", (int)i);
    } else if (kind == kFusedTT) {
      if (f.flags & 0x100u) {
        if (c.split_park) return fail("a 20-state tip-tip step must not park", (int)i);
        if (park(h_mv(f.pM, run))) return fail("park into an occupied register slot", (int)i);
      }
      run = h_mul(h_tip(f.cX, f.tX), h_tip(f.cY, f.tY));
    } else if (kind == kFusedRT) {
      run = h_mul(h_mv(f.pM, run), h_tip(f.cY, f.tY));
    } else {   // kFusedRP
      uint64_t sib;
      if (f.flags & 0x400u) { if (!s0_full) return fail("pop from an empty register slot", (int)i); sib = s0; s0_full = false; }
      else if (f.flags & 0x1000u) { if (!s1_full) return fail("pop from an empty register slot 1", (int)i); sib = s1; s1_full = false; }
      else if (placed && (f.flags & 0x40000u)) { if (!lds_full) return fail("pop from an empty LDS slot", (int)i); sib = lds; lds_full = false; }
      else { if (mem.empty()) return fail("pop from an empty stack", (int)i); sib = mem.back(); mem.pop_back(); }
      run = h_mul(h_mv(f.pM, run), sib);
    }
    for (int k = 0; marked && k < 2; ++k)
      if (f.flags & (0x8000u << k)) {
        if (kind == kFusedPark || !marked[k] || run != marked[k])
          return fail("a step is flagged as the root operation's child but does not leave it in the running CLV", (int)i, k);
        ++seen[k];
      }
  }
  for (int k = 0; marked && k < 2; ++k)
    if (seen[k] != (marked[k] ? 1u : 0u)) return fail("root child: flagged steps", k, (int)seen[k]);
  if (s0_full || s1_full || lds_full || !mem.empty()) return fail("entries left on the stack");
  if (run != want) return fail("the program does not compute the root CLV");
  if (!placed && mem_depth >= 2 && c.reg_levels == 1 && lds_pos >= mem_depth) return fail("lds_pos out of range", (int)lds_pos, (int)mem_depth);
  if (placed && mem_depth != c.mem_depth) return fail("private-segment depth", (int)mem_depth, (int)c.mem_depth);
  return 0;
}

static int check_compiler(std::mt19937 &rng, int &cases) {
  unsigned long placed_total = 0, placed_in_slots = 0, placed_by_levels = 0;
  for (int rep = 0; rep < 600; ++rep) {
    const bool k20 = rep % 3 == 2;
    // shapes: random joins (deep, unbalanced), balanced (deepest stacks), caterpillar (depth 1)
    unsigned n = 3 + rng() % (rep % 10 == 0 ? 1500 : 300);
    std::vector<rdamd_operation_t> ops = random_postorder(n, rng);
    if (rep % 5 == 1) {   // perfectly balanced: pair up level by level
      n = 1u << (2 + rng() % 9);
      ops.clear();
      std::vector<unsigned> level(n);
      for (unsigned i = 0; i < n; ++i) level[i] = i;
      unsigned next_clv = n, next_mat = 0;
      while (level.size() > 1) {
        std::vector<unsigned> up;
        for (size_t i = 0; i + 1 < level.size(); i += 2) {
          rdamd_operation_t o;
          o.parent_clv_index = next_clv; o.parent_scaler_index = -1;
          o.child1_clv_index = level[i]; o.child1_matrix_index = next_mat++; o.child1_scaler_index = -1;
          o.child2_clv_index = level[i + 1]; o.child2_matrix_index = next_mat++; o.child2_scaler_index = -1;
          ops.push_back(o);
          up.push_back(next_clv++);
        }
        level.swap(up);
      }
    }
    rdamd::Compiler c;
    c.ops = ops.data(); c.n_ops = (unsigned)ops.size(); c.tips = n; c.sites = 1000;
    c.tip_stride = k20 ? 1000 : 2000; c.rate_cats = 4;
    c.unit = c.rate_cats * (k20 ? 3200u : 128u);
    c.split_park = k20;
    c.wide_base = 8u * (2 * n) * c.rate_cats * 16u;
    for (unsigned i = 0; i < c.n_ops; ++i) c.producer[ops[i].parent_clv_index] = i;
    if (!k20 && rep % 2 == 0) {   // some inner nodes become pseudo-tips (their subtrees leave the program)
      unsigned rows = n, wide = 0;
      for (unsigned i = 0; i + 1 < c.n_ops; ++i)
        if (rng() % 6 == 0) {
          c.pseudo_row[ops[i].parent_clv_index] = rows++;
          if (rng() % 2) c.pseudo_wide[ops[i].parent_clv_index] = wide++;
        }
    }
    // the steps that compute the root operation's inner children carry 0x8000 / 0x10000 (4 states:
    // the exporting evaluator, rdamd_evaluate_root_children)
    uint64_t marked[2] = {0, 0};
    if (!k20) {
      const rdamd_operation_t &root = ops.back();
      const unsigned kid[2] = {root.child1_clv_index, root.child2_clv_index};
      for (int k = 0; k < 2; ++k)
        if (kid[k] >= n) c.mark_clv[k] = kid[k];
    }
    c.need.assign(c.n_ops, 0);
    c.compute_need(c.n_ops - 1);
    c.emit(c.n_ops - 1, false, 0);
    const unsigned first_pass_depth = c.max_depth;
    unsigned parks = 0;
    for (unsigned l = 0; l < 16; ++l) parks += c.parks_at[l];
    // (the two thresholds rdamd_schedule_create uses: kernels with / without private-segment levels)
    const unsigned beyond = k20 ? 0u : (rep % 4 < 2 ? 1u + rdamd::kFusedSpillLevels : 3u);
    c.place_parks = !k20 && rep % 4 < 2;   // (what rdamd_schedule_create asks for with 64-row table slots)
    const unsigned lds_pos = c.place_levels(beyond, rdamd::kFusedSpillLevels - 1u);
    const bool placed = !c.park_class.empty();
    if (c.max_depth != first_pass_depth) return fail("the second pass changed the stack depth");
    unsigned parks2 = 0, busiest = 0;
    for (unsigned l = 0; l < 16; ++l) { parks2 += c.parks_at[l]; busiest = std::max(busiest, c.parks_at[l]); }
    if (parks2 != parks) return fail("the second pass changed the number of parks");
    if (!placed && c.reg_levels == 1 && c.max_depth >= 1 && c.parks_at[c.reg_level] != busiest)
      return fail("the register slot is not on the busiest level", (int)c.reg_level);
    if (placed) {   // placed park by park: never fewer in the two slots than the two busiest levels would hold
      unsigned in_slots = 0, best = 0, second = 0;
      for (const rdamd::FusedOp &f : c.out) in_slots += (f.flags & 0x100u) && (f.flags & 0x20200u);
      for (unsigned l = 0; l < 16; ++l) {
        if (c.parks_at[l] > best) { second = best; best = c.parks_at[l]; }
        else if (c.parks_at[l] > second) second = c.parks_at[l];
      }
      if (in_slots < best + second) return fail("placing the parks one by one must not lose to the level rule", (int)in_slots, (int)(best + second));
      placed_total += parks; placed_in_slots += in_slots; placed_by_levels += best + second;
    }
    if (k20 && c.reg_levels != 1) return fail("20-state programs have one register level");
    if (!k20 && (c.reg_levels == 2) != (c.max_depth > beyond)) return fail("two register levels", (int)c.max_depth);
    unsigned mem_depth = 0;
    if (!k20) {
      const rdamd_operation_t &root = ops.back();
      const unsigned kid[2] = {root.child1_clv_index, root.child2_clv_index};
      for (int k = 0; k < 2; ++k)   // (a child folded into a pseudo-tip is no step of the program)
        if (kid[k] >= n && c.is_inner(kid[k])) marked[k] = expected(c, ops, c.producer.at(kid[k]));
    }
    if (replay(c, lds_pos, expected(c, ops, c.n_ops - 1), mem_depth, k20 ? nullptr : marked)) return 1;
    if (!placed && mem_depth != (c.max_depth > c.reg_levels ? c.max_depth - c.reg_levels : 0)) return fail("in-memory depth", (int)mem_depth, (int)c.max_depth);
    if (placed && mem_depth > rdamd::kFusedSpillLevels - 1u) return fail("more private-segment entries than a wave has room for", (int)mem_depth);
    if (c.reg_levels == 1 && !k20 && mem_depth + 1 > beyond) return fail("more in-memory entries than the kernel has places for");
    // steps: one per operation left in the program (+ one per park for 20 states)
    size_t real = 0;
    for (const rdamd::FusedOp &f : c.out) real += (f.flags & 3u) != rdamd::kFusedPark || !k20;
    if (c.pseudo_row.empty() && real != c.n_ops) return fail("operations lost", (int)real, (int)c.n_ops);
    ++cases;
  }
  std::fprintf(stderr, "parks placed one by one: %lu of %lu in the two slots (the two busiest levels: %lu)\n", placed_in_slots,
              placed_total, placed_by_levels);
  return 0;
}

int main() {
  std::mt19937 rng(20240603);
  int cases = 0;
  if (check_compiler(rng, cases)) return 1;
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
  // ---- list_levels (the 4-state traversal kernel's launches) --------------------------------
  for (int rep = 0; rep < 300; ++rep) {
    const unsigned n = 3 + rng() % 600;
    std::vector<rdamd_operation_t> ops = random_postorder(n, rng);
    const unsigned count = (unsigned)ops.size();
    const unsigned max_pieces = 2 + rng() % 31, small = 4 + rng() % 12, min_count = 6 + rng() % 24;
    rdamd::ListLevels lv;
    rdamd::list_levels(n, 2 * n, ops.data(), count, max_pieces, small, min_count, lv);
    if (lv.order.empty()) {
      if (!lv.seg.empty() || !lv.level.empty()) return fail("levels without an order");
      ++cases;
      continue;
    }
    if (lv.order.size() != count) return fail("levels: not a permutation (size)");
    if (lv.seg.front() != 0 || lv.seg.back() != count) return fail("levels: segment bounds");
    if (lv.level.front() != 0 || lv.level.back() != lv.seg.size() - 1) return fail("levels: launch bounds");
    if (lv.level.size() < 3) return fail("levels: a cut list has at least two launches");
    std::set<unsigned> done_before;   // CLVs of earlier launches
    std::set<unsigned> seen;
    for (size_t l = 0; l + 1 < lv.level.size(); ++l) {
      const unsigned s0 = lv.level[l], s1 = lv.level[l + 1];
      if (s1 <= s0 || s1 - s0 > std::max(max_pieces, 1u)) return fail("levels: pieces per launch", (int)(s1 - s0));
      if (l + 2 == lv.level.size() && s1 - s0 != 1) return fail("levels: the last launch is one list");
      std::set<unsigned> this_launch;
      for (unsigned seg = s0; seg < s1; ++seg) {
        if (lv.seg[seg + 1] <= lv.seg[seg]) return fail("levels: empty segment");
        std::set<unsigned> produced;
        for (unsigned i = lv.seg[seg]; i < lv.seg[seg + 1]; ++i) {
          const rdamd_operation_t &o = lv.order[i];
          if (!seen.insert(o.parent_clv_index).second) return fail("levels: operation twice");
          for (unsigned ch : {o.child1_clv_index, o.child2_clv_index})
            if (ch >= n && !produced.count(ch) && !done_before.count(ch))
              return fail("levels: a segment reads what neither it nor an earlier launch computed", (int)l, (int)i);
          produced.insert(o.parent_clv_index);
        }
        this_launch.insert(produced.begin(), produced.end());
      }
      done_before.insert(this_launch.begin(), this_launch.end());
    }
    if (lv.order[count - 1].parent_clv_index != ops[count - 1].parent_clv_index) return fail("levels: the root operation must stay last");
    ++cases;
  }
  {   // two operations writing one scale buffer may not end up side by side -> no split
    std::vector<rdamd_operation_t> ops = random_postorder(200, rng);
    ops[40].parent_scaler_index = ops[120].parent_scaler_index = 7;
    std::vector<rdamd_operation_t> order;
    std::vector<unsigned> bounds;
    rdamd::k20_split(200, 400, ops.data(), (unsigned)ops.size(), 8, order, bounds);
    if (!order.empty() || !bounds.empty()) return fail("a shared scale buffer must keep the list whole");
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
