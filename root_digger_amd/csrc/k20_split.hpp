// Cutting a post-order operation list into independent subtree pieces (the 20-state traversal
// kernel runs them side by side, kernels_clv_mfma.hip): pure host logic, no HIP, so that
// tests/cpp/host_logic_check.cpp can check it without a GPU.
#pragma once

#include <algorithm>
#include <cstddef>
#include <vector>

#include "../../include/root_digger_amd.h"

namespace rdamd {

// does operation i read, from memory, something one of the two operations before it (inside
// the same segment, which starts at `seg_start`) wrote and that cannot be forwarded in
// registers?  (20-state kernel, see rdamd_update_clvs)
inline bool k20_hazard(unsigned tips, const rdamd_operation_t *ops, unsigned i, unsigned seg_start) {
  const rdamd_operation_t &o = ops[i];
  const unsigned ch[2] = {o.child1_clv_index, o.child2_clv_index};
  const int chsc[2] = {o.child1_scaler_index, o.child2_scaler_index};
  bool hazard = false;
  for (int c = 0; c < 2; ++c) {
    if (ch[c] < tips) continue;
    int hits = 0, clean = 0;
    for (unsigned back = 1; back <= 2 && back <= i - seg_start; ++back) {
      const rdamd_operation_t &b = ops[i - back];
      const bool forwarded = ch[c] == b.parent_clv_index && chsc[c] == b.parent_scaler_index;
      const bool touches = ch[c] == b.parent_clv_index ||
                           (chsc[c] >= 0 && chsc[c] == b.parent_scaler_index);
      hits += touches;
      clean += forwarded;
    }
    hazard = hazard || hits != clean || hits > 1;
  }
  return hazard;
}

// The 20-state traversal kernel gives every 16-site tile ONE workgroup that walks the whole
// operation list: BASELINE config c3 (10 000 sites) has 625 of them for 256 CUs that could
// hold 768, and each is a 199-operation dependency chain.  A post-order list of a tree is a
// nest of contiguous subtree ranges, so it can be cut into independent PIECES -- subtrees of
// comparable size, longest first -- that run side by side (grid.y), followed by the few TOP
// operations that join them.  Returns the re-ordered list (pieces, then top) and the piece
// boundaries; leaves both empty when the list is no such nest (arbitrary operation orders,
// partial traversals) or too short to be worth a second launch.
// (`forwarding_hazards` = false: the 4-state kernel, which reads a memory child at use, behind
// every earlier store of the lane -- no piece can need a cut of its own there.  `small`: a piece
// of at most this many operations is not cut further; 0 = a quarter of the list, at least 12.
// `min_count`: shorter lists stay whole.  `external_children`: an inner child nothing in the list
// computes is taken as given, like a tip -- the top list of an earlier cut is such a list.)
inline void k20_split(unsigned tips, unsigned clv_buffers, const rdamd_operation_t *ops, unsigned count,
                      unsigned max_pieces, std::vector<rdamd_operation_t> &order, std::vector<unsigned> &bounds,
                      bool forwarding_hazards = true, unsigned small = 0, unsigned min_count = 24,
                      bool external_children = false) {
  order.clear();
  bounds.clear();
  if (count < min_count || count < 3) return;
  const unsigned nclv = tips + clv_buffers;
  std::vector<int> producer(nclv, -1), parent(count, -1);
  std::vector<unsigned> size(count, 1);
  std::vector<char> scaler_written;   // (pieces run side by side: no two operations may share a scale buffer)
  for (unsigned i = 0; i < count; ++i) {
    const rdamd_operation_t &o = ops[i];
    if (o.parent_clv_index < tips || o.parent_clv_index >= nclv || o.child1_clv_index >= nclv ||
        o.child2_clv_index >= nclv || producer[o.parent_clv_index] >= 0)
      return;
    if (o.parent_scaler_index >= 0) {
      const size_t sc = (size_t)o.parent_scaler_index;
      if (sc >= scaler_written.size()) scaler_written.resize(sc + 1, 0);
      if (scaler_written[sc]) return;
      scaler_written[sc] = 1;
    }
    // post-order: the later inner child sits right in front, the earlier one right in front of
    // the later one's whole subtree
    int kids[2], nk = 0;
    for (unsigned ch : {o.child1_clv_index, o.child2_clv_index})
      if (ch >= tips) {
        const int j = producer[ch];
        if (j < 0 && external_children) continue;
        if (j < 0 || parent[j] >= 0) return;   // not computed in this list / used twice
        kids[nk++] = j;
      }
    if (nk == 2 && kids[0] > kids[1]) std::swap(kids[0], kids[1]);
    if (nk >= 1 && kids[nk - 1] != (int)i - 1) return;
    if (nk == 2 && kids[0] != (int)i - 1 - (int)size[kids[1]]) return;
    for (int k = 0; k < nk; ++k) {
      parent[kids[k]] = (int)i;
      size[i] += size[kids[k]];
    }
    producer[o.parent_clv_index] = (int)i;
  }
  if (size[count - 1] != count) return;   // a forest
  // split the largest piece into its child subtrees until the pieces are comparable
  std::vector<unsigned> pieces{count - 1};   // by their last (root) operation
  std::vector<char> top(count, 0);
  for (;;) {
    size_t big = 0;
    for (size_t k = 1; k < pieces.size(); ++k)
      if (size[pieces[k]] > size[pieces[big]]) big = k;
    const unsigned r = pieces[big];
    if (size[r] <= (small ? small : std::max(12u, count / 4)) || pieces.size() + 1 > max_pieces) break;
    top[r] = 1;
    pieces.erase(pieces.begin() + (std::ptrdiff_t)big);
    for (unsigned j = 0; j < r; ++j)
      if (parent[j] == (int)r) pieces.push_back(j);
    if (pieces.empty()) return;   // (a root operation over two tips)
  }
  if (pieces.size() < 2) return;
  std::sort(pieces.begin(), pieces.end(), [&](unsigned a, unsigned b) { return size[a] > size[b]; });
  order.reserve(count);
  for (unsigned r : pieces) {
    bounds.push_back((unsigned)order.size());
    for (unsigned j = r + 1 - size[r]; j <= r; ++j) order.push_back(ops[j]);
  }
  bounds.push_back((unsigned)order.size());   // = where the top operations start
  for (unsigned j = 0; j < count; ++j)
    if (top[j]) order.push_back(ops[j]);
  // no piece may need a further cut of its own (then the plain order runs)
  for (size_t k = 0; forwarding_hazards && k + 1 < bounds.size(); ++k)
    for (unsigned i = bounds[k] + 1; i < bounds[k + 1]; ++i)
      if (k20_hazard(tips, order.data(), i, bounds[k])) {
        order.clear();
        bounds.clear();
        return;
      }
}

// The 4-state traversal kernel's launches for one list (kernels_clv.hip: every lane walks its
// list alone, so a launch lasts as long as its longest list): the list is cut into pieces, the
// operations that join them -- a list with external children -- are cut again, and so on until what
// is left is short.  `seg`: segment boundaries in `order` (segments = pieces, or a whole remaining
// list), `level`: the first segment of every launch (+ the segment count).  Empty `order`: the list
// runs as it is.
struct ListLevels {
  std::vector<rdamd_operation_t> order;
  std::vector<unsigned> seg, level;
};
// (`forwarding_hazards`: the 20-state kernel's pieces, see k20_split -- a level one of whose pieces
// would need a cut of its own is not cut.)
inline void list_levels(unsigned tips, unsigned clv_buffers, const rdamd_operation_t *ops, unsigned count,
                        unsigned max_pieces, unsigned small, unsigned min_count, ListLevels &out,
                        bool forwarding_hazards = false) {
  out.order.clear();
  out.seg.clear();
  out.level.clear();
  std::vector<rdamd_operation_t> cur(ops, ops + count), order;
  std::vector<unsigned> bounds;
  bool external = false;
  for (;;) {
    k20_split(tips, clv_buffers, cur.data(), (unsigned)cur.size(), max_pieces, order, bounds, forwarding_hazards,
              std::max(small, (unsigned)cur.size() / std::max(1u, max_pieces)), min_count, external);
    if (order.empty()) break;
    const unsigned base = (unsigned)out.order.size(), top = bounds.back();
    out.level.push_back((unsigned)out.seg.size());
    for (size_t k = 0; k + 1 < bounds.size(); ++k) out.seg.push_back(base + bounds[k]);
    out.order.insert(out.order.end(), order.begin(), order.begin() + top);
    cur.assign(order.begin() + top, order.end());
    external = true;
  }
  if (out.order.empty()) return;   // never cut
  out.level.push_back((unsigned)out.seg.size());   // what is left: one segment, one launch
  out.seg.push_back((unsigned)out.order.size());
  out.order.insert(out.order.end(), cur.begin(), cur.end());
  out.seg.push_back((unsigned)out.order.size());
  out.level.push_back((unsigned)out.seg.size() - 1);
}

}  // namespace rdamd
