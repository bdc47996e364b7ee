// Host-side rooted-tree surface: mirrors rooted_tree_t of the reference
// (/root/reference/src/tree.hpp:54-201) -- same method names, argument
// meaning and index conventions -- on an index-based half-edge mesh instead of
// coraxlib's pointer rings.
//
// Half-edge h:  next[h] walks the ring of its node (tips: next = -1),
// back[h] is the twin across the branch.  Each node's ring shares clv_index /
// scaler_index; the two half-edges of a branch share length / pmatrix_index.
//
// Index conventions (pinned by the reference's
// test/src/tree.cpp:142-212): tips clv 0..n-1 (scaler -1), inner nodes clv
// n..2n-3 and scaler 0..n-3 in post-order from the parse root, branch pmatrix
// index = clv index of the node below it; the virtual root adds CLV 2n-2,
// scaler n-2 and pmatrix 2n-3.
#pragma once

#include <string>
#include <tuple>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

#include "../../include/root_digger_amd.h"

namespace rdamd {

struct root_location_t {   // src/tree.hpp:24-52
  int    edge = -1;        // half-edge id (the reference holds a corax_unode_t*)
  size_t id = 0;
  double saved_brlen = 0.0;
  double brlen_ratio = 0.5;

  double brlen() const { return saved_brlen * brlen_ratio; }
  double brlen_compliment() const { return saved_brlen * (1 - brlen_ratio); }
  bool operator==(const root_location_t &o) const {
    return edge == o.edge && brlen_ratio == o.brlen_ratio;
  }
  bool operator!=(const root_location_t &o) const { return !(*this == o); }
};

using op_schedule_t = std::tuple<std::vector<rdamd_operation_t>,
                                 std::vector<unsigned int>, std::vector<double>>;

class rooted_tree_t {
public:
  rooted_tree_t() = default;
  // src/tree.hpp:58-66 takes a file name; from_newick takes the text itself.
  static rooted_tree_t from_newick(const std::string &newick);
  static rooted_tree_t from_file(const std::string &filename);

  root_location_t root_location(size_t index) const;
  root_location_t root_location(const std::string &label) const;
  root_location_t root_location() const { return _current_rl; }
  const std::vector<root_location_t> &roots() const { return _roots; }
  size_t root_count() const { return _roots.size(); }

  unsigned int tip_count() const { return _tip_count; }
  unsigned int inner_count() const { return _inner_count + 1; }   // src/tree.cpp:103-105
  unsigned int branch_count() const { return _tip_count * 2 - 2; }
  unsigned int root_clv_index() const { return _clv[_vroot]; }
  int          root_scaler_index() const { return _scaler[_vroot]; }

  std::string label(const root_location_t &rl) const;
  bool is_internal(const root_location_t &rl) const;
  std::unordered_map<std::string, unsigned int> label_map() const;
  std::unordered_set<std::string>               label_set() const;
  std::string tip_label(unsigned int clv_index) const;

  // src/tree.cpp:364-413, :415-441, :572-657
  op_schedule_t generate_operations(const root_location_t &);
  std::tuple<rdamd_operation_t, std::vector<unsigned int>, std::vector<double>>
  generate_derivative_operations(const root_location_t &root);
  op_schedule_t generate_root_update_operations(const root_location_t &new_root);

  // src/tree.cpp:271-358
  void root_by(unsigned int root_id) { root_by(_roots.at(root_id)); }
  void root_by(const root_location_t &);
  void update_root(root_location_t);
  void unroot();
  bool rooted() const;
  bool branch_length_sanity_check() const;   // src/tree.cpp:498-517
  bool sanity_check() const { return branch_length_sanity_check(); }

  std::string newick(bool annotations = true) const;   // src/tree.cpp:443-492
  void annotate_branch(const root_location_t &rl, const std::string &key,
                       const std::string &value) { annotate_branch(rl, key, value, value); }
  void annotate_branch(const root_location_t &rl, const std::string &key,
                       const std::string &left_value, const std::string &right_value);
  // src/tree.cpp:709-726
  void annotate_lh(const root_location_t &rl, double lh) { annotate_branch(rl, "LLH", std::to_string(lh)); }
  void annotate_ratio(const root_location_t &rl, double ratio) {
    annotate_branch(rl, "alpha", std::to_string(ratio), std::to_string(1 - ratio));
  }
  void clear_newick_annotations() { _annotations.clear(); }

  // All-directions schedule for sweeps at FIXED parameters (SURVEY.md 8f item 2,
  // the GPU analogue of move_root / compute_all_root_lh, src/model.cpp:823-889):
  // one conditional likelihood vector per DIRECTED inner half-edge -- the subtree
  // behind the node, looking away from the branch -- does not depend on where the
  // root is, so 3(n-2) operations + one root operation per branch give every
  // root's likelihood.  Indices refer to a partition of its own:
  //   clv   tips .. tips+3(n-2)-1     directed inner half-edges
  //         then one per root id      the root CLVs
  //   scaler the same numbering minus `tips`
  //   pmatrix 0 .. 2n-4 the branches; 2n-3+2*rid, +1 the two halves of root rid
  struct directional_schedule_t {
    std::vector<rdamd_operation_t> ops;          // directed ops in dependency order, then root ops by id
    std::vector<unsigned int>      matrix_indices;
    std::vector<double>            branch_lengths;
    std::vector<unsigned int>      root_clv;     // per root id
    std::vector<int>               root_scaler;
    unsigned int clv_buffers = 0, scale_buffers = 0, prob_matrices = 0;
  };
  // the roots' alpha values are the ones stored in roots() unless `ratios` gives others
  directional_schedule_t generate_directional_operations(const std::vector<double> *ratios = nullptr) const;

  // Root placements ranked by how well they balance the tree, best first
  // (src/tree.cpp:863-945): the starting points of the heuristic search.
  // midpoint: score of a branch = max over (tip left, tip right) pairs of
  // d·(1 − Δ²/d) with the root slid along the branch to balance the pair;
  // modified MAD: root-mean-square relative deviation of the pairs' balance.
  std::vector<root_location_t> rank_midpoints() const;
  std::vector<root_location_t> rank_modified_mad() const;
  root_location_t midpoint() const { return rank_midpoints().front(); }   // src/tree.cpp:903-905

  // tips below each side of a root edge (test/diagnostic helper, not in the
  // reference): labels reachable from rl.edge without crossing the branch.
  std::vector<std::string> side_tips(const root_location_t &rl) const;

private:
  std::vector<int> full_traverse() const;               // src/tree.cpp:256-269
  template <typename F> void traverse(int root, F &&visit, std::vector<int> &out) const;
  void tag_ring(int h, bool v);
  // tip distances on both sides of every root branch, folded per branch
  template <typename Map, typename Reduce>
  std::vector<root_location_t> rank_branches(Map &&pair_score, Reduce &&fold) const;
  bool find_path_recurse(int n1, int n2);
  void find_path(int n1, int n2);

  // half-edge mesh
  std::vector<int>          _next, _back;
  std::vector<double>       _length;
  std::vector<unsigned int> _clv, _pmatrix;
  std::vector<int>          _scaler;
  std::vector<std::string>  _label;
  std::vector<char>         _tag;
  std::vector<int>          _tip_edge;   // clv index -> tip half-edge
  int          _vroot = -1, _root_left = -1, _root_right = -1;
  unsigned int _tip_count = 0, _inner_count = 0, _edge_count = 0;
  root_location_t              _current_rl;
  std::vector<root_location_t> _roots;
  std::unordered_map<int, std::vector<std::pair<std::string, std::string>>> _annotations;
};

}  // namespace rdamd
