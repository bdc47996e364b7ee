// rooted_tree_t on a half-edge mesh; behaviour follows
// /root/reference/src/tree.cpp (cited per function) and the libpll/coraxlib
// utree conventions it builds on.
#include "tree.hpp"

#include <cmath>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <sstream>
#include <stdexcept>

namespace rdamd {

namespace {

struct pnode {   // parse tree
  std::string label;
  double length = 0.0;
  std::vector<int> children;
};

struct newick_parser {
  const std::string &s;
  size_t pos = 0;
  std::vector<pnode> nodes;
  explicit newick_parser(const std::string &text) : s(text) {}

  void skip_ws() {
    while (pos < s.size()) {
      if (isspace((unsigned char)s[pos])) { ++pos; continue; }
      if (s[pos] == '[') {   // comment
        while (pos < s.size() && s[pos] != ']') ++pos;
        if (pos < s.size()) ++pos;
        continue;
      }
      break;
    }
  }
  int parse_node() {
    skip_ws();
    int id = (int)nodes.size();
    nodes.emplace_back();
    if (pos < s.size() && s[pos] == '(') {
      ++pos;
      while (true) {
        int c = parse_node();
        nodes[id].children.push_back(c);
        skip_ws();
        if (pos >= s.size()) throw std::invalid_argument("newick: unexpected end");
        if (s[pos] == ',') { ++pos; continue; }
        if (s[pos] == ')') { ++pos; break; }
        throw std::invalid_argument("newick: expected ',' or ')'");
      }
    }
    skip_ws();
    size_t b = pos;
    if (pos < s.size() && (s[pos] == '\'' || s[pos] == '"')) {
      char q = s[pos++];
      b = pos;
      while (pos < s.size() && s[pos] != q) ++pos;
      nodes[id].label = s.substr(b, pos - b);
      if (pos < s.size()) ++pos;
    } else {
      while (pos < s.size() && !strchr(":,();[", s[pos]) && !isspace((unsigned char)s[pos])) ++pos;
      nodes[id].label = s.substr(b, pos - b);
    }
    skip_ws();
    if (pos < s.size() && s[pos] == ':') {
      ++pos;
      skip_ws();
      const char *start = s.c_str() + pos;
      char *end = nullptr;
      nodes[id].length = strtod(start, &end);
      if (end == start) throw std::invalid_argument("newick: bad branch length");
      pos += (size_t)(end - start);
    }
    return id;
  }
};

}  // namespace

// ---------------------------------------------------------------------------
// construction: parse, unroot a binary root the way
// corax_utree_parse_newick_unroot does (src/tree.cpp:11-13), assign indices,
// list the root locations (src/tree.cpp:173-189), reserve the virtual root
// (src/tree.cpp:213-236).
// ---------------------------------------------------------------------------
rooted_tree_t rooted_tree_t::from_file(const std::string &filename) {
  std::ifstream in(filename);
  if (!in) throw std::invalid_argument("Tree file could not be parsed");
  std::stringstream ss;
  ss << in.rdbuf();
  return from_newick(ss.str());
}

rooted_tree_t rooted_tree_t::from_newick(const std::string &text) {
  newick_parser ps(text);
  int proot;
  try {
    proot = ps.parse_node();
  } catch (const std::exception &) {
    throw std::invalid_argument("Tree file could not be parsed");
  }
  auto &pn = ps.nodes;
  rooted_tree_t t;

  auto new_half = [&t]() {
    int h = (int)t._next.size();
    t._next.push_back(-1); t._back.push_back(-1); t._length.push_back(0.0);
    t._clv.push_back(0); t._pmatrix.push_back(0); t._scaler.push_back(-1);
    t._label.emplace_back(); t._tag.push_back(0);
    return h;
  };
  auto link = [&t](int a, int b, double len) {
    t._back[a] = b; t._back[b] = a;
    t._length[a] = t._length[b] = len;
  };
  // builds the subtree of parse node `id`; returns its top half-edge
  std::function<int(int)> build = [&](int id) -> int {
    const pnode &n = pn[id];
    int top = new_half();
    t._label[top] = n.label;
    if (n.children.empty()) return top;
    if (n.children.size() != 2)
      throw std::invalid_argument("Tree file could not be parsed: only binary trees are supported");
    int a = new_half(), b = new_half();
    t._next[top] = a; t._next[a] = b; t._next[b] = top;
    t._label[a] = t._label[b] = n.label;
    int ca = build(n.children[0]);
    link(a, ca, pn[n.children[0]].length);
    int cb = build(n.children[1]);
    link(b, cb, pn[n.children[1]].length);
    return top;
  };

  const pnode &r = pn[proot];
  int h0 = new_half(), h1 = new_half(), h2 = new_half();
  t._next[h0] = h1; t._next[h1] = h2; t._next[h2] = h0;
  if (r.children.size() == 3) {
    t._label[h0] = t._label[h1] = t._label[h2] = r.label;
    for (int k = 0; k < 3; ++k) {
      int c = build(r.children[k]);
      link(k == 0 ? h0 : (k == 1 ? h1 : h2), c, pn[r.children[k]].length);
    }
  } else if (r.children.size() == 2) {
    // binary root: the first root child that has descendants becomes the
    // trifurcating node; the root branch lengths are summed.
    int left = r.children[0], right = r.children[1];
    int keep, other;
    if (!pn[left].children.empty()) { keep = left; other = right; }
    else if (!pn[right].children.empty()) { keep = right; other = left; }
    else throw std::invalid_argument("Tree requires at least three tips");
    if (pn[keep].children.size() != 2)
      throw std::invalid_argument("Tree file could not be parsed: only binary trees are supported");
    t._label[h0] = t._label[h1] = t._label[h2] = pn[keep].label;
    int co = build(other);
    link(h0, co, pn[left].length + pn[right].length);
    int c1 = build(pn[keep].children[0]);
    link(h1, c1, pn[pn[keep].children[0]].length);
    int c2 = build(pn[keep].children[1]);
    link(h2, c2, pn[pn[keep].children[1]].length);
  } else {
    throw std::invalid_argument("Tree file could not be parsed: root must have 2 or 3 children");
  }
  t._vroot = h0;

  // index assignment: post-order from vroot (back subtree first); tips in visit
  // order, inner nodes in post-order; a branch carries the clv index of the
  // node below it.
  unsigned tips = 0;
  for (size_t h = 0; h < t._next.size(); ++h)
    if (t._next[h] < 0) ++tips;
  if (tips < 3) throw std::invalid_argument("Tree requires at least three tips");
  t._tip_count = tips;
  t._inner_count = tips - 2;
  t._edge_count = 2 * tips - 3;
  t._tip_edge.assign(tips, -1);
  unsigned next_tip = 0, next_inner = tips;
  int next_scaler = 0;
  std::function<void(int)> number = [&](int top) {
    if (t._next[top] < 0) {
      t._clv[top] = next_tip; t._scaler[top] = RDAMD_SCALE_BUFFER_NONE;
      t._tip_edge[next_tip] = top;
      ++next_tip;
    } else {
      for (int s = t._next[top]; s != top; s = t._next[s]) number(t._back[s]);
      for (int s = top, first = 1; first || s != top; s = t._next[s], first = 0) {
        t._clv[s] = next_inner; t._scaler[s] = next_scaler;
      }
      ++next_inner; ++next_scaler;
    }
    t._pmatrix[top] = t._pmatrix[t._back[top]] = t._clv[top];
  };
  number(t._back[h0]);
  number(t._back[h1]);
  number(t._back[h2]);
  for (int s : {h0, h1, h2}) { t._clv[s] = next_inner; t._scaler[s] = next_scaler; }

  // root locations: one per branch, named by the half-edge met first in the
  // post-order traversal (src/tree.cpp:173-189)
  {
    auto trav = t.full_traverse();
    std::unordered_set<int> seen;
    size_t id = 0;
    for (int e : trav) {
      if (!seen.count(e) && !seen.count(t._back[e])) {
        seen.insert(e);
        t._roots.push_back({e, id++, t._length[e], 0.5});
      }
    }
  }
  // virtual root ring of two (src/tree.cpp:213-236)
  t._root_left = new_half();
  t._root_right = new_half();
  t._next[t._root_left] = t._root_right;
  t._next[t._root_right] = t._root_left;
  t._clv[t._root_left] = t._clv[t._root_right] = 2 * tips - 2;
  t._scaler[t._root_left] = t._scaler[t._root_right] = (int)tips - 2;
  t._pmatrix[t._root_right] = 2 * tips - 3;
  return t;
}

// ---------------------------------------------------------------------------
// traversal: post-order from `root`: subtree behind root.back first, then the
// ring of root, root last (corax_utree_traverse POSTORDER).  visit(h) == false
// prunes the subtree.
// ---------------------------------------------------------------------------
template <typename F>
void rooted_tree_t::traverse(int root, F &&visit, std::vector<int> &out) const {
  std::function<void(int)> rec = [&](int node) {
    if (!visit(node)) return;
    if (_next[node] >= 0)
      for (int s = _next[node]; s != node; s = _next[s]) rec(_back[s]);
    out.push_back(node);
  };
  rec(_back[root]);
  rec(root);
}

std::vector<int> rooted_tree_t::full_traverse() const {
  std::vector<int> out;
  out.reserve(2 * _tip_count);
  traverse(_vroot, [](int) { return true; }, out);
  return out;
}

root_location_t rooted_tree_t::root_location(size_t index) const {
  if (index >= _roots.size())
    throw std::invalid_argument("Invalid index for roots on this tree: " + std::to_string(index));
  return _roots[index];
}

root_location_t rooted_tree_t::root_location(const std::string &name) const {
  for (const auto &rl : _roots)
    if (_label[rl.edge] == name && !name.empty()) return rl;
  throw std::runtime_error("Can't find the root location with label: " + name);
}

std::string rooted_tree_t::label(const root_location_t &rl) const {
  return _label[rl.edge].empty() ? "(null)" : _label[rl.edge];
}

bool rooted_tree_t::is_internal(const root_location_t &rl) const {
  return _next[rl.edge] >= 0 && _next[_back[rl.edge]] >= 0;
}

std::unordered_map<std::string, unsigned int> rooted_tree_t::label_map() const {
  std::unordered_map<std::string, unsigned int> m;
  for (unsigned i = 0; i < _tip_count; ++i) m[_label[_tip_edge[i]]] = i;
  return m;
}

std::unordered_set<std::string> rooted_tree_t::label_set() const {
  std::unordered_set<std::string> m;
  for (unsigned i = 0; i < _tip_count; ++i) m.insert(_label[_tip_edge[i]]);
  return m;
}

std::string rooted_tree_t::tip_label(unsigned int clv_index) const {
  return _label[_tip_edge.at(clv_index)];
}

bool rooted_tree_t::rooted() const { return _next[_next[_vroot]] == _vroot; }

// src/tree.cpp:273-320
void rooted_tree_t::root_by(const root_location_t &rl) {
  if (rooted() && rl.edge == _current_rl.edge) {   // same branch: only alpha moves
    update_root(rl);
    return;
  }
  if (rooted()) unroot();
  const int left_child = rl.edge, right_child = _back[rl.edge];
  _back[left_child] = _root_left; _back[_root_left] = left_child;
  _length[left_child] = _length[_root_left] = rl.brlen();
  _back[right_child] = _root_right; _back[_root_right] = right_child;
  _length[right_child] = _length[_root_right] = rl.brlen_compliment();
  _inner_count += 1;
  _edge_count += 1;
  _vroot = _root_left;
  _clv[_root_left] = _clv[_root_right] = 2 * _tip_count - 2;
  _scaler[_root_left] = _scaler[_root_right] = (int)_inner_count - 1;
  _pmatrix[_root_left] = _pmatrix[left_child];
  _pmatrix[right_child] = _pmatrix[_root_right] = _edge_count - 1;
  _current_rl = rl;
}

// src/tree.cpp:322-332.  The reference only reaches its update_root when
// root.edge equals the (virtual) vroot, which a root location never does
// while rooted, so there it is dead code; here it is the live fast path for
// "same branch, new alpha" and leaves exactly the state a full unroot +
// root_by would.
void rooted_tree_t::update_root(root_location_t root) {
  if (!rooted() || root.edge != _current_rl.edge)
    throw std::runtime_error("Provided root doesn't match the current tree");
  int left_child = root.edge, right_child = _back[_root_right];
  _length[left_child] = _length[_root_left] = root.brlen();
  _length[right_child] = _length[_root_right] = root.brlen_compliment();
  _current_rl = root;
}

// src/tree.cpp:334-358
void rooted_tree_t::unroot() {
  if (!rooted()) return;
  int left_child = _back[_vroot], right_child = _back[_next[_vroot]];
  _back[right_child] = left_child;
  _back[left_child] = right_child;
  _length[right_child] = _length[left_child] = _current_rl.saved_brlen;
  _back[_root_left] = _back[_root_right] = -1;
  _length[_root_left] = _length[_root_right] = -1.0;
  _vroot = _next[left_child] >= 0 ? left_child : right_child;
  if (_next[_vroot] < 0) throw std::runtime_error("unrooted to a tip");
  _inner_count -= 1;
  _edge_count -= 1;
  _pmatrix[right_child] = _pmatrix[left_child];
}

namespace {
// corax_utree_create_operations restated: every node contributes its branch
// (length, pmatrix) unless it is the twin of the last node; inner nodes
// contribute an operation whose children are ring->next, ring->next->next.
void create_operations(const std::vector<int> &trav, size_t count,
                       const std::vector<int> &next, const std::vector<int> &back,
                       const std::vector<double> &length,
                       const std::vector<unsigned> &clv, const std::vector<unsigned> &pm,
                       const std::vector<int> &scaler, std::vector<rdamd_operation_t> &ops,
                       std::vector<unsigned> &pmatrix_indices, std::vector<double> &brlens) {
  for (size_t i = 0; i < count; ++i) {
    int node = trav[i];
    if (node != back[trav[count - 1]]) {
      brlens.push_back(length[node]);
      pmatrix_indices.push_back(pm[node]);
    }
    if (next[node] >= 0) {
      int c1 = back[next[node]], c2 = back[next[next[node]]];
      rdamd_operation_t op;
      op.parent_clv_index = clv[node]; op.parent_scaler_index = scaler[node];
      op.child1_clv_index = clv[c1]; op.child1_scaler_index = scaler[c1];
      op.child1_matrix_index = pm[c1];
      op.child2_clv_index = clv[c2]; op.child2_scaler_index = scaler[c2];
      op.child2_matrix_index = pm[c2];
      ops.push_back(op);
    }
  }
}
}  // namespace

// src/tree.cpp:364-413
op_schedule_t rooted_tree_t::generate_operations(const root_location_t &new_root) {
  root_by(new_root);
  auto trav = full_traverse();
  std::vector<rdamd_operation_t> ops;
  std::vector<unsigned> pmi;
  std::vector<double> brl;
  ops.reserve(trav.size()); pmi.reserve(trav.size()); brl.reserve(trav.size());
  create_operations(trav, trav.size() - 1, _next, _back, _length, _clv, _pmatrix, _scaler,
                    ops, pmi, brl);
  int root_node = trav.back();
  int c1 = _back[root_node], c2 = _back[_next[root_node]];
  rdamd_operation_t op;
  op.parent_clv_index = _clv[root_node]; op.parent_scaler_index = _scaler[root_node];
  op.child1_clv_index = _clv[c1]; op.child1_scaler_index = _scaler[c1];
  op.child1_matrix_index = _pmatrix[c1];
  op.child2_clv_index = _clv[c2]; op.child2_scaler_index = _scaler[c2];
  op.child2_matrix_index = _pmatrix[c2];
  ops.push_back(op);
  return std::make_tuple(std::move(ops), std::move(pmi), std::move(brl));
}

// src/tree.cpp:415-441
std::tuple<rdamd_operation_t, std::vector<unsigned int>, std::vector<double>>
rooted_tree_t::generate_derivative_operations(const root_location_t &root) {
  root_by(root);
  int c1 = _back[_vroot], c2 = _back[_next[_vroot]];
  rdamd_operation_t op;
  op.parent_clv_index = root_clv_index(); op.parent_scaler_index = root_scaler_index();
  op.child1_clv_index = _clv[c1]; op.child1_matrix_index = _pmatrix[c1];
  op.child1_scaler_index = _scaler[c1];
  op.child2_clv_index = _clv[c2]; op.child2_matrix_index = _pmatrix[c2];
  op.child2_scaler_index = _scaler[c2];
  std::vector<unsigned> pmi{_pmatrix[c1], _pmatrix[c2]};
  std::vector<double> brl{_length[c1], _length[c2]};
  return std::make_tuple(op, pmi, brl);
}

void rooted_tree_t::tag_ring(int h, bool v) {
  int s = h;
  do {
    _tag[s] = v;
    s = _next[s];
  } while (s >= 0 && s != h);
}

// src/tree.cpp:542-570: depth-first search for n2 starting behind n1, tagging
// the ring of every node on the way back up.
bool rooted_tree_t::find_path_recurse(int n1, int n2) {
  if (n1 == n2) { tag_ring(n1, true); return true; }
  if (_next[n1] >= 0) {
    int start = n1;
    n1 = _next[n1];
    while (start != n1) {
      if (n1 == n2) { tag_ring(n1, true); return true; }
      if (find_path_recurse(_back[n1], n2)) { tag_ring(n1, true); return true; }
      n1 = _next[n1];
    }
  }
  return false;
}

void rooted_tree_t::find_path(int n1, int n2) {
  int start = n1, cur = n1;
  do {
    if (find_path_recurse(_back[cur], n2)) break;
    cur = _next[cur];
  } while (cur >= 0 && cur != start);
}

// src/tree.cpp:572-657: only the nodes whose orientation changes when the root
// moves (the path old root -> new root, both root branches included) are
// re-evaluated.
op_schedule_t rooted_tree_t::generate_root_update_operations(const root_location_t &new_root) {
  if (_current_rl.edge < 0 || !rooted())
    throw std::runtime_error("generate_root_update_operations needs a rooted tree");
  // (the reference also compares against _current_rl.edge->back, which while
  // rooted is the virtual root and never a root location)
  if (new_root.edge == _current_rl.edge) return {};
  auto old_root = _current_rl;
  root_by(new_root);
  find_path(old_root.edge, _vroot);
  tag_ring(old_root.edge, true);
  tag_ring(_back[old_root.edge], true);
  tag_ring(_back[_vroot], true);
  tag_ring(_back[_next[_vroot]], true);

  std::vector<int> trav;
  traverse(_vroot,
           [this](int n) {
             if (_tag[n]) { _tag[n] = 0; return true; }
             return false;
           },
           trav);
  if (trav.empty())
    throw std::runtime_error("traversal buffer when updating the root had size zero");
  std::vector<rdamd_operation_t> ops;
  std::vector<unsigned> pmi;
  std::vector<double> brl;
  create_operations(trav, trav.size() - 1, _next, _back, _length, _clv, _pmatrix, _scaler,
                    ops, pmi, brl);
  int root_node = _vroot;
  int c1 = _back[root_node], c2 = _back[_next[root_node]];
  rdamd_operation_t op;
  op.parent_clv_index = _clv[root_node]; op.parent_scaler_index = _scaler[root_node];
  op.child1_clv_index = _clv[c1]; op.child1_scaler_index = _scaler[c1];
  op.child1_matrix_index = _pmatrix[c1];
  op.child2_clv_index = _clv[c2]; op.child2_scaler_index = _scaler[c2];
  op.child2_matrix_index = _pmatrix[c2];
  ops.push_back(op);
  std::fill(_tag.begin(), _tag.end(), 0);
  return std::make_tuple(std::move(ops), std::move(pmi), std::move(brl));
}

// src/tree.cpp:498-517
bool rooted_tree_t::branch_length_sanity_check() const {
  auto nodes = full_traverse();
  nodes.pop_back();
  std::vector<double> len;
  for (int n : nodes) len.push_back(_length[n]);
  std::sort(len.begin(), len.end());
  double median = (len[(len.size() - 1) / 2] + len[len.size() / 2]) / 2.0;
  if (median * 10.0 < len.back() || len.front() < median / 10.0) return false;
  return true;
}

// src/tree.cpp:443-492 (label:length with six decimals + NHX annotations)
std::string rooted_tree_t::newick(bool annotations) const {
  auto serialize = [&](int n) {
    char buf[64];
    snprintf(buf, sizeof(buf), "%f", _length[n]);
    std::string s = _label[n] + ":" + buf;
    if (annotations) {
      auto it = _annotations.find(n);
      if (it != _annotations.end() && !it->second.empty()) {
        s += "[&&NHX";
        for (auto &kv : it->second) s += ":" + kv.first + "=" + kv.second;
        s += "]";
      }
    }
    return s;
  };
  std::function<std::string(int)> sub = [&](int n) -> std::string {
    if (_next[n] < 0) return serialize(n);
    std::string s = "(";
    bool first = true;
    for (int k = _next[n]; k != n; k = _next[k]) {
      if (!first) s += ",";
      s += sub(_back[k]);
      first = false;
    }
    return s + ")" + serialize(n);
  };
  int root = _vroot;
  std::string s = "(" + sub(_back[root]);
  for (int k = _next[root]; k != root; k = _next[k]) s += "," + sub(_back[k]);
  return s + ")" + _label[root] + ";";
}

// src/tree.cpp:731-760
void rooted_tree_t::annotate_branch(const root_location_t &rl, const std::string &key,
                                    const std::string &value, const std::string &right_value) {
  _annotations[rl.edge].emplace_back(key, value);
  int other = _back[rl.edge];
  size_t ring = 1;
  if (_next[other] >= 0) {
    ring = 0;
    int c = other;
    do { ++ring; c = _next[c]; } while (c != other);
  }
  if (ring > 2) _annotations[other].emplace_back(key, right_value);
  else _annotations[_back[_next[other]]].emplace_back(key, right_value);
}

std::vector<std::string> rooted_tree_t::side_tips(const root_location_t &rl) const {
  std::vector<std::string> out;
  std::function<void(int)> rec = [&](int n) {
    if (_next[n] < 0) { out.push_back(_label[n]); return; }
    for (int k = _next[n]; k != n; k = _next[k]) rec(_back[k]);
  };
  rec(rl.edge);
  std::sort(out.begin(), out.end());
  return out;
}


// ---- balance-based rankings (src/tree.cpp:795-945) ----------------------------------
namespace {
// distances from the far end of half-edge h to every tip behind it, depth first
// in ring order; `depth` is the distance already walked before crossing h
void collect_tip_distances(const std::vector<int> &next, const std::vector<int> &back,
                           const std::vector<double> &length, int h, double depth,
                           std::vector<double> &out) {
  depth += length[h];
  if (next[h] < 0) {
    out.push_back(depth);
    return;
  }
  for (int k = next[h]; k != h; k = next[k]) collect_tip_distances(next, back, length, back[k], depth, out);
}
}  // namespace

template <typename Map, typename Reduce>
std::vector<root_location_t> rooted_tree_t::rank_branches(Map &&pair_score, Reduce &&fold) const {
  if (rooted()) {   // the rankings are a property of the unrooted tree
    rooted_tree_t bare(*this);
    bare.unroot();
    return bare.rank_branches(pair_score, fold);
  }
  std::vector<std::pair<double, size_t>> scored;
  std::vector<double> near, far, vals;
  for (size_t i = 0; i < _roots.size(); ++i) {
    const int e = _roots[i].edge;
    near.clear();
    far.clear();
    if (_next[e] < 0) near.push_back(0.0);   // the branch hangs off a tip
    else
      for (int k = _next[e]; k != e; k = _next[k])
        collect_tip_distances(_next, _back, _length, _back[k], 0.0, near);
    collect_tip_distances(_next, _back, _length, _back[e], -_length[e], far);
    vals.clear();
    for (double a : near)
      for (double b : far) vals.push_back(pair_score(a, b, _roots[i].saved_brlen));
    scored.emplace_back(fold(vals), i);
  }
  std::sort(scored.begin(), scored.end(),
            [](const std::pair<double, size_t> &a, const std::pair<double, size_t> &b) {
              return a.first > b.first;
            });
  std::vector<root_location_t> out;
  out.reserve(scored.size());
  for (const auto &sc : scored) out.push_back(_roots[sc.second]);
  return out;
}

std::vector<root_location_t> rooted_tree_t::rank_midpoints() const {
  auto pair_score = [](double longer, double shorter, double brlen) {
    if (longer < shorter) std::swap(longer, shorter);
    const double gap = longer - shorter;
    if (gap < brlen) {   // the branch can absorb the gap: what is left is shared
      shorter += gap;
      const double half = (brlen - gap) / 2.0;
      shorter += half;
      longer += half;
    } else {
      shorter += brlen;
    }
    const double span = shorter + longer;
    return (1 - (gap * gap) / span) * span;
  };
  auto fold = [](const std::vector<double> &v) { return *std::max_element(v.begin(), v.end()); };
  return rank_branches(pair_score, fold);
}

std::vector<root_location_t> rooted_tree_t::rank_modified_mad() const {
  auto pair_score = [](double a, double b, double brlen) {
    const double span = a + b + brlen;
    const double rho = std::min(std::max((span - 2 * a) / (2 * brlen), 0.0), 1.0);
    a = a + rho * brlen;
    return a / span - 1;
  };
  auto fold = [](const std::vector<double> &v) {
    double acc = 0.0;
    for (double x : v) acc += x * x;
    acc /= static_cast<double>(v.size());
    return std::sqrt(acc);
  };
  return rank_branches(pair_score, fold);
}


// ---- all-directions schedule ---------------------------------------------------------
rooted_tree_t::directional_schedule_t
rooted_tree_t::generate_directional_operations(const std::vector<double> *ratios) const {
  if (rooted()) {
    rooted_tree_t bare(*this);
    bare.unroot();
    return bare.generate_directional_operations(ratios);
  }
  if (ratios && ratios->size() != _roots.size())
    throw std::invalid_argument("generate_directional_operations: one ratio per root location");
  directional_schedule_t out;
  const unsigned n = _tip_count, n_roots = (unsigned)_roots.size();
  const size_t H = _next.size();
  // number the directed inner half-edges and the undirected branches
  std::vector<int> dir_id(H, -1), branch_id(H, -1);
  unsigned n_dir = 0, n_branch = 0;
  for (size_t h = 0; h < H; ++h) {
    if ((int)h == _root_left || (int)h == _root_right || _back[h] < 0) continue;   // spare root slots
    if (_next[h] >= 0) dir_id[h] = (int)n_dir++;
    if (branch_id[h] < 0) branch_id[h] = branch_id[_back[h]] = (int)n_branch++;
  }
  auto clv_of = [&](int h) { return _next[h] < 0 ? _clv[h] : n + (unsigned)dir_id[h]; };
  auto scaler_of = [&](int h) { return _next[h] < 0 ? -1 : dir_id[h]; };
  out.matrix_indices.resize(n_branch);
  out.branch_lengths.resize(n_branch);
  for (size_t h = 0; h < H; ++h)
    if (branch_id[h] >= 0) {
      out.matrix_indices[branch_id[h]] = (unsigned)branch_id[h];
      out.branch_lengths[branch_id[h]] = _length[h];
    }
  // D(h) needs D(back(k)) of the two other ring members k: depth-first, each once
  std::vector<char> done(H, 0);
  std::vector<std::pair<int, int>> stack;   // (half-edge, next ring member to look at: 0, 1, 2 = emit)
  for (size_t h0 = 0; h0 < H; ++h0) {
    if (dir_id[h0] < 0 || done[h0]) continue;
    stack.emplace_back((int)h0, 0);
    while (!stack.empty()) {
      auto &[h, stage] = stack.back();
      const int k1 = _next[h], k2 = _next[k1];
      if (stage < 2) {
        const int child = _back[stage == 0 ? k1 : k2];
        ++stage;
        if (_next[child] >= 0 && !done[child]) stack.emplace_back(child, 0);
        continue;
      }
      rdamd_operation_t op;
      op.parent_clv_index = clv_of(h); op.parent_scaler_index = scaler_of(h);
      op.child1_clv_index = clv_of(_back[k1]); op.child1_matrix_index = (unsigned)branch_id[k1];
      op.child1_scaler_index = scaler_of(_back[k1]);
      op.child2_clv_index = clv_of(_back[k2]); op.child2_matrix_index = (unsigned)branch_id[k2];
      op.child2_scaler_index = scaler_of(_back[k2]);
      if (!done[h]) out.ops.push_back(op);
      done[h] = 1;
      stack.pop_back();
    }
  }
  // one root operation per branch: the two directed CLVs that face each other
  // across it, through the two halves of the branch (root_by's child order)
  out.root_clv.resize(n_roots);
  out.root_scaler.resize(n_roots);
  for (unsigned rid = 0; rid < n_roots; ++rid) {
    root_location_t rl = _roots[rid];
    if (ratios) rl.brlen_ratio = (*ratios)[rid];
    const int a = rl.edge, b = _back[rl.edge];
    const unsigned m = n_branch + 2 * rid;
    out.matrix_indices.push_back(m);     out.branch_lengths.push_back(rl.brlen());
    out.matrix_indices.push_back(m + 1); out.branch_lengths.push_back(rl.brlen_compliment());
    rdamd_operation_t op;
    op.parent_clv_index = n + n_dir + rid; op.parent_scaler_index = (int)(n_dir + rid);
    op.child1_clv_index = clv_of(a); op.child1_matrix_index = m;     op.child1_scaler_index = scaler_of(a);
    op.child2_clv_index = clv_of(b); op.child2_matrix_index = m + 1; op.child2_scaler_index = scaler_of(b);
    out.ops.push_back(op);
    out.root_clv[rid] = op.parent_clv_index;
    out.root_scaler[rid] = op.parent_scaler_index;
  }
  out.clv_buffers = n_dir + n_roots;
  out.scale_buffers = n_dir + n_roots;
  out.prob_matrices = n_branch + 2 * n_roots;
  return out;
}

}  // namespace rdamd
