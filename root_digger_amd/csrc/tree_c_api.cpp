// C wrappers over rooted_tree_t (declared in include/root_digger_amd.h).
#include <cstdlib>
#include <cstring>

#include "common.hpp"
#include "tree.hpp"
#include "tree_c_api.hpp"

#include <algorithm>

struct rdamd_tree {
  rdamd::rooted_tree_t tree;
  mutable std::string  scratch;
};

using rdamd::root_location_t;

const rdamd::rooted_tree_t &rdamd_tree_cpp(const rdamd_tree_t *t) { return t->tree; }

namespace {
root_location_t to_cpp(const rdamd_root_location_t *rl) {
  root_location_t r;
  r.edge = rl->edge; r.id = (size_t)rl->id; r.saved_brlen = rl->saved_brlen;
  r.brlen_ratio = rl->brlen_ratio;
  return r;
}
void to_c(const root_location_t &r, rdamd_root_location_t *out) {
  out->edge = r.edge; out->id = r.id; out->saved_brlen = r.saved_brlen;
  out->brlen_ratio = r.brlen_ratio;
}
char *dup(const std::string &s) {
  char *r = (char *)malloc(s.size() + 1);
  if (r) memcpy(r, s.c_str(), s.size() + 1);
  return r;
}
int unpack(const rdamd::op_schedule_t &res, rdamd_operation_t *ops, unsigned *n_ops,
           unsigned *pmi, double *brl, unsigned *n_mat) {
  const auto &o = std::get<0>(res);
  const auto &p = std::get<1>(res);
  const auto &b = std::get<2>(res);
  for (size_t i = 0; i < o.size(); ++i) ops[i] = o[i];
  for (size_t i = 0; i < p.size(); ++i) { pmi[i] = p[i]; brl[i] = b[i]; }
  *n_ops = (unsigned)o.size();
  *n_mat = (unsigned)p.size();
  return RDAMD_SUCCESS;
}
}  // namespace

#define GUARD(...)                                      \
  try {                                                 \
    rdamd::clear_error();                               \
    __VA_ARGS__                                         \
  } catch (const std::exception &e) {                   \
    rdamd::set_error(20, "%s", e.what());               \
    return 0;                                           \
  }

extern "C" {

rdamd_tree_t *rdamd_tree_from_file(const char *filename) {
  GUARD({
    auto *t = new rdamd_tree{rdamd::rooted_tree_t::from_file(filename), {}};
    return t;
  })
}
rdamd_tree_t *rdamd_tree_from_newick(const char *newick) {
  GUARD({
    auto *t = new rdamd_tree{rdamd::rooted_tree_t::from_newick(newick), {}};
    return t;
  })
}
void rdamd_tree_destroy(rdamd_tree_t *t) { delete t; }
unsigned int rdamd_tree_tip_count(const rdamd_tree_t *t) { return t->tree.tip_count(); }
unsigned int rdamd_tree_inner_count(const rdamd_tree_t *t) { return t->tree.inner_count(); }
unsigned int rdamd_tree_branch_count(const rdamd_tree_t *t) { return t->tree.branch_count(); }
unsigned int rdamd_tree_root_count(const rdamd_tree_t *t) { return (unsigned)t->tree.root_count(); }
unsigned int rdamd_tree_root_clv_index(const rdamd_tree_t *t) { return t->tree.root_clv_index(); }
int rdamd_tree_root_scaler_index(const rdamd_tree_t *t) { return t->tree.root_scaler_index(); }

int rdamd_tree_root_location(const rdamd_tree_t *t, unsigned int index,
                             rdamd_root_location_t *out) {
  GUARD({ to_c(t->tree.root_location((size_t)index), out); return RDAMD_SUCCESS; })
}
int rdamd_tree_generate_directional_operations(const rdamd_tree_t *t, const double *ratios,
                                               rdamd_operation_t *ops, unsigned int *n_ops,
                                               unsigned int *matrix_indices,
                                               double *branch_lengths, unsigned int *n_matrices,
                                               unsigned int *root_clv, int *root_scaler,
                                               unsigned int sizes[3]) {
  GUARD({
    std::vector<double> r;
    if (ratios) r.assign(ratios, ratios + t->tree.root_count());
    const auto d = t->tree.generate_directional_operations(ratios ? &r : nullptr);
    std::copy(d.ops.begin(), d.ops.end(), ops);
    *n_ops = (unsigned)d.ops.size();
    std::copy(d.matrix_indices.begin(), d.matrix_indices.end(), matrix_indices);
    std::copy(d.branch_lengths.begin(), d.branch_lengths.end(), branch_lengths);
    *n_matrices = (unsigned)d.matrix_indices.size();
    std::copy(d.root_clv.begin(), d.root_clv.end(), root_clv);
    std::copy(d.root_scaler.begin(), d.root_scaler.end(), root_scaler);
    sizes[0] = d.clv_buffers; sizes[1] = d.scale_buffers; sizes[2] = d.prob_matrices;
    return RDAMD_SUCCESS;
  })
}
// rank_midpoints / rank_modified_mad: root ids, best first (root_count of them)
int rdamd_tree_rank_midpoints(const rdamd_tree_t *t, unsigned int *root_ids) {
  GUARD({
    const auto r = t->tree.rank_midpoints();
    for (size_t i = 0; i < r.size(); ++i) root_ids[i] = (unsigned)r[i].id;
    return RDAMD_SUCCESS;
  })
}
int rdamd_tree_rank_modified_mad(const rdamd_tree_t *t, unsigned int *root_ids) {
  GUARD({
    const auto r = t->tree.rank_modified_mad();
    for (size_t i = 0; i < r.size(); ++i) root_ids[i] = (unsigned)r[i].id;
    return RDAMD_SUCCESS;
  })
}
int rdamd_tree_root_location_by_label(const rdamd_tree_t *t, const char *label,
                                      rdamd_root_location_t *out) {
  GUARD({ to_c(t->tree.root_location(std::string(label)), out); return RDAMD_SUCCESS; })
}
const char *rdamd_tree_root_label(const rdamd_tree_t *t, unsigned int index) {
  GUARD({
    t->scratch = t->tree.label(t->tree.root_location((size_t)index));
    return t->scratch.c_str();
  })
}
int rdamd_tree_root_is_internal(const rdamd_tree_t *t, unsigned int index) {
  GUARD({ return t->tree.is_internal(t->tree.root_location((size_t)index)) ? 1 : 0; })
}
int rdamd_tree_tip_index(const rdamd_tree_t *t, const char *label) {
  auto m = t->tree.label_map();
  auto it = m.find(label);
  return it == m.end() ? -1 : (int)it->second;
}
const char *rdamd_tree_tip_label(const rdamd_tree_t *t, unsigned int clv_index) {
  GUARD({
    t->scratch = t->tree.tip_label(clv_index);
    return t->scratch.c_str();
  })
}
char *rdamd_tree_side_tips(const rdamd_tree_t *t, const rdamd_root_location_t *rl) {
  GUARD({
    std::string s;
    for (auto &l : t->tree.side_tips(to_cpp(rl))) { if (!s.empty()) s += "\n"; s += l; }
    return dup(s);
  })
}

int rdamd_tree_generate_operations(rdamd_tree_t *t, const rdamd_root_location_t *rl,
                                   rdamd_operation_t *ops, unsigned int *n_ops,
                                   unsigned int *pmi, double *brl, unsigned int *n_mat) {
  GUARD({ return unpack(t->tree.generate_operations(to_cpp(rl)), ops, n_ops, pmi, brl, n_mat); })
}
int rdamd_tree_generate_derivative_operations(rdamd_tree_t *t,
                                              const rdamd_root_location_t *rl,
                                              rdamd_operation_t *op, unsigned int *pmi,
                                              double *brl) {
  GUARD({
    auto res = t->tree.generate_derivative_operations(to_cpp(rl));
    *op = std::get<0>(res);
    for (int i = 0; i < 2; ++i) { pmi[i] = std::get<1>(res)[i]; brl[i] = std::get<2>(res)[i]; }
    return RDAMD_SUCCESS;
  })
}
int rdamd_tree_generate_root_update_operations(rdamd_tree_t *t,
                                               const rdamd_root_location_t *rl,
                                               rdamd_operation_t *ops, unsigned int *n_ops,
                                               unsigned int *pmi, double *brl,
                                               unsigned int *n_mat) {
  GUARD({
    return unpack(t->tree.generate_root_update_operations(to_cpp(rl)), ops, n_ops, pmi, brl, n_mat);
  })
}
int rdamd_tree_root_by(rdamd_tree_t *t, const rdamd_root_location_t *rl) {
  GUARD({ t->tree.root_by(to_cpp(rl)); return RDAMD_SUCCESS; })
}
void rdamd_tree_unroot(rdamd_tree_t *t) { t->tree.unroot(); }
int rdamd_tree_rooted(const rdamd_tree_t *t) { return t->tree.rooted() ? 1 : 0; }
int rdamd_tree_sanity_check(const rdamd_tree_t *t) { return t->tree.sanity_check() ? 1 : 0; }
char *rdamd_tree_newick(const rdamd_tree_t *t, int annotations) {
  GUARD({ return dup(t->tree.newick(annotations != 0)); })
}
int rdamd_tree_annotate_branch(rdamd_tree_t *t, const rdamd_root_location_t *rl,
                               const char *key, const char *value) {
  GUARD({ t->tree.annotate_branch(to_cpp(rl), key, value); return RDAMD_SUCCESS; })
}

int rdamd_tree_annotate_branch_lr(rdamd_tree_t *t, const rdamd_root_location_t *rl,
                                  const char *key, const char *left_value,
                                  const char *right_value) {
  GUARD({ t->tree.annotate_branch(to_cpp(rl), key, left_value, right_value); return RDAMD_SUCCESS; })
}

}  // extern "C"
