// C wrappers over model_t (declared in include/root_digger_amd.h).
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>

#include <algorithm>

#include "common.hpp"
#include "batch_combiner.hpp"
#include "checkpoint.hpp"
#include "lockstep_conductor.hpp"
#include "model.hpp"
#include "partition_info.hpp"
#include "tree_c_api.hpp"

struct rdamd_model {
  rdamd::model_t *model = nullptr;
  rdamd::msa_t    msa;                        // partition 0 (the only one unless partitioned)
  std::vector<rdamd::msa_t> more_msas;        // partitions 1.. of a partitioned model
  std::vector<rdamd::ratehet_opts_t> ratehets;   // per partition
  // what a replica needs (parallel exhaustive search)
  unsigned rate_cats = 1;
  std::vector<rdamd::msa_t> all_msas() const {
    std::vector<rdamd::msa_t> v{msa};
    v.insert(v.end(), more_msas.begin(), more_msas.end());
    return v;
  }
  std::vector<rdamd::ratehet_opts_t> all_ratehets() const {
    return ratehets.empty() ? std::vector<rdamd::ratehet_opts_t>{rdamd::ratehet_opts_t(rate_cats)}
                            : ratehets;
  }
  uint64_t seed = 0;
  bool     early_stop = false;
  void    *setulb = nullptr;
  rdamd::checkpoint_t *checkpoint = nullptr;
  std::unique_ptr<rdamd::model_t::progress_t> progress;   // set by rdamd_model_set_progress
  int lockstep_priority = 1;      // stream priority of the shared objective partition in a lock-stepped search
  bool children_only = true;      // rdamd_model_set_root_children_only
  unsigned lockstep_groups = 0;   // 0: the library's choice; 1: one group, blocking launches (rdamd_model_set_lockstep_groups)
  uint64_t lockstep_stats[4] = {0, 0, 0, 0};   // of the last lock-stepped search (rdamd_model_lockstep_stats)
  uint64_t round_stats[3] = {0, 0, 0};         // ... in rounds: rounds, collectives, second-pass redos
  double round_seconds[4] = {0, 0, 0, 0};      // ... and the host time of the rounds' phases
  struct async_reducer_t { rdamd_lnl_reducer_t queue; void *user; };   // rdamd_model_set_lnl_reducer_async
  std::unique_ptr<async_reducer_t> async_reducer;
  int lockstep_rounds = -1;       // -1: rounds for site-sharded models only; 0 never; 1 always (rdamd_model_set_lockstep_rounds)
  ~rdamd_model() { delete model; }
};

using rdamd::root_location_t;

rdamd::checkpoint_t *rdamd_checkpoint_cpp(rdamd_checkpoint_t *c);   // checkpoint_c_api.cpp

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
}  // namespace

#define GUARD(failret, ...)                             \
  try {                                                 \
    rdamd::clear_error();                               \
    __VA_ARGS__                                         \
  } catch (const std::exception &e) {                   \
    rdamd::set_error(50, "%s", e.what());               \
    return failret;                                     \
  }

extern "C" {

rdamd_model_t *rdamd_model_create(const rdamd_tree_t *tree, unsigned int n_taxa,
                                  const char *const *labels, const char *const *sequences,
                                  const unsigned int *weights, unsigned int states,
                                  const uint64_t *map, unsigned int rate_cats, uint64_t seed,
                                  int early_stop) {
  GUARD(nullptr, {
    auto *m = new rdamd_model();
    m->msa.states = states;
    m->msa.set_map(map);
    for (unsigned i = 0; i < n_taxa; ++i) {
      m->msa.labels.emplace_back(labels[i]);
      m->msa.sequences.emplace_back(sequences[i]);
    }
    if (weights) m->msa.weights.assign(weights, weights + m->msa.length());
    m->rate_cats = rate_cats; m->seed = seed; m->early_stop = early_stop != 0;
    try {
      m->model = new rdamd::model_t(rdamd_tree_cpp(tree), {m->msa},
                                    {rdamd::ratehet_opts_t(rate_cats)}, false, seed,
                                    early_stop != 0);
    } catch (...) {
      delete m;
      throw;
    }
    return m;
  })
}
static rdamd_model_t *create_from_file(const rdamd_tree_t *tree, const char *msa_filename,
                                       unsigned int states, const uint64_t *map,
                                       const rdamd::ratehet_opts_t &rc, uint64_t seed,
                                       int early_stop, int compress, unsigned int *n_patterns,
                                       unsigned int block = 0, unsigned int n_blocks = 1,
                                       unsigned int *n_columns = nullptr) {
  GUARD(nullptr, {
    auto *m = new rdamd_model();
    m->rate_cats = (unsigned)rc.rate_cats; m->seed = seed; m->early_stop = early_stop != 0;
    m->ratehets = {rc};
    try {
      if (n_blocks <= 1) {
        m->msa = rdamd::msa_t::from_file(msa_filename, map, states, compress != 0);
        if (n_columns) *n_columns = m->msa.total_weight();
      } else {   // one site block of a site-sharded run: cut first, compress the block
        if (block >= n_blocks) throw std::invalid_argument("site block index out of range");
        const rdamd::msa_t whole = rdamd::msa_t::from_file(msa_filename, map, states, false);
        const size_t len = whole.length(), size = len / n_blocks, mod = len % n_blocks;
        const size_t lo = size * block + std::min<size_t>(mod, block);           // src/model.cpp:1899-1907
        const size_t hi = size * (block + 1) + std::min<size_t>(mod, block + 1);
        if (lo == hi) throw std::invalid_argument("more site blocks than alignment columns");
        m->msa = whole.columns(lo, hi);
        if (compress) m->msa.compress();
        if (n_columns) *n_columns = (unsigned)len;
      }
      if (!m->msa.constiency_check(rdamd_tree_cpp(tree).label_set()))
        throw std::invalid_argument("Taxa on the tree and in the MSA are inconsistient");
      m->model = new rdamd::model_t(rdamd_tree_cpp(tree), {m->msa}, {rc}, false, seed,
                                    early_stop != 0);
    } catch (...) {
      delete m;
      throw;
    }
    if (n_patterns) *n_patterns = (unsigned)m->msa.length();
    return m;
  })
}
rdamd_model_t *rdamd_model_create_from_file(const rdamd_tree_t *tree, const char *msa_filename,
                                            unsigned int states, const uint64_t *map,
                                            unsigned int rate_cats, uint64_t seed,
                                            int early_stop, int compress,
                                            unsigned int *n_patterns) {
  return create_from_file(tree, msa_filename, states, map, rdamd::ratehet_opts_t(rate_cats), seed,
                          early_stop, compress, n_patterns);
}
rdamd_model_t *rdamd_model_create_from_file_ratehet(const rdamd_tree_t *tree,
                                                    const char *msa_filename,
                                                    unsigned int states, const uint64_t *map,
                                                    const rdamd_ratehet_opts_t *ratehet,
                                                    uint64_t seed, int early_stop, int compress,
                                                    unsigned int *n_patterns) {
  rdamd::ratehet_opts_t rc(ratehet->rate_cats ? ratehet->rate_cats : 1);
  rc.type = (rdamd::param_type)ratehet->type;
  rc.rate_category_type = (rdamd::rate_category)ratehet->rate_category_type;
  rc.alpha_init = ratehet->alpha_init != 0;
  rc.alpha = ratehet->alpha;
  return create_from_file(tree, msa_filename, states, map, rc, seed, early_stop, compress,
                          n_patterns);
}
rdamd_model_t *rdamd_model_create_from_file_block(const rdamd_tree_t *tree, const char *msa_filename,
                                                  unsigned int states, const uint64_t *map,
                                                  const rdamd_ratehet_opts_t *ratehet,
                                                  uint64_t seed, int early_stop, int compress,
                                                  unsigned int block, unsigned int n_blocks,
                                                  unsigned int *n_patterns,
                                                  unsigned int *n_columns) {
  rdamd::ratehet_opts_t rc(ratehet->rate_cats ? ratehet->rate_cats : 1);
  rc.type = (rdamd::param_type)ratehet->type;
  rc.rate_category_type = (rdamd::rate_category)ratehet->rate_category_type;
  rc.alpha_init = ratehet->alpha_init != 0;
  rc.alpha = ratehet->alpha;
  return create_from_file(tree, msa_filename, states, map, rc, seed, early_stop, compress,
                          n_patterns, block, n_blocks, n_columns);
}
int rdamd_model_set_lnl_reducer(rdamd_model_t *m, rdamd_lnl_reducer_t reduce, void *user,
                                int on_device) {
  GUARD(RDAMD_FAILURE, {
    m->model->set_lnl_reducer(reduce, user, on_device != 0);
    // the ready-made RCCL reducer comes with its two halves
    if (reduce == rdamd_comm_reducer && on_device) {
      m->model->set_lnl_reducer_async(rdamd_comm_reducer_queue, rdamd_comm_reducer_wait, user);
      m->model->set_lnl_reducer_abort([](void *u) { rdamd_comm_abort((rdamd_comm_t *)u); }, user);
    }
    return RDAMD_SUCCESS;
  })
}
int rdamd_model_set_lnl_reducer_abort(rdamd_model_t *m, rdamd_lnl_abort_t abort, void *user) {
  GUARD(RDAMD_FAILURE, {
    if (!m->model->site_sharded()) throw std::invalid_argument("set_lnl_reducer_abort: set the reducer first");
    m->model->set_lnl_reducer_abort(abort, user);
    return RDAMD_SUCCESS;
  })
}
// the blocking form of a two-halves reducer: queue, then wait for the stream
static int blocking_from_async(double *values, unsigned int n, void *stream, void *user) {
  auto *a = (rdamd_model::async_reducer_t *)user;
  if (a->queue(values, n, stream, a->user) != RDAMD_SUCCESS) return RDAMD_FAILURE;
  return hipStreamSynchronize((hipStream_t)stream) == hipSuccess ? RDAMD_SUCCESS : RDAMD_FAILURE;
}
int rdamd_model_set_lnl_reducer_async(rdamd_model_t *m, rdamd_lnl_reducer_t queue, rdamd_lnl_wait_t wait,
                                      void *user) {
  GUARD(RDAMD_FAILURE, {
    if (!queue || !wait) throw std::invalid_argument("set_lnl_reducer_async: both halves are required");
    m->async_reducer.reset(new rdamd_model::async_reducer_t{queue, user});
    m->model->set_lnl_reducer(blocking_from_async, m->async_reducer.get(), true);
    m->model->set_lnl_reducer_async(queue, wait, user);
    return RDAMD_SUCCESS;
  })
}
// The reference's partitioned set-up (src/main.cpp:512-555): the alignment is read
// whole, cut into the partition file's column ranges, each partition compressed
// on its own; rate categories come from each partition's model string.
rdamd_model_t *rdamd_model_create_partitioned(const rdamd_tree_t *tree, const char *msa_filename,
                                              const char *partition_filename,
                                              unsigned int states, const uint64_t *map,
                                              uint64_t seed, int early_stop,
                                              unsigned int *n_partitions) {
  GUARD(nullptr, {
    auto *m = new rdamd_model();
    m->seed = seed; m->early_stop = early_stop != 0;
    try {
      const rdamd::msa_t whole = rdamd::msa_t::from_file(msa_filename, map, states, false);
      const auto infos = rdamd::parse_partition_file(partition_filename);
      if (infos.empty()) throw std::runtime_error("The partition file holds no partitions");
      auto msas = rdamd::partition_msa(whole, infos, true);
      for (const auto &pi : infos) {
        rdamd::ratehet_opts_t rc = pi.model.ratehet_opts;
        if (rc.rate_cats == 0) rc.rate_cats = 1;
        m->ratehets.push_back(rc);
      }
      for (const auto &x : msas)
        if (!x.constiency_check(rdamd_tree_cpp(tree).label_set()))
          throw std::invalid_argument("Taxa on the tree and in the MSA are inconsistient");
      m->rate_cats = (unsigned)m->ratehets[0].rate_cats;
      m->msa = msas[0];
      m->more_msas.assign(msas.begin() + 1, msas.end());
      m->model = new rdamd::model_t(rdamd_tree_cpp(tree), msas, m->ratehets, false, seed,
                                    early_stop != 0);
    } catch (...) {
      delete m;
      throw;
    }
    if (n_partitions) *n_partitions = (unsigned)(1 + m->more_msas.size());
    return m;
  })
}

namespace {
void fill_model(const rdamd::model_info_t &mi, rdamd_partition_info_t *out) {
  std::snprintf(out->subst_str, sizeof out->subst_str, "%s", mi.subst_str.c_str());
  out->freq_type = (int)mi.freq_opts.type;
  out->invar_present = mi.invar_opts.present;
  out->invar_type = (int)mi.invar_opts.type;
  out->invar_user_prop = mi.invar_opts.user_prop;
  out->ratehet = {(int32_t)mi.ratehet_opts.type, (int32_t)mi.ratehet_opts.rate_category_type,
                  mi.ratehet_opts.rate_cats, mi.ratehet_opts.alpha_init ? 1 : 0,
                  mi.ratehet_opts.alpha};
  out->asc_present = mi.asc_opts.present;
  out->asc_type = (int)mi.asc_opts.type;
  out->asc_fels_weight = mi.asc_opts.fels_weight;
  out->n_stam_weights = (unsigned)std::min<size_t>(mi.asc_opts.stam_weights.size(), 32);
  for (unsigned i = 0; i < out->n_stam_weights; ++i) out->stam_weights[i] = mi.asc_opts.stam_weights[i];
}
}  // namespace

int rdamd_parse_model_info(const char *model_string, rdamd_partition_info_t *out) {
  GUARD(RDAMD_FAILURE, {
    std::memset(out, 0, sizeof *out);
    fill_model(rdamd::parse_model_info(model_string), out);
    std::snprintf(out->model_name, sizeof out->model_name, "%s", model_string);
    return RDAMD_SUCCESS;
  })
}
int rdamd_parse_partition_info(const char *line, rdamd_partition_info_t *out) {
  GUARD(RDAMD_FAILURE, {
    std::memset(out, 0, sizeof *out);
    const auto pi = rdamd::parse_partition_info(line);
    fill_model(pi.model, out);
    std::snprintf(out->model_name, sizeof out->model_name, "%s", pi.model_name.c_str());
    std::snprintf(out->partition_name, sizeof out->partition_name, "%s", pi.partition_name.c_str());
    if (pi.parts.size() > 64) throw std::runtime_error("more than 64 ranges in one partition line");
    out->n_ranges = (unsigned)pi.parts.size();
    for (size_t i = 0; i < pi.parts.size(); ++i) {
      out->ranges[i][0] = pi.parts[i].first;
      out->ranges[i][1] = pi.parts[i].second;
    }
    return RDAMD_SUCCESS;
  })
}
// msa_t::partition on a file: lengths (patterns if compress, columns otherwise) and
// total weights of the partitions the given lines describe
int rdamd_msa_partition_probe(const char *msa_filename, const uint64_t *map,
                              unsigned int n_lines, const char *const *lines, int compress,
                              unsigned int *lengths, unsigned int *total_weights) {
  GUARD(RDAMD_FAILURE, {
    const rdamd::msa_t whole = rdamd::msa_t::from_file(msa_filename, map, 4, false);
    rdamd::msa_partitions_t infos;
    for (unsigned i = 0; i < n_lines; ++i) infos.push_back(rdamd::parse_partition_info(lines[i]));
    const auto msas = rdamd::partition_msa(whole, infos, compress != 0);
    for (size_t i = 0; i < msas.size(); ++i) {
      if (lengths) lengths[i] = (unsigned)msas[i].length();
      if (total_weights) total_weights[i] = msas[i].total_weight();
    }
    return RDAMD_SUCCESS;
  })
}
int rdamd_model_partition_count(const rdamd_model_t *m) { return (int)(1 + m->more_msas.size()); }

int rdamd_msa_probe(const char *msa_filename, const uint64_t *map, int compress,
                    unsigned int *n_taxa, unsigned int *n_patterns,
                    unsigned int *total_weight) {
  GUARD(RDAMD_FAILURE, {
    rdamd::msa_t m = rdamd::msa_t::from_file(msa_filename, map, 4, compress != 0);
    if (n_taxa) *n_taxa = (unsigned)m.count();
    if (n_patterns) *n_patterns = (unsigned)m.length();
    if (total_weight) *total_weight = m.total_weight();
    return RDAMD_SUCCESS;
  })
}
void rdamd_model_destroy(rdamd_model_t *m) { delete m; }

int rdamd_model_initialize_partitions(rdamd_model_t *m, int uniform_freqs) {
  GUARD(RDAMD_FAILURE, {
    if (uniform_freqs) m->model->initialize_partitions_uniform_freqs(m->all_msas());
    else m->model->initialize_partitions(m->all_msas());
    return RDAMD_SUCCESS;
  })
}
int rdamd_model_set_subst_rates(rdamd_model_t *m, const double *rates) {
  GUARD(RDAMD_FAILURE, {
    unsigned k = m->msa.states;
    m->model->set_subst_rates(0, rdamd::model_params_t(rates, rates + k * k - k));
    return RDAMD_SUCCESS;
  })
}
int rdamd_model_set_subst_rates_uniform(rdamd_model_t *m) {
  GUARD(RDAMD_FAILURE, { m->model->set_subst_rates_uniform(); return RDAMD_SUCCESS; })
}
int rdamd_model_set_freqs(rdamd_model_t *m, const double *freqs) {
  GUARD(RDAMD_FAILURE, {
    m->model->set_freqs(0, rdamd::model_params_t(freqs, freqs + m->msa.states));
    return RDAMD_SUCCESS;
  })
}
int rdamd_model_set_empirical_freqs(rdamd_model_t *m) {
  GUARD(RDAMD_FAILURE, { m->model->set_empirical_freqs(); return RDAMD_SUCCESS; })
}
int rdamd_model_set_gamma_alpha(rdamd_model_t *m, double alpha) {
  GUARD(RDAMD_FAILURE, { m->model->set_gamma_rates(0, {alpha}); return RDAMD_SUCCESS; })
}
double rdamd_model_compute_lh(rdamd_model_t *m, const rdamd_root_location_t *rl) {
  GUARD(std::nan(""), { return m->model->compute_lh(to_cpp(rl)); })
}
double rdamd_model_compute_lh_root(rdamd_model_t *m, const rdamd_root_location_t *rl) {
  GUARD(std::nan(""), { return m->model->compute_lh_root(to_cpp(rl)); })
}
int rdamd_model_compute_dlh(rdamd_model_t *m, const rdamd_root_location_t *rl, double out[2]) {
  GUARD(RDAMD_FAILURE, {
    auto d = m->model->compute_dlh(to_cpp(rl));
    out[0] = d.lh; out[1] = d.dlh;
    return RDAMD_SUCCESS;
  })
}
int rdamd_model_move_root(rdamd_model_t *m, const rdamd_root_location_t *rl) {
  GUARD(RDAMD_FAILURE, { m->model->move_root(to_cpp(rl)); return RDAMD_SUCCESS; })
}
int rdamd_model_compute_all_root_lh(rdamd_model_t *m, double *out) {
  GUARD(RDAMD_FAILURE, {
    auto v = m->model->compute_all_root_lh();
    for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
    return RDAMD_SUCCESS;
  })
}
int rdamd_model_compute_all_root_lh_batched(rdamd_model_t *m, double *out) {
  GUARD(RDAMD_FAILURE, {
    auto v = m->model->compute_all_root_lh_batched();
    for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
    return RDAMD_SUCCESS;
  })
}
int rdamd_model_compute_all_root_lh_directional(rdamd_model_t *m, const double *ratios, double *out) {
  GUARD(RDAMD_FAILURE, {
    std::vector<double> r;
    if (ratios) r.assign(ratios, ratios + m->model->tree().root_count());
    const auto v = m->model->compute_all_root_lh_directional(ratios ? &r : nullptr);
    std::copy(v.begin(), v.end(), out);
    return RDAMD_SUCCESS;
  })
}
int rdamd_model_search(rdamd_model_t *m, unsigned int min_roots, double root_ratio, double atol,
                       double pgtol, double brtol, double factor,
                       rdamd_root_location_t *best_rl, double *best_llh) {
  GUARD(RDAMD_FAILURE, {
    auto best = m->model->search(min_roots, root_ratio, atol, pgtol, brtol, factor, nullptr);
    if (best_rl) to_c(best.first, best_rl);
    if (best_llh) *best_llh = best.second;
    return RDAMD_SUCCESS;
  })
}
int rdamd_model_optimize_alpha(rdamd_model_t *m, const rdamd_root_location_t *rl, double atol,
                               rdamd_root_location_t *out) {
  GUARD(RDAMD_FAILURE, { to_c(m->model->optimize_alpha(to_cpp(rl), atol), out); return RDAMD_SUCCESS; })
}
int rdamd_model_compute_lh_batch(rdamd_model_t *m, unsigned int n,
                                 const rdamd_root_location_t *rls, const double *subst,
                                 const double *freqs, const double *gamma_alpha, double *out) {
  GUARD(RDAMD_FAILURE, {
    const unsigned k = m->msa.states, np = k * k - k;
    std::vector<root_location_t> roots;
    std::vector<std::vector<rdamd::partition_parameters_t>> params(n);
    for (unsigned j = 0; j < n; ++j) {
      roots.push_back(to_cpp(&rls[j]));
      rdamd::partition_parameters_t pp;
      pp.subst_rates.assign(subst + (size_t)j * np, subst + (size_t)(j + 1) * np);
      pp.freqs.assign(freqs + (size_t)j * k, freqs + (size_t)(j + 1) * k);
      pp.gamma_alpha.assign(1, gamma_alpha ? gamma_alpha[j] : 1.0);
      params[j].push_back(pp);
    }
    auto v = m->model->compute_lh_batch(roots, params);
    for (unsigned j = 0; j < n; ++j) out[j] = v[j];
    return RDAMD_SUCCESS;
  })
}
void rdamd_model_set_lbfgsb(rdamd_model_t *m, void *fn) {
  m->setulb = fn;
  m->model->set_lbfgsb(reinterpret_cast<rdamd::model_t::setulb_fn>(fn));
}

// The candidate-root loop of exhaustive_search (src/model.cpp:1154) is
// embarrassingly parallel: the reference runs it on one MPI rank per chunk
// (src/model.cpp:1867-1911).  On one GPU the same thing is `workers` host
// threads, each with its own model replica (own partition, own HIP stream),
// pulling candidates from a shared counter -- their small launches (13-job
// L-BFGS-B batches, root-only Brent steps) overlap on the device.
// replicas keep only what the root-only steps read when the searches' compute_lh is the
// children-only one (rdamd_model_set_root_children_only, the default)
static bool replicas_are_sparse(const rdamd_model_t *m) { return m->children_only; }

unsigned int rdamd_model_max_replicas(const rdamd_model_t *m, unsigned int requested,
                                      uint64_t *replica_bytes) {
  const unsigned tips = m->model->tree().tip_count(), branches = m->model->tree().branch_count();
  uint64_t per_replica = 0;
  const auto ratehets = m->all_ratehets();
  const auto msas = m->all_msas();
  for (size_t i = 0; i < msas.size(); ++i) {
    // a replica's 4-state / binary partitions hold the root's two children and the root CLV, not
    // all 2n - 3 buffers (RDAMD_ATTRIB_SPARSE_CLVS; their pools start at four slots) -- plus the
    // evaluator's workspace for the one job that writes them
    const unsigned R = (unsigned)ratehets[std::min(i, ratehets.size() - 1)].rate_cats;
    const bool sparse = replicas_are_sparse(m) &&
                        (msas[i].states == 4 || msas[i].states == 2 || (msas[i].states == 20 && R <= 8));
    per_replica += rdamd_partition_footprint(tips, sparse ? 4u : branches, msas[i].states, (unsigned)msas[i].length(),
                                             branches, R, sparse ? 4u : branches);
    if (sparse) {
      const uint64_t K = msas[i].states == 20 ? 20 : 4;
      per_replica += (uint64_t)branches * R * (K * K + (K == 4 ? 64 : 64 * 24)) * 8 * 2 +
                     (uint64_t)msas[i].length() * R * 8 + ((uint64_t)2 << 20);
    }
  }
  if (replica_bytes) *replica_bytes = per_replica;
  uint64_t free_b = 0, total_b = 0;
  if (rdamd_device_memory(&free_b, &total_b) != RDAMD_SUCCESS || per_replica == 0) return requested ? requested : 1;
  const uint64_t fit = (uint64_t)(0.85 * (double)free_b) / per_replica;
  return (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(requested, fit));
}

// The shared objective partition(s) on LOW stream priority for the duration of a lock-stepped
// search: their launches fill every CU for a millisecond, and the kernels beside them (front
// halves of the other group's batch, the replicas' root-only steps and traversals) get the wave
// slots that become free instead of waiting for the launch to end.  Normal priority again when
// the search is over, however it ends (the stream is re-created: rdamd_partition_stream handles
// taken before are invalid, as the header says).
struct shared_priority_t {
  rdamd_model_t *m;
  bool active;
  shared_priority_t(rdamd_model_t *m_, bool lockstep) : m(m_), active(lockstep && m_->lockstep_priority != 0) {
    for (size_t pi = 0; active && pi < m->model->partition_count(); ++pi)
      if (rdamd_partition_set_stream_priority(m->model->partition(pi), m->lockstep_priority) != RDAMD_SUCCESS) {
        // (a constructor that throws has no destructor: put back what was switched so far)
        const std::string why = rdamd_errmsg();
        for (size_t pj = 0; pj < pi; ++pj) (void)rdamd_partition_set_stream_priority(m->model->partition(pj), 0);
        throw std::runtime_error("set_stream_priority: " + why);
      }
  }
  ~shared_priority_t() {
    for (size_t pi = 0; active && pi < m->model->partition_count(); ++pi)
      (void)rdamd_partition_set_stream_priority(m->model->partition(pi), 0);
  }
};

// Lock step in deterministic rounds (lockstep_conductor.hpp): what a site-sharded model's
// lock-stepped search is -- every rank of the site group runs this with the same candidates and
// forms the same rounds, one collective each.  Worker w is a host thread with a model replica
// (sparse: the root's children only); the objective partition is m's own.
static int search_in_rounds(rdamd_model_t *m, unsigned int workers, double atol, double pgtol, double brtol,
                            double factor, uint64_t *root_id, double *llh, double *alpha,
                            unsigned int *n_results, rdamd_root_location_t *best_rl, double *best_llh) {
  GUARD(RDAMD_FAILURE, {
    if (m->model->partition_count() != 1)
      throw std::runtime_error("lock step in rounds takes single-partition models");
    const std::vector<size_t> todo = m->model->assigned_indicies();
    if (workers < 1) workers = 1;
    workers = (unsigned)std::min<size_t>(workers, std::max<size_t>(todo.size(), 1));
    {
      // How many replicas fit is each rank's own finding (its free device memory), and every rank
      // of a site group must run the SAME number of candidates in flight: a different count on
      // one rank means other rounds, other vector lengths, collectives that do not match.  So
      // the group agrees before it starts -- through the reducer, like everything else:
      // [1, requested, requested^2, fit, fit^2] summed; all ranks equal <=> sum == G x own for
      // the value AND its square (then sum (x_i - own)^2 = 0), which every rank decides alike.
      uint64_t bytes = 0;
      const unsigned requested = workers;
      unsigned fit = rdamd_model_max_replicas(m, workers, &bytes);
      if (m->model->site_sharded()) {
        double v[5] = {1.0, (double)requested, (double)requested * requested, (double)fit, (double)fit * fit};
        m->model->sum_over_site_group(v, 5);
        const double G = v[0];
        if (v[1] != G * requested || v[2] != G * requested * requested)
          throw std::runtime_error("lock step in rounds: the ranks of the site group were asked for different numbers of "
                                   "candidates in flight (this rank: " + std::to_string(requested) + ")");
        if (v[3] != G * fit || v[4] != G * (double)fit * fit) {
          // they differ: everybody runs the smallest.  u[i] = 1 while i < fit; the sum is G exactly
          // where every rank still fits
          std::vector<double> u(requested, 0.0);
          for (unsigned i = 0; i < fit && i < requested; ++i) u[i] = 1.0;
          m->model->sum_over_site_group(u.data(), u.size());
          unsigned least = 0;
          while (least < requested && u[least] == G) ++least;
          std::fprintf(stderr, "rdamd: the ranks of the site group fit different numbers of replicas (this rank: %u of %.2f GB "
                               "each); all of them run %u candidates in flight\n", fit, (double)bytes / 1e9, std::max(least, 1u));
          fit = std::max(least, 1u);
        }
      }
      if (fit < workers) {
        std::fprintf(stderr, "rdamd: %u replicas of %.2f GB each do not fit the free device memory; running %u\n",
                     workers, (double)bytes / 1e9, fit);
        workers = fit;
      }
    }
    shared_priority_t shared_priority(m, true);
    rdamd::conductor_t::config_t cfg;
    cfg.shared = m->model->partition(0);
    cfg.n_workers = workers;
    // One worker group or two alternating ones (rdamd_model_set_lockstep_groups)?  Two hide the
    // hosts' steps behind the other group's launch; one makes every launch twice as large and
    // HALVES the collectives.  Measured on one GPU (profiles/r5_shard_search.txt): an unsharded c2
    // gains from two; c2 / 8's shard is faster with one (0.201 against 0.218 s per candidate, 244
    // against 397 collectives per candidate), the deep shards are evaluator-bound either way --
    // and on real links every round pays the collective's latency.  So: a site-sharded model
    // defaults to ONE group, an unsharded one to two.
    const unsigned want_groups = m->lockstep_groups ? m->lockstep_groups : (m->model->site_sharded() ? 1u : 2u);
    cfg.n_groups = workers >= 4 && want_groups >= 2 ? 2u : 1u;
    cfg.n_candidates = todo.size();
    const auto red = m->model->reducer();
    cfg.reduce = red.reduce; cfg.device = red.device; cfg.queue = red.queue; cfg.wait = red.wait;
    cfg.user = red.user; cfg.async_user = red.async_user;
    cfg.abort = red.abort; cfg.abort_user = red.abort_user;
    rdamd::conductor_t conductor(cfg);
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) throw std::runtime_error("no HIP device");
    std::mutex mu;
    std::vector<rdamd::rd_result_t> results;
    std::string first_error;
    auto work = [&](unsigned wid) {
      try {
        if (hipSetDevice(device) != hipSuccess) throw std::runtime_error("hipSetDevice failed");
        const auto msas = m->all_msas();
        const bool sparse = replicas_are_sparse(m);
        rdamd::model_t replica(m->model->tree(), msas, m->all_ratehets(), false, m->seed + wid, m->early_stop, sparse);
        // (no collective of its own: the group's frequencies come from the parent, every other
        // sum goes through the rounds)
        replica.adopt_empirical_freqs(*m->model);
        replica.initialize_partitions(msas);
        if (m->setulb) replica.set_lbfgsb(reinterpret_cast<rdamd::model_t::setulb_fn>(m->setulb));
        replica.set_checkpoint(m->checkpoint);
        replica.set_progress(m->progress.get());
        replica.set_root_children_only(m->children_only);
        if (m->lockstep_priority)
          for (size_t pi = 0; pi < replica.partition_count(); ++pi)
            if (rdamd_partition_set_stream_priority(replica.partition(pi), -1) != RDAMD_SUCCESS)
              throw std::runtime_error(std::string("set_stream_priority: ") + rdamd_errmsg());
        if (!sparse) replica.initialize();
        replica.set_conductor(&conductor, wid);
        for (;;) {
          const long k = conductor.next_candidate(wid);
          if (k < 0) break;
          replica.assign_indicies(std::vector<size_t>{todo[(size_t)k]});
          std::vector<rdamd::rd_result_t> r;
          replica.exhaustive_search(atol, pgtol, brtol, factor, &r);
          std::lock_guard<std::mutex> g(mu);
          results.insert(results.end(), r.begin(), r.end());
        }
      } catch (const std::exception &e) {
        conductor.fail(e.what());
        std::lock_guard<std::mutex> g(mu);
        if (first_error.empty()) first_error = e.what();
      }
    };
    std::vector<std::thread> pool;
    for (unsigned w = 0; w < workers; ++w) pool.emplace_back(work, w);
    for (auto &t : pool) t.join();
    const auto st = conductor.stats();
    m->lockstep_stats[0] = st.obj_launches; m->lockstep_stats[1] = st.obj_jobs;
    m->lockstep_stats[2] = st.root_launches; m->lockstep_stats[3] = st.root_steps;
    m->round_stats[0] = st.rounds; m->round_stats[1] = st.collectives; m->round_stats[2] = st.redos;
    for (int k = 0; k < 4; ++k) m->round_seconds[k] = st.seconds[k];
    if (!first_error.empty()) throw std::runtime_error(first_error);
    std::sort(results.begin(), results.end(),
              [](const rdamd::rd_result_t &a, const rdamd::rd_result_t &b) { return a.root_id < b.root_id; });
    double bl = -INFINITY;
    for (size_t i = 0; i < results.size(); ++i) {
      root_id[i] = results[i].root_id; llh[i] = results[i].llh; alpha[i] = results[i].alpha;
      if (results[i].llh > bl) {
        bl = results[i].llh;
        if (best_rl) {
          auto rl = m->model->tree().root_location(results[i].root_id);
          rl.brlen_ratio = results[i].alpha;
          to_c(rl, best_rl);
        }
      }
    }
    *n_results = (unsigned)results.size();
    if (best_llh) *best_llh = bl;
    return RDAMD_SUCCESS;
  })
}

static int search_with_replicas(rdamd_model_t *m, unsigned int workers, bool lockstep, double atol,
                                double pgtol, double brtol, double factor, uint64_t *root_id,
                                double *llh, double *alpha, unsigned int *n_results,
                                rdamd_root_location_t *best_rl, double *best_llh) {
  GUARD(RDAMD_FAILURE, {
    // lock-step: the replicas' objective batches meet in one launch on THIS
    // model's partition (batch_combiner.hpp); it does nothing else meanwhile
    if (m->model->site_sharded())
      throw std::runtime_error("the candidates of a site-sharded model advance in rounds "
                               "(rdamd_model_exhaustive_search_lockstep); free-running replicas would "
                               "reorder the site group's collectives");
    std::vector<std::unique_ptr<rdamd::batch_combiner_t>> combiner[2];   // [group][partition]
    std::unique_ptr<rdamd::root_combiner_t> root_combiner;
    const std::vector<size_t> todo = m->model->assigned_indicies();
    if (workers < 1) workers = 1;
    workers = (unsigned)std::min<size_t>(workers, std::max<size_t>(todo.size(), 1));
    {   // every replica is a full model (all CLV buffers): keep them inside the device memory
      uint64_t bytes = 0;
      const unsigned fit = rdamd_model_max_replicas(m, workers, &bytes);
      if (fit < workers) {
        std::fprintf(stderr, "rdamd: %u replicas of %.1f GB each do not fit the free device memory; "
                             "running %u\n", workers, (double)bytes / 1e9, fit);
        workers = fit;
      }
    }
    // From four candidates in flight on they form two groups whose batches alternate on the
    // shared partition (batch_combiner.hpp): one group's hosts work while the other's batch runs
    const unsigned n_groups = lockstep && workers >= 4 && m->lockstep_groups != 1 ? 2u : 1u;
    shared_priority_t shared_priority(m, lockstep);
    if (lockstep) {
      for (unsigned g = 0; g < n_groups; ++g)
        for (size_t pi = 0; pi < m->model->partition_count(); ++pi)
          combiner[g].emplace_back(new rdamd::batch_combiner_t(m->model->partition(pi), n_groups == 2 ? (int)g : -1));
      root_combiner.reset(new rdamd::root_combiner_t());
    }
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) throw std::runtime_error("no HIP device");
    std::atomic<size_t> next{0};
    std::mutex mu;
    std::vector<rdamd::rd_result_t> results;
    std::string first_error;
    auto work = [&](unsigned wid) {
      try {
        if (hipSetDevice(device) != hipSuccess) throw std::runtime_error("hipSetDevice failed");
        const auto msas = m->all_msas();
        const bool sparse = replicas_are_sparse(m);
        rdamd::model_t replica(m->model->tree(), msas, m->all_ratehets(), false, m->seed + wid,
                               m->early_stop, sparse);
        replica.adopt_empirical_freqs(*m->model);
        replica.initialize_partitions(msas);
        if (m->setulb) replica.set_lbfgsb(reinterpret_cast<rdamd::model_t::setulb_fn>(m->setulb));
        replica.set_checkpoint(m->checkpoint);
        replica.set_progress(m->progress.get());
        replica.set_root_children_only(m->children_only);
        {
          std::vector<rdamd::batch_combiner_t *> mine;
          for (auto &c : combiner[wid % n_groups]) mine.push_back(c.get());
          replica.set_combiners(mine);
        }
        if (lockstep && m->lockstep_priority)   // (the replicas' short kernels in front of the shared partition's long ones)
          for (size_t pi = 0; pi < replica.partition_count(); ++pi)
            if (rdamd_partition_set_stream_priority(replica.partition(pi), -1) != RDAMD_SUCCESS)
              throw std::runtime_error(std::string("set_stream_priority: ") + rdamd_errmsg());
        // (model_t::initialize is a full traversal whose CLVs the search never reads: a sparse
        // replica, whose first step writes the two it needs, does without)
        if (!sparse) replica.initialize();
        replica.set_root_combiner(root_combiner.get());   // (after initialize(): that evaluates on its own)
        for (;;) {
          const size_t k = next.fetch_add(1);
          if (k >= todo.size()) break;
          replica.assign_indicies(std::vector<size_t>{todo[k]});
          std::vector<rdamd::rd_result_t> r;
          replica.exhaustive_search(atol, pgtol, brtol, factor, &r);
          std::lock_guard<std::mutex> g(mu);
          results.insert(results.end(), r.begin(), r.end());
        }
      } catch (const std::exception &e) {
        std::lock_guard<std::mutex> g(mu);
        if (first_error.empty()) first_error = e.what();
        next.store(todo.size());
      }
    };
    std::vector<std::thread> pool;
    for (unsigned w = 0; w < workers; ++w) pool.emplace_back(work, w);
    for (auto &t : pool) t.join();
    if (lockstep) {
      m->lockstep_stats[0] = m->lockstep_stats[1] = 0;
      for (unsigned g = 0; g < n_groups; ++g)
        for (auto &c : combiner[g]) {
          m->lockstep_stats[0] += c->launches();
          m->lockstep_stats[1] += c->jobs();
        }
      m->lockstep_stats[2] = root_combiner->launches(); m->lockstep_stats[3] = root_combiner->steps();
    }
    if (!first_error.empty()) throw std::runtime_error(first_error);
    std::sort(results.begin(), results.end(),
              [](const rdamd::rd_result_t &a, const rdamd::rd_result_t &b) { return a.root_id < b.root_id; });
    double bl = -INFINITY;
    for (size_t i = 0; i < results.size(); ++i) {
      root_id[i] = results[i].root_id; llh[i] = results[i].llh; alpha[i] = results[i].alpha;
      if (results[i].llh > bl) {
        bl = results[i].llh;
        if (best_rl) {
          auto rl = m->model->tree().root_location(results[i].root_id);
          rl.brlen_ratio = results[i].alpha;
          to_c(rl, best_rl);
        }
      }
    }
    *n_results = (unsigned)results.size();
    if (best_llh) *best_llh = bl;
    return RDAMD_SUCCESS;
  })
}
int rdamd_model_exhaustive_search_parallel(rdamd_model_t *m, unsigned int workers, double atol,
                                           double pgtol, double brtol, double factor,
                                           uint64_t *root_id, double *llh, double *alpha,
                                           unsigned int *n_results,
                                           rdamd_root_location_t *best_rl, double *best_llh) {
  return search_with_replicas(m, workers, false, atol, pgtol, brtol, factor, root_id, llh, alpha,
                              n_results, best_rl, best_llh);
}
int rdamd_model_exhaustive_search_lockstep(rdamd_model_t *m, unsigned int in_flight, double atol,
                                           double pgtol, double brtol, double factor,
                                           uint64_t *root_id, double *llh, double *alpha,
                                           unsigned int *n_results,
                                           rdamd_root_location_t *best_rl, double *best_llh) {
  // a site-sharded model's candidates meet in deterministic rounds (the ranks of its site group
  // must form the same launches and collectives); others in arrival order, unless asked
  const bool rounds = m->lockstep_rounds < 0 ? m->model->site_sharded() : m->lockstep_rounds != 0;
  if (rounds)
    return search_in_rounds(m, in_flight, atol, pgtol, brtol, factor, root_id, llh, alpha, n_results, best_rl, best_llh);
  return search_with_replicas(m, in_flight, true, atol, pgtol, brtol, factor, root_id, llh, alpha,
                              n_results, best_rl, best_llh);
}
int rdamd_model_optimize_params(rdamd_model_t *m, const rdamd_root_location_t *rl, double pgtol,
                                double factor, int optimize_gamma, double *subst, double *freqs,
                                double *gamma_alpha, uint64_t *n_batches,
                                uint64_t *n_evaluations) {
  GUARD(RDAMD_FAILURE, {
    const unsigned k = m->msa.states;
    std::vector<rdamd::partition_parameters_t> params(1);
    params[0].subst_rates.assign(subst, subst + k * k - k);
    params[0].freqs.assign(freqs, freqs + k);
    params[0].gamma_alpha.assign(1, *gamma_alpha);
    const size_t b0 = m->model->objective_batches(), e0 = m->model->objective_evaluations();
    m->model->optimize_params(params, to_cpp(rl), pgtol, factor, optimize_gamma != 0);
    std::copy(params[0].subst_rates.begin(), params[0].subst_rates.end(), subst);
    std::copy(params[0].freqs.begin(), params[0].freqs.end(), freqs);
    *gamma_alpha = params[0].gamma_alpha[0];
    if (n_batches) *n_batches = m->model->objective_batches() - b0;
    if (n_evaluations) *n_evaluations = m->model->objective_evaluations() - e0;
    return RDAMD_SUCCESS;
  })
}
void rdamd_model_set_lockstep_groups(rdamd_model_t *m, unsigned int groups) { m->lockstep_groups = groups; }
void rdamd_model_set_lockstep_priority(rdamd_model_t *m, int level) { m->lockstep_priority = level; }
void rdamd_model_set_root_children_only(rdamd_model_t *m, int on) {
  m->children_only = on != 0;
  if (m->model) m->model->set_root_children_only(on != 0);
}
void rdamd_model_lockstep_stats(const rdamd_model_t *m, uint64_t out[4]) {
  for (int i = 0; i < 4; ++i) out[i] = m->lockstep_stats[i];
}
void rdamd_model_set_lockstep_rounds(rdamd_model_t *m, int mode) { m->lockstep_rounds = mode; }
void rdamd_model_round_seconds(const rdamd_model_t *m, double out[4]) {
  for (int i = 0; i < 4; ++i) out[i] = m->round_seconds[i];
}
void rdamd_model_round_stats(const rdamd_model_t *m, uint64_t out[4]) {
  for (int i = 0; i < 3; ++i) out[i] = m->round_stats[i];
  out[3] = m->model->collectives();
}
void rdamd_model_counters(const rdamd_model_t *m, uint64_t out[6]) {
  const auto c = m->model->counters();
  for (int i = 0; i < 6; ++i) out[i] = c[i];
}

int rdamd_model_set_progress(rdamd_model_t *m, int on) {
  if (on) {
    m->progress.reset(new rdamd::model_t::progress_t());
    m->progress->total = m->model->assigned_indicies().size();
  } else {
    m->progress.reset();
  }
  m->model->set_progress(m->progress.get());
  return RDAMD_SUCCESS;
}
int rdamd_model_set_checkpoint(rdamd_model_t *m, rdamd_checkpoint_t *c) {
  m->checkpoint = rdamd_checkpoint_cpp(c);
  m->model->set_checkpoint(m->checkpoint);
  return RDAMD_SUCCESS;
}
int rdamd_model_assign_by_rank_checkpoint(rdamd_model_t *m, unsigned int rank,
                                          unsigned int num_tasks, rdamd_checkpoint_t *c) {
  GUARD(RDAMD_FAILURE, {
    std::vector<size_t> done;
    if (c) done = rdamd_checkpoint_cpp(c)->completed_indicies();
    m->model->assign_indicies_by_rank_exhaustive(rank, num_tasks, done);
    return RDAMD_SUCCESS;
  })
}
int rdamd_model_assign_by_rank_search(rdamd_model_t *m, unsigned int min_roots, double root_ratio,
                                      unsigned int rank, unsigned int num_tasks,
                                      int initial_root_strategy, rdamd_checkpoint_t *c) {
  GUARD(RDAMD_FAILURE, {
    if (initial_root_strategy < 0 || initial_root_strategy > 2)
      throw std::runtime_error("The initial root strategy was not recognized");
    std::vector<size_t> done;
    if (c) done = rdamd_checkpoint_cpp(c)->completed_indicies();
    m->model->assign_indicies_by_rank_search(
        min_roots, root_ratio, rank, num_tasks,
        (rdamd::model_t::initial_root_strategy)initial_root_strategy, done);
    return RDAMD_SUCCESS;
  })
}
int rdamd_model_assigned(const rdamd_model_t *m, uint64_t *root_ids, unsigned int cap) {
  const auto &idx = m->model->assigned_indicies();
  for (size_t i = 0; i < idx.size() && i < cap; ++i) root_ids[i] = idx[i];
  return (int)idx.size();
}
int rdamd_model_assign_by_rank(rdamd_model_t *m, unsigned int rank, unsigned int num_tasks) {
  GUARD(RDAMD_FAILURE, {
    m->model->assign_indicies_by_rank_exhaustive(rank, num_tasks);
    return RDAMD_SUCCESS;
  })
}
int rdamd_model_exhaustive_search(rdamd_model_t *m, double atol, double pgtol, double brtol,
                                  double factor, uint64_t *root_id, double *llh, double *alpha,
                                  unsigned int *n_results, rdamd_root_location_t *best_rl,
                                  double *best_llh) {
  GUARD(RDAMD_FAILURE, {
    std::vector<rdamd::rd_result_t> res;
    auto best = m->model->exhaustive_search(atol, pgtol, brtol, factor, &res);
    for (size_t i = 0; i < res.size(); ++i) {
      root_id[i] = res[i].root_id; llh[i] = res[i].llh; alpha[i] = res[i].alpha;
    }
    *n_results = (unsigned)res.size();
    if (best_rl) to_c(best.first, best_rl);
    if (best_llh) *best_llh = best.second;
    return RDAMD_SUCCESS;
  })
}

}  // extern "C"
