// Host-side likelihood facade: mirrors model_t of the reference
// (/root/reference/src/model.hpp:47-277) for the rows of SURVEY.md section 8a
// that sit above the three library calls -- compute_lh (a4), the per-partition
// objective (a5), compute_lh_root (a6), compute_dlh (a7), move_root (a8),
// update_pmatrices (a9), the parameter setters (a16), optimize_alpha / Brent
// and the exhaustive outer loop (a17) -- on top of the C ABI of this library.
//
// The L-BFGS-B parameter optimiser of the reference (lib/lbfgsb, out of scope:
// SURVEY 2.1) is reached through a callback so the caller keeps its own;
// its objective is served in batches by compute_lh_batch().
#pragma once

#include <cstdint>
#include <functional>
#include <memory>
#include <random>
#include <stdexcept>
#include <array>
#include <atomic>
#include <chrono>
#include <string>
#include <unordered_set>
#include <utility>
#include <vector>

#include "../../include/root_digger_amd.h"
#include "tree.hpp"

namespace rdamd {

class checkpoint_t;
class batch_combiner_t;
class root_combiner_t;
class conductor_t;

typedef std::vector<double> model_params_t;

struct dlh_t {   // src/model.hpp:21-24
  double lh;
  double dlh;
};

// src/util.hpp:37-70.  Values and member order are the reference's: the struct
// goes into the checkpoint file as its raw 32 bytes (checkpoint.hpp).
enum class param_type : int32_t { emperical, estimate, equal, user };
enum class rate_category : int32_t { MEDIAN, MEAN, FREE };

struct ratehet_opts_t {
  param_type    type = param_type::estimate;
  rate_category rate_category_type = rate_category::MEAN;
  uint64_t      rate_cats = 1;
  bool          alpha_init = false;
  double        alpha = 1.0;
  ratehet_opts_t() = default;
  ratehet_opts_t(size_t cats) : rate_cats(cats) {}
};
static_assert(sizeof(ratehet_opts_t) == 32, "checkpoint layout");

struct partition_parameters_t {   // src/util.hpp:119-124
  model_params_t subst_rates, freqs, gamma_alpha, gamma_weights;
};

struct rd_result_t {   // src/util.hpp:126-130
  size_t root_id;
  double llh;
  double alpha;
};

// Alignment of one partition (src/msa.hpp:21-68): in memory, or read from a
// PHYLIP / FASTA file with optional site-pattern compression (msa.cpp).
struct msa_t {
  std::vector<std::string>  labels, sequences;
  std::vector<unsigned int> weights;        // pattern weights; empty = all 1
  unsigned int              states = 4;
  const uint64_t           *map = rdamd_map_nt;
  // a caller's character map is COPIED (the model -- and the replicas a parallel search makes
  // from it -- outlive the call that handed it over); copies of the msa share the storage
  std::shared_ptr<std::vector<uint64_t>> map_store;
  void set_map(const uint64_t *m) {
    if (!m || m == rdamd_map_nt) { map = rdamd_map_nt; map_store.reset(); return; }
    map_store = std::make_shared<std::vector<uint64_t>>(m, m + 256);
    map = map_store->data();
  }
  size_t length() const { return sequences.empty() ? 0 : sequences[0].size(); }
  int    count() const { return (int)sequences.size(); }
  unsigned int total_weight() const;

  static msa_t from_file(const std::string &filename, const uint64_t *map = nullptr,
                         unsigned int states = 4, bool compress_patterns = true);
  void compress();                                                    // src/msa.cpp:621-633
  // columns [lo, hi) of an UNCOMPRESSED alignment (a site block of a site-sharded run)
  msa_t columns(size_t lo, size_t hi) const;
  bool constiency_check(const std::unordered_set<std::string> &tree_labels) const;   // :641-668
  void valid_data() const;                                            // :670-686
};

model_params_t random_params(size_t size, uint64_t seed);   // src/model.cpp:87-93

class model_t {
public:
  // sparse_clvs: 4-state / binary partitions are created with RDAMD_ATTRIB_SPARSE_CLVS -- the
  // replicas of a lock-stepped search, which only ever hold the root's two children
  // (compute_lh_for_root_steps discards everything else before it writes them)
  model_t(rooted_tree_t t, const std::vector<msa_t> &msa,
          const std::vector<ratehet_opts_t> &rate_cats, bool invariant_sites,
          uint64_t seed, bool early_stop, bool sparse_clvs = false);
  ~model_t();
  model_t(const model_t &) = delete;
  model_t &operator=(const model_t &) = delete;

  // ---- likelihood facade (hot-path callers) -------------------------------
  double compute_lh(const root_location_t &root_location);      // src/model.cpp:384-413
  // the same value; only the root's two children are left materialised (model.cpp)
  double compute_lh_for_root_steps(const root_location_t &root_location);
  void set_root_children_only(bool on) { _children_only = on; }
  double compute_lh_root(const root_location_t &root);          // :415-452
  dlh_t  compute_dlh(const root_location_t &root_location);     // :481-519
  // several positions of one root branch per launch (optimize_alpha's opening and scan levels)
  std::vector<double> root_lh_at(const root_location_t &root, const std::vector<double> &ratios);
  std::vector<dlh_t>  compute_dlh_many(const std::vector<root_location_t> &roots);
  void   move_root(const root_location_t &new_root);            // :823-854
  std::vector<double> compute_all_root_lh();                    // :1737-1746
  // the same sweep at the current parameters as ONE fused launch (every root a
  // job; 4-state data): what the all-directions CLV cache of SURVEY 8f item 2
  // was meant to buy, without storing any CLV.  Partition state is untouched.
  std::vector<double> compute_all_root_lh_batched();
  // the same sweep through an all-directions CLV cache: a second partition holds
  // the 3(n-2) directed CLVs (rooted_tree_t::generate_directional_operations),
  // so every root costs one root operation.  ratios: alpha per root (nullptr =
  // the stored ones).  One 4-state / binary partition.
  std::vector<double> compute_all_root_lh_directional(const std::vector<double> *ratios = nullptr);

  // batched objective: lnL of (root, parameter set) pairs, all partitions,
  // through rdamd_evaluate_batch (one fused launch per partition)
  std::vector<double> compute_lh_batch(const std::vector<root_location_t> &roots,
                                       const std::vector<std::vector<partition_parameters_t>> &params);

  // ---- root placement ------------------------------------------------------
  root_location_t optimize_alpha(const root_location_t &root, double atol);   // :679-794
  std::pair<root_location_t, double> optimize_root_location(size_t min_roots,
                                                            double root_ratio);  // :796-821
  std::vector<root_location_t> suggest_roots_lh(size_t min, double ratio);   // :865-889

  // parameter optimiser hook: called where the reference calls optimize_params
  // (src/model.cpp:1174); receives the current parameters and root and returns
  // improved ones.  Default: none (parameters stay as initialised).
  using param_optimizer_t = std::function<void(model_t &, std::vector<partition_parameters_t> &,
                                               const root_location_t &, double pgtol,
                                               double factor, bool optimize_gamma)>;
  void set_param_optimizer(param_optimizer_t f) { _optimizer = std::move(f); }

  // The caller's L-BFGS-B (the reference vendors lib/lbfgsb; its reverse-
  // communication entry point setulb, lib/lbfgsb/lbfgsb.h:196-201).  Once set,
  // optimize_params() drives it exactly as bfgs_params does
  // (src/model.cpp:1430-1522), except that the objective and its n finite-
  // difference perturbations are ONE batched device launch per iteration.
  typedef int (*setulb_fn)(int *n, int *m, double *x, double *l, double *u, int *nbd, double *f,
                           double *g, double *factr, double *pgtol, double *wa, int *iwa,
                           int *task, int *iprint, int *csave, int *lsave, int *isave,
                           double *dsave);
  void set_lbfgsb(setulb_fn fn) { _setulb = fn; }
  // every finished candidate is appended to this result log (checkpoint.hpp), as
  // the reference's searches do (src/model.cpp:1107, :1215); not owned
  void set_checkpoint(checkpoint_t *c) { _checkpoint = c; }
  // progress lines of the reference's searches ("Step i / n, ETC: h" at its
  // default verbosity, src/model.cpp:1101-1105, :1219-1223) go to stdout when a
  // tracker is attached; shared by the replicas of a parallel search; not owned
  struct progress_t {
    std::atomic<size_t> done{0};
    size_t total = 0;
    std::chrono::steady_clock::time_point start = std::chrono::steady_clock::now();
    void step(const char *what);
  };
  void set_progress(progress_t *p) { _progress = p; }
  // the optimiser's objective batches go through this combiner (batch_combiner.hpp)
  // instead of being launched on this model's own partition; not owned
  // (one per partition: a partitioned model optimises its partitions one after the other, each
  // on its own objective, src/model.cpp:1935-1984 -- candidates meet per partition)
  void set_combiners(std::vector<batch_combiner_t *> c) { _combiners = std::move(c); }
  void set_combiner(batch_combiner_t *c) { _combiners.assign(c ? 1 : 0, c); }
  // ... and the root-only steps (compute_lh_root / compute_dlh) through this one
  void set_root_combiner(root_combiner_t *c) { _root_combiner = c; }
  // Site-sharded runs: this model holds one block of the alignment's
  // columns; every lnL it computes is summed over the ranks of its site group
  // through `fn` before any optimiser sees it (include/root_digger_amd.h,
  // rdamd_lnl_reducer_t).  device = true: fn takes device memory + the stream.
  void set_lnl_reducer(rdamd_lnl_reducer_t fn, void *user, bool device) {
    _reduce = fn; _reduce_user = user; _reduce_device = device;
    _reduce_queue = nullptr; _reduce_wait = nullptr;
    _reduce_abort = nullptr; _reduce_abort_user = nullptr;
    _empirical.clear();
  }
  // how to get a thread of this process OUT of the reducer (include/root_digger_amd.h,
  // rdamd_model_set_lnl_reducer_abort): a lock-stepped search that fails in one worker group's
  // round must not wait for the other group's collective, which the failed ranks never join
  void set_lnl_reducer_abort(rdamd_lnl_abort_t fn, void *user) { _reduce_abort = fn; _reduce_abort_user = user; }
  // the same reducer in two halves (include/root_digger_amd.h, rdamd_model_set_lnl_reducer_async):
  // what the lock-stepped search of a site-sharded model queues behind its rounds
  void set_lnl_reducer_async(rdamd_lnl_reducer_t queue, rdamd_lnl_wait_t wait, void *user) {
    _reduce_queue = queue; _reduce_wait = wait; _reduce_async_user = user;
  }
  bool site_sharded() const { return _reduce != nullptr; }
  struct reducer_t {
    rdamd_lnl_reducer_t reduce, queue; rdamd_lnl_wait_t wait; void *user, *async_user; bool device;
    rdamd_lnl_abort_t abort; void *abort_user;
  };
  reducer_t reducer() const {
    return {_reduce, _reduce_queue, _reduce_wait, _reduce_user, _reduce_async_user, _reduce_device, _reduce_abort, _reduce_abort_user};
  }
  // Lock step in deterministic rounds (lockstep_conductor.hpp): this model is the replica one
  // candidate in flight runs on; its objective batches, its root-only steps and every value it
  // needs summed over the site group go through the conductor as worker `worker`.  Not owned.
  void set_conductor(conductor_t *c, unsigned worker) { _conductor = c; _worker = worker; }
  // empirical frequencies a replica takes over from the model it was made from (they depend on
  // the data only; for a site-sharded model they are SUMS over the group, which a replica --
  // running on a thread of its own -- must not ask for by itself)
  // (kept across the replica's own tip load: set_tip_states forgets computed frequencies, not adopted ones)
  void adopt_empirical_freqs(const model_t &other) { _empirical = other._empirical; _empirical_adopted = true; }
  const std::vector<model_params_t> &empirical_freqs() const { return _empirical; }
  // host values summed over the site group in place (the model's own collective; no-op unsharded)
  void sum_over_site_group(double *values, size_t n) { reduce_values(values, n); }
  // collectives this model has asked its reducer for (a sequential site-sharded search: one per
  // request; the lock-stepped one counts in the conductor)
  uint64_t collectives() const { return _n_collectives; }
  // src/model.cpp:1925-1984
  void optimize_params(std::vector<partition_parameters_t> &params, const root_location_t &rl,
                       double pgtol, double factor, bool optimize_gamma);
  size_t objective_batches() const { return _objective_batches; }
  // work counters since construction: {objective batches, objective evaluations,
  // full traversals (compute_lh), root-only positions (compute_lh_root/compute_dlh),
  // move_root calls, L-BFGS-B iterations}
  std::array<uint64_t, 6> counters() const {
    return {_objective_batches, _objective_evals, _n_full, _n_root_positions, _n_move_root, _n_lbfgsb_iters};
  }
  size_t objective_evaluations() const { return _objective_evals; }

  // heuristic search, src/model.cpp:1008-1137 (needs set_lbfgsb)
  std::pair<root_location_t, double> search(size_t min_roots, double root_ratio, double atol,
                                            double pgtol, double brtol, double factor,
                                            std::vector<rd_result_t> *results = nullptr);
  // src/model.cpp:1139-1272; results (one per assigned root) are returned
  // instead of going through the checkpoint file.
  std::pair<root_location_t, double> exhaustive_search(double atol, double pgtol, double brtol,
                                                       double factor,
                                                       std::vector<rd_result_t> *results = nullptr);
  // LWR = exp(llh - max) / sum, src/model.cpp:1239-1258
  static std::vector<double> likelihood_weight_ratios(const std::vector<rd_result_t> &results);

  void initialize() { compute_lh(_tree.root_location(0)); }   // :1274
  void finalize() { _tree.unroot(); }                         // :1276

  // ---- parameters (a16) -----------------------------------------------------
  void initialize_partitions(const std::vector<msa_t> &);                 // :1297-1306
  void initialize_partitions_uniform_freqs(const std::vector<msa_t> &);   // :1308-1321
  void set_subst_rates(size_t p, const model_params_t &);                 // :184
  void set_subst_rates_uniform();                                         // :1748-1755
  void set_freqs(size_t p, const model_params_t &);                       // :341
  void set_freqs_all_free(size_t p, model_params_t);                      // :350
  void set_empirical_freqs(size_t p);                                     // :327
  void set_empirical_freqs();
  void set_gamma_rates(size_t p);                                         // :208
  void set_gamma_rates(size_t p, const model_params_t &alpha);            // :224
  void set_gamma_weights(size_t p, model_params_t w);                     // :199
  void set_model_params(const std::vector<partition_parameters_t> &);     // :1913-1923
  partition_parameters_t make_partition_parameters(size_t states, rate_category rc,
                                                   size_t rate_cat_count);   // :979-1005

  // ---- work assignment (src/model.cpp:1761-1911) ------------------------------
  void assign_indicies();
  void assign_indicies(const std::vector<size_t> &idx) { _assigned_idx = idx; }
  // starting roots of the heuristic search, src/model.cpp:941-962, :1809-1865
  enum class initial_root_strategy { random, midpoint, modified_mad };
  std::vector<size_t> shuffle_root_indicies();
  std::vector<size_t> suggest_root_indicies_midpoint() const;
  std::vector<size_t> suggest_root_indicies_modified_mad() const;
  void assign_indicies_by_rank_search(size_t min_roots, double root_ratio, size_t rank,
                                      size_t num_tasks, initial_root_strategy init_root,
                                      const std::vector<size_t> &completed = {});
  void assign_indicies_by_rank_exhaustive(size_t rank, size_t num_tasks,
                                          const std::vector<size_t> &completed = {});
  std::vector<size_t> assigned_indicies() const { return _assigned_idx; }

  rooted_tree_t       &tree() { return _tree; }
  size_t               partition_count() const { return _partitions.size(); }
  rdamd_partition_t   *partition(size_t i) { return _partitions[i]; }

private:
  std::pair<root_location_t, double> brents(root_location_t beg, dlh_t d_beg,
                                            root_location_t end, dlh_t d_end, double atol);
  void set_tip_states(size_t p, const msa_t &msa);   // :302-325
  void update_invariant_sites(size_t p);             // :292-300
  void update_pmatrices(const std::vector<unsigned int> &pmatrix_indices,
                        const std::vector<double> &branch_lengths);   // :357-382 (one call per partition)

  void   reduce_values(double *values, size_t n);   // no-op without a reducer
  double reduce_value(double v) { reduce_values(&v, 1); return v; }
  double *reduce_scratch(size_t n);                 // device buffer of >= n doubles

  rdamd_lnl_reducer_t                    _reduce = nullptr, _reduce_queue = nullptr;
  rdamd_lnl_wait_t                       _reduce_wait = nullptr;
  rdamd_lnl_abort_t                      _reduce_abort = nullptr;
  void                                  *_reduce_abort_user = nullptr;
  void                                  *_reduce_user = nullptr, *_reduce_async_user = nullptr;
  bool                                   _reduce_device = false;
  conductor_t                           *_conductor = nullptr;
  unsigned                               _worker = 0;
  std::vector<model_params_t>            _empirical;           // [partition]: empirical frequencies, once computed
  bool                                   _empirical_adopted = false;   // ... by the model this replica was made from
  uint64_t                               _n_collectives = 0;
  // the divergence guard of the model's OWN reductions (model.cpp, reduce_values): two words ride
  // behind every vector -- 1.0 and a 40-bit hash of the bits the previous reduction returned
  double                                 _guard_prev = 0.0;
  void guard_fill(double *tail) const { tail[0] = 1.0; tail[1] = _guard_prev; }
  void guard_check(const double *sums, size_t n);   // throws; then remembers sums[0 .. n)
  double                                *_d_reduce = nullptr;
  size_t                                 _d_reduce_cap = 0;
  double                                *_h_reduce = nullptr;   // pinned twin of _d_reduce
  rooted_tree_t                          _tree;
  std::vector<rdamd_partition_t *>       _partitions;
  std::vector<rate_category>             _rate_category_types;
  std::vector<model_params_t>            _rate_rates, _rate_weights;
  std::vector<bool>                      _rate_user_init;
  std::vector<std::vector<unsigned int>> _param_indicies;
  std::vector<size_t>                    _assigned_idx;
  std::minstd_rand                       _random_engine;
  rdamd_partition_t                     *_sweep = nullptr;   // all-directions cache (lazily built)
  std::vector<msa_t>                     _sweep_msa;          // what it needs to load its tips
  progress_t                            *_progress = nullptr;
  checkpoint_t                          *_checkpoint = nullptr;
  std::vector<batch_combiner_t *>        _combiners;          // [partition], or empty
  root_combiner_t                       *_root_combiner = nullptr;
  bool                                   _invariant_sites, _early_stop;   // +I is inert (:292-300)
  bool                                   _children_only = true;           // compute_lh_for_root_steps
  bool                                   _sparse = false;                 // RDAMD_ATTRIB_SPARSE_CLVS partitions
  uint64_t                               _seed;
  param_optimizer_t                      _optimizer;
  setulb_fn                              _setulb = nullptr;
  size_t                                 _objective_batches = 0, _objective_evals = 0;
  uint64_t _n_full = 0, _n_root_positions = 0, _n_move_root = 0, _n_lbfgsb_iters = 0;

  enum class bfgs_target { rates, freqs, gamma };
  // lnL at up to RDAMD_ROOT_MAX_POSITIONS positions of the root operation, all partitions, summed
  // over the site group: one launch (or one request to the combiner / conductor it meets the
  // other candidates' steps in)
  void root_positions(const rdamd_operation_t &op, const double *l1, const double *l2, unsigned n, double *total);
  double bfgs_params(model_params_t &initial, size_t partition, bfgs_target what,
                     rdamd_schedule_t *sched, batch_combiner_t *combiner, double p_min, double p_max,
                     double epsilon, double pgtol, double factor);
};

}  // namespace rdamd
