// model_t on the rdamd C ABI; behaviour follows /root/reference/src/model.cpp
// (cited per function).
#include "model.hpp"

#include <cstdio>
#include <hip/hip_runtime.h>
#include "batch_combiner.hpp"
#include "checkpoint.hpp"
#include "lockstep_conductor.hpp"

#include <algorithm>
#include <cmath>
#include <limits>
#include <numeric>
#include <unordered_map>

namespace rdamd {

namespace {
[[noreturn]] void fail(const std::string &what) {
  std::string m = what;
  const char *e = rdamd_errmsg();
  if (e && *e) m += std::string(": ") + e;
  throw std::runtime_error(m);
}
}  // namespace


model_params_t random_params(size_t size, uint64_t seed) {
  model_params_t mp(size);
  std::minstd_rand engine((std::minstd_rand::result_type)seed);
  std::uniform_real_distribution<> dist(1e-4, 1.0);
  for (auto &f : mp) f = dist(engine);
  return mp;
}

// src/model.cpp:99-176
model_t::model_t(rooted_tree_t tree, const std::vector<msa_t> &msas,
                 const std::vector<ratehet_opts_t> &rate_cats, bool invariant_sites,
                 uint64_t seed, bool early_stop, bool sparse_clvs)
    : _invariant_sites(invariant_sites), _early_stop(early_stop), _sparse(sparse_clvs), _seed(seed) {
  _random_engine = std::minstd_rand((std::minstd_rand::result_type)_seed);
  _tree = std::move(tree);
  if (rate_cats.size() != msas.size())
    throw std::invalid_argument("one rate-heterogeneity option per partition is required");
  for (const auto &rc : rate_cats) {
    _rate_rates.emplace_back(rc.rate_cats, rc.alpha);
    _rate_weights.emplace_back(rc.rate_cats, 1.0 / rc.rate_cats);
    _rate_category_types.push_back(rc.rate_category_type);
    _rate_user_init.push_back(rc.alpha_init);
    _param_indicies.emplace_back(rc.rate_cats, 0u);
  }
  auto labels = _tree.label_set();
  for (const auto &msa : msas) {
    if ((size_t)msa.count() != labels.size())
      throw std::invalid_argument("Taxa on the tree and in the MSA are inconsistient");
    for (const auto &l : msa.labels)
      if (!labels.count(l))
        throw std::invalid_argument("Taxa on the tree and in the MSA are inconsistient");
  }
  for (size_t p = 0; p < msas.size(); ++p) {
    const auto &msa = msas[p];
    unsigned int attributes = RDAMD_ATTRIB_NONREV;
    if (msa.states == 4) attributes |= RDAMD_ATTRIB_SITE_REPEATS;
    // (the shapes rdamd_evaluate_root_children takes: what compute_lh_for_root_steps runs on)
    if (_sparse && (msa.states == 4 || msa.states == 2 || (msa.states == 20 && _rate_rates[p].size() <= 8)))
      attributes |= RDAMD_ATTRIB_SPARSE_CLVS;
    rdamd_partition_t *part = rdamd_partition_create(
        _tree.tip_count(), _tree.branch_count(), msa.states, (unsigned)msa.length(), 1,
        _tree.branch_count(), (unsigned)_rate_rates[p].size(), _tree.branch_count(), attributes);
    if (!part) fail("partition_create");
    _partitions.push_back(part);
    set_gamma_rates(p);
  }
  assign_indicies();
}

model_t::~model_t() {
  for (auto p : _partitions)
    if (p) rdamd_partition_destroy(p);
  if (_sweep) rdamd_partition_destroy(_sweep);
  if (_d_reduce) (void)hipFree(_d_reduce);
  if (_h_reduce) (void)hipHostFree(_h_reduce);
}

// ---- site-group reduction ---------------------------------------------
double *model_t::reduce_scratch(size_t n) {
  if (n > _d_reduce_cap) {
    if (_d_reduce) (void)hipFree(_d_reduce);
    _d_reduce = nullptr;
    if (_h_reduce) (void)hipHostFree(_h_reduce);
    _h_reduce = nullptr;
    _d_reduce_cap = std::max<size_t>(n, 256);
    // (the pinned twin: scalar lnLs go up and come back without a pageable staging copy)
    if (hipMalloc((void **)&_d_reduce, _d_reduce_cap * sizeof(double)) != hipSuccess ||
        hipHostMalloc((void **)&_h_reduce, _d_reduce_cap * sizeof(double), hipHostMallocDefault) != hipSuccess) {
      _d_reduce_cap = 0;
      throw std::runtime_error("site-group reduction: hipMalloc failed");
    }
  }
  return _d_reduce;
}

// The guard: every rank of the group must receive the same BITS from the
// reducer -- the optimisers branch on them.  Two words ride behind every vector the model itself has
// summed: 1.0 (comes back as the number of ranks G) and a 40-bit hash of the bits the PREVIOUS
// reduction handed to this rank (must come back as G x own: exact, small integers in doubles).  A rank
// whose copy differed by one ulp is found by the next reduction, on every rank, before its requests
// can differ -- a failure with a reason instead of a collective that no longer matches.  (The rounds
// of a lock-stepped search carry their own: lockstep_conductor.hpp.)
constexpr size_t kGuardWords = 2;
void model_t::guard_check(const double *sums, size_t n) {
  const double ranks = sums[n];
  if (!(ranks >= 1.0) || ranks != std::floor(ranks) || sums[n + 1] != ranks * _guard_prev)
    throw std::runtime_error("site-group reduction " + std::to_string(_n_collectives) +
                             ": the ranks did not receive the same bits from the previous reduction "
                             "(every rank must get identical sums from the reducer; the library's default, "
                             "ncclAllGather + a sum in rank order, does by construction)");
  uint64_t h = 1469598103934665603ull;
  const unsigned char *b = reinterpret_cast<const unsigned char *>(sums);
  for (size_t i = 0; i < n * sizeof(double); ++i) h = (h ^ b[i]) * 1099511628211ull;
  _guard_prev = (double)((h ^ (h >> 40)) & ((1ull << 40) - 1));
}

// host values in, their sums over the site group out (identical on every rank)
void model_t::reduce_values(double *values, size_t n) {
  if (n == 0) return;
  if (_conductor) {   // a candidate in flight: its sums travel with the round's collective
    _conductor->reduce(_worker, values, (unsigned)n);
    return;
  }
  if (!_reduce) return;
  ++_n_collectives;
  if (!_reduce_device) {
    std::vector<double> v(values, values + n);
    v.resize(n + kGuardWords);
    guard_fill(v.data() + n);
    if (_reduce(v.data(), (unsigned)v.size(), nullptr, _reduce_user) != RDAMD_SUCCESS)
      throw std::runtime_error("site-group reduction failed");
    guard_check(v.data(), n);
    std::copy(v.begin(), v.begin() + n, values);
    return;
  }
  const size_t m = n + kGuardWords;
  double *d = reduce_scratch(m);
  hipStream_t st = (hipStream_t)rdamd_partition_stream(_partitions[0]);
  std::copy(values, values + n, _h_reduce);
  guard_fill(_h_reduce + n);
  if (hipMemcpyAsync(d, _h_reduce, m * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess ||
      _reduce(d, (unsigned)m, st, _reduce_user) != RDAMD_SUCCESS ||
      hipMemcpyAsync(_h_reduce, d, m * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess)
    throw std::runtime_error(std::string("site-group reduction failed: ") + rdamd_errmsg());
  guard_check(_h_reduce, n);
  std::copy(_h_reduce, _h_reduce + n, values);
}

// ---- setters ----------------------------------------------------------------
void model_t::set_subst_rates(size_t p, const model_params_t &mp) {
  rdamd_set_subst_params(_partitions[p], 0, mp.data());
}

void model_t::set_subst_rates_uniform() {
  for (size_t i = 0; i < _partitions.size(); ++i) {
    unsigned states = rdamd_partition_states(_partitions[i]);
    unsigned params = states * states - states;
    set_subst_rates(i, model_params_t(params, 1.0 / params));
  }
}

void model_t::set_gamma_weights(size_t p, model_params_t w) {
  double sum = 0.0;
  for (auto f : w) sum += f;
  for (auto &f : w) f /= sum;
  rdamd_set_category_weights(_partitions[p], w.data());
}

// src/model.cpp:208-290.  The reference's mean/median setters swap the two
// enum values: the initial call uses MEAN at alpha = 1,
// every later call uses MEDIAN at the given alpha, for both declared types.
// That observable behaviour is kept.
void model_t::set_gamma_rates(size_t p) {
  rdamd_set_category_weights(_partitions[p], _rate_weights[p].data());
  if (_rate_category_types[p] == rate_category::FREE) {
    for (auto &r : _rate_rates[p]) r = 1.0;
  } else {
    rdamd_compute_gamma_cats(1.0, (unsigned)_rate_rates[p].size(), _rate_rates[p].data(),
                             RDAMD_GAMMA_RATES_MEAN);
  }
  rdamd_set_category_rates(_partitions[p], _rate_rates[p].data());
}

void model_t::set_gamma_rates(size_t p, const model_params_t &alpha) {
  if (_rate_category_types[p] != rate_category::FREE)
    rdamd_compute_gamma_cats(alpha[0], (unsigned)_rate_rates[p].size(), _rate_rates[p].data(),
                             RDAMD_GAMMA_RATES_MEDIAN);
  // FREE: the reference normalises a local copy and uploads the unchanged
  // member (all ones), src/model.cpp:279-290 -- free rates never take effect
  rdamd_set_category_rates(_partitions[p], _rate_rates[p].data());
}

void model_t::update_invariant_sites(size_t p) {
  // src/model.cpp:292-300: the proportion is only ever 0.0 (+I is inert) -- with the flag the
  // reference re-derives which sites are invariant, which a proportion of 0 never reads
  (void)_invariant_sites;
  if (rdamd_update_invariant_sites_proportion(_partitions[p], 0, 0.0) != RDAMD_SUCCESS)
    fail("update_invariant_sites");
}

void model_t::set_tip_states(size_t p, const msa_t &msa) {
  auto label_map = _tree.label_map();
  for (int i = 0; i < msa.count(); ++i) {
    auto it = label_map.find(msa.labels[i]);
    if (it == label_map.end())
      throw std::runtime_error("Could not find taxa " + msa.labels[i] + " in tree");
    if (rdamd_set_tip_states(_partitions[p], it->second, msa.map, msa.sequences[i].c_str()) !=
        RDAMD_SUCCESS)
      fail("failed to set tip " + std::to_string(i));
  }
  if (!msa.weights.empty()) rdamd_set_pattern_weights(_partitions[p], msa.weights.data());
  // (new data: new empirical frequencies -- unless they are the whole alignment's, adopted from the
  // model this replica was made from: a replica's own block may even lack a state)
  if (!_empirical_adopted && p < _empirical.size()) _empirical[p].clear();
}

void model_t::set_empirical_freqs(size_t p) {
  unsigned states = rdamd_partition_states(_partitions[p]);
  // They depend on the data alone: computed once (the reference recomputes them for every
  // candidate, src/model.cpp:1157; the same numbers).  A site-sharded model's are sums over the
  // group -- one collective per model instead of one per candidate, and none at all from the
  // replicas of a lock-stepped search, which take them over (adopt_empirical_freqs).
  if (p < _empirical.size() && _empirical[p].size() == states) {
    rdamd_set_frequencies(_partitions[p], 0, _empirical[p].data());
    return;
  }
  double *f = rdamd_msa_empirical_frequencies(_partitions[p]);
  if (_reduce || _conductor) {   // one model for the whole alignment: blocks weighted by their columns
    const double w = rdamd_partition_weight_sum(_partitions[p]);
    std::vector<double> acc(states + 1);
    for (unsigned i = 0; i < states; ++i) acc[i] = f[i] * w;
    acc[states] = w;
    reduce_values(acc.data(), acc.size());
    for (unsigned i = 0; i < states; ++i) f[i] = acc[i] / acc[states];
  }
  for (unsigned i = 0; i < states; ++i)
    if (f[i] <= 0) {
      free(f);
      throw std::runtime_error("One of the state frequenices is zero while using emperical "
                               "frequencies");
    }
  rdamd_set_frequencies(_partitions[p], 0, f);
  if (_empirical.size() <= p) _empirical.resize(p + 1);
  _empirical[p].assign(f, f + states);
  free(f);
}

void model_t::set_empirical_freqs() {
  for (size_t i = 0; i < _partitions.size(); ++i) set_empirical_freqs(i);
}

void model_t::set_freqs(size_t p, const model_params_t &freqs) {
  for (auto f : freqs)
    if (f <= 0.0) throw std::runtime_error("Frequencies with 0 entries are not allowed");
  rdamd_set_frequencies(_partitions[p], 0, freqs.data());
}

void model_t::set_freqs_all_free(size_t p, model_params_t freqs) {
  double sum = 0.0;
  for (auto f : freqs) sum += f;
  for (auto &f : freqs) f /= sum;
  set_freqs(p, freqs);
}

void model_t::set_model_params(const std::vector<partition_parameters_t> &params) {
  for (size_t i = 0; i < params.size(); ++i) {
    set_subst_rates(i, params[i].subst_rates);
    // the optimiser's frequency vector is unnormalised (bfgs_freqs applies it
    // through set_freqs_all_free); the reference re-applies it here with the
    // plain setter (src/model.cpp:1917), which silently changes the model --
    // normalise instead so that the reported lnL reproduces
    set_freqs_all_free(i, params[i].freqs);
    set_gamma_rates(i, params[i].gamma_alpha);
    if (_rate_category_types[i] == rate_category::FREE) set_gamma_weights(i, params[i].gamma_weights);
  }
}

partition_parameters_t model_t::make_partition_parameters(size_t states, rate_category rc,
                                                          size_t cats) {
  // src/model.cpp:979-1005: uniform rates, uniform freqs, alpha 1; FREE draws
  // its category weights from the model's engine
  partition_parameters_t pp;
  pp.subst_rates.assign(states * states - states, 1.0 / (states * states - states));
  pp.freqs.assign(states, 1.0 / states);
  if (rc == rate_category::FREE) {
    pp.gamma_alpha.assign(cats, 1.0);
    std::uniform_real_distribution<> dis(0.0, 1.0);
    pp.gamma_weights.resize(cats);
    for (auto &v : pp.gamma_weights) v = dis(_random_engine);
  } else {
    pp.gamma_alpha.assign(1, 1.0);
  }
  return pp;
}

void model_t::initialize_partitions(const std::vector<msa_t> &msa) {
  _sweep_msa = msa;   // the all-directions cache loads the same tips when it is first used
  for (size_t p = 0; p < _partitions.size(); ++p) {
    set_tip_states(p, msa[p]);
    update_invariant_sites(p);
    set_empirical_freqs(p);
    set_subst_rates(p, random_params(msa[p].states * msa[p].states - msa[p].states,
                                     _random_engine()));
  }
}

void model_t::initialize_partitions_uniform_freqs(const std::vector<msa_t> &msa) {
  _sweep_msa = msa;
  for (size_t p = 0; p < _partitions.size(); ++p) {
    set_tip_states(p, msa[p]);
    update_invariant_sites(p);
    unsigned states = rdamd_partition_states(_partitions[p]);
    set_freqs(p, model_params_t(states, 1.0 / states));
    set_subst_rates(p, random_params(states * states - states, _random_engine()));
    set_gamma_rates(p);
  }
}

// ---- likelihood facade ----------------------------------------------------------
void model_t::update_pmatrices(const std::vector<unsigned int> &pmi,
                               const std::vector<double> &brl) {
  // the reference loops branch by branch under OpenMP (src/model.cpp:362-369);
  // here the whole list is one device launch per partition
  for (size_t i = 0; i < _partitions.size(); ++i)
    if (rdamd_update_prob_matrices(_partitions[i], _param_indicies[i].data(), pmi.data(),
                                   brl.data(), (unsigned)pmi.size()) != RDAMD_SUCCESS)
      fail("update_prob_matrices");
}

double model_t::compute_lh(const root_location_t &root_location) {
  auto sched = _tree.generate_operations(root_location);
  const auto &ops = std::get<0>(sched);
  update_pmatrices(std::get<1>(sched), std::get<2>(sched));
  ++_n_full;
  double lh = 0.0;
  for (size_t i = 0; i < _partitions.size(); ++i) {
    rdamd_update_clvs(_partitions[i], ops.data(), (unsigned)ops.size());
    if (rdamd_errno()) fail("update_clvs");
    lh += rdamd_compute_root_loglikelihood(_partitions[i], _tree.root_clv_index(),
                                           _tree.root_scaler_index(),
                                           _param_indicies[i].data(), nullptr);
  }
  return reduce_value(lh);
}

// compute_lh for the searches, between optimize_params and the root-only steps: the same
// value for the caller's convergence test, but only what those steps read is left behind --
// the CLVs and scalers of the root's two children (rdamd_evaluate_root_children: one job of
// the fused evaluator instead of a traversal that writes every CLV; 4-state, binary and -- up to
// four rate categories -- 20-state partitions).  Partitions the fused evaluators do not take
// (general K) go through the three calls of compute_lh.
double model_t::compute_lh_for_root_steps(const root_location_t &root_location) {
  if (!_children_only) return compute_lh(root_location);
  auto sched = _tree.generate_operations(root_location);
  const auto &ops = std::get<0>(sched);
  const auto &pmi = std::get<1>(sched);
  const auto &brl = std::get<2>(sched);
  ++_n_full;
  double lh = 0.0;
  for (size_t i = 0; i < _partitions.size(); ++i) {
    rdamd_partition_t *part = _partitions[i];
    const unsigned st = rdamd_partition_states(part);
    if (st == 4 || st == 2 || (st == 20 && rdamd_partition_rate_cats(part) <= 8)) {
      double v = 0.0;
      // (a replica's sparse partition: the children of the LAST root, and whatever else an earlier
      // call named, give their memory back -- nothing reads them after this call)
      if (_sparse) rdamd_partition_discard_clvs(part);
      if (rdamd_evaluate_root_children(part, ops.data(), (unsigned)ops.size(), pmi.data(), brl.data(),
                                       (unsigned)pmi.size(), rdamd_partition_subst_params(part, 0),
                                       rdamd_partition_frequencies(part, 0), _rate_rates[i].data(),
                                       _rate_weights[i].data(), &v) != RDAMD_SUCCESS)
        fail("evaluate_root_children");
      lh += v;
    } else {
      if (rdamd_update_prob_matrices(part, _param_indicies[i].data(), pmi.data(), brl.data(),
                                     (unsigned)pmi.size()) != RDAMD_SUCCESS)
        fail("update_prob_matrices");
      rdamd_update_clvs(part, ops.data(), (unsigned)ops.size());
      if (rdamd_errno()) fail("update_clvs");
      lh += rdamd_compute_root_loglikelihood(part, _tree.root_clv_index(), _tree.root_scaler_index(),
                                             _param_indicies[i].data(), nullptr);
    }
  }
  return reduce_value(lh);
}

void model_t::root_positions(const rdamd_operation_t &op, const double *l1, const double *l2, unsigned n,
                             double *total) {
  std::fill(total, total + n, 0.0);
  constexpr size_t P = RDAMD_ROOT_MAX_POSITIONS;
  if (_conductor || (_root_combiner && !_reduce)) {   // meets the other candidates' steps
    std::vector<const unsigned *> pidx;
    for (size_t i = 0; i < _partitions.size(); ++i) pidx.push_back(_param_indicies[i].data());
    if (_conductor) {   // (summed over partitions and site group there)
      _conductor->root(_worker, _partitions.data(), pidx.data(), (unsigned)_partitions.size(), op, l1, l2, n, total);
      return;
    }
    std::vector<double> v(_partitions.size() * P);
    _root_combiner->evaluate(_partitions.data(), pidx.data(), (unsigned)_partitions.size(), op, l1, l2, n, v.data());
    for (size_t i = 0; i < _partitions.size(); ++i)   // (summed in partition order, as the plain loop does)
      for (unsigned a = 0; a < n; ++a) total[a] += v[i * P + a];
  } else {
    double v[P];
    for (size_t i = 0; i < _partitions.size(); ++i) {
      if (rdamd_root_loglikelihood_fused(_partitions[i], &op, _param_indicies[i].data(), l1, l2, n, v) != RDAMD_SUCCESS)
        fail("root-only evaluation");
      for (unsigned a = 0; a < n; ++a) total[a] += v[a];
    }
  }
  reduce_values(total, n);
}

double model_t::compute_lh_root(const root_location_t &root) {
  auto res = _tree.generate_derivative_operations(root);
  const rdamd_operation_t &op = std::get<0>(res);
  const auto &brl = std::get<2>(res);
  ++_n_root_positions;
  double lh = 0.0;
  root_positions(op, &brl[0], &brl[1], 1, &lh);
  if (std::isnan(lh)) throw std::runtime_error("lh at root is not a number: " + std::to_string(lh));
  return lh;
}

// lnL with the root at `ratios` along root's branch: ONE launch per partition for up to eight
// positions (both child CLVs are read once whatever their number).  Tree and partitions are
// left at the LAST position, as a sequence of compute_lh_root calls would leave them; every
// value has the bits of its own compute_lh_root call.
std::vector<double> model_t::root_lh_at(const root_location_t &root, const std::vector<double> &ratios) {
  const size_t n = ratios.size();
  std::vector<double> l1(n), l2(n), total(n, 0.0);
  root_location_t at{root};
  for (size_t a = 0; a < n; ++a) {
    at.brlen_ratio = ratios[a];
    l1[a] = at.brlen();
    l2[a] = at.brlen_compliment();
  }
  auto res = _tree.generate_derivative_operations(at);   // (the last position)
  const rdamd_operation_t &op = std::get<0>(res);
  _n_root_positions += n;
  size_t step = RDAMD_ROOT_MAX_POSITIONS;   // (four at 8 rate categories)
  for (auto part : _partitions) step = std::min<size_t>(step, rdamd_partition_rate_cats(part) <= 4 ? 8 : 4);
  for (size_t lo = 0; lo < n; lo += step)
    root_positions(op, &l1[lo], &l2[lo], (unsigned)std::min(step, n - lo), &total[lo]);
  return total;
}

// the two positions compute_dlh evaluates for `root` -- alpha' first, alpha last -- and the
// sign of the difference (src/model.cpp:481-519)
static void dlh_positions(const root_location_t &root, double out[2], double *sign) {
  constexpr double EPSILON = 1e-8;
  double prime = root.brlen_ratio + EPSILON;
  *sign = 1.0;
  if (prime >= 1.0) {
    prime = root.brlen_ratio - EPSILON;
    *sign = -1.0;
  }
  out[0] = prime;
  out[1] = root.brlen_ratio;
}
static dlh_t dlh_from(double fx, double fxh, double sign, const root_location_t &root) {
  constexpr double EPSILON = 1e-8;
  if (std::isnan(fx))
    throw std::runtime_error("fx is not finite when computing derivative: " +
                             std::to_string(root.saved_brlen));
  if (std::isnan(fxh))
    throw std::runtime_error("fxh is not finite when computing derivative: " +
                             std::to_string(root.saved_brlen));
  if (std::isinf(fxh) && std::isinf(fx)) return {fx, 0};
  return {fx, (fxh - fx) / EPSILON * sign};
}

// compute_dlh for SEVERAL positions of one branch at once: two positions each, up to four
// derivatives per launch (optimize_alpha's scan levels; the same values as one call each)
std::vector<dlh_t> model_t::compute_dlh_many(const std::vector<root_location_t> &roots) {
  std::vector<double> ratios;
  std::vector<double> signs(roots.size());
  for (size_t i = 0; i < roots.size(); ++i) {
    double pos[2];
    dlh_positions(roots[i], pos, &signs[i]);
    ratios.push_back(pos[0]);
    ratios.push_back(pos[1]);
  }
  const std::vector<double> lh = root_lh_at(roots.front(), ratios);
  std::vector<dlh_t> out;
  for (size_t i = 0; i < roots.size(); ++i) out.push_back(dlh_from(lh[2 * i + 1], lh[2 * i], signs[i], roots[i]));
  return out;
}

// src/model.cpp:481-519: one-sided difference with EPSILON = 1e-8 (backward
// when alpha + eps would reach 1); both positions go to the device in one call.
dlh_t model_t::compute_dlh(const root_location_t &root) {
  constexpr double EPSILON = 1e-8;
  root_location_t root_prime{root};
  root_prime.brlen_ratio += EPSILON;
  double sign = 1.0;
  if (root_prime.brlen_ratio >= 1.0) {
    root_prime.brlen_ratio = root.brlen_ratio - EPSILON;
    sign = -1.0;
  }
  auto res = _tree.generate_derivative_operations(root);
  const rdamd_operation_t &op = std::get<0>(res);
  // evaluate alpha' first and alpha last: the partition is left at `root`,
  // where generate_derivative_operations put the tree
  const double l1[2] = {root_prime.brlen(), root.brlen()};
  const double l2[2] = {root_prime.brlen_compliment(), root.brlen_compliment()};
  _n_root_positions += 2;
  double both[2] = {0.0, 0.0};
  root_positions(op, l1, l2, 2, both);
  const double fxh = both[0], fx = both[1];
  if (std::isnan(fx))
    throw std::runtime_error("fx is not finite when computing derivative: " +
                             std::to_string(root.saved_brlen));
  if (std::isnan(fxh))
    throw std::runtime_error("fxh is not finite when computing derivative: " +
                             std::to_string(root.saved_brlen));
  if (std::isinf(fxh) && std::isinf(fx)) return {fx, 0};
  return {fx, (fxh - fx) / EPSILON * sign};
}

void model_t::move_root(const root_location_t &new_root) {
  auto sched = _tree.generate_root_update_operations(new_root);
  const auto &ops = std::get<0>(sched);
  if (ops.empty()) return;
  ++_n_move_root;
  for (size_t i = 0; i < _partitions.size(); ++i) {
    if (rdamd_update_prob_matrices(_partitions[i], _param_indicies[i].data(),
                                   std::get<1>(sched).data(), std::get<2>(sched).data(),
                                   (unsigned)std::get<1>(sched).size()) != RDAMD_SUCCESS)
      fail("move_root");
    rdamd_update_clvs(_partitions[i], ops.data(), (unsigned)ops.size());
    if (rdamd_errno()) fail("move_root");
  }
}

std::vector<double> model_t::compute_all_root_lh() {
  compute_lh(_tree.roots()[0]);
  std::vector<double> out;
  for (const auto &rl : _tree.roots()) {
    move_root(rl);
    out.push_back(compute_lh_root(rl));
  }
  return out;
}

std::vector<double> model_t::compute_all_root_lh_batched() {
  const auto &roots = _tree.roots();
  const size_t n = roots.size();
  std::vector<double> total(n, 0.0);
  rooted_tree_t scratch(_tree);   // schedules are generated on a copy: _tree keeps its rooting
  for (size_t p = 0; p < _partitions.size(); ++p) {
    rdamd_partition_t *part = _partitions[p];
    const unsigned K = rdamd_partition_states(part), NP = K * K - K;
    if (K != 4 && K != 2 && K != 20)
      throw std::runtime_error("compute_all_root_lh_batched: 4-state, binary or 20-state data only");
    const unsigned R = rdamd_partition_rate_cats(part);
    const double *cs = rdamd_partition_subst_params(part, 0), *cf = rdamd_partition_frequencies(part, 0);
    std::vector<rdamd_schedule_t *> owned;
    std::vector<const rdamd_schedule_t *> scheds(n);
    std::vector<double> subst(n * NP), freqs(n * K), rates(n * R), weights(n * R), out(n);
    for (size_t j = 0; j < n; ++j) {
      auto sc = scratch.generate_operations(roots[j]);
      rdamd_schedule_t *s = rdamd_schedule_create(
          part, std::get<0>(sc).data(), (unsigned)std::get<0>(sc).size(), std::get<1>(sc).data(),
          std::get<2>(sc).data(), (unsigned)std::get<1>(sc).size());
      if (!s) {
        for (auto o : owned) rdamd_schedule_destroy(o);
        fail("schedule_create");
      }
      owned.push_back(s);
      scheds[j] = s;
      std::copy(cs, cs + NP, subst.begin() + j * NP);
      std::copy(cf, cf + K, freqs.begin() + j * K);
      std::copy(_rate_rates[p].begin(), _rate_rates[p].end(), rates.begin() + j * R);
      std::copy(_rate_weights[p].begin(), _rate_weights[p].end(), weights.begin() + j * R);
    }
    int ok = rdamd_evaluate_batch(part, (unsigned)n, scheds.data(), subst.data(), freqs.data(),
                                  rates.data(), weights.data(), out.data());
    for (auto o : owned) rdamd_schedule_destroy(o);
    if (ok != RDAMD_SUCCESS) fail("evaluate_batch");
    for (size_t j = 0; j < n; ++j) total[j] += out[j];
  }
  reduce_values(total.data(), total.size());
  return total;
}

// SURVEY.md 8f item 2.  The cache partition mirrors partition 0's data and
// model; directed CLVs are recomputed on every call (they depend on the
// parameters), which is still 3(n-2) operations instead of (2n-3)(n-1).
std::vector<double> model_t::compute_all_root_lh_directional(const std::vector<double> *ratios) {
  if (_partitions.size() != 1 || _sweep_msa.size() != 1)
    throw std::runtime_error("compute_all_root_lh_directional: one initialised partition is required");
  rdamd_partition_t *src = _partitions[0];
  const auto d = _tree.generate_directional_operations(ratios);
  const unsigned R = rdamd_partition_rate_cats(src), K = rdamd_partition_states(src);
  if (!_sweep) {
    const msa_t &msa = _sweep_msa[0];
    _sweep = rdamd_partition_create(_tree.tip_count(), d.clv_buffers, K, (unsigned)msa.length(), 1,
                                    d.prob_matrices, R, d.scale_buffers, RDAMD_ATTRIB_NONREV);
    if (!_sweep) fail("partition_create (all-directions cache)");
    auto labels = _tree.label_map();
    for (int i = 0; i < msa.count(); ++i)
      if (rdamd_set_tip_states(_sweep, labels.at(msa.labels[i]), msa.map, msa.sequences[i].c_str()) !=
          RDAMD_SUCCESS)
        fail("set_tip_states (all-directions cache)");
    if (!msa.weights.empty()) rdamd_set_pattern_weights(_sweep, msa.weights.data());
  }
  rdamd_set_subst_params(_sweep, 0, rdamd_partition_subst_params(src, 0));
  rdamd_set_frequencies(_sweep, 0, rdamd_partition_frequencies(src, 0));
  rdamd_set_category_rates(_sweep, _rate_rates[0].data());
  rdamd_set_category_weights(_sweep, _rate_weights[0].data());
  if (rdamd_update_prob_matrices(_sweep, _param_indicies[0].data(), d.matrix_indices.data(),
                                 d.branch_lengths.data(), (unsigned)d.matrix_indices.size()) !=
      RDAMD_SUCCESS)
    fail("update_prob_matrices (all-directions cache)");
  rdamd_update_clvs(_sweep, d.ops.data(), (unsigned)d.ops.size());
  if (rdamd_errno()) fail("update_clvs (all-directions cache)");
  std::vector<double> out(d.root_clv.size());
  if (rdamd_compute_root_loglikelihoods(_sweep, (unsigned)out.size(), d.root_clv.data(),
                                        d.root_scaler.data(), _param_indicies[0].data(),
                                        out.data()) != RDAMD_SUCCESS)
    fail("compute_root_loglikelihoods");
  reduce_values(out.data(), out.size());
  return out;
}

std::vector<double> model_t::compute_lh_batch(
    const std::vector<root_location_t> &roots,
    const std::vector<std::vector<partition_parameters_t>> &params) {
  const size_t n = roots.size();
  if (params.size() != n) throw std::invalid_argument("one parameter vector per root");
  std::vector<double> total(n, 0.0);
  // schedules are per (partition, distinct root): compile once per root
  for (size_t p = 0; p < _partitions.size(); ++p) {
    const unsigned R = rdamd_partition_rate_cats(_partitions[p]);
    const unsigned K = rdamd_partition_states(_partitions[p]), NP = K * K - K;
    std::vector<rdamd_schedule_t *> owned;
    std::vector<const rdamd_schedule_t *> scheds(n);
    std::vector<double> subst(n * NP), freqs(n * K), rates(n * R), weights(n * R), out(n);
    for (size_t j = 0; j < n; ++j) {
      auto sc = _tree.generate_operations(roots[j]);
      rdamd_schedule_t *s = rdamd_schedule_create(
          _partitions[p], std::get<0>(sc).data(), (unsigned)std::get<0>(sc).size(),
          std::get<1>(sc).data(), std::get<2>(sc).data(), (unsigned)std::get<1>(sc).size());
      if (!s) {
        for (auto o : owned) rdamd_schedule_destroy(o);
        fail("schedule_create");
      }
      owned.push_back(s);
      scheds[j] = s;
      const partition_parameters_t &pp = params[j][p];
      std::copy(pp.subst_rates.begin(), pp.subst_rates.end(), subst.begin() + j * NP);
      double fs = 0.0;
      for (auto f : pp.freqs) fs += f;
      for (size_t k = 0; k < K; ++k) freqs[j * K + k] = pp.freqs[k] / fs;
      std::vector<double> r(R, 1.0);
      if (_rate_category_types[p] != rate_category::FREE)
        rdamd_compute_gamma_cats(pp.gamma_alpha.empty() ? 1.0 : pp.gamma_alpha[0], R, r.data(),
                                 RDAMD_GAMMA_RATES_MEDIAN);
      for (unsigned k = 0; k < R; ++k) {
        rates[j * R + k] = r[k];
        weights[j * R + k] = _rate_weights[p][k];
      }
    }
    int ok = rdamd_evaluate_batch(_partitions[p], (unsigned)n, scheds.data(), subst.data(),
                                  freqs.data(), rates.data(), weights.data(), out.data());
    for (auto o : owned) rdamd_schedule_destroy(o);
    if (ok != RDAMD_SUCCESS) fail("evaluate_batch");
    for (size_t j = 0; j < n; ++j) total[j] += out[j];
  }
  reduce_values(total.data(), total.size());
  return total;
}

// ---- root placement on one branch -------------------------------------------------
// Brent's method on d lnL / d alpha (src/model.cpp:606-676): bracket [a, b] with b
// the best iterate and c the previous contrapoint; inverse quadratic / secant step
// when it stays inside the bracket and shrinks fast enough, bisection otherwise;
// at most 64 derivative evaluations.
std::pair<root_location_t, double> model_t::brents(root_location_t beg, dlh_t d_beg,
                                                   root_location_t end, dlh_t d_end,
                                                   double atol) {
  if (!(d_beg.dlh * d_end.dlh < 0))
    throw std::runtime_error("Brents called with endpoints which don't bracket");
  struct pt { root_location_t rl; dlh_t f; };
  pt a{beg, d_beg}, b{end, d_end}, c{end, d_end};
  double step = b.rl.brlen_ratio - a.rl.brlen_ratio, prev_step = step;
  for (size_t it = 0; it < 64; ++it) {
    if (b.f.dlh * c.f.dlh > 0.0) {   // c must sit across the root from b
      c = a;
      step = prev_step = b.rl.brlen_ratio - a.rl.brlen_ratio;
    }
    if (std::fabs(c.f.dlh) < std::fabs(b.f.dlh)) {   // keep b the better of the two
      a = b; b = c; c = a;
    }
    const double tol = 2.0 * std::fabs(b.rl.brlen_ratio) * std::numeric_limits<double>::epsilon() +
                       0.5 * atol;
    const double half = 0.5 * (c.rl.brlen_ratio - b.rl.brlen_ratio);
    if (std::fabs(half) <= tol || std::fabs(b.f.dlh) <= 1e-12) return {b.rl, b.f.lh};
    if (std::fabs(prev_step) >= tol && std::fabs(a.f.dlh) > std::fabs(b.f.dlh)) {
      const double s = b.f.dlh / a.f.dlh;
      double p, q;
      if (std::fabs(a.rl.brlen_ratio - c.rl.brlen_ratio) < 1e-12) {   // secant
        p = 2.0 * half * s;
        q = 1.0 - s;
      } else {                                                        // inverse quadratic
        const double qa = a.f.dlh / c.f.dlh, r = b.f.dlh / c.f.dlh;
        p = s * (2.0 * half * qa * (qa - r) - (b.rl.brlen_ratio - a.rl.brlen_ratio) * (r - 1.0));
        q = (qa - 1.0) * (r - 1.0) * (s - 1.0);
      }
      if (p > 0.0) q = -q;
      p = std::fabs(p);
      const double lim1 = 3.0 * half * q - std::fabs(half * q), lim2 = std::fabs(prev_step * q);
      if (2.0 * p < std::min(lim1, lim2)) {
        prev_step = step;
        step = p / q;
      } else {
        step = prev_step = half;
      }
    } else {
      step = prev_step = half;
    }
    a = b;
    if (std::fabs(step) > tol) b.rl.brlen_ratio += step;
    else b.rl.brlen_ratio += half >= 0.0 ? tol : -tol;
    b.f = compute_dlh(b.rl);
  }
  throw std::runtime_error("Brents method failed to converge");
}

// src/model.cpp:679-794.  The reference's opening -- compute_lh_root(root), compute_dlh at
// alpha = 0 and at alpha = 1 -- is five independent positions of one branch: one launch here
// (same values, same checks in the same order, the partition left at alpha = 1 as there).
// The scan that follows when both ends have the same sign takes its levels one launch each.
root_location_t model_t::optimize_alpha(const root_location_t &root, double atol) {
  root_location_t beg{root}, end{root};
  beg.brlen_ratio = 0.0;
  end.brlen_ratio = 1.0;
  dlh_t d_beg, d_end;
  {
    double pb[2], pe[2], sb, se;
    dlh_positions(beg, pb, &sb);
    dlh_positions(end, pe, &se);
    const std::vector<double> lh = root_lh_at(root, {root.brlen_ratio, pb[0], pb[1], pe[0], pe[1]});
    // (the reference's text for this check, src/model.cpp:681-684)
    if (std::isnan(lh[0])) throw std::runtime_error("initial likelihood calculation is not finite");
    d_beg = dlh_from(lh[2], lh[1], sb, beg);
    d_end = dlh_from(lh[4], lh[3], se, end);
  }
  if (std::isnan(d_beg.dlh) || std::isnan(d_end.dlh))
    throw std::runtime_error("Initial derivatives failed when optimizing alpha: " +
                             std::to_string(root.saved_brlen));
  root_location_t best_endpoint = d_beg.lh >= d_end.lh ? beg : end;
  dlh_t lh_best_endpoint = d_beg.lh >= d_end.lh ? d_beg : d_end;
  if (std::fabs(d_beg.dlh) < atol || std::fabs(d_end.dlh) < atol) return best_endpoint;
  if (d_beg.dlh * d_end.dlh < 0.0) {
    auto mid = brents(beg, d_beg, end, d_end, atol);
    return lh_best_endpoint.lh > mid.second ? best_endpoint : mid.first;
  }
  // same sign at both ends: scan alpha = k/2, k/4, ... k/32 (odd k) for a sign
  // change and solve on both sides of it.  The positions of a level do not depend on each
  // other: they are evaluated together (four derivatives per launch) and looked at in the
  // reference's order, k ascending -- the first sign change decides, as there.
  const bool both_pos = d_beg.dlh > 0.0 && d_end.dlh > 0.0;
  dlh_t best_mid_lh{-std::numeric_limits<double>::infinity(), 0};
  root_location_t best_mid;
  bool found_mid = false;
  for (size_t parts = 2; parts <= 32; parts *= 2) {
    std::vector<root_location_t> mids;
    for (size_t k = 1; k <= parts; k += 2) {
      root_location_t mid{beg};
      mid.brlen_ratio = 1.0 / (double)parts * k;
      mids.push_back(mid);
    }
    std::vector<dlh_t> d_mids;
    for (size_t lo = 0; lo < mids.size(); lo += 4) {
      const std::vector<root_location_t> part(mids.begin() + (std::ptrdiff_t)lo,
                                              mids.begin() + (std::ptrdiff_t)std::min(lo + 4, mids.size()));
      const auto d = compute_dlh_many(part);
      d_mids.insert(d_mids.end(), d.begin(), d.end());
    }
    for (size_t j = 0; j < mids.size(); ++j) {
      const root_location_t &mid = mids[j];
      const dlh_t d_mid = d_mids[j];
      if (std::fabs(d_mid.dlh) < atol && best_mid_lh.lh < d_mid.lh) {
        best_mid_lh = d_mid;
        best_mid = mid;
        found_mid = true;
      }
      if ((both_pos && d_mid.dlh < 0.0) || (!both_pos && d_mid.dlh > 0.0)) {
        auto r1 = brents(beg, d_beg, mid, d_mid, atol);
        auto r2 = brents(mid, d_mid, end, d_end, atol);
        if (lh_best_endpoint.lh < best_mid_lh.lh) {
          lh_best_endpoint = best_mid_lh;
          best_endpoint = best_mid;
        }
        if (r1.second < r2.second) return lh_best_endpoint.lh >= r2.second ? best_endpoint : r2.first;
        return lh_best_endpoint.lh >= r1.second ? best_endpoint : r1.first;
      }
    }
  }
  if (found_mid) return best_mid;
  return both_pos ? end : beg;
}

std::vector<root_location_t> model_t::suggest_roots_lh(size_t min, double ratio) {
  std::vector<std::pair<root_location_t, double>> v;
  bool all_dna = true;
  for (auto p : _partitions) all_dna = all_dna && rdamd_partition_states(p) == 4;
  if (_partitions.size() == 1 && _sweep_msa.size() == 1) {
    // all-directions CLV cache: the values of the move_root sweep, bit for bit,
    // for 3(n-2) + (2n-3) operations; the model's own partition is not disturbed
    auto lh = compute_all_root_lh_directional();
    for (size_t i = 0; i < lh.size(); ++i) v.emplace_back(_tree.roots()[i], lh[i]);
  } else if (all_dna) {   // every root in one fused launch
    auto lh = compute_all_root_lh_batched();
    for (size_t i = 0; i < lh.size(); ++i) v.emplace_back(_tree.roots()[i], lh[i]);
  } else {
    for (auto rl : _tree.roots()) {
      move_root(rl);
      v.emplace_back(rl, compute_lh_root(rl));
    }
  }
  size_t keep = std::max((size_t)(v.size() * ratio), min);
  keep = std::min(keep, v.size());
  std::partial_sort(v.begin(), v.begin() + (std::ptrdiff_t)keep, v.end(),
                    [](const auto &a, const auto &b) { return a.second > b.second; });
  std::vector<root_location_t> out;
  for (size_t i = 0; i < keep; ++i) out.push_back(v[i].first);
  return out;
}

std::pair<root_location_t, double> model_t::optimize_root_location(size_t min_roots,
                                                                   double root_ratio) {
  std::pair<root_location_t, double> best;
  best.second = -std::numeric_limits<double>::infinity();
  for (auto &rl : suggest_roots_lh(min_roots, root_ratio)) {
    move_root(rl);
    rl = optimize_alpha(rl, 1e-14);
    double lh = compute_lh_root(rl);
    if (lh > best.second) best = {rl, lh};
  }
  return best;
}

// ---- work assignment -------------------------------------------------------------
void model_t::assign_indicies() {
  _assigned_idx.resize(_tree.root_count());
  std::iota(_assigned_idx.begin(), _assigned_idx.end(), 0);
}

std::vector<size_t> model_t::shuffle_root_indicies() {
  std::vector<size_t> idx(_tree.root_count());
  std::iota(idx.begin(), idx.end(), 0);
  std::shuffle(idx.begin(), idx.end(), _random_engine);
  return idx;
}

std::vector<size_t> model_t::suggest_root_indicies_midpoint() const {
  std::vector<size_t> ids;
  for (const auto &rl : _tree.rank_midpoints()) ids.push_back(rl.id);
  return ids;
}

std::vector<size_t> model_t::suggest_root_indicies_modified_mad() const {
  std::vector<size_t> ids;
  for (const auto &rl : _tree.rank_modified_mad()) ids.push_back(rl.id);
  return ids;
}

// src/model.cpp:1809-1865: the first max(root_count * root_ratio, min_roots)
// roots of the chosen ordering are the search's starting points; the ones a
// resumed run already holds are dropped, the rest is chunked over the ranks.
void model_t::assign_indicies_by_rank_search(size_t min_roots, double root_ratio, size_t rank,
                                             size_t num_tasks, initial_root_strategy init_root,
                                             const std::vector<size_t> &completed) {
  std::vector<size_t> order;
  switch (init_root) {
    case initial_root_strategy::random: order = shuffle_root_indicies(); break;
    case initial_root_strategy::midpoint: order = suggest_root_indicies_midpoint(); break;
    case initial_root_strategy::modified_mad: order = suggest_root_indicies_modified_mad(); break;
  }
  const size_t root_count = std::min(
      std::max(static_cast<size_t>(_tree.root_count() * root_ratio), min_roots), _tree.root_count());
  if (root_count < completed.size())
    throw std::runtime_error("There are too many results in the checkpoint for this search. Is "
                             "the checkpoint corrupted?");
  std::vector<size_t> done(completed);
  std::sort(done.begin(), done.end());
  const size_t work_left = root_count - done.size();
  std::vector<size_t> left;
  for (size_t i : order)
    if (!std::binary_search(done.begin(), done.end(), i)) left.push_back(i);
  // (as in the reference, the chunk bounds come from work_left while the list
  // still holds every unfinished root of the ordering)
  const size_t chunk = work_left / num_tasks, mod = work_left % num_tasks;
  const size_t beg = chunk * rank + std::min(mod, rank);
  const size_t end = std::min(chunk * (rank + 1) + std::min(mod, rank + 1), left.size());
  _assigned_idx.assign(left.begin() + (std::ptrdiff_t)std::min(beg, end),
                       left.begin() + (std::ptrdiff_t)end);
}

void model_t::assign_indicies_by_rank_exhaustive(size_t rank, size_t num_tasks,
                                                 const std::vector<size_t> &completed) {
  if (_tree.root_count() < completed.size())
    throw std::runtime_error("There are too many results in the checkpoint for this tree, are "
                             "you sure the checkpoint matches?");
  std::vector<size_t> done(completed);
  std::sort(done.begin(), done.end());
  std::vector<size_t> left;
  for (size_t i = 0; i < _tree.root_count(); ++i)
    if (!std::binary_search(done.begin(), done.end(), i)) left.push_back(i);
  const size_t chunk = left.size() / num_tasks, mod = left.size() % num_tasks;
  const size_t beg = chunk * rank + std::min(mod, rank);
  const size_t end = chunk * (rank + 1) + std::min(mod, rank + 1);
  _assigned_idx.assign(left.begin() + (std::ptrdiff_t)beg, left.begin() + (std::ptrdiff_t)end);
}

// ---- parameter optimisation (src/model.cpp:1430-1522, :1925-1984) ------------------
// The reference evaluates the objective once and then once per parameter for
// the one-sided finite-difference gradient (h = max(eps |x_i|, eps)), each a
// full traversal with different parameters.  Here those 1 + n evaluations are
// one rdamd_evaluate_batch call on a schedule compiled once per optimize_params.
double model_t::bfgs_params(model_params_t &initial, size_t pi, bfgs_target what,
                            rdamd_schedule_t *sched, batch_combiner_t *combiner, double p_min, double p_max,
                            double epsilon, double pgtol, double factor) {
  rdamd_partition_t *part = _partitions[pi];
  const unsigned R = rdamd_partition_rate_cats(part);
  int n = (int)initial.size();
  // state of the partition that this optimiser does not vary
  const double *cur_subst = rdamd_partition_subst_params(part, 0);
  const double *cur_freqs = rdamd_partition_frequencies(part, 0);
  const unsigned K = rdamd_partition_states(part), NP = K * K - K;
  const model_params_t base_subst(cur_subst, cur_subst + NP), base_freqs(cur_freqs, cur_freqs + K);
  const model_params_t base_rates(_rate_rates[pi]);

  auto apply = [&](const model_params_t &x) {   // set_func of the reference
    if (what == bfgs_target::rates) set_subst_rates(pi, x);
    else if (what == bfgs_target::freqs) set_freqs_all_free(pi, x);
    else set_gamma_rates(pi, x);
  };
  // -lnL for a list of parameter vectors, one fused launch
  auto objective = [&](const std::vector<model_params_t> &xs) {
    const size_t m = xs.size();
    std::vector<const rdamd_schedule_t *> scheds(m, sched);
    std::vector<double> subst(m * NP), freqs(m * K), rates(m * R), weights(m * R), out(m);
    for (size_t j = 0; j < m; ++j) {
      model_params_t s = base_subst, f = base_freqs, r = base_rates;
      if (what == bfgs_target::rates) s = xs[j];
      else if (what == bfgs_target::freqs) {
        f = xs[j];
        double sum = 0.0;
        for (auto v : f) sum += v;
        for (auto &v : f) v /= sum;
      } else if (_rate_category_types[pi] != rate_category::FREE) {
        rdamd_compute_gamma_cats(xs[j][0], R, r.data(), RDAMD_GAMMA_RATES_MEDIAN);
      }
      std::copy(s.begin(), s.end(), subst.begin() + j * NP);
      std::copy(f.begin(), f.end(), freqs.begin() + j * K);
      std::copy(r.begin(), r.end(), rates.begin() + j * R);
      std::copy(_rate_weights[pi].begin(), _rate_weights[pi].end(), weights.begin() + j * R);
    }
    if (_conductor) {   // meets the other candidates' requests in the round's launch; summed over the group
      _conductor->objective(_worker, (unsigned)m, sched, subst.data(), freqs.data(), rates.data(), weights.data(),
                            out.data());
    } else if (combiner) {   // meets the other candidates' requests in one launch
      combiner->evaluate((unsigned)m, sched, subst.data(), freqs.data(), rates.data(),
                          weights.data(), out.data());
    } else if (_reduce && _reduce_device) {
      ++_n_collectives;
      // site-sharded: the per-block lnLs stay on the device, the group's sum is queued
      // behind the batch on the partition's stream (the guard's two words behind the lnLs),
      // one copy brings the sums back
      double *d = reduce_scratch(m + kGuardWords);
      hipStream_t st = (hipStream_t)rdamd_partition_stream(part);
      if (rdamd_evaluate_batch_device(part, (unsigned)m, scheds.data(), subst.data(), freqs.data(),
                                      rates.data(), weights.data(), d) != RDAMD_SUCCESS)
        fail("evaluate_batch");
      guard_fill(_h_reduce + m);
      if (hipMemcpyAsync(d + m, _h_reduce + m, kGuardWords * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess ||
          _reduce(d, (unsigned)(m + kGuardWords), st, _reduce_user) != RDAMD_SUCCESS ||
          hipMemcpyAsync(_h_reduce, d, (m + kGuardWords) * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
          hipStreamSynchronize(st) != hipSuccess)
        throw std::runtime_error(std::string("site-group reduction failed: ") + rdamd_errmsg());
      guard_check(_h_reduce, m);
      std::copy(_h_reduce, _h_reduce + m, out.begin());
    } else {
      if (rdamd_evaluate_batch(part, (unsigned)m, scheds.data(), subst.data(), freqs.data(),
                               rates.data(), weights.data(), out.data()) != RDAMD_SUCCESS)
        fail("evaluate_batch");
      reduce_values(out.data(), m);
    }
    _objective_batches += 1;
    _objective_evals += m;
    for (auto &v : out) {
      if (std::isnan(v)) throw std::runtime_error("lh at root is not a number");
      v = -v;
    }
    return out;
  };

  int task = 1 /* START */, csave = 0, iprint = -1, m_corr = 20;
  // The reference evaluates the starting point before it calls the optimiser (src/model.cpp:1447)
  // and again as the value of its first FG request; f is deterministic and a job's value does not
  // depend on its launch, so the second one IS the first: no launch of its own for it (a
  // single-job launch per optimised block and round, a fifth of a lock-stepped search's launches)
  // unless the optimiser moved the point before asking.
  double score = 0.0, initial_score = 0.0;
  bool have_initial = false;
  std::vector<double> gradient((size_t)n, 0.0);
  std::vector<double> wa((2 * (size_t)m_corr + 5) * (size_t)n + 12 * (size_t)m_corr * ((size_t)m_corr + 1), 0.0);
  std::vector<int> iwa(3 * (size_t)n, 0), bound_type((size_t)n, 2);
  std::vector<double> x(initial), lo((size_t)n, p_min), hi((size_t)n, p_max);
  int lsave[4] = {0, 0, 0, 0}, isave[44] = {0};
  double dsave[29] = {0};
  // f(x) is deterministic: a point evaluated by the last FG request is not launched again
  model_params_t last_x;
  double last_f = 0.0;
  auto value_at = [&](const model_params_t &at) {
    if (at == last_x) return last_f;
    return objective({at})[0];
  };
  for (size_t iters = 0; iters < 500; ++iters) {
    _setulb(&n, &m_corr, x.data(), lo.data(), hi.data(), bound_type.data(), &score,
            gradient.data(), &factor, &pgtol, wa.data(), iwa.data(), &task, &iprint, &csave,
            lsave, isave, dsave);
    ++_n_lbfgsb_iters;
    const bool fg = task >= 10 && task <= 15;   // IS_FG, lib/lbfgsb/lbfgsb.h:84-86
    if (fg) {
      std::vector<model_params_t> xs(1, x);
      std::vector<double> h((size_t)n);
      for (int i = 0; i < n; ++i) {
        h[i] = std::max(epsilon * std::fabs(x[i]), epsilon);
        xs.push_back(x);
        xs.back()[i] += h[i];
      }
      auto f = objective(xs);
      score = f[0];
      last_x = x;
      last_f = score;
      if (!have_initial) {
        initial_score = x == initial ? score : objective({initial})[0];
        have_initial = true;
      }
      for (int i = 0; i < n; ++i) gradient[i] = (f[i + 1] - score) / h[i];
    } else {
      // the reference re-evaluates after every return (src/model.cpp:1500-1503);
      // NEW_X hands back the point of the last FG request, whose value is known
      score = value_at(x);
      if (task != 2 /* NEW_X */) break;
    }
  }
  score = value_at(x);
  if (!have_initial) initial_score = objective({initial})[0];   // (the optimiser never asked for a value)
  if (initial_score >= score) initial = x;   // improved (scores are -lnL)
  apply(initial);
  return score;
}

void model_t::optimize_params(std::vector<partition_parameters_t> &params,
                              const root_location_t &rl, double pgtol, double factor,
                              bool optimize_gamma) {
  if (!_setulb)
    throw std::runtime_error("optimize_params: no L-BFGS-B entry point set (set_lbfgsb)");
  auto sc = _tree.generate_operations(rl);
  if (!_combiners.empty() && _combiners.size() != _partitions.size())
    throw std::runtime_error("optimize_params: one batch combiner per partition is required");
  if (!_combiners.empty() && _reduce)
    throw std::runtime_error("optimize_params: the candidates of a site-sharded model meet in rounds "
                             "(lockstep_conductor.hpp), not in batch combiners");
  if (_conductor && _partitions.size() != 1)
    throw std::runtime_error("optimize_params: lock step in rounds takes single-partition models");
  for (size_t i = 0; i < _partitions.size(); ++i) {
    // The batched objective runs on the fused evaluators: 4-state and binary data, and 20
    // states with up to eight rate categories (the 381 finite-difference evaluations of a
    // 20-state rate matrix are one launch of fused20_eval_kernel, src/model.cpp:1490-1502).
    const unsigned st = rdamd_partition_states(_partitions[i]);
    if (st != 4 && st != 2 && !(st == 20 && rdamd_partition_rate_cats(_partitions[i]) <= 8))
      throw std::runtime_error("optimize_params: the batched objective handles 4-state and binary data, and "
                               "20-state data with up to 8 rate categories");
    batch_combiner_t *combiner = _combiners.empty() ? nullptr : _combiners[i];
    // (lock step: this candidate is inside partition i's objective phase from here on)
    batch_combiner_t::scope_t in_lockstep(combiner);
    set_subst_rates(i, params[i].subst_rates);
    set_freqs_all_free(i, params[i].freqs);
    set_gamma_rates(i, params[i].gamma_alpha);
    if (_rate_category_types[i] == rate_category::FREE) set_gamma_weights(i, params[i].gamma_weights);
    auto destroy = [&](rdamd_schedule_t *s) {
      if (combiner) combiner->schedule_destroy(s);
      else rdamd_schedule_destroy(s);
    };
    rdamd_schedule_t *sched =
        _conductor ? rdamd_schedule_create(_conductor->shared(), std::get<0>(sc).data(), (unsigned)std::get<0>(sc).size(),
                                           std::get<1>(sc).data(), std::get<2>(sc).data(),
                                           (unsigned)std::get<1>(sc).size()) :
        combiner ? combiner->schedule_create(std::get<0>(sc).data(), (unsigned)std::get<0>(sc).size(),
                                             std::get<1>(sc).data(), std::get<2>(sc).data(),
                                             (unsigned)std::get<1>(sc).size())
                 : rdamd_schedule_create(_partitions[i], std::get<0>(sc).data(),
                                         (unsigned)std::get<0>(sc).size(), std::get<1>(sc).data(),
                                         std::get<2>(sc).data(), (unsigned)std::get<1>(sc).size());
    if (!sched) fail("schedule_create");
    try {
      bfgs_params(params[i].subst_rates, i, bfgs_target::rates, sched, combiner, 1e-4, 1e4, 1e-4, pgtol, factor);
      bfgs_params(params[i].freqs, i, bfgs_target::freqs, sched, combiner, 1e-4, 1.0 - 1e-4 * 3, 1e-4, pgtol, factor);
      if (optimize_gamma && !_rate_user_init[i] && _rate_category_types[i] != rate_category::FREE)
        bfgs_params(params[i].gamma_alpha, i, bfgs_target::gamma, sched, combiner, 0.2, 10000.0, 1e-4, pgtol, factor);
    } catch (...) {
      destroy(sched);
      throw;
    }
    destroy(sched);
  }
}

void model_t::progress_t::step(const char *what) {
  const size_t i = ++done;
  const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
  const double etc_h = total > i ? elapsed / (double)i * (double)(total - i) / 3600.0 : 0.0;
  std::printf("[%.2f] %s %zu / %zu, ETC: %0.2fh\n", elapsed, what, i, total, etc_h);
  std::fflush(stdout);
}

// ---- heuristic search (src/model.cpp:1008-1137) ---------------------------------------
std::pair<root_location_t, double> model_t::search(size_t min_roots, double root_ratio,
                                                   double atol, double pgtol, double brtol,
                                                   double factor,
                                                   std::vector<rd_result_t> *results) {
  if (!_setulb && !_optimizer)
    throw std::runtime_error("search: no parameter optimiser set (set_lbfgsb)");
  set_subst_rates_uniform();
  set_empirical_freqs();
  std::vector<rd_result_t> local;
  std::vector<std::vector<partition_parameters_t>> local_params;
  for (auto rl_index : _assigned_idx) {
    root_location_t rl = _tree.root_location(rl_index);
    set_subst_rates_uniform();
    set_empirical_freqs();
    std::vector<partition_parameters_t> params, saved;
    for (size_t p = 0; p < _partitions.size(); ++p)
      params.push_back(make_partition_parameters(rdamd_partition_states(_partitions[p]),
                                                 _rate_category_types[p],
                                                 rdamd_partition_rate_cats(_partitions[p])));
    root_location_t cur_best_rl = rl;
    double cur_best_lh = -std::numeric_limits<double>::infinity();
    for (size_t iter = 0; iter < 1000; ++iter) {
      saved = params;
      if (_optimizer) _optimizer(*this, params, rl, pgtol, factor, true);
      else optimize_params(params, rl, pgtol, factor, true);
      compute_lh(rl);   // CLVs consistent with the new parameters before roots are moved
      auto cur = optimize_root_location(min_roots, root_ratio);
      if (cur.second < cur_best_lh) {   // no progress: restore and give up
        set_model_params(saved);
        params = saved;
        break;
      }
      if (_early_stop && rl.edge == cur.first.edge &&
          std::fabs(rl.brlen_ratio - cur.first.brlen_ratio) < brtol) {
        cur_best_rl = cur.first; cur_best_lh = cur.second;
        break;
      }
      if (std::fabs(cur.second - cur_best_lh) < atol) {
        cur_best_rl = cur.first; cur_best_lh = cur.second;
        break;
      }
      cur_best_rl = cur.first; cur_best_lh = cur.second;
      rl = cur_best_rl;
    }
    if (_checkpoint) _checkpoint->write({cur_best_rl.id, cur_best_lh, cur_best_rl.brlen_ratio}, params);
    if (_progress) _progress->step("Stage");
    local.push_back({cur_best_rl.id, cur_best_lh, cur_best_rl.brlen_ratio});
    local_params.push_back(params);
  }
  root_location_t best_rl;
  double best_llh = -std::numeric_limits<double>::infinity();
  for (size_t i = 0; i < local.size(); ++i)
    if (local[i].llh > best_llh) {
      best_llh = local[i].llh;
      best_rl = _tree.root_location(local[i].root_id);
      best_rl.brlen_ratio = local[i].alpha;
      set_model_params(local_params[i]);
    }
  if (!local.empty()) compute_lh(best_rl);
  if (results) *results = local;
  return {best_rl, best_llh};
}

// ---- exhaustive outer loop (src/model.cpp:1139-1272) ----------------------------------
std::pair<root_location_t, double> model_t::exhaustive_search(double atol, double pgtol,
                                                              double brtol, double factor,
                                                              std::vector<rd_result_t> *results) {
  root_location_t best_rl;
  double best_llh = -std::numeric_limits<double>::infinity();
  for (auto rl_index : _assigned_idx) {
    root_location_t rl = _tree.root_location(rl_index);
    set_subst_rates_uniform();
    set_empirical_freqs();
    _tree.root_by(rl);
    // (a caller-supplied optimiser may read any CLV; optimize_params reads none)
    if (_optimizer) compute_lh(rl);
    else compute_lh_for_root_steps(rl);
    std::vector<partition_parameters_t> params;
    for (size_t p = 0; p < _partitions.size(); ++p)
      params.push_back(make_partition_parameters(rdamd_partition_states(_partitions[p]),
                                                 _rate_category_types[p],
                                                 rdamd_partition_rate_cats(_partitions[p])));
    root_location_t cur_best_rl = rl;
    double cur_best_llh = -std::numeric_limits<double>::infinity();
    for (size_t iter = 0; iter < 1000; ++iter) {
      if (_optimizer) _optimizer(*this, params, rl, pgtol, factor, iter % 10 == 0);
      else if (_setulb) optimize_params(params, rl, pgtol, factor, iter % 10 == 0);
      if (std::fabs(compute_lh_for_root_steps(rl) - cur_best_llh) < atol) break;
      root_location_t cur_rl;
      double cur_llh;
      {   // (lock step: this candidate's root-only steps may now meet the others')
        root_combiner_t::scope_t placing(_root_combiner);
        cur_rl = optimize_alpha(rl, brtol);
        cur_llh = compute_lh_root(cur_rl);
      }
      if (_early_stop && std::fabs(rl.brlen_ratio - cur_rl.brlen_ratio) < brtol) {
        cur_best_rl = cur_rl;
        cur_best_llh = cur_llh;
        break;
      }
      if ((cur_llh - cur_best_llh) < atol) {
        if (cur_llh > cur_best_llh) { cur_best_rl = cur_rl; cur_best_llh = cur_llh; }
        break;
      }
      if (cur_llh > cur_best_llh) { cur_best_rl = cur_rl; cur_best_llh = cur_llh; }
      rl = cur_rl;
    }
    if (_checkpoint) _checkpoint->write({cur_best_rl.id, cur_best_llh, cur_best_rl.brlen_ratio}, params);
    if (_progress) _progress->step("Step");
    if (results) results->push_back({cur_best_rl.id, cur_best_llh, cur_best_rl.brlen_ratio});
    if (cur_best_llh > best_llh) { best_rl = cur_best_rl; best_llh = cur_best_llh; }
  }
  return {best_rl, best_llh};
}

std::vector<double> model_t::likelihood_weight_ratios(const std::vector<rd_result_t> &results) {
  double max_llh = -std::numeric_limits<double>::infinity();
  for (const auto &r : results) max_llh = std::max(max_llh, r.llh);
  double total = 0.0;
  for (const auto &r : results) total += std::exp(r.llh - max_llh);
  std::vector<double> out;
  for (const auto &r : results) out.push_back(std::exp(r.llh - max_llh) / total);
  return out;
}

}  // namespace rdamd
