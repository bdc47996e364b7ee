// C wrappers over checkpoint_t (declared in include/root_digger_amd.h).
#include <cstring>

#include "checkpoint.hpp"
#include "common.hpp"

struct rdamd_checkpoint {
  rdamd::checkpoint_t ckp;
  rdamd::cli_options_t loaded;                      // backs the strings of load_options
  std::vector<rdamd_ratehet_opts_t> loaded_cats;
  std::vector<rdamd::checkpoint_record_t> results;  // snapshot of the last read_results
  explicit rdamd_checkpoint(const std::string &prefix) : ckp(prefix) {}
};

#define GUARD(failret, ...)                             \
  try {                                                 \
    rdamd::clear_error();                               \
    __VA_ARGS__                                         \
  } catch (const std::exception &e) {                   \
    rdamd::set_error(60, "%s", e.what());               \
    return failret;                                     \
  }

namespace {
std::string str(const char *s) { return s ? std::string(s) : std::string(); }

std::vector<rdamd::partition_parameters_t> unflatten(unsigned n_partitions, const uint64_t *counts,
                                                     const double *values) {
  std::vector<rdamd::partition_parameters_t> pps(n_partitions);
  for (unsigned p = 0; p < n_partitions; ++p) {
    rdamd::model_params_t *dst[4] = {&pps[p].subst_rates, &pps[p].freqs, &pps[p].gamma_alpha,
                                     &pps[p].gamma_weights};
    for (int k = 0; k < 4; ++k) {
      dst[k]->assign(values, values + counts[4 * p + k]);
      values += counts[4 * p + k];
    }
  }
  return pps;
}
}  // namespace

rdamd::checkpoint_t *rdamd_checkpoint_cpp(rdamd_checkpoint_t *c) { return c ? &c->ckp : nullptr; }

extern "C" {

rdamd_checkpoint_t *rdamd_checkpoint_open(const char *prefix) {
  GUARD(nullptr, { return new rdamd_checkpoint(str(prefix)); })
}
void rdamd_checkpoint_close(rdamd_checkpoint_t *c) { delete c; }
int rdamd_checkpoint_existing(const rdamd_checkpoint_t *c) { return c->ckp.existing_checkpoint(); }
const char *rdamd_checkpoint_filename(rdamd_checkpoint_t *c) {
  static thread_local std::string name;
  name = c->ckp.get_filename();
  return name.c_str();
}

int rdamd_checkpoint_save_options(rdamd_checkpoint_t *c, const rdamd_cli_options_t *o) {
  GUARD(RDAMD_FAILURE, {
    rdamd::cli_options_t x;
    x.msa_filename = str(o->msa_filename); x.tree_filename = str(o->tree_filename);
    x.prefix = str(o->prefix); x.prefix_dir = str(o->prefix_dir);
    x.model_filename = str(o->model_filename); x.freqs_filename = str(o->freqs_filename);
    x.partition_filename = str(o->partition_filename); x.data_type = str(o->data_type);
    x.model_string = str(o->model_string);
    x.rate_cats.clear();
    for (uint64_t i = 0; i < o->n_rate_cats; ++i) {
      rdamd::ratehet_opts_t rc;
      rc.type = (rdamd::param_type)o->rate_cats[i].type;
      rc.rate_category_type = (rdamd::rate_category)o->rate_cats[i].rate_category_type;
      rc.rate_cats = o->rate_cats[i].rate_cats;
      rc.alpha_init = o->rate_cats[i].alpha_init != 0;
      rc.alpha = o->rate_cats[i].alpha;
      x.rate_cats.push_back(rc);
    }
    x.seed = o->seed; x.min_roots = o->min_roots; x.threads = o->threads;
    x.root_ratio = o->root_ratio; x.abs_tolerance = o->abs_tolerance; x.factor = o->factor;
    x.br_tolerance = o->br_tolerance; x.bfgs_tol = o->bfgs_tol;
    x.silent = o->silent; x.exhaustive = o->exhaustive; x.echo = o->echo;
    x.invariant_sites = o->invariant_sites;
    x.early_stop = (rdamd::early_stop_t)o->early_stop;
    x.initial_root_strategy = (rdamd::initial_root_strategy_t)o->initial_root_strategy;
    c->ckp.save_options(x);
    return RDAMD_SUCCESS;
  })
}

int rdamd_checkpoint_load_options(rdamd_checkpoint_t *c, rdamd_cli_options_t *o) {
  GUARD(RDAMD_FAILURE, {
    if (!c->ckp.existing_checkpoint()) return RDAMD_FAILURE;
    c->ckp.load_options(c->loaded);
    const rdamd::cli_options_t &x = c->loaded;
    o->msa_filename = x.msa_filename.c_str(); o->tree_filename = x.tree_filename.c_str();
    o->prefix = x.prefix.c_str(); o->prefix_dir = x.prefix_dir.c_str();
    o->model_filename = x.model_filename.c_str(); o->freqs_filename = x.freqs_filename.c_str();
    o->partition_filename = x.partition_filename.c_str(); o->data_type = x.data_type.c_str();
    o->model_string = x.model_string.c_str();
    c->loaded_cats.clear();
    for (const auto &rc : x.rate_cats)
      c->loaded_cats.push_back({(int32_t)rc.type, (int32_t)rc.rate_category_type, rc.rate_cats,
                                rc.alpha_init ? 1 : 0, rc.alpha});
    o->rate_cats = c->loaded_cats.data();
    o->n_rate_cats = c->loaded_cats.size();
    o->seed = x.seed; o->min_roots = x.min_roots; o->threads = x.threads;
    o->root_ratio = x.root_ratio; o->abs_tolerance = x.abs_tolerance; o->factor = x.factor;
    o->br_tolerance = x.br_tolerance; o->bfgs_tol = x.bfgs_tol;
    o->silent = x.silent; o->exhaustive = x.exhaustive; o->echo = x.echo;
    o->invariant_sites = x.invariant_sites;
    o->early_stop = (int)x.early_stop;
    o->initial_root_strategy = (int)x.initial_root_strategy;
    return RDAMD_SUCCESS;
  })
}

int rdamd_checkpoint_write(rdamd_checkpoint_t *c, uint64_t root_id, double llh, double alpha,
                           unsigned int n_partitions, const uint64_t *counts,
                           const double *values) {
  GUARD(RDAMD_FAILURE, {
    c->ckp.write({(size_t)root_id, llh, alpha}, unflatten(n_partitions, counts, values));
    return RDAMD_SUCCESS;
  })
}

int rdamd_checkpoint_read_results(rdamd_checkpoint_t *c, unsigned int *n_results) {
  GUARD(RDAMD_FAILURE, {
    c->results = c->ckp.read_results();
    *n_results = (unsigned)c->results.size();
    return RDAMD_SUCCESS;
  })
}

int rdamd_checkpoint_result(const rdamd_checkpoint_t *c, unsigned int index, uint64_t *root_id,
                            double *llh, double *alpha, unsigned int *n_partitions,
                            uint64_t *n_values) {
  if (index >= c->results.size()) return RDAMD_FAILURE;
  const auto &rec = c->results[index];
  *root_id = rec.first.root_id; *llh = rec.first.llh; *alpha = rec.first.alpha;
  *n_partitions = (unsigned)rec.second.size();
  uint64_t n = 0;
  for (const auto &pp : rec.second)
    n += pp.subst_rates.size() + pp.freqs.size() + pp.gamma_alpha.size() + pp.gamma_weights.size();
  *n_values = n;
  return RDAMD_SUCCESS;
}

int rdamd_checkpoint_result_params(const rdamd_checkpoint_t *c, unsigned int index,
                                   uint64_t *counts, double *values) {
  if (index >= c->results.size()) return RDAMD_FAILURE;
  for (const auto &pp : c->results[index].second) {
    for (const rdamd::model_params_t *v : {&pp.subst_rates, &pp.freqs, &pp.gamma_alpha, &pp.gamma_weights}) {
      *counts++ = v->size();
      values = std::copy(v->begin(), v->end(), values);
    }
  }
  return RDAMD_SUCCESS;
}

int rdamd_checkpoint_needs_cleaning(rdamd_checkpoint_t *c) {
  GUARD(-1, { return c->ckp.needs_cleaning() ? 1 : 0; })
}
int rdamd_checkpoint_clean(rdamd_checkpoint_t *c) {
  GUARD(RDAMD_FAILURE, {
    c->ckp.clean();
    return RDAMD_SUCCESS;
  })
}

uint32_t rdamd_checkpoint_checksum_result(uint64_t root_id, double llh, double alpha) {
  return rdamd::checkpoint_checksum(rdamd::rd_result_t{(size_t)root_id, llh, alpha});
}
uint32_t rdamd_checkpoint_checksum_params(unsigned int n_partitions, const uint64_t *counts,
                                          const double *values) {
  return rdamd::checkpoint_checksum(unflatten(n_partitions, counts, values));
}

}  // extern "C"
