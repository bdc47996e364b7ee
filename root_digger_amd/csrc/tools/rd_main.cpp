// rd_amd: the `rd` command line of the reference (/root/reference/src/main.cpp)
// as a native program on top of the C ABI -- include/root_digger_amd.h is the
// only project header it uses, so it doubles as the example of a whole-program
// caller.  Same option names as the reference; `--lbfgsb <lib>` names the shared
// library that exports the caller's L-BFGS-B entry point `setulb` (lib/lbfgsb of
// the reference builds as one).  Several processes (one per GPU: RANK,
// LOCAL_RANK, WORLD_SIZE in the environment, the way torch.distributed.run or
// any launcher sets them) split the candidate roots like the reference's MPI
// ranks (src/model.cpp:1867-1911) and meet in the <prefix>.ckp file; they
// synchronise over TCP at MASTER_ADDR:MASTER_PORT.
//
// --site-shards G (new here; north star / SURVEY 8e): the ranks form WORLD_SIZE/G
// candidate groups of G adjacent ranks.  A group shares its candidates; each
// member holds one contiguous block of the alignment's columns, and every
// log-likelihood is summed over the group -- on the device over RCCL (default:
// ncclAllGather + a sum in rank order, identical bits on every rank by
// construction; --site-reduce rccl-allreduce: one ncclAllReduce), or through the
// host with --site-reduce host (ranks that share one GPU).  G = WORLD_SIZE is BASELINE config c4's layout (site blocks
// only), 1 < G < WORLD_SIZE config c5's 2-D grid.
//
//   rd_amd --msa aln.fasta --tree t.nwk --prefix out --exhaustive --lbfgsb liblbfgsb.so
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <fstream>
#include <getopt.h>
#include <iostream>
#include <stdexcept>
#include <string>
#include <memory>
#include <thread>
#include <vector>

#include "root_digger_amd.h"
#include "rendezvous.hpp"

namespace {

struct options_t {
  std::string msa, tree, prefix, partition, model, lbfgsb, rate_cats_type = "mean";
  unsigned states = 4, rate_cats = 1, min_roots = 1, workers = 4;
  int lockstep = -1, device = -1, site_shards = 1, lockstep_rounds = -1, lockstep_groups = 0;
  bool site_reduce_host = false, site_allreduce = false, stats = false;
  uint64_t seed = 1;
  double root_ratio = 0.01, atol = 1e-7, bfgstol = 1e-7, brtol = 1e-12, factor = 1e4;
  int early_stop = 0;   // initialized_flag_t: 0 unset, 1 true, 2 false
  int strategy = 2;     // random, midpoint, modified-mad
  bool exhaustive = false, silent = false, clean = false, echo = false, no_checkpoint = false,
       invariant_sites = false;
};

[[noreturn]] void die(const std::string &what) {
  std::cout << "There was an error during processing:\n" << what << std::endl;
  std::exit(1);
}
void need(int rc, const char *what) {
  if (rc != RDAMD_SUCCESS) die(std::string(what) + ": " + rdamd_errmsg());
}

void usage() {
  std::puts(
      "Usage: rd_amd --msa <FILE> --tree <FILE> [--prefix <STRING>] [--exhaustive]\n"
      "  --partition <FILE>  --model <STRING>  --states {2,4}  --rate-cats <N>\n"
      "  --rate-cats-type {mean,median,free}  --seed <N>  --min-roots <N>  --root-ratio <X>\n"
      "  --atol <X>  --brtol <X>  --bfgstol <X>  --factor <X>  --early-stop  --no-early-stop\n"
      "  --initial-root-strategy {random,midpoint,modified-mad}  --threads <N>  --lockstep <N>\n"
      "  --site-shards <G>  --site-reduce {rccl,rccl-allreduce,host}  --lockstep-rounds {0,1}  --lockstep-groups {1,2}  --stats\n"
      "  --lbfgsb <LIB>  --device <N>  --silent  --echo  --clean  --no-checkpoint  --version");
}

options_t parse(int argc, char **argv) {
  static option long_opts[] = {
      {"msa", required_argument, 0, 0},          {"tree", required_argument, 0, 0},
      {"model", required_argument, 0, 0},        {"seed", required_argument, 0, 0},
      {"verbose", no_argument, 0, 0},            {"silent", no_argument, 0, 0},
      {"min-roots", required_argument, 0, 0},    {"root-ratio", required_argument, 0, 0},
      {"atol", required_argument, 0, 0},         {"brtol", required_argument, 0, 0},
      {"bfgstol", required_argument, 0, 0},      {"factor", required_argument, 0, 0},
      {"partition", required_argument, 0, 0},    {"prefix", required_argument, 0, 0},
      {"exhaustive", no_argument, 0, 0},         {"early-stop", no_argument, 0, 0},
      {"no-early-stop", no_argument, 0, 0},      {"rate-cats", required_argument, 0, 0},
      {"rate-cats-type", required_argument, 0, 0}, {"invariant-sites", no_argument, 0, 0},
      {"states", required_argument, 0, 0},       {"initial-root-strategy", required_argument, 0, 0},
      {"threads", required_argument, 0, 0},      {"version", no_argument, 0, 0},
      {"debug", no_argument, 0, 0},              {"mpi-debug", no_argument, 0, 0},
      {"clean", no_argument, 0, 0},              {"echo", no_argument, 0, 0},
      {"help", no_argument, 0, 0},               {"lbfgsb", required_argument, 0, 0},
      {"lockstep", required_argument, 0, 0},     {"device", required_argument, 0, 0},
      {"no-checkpoint", no_argument, 0, 0},      {"site-shards", required_argument, 0, 0},
      {"site-reduce", required_argument, 0, 0},  {"lockstep-rounds", required_argument, 0, 0},
      {"stats", no_argument, 0, 0},              {"lockstep-groups", required_argument, 0, 0},
      {0, 0, 0, 0}};
  options_t o;
  int index = 0;
  while (getopt_long_only(argc, argv, "", long_opts, &index) == 0) {
    const std::string name = long_opts[index].name;
    const char *v = optarg;
    if (name == "msa") o.msa = v;
    else if (name == "tree") o.tree = v;
    else if (name == "model") o.model = v;
    else if (name == "seed") o.seed = std::strtoull(v, nullptr, 10);
    else if (name == "silent") o.silent = true;
    else if (name == "min-roots") o.min_roots = (unsigned)std::atol(v);
    else if (name == "root-ratio") o.root_ratio = std::atof(v);
    else if (name == "atol") o.atol = std::atof(v);
    else if (name == "brtol") o.brtol = std::atof(v);
    else if (name == "bfgstol") o.bfgstol = std::atof(v);
    else if (name == "factor") o.factor = std::atof(v);
    else if (name == "partition") o.partition = v;
    else if (name == "prefix") o.prefix = v;
    else if (name == "exhaustive") o.exhaustive = true;
    else if (name == "early-stop") o.early_stop = 1;
    else if (name == "no-early-stop") o.early_stop = 2;
    else if (name == "rate-cats") o.rate_cats = (unsigned)std::atol(v);
    else if (name == "rate-cats-type") o.rate_cats_type = v;
    else if (name == "invariant-sites") o.invariant_sites = true;
    else if (name == "states") o.states = (unsigned)std::atol(v);
    else if (name == "initial-root-strategy") {
      const std::string s = v;
      o.strategy = s == "random" ? 0 : s == "midpoint" ? 1 : s == "modified-mad" ? 2 : -1;
      if (o.strategy < 0) die("An argument is required for --initial-root-strategy");
    } else if (name == "threads") o.workers = (unsigned)std::atol(v);
    else if (name == "version") { std::puts(rdamd_version()); std::exit(0); }
    else if (name == "clean") o.clean = true;
    else if (name == "echo") o.echo = true;
    else if (name == "help") { usage(); std::exit(0); }
    else if (name == "lbfgsb") o.lbfgsb = v;
    else if (name == "lockstep") o.lockstep = std::atoi(v);
    else if (name == "device") o.device = std::atoi(v);
    else if (name == "no-checkpoint") o.no_checkpoint = true;
    else if (name == "site-shards") o.site_shards = std::atoi(v);
    else if (name == "lockstep-rounds") o.lockstep_rounds = std::atoi(v);
    else if (name == "lockstep-groups") o.lockstep_groups = std::atoi(v);
    else if (name == "stats") o.stats = true;
    else if (name == "site-reduce") {
      const std::string s = v;
      if (s != "rccl" && s != "rccl-allreduce" && s != "host") die("--site-reduce takes rccl, rccl-allreduce or host");
      o.site_reduce_host = s == "host";
      o.site_allreduce = s == "rccl-allreduce";
    }
  }
  return o;
}

int env_int(const char *name, int fallback) {
  const char *v = std::getenv(name);
  return v ? std::atoi(v) : fallback;
}

}  // namespace

static int run(int argc, char **argv);

int main(int argc, char **argv) {
  try {
    return run(argc, argv);
  } catch (const std::exception &e) {
    die(e.what());
  }
}

static int run(int argc, char **argv) {
  using rdamd_tools::rendezvous_t;
  using rdamd_tools::site_group_t;
  const auto start = std::chrono::steady_clock::now();
  options_t o = parse(argc, argv);
  const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1);
  if (o.prefix.empty()) o.prefix = o.msa;
  if (world > 1 && o.no_checkpoint) die("--no-checkpoint: the ranks of a run meet in the checkpoint file");
  need(rdamd_set_device(o.device >= 0 ? o.device : env_int("LOCAL_RANK", 0)), "set_device");
  rendezvous_t ranks(rank, world);
  // candidate groups x site shards (see the header comment): rank = cgroup * G + srank
  const int G = o.site_shards;
  if (G < 1 || world % G) die("--site-shards must divide the number of ranks");
  const int cgroups = world / G, cgroup = rank / G, srank = rank % G;
  if (G > 1 && !o.partition.empty()) die("--site-shards: partition files are not supported");
  std::unique_ptr<site_group_t> host_group;
  rdamd_comm_t *comm = nullptr;
  if (G > 1 && o.site_reduce_host) {
    host_group.reset(new site_group_t(ranks, G));
    // (test hook, see rendezvous.hpp: RDAMD_FAULT_ULP=<rank>:<call>)
    if (const char *f = std::getenv("RDAMD_FAULT_ULP")) {
      int fr = -1;
      unsigned long long fc = 0;
      if (std::sscanf(f, "%d:%llu", &fr, &fc) == 2 && fr == rank) host_group->set_fault(fc);
    }
  } else if (G > 1) {   // the group leader's RCCL id reaches the members over the world star
    char id[128] = {0};
    if (srank == 0) need(rdamd_comm_unique_id(id), "RCCL unique id");
    std::vector<char> all;
    ranks.allgather(id, sizeof id, all);
    comm = rdamd_comm_create(all.data() + sizeof id * (size_t)(cgroup * G), srank, G);
    if (!comm) die(std::string("RCCL communicator: ") + rdamd_errmsg());
    if (o.site_allreduce) need(rdamd_comm_set_sum_mode(comm, RDAMD_COMM_SUM_ALLREDUCE), "sum mode");
  }

  // ---- checkpoint: mpi_create_checkpoint + merge_options_checkpoint, src/main.cpp:335-409
  rdamd_checkpoint_t *ckp = nullptr;
  std::vector<rdamd_ratehet_opts_t> header_cats;
  if (!o.no_checkpoint) {
    if (rank == 0) {
      ckp = rdamd_checkpoint_open(o.prefix.c_str());
      if (!ckp) die(std::string("checkpoint: ") + rdamd_errmsg());
      if (o.clean) {
        need(rdamd_checkpoint_clean(ckp), "checkpoint clean");
        return 0;
      }
      if (!rdamd_checkpoint_existing(ckp)) {
        rdamd_ratehet_opts_t rc{1, o.rate_cats_type == "median" ? 0 : o.rate_cats_type == "free" ? 2 : 1,
                                o.rate_cats, 0, 1.0};
        rdamd_cli_options_t h;
        std::memset(&h, 0, sizeof h);
        h.msa_filename = o.msa.c_str(); h.tree_filename = o.tree.c_str(); h.prefix = o.prefix.c_str();
        h.prefix_dir = ""; h.model_filename = ""; h.freqs_filename = "";
        h.partition_filename = o.partition.c_str(); h.data_type = o.states == 2 ? "bin" : "nt";
        h.model_string = o.model.c_str();
        h.rate_cats = &rc; h.n_rate_cats = 1;
        h.seed = o.seed; h.min_roots = o.min_roots; h.threads = o.workers;
        h.root_ratio = o.root_ratio; h.abs_tolerance = o.atol; h.factor = o.factor;
        h.br_tolerance = o.brtol; h.bfgs_tol = o.bfgstol;
        h.silent = o.silent; h.exhaustive = o.exhaustive; h.invariant_sites = o.invariant_sites;
        h.early_stop = o.early_stop; h.initial_root_strategy = o.strategy;
        need(rdamd_checkpoint_save_options(ckp, &h), "checkpoint save_options");
      }
      if (rdamd_checkpoint_needs_cleaning(ckp) == 1) need(rdamd_checkpoint_clean(ckp), "checkpoint clean");
    } else if (o.clean) {
      return 0;
    }
    ranks.barrier();
    if (rank != 0) {
      ckp = rdamd_checkpoint_open(o.prefix.c_str());
      if (!ckp) die(std::string("checkpoint: ") + rdamd_errmsg());
    }
    rdamd_cli_options_t h;
    if (rdamd_checkpoint_existing(ckp) && rdamd_checkpoint_load_options(ckp, &h) == RDAMD_SUCCESS) {
      if (!o.silent && rank == 0)
        std::cerr << "Loading options from the checkpoint file. Some cli options are ignored. If "
                     "the program is not working, try deleting the checkpoint file\n";
      o.msa = h.msa_filename; o.tree = h.tree_filename; o.partition = h.partition_filename;
      o.model = h.model_string;
      if (h.n_rate_cats) {
        o.rate_cats = (unsigned)h.rate_cats[0].rate_cats;
        o.rate_cats_type = h.rate_cats[0].rate_category_type == 0   ? "median"
                           : h.rate_cats[0].rate_category_type == 2 ? "free" : "mean";
      }
      o.seed = h.seed; o.atol = h.abs_tolerance; o.factor = h.factor; o.brtol = h.br_tolerance;
      o.bfgstol = h.bfgs_tol; o.exhaustive = h.exhaustive != 0; o.min_roots = (unsigned)h.min_roots;
      o.root_ratio = h.root_ratio; o.strategy = h.initial_root_strategy; o.early_stop = h.early_stop;
    }
  }
  // (again, now that a resumed run's saved options are in: a partition file that comes from
  // the checkpoint would build the whole partitioned model on every member of a site group
  // while the group's reduction is installed, and every lnL would be counted G times)
  if (G > 1 && !o.partition.empty())
    die("--site-shards: the checkpoint belongs to a run with a partition file, which a site group does not support");
  if (o.msa.empty()) { std::puts("No MSA was given, please supply an MSA"); usage(); return 1; }
  if (o.tree.empty()) { std::puts("No tree was given, please supply an tree"); usage(); return 1; }

  // ---- tree, alignment, model (src/main.cpp:484-584)
  rdamd_tree_t *tree = rdamd_tree_from_file(o.tree.c_str());
  if (!tree) die(rdamd_errmsg());
  const unsigned roots = rdamd_tree_root_count(tree);
  if (o.min_roots > roots) die("Min roots is larger than the number of roots on the tree");
  const bool early_stop = o.early_stop == 1 || (o.early_stop == 0 && !o.exhaustive);
  const uint64_t *map = o.states == 2 ? rdamd_map_bin : rdamd_map_nt;
  if (!o.model.empty()) {
    rdamd_partition_info_t mi;
    need(rdamd_parse_model_info(o.model.c_str(), &mi), "model string");
    o.rate_cats = mi.ratehet.rate_cats ? (unsigned)mi.ratehet.rate_cats : 1u;
  }
  rdamd_model_t *model = nullptr;
  if (!o.partition.empty()) {
    unsigned n_parts = 0;
    model = rdamd_model_create_partitioned(tree, o.msa.c_str(), o.partition.c_str(), o.states, map,
                                           o.seed, early_stop, &n_parts);
  } else {
    rdamd_ratehet_opts_t rc{1, o.rate_cats_type == "median" ? 0 : o.rate_cats_type == "free" ? 2 : 1,
                            o.rate_cats, 0, 1.0};
    unsigned patterns = 0, columns = 0;
    model = G > 1 ? rdamd_model_create_from_file_block(tree, o.msa.c_str(), o.states, map, &rc, o.seed,
                                                       early_stop, 1, (unsigned)srank, (unsigned)G,
                                                       &patterns, &columns)
                  : rdamd_model_create_from_file_ratehet(tree, o.msa.c_str(), o.states, map, &rc,
                                                         o.seed, early_stop, 1, &patterns);
    if (model && G > 1 && !o.silent)
      std::printf("[rank %d] candidate group %d/%d, site block %d/%d: %u patterns of %u columns\n",
                  rank, cgroup, cgroups, srank, G, patterns, columns);
  }
  if (!model) die(rdamd_errmsg());
  // before anything is evaluated: the empirical frequencies are reduced too
  if (host_group) {
    need(rdamd_model_set_lnl_reducer(model, site_group_t::reducer, host_group.get(), 0), "reducer");
    need(rdamd_model_set_lnl_reducer_abort(model, site_group_t::abort_hook, host_group.get()), "reducer abort");
  } else if (comm)
    need(rdamd_model_set_lnl_reducer(model, rdamd_comm_reducer, comm, 1), "reducer");
  if (o.echo) {
    char *nw = rdamd_tree_newick(tree, 1);
    std::cout << nw << std::endl;
    std::free(nw);
  }
  if (rdamd_model_initialize_partitions(model, 0) != RDAMD_SUCCESS)     // invalid empirical
    need(rdamd_model_initialize_partitions(model, 1), "initialize_partitions");   // frequencies -> uniform
  if (!o.lbfgsb.empty()) {
    void *lib = dlopen(o.lbfgsb.c_str(), RTLD_NOW | RTLD_GLOBAL);
    void *fn = lib ? dlsym(lib, "setulb") : nullptr;
    if (!fn) die("--lbfgsb: " + o.lbfgsb + " does not export setulb");
    rdamd_model_set_lbfgsb(model, fn);
  }
  rdamd_root_location_t rl0;
  need(rdamd_tree_root_location(tree, 0, &rl0), "root_location");
  rdamd_model_compute_lh(model, &rl0);                                   // model.initialize()
  if (ckp && srank == 0) rdamd_model_set_checkpoint(model, ckp);   // one record per candidate
  if (o.lockstep < 0) o.lockstep = !o.lbfgsb.empty() ? 32 : 0;   // (partitioned models lock-step too)
  // A site group's candidates advance in lock step in deterministic rounds (every rank of the
  // group forms the same launches and one collective per round; --lockstep 0: one candidate
  // at a time, a collective per request); free-running replicas would reorder the collectives.
  if (G > 1) o.workers = 0;
  if (o.lockstep_rounds >= 0) rdamd_model_set_lockstep_rounds(model, o.lockstep_rounds);
  if (o.lockstep_groups > 0) rdamd_model_set_lockstep_groups(model, (unsigned)o.lockstep_groups);

  // While the ranks search they exchange nothing over the rendezvous: the end of a
  // connection now means that a rank has died.  Nobody must stay behind inside a
  // collective (or wait for the dead rank's checkpoint records): abort the site group's
  // communicator and leave; closing our own connections passes the news on.
  const auto watch_ranks = [&ranks, comm, rank] {
    ranks.watch([comm, rank] {
      std::fprintf(stderr, "[rank %d] another rank of this run has gone; giving up\n", rank);
      if (comm) rdamd_comm_abort(comm);
      std::this_thread::sleep_for(std::chrono::seconds(comm ? 2 : 0));   // let a blocked reducer fail first
      std::_Exit(3);
    });
  };

  // ---- the search (src/main.cpp:586-635)
  std::vector<uint64_t> ids(roots);
  std::vector<double> llh(roots), alpha(roots);
  unsigned n_results = 0;
  rdamd_root_location_t best;
  double best_llh = -INFINITY;
  std::memset(&best, 0, sizeof best);
  if (o.exhaustive) {
    need(rdamd_model_assign_by_rank_checkpoint(model, (unsigned)cgroup, (unsigned)cgroups, ckp), "assign");
    ranks.barrier();
    watch_ranks();
    if (!o.silent && rank == 0) {
      std::puts("Starting exhaustive search");
      rdamd_model_set_progress(model, 1);
    }
    if (o.lockstep > 0)
      need(rdamd_model_exhaustive_search_lockstep(model, (unsigned)o.lockstep, o.atol, o.bfgstol, o.brtol,
                                                  o.factor, ids.data(), llh.data(), alpha.data(),
                                                  &n_results, &best, &best_llh), "exhaustive_search");
    else if (o.workers > 0)
      need(rdamd_model_exhaustive_search_parallel(model, o.workers, o.atol, o.bfgstol, o.brtol, o.factor,
                                                  ids.data(), llh.data(), alpha.data(), &n_results,
                                                  &best, &best_llh), "exhaustive_search");
    else
      need(rdamd_model_exhaustive_search(model, o.atol, o.bfgstol, o.brtol, o.factor, ids.data(),
                                         llh.data(), alpha.data(), &n_results, &best, &best_llh),
           "exhaustive_search");
  } else {
    if (o.lbfgsb.empty()) die("the heuristic search optimises the model parameters: it needs --lbfgsb");
    need(rdamd_model_assign_by_rank_search(model, o.min_roots, o.root_ratio, (unsigned)cgroup,
                                           (unsigned)cgroups, o.strategy, ckp), "assign");
    ranks.barrier();
    watch_ranks();
    std::vector<uint64_t> mine(roots);
    const int assigned = rdamd_model_assigned(model, mine.data(), roots);
    if (!o.silent && rank == 0) {
      std::puts("Starting root search");
      rdamd_model_set_progress(model, 1);
    }
    need(rdamd_model_search(model, o.min_roots, o.root_ratio, o.atol, o.bfgstol, o.brtol, o.factor,
                            &best, &best_llh), "search");
    if (assigned > 0) {
      ids[0] = best.id; llh[0] = best_llh; alpha[0] = best.brlen_ratio;
      n_results = 1;
    }
  }
  if (o.stats) {   // one line per rank on stderr: what the search launched and summed
    uint64_t ls[4] = {0, 0, 0, 0}, rs[4] = {0, 0, 0, 0}, ct[6] = {0, 0, 0, 0, 0, 0};
    rdamd_model_lockstep_stats(model, ls);
    rdamd_model_round_stats(model, rs);
    rdamd_model_counters(model, ct);
    double rsec[4] = {0, 0, 0, 0};
    rdamd_model_round_seconds(model, rsec);
    const std::chrono::duration<double> took = std::chrono::steady_clock::now() - start;
    uint64_t digest = 1469598103934665603ull;   // FNV-1a over the bits of this rank's (id, lnL, alpha) triples
    auto mix = [&digest](const void *ptr, size_t n) {
      for (size_t i = 0; i < n; ++i) digest = (digest ^ ((const unsigned char *)ptr)[i]) * 1099511628211ull;
    };
    for (unsigned i = 0; i < n_results; ++i) { mix(&ids[i], 8); mix(&llh[i], 8); mix(&alpha[i], 8); }
    std::fprintf(stderr, "[rank %d] stats: candidates=%u results_digest=%016llx lockstep=%d rounds=%llu collectives=%llu redos=%llu "
                         "own_collectives=%llu objective_launches=%llu objective_jobs=%llu root_launches=%llu "
                         "root_steps=%llu round_seconds=%.2f/%.2f/%.2f/%.2f seconds=%.3f\n",
                 rank, n_results, (unsigned long long)digest, o.lockstep, (unsigned long long)rs[0], (unsigned long long)rs[1],
                 (unsigned long long)rs[2], (unsigned long long)rs[3],
                 (unsigned long long)(ls[0] ? ls[0] : ct[0]), (unsigned long long)(ls[1] ? ls[1] : ct[1]),
                 (unsigned long long)ls[2], (unsigned long long)ls[3], rsec[0], rsec[1], rsec[2], rsec[3], took.count());
  }
  ranks.barrier();
  if (rank != 0) {
    rdamd_model_destroy(model);
    if (comm) rdamd_comm_destroy(comm);
    return 0;
  }

  // ---- rank 0: everybody's results from the log (src/model.cpp:1237-1268), the trees
  if (ckp) {
    unsigned n = 0;
    need(rdamd_checkpoint_read_results(ckp, &n), "checkpoint read_results");
    ids.assign(n, 0); llh.assign(n, 0.0); alpha.assign(n, 0.0);
    n_results = n;
    for (unsigned i = 0; i < n; ++i) {
      unsigned np = 0;
      uint64_t nv = 0;
      rdamd_checkpoint_result(ckp, i, &ids[i], &llh[i], &alpha[i], &np, &nv);
    }
    unsigned k = 0;
    for (unsigned i = 1; i < n; ++i)
      if (llh[i] > llh[k]) k = i;   // first maximum, as std::max_element
    if (n) {
      need(rdamd_tree_root_location(tree, (unsigned)ids[k], &best), "root_location");
      best.brlen_ratio = alpha[k];
      best_llh = llh[k];
    }
  }
  if (n_results == 0) die("no candidate root was evaluated");
  rdamd_tree_t *out = rdamd_tree_from_file(o.tree.c_str());
  std::string final_tree;
  if (o.exhaustive) {
    double mx = -INFINITY, total = 0.0;
    for (unsigned i = 0; i < n_results; ++i) mx = std::max(mx, llh[i]);
    for (unsigned i = 0; i < n_results; ++i) total += std::exp(llh[i] - mx);
    for (unsigned i = 0; i < n_results; ++i) {
      rdamd_root_location_t rl;
      need(rdamd_tree_root_location(out, (unsigned)ids[i], &rl), "root_location");
      rl.brlen_ratio = alpha[i];
      rdamd_tree_annotate_branch(out, &rl, "LWR", std::to_string(std::exp(llh[i] - mx) / total).c_str());
      rdamd_tree_annotate_branch(out, &rl, "LLH", std::to_string(llh[i]).c_str());
      rdamd_tree_annotate_branch_lr(out, &rl, "alpha", std::to_string(alpha[i]).c_str(),
                                    std::to_string(1 - alpha[i]).c_str());
    }
  }
  rdamd_root_location_t final_rl;
  need(rdamd_tree_root_location(out, (unsigned)best.id, &final_rl), "root_location");
  final_rl.brlen_ratio = best.brlen_ratio;
  if (o.exhaustive) {   // virtual_rooted_tree(final_rl).newick(): rooted there, then unrooted again
    need(rdamd_tree_root_by(out, &final_rl), "root_by");
    rdamd_tree_unroot(out);
    char *nw = rdamd_tree_newick(out, 1);
    final_tree = nw;
    std::free(nw);
    std::ofstream(o.prefix + ".lwr.tree") << final_tree;
  }
  need(rdamd_tree_root_by(out, &final_rl), "root_by");
  {
    char *nw = rdamd_tree_newick(out, 0);
    std::ofstream(o.prefix + ".rooted.tree") << nw;
    if (!o.exhaustive) final_tree = nw;
    std::free(nw);
  }
  if (!o.silent) std::printf("Final LogLH: %.5f\n", best_llh);
  std::cout << final_tree << std::endl;
  if (!o.silent) {
    const std::chrono::duration<double> took = std::chrono::steady_clock::now() - start;
    std::cout << "Inference took: " << took.count() << "s" << std::endl;
  }
  rdamd_tree_destroy(out);
  rdamd_model_destroy(model);
  if (comm) rdamd_comm_destroy(comm);
  rdamd_tree_destroy(tree);
  if (ckp) rdamd_checkpoint_close(ckp);
  return 0;
}
