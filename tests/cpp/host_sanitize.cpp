// Host-side code under AddressSanitizer + UBSan (CPU build only; GPU sanitizers
// are not available on the pool): the tree mesh, alignment ingest, partition /
// model string parser and checkpoint file, driven through their C++ classes.
// Built and run by tests/test_host_sanitizers.py; exit status 0 = clean.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <unistd.h>

#include "checkpoint.hpp"
#include "model.hpp"
#include "partition_info.hpp"
#include "tree.hpp"

// the one symbol these files take from the HIP side of the library
extern "C" const uint64_t rdamd_map_nt[256] = {};
static uint64_t nt_map[256];

#define CHECK(cond)                                                        \
  do {                                                                     \
    if (!(cond)) {                                                         \
      std::fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, #cond);      \
      return 1;                                                            \
    }                                                                      \
  } while (0)

int main(int argc, char **argv) {
  if (argc < 3) return 2;
  const std::string data = argv[1], tmp = argv[2];
  for (const char *c = "ACGTacgt"; *c; ++c) nt_map[(unsigned char)*c] = 1u << ((c - "ACGTacgt") % 4);
  nt_map[(unsigned char)'-'] = nt_map[(unsigned char)'N'] = 15;
  for (const char *c = "RYSWKMBDHVrywskmbdhv?OXnox"; *c; ++c) nt_map[(unsigned char)*c] = 15;

  using namespace rdamd;
  // ---- trees: every rooting, schedules, annotations, rankings, directional schedule
  for (const char *name : {"10.tree", "101.tree", "single.tree"}) {
    rooted_tree_t t = rooted_tree_t::from_file(data + "/" + name);
    const size_t roots = t.root_count();
    for (size_t i = 0; i < roots; ++i) {
      root_location_t rl = t.root_location(i);
      rl.brlen_ratio = 0.3;
      auto sched = t.generate_operations(rl);
      CHECK(std::get<0>(sched).size() == t.tip_count() - 1);
      auto upd = t.generate_root_update_operations(t.root_location((i * 7 + 3) % roots));
      (void)upd;
      t.annotate_lh(rl, -1.5);
      t.annotate_ratio(rl, 0.3);
      (void)t.branch_length_sanity_check();   // a spread heuristic, not an invariant
    }
    CHECK(!t.newick(true).empty());
    t.unroot();
    CHECK(t.rank_midpoints().size() == roots && t.rank_modified_mad().size() == roots);
    auto d = t.generate_directional_operations();
    CHECK(d.ops.size() == 3 * (t.tip_count() - 2) + roots);
  }
  bool threw = false;
  try { rooted_tree_t::from_newick("((a:1,b:1):1,c:1"); } catch (const std::exception &) { threw = true; }
  CHECK(threw);

  // ---- alignments, partitions
  msa_t fasta = msa_t::from_file(data + "/10.fasta", nt_map, 4, true);
  CHECK(fasta.count() == 10 && fasta.total_weight() == 1000);
  msa_t phy = msa_t::from_file(data + "/101.phy", nt_map, 4, false);
  msa_partitions_t parts{parse_partition_info("DNA+G4, a = 1-100, 500-520"),
                         parse_partition_info("UNREST+R2{0.2/0.8}{0.5/0.5}+FU{.1/.2/.3/.4}+IU{0.25}, b=200-300")};
  CHECK(parse_partition_info("DNA, c = 400, 410-420").parts.size() == 2);   // one-column range before a comma
  auto cut = partition_msa(phy, parts, true);
  CHECK(cut.size() == 2 && cut[0].total_weight() == 121 && cut[1].total_weight() == 101);
  for (const char *bad : {"", "DNA", "DNA,", "DNA, x", "DNA, x =", "DNA, x = 5-", "DNA+G{, x = 1-2",
                          "DNA+IU{, x=1-2", "DNA+ASC_S{1/, x=1-2", "+G, x=1-2"}) {
    threw = false;
    try { parse_partition_info(bad); } catch (const std::exception &) { threw = true; }
    CHECK(threw);
  }

  // ---- checkpoint: write, tear, clean, reread
  const std::string prefix = tmp + "/san";
  unlink((prefix + ".ckp").c_str());
  {
    checkpoint_t c(prefix);
    cli_options_t o;
    o.msa_filename = "m"; o.tree_filename = "t"; o.rate_cats = {ratehet_opts_t(4), ratehet_opts_t(1)};
    c.save_options(o);
    for (size_t i = 0; i < 20; ++i) {
      partition_parameters_t pp;
      pp.subst_rates.assign(12, 0.1 * (double)(i + 1));
      pp.freqs = {.1, .2, .3, .4};
      pp.gamma_alpha = {1.0 + (double)i};
      c.write({i, -100.0 - (double)i, 0.5}, {pp, pp});
    }
    CHECK(c.read_results().size() == 20 && !c.needs_cleaning());
  }
  {
    std::string path = prefix + ".ckp";
    std::ifstream in(path, std::ios::binary);
    std::string bytes((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    for (size_t cut_at : {bytes.size() - 1, bytes.size() - 40, bytes.size() - 333}) {
      std::ofstream(path, std::ios::binary | std::ios::trunc).write(bytes.data(), (std::streamsize)cut_at);
      checkpoint_t c(prefix);
      CHECK(c.needs_cleaning());
      const size_t kept = c.read_results().size();
      CHECK(kept >= 18 && kept < 20);
      c.clean();
      CHECK(!c.needs_cleaning() && c.completed_indicies().size() == kept);
      cli_options_t back;
      c.load_options(back);
      CHECK(back.rate_cats.size() == 2 && back.rate_cats[0].rate_cats == 4);
    }
    // garbage after a valid header never crashes the reader
    std::string junk = bytes.substr(0, 200);
    for (int i = 0; i < 300; ++i) junk.push_back((char)(i * 37));
    std::ofstream(path, std::ios::binary | std::ios::trunc).write(junk.data(), (std::streamsize)junk.size());
    try {
      checkpoint_t c(prefix);
      (void)c.needs_cleaning();
    } catch (const std::exception &) {
    }
  }
  unlink((prefix + ".ckp").c_str());
  std::puts("host code clean");
  return 0;
}
