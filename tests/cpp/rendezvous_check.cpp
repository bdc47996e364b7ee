// CPU check of root_digger_amd/csrc/tools/rendezvous.hpp (rd_amd's out-of-band
// channel): world allgather/barrier and the host-side site-group sum.
//   rendezvous_check <rank> <world> <group size>      (MASTER_ADDR/MASTER_PORT in the environment)
// prints: "<rank> gather=<r0,r1,...> sum=<v0,v1,v2>"
//   rendezvous_check <rank> <world> watch <victim>
// the "search phase": every rank watches its connections; rank <victim> dies without a word
// after 0.3 s, and every other rank must notice (prints "<rank> lost", exit code 3) instead
// of waiting for ever -- what stands between a dead rank and a hung ncclAllReduce.
#include <cstdio>
#include <cstdlib>
#include <string>

#include "rendezvous.hpp"

int main(int argc, char **argv) {
  if (argc == 5 && std::string(argv[3]) == "watch") {
    const int rank = std::atoi(argv[1]), world = std::atoi(argv[2]), victim = std::atoi(argv[4]);
    try {
      rdamd_tools::rendezvous_t ranks(rank, world);
      ranks.barrier();
      ranks.watch([rank] {
        std::printf("%d lost\n", rank);
        std::fflush(stdout);
        std::_Exit(3);
      });
      if (rank == victim) {
        std::this_thread::sleep_for(std::chrono::milliseconds(300));
        std::_Exit(9);                                       // no goodbye: the sockets just close
      }
      std::this_thread::sleep_for(std::chrono::seconds(30));   // "searching"
      std::printf("%d never noticed\n", rank);
      return 4;
    } catch (const std::exception &e) {
      std::fprintf(stderr, "rank %d: %s\n", rank, e.what());
      return 1;
    }
  }
  if (argc != 4) return 2;
  const int rank = std::atoi(argv[1]), world = std::atoi(argv[2]), G = std::atoi(argv[3]);
  try {
    rdamd_tools::rendezvous_t ranks(rank, world);
    rdamd_tools::site_group_t group(ranks, G);
    ranks.barrier();
    int32_t mine = 100 + rank;
    std::vector<char> all;
    ranks.allgather(&mine, sizeof mine, all);
    std::printf("%d gather=", rank);
    for (int r = 0; r < world; ++r) {
      int32_t v;
      std::memcpy(&v, all.data() + sizeof v * (size_t)r, sizeof v);
      std::printf("%d%s", v, r + 1 < world ? "," : "");
    }
    double vals[3] = {1.0 + rank, 0.5 * rank, 1e-3};
    for (int rep = 0; rep < 50; ++rep) {   // many small messages, like a search
      double v[3] = {vals[0], vals[1], vals[2]};
      if (rdamd_tools::site_group_t::reducer(v, 3, nullptr, &group) != 1) return 3;
      if (rep == 49) std::printf(" sum=%.17g,%.17g,%.17g\n", v[0], v[1], v[2]);
    }
    ranks.barrier();
  } catch (const std::exception &e) {
    std::fprintf(stderr, "rank %d: %s\n", rank, e.what());
    return 1;
  }
  return 0;
}
