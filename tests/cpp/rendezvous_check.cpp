// CPU check of root_digger_amd/csrc/tools/rendezvous.hpp (rd_amd's out-of-band
// channel): world allgather/barrier and the host-side site-group sum.
//   rendezvous_check <rank> <world> <group size>      (MASTER_ADDR/MASTER_PORT in the environment)
// prints: "<rank> gather=<r0,r1,...> sum=<v0,v1,v2>"
//   rendezvous_check <rank> <world> watch <victim>
// the "search phase": every rank watches its connections; rank <victim> dies without a word
// after 0.3 s, and every other rank must notice (prints "<rank> lost", exit code 3) instead
// of waiting for ever -- what stands between a dead rank and a hung ncclAllReduce.
//   rendezvous_check <rank> <world> mismatch <odd rank>
// one site group of all ranks; rank <odd rank> arrives with a vector one value longer (its rounds
// have diverged): the leader must REFUSE the reduction, every rank's reducer must fail -- none may
// be left waiting for bytes that never come (prints "<rank> refused", exit code 5).
//   rendezvous_check <rank> <world> abort <who>
// rank <who> never joins the reduction the others wait in; after 0.3 s every waiting rank's OWN
// abort hook is called from a second thread (what a failed lock-step round does): the blocked
// reducer must return failure at once (prints "<rank> aborted", exit code 6).
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
  if (argc == 5 && (std::string(argv[3]) == "mismatch" || std::string(argv[3]) == "abort")) {
    const int rank = std::atoi(argv[1]), world = std::atoi(argv[2]), who = std::atoi(argv[4]);
    const bool mismatch = std::string(argv[3]) == "mismatch";
    try {
      rdamd_tools::rendezvous_t ranks(rank, world);
      rdamd_tools::site_group_t group(ranks, world);
      ranks.barrier();
      double v[4] = {1.0, 2.0, 3.0, 4.0};
      if (rdamd_tools::site_group_t::reducer(v, 3, nullptr, &group) != 1) return 1;   // (a good one first)
      if (mismatch) {
        const int ok = rdamd_tools::site_group_t::reducer(v, rank == who ? 4u : 3u, nullptr, &group);
        if (ok == 1) { std::printf("%d accepted a mismatched reduction\n", rank); return 1; }
        std::printf("%d refused\n", rank);
        return 5;
      }
      if (rank == who) {   // never joins; stays alive so that nobody is helped by a closing socket
        std::this_thread::sleep_for(std::chrono::seconds(3));
        return 0;
      }
      std::thread hook([&group] {
        std::this_thread::sleep_for(std::chrono::milliseconds(300));
        rdamd_tools::site_group_t::abort_hook(&group);
      });
      const int ok = rdamd_tools::site_group_t::reducer(v, 3, nullptr, &group);
      hook.join();
      if (ok == 1) { std::printf("%d: the reduction completed without rank %d\n", rank, who); return 1; }
      std::printf("%d aborted\n", rank);
      return 6;
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
