// The out-of-band channel of a multi-process rd_amd run: a TCP star on
// MASTER_ADDR:MASTER_PORT (the variables every torch-style launcher exports;
// rank 0 listens).  The reference uses MPI_Barrier / MPI_Bcast
// (src/main.cpp:335-409); this program carries no message-passing runtime.
//
//  * barrier(), allgather(): over the world star (set-up traffic only: the
//    checkpoint hand-shake, the RCCL unique ids of the site groups);
//  * watch(): while the ranks search (no rendezvous traffic at all), a thread looks
//    for the END of a peer's connection -- a rank that died, for whatever reason,
//    closes its sockets -- and calls the program's handler, which aborts the RCCL
//    communicator its own rank may be blocked in and exits.  Rank 0 sees every rank
//    and goes down with any of them; the others see rank 0: one lost rank takes the
//    whole run down in two hops instead of leaving G-1 ranks inside ncclAllReduce;
//  * site_group_t: a second star inside one site group whose leader sums host
//    arrays of doubles in rank order and sends the sums back -- the HOST
//    fallback of the lnL all-reduce for runs whose ranks share one device (RCCL
//    refuses two ranks on one GPU); one node only (members reach the leader on
//    the loopback interface).  Real multi-GPU runs reduce with RCCL on the device
//    (rdamd_comm_*), this path is what the one-GPU test box exercises.
#pragma once
#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <cstdio>
#include <chrono>
#include <functional>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace rdamd_tools {

inline void send_all(int fd, const void *buf, size_t n) {
  const char *p = (const char *)buf;
  while (n) {
    const ssize_t k = send(fd, p, n, MSG_NOSIGNAL);
    if (k <= 0) throw std::runtime_error("rendezvous: a rank went away (send)");
    p += k;
    n -= (size_t)k;
  }
}
inline void recv_all(int fd, void *buf, size_t n) {
  if (n && recv(fd, buf, n, MSG_WAITALL) != (ssize_t)n)
    throw std::runtime_error("rendezvous: a rank went away (recv)");
}

// leader side: `members` connections, identified by the rank each peer announces
inline std::vector<int> accept_ranked(int listen_fd, int members, int first_rank) {
  std::vector<int> peers((size_t)members, -1);
  for (int i = 0; i < members; ++i) {
    const int fd = accept(listen_fd, nullptr, nullptr);
    if (fd < 0) throw std::runtime_error("rendezvous: accept failed");
    const int one = 1;
    setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
    int32_t r = -1;
    recv_all(fd, &r, sizeof r);
    const int slot = r - first_rank;
    if (slot < 0 || slot >= members || peers[(size_t)slot] >= 0)
      throw std::runtime_error("rendezvous: unexpected rank " + std::to_string(r));
    peers[(size_t)slot] = fd;
  }
  return peers;
}

inline int connect_retry(const char *addr, int port, int my_rank, double timeout_s) {
  sockaddr_in sa;
  std::memset(&sa, 0, sizeof sa);
  sa.sin_family = AF_INET;
  sa.sin_port = htons((uint16_t)port);
  if (inet_pton(AF_INET, addr, &sa.sin_addr) != 1)
    throw std::runtime_error(std::string("not an IPv4 address: ") + addr);
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {   // the leader may not be listening yet
    const int fd = socket(AF_INET, SOCK_STREAM, 0);
    if (fd >= 0 && connect(fd, (sockaddr *)&sa, sizeof sa) == 0) {
      const int one = 1;
      setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
      const int32_t r = my_rank;
      send_all(fd, &r, sizeof r);
      return fd;
    }
    if (fd >= 0) close(fd);
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s)
      throw std::runtime_error("rank " + std::to_string(my_rank) + " could not reach " + addr + ":" +
                               std::to_string(port));
    std::this_thread::sleep_for(std::chrono::milliseconds(20));
  }
}

inline int listen_on(int port, int backlog, int *bound_port = nullptr) {
  sockaddr_in sa;
  std::memset(&sa, 0, sizeof sa);
  sa.sin_family = AF_INET;
  sa.sin_port = htons((uint16_t)port);
  sa.sin_addr.s_addr = htonl(INADDR_ANY);
  const int ls = socket(AF_INET, SOCK_STREAM, 0);
  const int one = 1;
  setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
  if (ls < 0 || bind(ls, (sockaddr *)&sa, sizeof sa) != 0 || listen(ls, backlog) != 0)
    throw std::runtime_error("cannot listen on port " + std::to_string(port));
  if (bound_port) {
    socklen_t len = sizeof sa;
    getsockname(ls, (sockaddr *)&sa, &len);
    *bound_port = ntohs(sa.sin_port);
  }
  return ls;
}

class rendezvous_t {
public:
  rendezvous_t(int rank, int world) : _rank(rank), _world(world) {
    if (world <= 1) return;
    const char *addr = std::getenv("MASTER_ADDR");
    const char *port_s = std::getenv("MASTER_PORT");
    const int port = port_s ? std::atoi(port_s) : 29400;
    if (rank == 0) {
      const int ls = listen_on(port, world);
      _peers = accept_ranked(ls, world - 1, 1);
      close(ls);
    } else {
      _peers.push_back(connect_retry(addr ? addr : "127.0.0.1", port, rank, 600.0));
    }
  }
  ~rendezvous_t() {
    unwatch();
    for (int fd : _peers) close(fd);
  }
  // From now until unwatch(): `on_loss` runs (once, on the watcher thread) when a peer's
  // connection ends.  Bytes a peer sends meanwhile -- it has finished and waits in the
  // next barrier -- are left alone and end the watch on that connection.
  void watch(std::function<void()> on_loss) {
    if (_world <= 1 || _watcher.joinable()) return;
    _stop = false;
    _watcher = std::thread([this, on_loss] {
      std::vector<pollfd> fds;
      for (int fd : _peers) fds.push_back(pollfd{fd, POLLIN, 0});
      while (!_stop.load()) {
        if (poll(fds.data(), fds.size(), 100) <= 0) continue;
        for (pollfd &f : fds) {
          if (f.fd < 0 || !(f.revents & (POLLIN | POLLHUP | POLLERR))) continue;
          char byte;
          const ssize_t k = recv(f.fd, &byte, 1, MSG_PEEK | MSG_DONTWAIT);
          if (k > 0) { f.fd = -1; continue; }          // barrier traffic of a rank that is done
          if (k < 0 && (errno == EAGAIN || errno == EWOULDBLOCK)) continue;
          if (_stop.load()) return;
          on_loss();                                     // EOF or a broken connection
          return;
        }
      }
    });
  }
  void unwatch() {
    _stop = true;
    if (_watcher.joinable()) _watcher.join();
  }
  rendezvous_t(const rendezvous_t &) = delete;
  rendezvous_t &operator=(const rendezvous_t &) = delete;
  int rank() const { return _rank; }
  int world() const { return _world; }

  void barrier() {
    unwatch();   // (from here on the connections carry data again)
    char byte = 1;
    std::vector<char> all;
    allgather(&byte, 1, all);
  }
  // all = the `bytes`-sized contributions of ranks 0 .. world-1, in rank order
  void allgather(const void *mine, size_t bytes, std::vector<char> &all) {
    all.assign(bytes * (size_t)_world, 0);
    if (_world <= 1) {
      std::memcpy(all.data(), mine, bytes);
      return;
    }
    if (_rank == 0) {
      std::memcpy(all.data(), mine, bytes);
      // whichever rank is ready first is read first, so that the end of ANY connection is
      // seen at once (a rank that died while others are still searching)
      std::vector<pollfd> fds;
      for (int fd : _peers) fds.push_back(pollfd{fd, POLLIN, 0});
      for (int pending = _world - 1; pending > 0;) {
        if (poll(fds.data(), fds.size(), -1) < 0) {
          if (errno == EINTR) continue;
          throw std::runtime_error("rendezvous: poll failed");
        }
        for (size_t i = 0; i < fds.size(); ++i) {
          if (fds[i].fd < 0 || !(fds[i].revents & (POLLIN | POLLHUP | POLLERR))) continue;
          recv_all(fds[i].fd, all.data() + bytes * (i + 1), bytes);   // throws at the end of a connection
          fds[i].fd = -1;
          --pending;
        }
      }
      for (int fd : _peers) send_all(fd, all.data(), all.size());
    } else {
      send_all(_peers[0], mine, bytes);
      recv_all(_peers[0], all.data(), all.size());
    }
  }

private:
  int _rank, _world;
  std::vector<int> _peers;   // rank 0: peer of rank r at [r - 1]; others: rank 0
  std::thread _watcher;
  std::atomic<bool> _stop{false};
};

// Site group = ranks [leader, leader + size) of the world; host-side sum of
// double arrays through the group leader, in rank order (so every run with the
// same grouping produces the same bits, and all members receive identical sums).
class site_group_t {
public:
  site_group_t(rendezvous_t &world, int group_size)
      : _size(group_size), _srank(world.rank() % group_size) {
    if (group_size <= 1) return;
    const int leader = world.rank() - _srank;
    int32_t port = 0;
    int ls = -1;
    if (_srank == 0) {
      int bound = 0;
      ls = listen_on(0, group_size, &bound);   // ephemeral port, published below
      port = bound;
    }
    std::vector<char> all;
    world.allgather(&port, sizeof port, all);
    if (_srank == 0) {
      _peers = accept_ranked(ls, group_size - 1, world.rank() + 1);
      close(ls);
    } else {
      int32_t leader_port = 0;
      std::memcpy(&leader_port, all.data() + sizeof(int32_t) * (size_t)leader, sizeof leader_port);
      _peers.push_back(connect_retry("127.0.0.1", leader_port, world.rank(), 600.0));
    }
  }
  ~site_group_t() {
    for (int fd : _peers) close(fd);
  }
  site_group_t(const site_group_t &) = delete;
  site_group_t &operator=(const site_group_t &) = delete;
  int size() const { return _size; }
  int rank() const { return _srank; }

  void allreduce_sum(double *values, unsigned n) {
    if (_size <= 1 || n == 0) return;
    if (_aborted.load()) throw std::runtime_error("site group: aborted");
    const size_t bytes = sizeof(double) * n;
    const uint32_t len = n;
    if (_srank == 0) {
      _tmp.resize(n);
      for (int fd : _peers) {   // rank order: ((v0 + v1) + v2) + ...
        uint32_t theirs = 0;
        recv_all(fd, &theirs, sizeof theirs);
        // (ranks whose rounds have diverged: a device collective would hang or corrupt here)
        if (theirs != len) throw std::runtime_error("site group: the ranks' vectors differ in length");
        recv_all(fd, _tmp.data(), bytes);
        for (unsigned i = 0; i < n; ++i) values[i] += _tmp[i];
      }
      for (int fd : _peers) send_all(fd, values, bytes);
    } else {
      send_all(_peers[0], &len, sizeof len);
      send_all(_peers[0], values, bytes);
      recv_all(_peers[0], values, bytes);
    }
    // FAULT INJECTION for the divergence-guard test (tests/test_gpu_lockstep_rounds.py): this
    // rank's copy of the sums of its `_fault_call`-th reduction differs from the others' by one
    // unit in the last place of one value -- what a reducer without a bit-identity promise may do
    if (_fault_call && ++_calls == _fault_call) {
      uint64_t b;
      std::memcpy(&b, &values[0], 8);
      b ^= 1;
      std::memcpy(&values[0], &b, 8);
    }
  }
  void set_fault(uint64_t call) { _fault_call = call; }
  // rdamd_lnl_abort_t: every pending and future reduction of this process fails (any thread)
  void abort() {
    _aborted = true;
    for (int fd : _peers) shutdown(fd, SHUT_RDWR);
  }
  static void abort_hook(void *user) { ((site_group_t *)user)->abort(); }
  // rdamd_lnl_reducer_t with on_device = 0 and user = the site_group_t
  static int reducer(double *values, unsigned int n, void *, void *user) {
    try {
      ((site_group_t *)user)->allreduce_sum(values, n);
      return 1;
    } catch (const std::exception &e) {
      std::fprintf(stderr, "%s\n", e.what());
      return 0;
    }
  }

private:
  int _size, _srank;
  std::vector<int> _peers;
  std::vector<double> _tmp;
  std::atomic<bool> _aborted{false};
  uint64_t _fault_call = 0, _calls = 0;
};

}  // namespace rdamd_tools
