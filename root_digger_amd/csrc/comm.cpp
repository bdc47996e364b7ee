// RCCL communicator of one site group (include/root_digger_amd.h, "site-sharded
// runs").  The per-block log-likelihoods of a batch are summed over the ranks of
// the group with one collective queued on the partition's stream (north star: "an
// RCCL all-reduce of per-block log-likelihoods over xGMI").  The message is 8 x jobs
// bytes, i.e. latency-bound: there is nothing to bucket or overlap, the lever is the
// batch size of the launch in front of it.
//
// WHICH collective (rdamd_comm_set_sum_mode).  Every rank of a site group must end
// up with the same BITS: the optimisers above branch on these sums, and one ulp of
// difference between two ranks forks a trajectory -- the next rounds then have
// different lengths on different ranks and the group sits in a collective until the
// time limit.  ncclAllReduce promises a sum, not which sum: ring and tree algorithms
// hand every rank a copy of one result, direct / one-shot small-message paths let
// each rank add its peers' buffers itself, in its own order -- and a kilobyte
// message is where a library picks those.  So the DEFAULT is RDAMD_COMM_SUM_GATHER:
// ncclAllGather of the G vectors (bytes move, nothing is added) and
// `rank_order_sum_kernel` on the same stream, which adds them as
// ((v0 + v1) + v2) + ... in rank order.  Identical on every rank by construction,
// whatever algorithm RCCL picks, and the same sum as the host reducers of the tests
// (rendezvous.hpp site_group_t, dist.py): "rounds == sequential sharded search, bit
// for bit" holds on real links too.  RDAMD_COMM_SUM_ALLREDUCE (or
// RDAMD_COMM_SUM=allreduce in the environment) keeps the one-call ncclAllReduce;
// bench.py times both.  The conductor's divergence guard (lockstep_conductor.hpp)
// catches what either mode may still get wrong.
//
// Failure handling: a collective whose peer never arrives would block for ever, so
// rdamd_comm_reducer WAITS for its all-reduce by polling the stream -- with
// ncclCommGetAsyncError, a time limit (rdamd_comm_set_timeout, default 600 s, or
// RDAMD_COMM_TIMEOUT seconds) and an abort flag another thread may raise
// (rdamd_comm_abort: rd_amd's rendezvous watcher, when a rank of the run has gone).  On
// any of them the communicator is aborted (ncclCommAbort) and the call fails; the
// caller must exit and start afresh -- a communicator is not reusable after that.
//
// librccl is opened on first use (dlopen) so that single-GPU users carry no
// dependency on it; only the entry points used here are resolved.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.hpp"

namespace {
struct rccl_api {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                            hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;   // optional
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;                          // optional
  ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t *) = nullptr;   // optional
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::string load_error;   // dlerror() text of the failed load (dlerror() itself answers once)
};

rccl_api *rccl() {
  static rccl_api api;
  static std::once_flag once;
  std::call_once(once, [] {
    // RCCL must sit on the SAME HIP/HSA runtime instance this library runs on.  A
    // process may hold two ROCm installations (e.g. PyTorch's bundled one next to
    // /opt/rocm): look next to the libamdhip64 that is actually serving us first.
    std::string beside;
    Dl_info info;
    if (dladdr(reinterpret_cast<void *>(&hipGetDevice), &info) && info.dli_fname) {
      beside = info.dli_fname;
      const size_t slash = beside.rfind('/');
      beside = slash == std::string::npos ? std::string() : beside.substr(0, slash + 1);
    }
    // absolute candidates (beside the runtime, /opt/rocm) are tried only where the file
    // exists; the bare sonames go through the loader's search path
    const std::string candidates[] = {beside.empty() ? std::string() : beside + "librccl.so.1",
                                      beside.empty() ? std::string() : beside + "librccl.so",
                                      "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const std::string &name : candidates) {
      if (name.empty()) continue;
      if (name[0] == '/' && access(name.c_str(), R_OK) != 0) continue;
      api.lib = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (api.lib) break;
      const char *why = dlerror();
      if (why) api.load_error = why;
    }
    if (!api.lib) {
      if (api.load_error.empty()) api.load_error = "no librccl found";
      return;
    }
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
    api.AllReduce = (decltype(api.AllReduce))dlsym(api.lib, "ncclAllReduce");
    api.AllGather = (decltype(api.AllGather))dlsym(api.lib, "ncclAllGather");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
    api.CommAbort = (decltype(api.CommAbort))dlsym(api.lib, "ncclCommAbort");
    api.CommGetAsyncError = (decltype(api.CommGetAsyncError))dlsym(api.lib, "ncclCommGetAsyncError");
    if (!api.GetUniqueId || !api.CommInitRank || !api.AllReduce || !api.CommDestroy) {
      dlclose(api.lib);
      api.lib = nullptr;
      api.load_error = "the library lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy";
    }
  });
  if (!api.lib) {
    rdamd::set_error(60, "RCCL (librccl.so.1) could not be loaded: %s", api.load_error.c_str());
    return nullptr;
  }
  return &api;
}

bool ok(rccl_api *a, ncclResult_t r, const char *what) {
  if (r == ncclSuccess) return true;
  rdamd::set_error(61, "%s: %s", what, a->GetErrorString ? a->GetErrorString(r) : "RCCL error");
  return false;
}
}  // namespace

struct rdamd_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, n_ranks = 1;
  double timeout_s = 600.0;
  std::atomic<bool> abort_requested{false};
  std::atomic<bool> dead{false};   // aborted: no further collective may be queued
  std::atomic<int> sum_mode{RDAMD_COMM_SUM_GATHER};
  // RDAMD_COMM_SUM_GATHER: where the G vectors land, one buffer per stream the
  // communicator has been used on (work on ONE stream is ordered, so a buffer is free
  // again when the next collective of that stream starts; two streams must not share)
  struct gather_buf { hipStream_t stream = nullptr; double *d = nullptr; size_t cap = 0; };
  std::mutex gather_mu;
  std::vector<gather_buf> gather;
};

namespace {
// ncclCommAbort once; afterwards the communicator only remembers that it is dead
void abort_comm(rccl_api *a, rdamd_comm *c) {
  if (c->dead.exchange(true)) return;
  if (a->CommAbort && c->comm) (void)a->CommAbort(c->comm);
  c->comm = nullptr;
}

// Wait for everything queued on `stream` (the all-reduce last) without blocking in the
// runtime: see the header comment.
// (`event` not null: wait for that event instead of for the whole stream -- work queued on the
// stream BEHIND the collective, the next round of a lock-stepped search, is not waited for)
bool wait_for_collective(rccl_api *a, rdamd_comm *c, hipStream_t stream, hipEvent_t event = nullptr) {
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spin = 0;; ++spin) {
    const hipError_t q = event ? hipEventQuery(event) : hipStreamQuery(stream);
    if (q == hipSuccess) return true;
    if (q != hipErrorNotReady) {
      rdamd::set_error(63, "site-group all-reduce: %s", hipGetErrorString(q));
      abort_comm(a, c);
      return false;
    }
    if (c->abort_requested.load()) {
      rdamd::set_error(64, "site-group all-reduce aborted: a rank of the run has gone");
      abort_comm(a, c);
      return false;
    }
    if (spin < 2000) continue;   // the usual case: the sum arrives within microseconds
    ncclResult_t async = ncclSuccess;
    if (a->CommGetAsyncError && c->comm && a->CommGetAsyncError(c->comm, &async) == ncclSuccess &&
        async != ncclSuccess && async != ncclInProgress) {
      rdamd::set_error(65, "site-group all-reduce: %s",
                       a->GetErrorString ? a->GetErrorString(async) : "asynchronous RCCL error");
      abort_comm(a, c);
      return false;
    }
    const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (waited > c->timeout_s) {
      rdamd::set_error(66, "site-group all-reduce: no answer from the other %d rank(s) within %.0f s",
                       c->n_ranks - 1, c->timeout_s);
      abort_comm(a, c);
      return false;
    }
    std::this_thread::sleep_for(std::chrono::microseconds(waited < 0.01 ? 5 : 200));
  }
}
}  // namespace

namespace {
// out[i] = ((g[0][i] + g[1][i]) + g[2][i]) + ... : the site group's sum in RANK ORDER, the
// same additions in the same order on every rank (and in the host reducers of the tests)
__global__ void rank_order_sum_kernel(const double *gathered, double *out,
                                      unsigned n, unsigned ranks) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double acc = gathered[i];
  for (unsigned r = 1; r < ranks; ++r) acc += gathered[(size_t)r * n + i];
  out[i] = acc;
}

// the stream's gather buffer, at least ranks x n doubles
double *gather_buffer(rdamd_comm *c, hipStream_t stream, size_t n) {
  std::lock_guard<std::mutex> g(c->gather_mu);
  rdamd_comm::gather_buf *b = nullptr;
  for (auto &e : c->gather)
    if (e.stream == stream) b = &e;
  if (!b) {
    c->gather.push_back({});
    b = &c->gather.back();
    b->stream = stream;
  }
  const size_t want = n * (size_t)c->n_ranks;
  if (want > b->cap) {
    // (a collective queued earlier may still read the old buffer)
    if (b->d && (hipStreamSynchronize(stream) != hipSuccess || hipFree(b->d) != hipSuccess)) return nullptr;
    b->d = nullptr;
    b->cap = std::max<size_t>(2 * want, 4096);
    if (hipMalloc((void **)&b->d, b->cap * sizeof(double)) != hipSuccess) {
      b->cap = 0;
      return nullptr;
    }
  }
  return b->d;
}
}  // namespace

static_assert(sizeof(ncclUniqueId) == 128, "rdamd_comm_unique_id hands out 128 bytes");

extern "C" {

int rdamd_comm_unique_id(char id[128]) {
  rdamd::clear_error();
  rccl_api *a = rccl();
  if (!a) return RDAMD_FAILURE;
  ncclUniqueId u;
  if (!ok(a, a->GetUniqueId(&u), "ncclGetUniqueId")) return RDAMD_FAILURE;
  std::memcpy(id, &u, sizeof u);
  return RDAMD_SUCCESS;
}

rdamd_comm_t *rdamd_comm_create(const char id[128], int rank, int n_ranks) {
  rdamd::clear_error();
  rccl_api *a = rccl();
  if (!a) return nullptr;
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks) {
    rdamd::set_error(62, "rdamd_comm_create: rank %d of %d", rank, n_ranks);
    return nullptr;
  }
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof u);
  auto *c = new rdamd_comm();
  c->rank = rank;
  c->n_ranks = n_ranks;
  if (const char *t = std::getenv("RDAMD_COMM_TIMEOUT"))
    if (std::atof(t) > 0.0) c->timeout_s = std::atof(t);
  if (const char *t = std::getenv("RDAMD_COMM_SUM")) {
    if (!std::strcmp(t, "allreduce")) c->sum_mode = RDAMD_COMM_SUM_ALLREDUCE;
    else if (std::strcmp(t, "gather")) {
      rdamd::set_error(62, "RDAMD_COMM_SUM takes gather or allreduce, not '%s'", t);
      delete c;
      return nullptr;
    }
  }
  if (!ok(a, a->CommInitRank(&c->comm, n_ranks, u, rank), "ncclCommInitRank")) {
    delete c;
    return nullptr;
  }
  return c;
}

int rdamd_comm_allreduce_sum(rdamd_comm_t *c, double *device_values, unsigned int n, void *stream) {
  rccl_api *a = rccl();
  if (!a || !c) return RDAMD_FAILURE;
  if (n == 0) return RDAMD_SUCCESS;
  if (c->dead.load() || c->abort_requested.load()) {
    rdamd::set_error(64, "site-group all-reduce: the communicator was aborted");
    return RDAMD_FAILURE;
  }
  if (c->sum_mode.load() == RDAMD_COMM_SUM_ALLREDUCE)
    return ok(a, a->AllReduce(device_values, device_values, n, ncclDouble, ncclSum, c->comm,
                              (hipStream_t)stream), "ncclAllReduce")
               ? RDAMD_SUCCESS : RDAMD_FAILURE;
  // the deterministic form (header comment): gather, then add in rank order
  if (!a->AllGather) {
    rdamd::set_error(60, "this librccl has no ncclAllGather (RDAMD_COMM_SUM=allreduce selects the one-call form)");
    return RDAMD_FAILURE;
  }
  double *g = gather_buffer(c, (hipStream_t)stream, n);
  if (!g) {
    rdamd::set_error(63, "site-group sum: no device memory for %d x %u values", c->n_ranks, n);
    return RDAMD_FAILURE;
  }
  if (!ok(a, a->AllGather(device_values, g, n, ncclDouble, c->comm, (hipStream_t)stream), "ncclAllGather"))
    return RDAMD_FAILURE;
  rank_order_sum_kernel<<<dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(g, device_values, n,
                                                                                     (unsigned)c->n_ranks);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    rdamd::set_error(63, "site-group sum: %s", hipGetErrorString(e));
    return RDAMD_FAILURE;
  }
  return RDAMD_SUCCESS;
}

int rdamd_comm_set_sum_mode(rdamd_comm_t *c, int mode) {
  if (!c || (mode != RDAMD_COMM_SUM_GATHER && mode != RDAMD_COMM_SUM_ALLREDUCE)) {
    rdamd::set_error(62, "rdamd_comm_set_sum_mode: RDAMD_COMM_SUM_GATHER or RDAMD_COMM_SUM_ALLREDUCE");
    return RDAMD_FAILURE;
  }
  c->sum_mode = mode;
  return RDAMD_SUCCESS;
}
int rdamd_comm_sum_mode(const rdamd_comm_t *c) { return c ? c->sum_mode.load() : -1; }

// The rank-order sum by itself: out[i] = ((g[0][i] + g[1][i]) + ...) over `ranks` vectors of n
// doubles that lie one behind the other in device memory -- what RDAMD_COMM_SUM_GATHER queues
// behind its ncclAllGather, callable without a communicator (the tests compare it with the host
// reducers bit for bit; `out` may be the first vector).
int rdamd_rank_order_sum(const double *gathered, double *out, unsigned int n, unsigned int ranks, void *stream) {
  rdamd::clear_error();
  if (!gathered || !out || ranks < 1) {
    rdamd::set_error(62, "rdamd_rank_order_sum: null pointer or no ranks");
    return RDAMD_FAILURE;
  }
  if (n == 0) return RDAMD_SUCCESS;
  rank_order_sum_kernel<<<dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(gathered, out, n, ranks);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    rdamd::set_error(63, "rdamd_rank_order_sum: %s", hipGetErrorString(e));
    return RDAMD_FAILURE;
  }
  return RDAMD_SUCCESS;
}

// the reducer model_t calls: queue the sum, then wait for it (a lost peer must not hang us)
int rdamd_comm_reducer(double *values, unsigned int n, void *stream, void *user) {
  rdamd_comm_t *c = (rdamd_comm_t *)user;
  if (rdamd_comm_allreduce_sum(c, values, n, stream) != RDAMD_SUCCESS) return RDAMD_FAILURE;
  rccl_api *a = rccl();
  if (c->n_ranks <= 1 || !a) return RDAMD_SUCCESS;
  return wait_for_collective(a, c, (hipStream_t)stream) ? RDAMD_SUCCESS : RDAMD_FAILURE;
}

// the two halves on their own (rdamd_model_set_lnl_reducer_async): queue only ...
int rdamd_comm_reducer_queue(double *values, unsigned int n, void *stream, void *user) {
  return rdamd_comm_allreduce_sum((rdamd_comm_t *)user, values, n, stream);
}
// ... and wait for the HIP event recorded behind it, the way rdamd_comm_reducer waits
int rdamd_comm_reducer_wait(void *event, void *user) {
  rdamd_comm_t *c = (rdamd_comm_t *)user;
  rccl_api *a = rccl();
  if (!a || !c || !event) return RDAMD_FAILURE;
  if (c->n_ranks <= 1) {
    if (hipEventSynchronize((hipEvent_t)event) == hipSuccess) return RDAMD_SUCCESS;
    rdamd::set_error(63, "site-group all-reduce: the stream failed");
    return RDAMD_FAILURE;
  }
  return wait_for_collective(a, c, nullptr, (hipEvent_t)event) ? RDAMD_SUCCESS : RDAMD_FAILURE;
}

void rdamd_comm_set_timeout(rdamd_comm_t *c, double seconds) {
  if (c && seconds > 0.0) c->timeout_s = seconds;
}

// callable from any thread: the rank's next (or current) rdamd_comm_reducer call fails
void rdamd_comm_abort(rdamd_comm_t *c) {
  if (c) c->abort_requested = true;
}

void rdamd_comm_destroy(rdamd_comm_t *c) {
  if (!c) return;
  rccl_api *a = rccl();
  if (a && c->comm && !c->dead.load()) (void)a->CommDestroy(c->comm);
  for (auto &b : c->gather)
    if (b.d) (void)hipFree(b.d);
  delete c;
}

}  // extern "C"
