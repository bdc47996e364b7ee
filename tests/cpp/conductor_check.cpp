// CPU check of root_digger_amd/csrc/lockstep_conductor.hpp: the PROTOCOL of lock step in
// deterministic rounds -- which requests form a round, in which order the rounds of two worker
// groups reach the site group's reducer, how candidates are handed out -- without a GPU.
//
// G simulated ranks of one site group live in ONE process, each with its own conductor and W
// worker threads; the library calls the conductor makes (objective batch, combined root step) and
// the HIP runtime calls around them are stand-ins defined in this file (they compute
// deterministic values per "rank" on the host).  The site group's reducer is a rendezvous of
// the G ranks that REFUSES a mismatch: every rank must arrive with a vector of the same length,
// round after round -- exactly what a real all-reduce needs and what thread timing must not
// change.  Every worker runs a synthetic candidate: a chain of requests whose kind and size
// depend on the SUMS it got back (as an optimiser's next step depends on the reduced lnL), for a
// number of steps that differs per candidate, under random delays.
//
//   conductor_check <ranks G> <workers W> <groups 1|2> <candidates> <seed> [host|device] [<rank>:<collective>]
// prints "conductor OK rounds=<n> collectives=<n> redos=<n> digest=<hex>"; exit code 0.  The digest
// covers every value every worker received: the test driver compares it across seeds of the DELAYS.
//
// `device`: the path real RCCL runs take -- rdamd_evaluate_batch_submit_device / _redo_device /
// _finish_device stand-ins whose SECOND-PASS FLAG is raised at random, differently on every rank
// (so the flag must travel through the sum and every rank must repeat the collective, in its
// worker group's turn), and a reducer in two halves (queue / wait).  The rendezvous then also
// asserts that every rank's vector carries the same guard word for (worker group, round, requests)
// -- i.e. that all ranks issue the SAME SEQUENCE of collectives, redos included.
// `<rank>:<collective>`: that rank's copy of the sums of its <collective>-th reduction (and the
// two after it) differs by one unit in the last place (a reducer without a bit-identity promise).  The conductor's
// divergence guard must fail the run on every rank at once, naming the round: exit code 1, "has
// diverged" on stderr.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <random>
#include <thread>
#include <vector>

#include "lockstep_conductor.hpp"

// ---- stand-ins for the HIP runtime (host path of the conductor only) ---------------------------
extern "C" {
hipError_t hipMalloc(void **p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned int) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void *p) { std::free(p); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = (hipEvent_t)std::malloc(8); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { std::free((void *)e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
const char *hipGetErrorString(hipError_t) { return "mock"; }
}

// ---- stand-ins for the library: a "partition" is the rank's number ---------------------------------
struct rdamd_partition { int rank; };
struct rdamd_schedule { double tag; };
static thread_local const char *g_err = "";
extern "C" {
const char *rdamd_errmsg(void) { return g_err; }
void *rdamd_partition_stream(const rdamd_partition_t *) { return nullptr; }
unsigned int rdamd_partition_rate_cats(const rdamd_partition_t *) { return 1; }
unsigned int rdamd_partition_states(const rdamd_partition_t *) { return 2; }   // NP = 2 parameters per job
}
static std::mutex g_slot_mu;
static std::vector<double> g_slot[64][2];   // [rank][slot]: results of the batch in flight
static double job_value(int rank, double tag, const double *subst) {
  // a rank's share of a job's "lnL": depends on the rank (its site block), the schedule, the parameters
  return -(1.0 + 0.25 * rank) * (tag + subst[0] * 3.0 + subst[1]);
}
extern "C" {
int rdamd_evaluate_batch_submit(rdamd_partition_t *p, unsigned int slot, unsigned int n, const rdamd_schedule_t *const *scheds,
                                const double *subst, const double *, const double *, const double *) {
  std::vector<double> out(n);
  for (unsigned j = 0; j < n; ++j) out[j] = job_value(p->rank, scheds[j]->tag, subst + 2 * j);
  std::lock_guard<std::mutex> g(g_slot_mu);
  if (!g_slot[p->rank][slot].empty()) { g_err = "slot busy"; return RDAMD_FAILURE; }
  g_slot[p->rank][slot] = out;
  return RDAMD_SUCCESS;
}
int rdamd_evaluate_batch_wait(rdamd_partition_t *p, unsigned int slot, double *out) {
  std::lock_guard<std::mutex> g(g_slot_mu);
  std::vector<double> &v = g_slot[p->rank][slot];
  std::copy(v.begin(), v.end(), out);
  v.clear();
  return RDAMD_SUCCESS;
}
// The device path: "device memory" is host memory here and the stream runs at once.  A batch's
// second-pass flag is up with probability 1/6, decided per (rank, batch) -- the ranks disagree,
// as real shards do; until its redo a flagged batch shows values that are NOT final.
static std::atomic<uint64_t> g_batches[64];
int rdamd_evaluate_batch_submit_device(rdamd_partition_t *p, unsigned int slot, unsigned int n, const rdamd_schedule_t *const *scheds,
                                       const double *subst, const double *, const double *, const double *, void *d_vec) {
  std::vector<double> out(n);
  for (unsigned j = 0; j < n; ++j) out[j] = job_value(p->rank, scheds[j]->tag, subst + 2 * j);
  const uint64_t k = g_batches[p->rank]++;
  const bool flagged = ((k * 2654435761ull + (uint64_t)p->rank * 40503ull) >> 7) % 6 == 0;
  double *d = (double *)d_vec;
  for (unsigned j = 0; j < n; ++j) d[j] = flagged ? out[j] - 1000.0 : out[j];
  d[n] = flagged ? 1.0 : 0.0;
  std::lock_guard<std::mutex> g(g_slot_mu);
  if (!g_slot[p->rank][slot].empty()) { g_err = "slot busy"; return RDAMD_FAILURE; }
  g_slot[p->rank][slot] = out;
  return RDAMD_SUCCESS;
}
int rdamd_evaluate_batch_redo_device(rdamd_partition_t *p, unsigned int slot, void *d_vec) {
  std::lock_guard<std::mutex> g(g_slot_mu);
  const std::vector<double> &v = g_slot[p->rank][slot];
  if (v.empty()) { g_err = "nothing in flight"; return RDAMD_FAILURE; }
  std::copy(v.begin(), v.end(), (double *)d_vec);
  ((double *)d_vec)[v.size()] = 0.0;
  return RDAMD_SUCCESS;
}
int rdamd_evaluate_batch_finish_device(rdamd_partition_t *p, unsigned int slot) {
  std::lock_guard<std::mutex> g(g_slot_mu);
  if (g_slot[p->rank][slot].empty()) { g_err = "nothing in flight"; return RDAMD_FAILURE; }
  g_slot[p->rank][slot].clear();
  return RDAMD_SUCCESS;
}
int rdamd_root_loglikelihood_fused_multi(unsigned int n_items, rdamd_partition_t *const *parts, const rdamd_operation_t *ops,
                                         const unsigned int *const *, const double *l1, const double *l2,
                                         const unsigned int *npos, double *out) {
  for (unsigned i = 0; i < n_items; ++i)
    for (unsigned a = 0; a < npos[i]; ++a)
      out[8 * i + a] = -(1.0 + 0.25 * parts[i]->rank) * (ops[i].parent_clv_index + l1[8 * i + a] * 2.0 + l2[8 * i + a]);
  return RDAMD_SUCCESS;
}
}

// ---- the site group's reducer: all G ranks meet, lengths must agree, sums in rank order -------------
struct group_t {
  int G;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  uint64_t generation = 0;
  unsigned n = 0;
  std::vector<double> sum;
  std::vector<std::vector<double>> parts;
  bool failed = false;
  uint64_t collectives = 0;
  bool check_sequence = false;   // every rank's vector must carry the same round identity
};
struct rank_ctx { group_t *g; int rank; uint64_t calls = 0, fault_call = 0; };
static int reducer_body(double *values, unsigned int n, void *, void *user);
static int reducer(double *values, unsigned int n, void *stream, void *user) {
  rank_ctx *c = (rank_ctx *)user;
  if (reducer_body(values, n, stream, user) != RDAMD_SUCCESS) return RDAMD_FAILURE;
  // this rank's copy of the sums: one ulp off (three reductions in a row: a first pass that is
  // repeated for somebody's second-pass flag is thrown away, and legitimately leaves no trace)
  if (c->fault_call && ++c->calls >= c->fault_call && c->calls < c->fault_call + 3) {
    uint64_t b;
    std::memcpy(&b, &values[0], 8);
    b ^= 1;
    std::memcpy(&values[0], &b, 8);
  }
  return RDAMD_SUCCESS;
}
static int reducer_wait(void *, void *) { return RDAMD_SUCCESS; }   // (the stand-in stream has run everything already)
static void reducer_abort(void *user) {
  group_t &g = *((rank_ctx *)user)->g;
  std::lock_guard<std::mutex> lk(g.mu);
  g.failed = true;
  g.cv.notify_all();
}
static int reducer_body(double *values, unsigned int n, void *, void *user) {
  rank_ctx *c = (rank_ctx *)user;
  group_t &g = *c->g;
  std::unique_lock<std::mutex> lk(g.mu);
  if (g.failed) return RDAMD_FAILURE;
  if (g.arrived == 0) { g.n = n; g.parts.assign(g.G, {}); }
  if (n != g.n) {   // the ranks' rounds have diverged: a real all-reduce would hang or corrupt
    std::fprintf(stderr, "reducer: rank %d arrived with %u values, the round has %u\n", c->rank, n, g.n);
    g.failed = true;
    g.cv.notify_all();
    return RDAMD_FAILURE;
  }
  g.parts[c->rank].assign(values, values + n);
  const uint64_t gen = g.generation;
  if (++g.arrived == g.G) {
    // (the conductor's guard: [.. | 1.0 | hash(worker group, round, requests) | hash(previous results)])
    for (int r = 1; g.check_sequence && r < g.G; ++r)
      if (n < 3 || g.parts[r][n - 2] != g.parts[0][n - 2]) {
        std::fprintf(stderr, "reducer: rank %d's collective is not the one rank 0 issued (order of the rounds differs)\n", r);
        g.failed = true;
        g.cv.notify_all();
        return RDAMD_FAILURE;
      }
    g.sum.assign(n, 0.0);
    for (int r = 0; r < g.G; ++r)
      for (unsigned i = 0; i < n; ++i) g.sum[i] += g.parts[r][i];
    g.arrived = 0;
    ++g.generation;
    ++g.collectives;
    g.cv.notify_all();
  } else {
    // (a plain wait: the timed waits of this toolchain's libstdc++ go through pthread_cond_clockwait,
    // which its ThreadSanitizer does not know -- ranks that fall out of step are caught by the length
    // check above or by the test driver's time limit)
    g.cv.wait(lk, [&] { return g.generation != gen || g.failed; });
    if (g.failed) return RDAMD_FAILURE;
  }
  std::copy(g.sum.begin(), g.sum.begin() + n, values);
  return RDAMD_SUCCESS;
}

static uint64_t mix(uint64_t h, double v) {
  uint64_t b;
  std::memcpy(&b, &v, 8);
  for (int i = 0; i < 8; ++i) h = (h ^ ((b >> (8 * i)) & 0xff)) * 1099511628211ull;
  return h;
}

int main(int argc, char **argv) {
  if (argc < 6 || argc > 8) return 2;
  const int G = std::atoi(argv[1]), W = std::atoi(argv[2]), groups = std::atoi(argv[3]), ncand = std::atoi(argv[4]);
  const unsigned seed = (unsigned)std::atoi(argv[5]);
  const bool device = argc > 6 && !std::strcmp(argv[6], "device");
  int fault_rank = -1;
  unsigned long long fault_call = 0;
  if (argc > 7 && std::sscanf(argv[7], "%d:%llu", &fault_rank, &fault_call) != 2) return 2;
  group_t group;
  group.G = G;
  group.check_sequence = true;
  std::vector<rdamd_partition> parts(G);
  std::vector<rank_ctx> ctx(G);
  std::vector<std::unique_ptr<rdamd::conductor_t>> cond;
  for (int r = 0; r < G; ++r) {
    parts[r].rank = r;
    ctx[r].g = &group; ctx[r].rank = r;
    if (r == fault_rank) ctx[r].fault_call = fault_call;
    rdamd::conductor_t::config_t cfg;
    cfg.shared = &parts[r];
    cfg.n_workers = (unsigned)W;
    cfg.n_groups = (unsigned)groups;
    cfg.n_candidates = (size_t)ncand;
    cfg.reduce = reducer;
    cfg.device = device;
    cfg.user = &ctx[r];
    if (device) { cfg.queue = reducer; cfg.wait = reducer_wait; cfg.async_user = &ctx[r]; }
    cfg.abort = reducer_abort;
    cfg.abort_user = &ctx[r];
    cond.emplace_back(new rdamd::conductor_t(cfg));
  }
  std::vector<uint64_t> digest((size_t)G * W, 1469598103934665603ull);
  std::vector<std::vector<long>> took((size_t)G * W);
  std::atomic<int> errors{0};
  auto worker = [&](int r, int w) {
    std::minstd_rand delay(seed * 7919u + (unsigned)(r * 131 + w));   // timing only: must not change any value
    auto nap = [&] {
      const unsigned k = delay() % 7;
      if (k == 0) std::this_thread::sleep_for(std::chrono::microseconds(delay() % 300));
      else if (k == 1) std::this_thread::yield();
    };
    uint64_t &h = digest[(size_t)r * W + w];
    rdamd_partition own{r};
    rdamd_partition *own_p = &own;
    const unsigned pidx0 = 0;
    const unsigned *pidx = &pidx0;
    try {
      for (;;) {
        nap();
        const long k = cond[r]->next_candidate((unsigned)w);
        if (k < 0) break;
        took[(size_t)r * W + w].push_back(k);
        rdamd_schedule sched{(double)(k + 1)};
        double state = 0.5 + 0.01 * (double)k;       // driven by the SUMS only: the same on every rank
        const int steps = 20 + (int)((k * 37) % 60);   // candidates differ in length
        for (int s = 0; s < steps; ++s) {
          nap();
          const unsigned pick = (unsigned)(std::fabs(state) * 1000.0) % 10u;
          if (pick < 5) {   // an optimiser step: 1 + n evaluations
            const unsigned n = 2 + pick * 3;
            std::vector<double> subst(2 * n), f(2 * n, 0.5), rt(n, 1.0), wt(n, 1.0), out(n);
            for (unsigned j = 0; j < n; ++j) { subst[2 * j] = state + 1e-3 * j; subst[2 * j + 1] = 0.1 * (s + 1); }
            cond[r]->objective((unsigned)w, n, &sched, subst.data(), f.data(), rt.data(), wt.data(), out.data());
            for (double v : out) { h = mix(h, v); state = 0.7 * state + 1e-3 * v; }
          } else if (pick < 8) {   // root positions of the candidate's branch
            const unsigned n = 1 + pick % 4;
            rdamd_operation_t op{};
            op.parent_clv_index = (unsigned)k;
            double l1[8] = {0}, l2[8] = {0}, out[8] = {0};
            for (unsigned a = 0; a < n; ++a) { l1[a] = state * (a + 1); l2[a] = 1.0 - 0.1 * a; }
            cond[r]->root((unsigned)w, &own_p, &pidx, 1, op, l1, l2, n, out);
            for (unsigned a = 0; a < n; ++a) { h = mix(h, out[a]); state = 0.9 * state - 1e-3 * out[a]; }
          } else {   // a value that only needs summing (compute_lh's lnL)
            double v[2] = {-(1.0 + 0.25 * r) * state, (double)s};
            cond[r]->reduce((unsigned)w, v, 2);
            h = mix(mix(h, v[0]), v[1]);
            state = 0.5 * state + 1e-3 * v[0];
          }
        }
      }
    } catch (const std::exception &e) {
      std::fprintf(stderr, "rank %d worker %d: %s\n", r, w, e.what());
      cond[r]->fail(e.what());
      ++errors;
    }
  };
  std::vector<std::thread> pool;
  for (int r = 0; r < G; ++r)
    for (int w = 0; w < W; ++w) pool.emplace_back(worker, r, w);
  for (auto &t : pool) t.join();
  if (errors.load() || group.failed) {
    std::printf("conductor FAILED\n");
    return 1;
  }
  // every rank: the same candidates on the same workers, the same values
  uint64_t all = 1469598103934665603ull;
  for (int w = 0; w < W; ++w) {
    for (int r = 1; r < G; ++r)
      if (digest[(size_t)r * W + w] != digest[w] || took[(size_t)r * W + w] != took[w]) {
        std::printf("conductor FAILED: rank %d worker %d differs from rank 0\n", r, w);
        return 1;
      }
    all = (all ^ digest[w]) * 1099511628211ull;
    for (long k : took[w]) all = (all ^ (uint64_t)k) * 1099511628211ull;
  }
  size_t handed = 0;
  for (int w = 0; w < W; ++w) handed += took[w].size();
  const auto st = cond[0]->stats();
  if (handed != (size_t)ncand || st.collectives != group.collectives) {
    std::printf("conductor FAILED: %zu of %d candidates handed out, %llu collectives counted, %llu seen\n", handed, ncand,
                (unsigned long long)st.collectives, (unsigned long long)group.collectives);
    return 1;
  }
  if (st.group_size != (uint64_t)G) {
    std::printf("conductor FAILED: the guard counted %llu ranks, there are %d\n", (unsigned long long)st.group_size, G);
    return 1;
  }
  std::printf("conductor OK rounds=%llu collectives=%llu redos=%llu digest=%016llx\n", (unsigned long long)st.rounds,
              (unsigned long long)st.collectives, (unsigned long long)st.redos, (unsigned long long)all);
  return 0;
}
