// Lock step in DETERMINISTIC ROUNDS: the meeting point of the candidates a site-sharded model
// has in flight (SURVEY.md 8e + 8(f)1; north star: "the outer root-edge loop batches lnL
// evaluations so candidate edges and site blocks shard across the 8 GPUs ... with an RCCL
// all-reduce of per-block log-likelihoods").
//
// The reference walks its candidate roots one after the other (src/model.cpp:1154-1229) and every
// likelihood it looks at is one traversal.  The batch combiners of batch_combiner.hpp let the
// candidates of ONE process meet in whatever launches their arrival order produces -- fine while
// nobody else has to agree with that order.  The ranks of a site group do have to: each holds a
// block of the alignment's columns, every value any optimiser sees is the SUM over the group,
// and a collective only works if all ranks issue the same collectives in the same order.
//
// What makes that possible: the ranks of a group run the same candidates on the same reduced
// values, so every candidate's SEQUENCE of requests is the same on every rank.  A round closes
// when EVERY live candidate of a worker group has posted its next request, whatever it is --
//   * the n + 1 evaluations of an L-BFGS-B step (src/model.cpp:1430-1522),
//   * up to eight root positions of its branch (compute_lh_root / compute_dlh / Brent /
//     optimize_alpha's scans, src/model.cpp:415-519, :606-794),
//   * values it only needs summed (compute_lh's lnL, src/model.cpp:384-413),
//   * or "my candidate is done, give me the next one" --
// so its composition is a function of the data, not of thread timing.  The round then is: the
// objective jobs of all candidates in worker order as ONE launch of the fused evaluator on the
// shared partition, the root positions of all candidates as ONE root_multi launch beside it,
// and ONE all-reduce over [objective lnLs | second-pass flag | root lnLs | plain values] queued
// behind them on the shared partition's stream: one collective per round instead of one per
// request (13 / 5 / 2-job batches and single Brent steps in the sequential loop).
//
// Two worker groups alternate STRICTLY (A0 B0 A1 B1 ...): while one group's round is on the
// device and in the collective, the other group's hosts take their optimiser steps; the turn is
// handed over when a round has been QUEUED, so the order of everything on the stream -- and of
// the collectives -- is the same on every rank.  Candidates are handed out inside the rounds, in
// worker order, from one counter: which worker runs which candidate is deterministic too.
//
// The evaluator's second pass (jobs whose tables make the per-site rescaling rule matter,
// fused.hpp) is the one decision a batch used to take on the host.  Here the flag travels
// through the all-reduce as one more summand (rdamd_evaluate_batch_submit_device): all ranks
// learn together that some rank needs the pass, all redo the collective, in the same place of
// the order (the redo waits for the group's turn).
//
// THE ASSUMPTION, AND ITS GUARD.  All of the above rests on one thing: every rank of the group
// receives the same BITS from the reducer.  The library's RCCL reducer makes that true by
// construction (comm.cpp, RDAMD_COMM_SUM_GATHER: gather + a sum in rank order); a caller's
// reducer, or RDAMD_COMM_SUM_ALLREDUCE under an algorithm that lets each rank add for itself, may
// not -- and one ulp of difference forks an optimiser's trajectory: some later round has other
// requests on one rank than on the others, the collectives no longer match, and the group would
// sit in one until the communicator's time limit.  So every round's vector ends in three GUARD
// words that travel through the same sum:
//   [ 1.0 | a 40-bit hash of (worker group, its round number, every request's worker / kind /
//     length) | a 40-bit hash of the bits of the PREVIOUS round's results ]
// and after the sum the round checks  word1 == G' x its own  and  word2 == G' x its own, with
// G' = the summed first word (small integers in doubles: these sums are exact in any order).  A
// rank whose results differed from the others' by a single bit is found in the very next round
// of its worker group, before its candidates' requests have had a chance to differ; requests
// that differ in kind or worker are found in the round they occur in.  The round fails AT ONCE on
// every rank, with its number in the message.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/root_digger_amd.h"

namespace rdamd {

class conductor_t {
public:
  struct config_t {
    rdamd_partition_t *shared = nullptr;   // the objective partition (the parent model's; not owned)
    unsigned n_workers = 1, n_groups = 1;
    size_t n_candidates = 0;
    // the site group's sum (all null: a one-rank "group", nothing to sum)
    rdamd_lnl_reducer_t reduce = nullptr;   // blocking form; host arrays unless `device`
    bool device = false;
    rdamd_lnl_reducer_t queue = nullptr;    // two halves of a device-side reducer (optional)
    rdamd_lnl_wait_t wait = nullptr;
    void *user = nullptr, *async_user = nullptr;   // of `reduce` / of `queue` and `wait`
    // called once when the search fails: makes pending and future reducer calls of THIS process
    // return failure (the other worker group's round may sit in a collective the ranks that failed
    // a moment earlier never join)
    void (*abort)(void *) = nullptr;
    void *abort_user = nullptr;
  };

  explicit conductor_t(const config_t &c) : _c(c) {
    if (!_c.shared || _c.n_workers < 1 || _c.n_groups < 1 || _c.n_groups > 2)
      throw std::invalid_argument("conductor: a shared partition, >= 1 workers, 1 or 2 groups");
    _c.n_groups = std::min(_c.n_groups, _c.n_workers);
    for (unsigned w = 0; w < _c.n_workers; ++w) ++_g[w % _c.n_groups].live;
  }
  ~conductor_t() {
    for (group_t &g : _g) {
      if (g.d_vec) (void)hipFree(g.d_vec);
      if (g.h_in) (void)hipHostFree(g.h_in);
      if (g.h_res) (void)hipHostFree(g.h_res);
      if (g.ev) (void)hipEventDestroy(g.ev);
    }
  }
  conductor_t(const conductor_t &) = delete;
  conductor_t &operator=(const conductor_t &) = delete;

  rdamd_partition_t *shared() const { return _c.shared; }

  // ---- what a worker (one candidate in flight, on its own model replica) may ask for ---------
  // index of its next candidate, or -1: nothing left, the worker has left its group
  long next_candidate(unsigned worker) {
    request_t r;
    r.kind = NEXT; r.worker = worker;
    post(r);
    return r.candidate;
  }
  // n jobs of one schedule of the shared partition (rdamd_evaluate_batch's blocks); out[j] =
  // the lnL of job j summed over the site group
  void objective(unsigned worker, unsigned n, const rdamd_schedule_t *sched, const double *subst,
                 const double *freqs, const double *rates, const double *weights, double *out) {
    request_t r;
    r.kind = OBJECTIVE; r.worker = worker; r.n = n; r.sched = sched;
    r.subst = subst; r.freqs = freqs; r.rates = rates; r.weights = weights; r.out = out;
    post(r);
  }
  // n <= 8 positions of root operation `op` on the worker's own partitions; out[a] = the lnL of
  // position a, summed over the partitions (in their order) and then over the site group
  void root(unsigned worker, rdamd_partition_t *const *parts, const unsigned *const *params_idx,
            unsigned n_parts, const rdamd_operation_t &op, const double *l1, const double *l2,
            unsigned n, double *out) {
    request_t r;
    r.kind = ROOT; r.worker = worker; r.n = n; r.parts = parts; r.params_idx = params_idx;
    r.n_parts = n_parts; r.op = op; r.out = out;
    for (unsigned a = 0; a < n; ++a) { r.l1[a] = l1[a]; r.l2[a] = l2[a]; }
    post(r);
  }
  // values[0 .. n) summed over the site group, in place
  void reduce(unsigned worker, double *values, unsigned n) {
    request_t r;
    r.kind = REDUCE; r.worker = worker; r.n = n; r.out = values;
    post(r);
  }
  // a worker that dies takes the search down: nobody may be left waiting for its request
  void fail(const std::string &what) {
    std::lock_guard<std::mutex> lk(_mu);
    set_error(what.empty() ? "a candidate failed" : what);
    _cv.notify_all();
  }

  // rounds closed, collectives queued, objective launches / their jobs, root launches / their steps
  // seconds[]: host time of the rounds' phases, summed: [0] the objective batch queued, [1] the
  // root-only launch (blocking), [2] the sum queued (host reducer: batch waited for and summed),
  // [3] waiting for the round's event and handing out the results
  struct stats_t {
    uint64_t rounds = 0, collectives = 0, obj_launches = 0, obj_jobs = 0, root_launches = 0, root_steps = 0, redos = 0;
    uint64_t group_size = 0;   // ranks the guard's counting word has seen in the sums (0: no round with a sum yet)
    double seconds[4] = {0, 0, 0, 0};
  };
  stats_t stats() const {
    std::lock_guard<std::mutex> lk(_mu);
    return _stats;
  }

private:
  enum kind_t { NEXT, OBJECTIVE, ROOT, REDUCE };
  struct request_t {
    kind_t kind = NEXT;
    unsigned worker = 0, n = 0;
    // objective
    const rdamd_schedule_t *sched = nullptr;
    const double *subst = nullptr, *freqs = nullptr, *rates = nullptr, *weights = nullptr;
    // root
    rdamd_partition_t *const *parts = nullptr;
    const unsigned *const *params_idx = nullptr;
    unsigned n_parts = 0;
    rdamd_operation_t op{};
    double l1[RDAMD_ROOT_MAX_POSITIONS] = {0}, l2[RDAMD_ROOT_MAX_POSITIONS] = {0};
    double *out = nullptr;    // objective / root: results; reduce: the values, in place
    long candidate = -1;      // next
    bool done = false;
    bool in_round = false;    // taken out of `posted` by the thread that runs the group's round: it holds the
                              // address until it sets `done` -- the poster must not leave before that
  };
  struct group_t {
    unsigned live = 0;
    bool busy = false;
    std::vector<request_t *> posted;
    // the round's vector: [objective lnLs | flag | root lnLs | plain values]
    double *d_vec = nullptr, *h_in = nullptr, *h_res = nullptr;
    size_t cap = 0;
    hipEvent_t ev = nullptr;   // behind the round's collective and the copy of its sums
    // the divergence guard (header comment): rounds of this group that carried a sum so far, and
    // the hash of the last one's results
    uint64_t seq = 0;
    double prev_word = 0.0;
  };
  static constexpr size_t GUARD_WORDS = 3;
  static uint64_t fnv(uint64_t h, const void *p, size_t n) {
    for (size_t i = 0; i < n; ++i) h = (h ^ ((const unsigned char *)p)[i]) * 1099511628211ull;
    return h;
  }
  template <class T> static uint64_t fnv(uint64_t h, const T &v) { return fnv(h, &v, sizeof v); }
  // 40 bits: G x word and the sum of G words are exact in a double for any G a node can hold
  static double word40(uint64_t h) { return (double)((h ^ (h >> 40)) & ((1ull << 40) - 1)); }

  [[noreturn]] static void hip_fail(const char *what, hipError_t e) {
    throw std::runtime_error(std::string("lock-step round: ") + what + ": " + hipGetErrorString(e));
  }
#define RDAMD_ROUND_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) hip_fail(#expr, e_); } while (0)

  // under _mu; the first error stays
  void set_error(const std::string &what) {
    if (!_error.empty()) return;
    _error = what;
    if (_c.abort) _c.abort(_c.abort_user);
  }
  bool my_turn(unsigned g) const { return _c.n_groups == 1 || _turn == g || _g[g ^ 1u].live == 0; }
  void pass_turn(unsigned g) {
    if (_c.n_groups == 2 && _g[g ^ 1u].live > 0) _turn = g ^ 1u;
    else _turn = g;
  }

  void post(request_t &req) {
    const unsigned g = req.worker % _c.n_groups;
    group_t &grp = _g[g];
    std::unique_lock<std::mutex> lk(_mu);
    if (!_error.empty()) throw std::runtime_error(_error);
    grp.posted.push_back(&req);
    _cv.notify_all();
    for (;;) {
      if (req.done) break;
      if (!_error.empty() && !req.in_round) {
        // (`req` lives on the caller's stack: nothing may keep its address -- and while a round that was
        // started before the error holds it, this thread stays: the round's thread ends it, error or not)
        grp.posted.erase(std::remove(grp.posted.begin(), grp.posted.end(), &req), grp.posted.end());
        throw std::runtime_error(_error);
      }
      if (!grp.busy && !grp.posted.empty() && grp.posted.size() == grp.live && my_turn(g)) {
        // everybody of the group has asked and it is the group's turn: this thread runs the round
        std::vector<request_t *> round;
        round.swap(grp.posted);
        for (request_t *r : round) r->in_round = true;
        grp.busy = true;
        try {
          run_round(g, round, lk);
        } catch (const std::exception &e) {
          if (!lk.owns_lock()) lk.lock();
          set_error(e.what());
        }
        grp.busy = false;
        for (request_t *r : round) r->done = true;
        _cv.notify_all();
        continue;
      }
      if (!_debug) {
        _cv.wait(lk);
      } else if (_cv.wait_for(lk, std::chrono::seconds(20)) == std::cv_status::timeout) {
        // RDAMD_LOCKSTEP_DEBUG: who is everybody waiting for?
        std::fprintf(stderr, "[lockstep] worker %u (group %u, kind %d) still waiting: turn %u", req.worker, g, (int)req.kind, _turn);
        for (unsigned k = 0; k < _c.n_groups; ++k) {
          std::fprintf(stderr, " | group %u live %u busy %d posted", k, _g[k].live, (int)_g[k].busy);
          for (const request_t *r : _g[k].posted) std::fprintf(stderr, " %u:%d", r->worker, (int)r->kind);
        }
        std::fprintf(stderr, " | rounds %llu next %zu/%zu\n", (unsigned long long)_stats.rounds, _next, _c.n_candidates);
      }
    }
    if (!_error.empty() && req.kind != NEXT) throw std::runtime_error(_error);
  }

  void ensure_vec(group_t &grp, size_t n, hipStream_t stream) {
    if (n <= grp.cap) return;
    RDAMD_ROUND_TRY(hipStreamSynchronize(stream));
    if (grp.d_vec) (void)hipFree(grp.d_vec);
    if (grp.h_in) (void)hipHostFree(grp.h_in);
    if (grp.h_res) (void)hipHostFree(grp.h_res);
    grp.d_vec = grp.h_in = grp.h_res = nullptr;
    grp.cap = std::max<size_t>(2 * n, 1024);
    RDAMD_ROUND_TRY(hipMalloc((void **)&grp.d_vec, grp.cap * sizeof(double)));
    RDAMD_ROUND_TRY(hipHostMalloc((void **)&grp.h_in, grp.cap * sizeof(double), hipHostMallocDefault));
    RDAMD_ROUND_TRY(hipHostMalloc((void **)&grp.h_res, grp.cap * sizeof(double), hipHostMallocDefault));
  }

  // `lk` is held on entry and on return
  void run_round(unsigned g, std::vector<request_t *> &round, std::unique_lock<std::mutex> &lk) {
    group_t &grp = _g[g];
    // (asked for every round: rdamd_partition_set_stream_priority re-creates the stream; a local --
    // the other group's round may be under way)
    const hipStream_t stream = (hipStream_t)rdamd_partition_stream(_c.shared);
    if (!grp.ev) RDAMD_ROUND_TRY(hipEventCreateWithFlags(&grp.ev, hipEventDisableTiming));
    std::sort(round.begin(), round.end(), [](const request_t *a, const request_t *b) { return a->worker < b->worker; });
    // ---- candidates, in worker order (under the lock: the counter is shared by the groups,
    // and the turn makes the order of the groups' rounds the same everywhere)
    std::vector<request_t *> obj, root, red;
    uint64_t comp = fnv(fnv(1469598103934665603ull, g), grp.seq);
    for (request_t *r : round) {
      comp = fnv(fnv(fnv(comp, r->worker), (unsigned)r->kind), r->n);
      switch (r->kind) {
        case NEXT:
          if (_next < _c.n_candidates) r->candidate = (long)_next++;
          else { r->candidate = -1; --grp.live; }
          break;
        case OBJECTIVE: obj.push_back(r); break;
        case ROOT: root.push_back(r); break;
        case REDUCE: red.push_back(r); break;
      }
    }
    ++_stats.rounds;
    if (obj.empty() && root.empty() && red.empty()) {
      pass_turn(g);
      return;
    }
    lk.unlock();
    double phase[4] = {0, 0, 0, 0};
    auto t_mark = std::chrono::steady_clock::now();
    auto lap = [&](int k) {
      const auto now = std::chrono::steady_clock::now();
      phase[k] += std::chrono::duration<double>(now - t_mark).count();
      t_mark = now;
    };

    // ---- the round's vector
    size_t m = 0, n_root = 0, n_red = 0;
    for (request_t *r : obj) m += r->n;
    for (request_t *r : root) n_root += r->n;
    for (request_t *r : red) n_red += r->n;
    const size_t o_flag = m, o_root = m + 1, o_red = o_root + n_root, o_guard = o_red + n_red, total = o_guard + GUARD_WORDS;
    ensure_vec(grp, total, stream);
    const uint64_t round_no = grp.seq++;   // (of this worker group, counting the rounds that carry a sum)
    // what an objective batch in flight needs when the round fails after it was queued: its slot
    // released and the partition's batch sequence advanced (parked schedule blocks wait for that)
    struct batch_in_flight_t {
      rdamd_partition_t *p; unsigned slot; bool device, armed = false;
      ~batch_in_flight_t() {
        if (!armed) return;
        if (device) (void)rdamd_evaluate_batch_finish_device(p, slot);
        else { std::vector<double> sink(n); (void)rdamd_evaluate_batch_wait(p, slot, sink.data()); }
      }
      size_t n = 0;
    } in_flight{_c.shared, g, false};
    const bool two_phase = _c.device && _c.queue && _c.wait;
    const bool device_path = !_c.reduce || _c.device;   // (no reducer at all: the device path without a collective)

    // ---- the objective jobs: one launch on the shared partition
    if (m) {
      const unsigned R = rdamd_partition_rate_cats(_c.shared);
      const unsigned K = rdamd_partition_states(_c.shared), NP = K * K - K;
      std::vector<const rdamd_schedule_t *> scheds;
      std::vector<double> subst, freqs, rates, weights;
      scheds.reserve(m);
      for (request_t *r : obj) {
        scheds.insert(scheds.end(), r->n, r->sched);
        subst.insert(subst.end(), r->subst, r->subst + (size_t)r->n * NP);
        freqs.insert(freqs.end(), r->freqs, r->freqs + (size_t)r->n * K);
        rates.insert(rates.end(), r->rates, r->rates + (size_t)r->n * R);
        weights.insert(weights.end(), r->weights, r->weights + (size_t)r->n * R);
      }
      const int rc = device_path
          ? rdamd_evaluate_batch_submit_device(_c.shared, g, (unsigned)m, scheds.data(), subst.data(), freqs.data(),
                                               rates.data(), weights.data(), grp.d_vec)
          : rdamd_evaluate_batch_submit(_c.shared, g, (unsigned)m, scheds.data(), subst.data(), freqs.data(),
                                        rates.data(), weights.data());
      if (rc != RDAMD_SUCCESS) throw std::runtime_error(std::string("lock-step round: objective batch: ") + rdamd_errmsg());
      in_flight.device = device_path; in_flight.n = m; in_flight.armed = true;
    }
    lap(0);
    // ---- the root positions: one launch over the candidates' own partitions, beside it
    if (!root.empty()) {
      constexpr unsigned P = RDAMD_ROOT_MAX_POSITIONS;
      std::vector<rdamd_partition_t *> parts;
      std::vector<rdamd_operation_t> ops;
      std::vector<const unsigned *> pidx;
      std::vector<double> l1, l2;
      std::vector<unsigned> npos;
      for (request_t *r : root)
        for (unsigned i = 0; i < r->n_parts; ++i) {
          parts.push_back(r->parts[i]); ops.push_back(r->op); pidx.push_back(r->params_idx[i]);
          npos.push_back(r->n);
          l1.insert(l1.end(), r->l1, r->l1 + P);
          l2.insert(l2.end(), r->l2, r->l2 + P);
        }
      std::vector<double> v(P * parts.size());
      if (rdamd_root_loglikelihood_fused_multi((unsigned)parts.size(), parts.data(), ops.data(), pidx.data(), l1.data(),
                                               l2.data(), npos.data(), v.data()) != RDAMD_SUCCESS)
        throw std::runtime_error(std::string("lock-step round: root step: ") + rdamd_errmsg());
      size_t item = 0, at = o_root;
      for (request_t *r : root) {   // (a candidate's partitions summed in their order, as compute_lh_root does)
        for (unsigned a = 0; a < r->n; ++a) grp.h_in[at + a] = 0.0;
        for (unsigned i = 0; i < r->n_parts; ++i, ++item)
          for (unsigned a = 0; a < r->n; ++a) grp.h_in[at + a] += v[P * item + a];
        at += r->n;
      }
    }
    {
      size_t at = o_red;
      for (request_t *r : red) {
        std::copy(r->out, r->out + r->n, grp.h_in + at);
        at += r->n;
      }
    }
    // ---- the guard words (header comment)
    const double comp_word = word40(comp), prev_word = grp.prev_word;
    grp.h_in[o_guard] = 1.0;
    grp.h_in[o_guard + 1] = comp_word;
    grp.h_in[o_guard + 2] = prev_word;
    const auto check_guard = [&](const double *sums) {
      const double ranks = sums[o_guard];
      const char *what = nullptr;
      if (!(ranks >= 1.0) || ranks != std::floor(ranks) || ranks > 65536.0) what = "the ranks do not add up to a whole number";
      else if (_group_size.load() && (double)_group_size.load() != ranks) what = "the number of ranks in the sum has changed";
      else if (sums[o_guard + 1] != ranks * comp_word) what = "the ranks' candidates have posted different requests";
      else if (sums[o_guard + 2] != ranks * prev_word)
        what = "the ranks did not receive the same bits from the group's previous sum";
      if (what) {
        char msg[512];
        std::snprintf(msg, sizeof msg,
                      "lock-step round %llu of worker group %u (%zu values): the site group has diverged -- %s.  "
                      "Every rank must get identical sums from the reducer (the library's default, ncclAllGather + a sum "
                      "in rank order, does by construction)", (unsigned long long)round_no, g, total, what);
        throw std::runtime_error(msg);
      }
      _group_size = (uint64_t)ranks;
    };

    lap(1);
    // ---- the sum over the site group
    double *res = grp.h_res;
    bool need_redo_check = false;
    if (device_path) {
      if (!m) grp.h_in[o_flag] = 0.0;
      // (host-made values go up behind the batch; the front of the vector is the batch's)
      const size_t lo = m ? o_root : 0;
      if (total > lo)
        RDAMD_ROUND_TRY(hipMemcpyAsync(grp.d_vec + lo, grp.h_in + lo, (total - lo) * sizeof(double), hipMemcpyHostToDevice, stream));
      queue_sum(grp.d_vec, total, two_phase, stream);
      RDAMD_ROUND_TRY(hipMemcpyAsync(grp.h_res, grp.d_vec, total * sizeof(double), hipMemcpyDeviceToHost, stream));
      RDAMD_ROUND_TRY(hipEventRecord(grp.ev, stream));
      need_redo_check = m > 0;
    } else {
      // host reducer (ranks that share a device): the batch's own wait runs the second pass where
      // this rank needs it -- its values are final before they are summed
      in_flight.armed = false;   // (the wait releases the slot whether it succeeds or not)
      if (m && rdamd_evaluate_batch_wait(_c.shared, g, grp.h_in) != RDAMD_SUCCESS)
        throw std::runtime_error(std::string("lock-step round: objective batch: ") + rdamd_errmsg());
      grp.h_in[o_flag] = 0.0;
      if (_c.reduce(grp.h_in, (unsigned)total, nullptr, _c.user) != RDAMD_SUCCESS)
        throw std::runtime_error(std::string("lock-step round: site-group reduction failed: ") + rdamd_errmsg());
      res = grp.h_in;
      check_guard(res);
    }

    lap(2);
    // ---- queued: the other group may go
    lk.lock();
    ++_stats.collectives;
    if (m) { ++_stats.obj_launches; _stats.obj_jobs += m; }
    if (!root.empty()) { ++_stats.root_launches; _stats.root_steps += root.size(); }
    pass_turn(g);
    _cv.notify_all();
    lk.unlock();

    // ---- results
    if (device_path) {
      wait_round(grp, two_phase);
      check_guard(grp.h_res);
      if (need_redo_check && grp.h_res[o_flag] != 0.0) {
        // some rank's batch wants its second pass: every rank repeats the collective -- in the
        // group's turn, so that it sits at the same place of the stream's order everywhere
        lk.lock();
        while (!my_turn(g) && _error.empty()) _cv.wait(lk);
        if (!_error.empty()) throw std::runtime_error(_error);
        ++_stats.redos; ++_stats.collectives;
        lk.unlock();   // (the turn stays here: the other group passes it only when it has it)
        if (rdamd_evaluate_batch_redo_device(_c.shared, g, grp.d_vec) != RDAMD_SUCCESS)
          throw std::runtime_error(std::string("lock-step round: second pass: ") + rdamd_errmsg());
        if (total > o_root)
          RDAMD_ROUND_TRY(hipMemcpyAsync(grp.d_vec + o_root, grp.h_in + o_root, (total - o_root) * sizeof(double), hipMemcpyHostToDevice, stream));
        queue_sum(grp.d_vec, total, two_phase, stream);
        RDAMD_ROUND_TRY(hipMemcpyAsync(grp.h_res, grp.d_vec, total * sizeof(double), hipMemcpyDeviceToHost, stream));
        RDAMD_ROUND_TRY(hipEventRecord(grp.ev, stream));
        wait_round(grp, two_phase);
        check_guard(grp.h_res);
      }
      in_flight.armed = false;
      if (m && rdamd_evaluate_batch_finish_device(_c.shared, g) != RDAMD_SUCCESS)
        throw std::runtime_error(std::string("lock-step round: objective batch: ") + rdamd_errmsg());
    }
    // (what the NEXT round of this group vouches for: the bits every candidate is about to see)
    grp.prev_word = word40(fnv(1469598103934665603ull, res, o_guard * sizeof(double)));
    {
      size_t at = 0;
      for (request_t *r : obj) { std::copy(res + at, res + at + r->n, r->out); at += r->n; }
      at = o_root;
      for (request_t *r : root) { std::copy(res + at, res + at + r->n, r->out); at += r->n; }
      at = o_red;
      for (request_t *r : red) { std::copy(res + at, res + at + r->n, r->out); at += r->n; }
    }
    lap(3);
    lk.lock();
    for (int k = 0; k < 4; ++k) _stats.seconds[k] += phase[k];
    _stats.group_size = _group_size.load();
  }

  // the site group's sum over d[0 .. n), queued on the shared partition's stream
  void queue_sum(double *d, size_t n, bool two_phase, hipStream_t stream) {
    if (!_c.reduce && !_c.queue) return;   // a one-rank group
    const int rc = two_phase ? _c.queue(d, (unsigned)n, (void *)stream, _c.async_user)
                             : _c.reduce(d, (unsigned)n, (void *)stream, _c.user);
    if (rc != RDAMD_SUCCESS)
      throw std::runtime_error(std::string("lock-step round: site-group reduction failed: ") + rdamd_errmsg());
  }
  // (the round's event, not the stream: the other group's round may be queued behind this one)
  void wait_round(group_t &grp, bool two_phase) {
    if (two_phase) {
      if (_c.wait((void *)grp.ev, _c.async_user) != RDAMD_SUCCESS)
        throw std::runtime_error(std::string("lock-step round: site-group reduction failed: ") + rdamd_errmsg());
    } else {
      RDAMD_ROUND_TRY(hipEventSynchronize(grp.ev));
    }
  }
#undef RDAMD_ROUND_TRY

  config_t _c;
  mutable std::mutex _mu;
  std::condition_variable _cv;
  group_t _g[2];
  unsigned _turn = 0;
  size_t _next = 0;
  std::atomic<uint64_t> _group_size{0};   // the guard's counting word, once seen (the two worker groups' rounds overlap)
  const bool _debug = std::getenv("RDAMD_LOCKSTEP_DEBUG") != nullptr;
  std::string _error;
  stats_t _stats;
};

}  // namespace rdamd
