// Lock-step batching of the optimiser's objective across candidate roots
// (SURVEY.md 8(f)1).  The reference optimises one candidate root per MPI rank
// (src/model.cpp:1154-1229, :1899-1907); each L-BFGS-B step asks for n+1
// likelihood evaluations (:1430-1522).  Here several candidates run on host
// threads of one process, and instead of each launching its own 13-job batch,
// their requests meet in this combiner: when every candidate that is currently
// inside optimize_params has asked, ONE fused launch on a shared partition
// evaluates all of them.  Small alignments, whose single batches are
// launch-bound, gain the most; large ones keep the GPU at its wide-batch rate
// with a single copy of the tip data for the objective.
#pragma once

#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/root_digger_amd.h"

namespace rdamd {

//
// Two combiners on ONE shared partition form a pipeline (rdamd_evaluate_batch_submit / _wait,
// slots 0 and 1): the candidates in flight are split into two groups, and while one group's
// batch is on the device the other group's host threads take their L-BFGS-B steps and queue
// the next batch behind it -- the evaluator launches follow each other without the gap a
// single group leaves between its launches (results back, n setulb calls, parameters up,
// P-matrices, clade tables: ~250 us on c2 against 1.25 ms of evaluator per launch).
class batch_combiner_t {
public:
  // `shared`: the partition every combined launch runs on (not owned); slot < 0: blocking
  // launches (one group), 0 / 1: this group's slot of the pipeline
  explicit batch_combiner_t(rdamd_partition_t *shared, int slot = -1) : _part(shared), _slot(slot) {}

  rdamd_partition_t *partition() const { return _part; }

  // a candidate enters / leaves the phase in which it submits requests
  void enter() {
    std::lock_guard<std::mutex> g(_mu);
    ++_active;
  }
  void leave() {
    {
      std::lock_guard<std::mutex> g(_mu);
      --_active;
    }
    _cv.notify_all();   // the others may now be complete without this one
  }
  struct scope_t {   // RAII enter/leave
    batch_combiner_t *c;
    explicit scope_t(batch_combiner_t *c_) : c(c_) { if (c) c->enter(); }
    ~scope_t() { if (c) c->leave(); }
    scope_t(const scope_t &) = delete;
    scope_t &operator=(const scope_t &) = delete;
  };

  // schedules live on the shared partition (the library serialises what touches it)
  rdamd_schedule_t *schedule_create(const rdamd_operation_t *ops, unsigned n_ops,
                                    const unsigned *matrix_indices, const double *branch_lengths,
                                    unsigned n_matrices) {
    return rdamd_schedule_create(_part, ops, n_ops, matrix_indices, branch_lengths, n_matrices);
  }
  void schedule_destroy(rdamd_schedule_t *s) { rdamd_schedule_destroy(s); }

  // n jobs of one candidate (same schedule): subst [n][K*K-K], freqs [n][K],
  // rates / weights [n][R].  Returns when the launch that carried them is done.
  void evaluate(unsigned n, const rdamd_schedule_t *sched, const double *subst,
                const double *freqs, const double *rates, const double *weights, double *out) {
    request_t req{n, sched, subst, freqs, rates, weights, out};
    std::unique_lock<std::mutex> lk(_mu);
    _pending.push_back(&req);
    for (;;) {
      if (req.done) break;
      if (!_launching && !_pending.empty() && _pending.size() >= (size_t)_active) {
        // everyone who can ask has asked: this thread launches for all of them
        std::vector<request_t *> batch;
        batch.swap(_pending);
        _launching = true;
        lk.unlock();
        std::string err;
        try {
          err = launch(batch);
        } catch (const std::exception &e) {   // nobody may be left waiting on this batch
          err = std::string("combined evaluate_batch failed: ") + e.what();
        }
        lk.lock();
        _launching = false;
        for (request_t *r : batch) {
          r->error = err;
          r->done = true;
        }
        ++_launches;
        _jobs += count(batch);
        _cv.notify_all();
        continue;
      }
      _cv.wait(lk);
    }
    if (!req.error.empty()) throw std::runtime_error(req.error);
  }

  size_t launches() const { return _launches; }
  size_t jobs() const { return _jobs; }

private:
  struct request_t {
    unsigned n;
    const rdamd_schedule_t *sched;
    const double *subst, *freqs, *rates, *weights;
    double *out;
    bool done = false;
    std::string error;
  };
  static size_t count(const std::vector<request_t *> &b) {
    size_t n = 0;
    for (auto *r : b) n += r->n;
    return n;
  }
  std::string launch(const std::vector<request_t *> &batch) {
    const unsigned R = rdamd_partition_rate_cats(_part);
    const unsigned K = rdamd_partition_states(_part), NP = K * K - K;
    const size_t total = count(batch);
    std::vector<const rdamd_schedule_t *> scheds;
    std::vector<double> subst, freqs, rates, weights, out(total);
    scheds.reserve(total);
    for (auto *r : batch) {
      scheds.insert(scheds.end(), r->n, r->sched);
      subst.insert(subst.end(), r->subst, r->subst + (size_t)r->n * NP);
      freqs.insert(freqs.end(), r->freqs, r->freqs + (size_t)r->n * K);
      rates.insert(rates.end(), r->rates, r->rates + (size_t)r->n * R);
      weights.insert(weights.end(), r->weights, r->weights + (size_t)r->n * R);
    }
    if (_slot < 0) {
      if (rdamd_evaluate_batch(_part, (unsigned)total, scheds.data(), subst.data(), freqs.data(),
                               rates.data(), weights.data(), out.data()) != RDAMD_SUCCESS)
        return std::string("combined evaluate_batch failed: ") + rdamd_errmsg();
    } else {
      if (rdamd_evaluate_batch_submit(_part, (unsigned)_slot, (unsigned)total, scheds.data(), subst.data(),
                                      freqs.data(), rates.data(), weights.data()) != RDAMD_SUCCESS ||
          rdamd_evaluate_batch_wait(_part, (unsigned)_slot, out.data()) != RDAMD_SUCCESS)
        return std::string("combined evaluate_batch failed: ") + rdamd_errmsg();
    }
    size_t at = 0;
    for (auto *r : batch) {
      std::copy(out.begin() + (std::ptrdiff_t)at, out.begin() + (std::ptrdiff_t)(at + r->n), r->out);
      at += r->n;
    }
    return std::string();
  }

  rdamd_partition_t *_part;
  int _slot;
  std::mutex _mu;
  std::condition_variable _cv;
  std::vector<request_t *> _pending;
  int _active = 0;
  bool _launching = false;
  size_t _launches = 0, _jobs = 0;
};

// The same meeting point for the ROOT-ONLY steps (compute_lh_root / compute_dlh: Brent's
// method on the root position, src/model.cpp:606-794, and optimize_alpha's scans).  Every
// candidate in flight sits on its own model replica -- its own child CLVs, parameters and
// stream --, so a combined step is one launch over several partitions
// (rdamd_root_loglikelihood_fused_multi) instead of one per candidate: a c2 search issues
// ~14 000 of them for 13 candidates.  A candidate enters when it starts to place the root on
// its branch and leaves when it is done; when everyone inside has asked, the last one launches.
class root_combiner_t {
public:
  void enter() {
    std::lock_guard<std::mutex> g(_mu);
    ++_active;
  }
  void leave() {
    {
      std::lock_guard<std::mutex> g(_mu);
      --_active;
    }
    _cv.notify_all();
  }
  struct scope_t {
    root_combiner_t *c;
    explicit scope_t(root_combiner_t *c_) : c(c_) { if (c) c->enter(); }
    ~scope_t() { if (c) c->leave(); }
    scope_t(const scope_t &) = delete;
    scope_t &operator=(const scope_t &) = delete;
  };

  // n <= 8 root positions (4 at 8 rate categories) of `op` on every one of the model's
  // `n_parts` partitions: branch lengths l1 / l2; out[8 i + a] = lnL of partition i at
  // position a (the caller sums over its partitions)
  void evaluate(rdamd_partition_t *const *parts, const unsigned *const *params_idx, unsigned n_parts,
                const rdamd_operation_t &op, const double *l1, const double *l2, unsigned n, double *out) {
    request_t req{parts, params_idx, n_parts, op, {0}, {0}, n, out};
    for (unsigned a = 0; a < n; ++a) { req.l1[a] = l1[a]; req.l2[a] = l2[a]; }
    std::unique_lock<std::mutex> lk(_mu);
    _pending.push_back(&req);
    for (;;) {
      if (req.done) break;
      if (!_launching && !_pending.empty() && _pending.size() >= (size_t)std::max(_active, 1)) {
        std::vector<request_t *> batch;
        batch.swap(_pending);
        _launching = true;
        lk.unlock();
        std::string err;
        try {
          err = launch(batch);
        } catch (const std::exception &e) {
          err = std::string("combined root step failed: ") + e.what();
        }
        lk.lock();
        _launching = false;
        for (request_t *r : batch) {
          r->error = err;
          r->done = true;
        }
        ++_launches;
        _steps += batch.size();
        _cv.notify_all();
        continue;
      }
      _cv.wait(lk);
    }
    if (!req.error.empty()) throw std::runtime_error(req.error);
  }
  size_t launches() const { return _launches; }
  size_t steps() const { return _steps; }

private:
  struct request_t {
    rdamd_partition_t *const *parts;
    const unsigned *const *params_idx;
    unsigned n_parts;
    rdamd_operation_t op;
    double l1[RDAMD_ROOT_MAX_POSITIONS], l2[RDAMD_ROOT_MAX_POSITIONS];
    unsigned n;
    double *out;
    bool done = false;
    std::string error;
  };
  static std::string launch(const std::vector<request_t *> &batch) {
    constexpr unsigned P = RDAMD_ROOT_MAX_POSITIONS;
    size_t m = 0;
    for (auto *r : batch) m += r->n_parts;
    std::vector<rdamd_partition_t *> parts;
    std::vector<rdamd_operation_t> ops;
    std::vector<const unsigned *> pidx;
    std::vector<double> l1, l2, out(P * m);
    std::vector<unsigned> npos;
    for (auto *r : batch)
      for (unsigned i = 0; i < r->n_parts; ++i) {   // one item per (candidate, partition)
        parts.push_back(r->parts[i]); ops.push_back(r->op); pidx.push_back(r->params_idx[i]);
        npos.push_back(r->n);
        l1.insert(l1.end(), r->l1, r->l1 + P);
        l2.insert(l2.end(), r->l2, r->l2 + P);
      }
    if (rdamd_root_loglikelihood_fused_multi((unsigned)m, parts.data(), ops.data(), pidx.data(), l1.data(),
                                             l2.data(), npos.data(), out.data()) != RDAMD_SUCCESS)
      return std::string("combined root step failed: ") + rdamd_errmsg();
    size_t at = 0;
    for (auto *r : batch)
      for (unsigned i = 0; i < r->n_parts; ++i, ++at)
        for (unsigned a = 0; a < r->n; ++a) r->out[P * i + a] = out[P * at + a];
    return std::string();
  }
  std::mutex _mu;
  std::condition_variable _cv;
  std::vector<request_t *> _pending;
  int _active = 0;
  bool _launching = false;
  size_t _launches = 0, _steps = 0;
};

}  // namespace rdamd
