// Batched full-traversal evaluation through the fused kernel: the C ABI around
// kernels_fused.hip.  A "schedule" is a traversal compiled once on the host
// (operation order chosen to minimise the LDS stack, Sethi-Ullman style) and
// kept in HBM; a batch evaluates many (schedule, parameter set) jobs in one
// launch, which is what the L-BFGS-B finite-difference gradient
// (/root/reference/src/model.cpp:1488-1502) and the candidate-root loop
// (src/model.cpp:1154) ask for.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <unordered_map>

#include <cstdlib>
#include "common.hpp"
#include "fused.hpp"

struct rdamd_schedule {
  rdamd_partition *part = nullptr;
  rdamd::FusedOp *d_prog = nullptr;
  double *d_brlen = nullptr;
  unsigned n_ops = 0, depth = 0;      // depth: LDS stack levels the program needs
  unsigned reg_levels = 1;            // register stack levels it was compiled for
  std::vector<rdamd::FusedOp> prog;   // host copy (tests / debugging)
};

namespace rdamd {

struct FusedWorkspace {
  unsigned cap_jobs = 0, blocks_x = 0;
  char *d_in = nullptr;   // per-batch inputs (see ensure_workspace)
  FusedJob *d_jobs = nullptr;   // ... and where this batch's pieces sit inside it
  double *d_q = nullptr, *d_rates = nullptr, *d_freqs = nullptr, *d_rw = nullptr;
  double *d_pmat = nullptr, *d_tiptab = nullptr, *d_partials = nullptr, *d_out = nullptr;
  double *h_out = nullptr;   // pinned
  char *h_in = nullptr;      // pinned parameter staging
  size_t h_in_bytes = 0;
};

void fused_workspace_free(FusedWorkspace *w) {
  if (!w) return;
  void *dev[] = {w->d_in, w->d_pmat, w->d_tiptab, w->d_partials, w->d_out};
  for (void *d : dev)
    if (d) (void)hipFree(d);
  if (w->h_out) (void)hipHostFree(w->h_out);
  if (w->h_in) (void)hipHostFree(w->h_in);
  delete w;
}

static hipError_t ensure_workspace(rdamd_partition *p, unsigned n_jobs) {
  if (!p->fused) p->fused = new FusedWorkspace();
  FusedWorkspace *w = p->fused;
  if (n_jobs <= w->cap_jobs) return hipSuccess;
  hipError_t e = hipStreamSynchronize(p->stream);
  if (e != hipSuccess) return e;
  void *dev[] = {w->d_in, w->d_pmat, w->d_tiptab, w->d_partials, w->d_out};
  for (void *d : dev)
    if (d) (void)hipFree(d);
  if (w->h_out) (void)hipHostFree(w->h_out);
  if (w->h_in) (void)hipHostFree(w->h_in);
  *w = FusedWorkspace();
  const unsigned cap = std::max(16u, n_jobs + n_jobs / 2);
  const unsigned R = p->rate_cats, K = p->states;
  // 4 states: site blocks padded to a multiple of 16 (8 at two sites per lane) so that blockIdx.x % 8 (the
  // XCD a workgroup lands on) is the same for every job: each XCD's L2 then only
  // ever sees 1/8 of the tip codes.  20 states: one partial per 16-site tile.
  const unsigned per_block = 64;
  w->blocks_x = K == 4 ? ((p->sites + per_block - 1) / per_block + 15) / 16 * 16 : (p->sites + 15) / 16;
  const size_t pm_per_job = (size_t)p->prob_matrices * R * K * K;
#define A(ptr, bytes) do { e = hipMalloc((void **)&(ptr), (bytes)); if (e != hipSuccess) return e; } while (0)
  // jobs + Q + frequencies + rates + rate weights of a batch live in ONE device block
  // with the layout of the pinned staging block: one copy per batch instead of five
  A(w->d_in, (size_t)cap * (sizeof(FusedJob) + sizeof(double) * (K * K + K + 2 * R)));
  A(w->d_pmat, sizeof(double) * pm_per_job * cap);
  A(w->d_tiptab, sizeof(double) * (K == 4 ? pm_per_job * 4
                                           : (size_t)p->prob_matrices * R * kFused20TabDoubles) * cap);
  A(w->d_partials, sizeof(double) * w->blocks_x * cap);
  A(w->d_out, sizeof(double) * cap);
#undef A
  e = hipHostMalloc((void **)&w->h_out, sizeof(double) * cap, hipHostMallocDefault);
  if (e != hipSuccess) return e;
  w->h_in_bytes = (size_t)cap * (sizeof(FusedJob) + sizeof(double) * (K * K + K + 2 * R));
  e = hipHostMalloc((void **)&w->h_in, w->h_in_bytes, hipHostMallocDefault);
  if (e != hipSuccess) return e;
  w->cap_jobs = cap;
  return hipSuccess;
}

// ---- traversal compiler -----------------------------------------------------
// Input: operations in dependency order, the last one being the root.  Output:
// the same operations re-ordered so that, at every inner-inner node, the child
// needing the deeper stack is evaluated first (its result is parked in LDS
// while the other child runs), plus the flags the kernel interprets.
struct Compiler {
  const rdamd_operation_t *ops;
  unsigned n_ops, tips, sites, tip_stride, rate_cats;
  unsigned unit = 0;         // bytes between the [rate 0] entries of consecutive matrices
  bool split_park = false;   // 20-state programs: parking is a step of its own
  unsigned reg_levels = 1;   // stack levels the kernel keeps in registers (4 states: 1 or 2)
  std::unordered_map<unsigned, unsigned> producer;   // clv -> op index
  std::vector<unsigned> need;                        // stack slots a subtree needs
  std::vector<FusedOp> out;
  unsigned depth = 0, max_depth = 0;
  bool ok = true;

  bool is_inner(unsigned clv) const { return clv >= tips; }

  unsigned compute_need(unsigned i) {
    const rdamd_operation_t &o = ops[i];
    unsigned n1 = 0, n2 = 0;
    const bool i1 = is_inner(o.child1_clv_index), i2 = is_inner(o.child2_clv_index);
    if (i1) n1 = compute_need(producer.at(o.child1_clv_index));
    if (i2) n2 = compute_need(producer.at(o.child2_clv_index));
    unsigned r;
    if (i1 && i2) r = std::max(std::max(n1, n2), std::min(n1, n2) + 1);
    else r = i1 ? n1 : (i2 ? n2 : 0);
    need[i] = r;
    return r;
  }

  // park_mat: the matrix the CURRENTLY running CLV will meet at its parent if it
  // has to be parked while this subtree is evaluated
  void emit(unsigned i, bool live, unsigned park_mat) {
    const rdamd_operation_t &o = ops[i];
    const bool i1 = is_inner(o.child1_clv_index), i2 = is_inner(o.child2_clv_index);
    FusedOp f;
    memset(&f, 0, sizeof(f));
    unsigned matM = 0, matX = 0, matY = 0, kind = 0, spill = 0, tipX_row = 0, tipY_row = 0;
    if (!i1 && !i2) {
      kind = kFusedTT;
      spill = live ? 1 : 0;
      tipX_row = o.child1_clv_index; matX = o.child1_matrix_index;
      tipY_row = o.child2_clv_index; matY = o.child2_matrix_index;
      if (live) {
        matM = park_mat;              // pre-multiply the parked CLV
        if (depth == 0) spill |= 2;   // level 0 is a register slot in the kernel
        else if (depth == 1 && reg_levels >= 2) spill |= 8;   // ... and level 1 in programs compiled for two
        ++depth;
        max_depth = std::max(max_depth, depth);
        if (split_park) {
          FusedOp park;
          memset(&park, 0, sizeof(park));
          park.pM = matM * unit;
          park.flags = kFusedPark | ((spill & 2) ? 0x200u : 0u);   // 0x200: into the register slot
          out.push_back(park);
          matM = 0;
          spill = 0;
        }
      }
    } else if (i1 != i2) {
      const bool first_inner = i1;
      emit(producer.at(first_inner ? o.child1_clv_index : o.child2_clv_index), live, park_mat);
      kind = kFusedRT;
      matM = first_inner ? o.child1_matrix_index : o.child2_matrix_index;
      tipY_row = first_inner ? o.child2_clv_index : o.child1_clv_index;
      matY = first_inner ? o.child2_matrix_index : o.child1_matrix_index;
    } else {
      const unsigned a = producer.at(o.child1_clv_index), b = producer.at(o.child2_clv_index);
      const bool a_first = need[a] >= need[b];
      const unsigned first = a_first ? a : b, second = a_first ? b : a;
      const unsigned mat_first = a_first ? o.child1_matrix_index : o.child2_matrix_index;
      emit(first, live, park_mat);    // parked (times mat_first) by the first TT op of `second`
      emit(second, true, mat_first);
      kind = kFusedRP;                // running CLV = second; popped = mat_first . first
      matM = a_first ? o.child2_matrix_index : o.child1_matrix_index;
      --depth;
      if (depth == 0) spill |= 4;     // the popped sibling sits in the register slot
      else if (depth == 1 && reg_levels >= 2) spill |= 16;
    }
    f.pM = matM * unit;
    f.tX = matX * unit;
    f.tY = matY * unit;
    // 20 states: the tip tables' byte offsets ([matrix][rate 0], 12288 B per (matrix, rate))
    f.pad[0] = matX * rate_cats * (kFused20TabDoubles * 8u);
    f.pad[1] = matY * rate_cats * (kFused20TabDoubles * 8u);
    f.cX = tipX_row * tip_stride;
    f.cY = tipY_row * tip_stride;
    f.flags = kind | (spill << 8);   // (a 20-state TT never parks: its spill bits were moved to the park step)
    out.push_back(f);
  }
};

}  // namespace rdamd

using namespace rdamd;

extern "C" {

rdamd_schedule_t *rdamd_schedule_create(rdamd_partition_t *p, const rdamd_operation_t *ops,
                                        unsigned int n_ops,
                                        const unsigned int *matrix_indices,
                                        const double *branch_lengths,
                                        unsigned int n_matrices) {
  clear_error();
  const bool k20 = p->states == 20 && p->rate_cats <= 4;
  if (p->states != 4 && !k20) {
    set_error(40, "rdamd_schedule_create: the fused evaluator handles 4-state data and 20-state "
                  "data with up to 4 rate categories; use rdamd_update_clvs for %u states, %u "
                  "categories", p->states, p->rate_cats);
    return nullptr;
  }
  if (n_ops == 0 || (size_t)p->tips * p->tip_stride() > 0xffffffffu ||
      (size_t)p->prob_matrices * p->rate_cats * (k20 ? 12288 : 512) > 0xffffffffu) {
    set_error(41, "rdamd_schedule_create: empty operation list, or partition too large for "
                  "32-bit offsets (tips*sites or matrices*rates*512 >= 4 GiB)");
    return nullptr;
  }
  Compiler c;
  c.ops = ops; c.n_ops = n_ops; c.tips = p->tips; c.sites = p->sites; c.tip_stride = p->tip_stride(); c.rate_cats = p->rate_cats;
  c.unit = p->rate_cats * (k20 ? 3200u : 128u);
  c.split_park = k20;
  const unsigned nclv = p->tips + p->clv_buffers;
  for (unsigned i = 0; i < n_ops; ++i) {
    const rdamd_operation_t &o = ops[i];
    bool bad = o.parent_clv_index < p->tips || o.parent_clv_index >= nclv ||
               o.child1_clv_index >= nclv || o.child2_clv_index >= nclv ||
               o.child1_matrix_index >= p->prob_matrices ||
               o.child2_matrix_index >= p->prob_matrices;
    for (unsigned ch : {o.child1_clv_index, o.child2_clv_index})
      if (ch >= p->tips && !c.producer.count(ch)) bad = true;   // not yet computed
    if (c.producer.count(o.parent_clv_index)) bad = true;       // written twice
    if (bad) {
      set_error(42, "rdamd_schedule_create: operation %u is not part of a valid post-order "
                    "traversal (the fused evaluator needs the full schedule of "
                    "generate_operations)", i);
      return nullptr;
    }
    c.producer[o.parent_clv_index] = i;
  }
  c.need.assign(n_ops, 0);
  c.compute_need(n_ops - 1);
  c.out.reserve(n_ops);
  c.emit(n_ops - 1, false, 0);
  // 4 states: a program that would need three or more LDS stack levels is compiled for
  // TWO register levels instead (kernels_fused.hip: the LDS saved buys more resident
  // waves than the 18 extra registers cost)
  if (!k20 && c.max_depth >= 4) {
    c.out.clear();
    c.depth = c.max_depth = 0;
    c.reg_levels = 2;
    c.emit(n_ops - 1, false, 0);
  }
  size_t n_steps = c.out.size(), n_real = 0;
  for (const FusedOp &f : c.out)
    if (!c.split_park || (f.flags & 3u) != kFusedPark) ++n_real;
  if (n_real != n_ops) {
    set_error(43, "rdamd_schedule_create: %u of %u operations are not reachable from the "
                  "root operation", (unsigned)(n_ops - n_real), n_ops);
    return nullptr;
  }
  std::vector<double> brlen(p->prob_matrices, 0.0);
  for (unsigned m = 0; m < n_matrices; ++m) {
    if (matrix_indices[m] >= p->prob_matrices || !(branch_lengths[m] >= 0.0) ||
        !std::isfinite(branch_lengths[m])) {
      set_error(9, "rdamd_schedule_create: invalid branch (matrix %u, length %g)",
                matrix_indices[m], branch_lengths[m]);
      return nullptr;
    }
    brlen[matrix_indices[m]] = branch_lengths[m];
  }
  rdamd_schedule *s = new rdamd_schedule();
  // LDS levels = stack depth minus the register level (at least one is allocated)
  // (20 states: parking steps count as steps)
  s->part = p; s->n_ops = (unsigned)n_steps;
  s->depth = std::max(1u, c.max_depth > c.reg_levels ? c.max_depth - c.reg_levels : 0);
  s->reg_levels = c.reg_levels;
  n_ops = (unsigned)n_steps;
  s->prog = c.out;
  // harmless tail entries: the kernel prefetches descriptors up to i + 3
  for (int k = 0; k < 4; ++k) c.out.push_back(c.out.back());
#define TRY(expr) RDAMD_HIP_TRY(expr, (rdamd_schedule_destroy(s), nullptr))
  TRY(hipMalloc((void **)&s->d_prog, sizeof(FusedOp) * (n_ops + 4)));
  TRY(hipMalloc((void **)&s->d_brlen, sizeof(double) * p->prob_matrices));
  TRY(hipMemcpy(s->d_prog, c.out.data(), sizeof(FusedOp) * (n_ops + 4), hipMemcpyHostToDevice));
  TRY(hipMemcpy(s->d_brlen, brlen.data(), sizeof(double) * p->prob_matrices, hipMemcpyHostToDevice));
#undef TRY
  return s;
}

void rdamd_schedule_destroy(rdamd_schedule_t *s) {
  if (!s) return;
  if (s->part && s->part->stream) (void)hipStreamSynchronize(s->part->stream);
  if (s->d_prog) (void)hipFree(s->d_prog);
  if (s->d_brlen) (void)hipFree(s->d_brlen);
  delete s;
}

unsigned int rdamd_schedule_stack_depth(const rdamd_schedule_t *s) { return s->depth; }

static int evaluate_batch_impl(rdamd_partition_t *p, unsigned int n_jobs,
                               const rdamd_schedule_t *const *schedules,
                               const double *subst, const double *freqs,
                               const double *rates, const double *rate_weights,
                               double *lnl_host, void *lnl_device) {
  clear_error();
  if (n_jobs == 0) return RDAMD_SUCCESS;
  const bool k20 = p->states == 20 && p->rate_cats <= 4;
  if (p->states != 4 && !k20) {
    set_error(40, "rdamd_evaluate_batch: 4-state data, or 20-state data with up to 4 rate categories");
    return RDAMD_FAILURE;
  }
  const unsigned R = p->rate_cats, K = p->states, NP = K * K - K;
  if (p->sites == 0) {   // an empty alignment has likelihood 1
    if (lnl_host) std::fill(lnl_host, lnl_host + n_jobs, 0.0);
    if (lnl_device)
      RDAMD_HIP_TRY(hipMemset(lnl_device, 0, sizeof(double) * n_jobs), RDAMD_FAILURE);
    return RDAMD_SUCCESS;
  }
  RDAMD_HIP_TRY(ensure_workspace(p, n_jobs), RDAMD_FAILURE);
  FusedWorkspace *w = p->fused;
  // the pinned input block must not be rewritten while an earlier batch's
  // copies are still in flight
  RDAMD_HIP_TRY(hipStreamSynchronize(p->stream), RDAMD_FAILURE);
  char *h = w->h_in;
  FusedJob *hj = (FusedJob *)h;          h += sizeof(FusedJob) * n_jobs;
  double *hq = (double *)h;              h += sizeof(double) * K * K * n_jobs;
  double *hf = (double *)h;              h += sizeof(double) * K * n_jobs;
  double *hr = (double *)h;              h += sizeof(double) * R * n_jobs;
  double *hw = (double *)h;
  unsigned max_depth = 1, reg_levels = 1;
  for (unsigned j = 0; j < n_jobs; ++j) {
    const rdamd_schedule_t *s = schedules[j];
    if (!s || s->part != p) {
      set_error(44, "rdamd_evaluate_batch: job %u has no schedule of this partition", j);
      return RDAMD_FAILURE;
    }
    hj[j].prog = s->d_prog; hj[j].brlen = s->d_brlen; hj[j].n_ops = s->n_ops;
    hj[j].depth = 0;   // patched below: every block uses the launch-wide depth
    hj[j].tt_unsafe = 0; hj[j].pad = 0;   // (set again by the P-matrix step of this batch)
    max_depth = std::max(max_depth, s->depth);
    reg_levels = std::max(reg_levels, s->reg_levels);
    double wide_s[12] = {0}, wide_f[4] = {0};
    const double *sj = subst + (size_t)j * NP, *fj = freqs + (size_t)j * K;
    if (p->embedded()) {   // caller passes [n][2] / [n][2]: into the 4-state shapes
      wide_s[0] = subst[(size_t)j * 2];
      wide_s[3] = subst[(size_t)j * 2 + 1];
      wide_f[0] = freqs[(size_t)j * 2];
      wide_f[1] = freqs[(size_t)j * 2 + 1];
      sj = wide_s;
      fj = wide_f;
    }
    build_q_host(K, sj, fj, hq + (size_t)j * K * K);
    for (unsigned k = 0; k < K; ++k) hf[(size_t)j * K + k] = fj[k];
    for (unsigned r = 0; r < R; ++r) {
      hr[(size_t)j * R + r] = rates ? rates[(size_t)j * R + r] : p->rates[r];
      hw[(size_t)j * R + r] = rate_weights ? rate_weights[(size_t)j * R + r] : p->rate_weights[r];
    }
  }
  for (unsigned j = 0; j < n_jobs; ++j) hj[j].depth = max_depth;
  {   // the device block mirrors the staging block: one copy
    const size_t in_bytes = (size_t)n_jobs * (sizeof(FusedJob) + sizeof(double) * (K * K + K + 2 * R));
    RDAMD_HIP_TRY(hipMemcpyAsync(w->d_in, w->h_in, in_bytes, hipMemcpyHostToDevice, p->stream), RDAMD_FAILURE);
    w->d_jobs = (FusedJob *)(w->d_in + ((char *)hj - w->h_in));
    w->d_q = (double *)(w->d_in + ((char *)hq - w->h_in));
    w->d_freqs = (double *)(w->d_in + ((char *)hf - w->h_in));
    w->d_rates = (double *)(w->d_in + ((char *)hr - w->h_in));
    w->d_rw = (double *)(w->d_in + ((char *)hw - w->h_in));
  }

  double *d_out = lnl_device ? (double *)lnl_device : w->d_out;
  hipError_t e;
  if (k20) {
    Fused20Args b;
    b.jobs = w->d_jobs; b.tipcodes = p->d_tipcodes; b.tip_stride = p->tip_stride();
    b.codemask = p->d_codemask; b.pattern_weights = p->d_pattern_weights;
    b.pmat = w->d_pmat; b.tiptab = w->d_tiptab; b.ncodes = p->ncodes;
    b.tiptab_job_stride = (size_t)p->prob_matrices * R * kFused20TabDoubles;
    b.freqs = w->d_freqs; b.rate_weights = w->d_rw; b.partials = w->d_partials;
    b.pmat_job_stride = (size_t)p->prob_matrices * R * K * K;
    b.sites = p->sites; b.rate_cats = R; b.tiles = w->blocks_x;
    p->prof_begin(4);
    e = launch_fused20_pmatrix(b, w->d_q, w->d_rates, n_jobs, p->prob_matrices, p->stream);
    p->prof_end();
    RDAMD_HIP_TRY(e, RDAMD_FAILURE);
    p->prof_begin(3);
    e = launch_fused20_eval(b, n_jobs, max_depth, d_out, p->stream);
    p->prof_end();
    RDAMD_HIP_TRY(e, RDAMD_FAILURE);
  } else {
  FusedArgs a;
  a.jobs = w->d_jobs; a.tipcodes = p->d_tipcodes16; a.pattern_weights = p->d_pattern_weights;
  a.pmat = w->d_pmat; a.tiptab = w->d_tiptab; a.freqs = w->d_freqs; a.rate_weights = w->d_rw;
  a.partials = w->d_partials; a.persite = nullptr;
  a.pmat_job_stride = (size_t)p->prob_matrices * R * 16;
  a.sites = p->sites; a.rate_cats = R;
  a.tipcodes_bytes = (unsigned)std::min<size_t>((size_t)p->tips * p->tip_stride(), 0xffffffffu);
  p->prof_begin(4);
  e = launch_fused_pmatrix(a, w->d_q, w->d_rates, n_jobs, p->prob_matrices, p->stream);
  p->prof_end();
  RDAMD_HIP_TRY(e, RDAMD_FAILURE);
  p->prof_begin(3);
  // two sites per lane (kernels_fused.hip) when half the waves still fill the chip:
  // >= 4 waves for each of the 1024 SIMDs.
  unsigned ns = (size_t)n_jobs * w->blocks_x >= 8192 ? 2u : 1u;
#ifdef RDAMD_ABLATION   // A/B runs only (`make ablation`): RDAMD_FUSED_NS=1|2 overrides
  static const int force_ns = getenv("RDAMD_FUSED_NS") ? atoi(getenv("RDAMD_FUSED_NS")) : 0;
  if (force_ns) ns = (unsigned)force_ns;
#endif
  e = launch_fused_eval(a, n_jobs, max_depth, w->blocks_x, ns, reg_levels, d_out, p->stream);
  p->prof_end();
  RDAMD_HIP_TRY(e, RDAMD_FAILURE);
  }
  if (lnl_host) {
    RDAMD_HIP_TRY(hipMemcpyAsync(w->h_out, d_out, sizeof(double) * n_jobs, hipMemcpyDeviceToHost, p->stream), RDAMD_FAILURE);
    RDAMD_HIP_TRY(hipStreamSynchronize(p->stream), RDAMD_FAILURE);
    memcpy(lnl_host, w->h_out, sizeof(double) * n_jobs);
  } else {
    RDAMD_HIP_TRY(hipStreamSynchronize(p->stream), RDAMD_FAILURE);
  }
  return RDAMD_SUCCESS;
}

int rdamd_evaluate_batch(rdamd_partition_t *p, unsigned int n_jobs,
                         const rdamd_schedule_t *const *schedules, const double *subst,
                         const double *freqs, const double *rates,
                         const double *rate_weights, double *lnl_out) {
  return evaluate_batch_impl(p, n_jobs, schedules, subst, freqs, rates, rate_weights, lnl_out, nullptr);
}

int rdamd_evaluate_batch_device(rdamd_partition_t *p, unsigned int n_jobs,
                                const rdamd_schedule_t *const *schedules,
                                const double *subst, const double *freqs,
                                const double *rates, const double *rate_weights,
                                void *d_lnl_out) {
  return evaluate_batch_impl(p, n_jobs, schedules, subst, freqs, rates, rate_weights, nullptr, d_lnl_out);
}

}  // extern "C"
