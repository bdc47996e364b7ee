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
#include "clades.hpp"
#include "common.hpp"
#include "fused.hpp"
#include "traversal_compiler.hpp"

struct rdamd_schedule {
  rdamd_partition *part = nullptr;
  // everything of the schedule that lives on the device is ONE block out of the partition's
  // pool (schedule_block_alloc): [program | plain program | clade steps | clade groups | lengths]
  char *d_block = nullptr;
  size_t block_bytes = 0;
  rdamd::FusedOp *d_prog = nullptr;
  double *d_brlen = nullptr;
  unsigned n_ops = 0, depth = 0;      // depth: LDS stack levels the program needs
  unsigned reg_levels = 1;            // register stack levels it was compiled for
  std::vector<rdamd::FusedOp> prog;   // host copy (tests / debugging)
  // subtree site repeats (clades.hpp): d_prog is then the program WITH pseudo-tips and the
  // plain program of the caller's full operation list is kept beside it (it runs when a
  // launch's tables make the pseudo-tips' missing rescale counts matter, fused.hpp)
  rdamd::FusedOp *d_prog_plain = nullptr;     // == d_prog without pseudo-tips
  unsigned n_ops_plain = 0, depth_plain = 0, reg_levels_plain = 1;
  unsigned lds_pos = 0, lds_pos_plain = 0;    // FusedJob::lds_pos of the two programs
  rdamd::CladeStep *d_steps = nullptr;
  rdamd::CladeGroup *d_groups = nullptr;
  uint32_t *d_tipmask = nullptr;              // 20 states: bit m = P-matrix m belongs to a branch that ends in a tip
  unsigned n_steps = 0, n_groups = 0;
  unsigned tip_generation = 0;
  unsigned table_rows = 16;           // 16: 8-bit code arena, 16-row tables only; 64: 16-bit arena, 64-row tables too
  unsigned n_wide = 0;                // 64-row tables per job
  // what one (site, rate) executes: operations and matrix-vector products per traversal
  unsigned matvecs = 0, matvecs_plain = 0, clade_rows = 0;
  // the root operation's children (rdamd_evaluate_root_children): clv and scaler indices as the caller named them
  unsigned root_child_clv[2] = {0, 0};
  int root_child_sc[2] = {-1, -1};
};

namespace rdamd {

// doubles in front of the tables: the 64-row evaluator's buffer descriptor starts kFusedDmaBias
// bytes before a job's tables (kernels_fused.hip, RDAMD_LOAD_TABS64)
constexpr size_t kTiptabPad = kFusedDmaBias / 8;
// tips up to which a partition speculates by default (include/root_digger_amd.h, rdamd_partition_set_rescale_speculation)
constexpr unsigned kSpeculateTips = 256;

struct FusedWorkspace {
  unsigned cap_jobs = 0, blocks_x = 0;
  char *d_in = nullptr;   // per-batch inputs (see ensure_workspace)
  unsigned *d_any_unsafe = nullptr;   // inside d_in, behind the batch's inputs
  FusedJob *d_jobs = nullptr;   // ... and where this batch's pieces sit inside it
  double *d_q = nullptr, *d_rates = nullptr, *d_freqs = nullptr, *d_rw = nullptr;
  double *d_qpow20 = nullptr;   // 20 states: the powers of every job's Q (kernels_fused_k20.hip)
  double *d_pmat = nullptr, *d_tiptab = nullptr, *d_partials = nullptr, *d_out = nullptr;
  double *d_clade_scratch = nullptr;   // nested clade tables of a launch: [job][step][rate][rows][4]
  size_t clade_scratch_doubles = 0;
  unsigned *d_export_cnt = nullptr;    // rdamd_evaluate_root_children: rescale counts [2 children][site][rate]
  size_t tiptab_doubles = 0;           // allocated behind d_tiptab
  double *h_out = nullptr;   // pinned
  char *h_in = nullptr;      // pinned parameter staging
  size_t h_in_bytes = 0;
  hipEvent_t ev_ready = nullptr, ev_done = nullptr;   // pipelined batches (batch_submit / batch_wait)
  // the batch this slot has in flight: what batch_wait needs for the results and, should the
  // batch have raised a flag, for the second pass
  struct Pending {
    bool active = false, k20 = false, host_out = false;
    unsigned n_jobs = 0, ns = 1, max_depth[2] = {1, 1}, reg_levels[2] = {1, 1};
    FusedArgs a;
    double *d_out = nullptr;
    double *mirror = nullptr;   // rdamd_evaluate_batch_submit_device: the caller's device copy, [n_jobs] + the flag
    uint64_t seq = 0;
  } pend;
};

void fused_workspace_free(FusedWorkspace *w) {
  if (!w) return;
  void *dev[] = {w->d_in, w->d_pmat, w->d_tiptab, w->d_partials, w->d_out, w->d_clade_scratch, w->d_export_cnt, w->d_qpow20};
  for (void *d : dev)
    if (d) (void)hipFree(d);
  if (w->h_out) (void)hipHostFree(w->h_out);
  if (w->h_in) (void)hipHostFree(w->h_in);
  if (w->ev_ready) (void)hipEventDestroy(w->ev_ready);
  if (w->ev_done) (void)hipEventDestroy(w->ev_done);
  delete w;
}

// ---- device blocks of schedules -------------------------------------------------------------
// A lock-stepped search compiles and drops a schedule per candidate and round while batches
// are in flight; hipMalloc / hipFree there would wait for the whole device.  Blocks of dropped
// schedules are parked until the batches that may still read them have finished
// (rdamd_partition::batch_completed) and handed to the next schedule that fits.
static char *schedule_block_alloc(rdamd_partition *p, size_t bytes, size_t *got) {
  const uint64_t done = p->batch_completed.load(std::memory_order_acquire);
  for (size_t i = 0; i < p->sched_retired.size();) {
    if (p->sched_retired[i].seq <= done) {
      p->sched_pool.push_back(p->sched_retired[i]);
      p->sched_retired[i] = p->sched_retired.back();
      p->sched_retired.pop_back();
    } else {
      ++i;
    }
  }
  size_t best = p->sched_pool.size();
  for (size_t i = 0; i < p->sched_pool.size(); ++i)
    if (p->sched_pool[i].bytes >= bytes && p->sched_pool[i].bytes <= 2 * bytes + 4096 &&
        (best == p->sched_pool.size() || p->sched_pool[i].bytes < p->sched_pool[best].bytes))
      best = i;
  if (best < p->sched_pool.size()) {
    char *ptr = p->sched_pool[best].ptr;
    *got = p->sched_pool[best].bytes;
    p->sched_pool[best] = p->sched_pool.back();
    p->sched_pool.pop_back();
    return ptr;
  }
  const size_t rounded = (bytes + 4095) & ~(size_t)4095;
  char *ptr = nullptr;
  if (hipMalloc((void **)&ptr, rounded) != hipSuccess) return nullptr;
  *got = rounded;
  return ptr;
}
static void schedule_block_release(rdamd_partition *p, char *ptr, size_t bytes) {
  if (!ptr) return;
  // (batches in flight when the schedule is dropped may hold its programs)
  const uint64_t seq = p->batch_submitted;
  if (p->batch_completed.load(std::memory_order_acquire) >= seq) p->sched_pool.push_back({ptr, bytes, 0});
  else p->sched_retired.push_back({ptr, bytes, seq});
}

static hipError_t ensure_workspace(rdamd_partition *p, FusedWorkspace *&slot, unsigned n_jobs) {
  if (!slot) slot = new FusedWorkspace();
  FusedWorkspace *w = slot;
  if (n_jobs <= w->cap_jobs) return hipSuccess;
  hipError_t e = sync_streams(p);
  if (e != hipSuccess) return e;
  void *dev[] = {w->d_in, w->d_pmat, w->d_tiptab, w->d_partials, w->d_out, w->d_clade_scratch, w->d_export_cnt, w->d_qpow20};
  for (void *d : dev)
    if (d) (void)hipFree(d);
  if (w->h_out) (void)hipHostFree(w->h_out);
  if (w->h_in) (void)hipHostFree(w->h_in);
  {
    hipEvent_t r = w->ev_ready, d = w->ev_done;   // (the events survive a larger workspace)
    *w = FusedWorkspace();
    w->ev_ready = r; w->ev_done = d;
  }
  // (a model replica only ever runs one-job batches -- rdamd_evaluate_root_children -- and there
  // may be 32 of them on a device: no 16-job floor)
  const unsigned cap = std::max(2u, n_jobs + n_jobs / 2);
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
  // (+ 8 bytes: the batch's any-unsafe word, zeroed by the same copy, FusedArgs::any_unsafe)
  A(w->d_in, (size_t)cap * (sizeof(FusedJob) + sizeof(double) * (K * K + K + 2 * R)) + 8);
  A(w->d_pmat, sizeof(double) * pm_per_job * cap);
  if (K == 20) A(w->d_qpow20, sizeof(double) * fused20_qpow_doubles() * cap);
  w->tiptab_doubles = (K == 4 ? pm_per_job * 4 : (size_t)p->prob_matrices * R * kFused20TabDoubles) * cap;
  A(w->d_tiptab, sizeof(double) * (w->tiptab_doubles + kTiptabPad));
  A(w->d_partials, sizeof(double) * w->blocks_x * cap);
  A(w->d_out, sizeof(double) * cap);
#undef A
  e = hipHostMalloc((void **)&w->h_out, sizeof(double) * (cap + 1), hipHostMallocDefault);   // (+ the any-unsafe word)
  if (e != hipSuccess) return e;
  w->h_in_bytes = (size_t)cap * (sizeof(FusedJob) + sizeof(double) * (K * K + K + 2 * R)) + 8;
  e = hipHostMalloc((void **)&w->h_in, w->h_in_bytes, hipHostMallocDefault);
  if (e != hipSuccess) return e;
  w->cap_jobs = cap;
  return hipSuccess;
}

}  // namespace rdamd

using namespace rdamd;

// (the partition's launch_mu is held: the block goes back to the pool, nothing waits)
static void rdamd_schedule_destroy_locked(rdamd_schedule *s) {
  if (!s) return;
  if (s->part) schedule_block_release(s->part, s->d_block, s->block_bytes);
  delete s;
}

// allow_repeats = false: the plain program only, 16-row tables, nothing of the clade cache is
// touched (rdamd_evaluate_root_children: a schedule that lives for one launch)
static rdamd_schedule_t *schedule_create_impl(rdamd_partition_t *p, const rdamd_operation_t *ops,
                                              unsigned int n_ops,
                                              const unsigned int *matrix_indices,
                                              const double *branch_lengths,
                                              unsigned int n_matrices, bool allow_repeats) {
  // (clade cache, code arenas and the block pool belong to the partition: one thread at a time)
  std::lock_guard<std::mutex> guard(p->launch_mu);
  const bool k20 = p->states == 20 && p->rate_cats <= 8;
  if (p->states != 4 && !k20) {
    set_error(40, "rdamd_schedule_create: the fused evaluator handles 4-state data and 20-state "
                  "data with up to 8 rate categories; use rdamd_update_clvs for %u states, %u "
                  "categories", p->states, p->rate_cats);
    return nullptr;
  }
  if (n_ops == 0 || (size_t)p->tips * p->tip_stride() > 0xffffffffu ||
      (size_t)p->prob_matrices * p->rate_cats * (k20 ? 12288 : 512) > 0xffffffffu) {
    set_error(41, "rdamd_schedule_create: empty operation list, or partition too large for "
                  "32-bit offsets (tips*sites or matrices*rates*512 >= 4 GiB)");
    return nullptr;
  }
  // ---- validation: a full post-order traversal ------------------------------------------
  const unsigned nclv = p->tips + p->clv_buffers;
  std::unordered_map<unsigned, unsigned> producer;   // clv -> op index
  std::vector<int> consumer(n_ops, -1);              // op -> the op that takes its result
  for (unsigned i = 0; i < n_ops; ++i) {
    const rdamd_operation_t &o = ops[i];
    bool bad = o.parent_clv_index < p->tips || o.parent_clv_index >= nclv ||
               o.child1_clv_index >= nclv || o.child2_clv_index >= nclv ||
               o.child1_matrix_index >= p->prob_matrices ||
               o.child2_matrix_index >= p->prob_matrices;
    for (unsigned ch : {o.child1_clv_index, o.child2_clv_index}) {
      if (ch < p->tips) continue;
      auto it = producer.find(ch);
      if (it == producer.end() || consumer[it->second] >= 0) bad = true;   // not yet computed / used twice
      else consumer[it->second] = (int)i;
    }
    if (producer.count(o.parent_clv_index)) bad = true;       // written twice
    if (bad) {
      set_error(42, "rdamd_schedule_create: operation %u is not part of a valid post-order "
                    "traversal (the fused evaluator needs the full schedule of "
                    "generate_operations)", i);
      return nullptr;
    }
    producer[o.parent_clv_index] = i;
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

  // ---- one program from one operation list -------------------------------------------------
  struct Program {
    std::vector<FusedOp> steps;
    unsigned depth = 1, reg_levels = 1, matvecs = 0;
    unsigned lds_pos = 0;   // 4 states, stacks with private-segment levels: which in-memory entry sits in LDS
  };
  // 64-row tables (and the 16-bit code arena that goes with them) when the partition's class
  // limit asks for them; every program of the schedule then addresses that arena
  // A pseudo-tip's table is written into the tip-table slot of the branch above it
  // (kernels_clade.hip): that slot is only free when the branch's matrix index is used by this
  // one child.  The C ABI (like coraxlib's) lets a caller share a matrix index between
  // branches; such a list is evaluated without folding -- its tip tables are all code-indexed.
  bool matrix_shared = false;
  {
    std::vector<unsigned char> uses(p->prob_matrices, 0);
    for (unsigned i = 0; i < n_ops; ++i)
      for (unsigned m : {ops[i].child1_matrix_index, ops[i].child2_matrix_index})
        if (uses[m]++) matrix_shared = true;
  }
  const bool repeats = allow_repeats && !k20 && (p->attributes & RDAMD_ATTRIB_SITE_REPEATS) && p->sites > 0;
  if (repeats && !p->clades) p->clades = new CladeCache();
  // (the 16-bit arena needs twice the bytes per row: a partition whose tip rows alone would not
  // fit 32-bit offsets there keeps the 8-bit arena and 16-row tables -- for every schedule, so
  // that all of them can share a launch)
  const bool wide_fits = ((size_t)p->tips + 16) * p->tip_stride() * 2 + kTipcodePad <= 0xffffffffull;
  const bool wide_mode = repeats && p->clades->max_classes > 16 && wide_fits;
  const unsigned class_limit = wide_mode ? 64u : 16u;   // what this schedule's table slots hold
  auto compile = [&](const std::vector<rdamd_operation_t> &list,
                     const std::unordered_map<unsigned, unsigned> &pseudo_row,
                     const std::unordered_map<unsigned, unsigned> &pseudo_wide, Program &out) -> bool {
    Compiler c;
    c.ops = list.data(); c.n_ops = (unsigned)list.size(); c.tips = p->tips; c.sites = p->sites;
    c.tip_stride = p->tip_stride() * (wide_mode ? 2u : 1u); c.rate_cats = p->rate_cats;
    c.unit = p->rate_cats * (k20 ? 3200u : 128u);
    c.split_park = k20;
    c.pseudo_row = pseudo_row;
    c.pseudo_wide = pseudo_wide;
    c.wide_base = 8u * p->prob_matrices * p->rate_cats * 16u;
    c.dma_offsets = !k20 && wide_mode;
    c.place_parks = !k20 && wide_mode;   // (the kernels of these programs: one register slot, one LDS slot, a private-segment stack)
    {   // the steps that compute the root operation's inner children (fused.hpp, 0x8000 / 0x10000)
      const rdamd_operation_t &root = list.back();
      if (root.child1_clv_index >= p->tips) c.mark_clv[0] = root.child1_clv_index;
      if (root.child2_clv_index >= p->tips) c.mark_clv[1] = root.child2_clv_index;
    }
    for (unsigned i = 0; i < c.n_ops; ++i) c.producer[list[i].parent_clv_index] = i;
    c.need.assign(c.n_ops, 0);
    c.compute_need(c.n_ops - 1);
    c.out.reserve(c.n_ops);
    c.emit(c.n_ops - 1, false, 0);
    // second pass: the register slot to the busiest stack level, the LDS slot to the runner-up
    // (traversal_compiler.hpp)
    // (two register levels: from 8 in-memory entries on where the kernel has private-segment
    // levels, i.e. 64-row table slots; from 3 on an all-LDS stack -- kernels_fused.hip)
    const unsigned lds_pos = c.place_levels(k20 ? 0u : (wide_mode ? 1u + kFusedSpillLevels : 3u), kFusedSpillLevels - 1u);
    size_t n_real = 0;
    for (const FusedOp &f : c.out)
      if (!c.split_park || (f.flags & 3u) != kFusedPark) ++n_real;
    if (n_real != c.n_ops) {
      set_error(43, "rdamd_schedule_create: %u of %u operations are not reachable from the "
                    "root operation", (unsigned)(c.n_ops - n_real), c.n_ops);
      return false;
    }
    // LDS levels = stack depth minus the register levels (at least one is allocated)
    // (20 states: parking steps count as steps)
    // (parks placed one by one: the LDS slot + the private-segment entries)
    out.depth = !c.park_class.empty() ? 1u + c.mem_depth
                                      : std::max(1u, c.max_depth > c.reg_levels ? c.max_depth - c.reg_levels : 0);
    out.reg_levels = c.reg_levels;
    out.lds_pos = lds_pos;
    out.matvecs = c.matvecs;
    out.steps = std::move(c.out);
    return true;
  };

  const std::vector<rdamd_operation_t> all_ops(ops, ops + n_ops);
  Program plain, folded;
  if (!compile(all_ops, {}, {}, plain)) return nullptr;

  // ---- subtree site repeats: which clades become pseudo-tips (clades.hpp) -----------------
  // A node is SMALL when the sites fall into at most max_classes classes below it; small is
  // inherited downwards, so the small nodes form whole subtrees and the topmost small node
  // of each is the pseudo-tip.  The root operation is never folded (a program has >= 1 step).
  std::vector<char> small(n_ops, 0);
  std::vector<unsigned> node_id(n_ops, 0);
  std::vector<CladeStep> steps;
  std::vector<CladeGroup> groups;
  std::unordered_map<unsigned, unsigned> pseudo_row, pseudo_wide;
  unsigned clade_rows = 0, n_wide = 0;
  if (repeats) {
    auto id_of = [&](unsigned clv) { return clv < p->tips ? clv : node_id[producer.at(clv)]; };
    for (unsigned i = 0; i < n_ops; ++i) {
      const rdamd_operation_t &o = ops[i];
      node_id[i] = clade_intern(p, id_of(o.child1_clv_index), id_of(o.child2_clv_index),
                                o.child1_matrix_index, o.child2_matrix_index);
      const unsigned nc = clade_node(*p->clades, p->tips, node_id[i])->n_classes;
      // (a parent has at least as many classes as either child: small is inherited downwards
      // under any limit)
      small[i] = i + 1 < n_ops && !matrix_shared && nc > 0 && nc <= class_limit;
    }
    // the branch above operation i: the matrix index its consumer uses for it
    auto mat_above = [&](unsigned i) {
      const rdamd_operation_t &c = ops[consumer[i]];
      return c.child1_clv_index == ops[i].parent_clv_index ? c.child1_matrix_index : c.child2_matrix_index;
    };
    for (unsigned i = 0; i + 1 < n_ops; ++i) {
      if (!small[i] || small[consumer[i]]) continue;     // not a pseudo-tip
      CladeGroup g;
      g.first = (unsigned)steps.size();
      // post-order over the small subtree below i; local index = position inside the group
      std::function<unsigned(unsigned)> walk = [&](unsigned j) -> unsigned {
        const rdamd_operation_t &o = ops[j];
        CladeStep st;
        memset(&st, 0, sizeof st);
        const unsigned ch[2] = {o.child1_clv_index, o.child2_clv_index};
        const unsigned mt[2] = {o.child1_matrix_index, o.child2_matrix_index};
        for (int k = 0; k < 2; ++k)
          st.src[k] = ch[k] < p->tips ? mt[k] : (0x80000000u | walk(producer.at(ch[k])));
        const CladeNode *node = clade_node(*p->clades, p->tips, node_id[j]);
        st.n_classes = node->n_classes;
        st.out_mat = mat_above(j);
        st.last = j == i ? 1u : 0u;
        st.wide_slot = 0xffffffffu;
        if (j == i && node->n_classes > 16) {
          st.wide_slot = n_wide;
          pseudo_wide[o.parent_clv_index] = n_wide++;
        }
        st.pad = node_id[j];   // (host only: the map offset is filled in below)
        clade_rows += node->n_classes;
        steps.push_back(st);
        return (unsigned)steps.size() - 1 - g.first;
      };
      walk(i);
      g.count = (unsigned)steps.size() - g.first;
      groups.push_back(g);
    }
  }
  rdamd_schedule *s = new rdamd_schedule();
  s->part = p;
  s->tip_generation = p->tip_generation;
  s->root_child_clv[0] = ops[n_ops - 1].child1_clv_index; s->root_child_sc[0] = ops[n_ops - 1].child1_scaler_index;
  s->root_child_clv[1] = ops[n_ops - 1].child2_clv_index; s->root_child_sc[1] = ops[n_ops - 1].child2_scaler_index;
#define TRY(expr) RDAMD_HIP_TRY(expr, (rdamd_schedule_destroy_locked(s), nullptr))
  s->table_rows = wide_mode ? 64u : 16u;
  s->n_wide = n_wide;
  if (wide_mode) TRY(ensure_wide_arena(p));
  if (!groups.empty()) {
    for (CladeStep &st : steps) {
      TRY(clade_upload_map(p, st.pad));
      st.map_off = (uint32_t)clade_node(*p->clades, p->tips, st.pad)->map_off;
      st.pad = 0;
    }
    std::vector<rdamd_operation_t> kept;
    for (unsigned i = 0; i < n_ops; ++i) {
      if (!small[i]) { kept.push_back(ops[i]); continue; }
      if (small[consumer[i]]) continue;
      const hipError_t ce = clade_upload_codes(p, node_id[i], wide_mode);
      if (ce != hipSuccess && p->code_arena_full) {
        // no room for another row of class codes within 32-bit offsets: this schedule runs its
        // plain program (rows handed out earlier stay valid for the schedules that hold them)
        groups.clear(); steps.clear(); pseudo_row.clear(); pseudo_wide.clear();
        n_wide = 0; clade_rows = 0;
        break;
      }
      TRY(ce);
      pseudo_row[ops[i].parent_clv_index] = (unsigned)clade_node(*p->clades, p->tips, node_id[i])->code_row[wide_mode];
    }
    if (!groups.empty() && !compile(kept, pseudo_row, pseudo_wide, folded)) { rdamd_schedule_destroy_locked(s); return nullptr; }
  }
  s->n_wide = n_wide;
  const Program &main_prog = groups.empty() ? plain : folded;
  s->n_ops = (unsigned)main_prog.steps.size();
  s->depth = main_prog.depth; s->reg_levels = main_prog.reg_levels; s->matvecs = main_prog.matvecs;
  s->n_ops_plain = (unsigned)plain.steps.size();
  s->depth_plain = plain.depth; s->reg_levels_plain = plain.reg_levels; s->matvecs_plain = plain.matvecs;
  s->lds_pos = main_prog.lds_pos; s->lds_pos_plain = plain.lds_pos;
  s->n_steps = (unsigned)steps.size(); s->n_groups = (unsigned)groups.size(); s->clade_rows = clade_rows;
  s->prog = main_prog.steps;
  // ---- the device block: [program | plain program | clade steps | clade groups | lengths] ----
  {
    auto padded = [](const Program &pr) {
      std::vector<FusedOp> v = pr.steps;
      // harmless tail entries: the kernel prefetches descriptors up to i + 3
      for (int k = 0; k < 4; ++k) v.push_back(v.back());
      return v;
    };
    const std::vector<FusedOp> pm = padded(main_prog), pp = groups.empty() ? std::vector<FusedOp>() : padded(plain);
    auto up = [](size_t b) { return (b + 63) & ~(size_t)63; };
    const size_t o_prog = 0, o_plain = up(o_prog + sizeof(FusedOp) * pm.size()),
                 o_steps = up(o_plain + sizeof(FusedOp) * pp.size()),
                 o_groups = up(o_steps + sizeof(CladeStep) * steps.size()),
                 o_brlen = up(o_groups + sizeof(CladeGroup) * groups.size()),
                 o_tipmask = up(o_brlen + sizeof(double) * p->prob_matrices),
                 total = up(o_tipmask + (k20 ? sizeof(uint32_t) * ((p->prob_matrices + 31) / 32) : 0));
    std::vector<char> host(total, 0);
    memcpy(host.data() + o_prog, pm.data(), sizeof(FusedOp) * pm.size());
    if (!pp.empty()) memcpy(host.data() + o_plain, pp.data(), sizeof(FusedOp) * pp.size());
    if (!steps.empty()) memcpy(host.data() + o_steps, steps.data(), sizeof(CladeStep) * steps.size());
    if (!groups.empty()) memcpy(host.data() + o_groups, groups.data(), sizeof(CladeGroup) * groups.size());
    memcpy(host.data() + o_brlen, brlen.data(), sizeof(double) * p->prob_matrices);
    if (k20) {   // which branches end in a tip: only their tip tables are ever read (fused20_pmatrix_kernel)
      uint32_t *mask = (uint32_t *)(host.data() + o_tipmask);
      for (unsigned i = 0; i < n_ops; ++i) {
        if (ops[i].child1_clv_index < p->tips) mask[ops[i].child1_matrix_index >> 5] |= 1u << (ops[i].child1_matrix_index & 31u);
        if (ops[i].child2_clv_index < p->tips) mask[ops[i].child2_matrix_index >> 5] |= 1u << (ops[i].child2_matrix_index & 31u);
      }
    }
    s->d_block = schedule_block_alloc(p, total, &s->block_bytes);
    if (!s->d_block) {
      set_error(100 + (int)hipErrorOutOfMemory, "rdamd_schedule_create: no device memory for a %zu-byte schedule", total);
      rdamd_schedule_destroy_locked(s);
      return nullptr;
    }
    TRY(hipMemcpy(s->d_block, host.data(), total, hipMemcpyHostToDevice));
    s->d_prog = (FusedOp *)(s->d_block + o_prog);
    s->d_prog_plain = groups.empty() ? s->d_prog : (FusedOp *)(s->d_block + o_plain);
    s->d_steps = groups.empty() ? nullptr : (CladeStep *)(s->d_block + o_steps);
    s->d_groups = groups.empty() ? nullptr : (CladeGroup *)(s->d_block + o_groups);
    s->d_brlen = (double *)(s->d_block + o_brlen);
    s->d_tipmask = k20 ? (uint32_t *)(s->d_block + o_tipmask) : nullptr;
  }
#undef TRY
  return s;
}

extern "C" {

rdamd_schedule_t *rdamd_schedule_create(rdamd_partition_t *p, const rdamd_operation_t *ops,
                                        unsigned int n_ops,
                                        const unsigned int *matrix_indices,
                                        const double *branch_lengths,
                                        unsigned int n_matrices) {
  clear_error();
  return schedule_create_impl(p, ops, n_ops, matrix_indices, branch_lengths, n_matrices, true);
}

void rdamd_schedule_destroy(rdamd_schedule_t *s) {
  if (!s) return;
  if (s->part) {
    std::lock_guard<std::mutex> g(s->part->launch_mu);
    rdamd_schedule_destroy_locked(s);
  } else {
    delete s;
  }
}

unsigned int rdamd_schedule_stack_depth(const rdamd_schedule_t *s) { return s->depth; }

int rdamd_schedule_stats(const rdamd_schedule_t *s, rdamd_schedule_stats_t *out) {
  if (!s || !out) return RDAMD_FAILURE;
  const bool k20 = s->part->states == 20;
  unsigned parks = 0, parks_plain = 0;   // (20-state programs count parking as steps of their own)
  if (k20)
    for (const rdamd::FusedOp &f : s->prog) parks += (f.flags & 3u) == rdamd::kFusedPark;
  parks_plain = parks;
  out->operations = s->n_ops_plain - parks_plain;
  out->steps = s->n_ops - parks;
  out->matvecs = s->matvecs;
  out->matvecs_plain = s->matvecs_plain;
  out->pseudo_tips = s->n_groups;
  out->clade_nodes = s->n_steps;
  out->clade_rows = s->clade_rows;
  out->stack_depth = s->depth;
  out->stack_depth_plain = s->depth_plain;
  out->parks = out->parks_in_registers = out->parks_in_lds_slot = 0;
  for (const rdamd::FusedOp &f : s->prog) {
    const bool park = k20 ? (f.flags & 3u) == rdamd::kFusedPark : (f.flags & 0x100u) != 0u;
    out->parks += park;
    out->parks_in_registers += park && (f.flags & 0xa00u) != 0u;
    out->parks_in_lds_slot += park && (f.flags & 0x20000u) != 0u;
  }
  return RDAMD_SUCCESS;
}

unsigned int rdamd_partition_site_repeats(const rdamd_partition_t *p) {
  if (p->states != 4 || !(p->attributes & RDAMD_ATTRIB_SITE_REPEATS)) return 0;
  return p->clades ? p->clades->max_classes : rdamd::CladeCache().max_classes;
}

int rdamd_partition_set_site_repeats(rdamd_partition_t *p, unsigned int max_classes) {
  clear_error();
  std::lock_guard<std::mutex> guard(p->launch_mu);
  if (max_classes > 64) {
    set_error(46, "rdamd_partition_set_site_repeats: at most 64 classes per pseudo-tip (got %u)", max_classes);
    return RDAMD_FAILURE;
  }
  if (p->states != 4) return RDAMD_SUCCESS;
  if (max_classes == 0) {
    p->attributes &= ~RDAMD_ATTRIB_SITE_REPEATS;
    return RDAMD_SUCCESS;
  }
  p->attributes |= RDAMD_ATTRIB_SITE_REPEATS;
  if (p->clades && p->clades->max_classes != max_classes) {
    // classes were counted against the old limit: start the cache again (schedules keep
    // their own device copies; code rows already handed out stay where they are)
    RDAMD_HIP_TRY(sync_streams(p), RDAMD_FAILURE);
    for (rdamd::CladeNode &n : p->clades->nodes) { n.cls.clear(); n.cmap.clear(); }
    p->clades->intern.clear();
    // (node ids of the old generation stay valid for the schedules that hold them: nodes are
    // only appended, never re-used)
  }
  if (!p->clades) p->clades = new rdamd::CladeCache();
  p->clades->max_classes = max_classes;
  return RDAMD_SUCCESS;
}

// A batch in two halves.  batch_submit queues everything a batch needs -- inputs, P-matrices,
// clade tables, the evaluator, the copies of the results and of the batch's any-unsafe word --
// and returns; batch_wait blocks until that work is done, runs the second evaluator pass if the
// word came back set, and hands out the results.  rdamd_evaluate_batch is one after the other
// on slot 0, everything on the partition's stream.  `pipelined`: the slot form
// (rdamd_evaluate_batch_submit): the part in front of the evaluator goes to stream_pre, so that
// it runs beside the evaluator of the batch submitted before, and the evaluators follow each
// other on the partition's stream (a stream is a FIFO: batches finish in submission order).
static int batch_submit_impl(rdamd_partition_t *p, FusedWorkspace *&slot, bool pipelined, unsigned int n_jobs,
                             const rdamd_schedule_t *const *schedules,
                             const double *subst, const double *freqs,
                             const double *rates, const double *rate_weights,
                             bool host_out, void *lnl_device, bool export_children, double *mirror) {
  const bool k20 = p->states == 20 && p->rate_cats <= 8;
  if (p->states != 4 && !k20) {
    set_error(40, "rdamd_evaluate_batch: 4-state data, or 20-state data with up to 8 rate categories");
    return RDAMD_FAILURE;
  }
  const unsigned R = p->rate_cats, K = p->states, NP = K * K - K;
  std::lock_guard<std::mutex> guard(p->launch_mu);
  if (slot && slot->pend.active) {
    set_error(49, "rdamd_evaluate_batch_submit: the slot's last batch has not been waited for");
    return RDAMD_FAILURE;
  }
  RDAMD_HIP_TRY(ensure_workspace(p, slot, std::max(n_jobs, 1u)), RDAMD_FAILURE);
  FusedWorkspace *w = slot;
  w->pend = FusedWorkspace::Pending();
  w->pend.n_jobs = n_jobs; w->pend.k20 = k20; w->pend.host_out = host_out;
  w->pend.d_out = lnl_device ? (double *)lnl_device : w->d_out;
  w->pend.mirror = mirror;
  if (n_jobs == 0 || p->sites == 0) {   // an empty alignment has likelihood 1
    w->pend.active = true;
    w->pend.n_jobs = p->sites == 0 ? n_jobs : 0;
    w->pend.seq = 0;
    if (p->sites == 0 && lnl_device && n_jobs)
      RDAMD_HIP_TRY(hipMemset(lnl_device, 0, sizeof(double) * n_jobs), RDAMD_FAILURE);
    if (mirror)   // (zeros and a flag that is down, in stream order like a real batch's results)
      RDAMD_HIP_TRY(hipMemsetAsync(mirror, 0, sizeof(double) * ((size_t)n_jobs + 1), p->stream), RDAMD_FAILURE);
    return RDAMD_SUCCESS;
  }
  hipStream_t pre = p->stream;
  if (pipelined) {
    if (!p->stream_pre) {
      // high priority: its short kernels must find wave slots WHILE the other slot's evaluator
      // fills the device (at equal priority they start when that launch ends: measured, the
      // front half of every batch then sits in the gap between two evaluators)
      int least = 0, greatest = 0;
      RDAMD_HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest), RDAMD_FAILURE);
      RDAMD_HIP_TRY(hipStreamCreateWithPriority(&p->stream_pre, hipStreamNonBlocking, greatest), RDAMD_FAILURE);
    }
    if (!w->ev_ready) RDAMD_HIP_TRY(hipEventCreateWithFlags(&w->ev_ready, hipEventDisableTiming), RDAMD_FAILURE);
    if (!w->ev_done) RDAMD_HIP_TRY(hipEventCreateWithFlags(&w->ev_done, hipEventDisableTiming), RDAMD_FAILURE);
    pre = p->stream_pre;
  } else {
    // the pinned input block must not be rewritten while an earlier batch's
    // copies are still in flight
    RDAMD_HIP_TRY(hipStreamSynchronize(p->stream), RDAMD_FAILURE);
  }
  char *h = w->h_in;
  FusedJob *hj = (FusedJob *)h;          h += sizeof(FusedJob) * n_jobs;
  double *hq = (double *)h;              h += sizeof(double) * K * K * n_jobs;
  double *hf = (double *)h;              h += sizeof(double) * K * n_jobs;
  double *hr = (double *)h;              h += sizeof(double) * R * n_jobs;
  double *hw = (double *)h;
  unsigned max_depth[2] = {1, 1}, reg_levels[2] = {1, 1};   // [0] programs with pseudo-tips, [1] plain
  unsigned max_groups = 0, max_steps = 0, max_wide = 0, table_rows = 16;
  for (unsigned j = 0; j < n_jobs; ++j) {
    const rdamd_schedule_t *s = schedules[j];
    if (!s || s->part != p) {
      set_error(44, "rdamd_evaluate_batch: job %u has no schedule of this partition", j);
      return RDAMD_FAILURE;
    }
    if (s->n_groups && s->tip_generation != p->tip_generation) {
      set_error(45, "rdamd_evaluate_batch: job %u: the tip states changed after its schedule was "
                    "compiled (its site-repeat classes are stale); create the schedule again", j);
      return RDAMD_FAILURE;
    }
    hj[j].prog = s->d_prog; hj[j].brlen = s->d_brlen; hj[j].n_ops = s->n_ops;
    hj[j].prog_plain = s->d_prog_plain; hj[j].n_ops_plain = s->n_ops_plain;
    hj[j].clade_steps = s->d_steps; hj[j].clade_groups = s->d_groups;
    if (k20) hj[j].clade_steps = reinterpret_cast<const CladeStep *>(s->d_tipmask);   // (20 states: FusedJob::clade_steps carries the tip mask)
    hj[j].n_groups = s->n_groups; hj[j].n_clade_steps = s->n_steps;
    hj[j].depth = hj[j].depth_plain = 0;   // patched below: every block uses the launch-wide depth
    hj[j].tt_unsafe = export_children ? 1u : 0u;   // (set again by the P-matrix / clade-table steps of this batch;
                                                   // the exporting variant is one with every rescale test)
    hj[j].lds_pos = s->lds_pos | (s->lds_pos_plain << 16);
    max_depth[0] = std::max(max_depth[0], s->depth);
    max_depth[1] = std::max(max_depth[1], s->depth_plain);
    reg_levels[0] = std::max(reg_levels[0], s->reg_levels);
    reg_levels[1] = std::max(reg_levels[1], s->reg_levels_plain);
    max_groups = std::max(max_groups, s->n_groups);
    max_steps = std::max(max_steps, s->n_steps);
    max_wide = std::max(max_wide, s->n_wide);
    if (j && s->table_rows != table_rows) {
      set_error(47, "rdamd_evaluate_batch: job %u was compiled for another site-repeat class limit "
                    "than job 0 (16-row and 64-row schedules cannot share a launch)", j);
      return RDAMD_FAILURE;
    }
    table_rows = s->table_rows;
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
  for (unsigned j = 0; j < n_jobs; ++j) {
    hj[j].depth = max_depth[0];
    hj[j].depth_plain = max_depth[1];
  }
  const size_t clade_scratch_job = (size_t)max_steps * R * table_rows * 4;   // doubles: [step][rate][rows][4]
  // the job's tables: one 16-row table per (matrix, rate), then its 64-row tables
  const size_t tiptab_job = k20 ? (size_t)p->prob_matrices * R * kFused20TabDoubles
                                : (size_t)p->prob_matrices * R * 64 + (size_t)max_wide * R * 256;
  if (tiptab_job * n_jobs > w->tiptab_doubles) {
    RDAMD_HIP_TRY(sync_streams(p), RDAMD_FAILURE);
    if (w->d_tiptab) (void)hipFree(w->d_tiptab);
    w->d_tiptab = nullptr;
    w->tiptab_doubles = tiptab_job * std::max(n_jobs, w->cap_jobs);
    RDAMD_HIP_TRY(hipMalloc((void **)&w->d_tiptab, (w->tiptab_doubles + kTiptabPad) * sizeof(double)), RDAMD_FAILURE);
  }
  if (clade_scratch_job * n_jobs > w->clade_scratch_doubles) {
    RDAMD_HIP_TRY(sync_streams(p), RDAMD_FAILURE);
    if (w->d_clade_scratch) (void)hipFree(w->d_clade_scratch);
    w->d_clade_scratch = nullptr;
    w->clade_scratch_doubles = clade_scratch_job * std::max(n_jobs, w->cap_jobs);
    RDAMD_HIP_TRY(hipMalloc((void **)&w->d_clade_scratch, w->clade_scratch_doubles * sizeof(double)), RDAMD_FAILURE);
  }
  {   // the device block mirrors the staging block: one copy
    const size_t in_bytes = (size_t)n_jobs * (sizeof(FusedJob) + sizeof(double) * (K * K + K + 2 * R));
    memset(w->h_in + in_bytes, 0, 8);   // the any-unsafe word behind the block
    w->d_any_unsafe = (unsigned *)(w->d_in + in_bytes);
    p->stream_dirty = true;
    RDAMD_HIP_TRY(hipMemcpyAsync(w->d_in, w->h_in, in_bytes + 8, hipMemcpyHostToDevice, pre), RDAMD_FAILURE);
    w->d_jobs = (FusedJob *)(w->d_in + ((char *)hj - w->h_in));
    w->d_q = (double *)(w->d_in + ((char *)hq - w->h_in));
    w->d_freqs = (double *)(w->d_in + ((char *)hf - w->h_in));
    w->d_rates = (double *)(w->d_in + ((char *)hr - w->h_in));
    w->d_rw = (double *)(w->d_in + ((char *)hw - w->h_in));
  }

  double *d_out = w->pend.d_out;
  hipError_t e;
  // (the profiling spans are events on the partition's stream: a pipelined batch keeps the one
  // around its evaluator)
  if (k20) {
    Fused20Args b;
    b.jobs = w->d_jobs; b.tipcodes = p->d_tipcodes; b.tip_stride = p->tip_stride();
    b.codemask = p->d_codemask; b.pattern_weights = p->d_pattern_weights;
    b.pmat = w->d_pmat; b.tiptab = w->d_tiptab; b.ncodes = p->ncodes;
    b.tiptab_job_stride = (size_t)p->prob_matrices * R * kFused20TabDoubles;
    b.freqs = w->d_freqs; b.rate_weights = w->d_rw; b.partials = w->d_partials;
    b.pmat_job_stride = (size_t)p->prob_matrices * R * K * K;
    b.sites = p->sites; b.rate_cats = R; b.tiles = w->blocks_x;
    if (!pipelined) p->prof_begin(4);
    e = launch_fused20_pmatrix(b, w->d_q, w->d_qpow20, w->d_rates, n_jobs, p->prob_matrices, pre);
    if (!pipelined) p->prof_end();
    RDAMD_HIP_TRY(e, RDAMD_FAILURE);
    if (pipelined) {
      RDAMD_HIP_TRY(hipEventRecord(w->ev_ready, pre), RDAMD_FAILURE);
      RDAMD_HIP_TRY(hipStreamWaitEvent(p->stream, w->ev_ready, 0), RDAMD_FAILURE);
    }
    unsigned *export_scaler20[2] = {nullptr, nullptr};
    b.export_clv[0] = b.export_clv[1] = nullptr;
    b.export_cnt[0] = b.export_cnt[1] = nullptr;
    if (export_children) {
      // where the root operation's inner children go: the partition's own CLV / scaler buffers
      // (operand layout: rdamd_evaluate_root_children only takes partitions that keep it)
      const rdamd_schedule_t *s0 = schedules[0];
      if (n_jobs != 1 || pipelined || !p->mfma_layout) {
        set_error(50, "rdamd_evaluate_root_children: one job on the partition's stream (20 states: up to 8 rate categories)");
        return RDAMD_FAILURE;
      }
      unsigned phys_clv[2] = {0, 0};
      int phys_sc[2] = {-1, -1};
      for (int k = 0; k < 2; ++k) {
        if (s0->root_child_clv[k] < p->tips) continue;   // a tip: nothing to leave behind
        if (s0->root_child_sc[k] < 0 || (unsigned)s0->root_child_sc[k] >= p->scale_buffers) {
          set_error(50, "rdamd_evaluate_root_children: child %d of the root operation needs a scale buffer", k + 1);
          return RDAMD_FAILURE;
        }
        RDAMD_HIP_TRY(clv_phys(p, s0->root_child_clv[k], &phys_clv[k]), RDAMD_FAILURE);
        RDAMD_HIP_TRY(scaler_phys(p, s0->root_child_sc[k], &phys_sc[k]), RDAMD_FAILURE);
      }
      for (int k = 0; k < 2; ++k) {
        if (s0->root_child_clv[k] < p->tips) continue;
        if (!w->d_export_cnt)
          RDAMD_HIP_TRY(hipMalloc((void **)&w->d_export_cnt, sizeof(unsigned) * 2 * (size_t)p->sites * R), RDAMD_FAILURE);
        b.export_clv[k] = p->d_clv + (size_t)(phys_clv[k] - p->tips) * p->clv_doubles();
        b.export_cnt[k] = w->d_export_cnt + (size_t)k * p->sites * R;
        export_scaler20[k] = p->d_scaler + (size_t)phys_sc[k] * p->sites;
      }
    }
    p->prof_begin(3);
    e = export_children ? launch_fused20_export(b, max_depth[0], export_scaler20, d_out, p->stream)
                        : launch_fused20_eval(b, n_jobs, max_depth[0], d_out, p->stream);
    p->prof_end();
    RDAMD_HIP_TRY(e, RDAMD_FAILURE);
    if (host_out)
      RDAMD_HIP_TRY(hipMemcpyAsync(w->h_out, d_out, sizeof(double) * n_jobs, hipMemcpyDeviceToHost, p->stream), RDAMD_FAILURE);
    if (mirror) {   // (20 states: no second pass, the flag stays down)
      RDAMD_HIP_TRY(hipMemcpyAsync(mirror, d_out, sizeof(double) * n_jobs, hipMemcpyDeviceToDevice, p->stream), RDAMD_FAILURE);
      RDAMD_HIP_TRY(hipMemsetAsync(mirror + n_jobs, 0, sizeof(double), p->stream), RDAMD_FAILURE);
    }
  } else {
    FusedArgs a = {};
    const bool wide_codes = table_rows > 16;
    // (rdamd_set_tip_states drops the 16-bit arena; a 64-row schedule WITHOUT pseudo-tips
    // survives that call -- its programs only address tip rows -- and finds the arena rebuilt here)
    if (wide_codes) RDAMD_HIP_TRY(ensure_wide_arena(p), RDAMD_FAILURE);
    a.jobs = w->d_jobs; a.tipcodes = wide_codes ? p->d_codes_wide : p->d_tipcodes16;
    a.pattern_weights = p->d_pattern_weights;
    a.table_rows = table_rows;
    a.rates_across_waves = 0;   // (set below, once the size of the code arena is known)

    a.tiptab_job_stride = tiptab_job;
    a.pmat = w->d_pmat; a.tiptab = w->d_tiptab + kTiptabPad; a.freqs = w->d_freqs; a.rate_weights = w->d_rw;
    a.partials = w->d_partials; a.persite = nullptr;
    a.any_unsafe = w->d_any_unsafe;
    a.pmat_job_stride = (size_t)p->prob_matrices * R * 16;
    a.sites = p->sites; a.rate_cats = R;
    a.tipcodes_bytes = (unsigned)std::min<size_t>(wide_codes ? (size_t)p->wide_rows * p->tip_stride() * 2
                                                             : (size_t)p->code_rows * p->tip_stride(), 0xffffffffu);
    // One wave per rate category (kernels_fused.hip, RW) where re-reading the code arena once
    // per rate pass is what hurts: when it is far beyond every cache level.  Measured (one
    // box): c4 (850 MB of codes) 200.6 -> 181.1 ms per launch; c5 (340 MB) 77.9 -> 78.2; c2
    // (16 MB) 2.88 -> 3.08; 125.phy 1.86 -> 1.80.
    a.rates_across_waves = R >= 2 && R <= 8 && a.tipcodes_bytes >= (512u << 20);
    a.n_jobs = n_jobs;
    // Which workgroups share an XCD (and its 4 MB of L2): an eighth of the sites of EVERY job (the
    // default: each L2 then only ever sees an eighth of the code arena), or whole jobs, job j on
    // XCD j % 8 (its L2 then sees the tables of the one or two jobs it is working on).  The second
    // pays once the tables of the jobs that are on the device together outgrow the L2s: measured
    // (one box, round 4) c4 / 8 +2.4 %, c5 shard +0.8 %, 125.phy +2.3 %, c2 -2.6 %.
    {
      const double jobs_in_flight = 4096.0 / std::max(1u, w->blocks_x / 2u);   // ~16 one-wave workgroups on 256 CUs
      a.job_major = n_jobs >= 16 && jobs_in_flight * (double)tiptab_job * 8.0 > 16.0 * (1 << 20);
    }
#ifdef RDAMD_ABLATION
    if (getenv("RDAMD_FUSED_JOBMAJOR")) a.job_major = atoi(getenv("RDAMD_FUSED_JOBMAJOR")) != 0;
#endif
#ifdef RDAMD_ABLATION
    if (getenv("RDAMD_FUSED_RW")) a.rates_across_waves = atoi(getenv("RDAMD_FUSED_RW")) != 0 && R >= 2 && R <= 8;
#endif
    // No rescale tests in the first pass, a check of every site's sum at the root instead
    // (kernels_fused.hip, SPEC; include/root_digger_amd.h): where a tree is small enough that the
    // check practically never fails -- per-site likelihoods of a 256-tip alignment stay hundreds of
    // binades above 2^-900.  The rule looks at the partition's mode and tip count only -- nothing a
    // batch, or a code arena that grows as schedules are compiled, could change; the speculative
    // kernels are one-wave workgroups, so such a launch does without one wave per rate category.
    a.speculate = p->rescale_speculation < 0 ? p->tips <= kSpeculateTips : p->rescale_speculation > 0;
    if (a.speculate) a.rates_across_waves = 0;
    unsigned *export_scaler[2] = {nullptr, nullptr};
    if (export_children) {
      // where the root operation's inner children go: the partition's own CLV / scaler buffers
      const rdamd_schedule_t *s0 = schedules[0];
      if (n_jobs != 1 || pipelined || table_rows != 16 || p->mfma_layout) {
        set_error(50, "rdamd_evaluate_root_children: one job of a 16-row schedule on the partition's stream");
        return RDAMD_FAILURE;
      }
      unsigned phys_clv[2] = {0, 0};   // (sparse partitions: the children's pool slots, taken now -- both
      int phys_sc[2] = {-1, -1};       // before any address is formed: a pool that grows moves)
      for (int k = 0; k < 2; ++k) {
        if (s0->root_child_clv[k] < p->tips) continue;   // a tip: nothing to leave behind
        if (s0->root_child_sc[k] < 0 || (unsigned)s0->root_child_sc[k] >= p->scale_buffers) {
          set_error(50, "rdamd_evaluate_root_children: child %d of the root operation needs a scale buffer", k + 1);
          return RDAMD_FAILURE;
        }
        RDAMD_HIP_TRY(clv_phys(p, s0->root_child_clv[k], &phys_clv[k]), RDAMD_FAILURE);
        RDAMD_HIP_TRY(scaler_phys(p, s0->root_child_sc[k], &phys_sc[k]), RDAMD_FAILURE);
      }
      for (int k = 0; k < 2; ++k) {
        if (s0->root_child_clv[k] < p->tips) continue;
        if (!w->d_export_cnt)
          RDAMD_HIP_TRY(hipMalloc((void **)&w->d_export_cnt, sizeof(unsigned) * 2 * (size_t)p->sites * R), RDAMD_FAILURE);
        a.export_clv[k] = p->d_clv + (size_t)(phys_clv[k] - p->tips) * p->clv_doubles();
        a.export_cnt[k] = w->d_export_cnt + (size_t)k * p->sites * R;
        export_scaler[k] = p->d_scaler + (size_t)phys_sc[k] * p->sites;
      }
      a.rates_across_waves = 0;
      a.job_major = 0;
      a.speculate = 0;
    }
    if (!pipelined) p->prof_begin(4);
    e = launch_fused_pmatrix(a, w->d_q, w->d_rates, n_jobs, p->prob_matrices, pipelined, pre);
    if (e == hipSuccess && max_groups && !export_children)   // the pseudo-tips' tables, from the P-matrices and tip tables just made
      e = launch_clade_tables(a, p->clades->d_maps, w->d_clade_scratch, clade_scratch_job, n_jobs, max_groups, pipelined, pre);
    if (!pipelined) p->prof_end();
    RDAMD_HIP_TRY(e, RDAMD_FAILURE);
    if (pipelined) {
      RDAMD_HIP_TRY(hipEventRecord(w->ev_ready, pre), RDAMD_FAILURE);
      RDAMD_HIP_TRY(hipStreamWaitEvent(p->stream, w->ev_ready, 0), RDAMD_FAILURE);
    }
    p->prof_begin(3);
    // two sites per lane (kernels_fused.hip) when half the waves still fill the chip:
    // >= 4 waves for each of the 1024 SIMDs.
    unsigned ns = (size_t)n_jobs * w->blocks_x >= 8192 ? 2u : 1u;
#ifdef RDAMD_ABLATION   // A/B runs only (`make ablation`): RDAMD_FUSED_NS=1|2 overrides
    static const int force_ns = getenv("RDAMD_FUSED_NS") ? atoi(getenv("RDAMD_FUSED_NS")) : 0;
    if (force_ns) ns = (unsigned)force_ns;
#endif
#ifdef RDAMD_ABLATION   // timing only (results are garbage): what would fewer LDS stack levels / registers buy?
    if (getenv("RDAMD_FUSED_DEPTH")) max_depth[0] = (unsigned)atoi(getenv("RDAMD_FUSED_DEPTH"));
    if (getenv("RDAMD_FUSED_RL")) reg_levels[0] = (unsigned)atoi(getenv("RDAMD_FUSED_RL"));
#endif
    unsigned *h_flag = (unsigned *)(w->h_out + w->cap_jobs);
    if (export_children) {
      *h_flag = 0;   // (no second pass behind this one: batch_wait reads the word)
      e = launch_fused_export(a, max_depth[1], w->blocks_x, reg_levels[1], export_scaler, d_out,
                              host_out ? w->h_out : nullptr, p->stream);
    } else
    e = launch_fused_eval(a, n_jobs, max_depth, w->blocks_x, ns, reg_levels, false, d_out,
                          mirror ? mirror : (host_out ? w->h_out : nullptr), h_flag, p->stream,
                          mirror ? mirror + n_jobs : nullptr);
    p->prof_end();
    RDAMD_HIP_TRY(e, RDAMD_FAILURE);
    // The jobs whose tt_unsafe flag went up in this batch (a table entry in (0, 2^-128), a
    // pseudo-tip class that would have been rescaled: fused.hpp) were skipped by that pass;
    // the word that says whether there are any comes back with the results, and only then is
    // the second pass -- plain programs, tip-tip rescale test -- queued over the batch
    // (batch_wait).
    // (results and word are written into the pinned block by the finishing kernel itself)
    w->pend.a = a; w->pend.ns = ns;
    for (int k = 0; k < 2; ++k) { w->pend.max_depth[k] = max_depth[k]; w->pend.reg_levels[k] = reg_levels[k]; }
  }
  if (pipelined) RDAMD_HIP_TRY(hipEventRecord(w->ev_done, p->stream), RDAMD_FAILURE);
  w->pend.seq = ++p->batch_submitted;
  w->pend.active = true;
  return RDAMD_SUCCESS;
}

// Everything queued on the partition up to now is waited for, then counted as finished (what was
// submitted AFTER the count was taken is not: another thread's batch may have come in meanwhile).
static void drain_after_failure(rdamd_partition_t *p, uint64_t at_least) {
  uint64_t upto;
  {
    std::lock_guard<std::mutex> guard(p->launch_mu);
    upto = std::max(p->batch_submitted, at_least);
  }
  (void)sync_streams(p);
  uint64_t cur = p->batch_completed.load(std::memory_order_relaxed);
  while (cur < upto && !p->batch_completed.compare_exchange_weak(cur, upto, std::memory_order_release)) {}
}

// A submit that fails half-way may have queued copies and kernels on the slot's workspace: nothing
// may reuse it (or a schedule block those kernels read) before they have drained.
static int batch_submit(rdamd_partition_t *p, FusedWorkspace *&slot, bool pipelined, unsigned int n_jobs,
                        const rdamd_schedule_t *const *schedules,
                        const double *subst, const double *freqs,
                        const double *rates, const double *rate_weights,
                        bool host_out, void *lnl_device, bool export_children = false, double *mirror = nullptr) {
  const int rc = batch_submit_impl(p, slot, pipelined, n_jobs, schedules, subst, freqs, rates, rate_weights, host_out,
                                   lnl_device, export_children, mirror);
  if (rc != RDAMD_SUCCESS) drain_after_failure(p, 0);
  return rc;
}

static int batch_wait(rdamd_partition_t *p, FusedWorkspace *w, bool pipelined, double *lnl_host) {
  if (!w || !w->pend.active) {
    set_error(49, "rdamd_evaluate_batch_wait: no batch was submitted on this slot");
    return RDAMD_FAILURE;
  }
  FusedWorkspace::Pending &pd = w->pend;
  struct done_t {   // whatever happens, the slot is free again afterwards
    FusedWorkspace::Pending &pd;
    ~done_t() { pd.active = false; }
  } done{pd};
  if (pd.seq == 0) {   // nothing was queued (no jobs / no sites)
    if (lnl_host && pd.host_out) std::fill(lnl_host, lnl_host + pd.n_jobs, 0.0);
    return RDAMD_SUCCESS;
  }
  auto completed = [&] {   // (batches of one stream finish in submission order)
    uint64_t cur = p->batch_completed.load(std::memory_order_relaxed);
    while (cur < pd.seq && !p->batch_completed.compare_exchange_weak(cur, pd.seq, std::memory_order_release)) {}
  };
  // A failed wait or second pass must not leave the sequence number behind (parked schedule
  // blocks would wait for it for ever): drain the partition's streams, then everything queued
  // so far HAS finished, one way or the other.
  auto failed = [&] {
    drain_after_failure(p, pd.seq);
    return RDAMD_FAILURE;
  };
#define WAIT_TRY(expr) RDAMD_HIP_TRY(expr, failed())
  if (pipelined) WAIT_TRY(hipEventSynchronize(w->ev_done));
  else WAIT_TRY(sync_main(p));
  const unsigned *h_flag = (const unsigned *)(w->h_out + w->cap_jobs);
  if (!pd.k20 && *h_flag) {
    p->second_passes.fetch_add(1, std::memory_order_relaxed);
    hipError_t e = hipSuccess;
    {
      std::lock_guard<std::mutex> guard(p->launch_mu);
      // Between submit and here another thread may have compiled a schedule that GREW the code
      // arena (kernels_clade.hip, ensure_code_rows: new block, old one freed): the pointer the
      // first pass was queued with may be gone.  The plain programs of this pass address tip rows
      // only, which every generation of the arena holds at the same offsets.
      const bool wide_codes = pd.a.table_rows > 16;
      if (wide_codes) e = ensure_wide_arena(p);
      pd.a.tipcodes = wide_codes ? p->d_codes_wide : p->d_tipcodes16;
      pd.a.tipcodes_bytes = (unsigned)std::min<size_t>(wide_codes ? (size_t)p->wide_rows * p->tip_stride() * 2
                                                                  : (size_t)p->code_rows * p->tip_stride(), 0xffffffffu);
      if (e == hipSuccess) {
        p->prof_begin(3);
        e = launch_fused_eval(pd.a, pd.n_jobs, pd.max_depth, w->blocks_x, pd.ns, pd.reg_levels, true, pd.d_out,
                              pd.host_out ? w->h_out : nullptr, nullptr, p->stream);
        p->prof_end();
      }
      if (e == hipSuccess && pipelined) e = hipEventRecord(w->ev_done, p->stream);
    }
    WAIT_TRY(e);
    if (pipelined) WAIT_TRY(hipEventSynchronize(w->ev_done));
    else WAIT_TRY(hipStreamSynchronize(p->stream));
  }
#undef WAIT_TRY
  completed();
  if (lnl_host && pd.host_out) memcpy(lnl_host, w->h_out, sizeof(double) * pd.n_jobs);
  return RDAMD_SUCCESS;
}

int rdamd_partition_set_rescale_speculation(rdamd_partition_t *p, int mode) {
  clear_error();
  if (mode < -1 || mode > 1) {
    set_error(51, "rdamd_partition_set_rescale_speculation: mode %d (-1 default, 0 off, 1 on)", mode);
    return RDAMD_FAILURE;
  }
  std::lock_guard<std::mutex> guard(p->launch_mu);
  p->rescale_speculation = mode;
  return RDAMD_SUCCESS;
}

unsigned long long rdamd_evaluate_second_passes(const rdamd_partition_t *p) {
  return p->second_passes.load(std::memory_order_relaxed);
}

int rdamd_evaluate_batch(rdamd_partition_t *p, unsigned int n_jobs,
                         const rdamd_schedule_t *const *schedules, const double *subst,
                         const double *freqs, const double *rates,
                         const double *rate_weights, double *lnl_out) {
  clear_error();
  if (batch_submit(p, p->fused, false, n_jobs, schedules, subst, freqs, rates, rate_weights, true, nullptr) != RDAMD_SUCCESS)
    return RDAMD_FAILURE;
  return batch_wait(p, p->fused, false, lnl_out);
}

int rdamd_evaluate_root_children(rdamd_partition_t *p, const rdamd_operation_t *ops, unsigned int n_ops,
                                 const unsigned int *matrix_indices, const double *branch_lengths,
                                 unsigned int n_matrices, const double *subst, const double *freqs,
                                 const double *rates, const double *rate_weights, double *lnl_out) {
  clear_error();
  if (!(p->states == 4 && !p->mfma_layout) && !(p->states == 20 && p->rate_cats <= 8 && p->mfma_layout)) {
    set_error(50, "rdamd_evaluate_root_children: 4-state (or binary) partitions, and 20-state ones with up to 8 rate categories");
    return RDAMD_FAILURE;
  }
  // a schedule for this one launch: the plain program (its block comes from the partition's pool)
  rdamd_schedule_t *s = schedule_create_impl(p, ops, n_ops, matrix_indices, branch_lengths, n_matrices, false);
  if (!s) return RDAMD_FAILURE;
  const rdamd_schedule_t *list[1] = {s};
  int rc = batch_submit(p, p->fused, false, 1, list, subst, freqs, rates, rate_weights, true, nullptr, true);
  if (rc == RDAMD_SUCCESS) rc = batch_wait(p, p->fused, false, lnl_out);
  rdamd_schedule_destroy(s);
  return rc;
}

int rdamd_evaluate_batch_device(rdamd_partition_t *p, unsigned int n_jobs,
                                const rdamd_schedule_t *const *schedules,
                                const double *subst, const double *freqs,
                                const double *rates, const double *rate_weights,
                                void *d_lnl_out) {
  clear_error();
  if (batch_submit(p, p->fused, false, n_jobs, schedules, subst, freqs, rates, rate_weights, false, d_lnl_out) != RDAMD_SUCCESS)
    return RDAMD_FAILURE;
  return batch_wait(p, p->fused, false, nullptr);
}

int rdamd_evaluate_batch_submit(rdamd_partition_t *p, unsigned int slot, unsigned int n_jobs,
                                const rdamd_schedule_t *const *schedules,
                                const double *subst, const double *freqs,
                                const double *rates, const double *rate_weights) {
  clear_error();
  if (slot > 1) {
    set_error(49, "rdamd_evaluate_batch_submit: slot %u (a partition has slots 0 and 1)", slot);
    return RDAMD_FAILURE;
  }
  return batch_submit(p, slot ? p->fused1 : p->fused, true, n_jobs, schedules, subst, freqs, rates, rate_weights, true, nullptr);
}

// ---- the stream-ordered form (include/root_digger_amd.h) ------------------------------------
int rdamd_evaluate_batch_submit_device(rdamd_partition_t *p, unsigned int slot, unsigned int n_jobs,
                                       const rdamd_schedule_t *const *schedules,
                                       const double *subst, const double *freqs,
                                       const double *rates, const double *rate_weights, void *d_lnl_out) {
  clear_error();
  if (slot > 1 || !d_lnl_out) {
    set_error(49, "rdamd_evaluate_batch_submit_device: slot %u (0 or 1), device destination %p", slot, d_lnl_out);
    return RDAMD_FAILURE;
  }
  return batch_submit(p, slot ? p->fused1 : p->fused, true, n_jobs, schedules, subst, freqs, rates, rate_weights, false,
                      nullptr, false, (double *)d_lnl_out);
}

int rdamd_evaluate_batch_redo_device(rdamd_partition_t *p, unsigned int slot, void *d_lnl_out) {
  clear_error();
  FusedWorkspace *w = slot > 1 ? nullptr : (slot ? p->fused1 : p->fused);
  if (!w || !w->pend.active || !w->pend.mirror || !d_lnl_out) {
    set_error(49, "rdamd_evaluate_batch_redo_device: no device batch on slot %u", slot);
    return RDAMD_FAILURE;
  }
  FusedWorkspace::Pending &pd = w->pend;
  if (pd.seq == 0) {
    RDAMD_HIP_TRY(hipMemsetAsync(d_lnl_out, 0, sizeof(double) * ((size_t)pd.n_jobs + 1), p->stream), RDAMD_FAILURE);
    return RDAMD_SUCCESS;
  }
  // (the caller has seen the batch's summed flag: the batch itself is done, the pinned word is in)
  RDAMD_HIP_TRY(hipEventSynchronize(w->ev_done), RDAMD_FAILURE);
  const unsigned *h_flag = (const unsigned *)(w->h_out + w->cap_jobs);
  std::lock_guard<std::mutex> guard(p->launch_mu);
  if (!pd.k20 && *h_flag) {   // THIS rank's jobs need the pass (batch_wait has the notes)
    p->second_passes.fetch_add(1, std::memory_order_relaxed);
    const bool wide_codes = pd.a.table_rows > 16;
    if (wide_codes) RDAMD_HIP_TRY(ensure_wide_arena(p), RDAMD_FAILURE);
    pd.a.tipcodes = wide_codes ? p->d_codes_wide : p->d_tipcodes16;
    pd.a.tipcodes_bytes = (unsigned)std::min<size_t>(wide_codes ? (size_t)p->wide_rows * p->tip_stride() * 2
                                                                : (size_t)p->code_rows * p->tip_stride(), 0xffffffffu);
    p->prof_begin(3);
    hipError_t e = launch_fused_eval(pd.a, pd.n_jobs, pd.max_depth, w->blocks_x, pd.ns, pd.reg_levels, true, pd.d_out,
                                     nullptr, nullptr, p->stream);
    p->prof_end();
    RDAMD_HIP_TRY(e, RDAMD_FAILURE);
  }
  // this rank's results again (a collective has summed over the first copy in place), flag down
  RDAMD_HIP_TRY(hipMemcpyAsync(d_lnl_out, pd.d_out, sizeof(double) * pd.n_jobs, hipMemcpyDeviceToDevice, p->stream),
                RDAMD_FAILURE);
  RDAMD_HIP_TRY(hipMemsetAsync((double *)d_lnl_out + pd.n_jobs, 0, sizeof(double), p->stream), RDAMD_FAILURE);
  RDAMD_HIP_TRY(hipEventRecord(w->ev_done, p->stream), RDAMD_FAILURE);
  return RDAMD_SUCCESS;
}

int rdamd_evaluate_batch_finish_device(rdamd_partition_t *p, unsigned int slot) {
  clear_error();
  FusedWorkspace *w = slot > 1 ? nullptr : (slot ? p->fused1 : p->fused);
  if (!w || !w->pend.active || !w->pend.mirror) {
    set_error(49, "rdamd_evaluate_batch_finish_device: no device batch on slot %u", slot);
    return RDAMD_FAILURE;
  }
  FusedWorkspace::Pending &pd = w->pend;
  pd.active = false;
  if (pd.seq == 0) return RDAMD_SUCCESS;
  if (hipEventSynchronize(w->ev_done) != hipSuccess) {
    set_error(100, "rdamd_evaluate_batch_finish_device: the batch did not complete");
    drain_after_failure(p, pd.seq);
    return RDAMD_FAILURE;
  }
  uint64_t cur = p->batch_completed.load(std::memory_order_relaxed);
  while (cur < pd.seq && !p->batch_completed.compare_exchange_weak(cur, pd.seq, std::memory_order_release)) {}
  return RDAMD_SUCCESS;
}

int rdamd_evaluate_batch_wait(rdamd_partition_t *p, unsigned int slot, double *lnl_out) {
  clear_error();
  if (slot > 1) {
    set_error(49, "rdamd_evaluate_batch_wait: slot %u (a partition has slots 0 and 1)", slot);
    return RDAMD_FAILURE;
  }
  return batch_wait(p, slot ? p->fused1 : p->fused, true, lnl_out);
}

}  // extern "C"
