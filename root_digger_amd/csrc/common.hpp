// Internal declarations shared by the HIP translation units of librdamd.
// Not part of the C ABI (see include/root_digger_amd.h).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/root_digger_amd.h"

namespace rdamd {

struct FusedWorkspace;
void fused_workspace_free(FusedWorkspace *w);
struct CladeCache;
void clade_cache_free(CladeCache *c);

// 2^256 and 2^-256: the per-site scaling constants.
constexpr double kScaleFactor =
    115792089237316195423570985008687907853269984665640564039457584007913129639936.0;
constexpr double kScaleThreshold = 1.0 / kScaleFactor;
// log(2^-256)
constexpr double kLogScaleThreshold = -177.44567822334599;
// scratch P-matrix slots behind the caller's (root alpha variants)
constexpr unsigned kTipcodePad = 256;   // bytes of slack after the tip-code rows

void set_error(int code, const char *fmt, ...);
void clear_error();

#define RDAMD_HIP_TRY(expr, failret)                                            \
  do {                                                                          \
    hipError_t e_ = (expr);                                                     \
    if (e_ != hipSuccess) {                                                     \
      ::rdamd::set_error(100 + (int)e_, "%s failed: %s (%s:%d)", #expr,         \
                         hipGetErrorString(e_), __FILE__, __LINE__);            \
      return failret;                                                           \
    }                                                                           \
  } while (0)

// Device-side view of one partition: everything the kernels need.
struct DeviceView {
  unsigned tips, states, sites, rate_cats, ncodes_cap;
  const uint8_t *tipcodes;  // [tips][tip_stride] code index per tip character
  unsigned tip_stride;      // sites rounded up to a multiple of 4 (dword-aligned rows)
  double        *clv;       // [clv_buffers][sites][rate_cats][states]; 20-state matrix-core partitions:
                            // [clv_buffers][rate_cats][tiles][2560 B] (rdamd_partition::mfma_layout)
  unsigned      *scaler;    // [scale_buffers][sites]
  double        *pmat;      // [prob_matrices][rate_cats][states][states]
  double        *tiptab;    // [prob_matrices][rate_cats][ncodes_cap][states]
  const uint64_t *codemask; // [256]
  size_t clv_stride;        // doubles per CLV buffer (rdamd_partition::clv_doubles)
};


// ---- 20-state matrix-core layouts (kernels_clv_mfma.hip, kernels_fused_k20.hip) ----------
// Which state lane group g (= lane / 16) holds as its t-th operand (t = 0..4): the MFMA only
// needs the 20 states dealt out as 5 sets of 4, one member per lane group; this dealing
// gives group g the contiguous states 5g .. 5g+4, odd groups rotated by one.
__host__ __device__ constexpr unsigned k20_state_of(unsigned g, unsigned t) {
  return 5u * g + (t + (g & 1u)) % 5u;
}
// A CLV tile (16 sites of one rate, 320 doubles) as the wave holds it: piece 0 = operands
// t = 0, 1 of every lane (64 x 16 B), piece 1 = t = 2, 3, piece 2 = t = 4 (64 x 8 B);
// lane = 16 g + site % 16.  Index (in doubles) of (site in tile, state):
__host__ __device__ constexpr unsigned k20_tile_index(unsigned site_in_tile, unsigned state) {
  const unsigned g = state / 5u, t = (state % 5u + 5u - (g & 1u)) % 5u, lane = 16u * g + site_in_tile;
  return t < 4u ? (t >> 1) * 128u + lane * 2u + (t & 1u) : 256u + lane;
}
// A tip-table row in the same spirit: positions 0-7 = operands t = 0, 1 of lane groups
// 0..3, 8-15 = t = 2, 3, 16-19 = t = 4 -- the four lanes of a site read 64 + 64 + 32
// contiguous bytes.  State at row position w:
__host__ __device__ constexpr unsigned k20_row_state(unsigned w) {
  return w < 16u ? k20_state_of((w & 7u) >> 1, 2u * (w >> 3) + (w & 1u)) : k20_state_of(w - 16u, 4u);
}
}  // namespace rdamd

struct rdamd_partition {
  // Binary (2-state) data runs on the 4-state machinery: states 2 and 3 are
  // dummies with frequency 0, no substitutions and tip bit 0, so every kernel
  // computes exactly the 2-state recursion in entries 0-1 and zeros in 2-3.
  // `states` is what the kernels see (4 then), `api_states` what the caller set.
  unsigned api_states = 0;
  bool embedded() const { return api_states != states; }
  unsigned tips = 0, clv_buffers = 0, states = 0, sites = 0, rate_matrices = 0,
           prob_matrices = 0, rate_cats = 0, scale_buffers = 0, attributes = 0;
  int device = 0;
  hipStream_t stream = nullptr;
  // Something may be queued on `stream` that no host-side wait has covered yet.  Set by
  // everything that queues there (uploads, launches), cleared where the host has waited for the
  // stream (rdamd::sync_main).  rdamd_root_loglikelihood_fused_multi reads the partitions of MANY
  // streams from one launch on the leader's: it waits only for the streams that are dirty -- the
  // root-only steps of 32 candidates in lock step cost 32 stream synchronisations per launch
  // before, nearly all of them on idle streams.
  std::atomic<bool> stream_dirty{true};
  // rdamd_partition_stream() has handed the stream out: the caller may queue on it what the
  // library never sees -- always treated as dirty from then on
  std::atomic<bool> stream_external{false};

  // ---- HBM-resident state ------------------------------------------------
  uint8_t  *d_tipcodes = nullptr;
  uint8_t  *d_tipcodes16 = nullptr;   // 4 states: code x 16 = the byte offset of the code's row in the fused
                                      // evaluator's LDS tip tables (no shift per tip child and step).
                                      // An ARENA of rows: [0, tips) the tips, behind them the class codes of
                                      // the pseudo-tips (clades.hpp), appended as schedules discover them
  unsigned  code_rows = 0, code_rows_cap = 0;   // rows in use / allocated (tip_stride() bytes each)
  // the same arena with 16-bit entries (rows of 2 tip_stride() bytes), for schedules whose
  // pseudo-tips have up to 64 classes (offsets up to 1008); built on first use
  uint8_t  *d_codes_wide = nullptr;
  unsigned  wide_rows = 0, wide_rows_cap = 0;
  bool      code_arena_full = false;            // a code arena hit the 32-bit offset limit: no more pseudo-tip rows
  rdamd::CladeCache *clades = nullptr;          // subtree site repeats (RDAMD_ATTRIB_SITE_REPEATS)
  unsigned  tip_generation = 0;                 // bumped by rdamd_set_tip_states: schedules with pseudo-tips go stale
  double   *d_clv = nullptr;
  // CLV layout on the device.  The C ABI's layout ([site][rate][state], coraxlib's) is what
  // rdamd_get_clv hands out.  Partitions that run on clv_k20_traversal_kernel keep a CLV in
  // the matrix-core OPERAND layout instead (rdamd::k20_tile_index): [rate][tile of 16
  // sites][2 560 B], a tile holding what each of the wave's 64 lanes keeps of it in
  // registers -- loads and stores are 1 KB-contiguous instructions with no transposition.
  // Fixed at creation; the readers of d_clv are that kernel, the root reductions
  // (kernels_root.hip) and rdamd_get_clv.
  bool mfma_layout = false;
  unsigned clv_tiles() const { return (sites + 15u) / 16u; }
  size_t clv_doubles() const {   // one CLV buffer
    return mfma_layout ? (size_t)rate_cats * clv_tiles() * 16u * states : (size_t)sites * rate_cats * states;
  }
  unsigned *d_scaler = nullptr;
  // RDAMD_ATTRIB_SPARSE_CLVS: d_clv / d_scaler are POOLS of slots; a caller's buffer index gets a
  // slot when an entry point first names it (rdamd::op_phys and friends, partition.hip), and
  // rdamd_partition_discard_clvs hands all slots back.  The kernels never know: they see slot
  // numbers where a dense partition shows them the caller's indices.
  bool      sparse = false;
  std::vector<int> clv_slot, sc_slot;            // caller's index -> slot, or -1
  unsigned  clv_slots_used = 0, clv_slots_cap = 0, sc_slots_used = 0, sc_slots_cap = 0;
  unsigned  last_clv_launches = 0;   // rdamd_update_clvs_launches
  int       rescale_speculation = -1;   // rdamd_partition_set_rescale_speculation
  std::atomic<unsigned long long> second_passes{0};   // rdamd_evaluate_second_passes
  double   *d_pmat = nullptr;
  double   *d_tiptab = nullptr;
  double   *d_pmat_mfma = nullptr;   // 20-state only: MFMA-ready copy of d_pmat
  uint64_t *d_codemask = nullptr;
  double   *d_q = nullptr;        // [rate_matrices][K][K]
  double   *d_freqs = nullptr;    // [rate_matrices][K]
  double   *d_rates = nullptr;    // [R]
  double   *d_rate_weights = nullptr;
  unsigned *d_pattern_weights = nullptr;
  double   *d_tipclv_scratch = nullptr;  // expanded tip CLV for get_clv
  // scratch
  void     *d_scratch = nullptr;  // ops / matrix lists
  size_t    scratch_bytes = 0;
  double   *d_partials = nullptr; // per-block partial sums + result slots
  unsigned *d_counter = nullptr;  // arrival counter of the single-launch root kernel (stays 0 between launches)
  double   *d_result = nullptr;
  double   *d_persite = nullptr;
  double   *h_result = nullptr;   // pinned
  char     *h_stage = nullptr;    // pinned staging ring for small H2D copies
  size_t    stage_bytes = 0, stage_off = 0;

  // evaluate.hip: the batch workspaces.  Slot 0 serves rdamd_evaluate_batch; slots 0 and 1
  // alternate under rdamd_evaluate_batch_submit / _wait (two batches in flight: the second one's
  // inputs, P-matrices and clade tables are made on stream_pre while the first one's evaluator
  // runs on `stream`, and the evaluators follow each other on `stream` without a gap)
  rdamd::FusedWorkspace *fused = nullptr, *fused1 = nullptr;
  hipStream_t stream_pre = nullptr;                 // created with the first pipelined batch (high priority)
  int stream_priority = 0;                          // of `stream` (rdamd_partition_set_stream_priority)
  std::mutex launch_mu;                             // queueing on `stream` / stream_pre from several host threads
  uint64_t batch_submitted = 0;                     // batches queued so far (under launch_mu)
  std::atomic<uint64_t> batch_completed{0};         // ... and known to have finished (a stream is FIFO)
  // device blocks of destroyed schedules: reused by the next schedule once no batch in flight
  // can still read them (no hipFree -- it waits for the whole device -- in the steady state)
  struct PoolBlock { char *ptr; size_t bytes; uint64_t seq; };
  std::vector<PoolBlock> sched_pool, sched_retired;
  void *d_root_items = nullptr, *h_root_items = nullptr;   // rdamd_root_loglikelihood_fused_multi (the leading partition's)
  unsigned root_items_cap = 0;

  // ---- measurement (rdamd_profile_*) -----------------------------------------
  struct ProfSpan { hipEvent_t a, b; int kind; };
  std::vector<ProfSpan> prof_spans;
  std::vector<hipEvent_t> prof_pool;
  bool profiling = false;
  hipEvent_t prof_begin(int kind);
  void prof_end();

  // ---- host mirrors --------------------------------------------------------
  std::vector<std::vector<double>> subst, freqs;
  std::vector<double> rates, rate_weights, prop_invar;
  std::vector<unsigned> pattern_weights;
  std::vector<std::vector<double>> api_subst, api_freqs;   // caller-shaped copies (embedded only)
  std::vector<uint8_t> tipcodes;     // [tips][sites]
  std::vector<uint64_t> codemask;    // code index -> state mask
  unsigned ncodes = 0, ncodes_cap = 0;
  std::vector<char> q_dirty;         // per rate matrix
  bool tiptab_stale = false;

  unsigned tip_stride() const { return (sites + 3u) & ~3u; }
  rdamd::DeviceView view() const {
    rdamd::DeviceView v;
    v.tips = tips; v.states = states; v.sites = sites; v.rate_cats = rate_cats;
    v.ncodes_cap = ncodes_cap;
    v.tipcodes = d_tipcodes; v.tip_stride = tip_stride(); v.clv = d_clv; v.scaler = d_scaler; v.pmat = d_pmat;
    v.tiptab = d_tiptab; v.codemask = d_codemask;
    v.clv_stride = clv_doubles();
    return v;
  }
};

namespace rdamd {

// the partition's own stream, waited for on the host
inline hipError_t sync_main(rdamd_partition *p) {
  // (cleared BEFORE the wait: what another thread queues while we wait marks it again)
  p->stream_dirty = false;
  const hipError_t e = hipStreamSynchronize(p->stream);
  if (e != hipSuccess) p->stream_dirty = true;
  return e;
}
// everything queued for the partition, pipelined batches' front halves included
inline hipError_t sync_streams(rdamd_partition *p) {
  hipError_t e = sync_main(p);
  if (e == hipSuccess && p->stream_pre) e = hipStreamSynchronize(p->stream_pre);
  return e;
}

// partition.hip: the device buffer behind a caller's CLV / scaler index.  Dense partitions: the
// index itself; sparse ones (rdamd_partition::sparse): its pool slot, taken now if it has none
// (the pool grows -- a device-wide wait -- when it is full).  hipErrorInvalidValue: out of range.
hipError_t clv_phys(rdamd_partition *p, unsigned clv_index, unsigned *phys);
hipError_t scaler_phys(rdamd_partition *p, int scaler_index, int *phys);
hipError_t op_phys(rdamd_partition *p, const rdamd_operation_t &o, rdamd_operation_t *out);

// kernels_pmatrix.hip
// Rebuild Q for one rate matrix on the host into q[K*K].
void build_q_host(unsigned K, const double *subst, const double *freqs, double *q);
hipError_t launch_pmatrix(rdamd_partition *p, const unsigned *d_params_indices,
                          const unsigned *d_matrix_indices,
                          const double *d_branch_lengths, unsigned count);
hipError_t launch_tiptab_all(rdamd_partition *p);

// kernels_clv.hip
struct LevelOp {   // device-side op descriptor
  unsigned parent_clv, child1_clv, child2_clv;     // absolute clv indices
  unsigned child1_mat, child2_mat;
  int parent_sc, child1_sc, child2_sc;
  // where each child comes from: 0 tip, 1 memory, 2 register (= parent of the
  // previous op), 3+s = LDS parking slot s (4-state kernel only)
  unsigned src1, src2;
  // 4-state kernel: park = 1+s: also park the parent in LDS slot s (0: do not); noop = padding
  // entry (lists are padded to whole chunks).  20-state kernel (which has neither): the tip
  // indices of the NEXT operation's children (0 where a child is no tip, or there is no
  // next operation) -- it fetches tip codes through the scalar cache two operations ahead,
  // and taking the address from the operation in front means no scalar load has to wait
  // for another one.
  union { unsigned park; unsigned ahead1; };
  union { unsigned noop; unsigned ahead2; };
  // 4-state kernel only: byte offsets worked out on the host, so the kernel's
  // scalar unit does no 64-bit index arithmetic.  *_off of a child: its row in
  // the tip codes (tip) or its CLV (memory); kNoOffset where there is none.
  uint64_t parent_off, parent_sc_off;
  uint64_t child1_off, child1_sc_off;
  uint64_t child2_off, child2_sc_off;
};
static_assert(sizeof(LevelOp) == 96, "LevelOp: 12 words + 6 offsets");
constexpr uint64_t kNoOffset = ~0ull;
// LDS parking slots per lane the 4-state traversal kernel will have for this
// partition (0 for the other kernels): an older sibling waits there instead of
// being read back from HBM.
unsigned clv_traversal_slots(const rdamd_partition *p);
// the 4-state kernel wants its list padded with no-ops to a multiple of this (else 1)
unsigned clv_traversal_chunk(const rdamd_partition *p);
// independent pieces of one operation list, run side by side (grid.y): [start, start + len) each
constexpr unsigned kMaxListPieces = 32;
struct ListPieces {
  unsigned n = 0;
  unsigned start[kMaxListPieces] = {0}, len[kMaxListPieces] = {0};
};
// most pieces per launch a list of `count` operations of this partition is worth cutting into
// (0: run the list as it is)
unsigned clv_traversal_pieces(const rdamd_partition *p, unsigned count);
hipError_t launch_clv_traversal(rdamd_partition *p, const LevelOp *d_ops, const ListPieces &pieces,
                                unsigned slots);

// kernels_clv_mfma.hip (20 states)
hipError_t launch_pmat_to_mfma(rdamd_partition *p, const unsigned *d_matrix_indices, unsigned count);
// most pieces per launch / the size a piece keeps, for a list of `count` operations (pieces 0: run it whole)
void clv_k20_traversal_cut(const rdamd_partition *p, unsigned count, unsigned *pieces, unsigned *small, unsigned *min_count);
hipError_t launch_clv_k20_traversal(rdamd_partition *p, const LevelOp *d_ops, const ListPieces &pieces);
size_t k20_mfma_copy_doubles();               // doubles per (matrix, rate) in d_pmat_mfma

// kernels_root.hip
hipError_t launch_root_lnl(rdamd_partition *p, unsigned clv_index, int scaler_index,
                           const unsigned *d_freqs_indices, double *d_persite,
                           double *d_out);
// the whole root-only evaluation (P-matrices, root op, reduction) as one launch
hipError_t launch_root_single(rdamd_partition *p, const LevelOp &op, const double *len1,
                              const double *len2, unsigned n_positions,
                              const unsigned *params_indices, unsigned *d_counter, double *result);
// root positions one launch of the fused root kernels takes: 8 for up to 4 rate categories,
// 4 for 8 (one lane of a wave exponentiates one of the 2 R matrices of a position)
constexpr unsigned kRootMaxPositions = 8;
inline unsigned root_single_max_positions(unsigned R) { return R <= 4 ? 8u : 4u; }
struct RootSingleArgs {
  double len1[kRootMaxPositions], len2[kRootMaxPositions];      // child1 / child2 branch length per position
  unsigned params_idx[8];       // rate -> rate matrix (also the frequency set)
  unsigned n_positions;
#ifdef RDAMD_ABLATION
  unsigned abl;                 // timing experiments (profiles/root_interference.py): 1 = no exponentiation
#endif
};
// one row of root_multi_dna_kernel: everything root_single_dna_kernel takes as arguments
struct RootItem {
  DeviceView v;
  LevelOp op;
  RootSingleArgs ra;
  const double *q, *rates, *freqs, *rate_w;
  const unsigned *pw;
  const uint64_t *codemask;
  double *partials;
  unsigned *counter;
  double *result;      // [kRootMaxPositions]
  unsigned blocks, pad;
};
unsigned root_single_blocks(const rdamd_partition *p);
hipError_t launch_root_multi(const RootItem *d_items, unsigned n_items, unsigned R, unsigned max_positions,
                             unsigned max_blocks, hipStream_t stream);
// many root CLVs of one partition in one launch (bit-identical to launch_root_lnl each)
unsigned root_lnl_blocks(const rdamd_partition *p);
hipError_t launch_root_lnl_batch(rdamd_partition *p, unsigned count, const unsigned *d_clv_rel,
                                 const int *d_scaler_idx, const unsigned *d_fidx,
                                 double *d_partials, double *d_out);
}  // namespace rdamd
