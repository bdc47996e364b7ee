// Device-side records of the fused full-traversal evaluator (kernels_fused.hip)
// and the host objects that own them (evaluate.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

namespace rdamd {

enum : uint32_t {
  kFusedTT = 0,   // both children are tips
  kFusedRT = 1,   // X = running CLV (register), Y = tip
  kFusedRP = 2,   // X = running CLV (register), Y = popped from the LDS stack
  kFusedPark = 3, // 20-state programs only: push M . (running CLV) as a step of its own
};

// 64-site blocks are the unit blocks_x counts; a lane of the fused kernel carries 1
// or 2 sites (launch_fused_eval's sites_per_lane).

// One step of a compiled traversal ("program"), 32 bytes, one scalar load.
// All offsets are precomputed on the host so the kernel adds at most the rate.
// Every operation multiplies the running CLV by at most ONE matrix (pM):
//   RT / RP : the branch matrix of the running (register) child;
//   TT that parks the running CLV first: the branch matrix the parked CLV will
//   meet at its parent -- it is applied BEFORE parking, so the later pop needs
//   no matrix at all (one 4x4 matrix in SGPRs at a time, prefetchable).
struct FusedOp {
  uint32_t pM;           // byte offset of [matrix][rate 0] for the running-CLV product
  uint32_t tX, tY;       // byte offsets (x4 = tip-table offsets) of the tip operands' matrices
  uint32_t cX, cY;       // byte offset of the tip rows inside tipcodes
  uint32_t flags;        // kind | 0x100 park first | 0x200 park in register level 0 | 0x400 pop it |
                         // 0x800 park in register level 1 | 0x1000 pop it (4-state programs compiled for two)
                         // | 0x2000 / 0x4000 X / Y table has 64 rows | 0x8000 / 0x10000 this step computes the
                         // root operation's child 1 / child 2 (read by the exporting variant only)
                         // | 0x20000 park in the ONE LDS slot of the kernels with a private-segment stack | 0x40000
                         // pop it (programs whose parks are placed one by one, traversal_compiler.hpp)
  // 4-state programs compiled for 64-row table slots: the SCALAR OFFSETS of the LDS-DMA loads
  // that bring the X / Y operand's table of rate 0 into its slot -- tX * 4 + kFusedDmaBias, and
  // for Y minus kFusedDmaYSlot: the evaluator reaches the Y slot through the instruction offset,
  // which moves the source address along (kernels_fused.hip, RDAMD_LOAD_TABS64).
  // 20-state programs: the tip children's table offsets.
  uint32_t pad[2];
};
constexpr uint32_t kFusedDmaBias = 4096;    // the tables' buffer descriptor starts this far in front of them
constexpr uint32_t kFusedDmaYSlot = 2048;   // LDS byte offset of a wave's Y slot at 64 rows (32 TR)

struct CladeStep;
struct CladeGroup;
struct FusedJob {
  const FusedOp *prog;   // device
  const double  *brlen;  // device, indexed by P-matrix index
  uint32_t n_ops, depth;
  // 4 states: tt_unsafe is set by the P-matrix step when a tip-table entry of THIS JOB lies in
  // (0, 2^-128), and by the clade-table step when one of its pseudo-tips' tables has such an
  // entry or a class of a pseudo-tip would have been rescaled inside its clade.  While it is
  // 0, the product of two table rows is either 0 or >= 2^-256, a tip-tip step can never need
  // a rescale, and the evaluator variant without that test runs the job (kernels_fused.hip)
  // -- on `prog`, the program with the pseudo-tips.  When it is up, the variant with the test
  // runs it on `prog_plain`: every operation of the caller's list, tips only, every rescale
  // exactly where the reference rule puts it.  (A variant is launched over all jobs, a
  // workgroup whose job belongs to the other one returns at once; the second variant's pass
  // is only queued when the batch raised a flag at all, FusedArgs::any_unsafe.)
  uint32_t tt_unsafe;
  // (rounds 3 - 4: which in-memory stack level of the program sits in the one LDS slot of the kernels
  // with a private-segment stack.  Since round 5 the steps say it themselves, FusedOp::flags 0x20000 /
  // 0x40000; the word keeps the struct's layout)
  uint32_t lds_pos;
  // subtree site repeats (clades.hpp); without pseudo-tips prog_plain == prog, n_groups == 0
  const FusedOp    *prog_plain;
  const CladeStep  *clade_steps;     // (20-state jobs, which have no clades: a bitmap over the P-matrix indices, bit m
                                     // set = matrix m belongs to a branch that ends in a tip and needs its tip table)
  const CladeGroup *clade_groups;
  uint32_t n_ops_plain, depth_plain, n_groups, n_clade_steps;
};
static_assert(sizeof(FusedJob) == 72, "FusedJob: scalar loads at fixed offsets");

struct FusedArgs {
  const FusedJob *jobs;
  const uint8_t  *tipcodes;          // [tips + pseudo-tips][tip_stride], code x 16 (rdamd_partition::d_tipcodes16);
                                     // table_rows = 64: 16-bit entries, rows of 2 tip_stride bytes (d_codes_wide)
  const unsigned *pattern_weights;   // [sites]
  const double   *pmat;              // [job][matrix][rate][16]
  const double   *tiptab;            // [job][matrix][rate][16 codes][4]
  const double   *freqs;             // [job][4]
  const double   *rate_weights;      // [job][R]
  double         *partials;          // [job][blocks_x]
  double         *persite;           // [job][sites] or null
  size_t   pmat_job_stride;
  size_t   tiptab_job_stride;        // doubles per job: 4 pmat_job_stride (one 16-row table per matrix and rate)
                                     // + the job's 64-row tables [wide pseudo-tip][rate][2 halves][64][2]
  unsigned sites, rate_cats;
  unsigned tipcodes_bytes;           // rows in use * row bytes
  unsigned table_rows;               // 16 or 64: rows per LDS table slot (and the code arena's entry width)
  unsigned rates_across_waves;       // 1: a workgroup is R waves, one per rate category (kernels_fused.hip, RW)
  unsigned *any_unsafe;              // device word, 0 when the batch starts: set with the first FusedJob::tt_unsafe
  unsigned n_jobs;                   // jobs of the launch (grid.y may be padded, see job_major)
  unsigned job_major;                // 1: every XCD walks WHOLE jobs (job j on XCD j % 8) instead of an eighth of
                                     // the sites of every job: its L2 then sees the tables of the one or two jobs it
                                     // is working on (deep trees: 4 MB of tables per job) -- grid.y is padded to x8
  unsigned speculate;                // 1: the first pass runs WITHOUT rescale tests and sends a job whose smallest site sum
                                     // is below 2^-900 to the second pass (kernels_fused.hip, SPEC; evaluate.hip decides)
  // the exporting variant only (rdamd_evaluate_root_children): where the CLVs of the root operation's
  // two children go ([site][rate][4], the partition's own buffers; null for a tip child) and their
  // rescale counts per (site, rate) ([site][rate]; launch_fused_export turns them into per-site scalers)
  double   *export_clv[2];
  unsigned *export_cnt[2];
};

// ---- 20-state variant (kernels_fused_k20.hip) ---------------------------------
// Same programs (pM / tX / tY are byte offsets of [matrix][rate 0] inside one
// job's MFMA-ready P copies, 3200 bytes per (matrix, rate)); a TT step never
// parks -- the compiler emits a kFusedPark step before it instead, so a step
// multiplies by at most two matrices.
struct Fused20Args {
  const FusedJob *jobs;
  const uint8_t  *tipcodes;          // [tips][tip_stride]
  unsigned        tip_stride;
  const uint64_t *codemask;          // [256] code -> state mask
  const unsigned *pattern_weights;   // [sites]
  const double   *pmat;              // [job][matrix][rate][400]  MFMA-ready (kernels_clv_mfma.hip)
  const double   *tiptab;            // [job][matrix][rate][64 codes][24]: a 192-byte row per code, laid out
                                     // for the evaluator's loads (fused20_pmatrix_kernel)
  const double   *freqs;             // [job][20]
  const double   *rate_weights;      // [job][R]
  double         *partials;          // [job][tiles]
  size_t   pmat_job_stride;          // doubles per job
  size_t   tiptab_job_stride;        // doubles per job
  unsigned sites, rate_cats, tiles, ncodes;
  // the exporting variant only (rdamd_evaluate_root_children on a 20-state partition): where the
  // CLVs of the root operation's two children go -- the partition's own buffers, in its matrix-core
  // operand layout [rate][tile][320] (common.hpp, k20_tile_index); null for a tip child -- and
  // their rescale counts per (site, rate) ([site][rate]; the fix-up kernel turns them into the
  // reference's per-site scalers)
  double   *export_clv[2];
  unsigned *export_cnt[2];
};
// 4 states: in-memory stack entries of a program whose stack lives in one register slot, one
// LDS slot and the wave's private segment (kernels_fused.hip, SP; the private segment has a
// slot for every in-memory entry, the LDS one's stays unused); deeper programs take two
// register levels and all-LDS stacks
constexpr unsigned kFusedSpillLevels = 7;
constexpr unsigned kFused20TabCodes = 64;                        // rows per (matrix, rate)
constexpr unsigned kFused20TabRow = 4 * 6;                       // doubles per code
constexpr unsigned kFused20TabDoubles = kFused20TabCodes * kFused20TabRow;   // per (matrix, rate)
// d_qpow: n_jobs x fused20_qpow_doubles() doubles of scratch -- the powers Q^1 .. Q^16 of every job's Q
// (and its 1-norm), made in front of the P-matrices they all share
size_t fused20_qpow_doubles();
hipError_t launch_fused20_pmatrix(const Fused20Args &a, const double *d_q, double *d_qpow, const double *d_rates,
                                  unsigned n_jobs, unsigned n_mat, hipStream_t stream);
hipError_t launch_fused20_eval(const Fused20Args &a, unsigned n_jobs, unsigned max_depth,
                               double *d_out, hipStream_t stream);
// ONE job through the exporting variant: the evaluation, and the root operation's inner children left
// in a.export_clv / d_scaler[] as a traversal with per-site scalers leaves them (launch_fused_export's
// 20-state sibling)
hipError_t launch_fused20_export(const Fused20Args &a, unsigned max_depth, unsigned *const d_scaler[2],
                                 double *d_out, hipStream_t stream);

hipError_t launch_fused_pmatrix(const FusedArgs &a, const double *d_q, const double *d_rates,
                                unsigned n_jobs, unsigned n_mat, bool slim, hipStream_t stream);
// depth / reg_levels: [0] of the programs with pseudo-tips (FusedJob::prog), [1] of the plain
// programs (prog_plain); the two evaluator variants are launched with their own LDS sizes
// unsafe_pass: false = the jobs whose tt_unsafe flag is down (programs with pseudo-tips, no
// tip-tip rescale test), true = the others (plain programs, with the test).  The caller runs
// the second pass only when FusedArgs::any_unsafe came back set.
// h_out / h_flag (pinned host memory, or null): the finishing kernel writes the results and the
// batch's any-unsafe word there as well -- no copy launches behind the batch.  flag_f64 (device
// memory, or null): the word once more, as 0.0 / 1.0 (rdamd_evaluate_batch_submit_device).
hipError_t launch_fused_eval(const FusedArgs &a, unsigned n_jobs, const unsigned max_depth[2],
                             unsigned blocks_x, unsigned sites_per_lane, const unsigned reg_levels[2],
                             bool unsafe_pass, double *d_out, double *h_out, unsigned *h_flag,
                             hipStream_t stream, double *flag_f64 = nullptr);
// ONE job (its tt_unsafe word set by the caller: the plain program, every rescale test) through the
// exporting variant: the evaluation as above, and the root operation's inner children left in
// a.export_clv / d_scaler[] as a traversal with per-site scalers leaves them.  16-row schedules only.
hipError_t launch_fused_export(const FusedArgs &a, unsigned max_depth, unsigned blocks_x, unsigned reg_levels,
                               unsigned *const d_scaler[2], double *d_out, double *h_out, hipStream_t stream);


}  // namespace rdamd
