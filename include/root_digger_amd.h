/*
 * root_digger_amd.h -- C ABI of the MI355X (gfx950) likelihood core for
 * RootDigger's root search.
 *
 * This is the drop-in boundary: plain C, plain pointers and sizes, no torch or
 * HIP types.  Every entry point names the reference interface it replaces.
 * The reference reaches its likelihood library (coraxlib, a C library) from
 * src/model.cpp and src/tree.cpp; citations below are relative to
 * /root/reference/.  A maintainer swaps `corax_` for `rdamd_` at the call
 * sites listed.
 *
 * All CLV / P-matrix / scaler storage lives in HBM; host pointers passed in
 * are borrowed for the duration of the call only.  Functions are NOT
 * re-entrant on the same partition (one HIP stream per partition); different
 * partitions may be driven from different host threads, as the reference does
 * (src/model.cpp:397, :429, :1935).
 */
#ifndef ROOT_DIGGER_AMD_H_
#define ROOT_DIGGER_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RDAMD_SUCCESS 1 /* CORAX_SUCCESS */
#define RDAMD_FAILURE 0 /* CORAX_FAILURE */
#define RDAMD_SCALE_BUFFER_NONE (-1)

/* attribute bits accepted for source compatibility with src/model.cpp:145-157;
 * the SIMD ones are meaningless on a GPU and ignored. */
#define RDAMD_ATTRIB_ARCH_SSE (1u << 0)
#define RDAMD_ATTRIB_ARCH_AVX (1u << 1)
#define RDAMD_ATTRIB_ARCH_AVX2 (1u << 2)
#define RDAMD_ATTRIB_SITE_REPEATS (1u << 10)
#define RDAMD_ATTRIB_NONREV (1u << 11)
/* New here (coraxlib has no such thing): the partition's CLV and scale buffers get device memory
 * when a call first names them instead of at creation, and rdamd_partition_discard_clvs() returns
 * it all.  Indices, results and every other call are unchanged.  For partitions of which only a
 * few buffers are ever live -- the model replicas of a lock-stepped search keep the root's two
 * children and nothing else (rdamd_evaluate_root_children): megabytes instead of the
 * clv_buffers x sites x rates x states x 8 bytes of a dense partition. */
#define RDAMD_ATTRIB_SPARSE_CLVS (1u << 20)

#define RDAMD_GAMMA_RATES_MEAN 0   /* CORAX_GAMMA_RATES_MEAN   */
#define RDAMD_GAMMA_RATES_MEDIAN 1 /* CORAX_GAMMA_RATES_MEDIAN */

/* error channel: replaces the corax_errno / corax_errmsg globals read at
 * src/model.cpp:439, :849 and src/msa.cpp:628-629 (thread-local here). */
int         rdamd_errno(void);
const char *rdamd_errmsg(void);

/* replaces corax_operation_t (filled at src/tree.cpp:399-410, :425-436,
 * :636-647; golden indices in test/src/tree.cpp:154-179). */
typedef struct rdamd_operation {
  unsigned int parent_clv_index;
  int          parent_scaler_index;
  unsigned int child1_clv_index;
  unsigned int child1_matrix_index;
  int          child1_scaler_index;
  unsigned int child2_clv_index;
  unsigned int child2_matrix_index;
  int          child2_scaler_index;
} rdamd_operation_t;

typedef struct rdamd_partition rdamd_partition_t;

/* ------------------------------------------------------------------------
 * Partition life cycle and setters
 * --------------------------------------------------------------------- */

/* replaces corax_partition_create, src/model.cpp:159-168.  Returns NULL and
 * sets rdamd_errmsg on failure (no GPU, out of memory, bad sizes).
 * states == 2 (binary characters, `rd --states 2`, src/main.cpp:484-488) runs
 * on the 4-state kernels with two inert states; every array that crosses this
 * boundary keeps its 2-state shape (2 substitution rates, 2 frequencies,
 * [S][R][2] CLVs, [R][2][2] P-matrices), including rdamd_evaluate_batch. */
rdamd_partition_t *rdamd_partition_create(unsigned int tips,
                                          unsigned int clv_buffers,
                                          unsigned int states,
                                          unsigned int sites,
                                          unsigned int rate_matrices,
                                          unsigned int prob_matrices,
                                          unsigned int rate_cats,
                                          unsigned int scale_buffers,
                                          unsigned int attributes);
/* replaces corax_partition_destroy, src/model.cpp:180 */
void rdamd_partition_destroy(rdamd_partition_t *p);
/* RDAMD_ATTRIB_SPARSE_CLVS partitions: every CLV and scale buffer becomes undefined (scalers read
 * as 0 again) and its device memory is free for the buffers named next.  Queued work of the
 * partition is not disturbed.  A no-op on dense partitions (their buffers keep their contents). */
void rdamd_partition_discard_clvs(rdamd_partition_t *p);
/* device bytes the CLV and scale buffers of the partition hold right now */
uint64_t rdamd_partition_clv_bytes(const rdamd_partition_t *p);

/* replaces corax_set_tip_states, src/model.cpp:310.  `map` is a 256-entry
 * char -> state-bitmask table (rdamd_map_nt / rdamd_map_bin or the caller's).
 * Returns RDAMD_FAILURE on a character the map does not know. */
int rdamd_set_tip_states(rdamd_partition_t *p, unsigned int tip_index,
                         const uint64_t *map, const char *sequence);
/* replaces corax_set_pattern_weights, src/model.cpp:324 */
void rdamd_set_pattern_weights(rdamd_partition_t *p, const unsigned int *w);
/* replaces corax_set_subst_params, src/model.cpp:185 (K*K-K values) */
void rdamd_set_subst_params(rdamd_partition_t *p, unsigned int params_index,
                            const double *params);
/* replaces corax_set_frequencies, src/model.cpp:337, :347 */
void rdamd_set_frequencies(rdamd_partition_t *p, unsigned int params_index,
                           const double *freqs);
/* replaces corax_set_category_rates, src/model.cpp:244-289 */
void rdamd_set_category_rates(rdamd_partition_t *p, const double *rates);
/* replaces corax_set_category_weights, src/model.cpp:205, :209 */
void rdamd_set_category_weights(rdamd_partition_t *p, const double *weights);
/* replaces corax_update_invariant_sites_proportion, src/model.cpp:297.  The
 * reference only ever passes 0.0; any other value is rejected. */
int rdamd_update_invariant_sites_proportion(rdamd_partition_t *p,
                                            unsigned int       params_index,
                                            double             prop_invar);
/* replaces corax_msa_empirical_frequencies, src/model.cpp:329; malloc'd
 * double[states], caller frees (src/model.cpp:338). */
double *rdamd_msa_empirical_frequencies(rdamd_partition_t *p);
/* replaces corax_compute_gamma_cats, src/model.cpp:239-270 */
int rdamd_compute_gamma_cats(double alpha, unsigned int categories,
                             double *output_rates, int rates_mode);

/* struct reads the reference performs directly on corax_partition_t
 * (src/model.cpp:330, :1043-1045, :1315): exposed as accessors. */
unsigned int  rdamd_partition_states(const rdamd_partition_t *p);
unsigned int  rdamd_partition_rate_cats(const rdamd_partition_t *p);
unsigned int  rdamd_partition_sites(const rdamd_partition_t *p);
unsigned int  rdamd_partition_tips(const rdamd_partition_t *p);
/* sum of the pattern weights (= alignment columns the partition stands for) */
double        rdamd_partition_weight_sum(const rdamd_partition_t *p);
/* the HIP stream handle (opaque here; HIP's stream type) every launch of this partition is queued on:
 * a collective queued on it runs after the batch that produced its input */
void         *rdamd_partition_stream(const rdamd_partition_t *p);
/* Dispatch priority of the partition's stream: -1 high, 0 normal (default), +1 low.  When
 * several partitions' kernels compete for the device, workgroups of a higher-priority stream
 * take the wave slots that become free first.  The lock-stepped search puts the shared
 * objective partition -- long launches that fill every CU -- on LOW priority, so that the
 * short kernels beside it (the other group's P-matrices and clade tables, the candidates'
 * root-only steps) start when a slot frees up instead of when the launch ends.  The stream is
 * re-created: call it while nothing is queued on the partition (it waits for that).  A handle
 * rdamd_partition_stream() returned before the call is INVALID afterwards: ask again. */
int           rdamd_partition_set_stream_priority(rdamd_partition_t *p, int level);
const double *rdamd_partition_subst_params(const rdamd_partition_t *p,
                                           unsigned int params_index);
const double *rdamd_partition_frequencies(const rdamd_partition_t *p,
                                          unsigned int params_index);

/* ------------------------------------------------------------------------
 * The hot path
 * --------------------------------------------------------------------- */

/* replaces corax_update_prob_matrices, src/model.cpp:367, :432, :842.
 * P[m][r] = exp(Q * rate_r * t_m) computed on the device for the whole list in
 * one launch. */
int rdamd_update_prob_matrices(rdamd_partition_t  *p,
                               const unsigned int *params_indices,
                               const unsigned int *matrix_indices,
                               const double       *branch_lengths,
                               unsigned int        count);
/* replaces corax_update_clvs, src/model.cpp:402, :440, :461, :851.  `ops` must
 * be in dependency (post-) order, as corax_utree_create_operations emits. */
void rdamd_update_clvs(rdamd_partition_t *p, const rdamd_operation_t *ops,
                       unsigned int count);
/* Diagnostic: kernel launches the partition's last rdamd_update_clvs call took.  A full traversal
 * that leaves most of the device's wave slots empty is cut into independent subtrees that run
 * side by side, level by level -- one launch per level (a profiler sees that many launches of the
 * traversal kernel per call). */
unsigned int rdamd_update_clvs_launches(const rdamd_partition_t *p);
/* replaces corax_compute_root_loglikelihood, src/model.cpp:406, :441, :466.
 * persite_lnl may be NULL (the reference always passes nullptr). */
double rdamd_compute_root_loglikelihood(rdamd_partition_t  *p,
                                        unsigned int        clv_index,
                                        int                 scaler_index,
                                        const unsigned int *freqs_indices,
                                        double             *persite_lnl);

/* Fused form of the three calls of model_t::compute_lh_root
 * (src/model.cpp:432-445): two P-matrices + one root op + reduction in a
 * single launch, for `n_alpha` root positions on the same edge at once
 * (compute_dlh, src/model.cpp:481-519, needs two).  The root CLV/scaler of the
 * LAST position is left in the partition exactly as the unfused calls would.
 * lengths1/lengths2: child1/child2 branch length per position. */
int rdamd_root_loglikelihood_fused(rdamd_partition_t       *p,
                                   const rdamd_operation_t *root_op,
                                   const unsigned int      *params_indices,
                                   const double            *lengths1,
                                   const double            *lengths2,
                                   unsigned int             n_alpha,
                                   double                  *lnl_out);
/* The same for SEVERAL partitions of one device in ONE launch: item i = partition parts[i],
 * its root operation ops[i], n_positions[i] <= 8 root positions (RDAMD_ROOT_MAX_POSITIONS;
 * <= 4 at 8 rate categories) with branch lengths lengths1[8 i + a] / lengths2[8 i + a];
 * lnl_out[8 i + a].  A lock-stepped search
 * (rdamd_model_exhaustive_search_lockstep) serves the Brent / finite-difference steps
 * (src/model.cpp:606-794) of all candidates in flight -- each on its own model replica --
 * with it.  Every partition is left as rdamd_root_loglikelihood_fused leaves it and every
 * value has that call's bits; the partitions must be idle (no call of another thread in
 * progress on them). */
#define RDAMD_ROOT_MAX_POSITIONS 8
int rdamd_root_loglikelihood_fused_multi(unsigned int              n_items,
                                         rdamd_partition_t *const *parts,
                                         const rdamd_operation_t  *ops,
                                         const unsigned int *const *params_indices,
                                         const double             *lengths1,
                                         const double             *lengths2,
                                         const unsigned int       *n_positions,
                                         double                   *lnl_out);

/* corax_compute_root_loglikelihood for MANY root CLVs of the partition in one
 * launch; every value is bit-identical to a separate call on that CLV.  Used by
 * the all-directions sweep (rdamd_model_compute_all_root_lh_directional). */
int rdamd_compute_root_loglikelihoods(rdamd_partition_t *p, unsigned int count,
                                      const unsigned int *clv_indices, const int *scaler_indices,
                                      const unsigned int *freqs_indices, double *lnl_out);

/* ------------------------------------------------------------------------
 * Batched full-traversal evaluation (the exhaustive-search inner loop)
 *
 * model_t::compute_lh_partition (src/model.cpp:454-476) is the objective the
 * L-BFGS-B driver calls 1 + n_params times per iteration with a pre-generated
 * operation list (src/model.cpp:1488-1502), and the candidate-root loop
 * (src/model.cpp:1154) repeats that per root.  These entry points evaluate a
 * whole batch of such calls -- each job = one schedule (root placement) + one
 * parameter set -- in ONE fused launch that never writes a CLV to HBM.  They
 * are stateless with respect to the partition: its CLV / P-matrix / parameter
 * state is neither read nor changed (tip states and pattern weights are).
 * 4-state (and embedded binary) data, and 20-state data with up to 8 rate
 * categories; other shapes use the three calls above.
 * --------------------------------------------------------------------- */
typedef struct rdamd_schedule rdamd_schedule_t;

/* Compiles the (ops, matrix_indices, branch_lengths) triple that
 * rooted_tree_t::generate_operations returns (src/tree.cpp:364-413) into a
 * device-resident traversal program.  Borrowed inputs; NULL + rdamd_errmsg on a
 * list that is not a complete post-order traversal ending in the root op. */
rdamd_schedule_t *rdamd_schedule_create(rdamd_partition_t       *p,
                                        const rdamd_operation_t *ops,
                                        unsigned int             n_ops,
                                        const unsigned int      *matrix_indices,
                                        const double            *branch_lengths,
                                        unsigned int             n_matrices);
void         rdamd_schedule_destroy(rdamd_schedule_t *s);
/* LDS stack slots per site the compiled traversal needs (diagnostic). */
unsigned int rdamd_schedule_stack_depth(const rdamd_schedule_t *s);

/* Subtree site repeats (coraxlib's CORAX_ATTRIB_SITE_REPEATS, which the reference sets for
 * every 4-state partition, src/model.cpp:145-149).  A partition created with
 * RDAMD_ATTRIB_SITE_REPEATS compiles its schedules with them: a directed subtree below
 * which the alignment's columns fall into at most `max_classes` distinct tip patterns is
 * evaluated once per pattern class and job (a small table) instead of once per site, and the
 * traversal meets it as one more tip.  Results are those of the plain traversal; 4-state
 * (and embedded binary) partitions only, ignored otherwise.
 * rdamd_partition_set_site_repeats changes the class limit for schedules compiled from then
 * on: 0 = off; up to 16 = pseudo-tips share the tips' 16-row tables and 8-bit codes; up to
 * 64 (the attribute's default and the largest accepted value) = 64-row tables as well, 16-bit
 * codes.  Schedules compiled under limits on different sides of 16 cannot share a batch. */
int rdamd_partition_set_site_repeats(rdamd_partition_t *p, unsigned int max_classes);
/* the class limit in force (64 unless rdamd_partition_set_site_repeats changed it); 0: no
 * site repeats on this partition */
unsigned int rdamd_partition_site_repeats(const rdamd_partition_t *p);
/* What one (site, rate) executes per traversal of a compiled schedule, and what the
 * repeats saved: the denominators of the roofline figures (bench.py). */
typedef struct rdamd_schedule_stats {
  unsigned int operations;     /* operations of the caller's list (n - 1) */
  unsigned int steps;          /* operations left after folding clades into pseudo-tips */
  unsigned int matvecs;        /* 4x4 matrix-vector products per (site, rate) in those steps */
  unsigned int matvecs_plain;  /* ... in the plain program (= inner children of the list) */
  unsigned int pseudo_tips;    /* clades folded */
  unsigned int clade_nodes;    /* their nodes = operations evaluated per class instead of per site */
  unsigned int clade_rows;     /* table rows (class x node) computed per job and rate category */
  unsigned int stack_depth;    /* LDS stack levels of the program that runs */
  unsigned int stack_depth_plain;
  /* the program's parks (a subtree's result waits while its sibling is evaluated): all of them, those
   * that wait in the evaluator's register slot(s), those in its one LDS slot (kernels with a
   * private-segment stack); the rest wait on the in-memory stack -- LDS levels or the private segment */
  unsigned int parks, parks_in_registers, parks_in_lds_slot;
} rdamd_schedule_stats_t;
int rdamd_schedule_stats(const rdamd_schedule_t *s, rdamd_schedule_stats_t *out);

/* lnl_out[j] = log-likelihood of job j.  Row-major parameter blocks, K = the
 * partition's states: subst [n_jobs][K*K-K] (corax_set_subst_params order),
 * freqs [n_jobs][K],
 * rates [n_jobs][rate_cats] and rate_weights [n_jobs][rate_cats] (either may be
 * NULL: the partition's current category rates / weights are used). */
int rdamd_evaluate_batch(rdamd_partition_t *p, unsigned int n_jobs,
                         const rdamd_schedule_t *const *schedules,
                         const double *subst, const double *freqs,
                         const double *rates, const double *rate_weights,
                         double *lnl_out);
/* The 4-state evaluator's rescaling policy.  The reference's 2^256 rule keeps
 * CLVs of large trees inside the FP64 range; up to a few hundred tips nothing comes near its end
 * (2^-1022), and the rule only moves exponents.  mode 1: a batch's first pass runs WITHOUT the
 * rescale tests and checks every site's rate sum at the root instead; a job with a sum below 2^-900
 * (or zero) is evaluated again by the second pass -- plain program, every test --, as a job with
 * tiny table entries always was: no result is ever taken from a traversal that came near the
 * range's end.  mode 0: tests on every step (rounds 1 - 5).  mode -1 (the default): 1 for
 * partitions of up to 256 tips, 0 beyond (a 500-tip tree sends every job to the second pass).
 * A job's value depends on the job and the partition's mode only, never on what shares its batch.
 * Partitions a caller does not create itself (rdamd_model_t's and its replicas') take their mode
 * from the environment when they are created: RDAMD_RESCALE_SPECULATION=0 (tests on every step,
 * whatever the tree) or =1; unset: the default rule.
 * rdamd_evaluate_second_passes: batches of this partition that needed their second pass so far. */
int rdamd_partition_set_rescale_speculation(rdamd_partition_t *p, int mode);
unsigned long long rdamd_evaluate_second_passes(const rdamd_partition_t *p);
/* Same, but the n_jobs results are left in DEVICE memory at d_lnl_out (e.g. a
 * tensor that an RCCL all-reduce sums over site-sharded ranks next).  The call BLOCKS like
 * rdamd_evaluate_batch: it returns after the partition's stream has finished writing the
 * results (the batch's second evaluator pass, needed when a job's tables hold entries small
 * enough for the per-site rescaling rule to matter, is decided on the host from a word that
 * comes back with the batch).  Until it returns, d_lnl_out[j] of such a job is unspecified. */
int rdamd_evaluate_batch_device(rdamd_partition_t *p, unsigned int n_jobs,
                                const rdamd_schedule_t *const *schedules,
                                const double *subst, const double *freqs,
                                const double *rates, const double *rate_weights,
                                void *d_lnl_out);

/* The same batch in two halves, for callers that keep TWO batches of one partition in flight
 * (the lock-stepped exhaustive search, src/model.cpp:1139-1272 run for several candidates at
 * once: while one group's batch is on the device the other group's hosts take their L-BFGS-B
 * steps).  _submit queues the batch on `slot` (0 or 1) and returns; _wait blocks until that
 * batch is done and hands out its results.  The part of a batch in front of the evaluator
 * (parameter upload, P-matrices, clade tables) runs beside the evaluator of the batch
 * submitted before it; the evaluators follow each other in submission order.  A slot takes a
 * new batch only after its last one has been waited for.  Both calls may be made from
 * different host threads (one per slot); results are those of rdamd_evaluate_batch, bit for bit. */
int rdamd_evaluate_batch_submit(rdamd_partition_t *p, unsigned int slot, unsigned int n_jobs,
                                const rdamd_schedule_t *const *schedules,
                                const double *subst, const double *freqs,
                                const double *rates, const double *rate_weights);
int rdamd_evaluate_batch_wait(rdamd_partition_t *p, unsigned int slot, double *lnl_out);

/* The STREAM-ORDERED form of a slot's batch, for callers that queue more work behind it on
 * rdamd_partition_stream(p) -- the all-reduce of a site group's per-block lnLs, a
 * copy -- without a host round trip in between.  _submit_device queues the batch like _submit;
 * its finishing kernel writes the n_jobs results to DEVICE memory d_lnl_out[0 .. n_jobs) and,
 * into d_lnl_out[n_jobs], 1.0 if some job of the batch needs the second evaluator pass (see
 * rdamd_evaluate_batch_device) and 0.0 otherwise -- so a sum over the ranks of a site group
 * carries the answer "did anybody's?" to all of them together with the sums.  d_lnl_out must
 * hold n_jobs + 1 doubles.  Nothing is decided on the host in between: whatever the caller
 * queues on the stream next runs right behind the batch.
 *   flag (summed) == 0, the common case: the results are final; _finish_device releases the slot.
 *   flag != 0: EVERY rank of the group calls _redo_device -- it runs the second pass where THIS
 *   rank's batch asked for it, writes this rank's results to d_lnl_out[0 .. n_jobs) again (the
 *   collective summed over the first copy in place) and 0.0 behind them, all in stream order --
 *   and queues its collective again; then _finish_device.
 * _finish_device waits for what _submit_device / _redo_device queued and frees the slot.
 * Results are those of rdamd_evaluate_batch, bit for bit. */
int rdamd_evaluate_batch_submit_device(rdamd_partition_t *p, unsigned int slot, unsigned int n_jobs,
                                       const rdamd_schedule_t *const *schedules,
                                       const double *subst, const double *freqs,
                                       const double *rates, const double *rate_weights,
                                       void *d_lnl_out);
int rdamd_evaluate_batch_redo_device(rdamd_partition_t *p, unsigned int slot, void *d_lnl_out);
int rdamd_evaluate_batch_finish_device(rdamd_partition_t *p, unsigned int slot);

/* compute_lh for a caller that goes on with root-only steps (model_t::exhaustive_search between
 * optimize_params and optimize_alpha, src/model.cpp:1154-1229): ONE evaluation of the whole
 * operation list -- the last operation being the root's -- through the fused evaluator, which
 * also LEAVES THE ROOT OPERATION'S TWO CHILDREN BEHIND: their CLVs and per-site scalers are
 * written to the buffers the root operation names (child*_clv_index / child*_scaler_index; a
 * tip child has nothing to write), in the layout and with the meaning rdamd_update_clvs gives
 * them, so that rdamd_root_loglikelihood_fused / rdamd_compute_root_loglikelihood-style calls
 * on that operation find what a full traversal would have left.  NO OTHER CLV, scaler or
 * P-matrix of the partition is touched (they keep whatever an earlier call left): this replaces
 * corax_update_prob_matrices + corax_update_clvs + corax_compute_root_loglikelihood
 * (src/model.cpp:357-409) where only the root's children are read afterwards -- 13 MB written
 * instead of 1.3 GB moved on BASELINE c2.  Parameters as one job of rdamd_evaluate_batch.
 * 4-state and binary partitions, and 20-state ones with up to 8 rate categories (the fused
 * evaluators' shapes); the children need scale buffers.  *lnl_out = the tree's
 * log-likelihood (every operation of the list evaluated, the reference's rescaling rule at
 * every step). */
int rdamd_evaluate_root_children(rdamd_partition_t *p, const rdamd_operation_t *ops, unsigned int n_ops,
                                 const unsigned int *matrix_indices, const double *branch_lengths,
                                 unsigned int n_matrices, const double *subst, const double *freqs,
                                 const double *rates, const double *rate_weights, double *lnl_out);


/* parity/debug views: copy device buffers to host.  rdamd_get_clv returns coraxlib's
 * layout, out[site][rate][state] (what partition->clv[i] holds in the reference),
 * whatever layout the device keeps the CLV in. */
int rdamd_get_clv(rdamd_partition_t *p, unsigned int clv_index, double *out);
int rdamd_get_scaler(rdamd_partition_t *p, unsigned int scaler_index,
                     unsigned int *out);
int rdamd_get_pmatrix(rdamd_partition_t *p, unsigned int matrix_index,
                      double *out);

/* Measurement hooks (bench.py): while enabled, every kernel launch of the
 * partition is bracketed by HIP events on the partition's stream.
 * rdamd_profile_read synchronises, returns accumulated kernel milliseconds and
 * launch counts per kernel family {0: CLV level, 1: P-matrix, 2: root lnL,
 * 3: fused traversal (+ its finishing reduction), 4: batched P-matrix, 5-7:
 * reserved}, and resets the accumulators. */
void rdamd_profile_enable(rdamd_partition_t *p, int on);
int  rdamd_profile_read(rdamd_partition_t *p, double ms_out[8],
                        unsigned int launches_out[8]);

/* blocks until all queued device work of the partition has finished. */
void rdamd_partition_sync(rdamd_partition_t *p);


/* ------------------------------------------------------------------------
 * Host-side schedule generation: rooted_tree_t (src/tree.hpp:54-201)
 *
 * The C++ class lives in root_digger_amd/csrc/tree.hpp; these wrappers expose
 * it to C / ctypes callers.  Functions returning int give RDAMD_SUCCESS or
 * RDAMD_FAILURE (+ rdamd_errmsg) where the C++ method would throw.
 * --------------------------------------------------------------------- */
typedef struct rdamd_tree rdamd_tree_t;

/* replaces root_location_t (src/tree.hpp:24-52); `edge` is a half-edge id. */
typedef struct rdamd_root_location {
  int      edge;
  uint64_t id;
  double   saved_brlen;
  double   brlen_ratio;
} rdamd_root_location_t;

/* rooted_tree_t(const std::string &tree_filename), src/tree.hpp:58-66 */
rdamd_tree_t *rdamd_tree_from_file(const char *filename);
rdamd_tree_t *rdamd_tree_from_newick(const char *newick);
void          rdamd_tree_destroy(rdamd_tree_t *t);
unsigned int  rdamd_tree_tip_count(const rdamd_tree_t *t);    /* src/tree.cpp:102 */
unsigned int  rdamd_tree_inner_count(const rdamd_tree_t *t);  /* src/tree.cpp:103 */
unsigned int  rdamd_tree_branch_count(const rdamd_tree_t *t); /* src/tree.cpp:106 */
unsigned int  rdamd_tree_root_count(const rdamd_tree_t *t);   /* src/tree.cpp:100 */
unsigned int  rdamd_tree_root_clv_index(const rdamd_tree_t *t);   /* :110 */
int           rdamd_tree_root_scaler_index(const rdamd_tree_t *t); /* :113 */
int rdamd_tree_root_location(const rdamd_tree_t *t, unsigned int index,
                             rdamd_root_location_t *out);        /* :74-82 */
int rdamd_tree_root_location_by_label(const rdamd_tree_t *t, const char *label,
                                      rdamd_root_location_t *out); /* :84-91 */
/* rank_midpoints (:863-901) / rank_modified_mad (:907-945): the root ids ordered
 * best first by how well each branch balances the tree; root_ids holds
 * root_count entries.  midpoint() is the first of rank_midpoints. */
int rdamd_tree_rank_midpoints(const rdamd_tree_t *t, unsigned int *root_ids);
int rdamd_tree_rank_modified_mad(const rdamd_tree_t *t, unsigned int *root_ids);
/* label of a root location ("(null)" when unlabeled), src/tree.hpp:36-38 */
const char *rdamd_tree_root_label(const rdamd_tree_t *t, unsigned int index);
int rdamd_tree_root_is_internal(const rdamd_tree_t *t, unsigned int index);
/* label_map(), src/tree.cpp:117-125: tip label -> tip clv index (or -1) */
int         rdamd_tree_tip_index(const rdamd_tree_t *t, const char *label);
const char *rdamd_tree_tip_label(const rdamd_tree_t *t, unsigned int clv_index);
/* tip labels on the rl.edge side of a root branch, '\n'-joined, malloc'd */
char *rdamd_tree_side_tips(const rdamd_tree_t *t, const rdamd_root_location_t *rl);

/* generate_operations, src/tree.cpp:364-413.  ops must hold tip_count
 * entries, pmatrix_indices/branch_lengths 2*tip_count entries. */
int rdamd_tree_generate_operations(rdamd_tree_t *t, const rdamd_root_location_t *rl,
                                   rdamd_operation_t *ops, unsigned int *n_ops,
                                   unsigned int *pmatrix_indices,
                                   double *branch_lengths, unsigned int *n_matrices);
/* generate_derivative_operations, src/tree.cpp:415-441 (1 op, 2 matrices) */
int rdamd_tree_generate_derivative_operations(rdamd_tree_t *t,
                                              const rdamd_root_location_t *rl,
                                              rdamd_operation_t *op,
                                              unsigned int *pmatrix_indices,
                                              double *branch_lengths);
/* generate_root_update_operations, src/tree.cpp:572-657 */
int rdamd_tree_generate_root_update_operations(rdamd_tree_t *t,
                                               const rdamd_root_location_t *rl,
                                               rdamd_operation_t *ops, unsigned int *n_ops,
                                               unsigned int *pmatrix_indices,
                                               double *branch_lengths,
                                               unsigned int *n_matrices);
int  rdamd_tree_root_by(rdamd_tree_t *t, const rdamd_root_location_t *rl); /* :273 */
void rdamd_tree_unroot(rdamd_tree_t *t);                                   /* :334 */
int  rdamd_tree_rooted(const rdamd_tree_t *t);                             /* :360 */
int  rdamd_tree_sanity_check(const rdamd_tree_t *t);                       /* :519 */
/* newick(annotations), src/tree.cpp:443-492; malloc'd, caller frees */
char *rdamd_tree_newick(const rdamd_tree_t *t, int annotations);
int   rdamd_tree_annotate_branch(rdamd_tree_t *t, const rdamd_root_location_t *rl,
                                 const char *key, const char *value); /* :731 */
/* annotate_branch(rl, key, left_value, right_value), src/tree.cpp:737-760 */
int   rdamd_tree_annotate_branch_lr(rdamd_tree_t *t, const rdamd_root_location_t *rl,
                                    const char *key, const char *left_value,
                                    const char *right_value);

/* ------------------------------------------------------------------------
 * Host-side likelihood facade: model_t (src/model.hpp:47-277)
 *
 * C wrappers over the C++ class in root_digger_amd/csrc/model.hpp for one
 * partition (the C++ class takes any number).  Rows a4-a9, a16, a17 of
 * SURVEY.md section 8a.
 * --------------------------------------------------------------------- */
typedef struct rdamd_model rdamd_model_t;

/* model_t(tree, {msa}, {rate_cats}, invariant_sites=false, seed, early_stop),
 * src/model.cpp:99-176; the tree is copied.  weights may be NULL (all 1). */
rdamd_model_t *rdamd_model_create(const rdamd_tree_t *tree, unsigned int n_taxa,
                                  const char *const *labels, const char *const *sequences,
                                  const unsigned int *weights, unsigned int states,
                                  const uint64_t *map, unsigned int rate_cats,
                                  uint64_t seed, int early_stop);
/* Same, reading the alignment from a PHYLIP / FASTA file with site-pattern
 * compression (msa_t(filename), src/msa.hpp:23-37).  *n_patterns (may be NULL)
 * receives the compressed length. */
rdamd_model_t *rdamd_model_create_from_file(const rdamd_tree_t *tree, const char *msa_filename,
                                            unsigned int states, const uint64_t *map,
                                            unsigned int rate_cats, uint64_t seed,
                                            int early_stop, int compress,
                                            unsigned int *n_patterns);
/* Parses an alignment file the way msa_t(filename) does and reports its shape
 * (host only; no GPU needed). */
int rdamd_msa_probe(const char *msa_filename, const uint64_t *map, int compress,
                    unsigned int *n_taxa, unsigned int *n_patterns,
                    unsigned int *total_weight);
void rdamd_model_destroy(rdamd_model_t *m);
/* initialize_partitions / initialize_partitions_uniform_freqs, :1297-1321 */
int rdamd_model_initialize_partitions(rdamd_model_t *m, int uniform_freqs);
int rdamd_model_set_subst_rates(rdamd_model_t *m, const double *rates);      /* :184 */
int rdamd_model_set_subst_rates_uniform(rdamd_model_t *m);                   /* :1748 */
int rdamd_model_set_freqs(rdamd_model_t *m, const double *freqs);            /* :341 */
int rdamd_model_set_empirical_freqs(rdamd_model_t *m);                       /* :327 */
int rdamd_model_set_gamma_alpha(rdamd_model_t *m, double alpha);             /* :224 */
/* compute_lh / compute_lh_root, :384-452 (NaN + rdamd_errmsg on failure) */
double rdamd_model_compute_lh(rdamd_model_t *m, const rdamd_root_location_t *rl);
double rdamd_model_compute_lh_root(rdamd_model_t *m, const rdamd_root_location_t *rl);
/* compute_dlh, :481-519: out = {lh, dlh} */
int rdamd_model_compute_dlh(rdamd_model_t *m, const rdamd_root_location_t *rl, double out[2]);
int rdamd_model_move_root(rdamd_model_t *m, const rdamd_root_location_t *rl);  /* :823 */
/* compute_all_root_lh, :1737-1746: out holds root_count values */
int rdamd_model_compute_all_root_lh(rdamd_model_t *m, double *out);
/* the same sweep as ONE fused launch (every root a job), partition untouched */
int rdamd_model_compute_all_root_lh_batched(rdamd_model_t *m, double *out);
/* the same sweep through an all-directions CLV cache: a
 * partition of its own holds the 3(n-2) directed CLVs, so the 2n-3 likelihoods
 * cost 3(n-2) + (2n-3) operations and one batched root reduction.  4-state /
 * binary single-partition models; `ratios` (root_count values) overrides the
 * stored alpha of every root when not NULL. */
int rdamd_model_compute_all_root_lh_directional(rdamd_model_t *m, const double *ratios, double *out);
/* generate_directional_operations of the tree (csrc/tree.hpp): ops holds
 * 3(n-2) + (2n-3) entries, the matrix arrays (2n-3) + 2(2n-3); sizes[3] =
 * {clv_buffers, scale_buffers, prob_matrices} the partition needs */
int rdamd_tree_generate_directional_operations(const rdamd_tree_t *t, const double *ratios,
                                               rdamd_operation_t *ops, unsigned int *n_ops,
                                               unsigned int *matrix_indices,
                                               double *branch_lengths, unsigned int *n_matrices,
                                               unsigned int *root_clv, int *root_scaler,
                                               unsigned int sizes[3]);
/* heuristic search(min_roots, root_ratio, atol, pgtol, brtol, factor), :1008-1137
 * (needs rdamd_model_set_lbfgsb); best placement and its lnL are returned. */
int rdamd_model_search(rdamd_model_t *m, unsigned int min_roots, double root_ratio, double atol,
                       double pgtol, double brtol, double factor,
                       rdamd_root_location_t *best_rl, double *best_llh);
/* optimize_alpha, :679-794 */
int rdamd_model_optimize_alpha(rdamd_model_t *m, const rdamd_root_location_t *rl, double atol,
                               rdamd_root_location_t *out);
/* batched objective over (root, parameter) pairs: subst [n][K*K-K], freqs
 * [n][K], gamma_alpha [n] (NULL = 1.0); through rdamd_evaluate_batch */
int rdamd_model_compute_lh_batch(rdamd_model_t *m, unsigned int n,
                                 const rdamd_root_location_t *rls, const double *subst,
                                 const double *freqs, const double *gamma_alpha, double *out);
/* Hands the model the caller's L-BFGS-B entry point (the reference's vendored
 * `setulb`, lib/lbfgsb/lbfgsb.h:196-201).  exhaustive_search then optimises
 * rates / frequencies / gamma alpha as optimize_params does
 * (src/model.cpp:1925-1984), with every objective + finite-difference
 * gradient evaluation (src/model.cpp:1488-1502) as ONE batched launch. */
void rdamd_model_set_lbfgsb(rdamd_model_t *m, void *setulb_entry_point);
/* optimize_params for the current parameters at root rl; returns them
 * (subst [12], freqs [4], gamma_alpha [1]) and the objective statistics. */
int rdamd_model_optimize_params(rdamd_model_t *m, const rdamd_root_location_t *rl,
                                double pgtol, double factor, int optimize_gamma,
                                double *subst, double *freqs, double *gamma_alpha,
                                uint64_t *n_batches, uint64_t *n_evaluations);
/* work counters since the model was created: out[0] objective batches (fused
 * launches), [1] objective evaluations, [2] full traversals through
 * compute_lh, [3] root-only positions (compute_lh_root / compute_dlh),
 * [4] move_root calls, [5] setulb calls.  Diagnostic. */
void rdamd_model_counters(const rdamd_model_t *m, uint64_t out[6]);
/* the last rdamd_model_exhaustive_search_lockstep: out[0] combined objective launches,
 * [1] the jobs they carried, [2] combined root-only launches
 * (rdamd_root_loglikelihood_fused_multi), [3] the candidates' root-only steps they carried */
void rdamd_model_lockstep_stats(const rdamd_model_t *m, uint64_t out[4]);
/* How the candidates in flight of a lock-stepped search meet: 0 (default) = the library's
 * choice -- from four candidates on, TWO groups whose objective batches alternate on the
 * shared partition (rdamd_evaluate_batch_submit / _wait: one group's hosts take their
 * L-BFGS-B steps while the other group's batch runs), except for a site-sharded model, whose
 * rounds each end in a collective: ONE group there (launches twice as large, half the
 * collectives); 1 = one group; 2 = two groups.
 * The records are the same either way (a job's value does not depend on its launch). */
void rdamd_model_set_lockstep_groups(rdamd_model_t *m, unsigned int groups);
/* How a lock-stepped search forms its launches.  In ARRIVAL ORDER (batch combiners: whoever has
 * asked when the others are busy elsewhere goes), or in DETERMINISTIC ROUNDS: a round closes when
 * every candidate in flight has posted its next request -- an L-BFGS-B step's evaluations, root
 * positions of its branch, a value to be summed, or "next candidate" --, the requests run as one
 * objective launch plus one root-only launch, and ONE collective sums everything the round
 * produced over the site group (csrc/lockstep_conductor.hpp).  The ranks of a site group then
 * form identical rounds, which is what lets a SITE-SHARDED model search in lock step at all.
 * mode -1 (default): rounds for site-sharded models, arrival order otherwise; 0: arrival order
 * (a site-sharded model refuses); 1: rounds always.  Records are the sequential search's either
 * way.  Rounds take single-partition models. */
void rdamd_model_set_lockstep_rounds(rdamd_model_t *m, int mode);
/* the last search in rounds: out[0] rounds closed, [1] collectives queued (rounds that had
 * anything to sum, plus repeats), [2] rounds repeated because a rank's batch needed its second
 * evaluator pass; [3] (any search) collectives THIS model asked its reducer for by itself --
 * the sequential site-sharded search's count, one per request */
void rdamd_model_round_stats(const rdamd_model_t *m, uint64_t out[4]);
/* ... and where its rounds spent their host time, seconds summed over the rounds: out[0] queueing
 * the objective batch, [1] the root-only launch (it blocks), [2] queueing the sum (a host
 * reducer: waiting for the batch and summing), [3] waiting for the round's event */
void rdamd_model_round_seconds(const rdamd_model_t *m, double out[4]);
/* stream priority (rdamd_partition_set_stream_priority) of the shared objective partition
 * during a lock-stepped search: +1 low (default), 0 leave it as it is */
void rdamd_model_set_lockstep_priority(rdamd_model_t *m, int level);
/* The searches' compute_lh between optimize_params and the root-only steps: 1 (default) =
 * rdamd_evaluate_root_children where the partition allows it (4-state / binary), 0 = always
 * the full traversal that materialises every CLV (rdamd_update_clvs). */
void rdamd_model_set_root_children_only(rdamd_model_t *m, int on);
/* assign_indicies_by_rank_exhaustive, :1867-1911 */
int rdamd_model_assign_by_rank(rdamd_model_t *m, unsigned int rank, unsigned int num_tasks);
/* exhaustive_search, :1139-1272, over the assigned roots.  root_id / llh /
 * alpha hold root_count entries; *n_results is set; best_* may be NULL. */
int rdamd_model_exhaustive_search(rdamd_model_t *m, double atol, double pgtol, double brtol,
                                  double factor, uint64_t *root_id, double *llh, double *alpha,
                                  unsigned int *n_results, rdamd_root_location_t *best_rl,
                                  double *best_llh);
/* The same loop with `workers` host threads, each driving its own replica of the
 * model (own partition, own HIP stream) and pulling candidates from a shared
 * counter: the one-GPU form of the reference's one-MPI-rank-per-chunk split
 * (src/model.cpp:1867-1911).  Results come back sorted by root id. */
int rdamd_model_exhaustive_search_parallel(rdamd_model_t *m, unsigned int workers,
                                           double atol, double pgtol, double brtol,
                                           double factor, uint64_t *root_id, double *llh,
                                           double *alpha, unsigned int *n_results,
                                           rdamd_root_location_t *best_rl, double *best_llh);
/* Lock-stepped form: `in_flight` candidates advance together; whenever all of
 * those that are inside optimize_params have asked for their next objective
 * batch (the n+1 evaluations of one L-BFGS-B step, src/model.cpp:1430-1522),
 * ONE fused launch on m's own partition serves them all.  Same results as the
 * sequential loop; launch-bound (small) alignments gain the most. */
int rdamd_model_exhaustive_search_lockstep(rdamd_model_t *m, unsigned int in_flight,
                                           double atol, double pgtol, double brtol,
                                           double factor, uint64_t *root_id, double *llh,
                                           double *alpha, unsigned int *n_results,
                                           rdamd_root_location_t *best_rl, double *best_llh);

/* ------------------------------------------------------------------------
 * Result log / checkpoint: byte-compatible with the reference's <prefix>.ckp
 * (checkpoint_t, src/checkpoint.hpp:231-300, src/checkpoint.cpp).  A search
 * appends (root id, lnL, alpha, parameters) per finished candidate under an
 * fcntl lock, so one file is shared by every process of a multi-GPU run the
 * way the reference's MPI ranks share it, and an interrupted run resumes from
 * it (by either program).
 * ---------------------------------------------------------------------- */
typedef struct rdamd_checkpoint rdamd_checkpoint_t;

typedef struct {   /* ratehet_opts_t, src/util.hpp:50-70 */
  int32_t  type;                 /* param_type: 0 emperical 1 estimate 2 equal 3 user */
  int32_t  rate_category_type;   /* 0 MEDIAN 1 MEAN 2 FREE */
  uint64_t rate_cats;
  int32_t  alpha_init;
  double   alpha;
} rdamd_ratehet_opts_t;

typedef struct {   /* the serialised fields of cli_options_t, src/checkpoint.cpp:60-91 */
  const char *msa_filename, *tree_filename, *prefix, *prefix_dir, *model_filename,
             *freqs_filename, *partition_filename, *data_type, *model_string;
  const rdamd_ratehet_opts_t *rate_cats;
  uint64_t n_rate_cats;
  uint64_t seed, min_roots, threads;
  double   root_ratio, abs_tolerance, factor, br_tolerance, bfgs_tol;
  int32_t  silent, exhaustive, echo, invariant_sites;
  int32_t  early_stop;             /* 0 unset, 1 true, 2 false (initialized_flag_t) */
  int32_t  initial_root_strategy;  /* 0 random, 1 midpoint, 2 modified_mad */
} rdamd_cli_options_t;

/* checkpoint_t(prefix): opens or creates <prefix>.ckp */
rdamd_checkpoint_t *rdamd_checkpoint_open(const char *prefix);
void        rdamd_checkpoint_close(rdamd_checkpoint_t *c);
int         rdamd_checkpoint_existing(const rdamd_checkpoint_t *c);   /* existing_checkpoint() */
const char *rdamd_checkpoint_filename(rdamd_checkpoint_t *c);
/* save_options (new file only) / load_options (existing file only; the strings
 * stay valid until the next load or close) */
int rdamd_checkpoint_save_options(rdamd_checkpoint_t *c, const rdamd_cli_options_t *o);
int rdamd_checkpoint_load_options(rdamd_checkpoint_t *c, rdamd_cli_options_t *o);
/* write(result, parameters): counts[n_partitions][4] = lengths of subst_rates,
 * freqs, gamma_alpha, gamma_weights; values = those vectors back to back */
int rdamd_checkpoint_write(rdamd_checkpoint_t *c, uint64_t root_id, double llh, double alpha,
                           unsigned int n_partitions, const uint64_t *counts,
                           const double *values);
/* read_results(): takes a snapshot; its entries are then read one by one */
int rdamd_checkpoint_read_results(rdamd_checkpoint_t *c, unsigned int *n_results);
int rdamd_checkpoint_result(const rdamd_checkpoint_t *c, unsigned int index, uint64_t *root_id,
                            double *llh, double *alpha, unsigned int *n_partitions,
                            uint64_t *n_values);
int rdamd_checkpoint_result_params(const rdamd_checkpoint_t *c, unsigned int index,
                                   uint64_t *counts, double *values);
int rdamd_checkpoint_needs_cleaning(rdamd_checkpoint_t *c);   /* 1 / 0, -1 on error */
int rdamd_checkpoint_clean(rdamd_checkpoint_t *c);
/* the file's record checksums (the reference's Adler-32 variant) */
uint32_t rdamd_checkpoint_checksum_result(uint64_t root_id, double llh, double alpha);
uint32_t rdamd_checkpoint_checksum_params(unsigned int n_partitions, const uint64_t *counts,
                                          const double *values);
/* on: the searches print the reference's progress lines ("Step i / n, ETC: h",
 * src/model.cpp:1219-1223) to stdout, counting over the roots assigned at the
 * time of the call; call it after the assign function.  off: silent. */
int rdamd_model_set_progress(rdamd_model_t *m, int on);
/* searches of this model append every finished candidate to `c` (NULL detaches),
 * src/model.cpp:1107 and :1215 */
int rdamd_model_set_checkpoint(rdamd_model_t *m, rdamd_checkpoint_t *c);
/* assign_indicies_by_rank_exhaustive(rank, num_tasks, checkpoint), :1867-1911:
 * the roots already in the log are skipped */
int rdamd_model_assign_by_rank_checkpoint(rdamd_model_t *m, unsigned int rank,
                                          unsigned int num_tasks, rdamd_checkpoint_t *c);
/* assign_indicies_by_rank_search, :1809-1865: the heuristic search's starting
 * roots = the first max(root_count*root_ratio, min_roots) of an ordering
 * (0 random shuffle with the model's seed, 1 midpoint rank, 2 modified MAD
 * rank), minus the roots already in `c` (may be NULL), chunked over the ranks */
int rdamd_model_assign_by_rank_search(rdamd_model_t *m, unsigned int min_roots, double root_ratio,
                                      unsigned int rank, unsigned int num_tasks,
                                      int initial_root_strategy, rdamd_checkpoint_t *c);
/* the root ids currently assigned to this model; returns how many there are */
int rdamd_model_assigned(const rdamd_model_t *m, uint64_t *root_ids, unsigned int cap);

/* ------------------------------------------------------------------------
 * Partition file / model string front end (src/msa.cpp:91-522): which columns
 * form each partition and what its model string asks for.
 * ---------------------------------------------------------------------- */
typedef struct {   /* partition_info_t + model_info_t, src/util.hpp:86-100 */
  char     model_name[256], partition_name[128], subst_str[64];
  unsigned int n_ranges;
  uint64_t ranges[64][2];          /* 1-based, inclusive */
  int32_t  freq_type;              /* param_type: 0 emperical 1 estimate 2 equal 3 user */
  int32_t  invar_present, invar_type;
  float    invar_user_prop;
  rdamd_ratehet_opts_t ratehet;    /* rate_cats == 0: the string has no +G / +R */
  int32_t  asc_present, asc_type;  /* 0 lewis 1 fels 2 stam */
  double   asc_fels_weight;
  unsigned int n_stam_weights;
  double   stam_weights[32];
} rdamd_partition_info_t;
/* parse_model_info, :364-415 / parse_partition_info, :417-506 */
int rdamd_parse_model_info(const char *model_string, rdamd_partition_info_t *out);
int rdamd_parse_partition_info(const char *line, rdamd_partition_info_t *out);
/* msa_t::partition, :522-591: length (patterns when compress != 0) and total
 * weight of each partition described by `lines` */
int rdamd_msa_partition_probe(const char *msa_filename, const uint64_t *map,
                              unsigned int n_lines, const char *const *lines, int compress,
                              unsigned int *lengths, unsigned int *total_weights);
/* the partitioned model of src/main.cpp:512-555: one partition per line of the
 * partition file, rate categories from each line's model string */
rdamd_model_t *rdamd_model_create_partitioned(const rdamd_tree_t *tree, const char *msa_filename,
                                              const char *partition_filename,
                                              unsigned int states, const uint64_t *map,
                                              uint64_t seed, int early_stop,
                                              unsigned int *n_partitions);
int rdamd_model_partition_count(const rdamd_model_t *m);
/* rdamd_model_create_from_file with the full rate-heterogeneity option
 * (`rd --rate-cats N --rate-cats-type {mean,median,free}`, src/main.cpp:256-266) */
rdamd_model_t *rdamd_model_create_from_file_ratehet(const rdamd_tree_t *tree,
                                                    const char *msa_filename,
                                                    unsigned int states, const uint64_t *map,
                                                    const rdamd_ratehet_opts_t *ratehet,
                                                    uint64_t seed, int early_stop, int compress,
                                                    unsigned int *n_patterns);

/* character maps (replace corax_map_nt / corax_map_bin, src/main.cpp:484) */
extern const uint64_t rdamd_map_nt[256];
extern const uint64_t rdamd_map_bin[256];

/* ---- site-sharded runs ------------------------------------------------------
 * North star / SURVEY 8(e): "candidate edges and site blocks shard across the 8
 * GPUs of one node with an RCCL all-reduce of per-block log-likelihoods".  The
 * reference has no site sharding (its MPI ranks split candidates only,
 * src/model.cpp:1867-1911); here each rank of a SITE GROUP builds its model on
 * one contiguous block of alignment columns and every log-likelihood the model
 * hands to its optimisers (compute_lh, compute_lh_root, compute_dlh, the
 * L-BFGS-B objective batches of src/model.cpp:1488-1502, the root sweeps) is
 * summed over the group through this hook before it is used, so all ranks of a
 * group follow the same trajectory bit for bit.  Empirical frequencies are
 * combined the same way (weighted by each block's column count).
 *
 * on_device = 0: `values` is a host array; sum it over the group in place.
 * on_device = 1: `values` is DEVICE memory and `stream` the HIP stream handle the
 *   producing launch was queued on: queue the collective on that stream
 *   (rdamd_comm_allreduce_sum: ncclAllGather + a rank-order sum kernel by default, or
 *   ncclAllReduce(values, values, n, ncclDouble, ncclSum, comm, stream)).
 * EVERY RANK OF THE GROUP MUST RECEIVE THE SAME BITS: the optimisers branch on these sums.  A
 * reducer that cannot promise that (see RDAMD_COMM_SUM_ALLREDUCE) forks the ranks' trajectories;
 * the searches check for it with every reduction and fail at once (divergence guard: two or three
 * extra words ride behind the values of every vector handed to the reducer -- csrc/model.cpp
 * guard_check, csrc/lockstep_conductor.hpp) instead of waiting in a collective that no longer
 * matches.
 * Return RDAMD_SUCCESS.  A site-sharded model runs its candidates sequentially
 * (rdamd_model_exhaustive_search) or in lock step in deterministic rounds
 * (rdamd_model_exhaustive_search_lockstep; every rank of the group with the same `in_flight`);
 * free-running replicas (rdamd_model_exhaustive_search_parallel) would reorder the collectives. */
typedef int (*rdamd_lnl_reducer_t)(double *values, unsigned int n, void *stream, void *user);
int rdamd_model_set_lnl_reducer(rdamd_model_t *m, rdamd_lnl_reducer_t reduce, void *user,
                                int on_device);
/* A device-side reducer in TWO halves, for the lock-stepped search of a site-sharded model
 * (rdamd_model_exhaustive_search_lockstep): `queue` only queues the collective on the stream it
 * is given and returns; `wait(event, user)` blocks until the HIP event `event` -- recorded by the
 * library on that stream, behind the collective and the copy that brings the sums back -- has
 * happened, or fails.  While one group of candidates waits for its sums the other group's round
 * is queued behind them.  The blocking form every other path uses is `queue` followed by a
 * synchronisation of the stream.
 * rdamd_model_set_lnl_reducer(m, rdamd_comm_reducer, comm, 1) installs rdamd_comm_reducer_queue /
 * rdamd_comm_reducer_wait by itself. */
typedef int (*rdamd_lnl_wait_t)(void *event, void *user);
int rdamd_model_set_lnl_reducer_async(rdamd_model_t *m, rdamd_lnl_reducer_t queue, rdamd_lnl_wait_t wait,
                                      void *user);
/* How to get this process OUT of the reducer (optional).  A lock-stepped search whose round
 * fails -- the divergence guard of csrc/lockstep_conductor.hpp, a failed launch -- fails on every
 * rank in the same round, but another worker group's round may already sit in a collective that
 * the ranks which failed a moment earlier will never join.  `abort(user)` is called once, from
 * the failing thread, and must make every pending and future call of the reducer / `wait` return
 * failure promptly; it may be called from any thread.  rdamd_model_set_lnl_reducer(m,
 * rdamd_comm_reducer, comm, 1) installs rdamd_comm_abort by itself. */
typedef void (*rdamd_lnl_abort_t)(void *user);
int rdamd_model_set_lnl_reducer_abort(rdamd_model_t *m, rdamd_lnl_abort_t abort, void *user);
/* rdamd_model_create_from_file_ratehet on block `block` of `n_blocks` contiguous
 * column blocks of the alignment (chunking of src/model.cpp:1899-1907 applied to
 * columns; the block is cut BEFORE pattern compression).  n_columns: optional,
 * the whole alignment's column count. */
rdamd_model_t *rdamd_model_create_from_file_block(const rdamd_tree_t *tree, const char *msa_filename,
                                                  unsigned int states, const uint64_t *map,
                                                  const rdamd_ratehet_opts_t *ratehet,
                                                  uint64_t seed, int early_stop, int compress,
                                                  unsigned int block, unsigned int n_blocks,
                                                  unsigned int *n_patterns,
                                                  unsigned int *n_columns);

/* RCCL communicator of one site group (librccl is loaded on first use; the
 * library has no link-time dependency on it).  Rank 0 of the group calls
 * rdamd_comm_unique_id and hands the 128 bytes to the others by any means
 * (rd_amd: its TCP rendezvous); every rank then calls rdamd_comm_create on its
 * own device.  rdamd_comm_reducer is a ready-made rdamd_lnl_reducer_t
 * (on_device = 1) whose `user` is the communicator.
 * A lost peer must not leave the other ranks of the group inside the collective:
 * rdamd_comm_reducer waits for its all-reduce by polling the stream and FAILS (after
 * ncclCommAbort; rdamd_errmsg says why) when RCCL reports an asynchronous error, when
 * nothing has arrived within the time limit (rdamd_comm_set_timeout; default 600 s, or
 * RDAMD_COMM_TIMEOUT seconds in the environment), or when another thread has called
 * rdamd_comm_abort -- rd_amd does from the thread that watches its rendezvous
 * connections.  An aborted communicator is unusable: exit and start a new process. */
typedef struct rdamd_comm rdamd_comm_t;
int           rdamd_comm_unique_id(char id[128]);
rdamd_comm_t *rdamd_comm_create(const char id[128], int rank, int n_ranks);
/* device_values[0 .. n) summed over the group in place, queued on `stream`.  HOW is the
 * communicator's sum mode:
 *   RDAMD_COMM_SUM_GATHER (default): ncclAllGather of the G vectors + a kernel on the same stream
 *     that adds them in RANK ORDER, ((v0 + v1) + v2) + ...: every rank holds the same bits by
 *     construction, whatever algorithm RCCL picks for a message of this size, and the sum is the
 *     one a host loop over the ranks makes (rd_amd --site-reduce host, dist.py) -- so a
 *     lock-stepped search gives the sequential sharded search's records bit for bit on real links;
 *   RDAMD_COMM_SUM_ALLREDUCE: one ncclAllReduce(values, values, n, ncclDouble, ncclSum).  The
 *     sum's association is RCCL's; identical bits on all ranks are NOT promised by the interface
 *     (ring / tree algorithms deliver them, direct small-message paths need not).
 * RDAMD_COMM_SUM=gather|allreduce in the environment sets the mode a new communicator starts in.
 * All ranks of a group must use the same mode. */
#define RDAMD_COMM_SUM_GATHER    0
#define RDAMD_COMM_SUM_ALLREDUCE 1
int           rdamd_comm_set_sum_mode(rdamd_comm_t *c, int mode);
int           rdamd_comm_sum_mode(const rdamd_comm_t *c);
int           rdamd_comm_allreduce_sum(rdamd_comm_t *c, double *device_values, unsigned int n,
                                       void *stream);
/* the second half of RDAMD_COMM_SUM_GATHER by itself: `ranks` vectors of n doubles, one behind
 * the other in device memory, added in rank order into out[0 .. n) (may be the first vector) */
int           rdamd_rank_order_sum(const double *gathered, double *out, unsigned int n,
                                   unsigned int ranks, void *stream);
int           rdamd_comm_reducer(double *values, unsigned int n, void *stream, void *user);
/* its two halves (rdamd_model_set_lnl_reducer_async): queue the all-reduce and return; wait for
 * a HIP event recorded behind it, with rdamd_comm_reducer's failure handling */
int           rdamd_comm_reducer_queue(double *values, unsigned int n, void *stream, void *user);
int           rdamd_comm_reducer_wait(void *event, void *user);
void          rdamd_comm_set_timeout(rdamd_comm_t *c, double seconds);
void          rdamd_comm_abort(rdamd_comm_t *c);   /* any thread */
void          rdamd_comm_destroy(rdamd_comm_t *c);

/* library / device info */
const char *rdamd_version(void);
/* Path of the HIP runtime (libamdhip64) this library runs on.  A caller that hands over
 * DEVICE pointers it got elsewhere (rdamd_evaluate_batch_device, rdamd_comm_allreduce_sum)
 * must have got them from the same runtime instance: a process can hold two (ML frameworks
 * bundle their own ROCm next to /opt/rocm); bench.py checks before it passes a tensor. */
const char *rdamd_hip_runtime_path(void);
int         rdamd_device_count(void);
/* selects the HIP device later rdamd_partition_create calls of this thread use
 * (one process per GPU: call with LOCAL_RANK). */
int         rdamd_set_device(int device);
/* free / total memory of the current device in bytes (hipMemGetInfo) */
int         rdamd_device_memory(uint64_t *free_bytes, uint64_t *total_bytes);
/* device bytes a partition of this shape takes (rdamd_partition_create's buffers) */
uint64_t    rdamd_partition_footprint(unsigned int tips, unsigned int clv_buffers,
                                      unsigned int states, unsigned int sites,
                                      unsigned int prob_matrices, unsigned int rate_cats,
                                      unsigned int scale_buffers);
/* Replicas rdamd_model_exhaustive_search_parallel / _lockstep may hold for this
 * model on the current device: `requested`, clamped so that the replicas'
 * partitions fit in 85 % of the device memory that is free now; at least 1.  A replica's
 * 4-state / binary partitions are RDAMD_ATTRIB_SPARSE_CLVS ones while the searches'
 * compute_lh is the children-only one (the default): three CLVs, not 2n - 3.  replica_bytes (optional): what one
 * replica takes.  The two searches apply this clamp themselves and say so on
 * stderr when it bites. */
unsigned int rdamd_model_max_replicas(const rdamd_model_t *m, unsigned int requested,
                                      uint64_t *replica_bytes);

#ifdef __cplusplus
}
#endif
#endif /* ROOT_DIGGER_AMD_H_ */
