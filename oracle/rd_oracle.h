/*
 * rd_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the likelihood arithmetic RootDigger obtains from
 * coraxlib (corax_update_prob_matrices / corax_update_clvs /
 * corax_compute_root_loglikelihood and the setters that feed them).  coraxlib
 * is an EMPTY, un-pinned submodule in the reference mount
 * (/root/reference/.gitmodules:4-6, lib/coraxlib/ has no files), so this file
 * restates the published libpll-2/coraxlib algorithm and anchors on the
 * reference's own call sites (src/model.cpp:159-168, :184-355, :357-476).
 *
 * PARITY STATUS: "parity unpinned" for ABSOLUTE lnL against coraxlib itself
 * (the reference's tests hold no known-answer lnL, test/src/model.cpp holds
 * property checks only).  The oracle IS pinned against
 *   - closed-form JC69 likelihoods,
 *   - an independent SciPy (scipy.linalg.expm) Felsenstein pruning
 *     (oracle/gen_golden.py writes the JSON files under tests/golden/),
 *   - the reference's property tests (determinism, full == root-only,
 *     root invariance under a reversible model, test/src/model.cpp:59-75,
 *     :271-288, :367-387).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product path (root_digger_amd/) never does.
 */
#ifndef RD_ORACLE_H_
#define RD_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_SUCCESS 1
#define ORC_FAILURE 0
#define ORC_SCALE_BUFFER_NONE (-1)
#define ORC_GAMMA_RATES_MEAN 0
#define ORC_GAMMA_RATES_MEDIAN 1

/* same field order as corax_operation_t (filled at src/tree.cpp:399-410) */
typedef struct orc_operation {
  unsigned int parent_clv_index;
  int          parent_scaler_index;
  unsigned int child1_clv_index;
  unsigned int child1_matrix_index;
  int          child1_scaler_index;
  unsigned int child2_clv_index;
  unsigned int child2_matrix_index;
  int          child2_scaler_index;
} orc_operation_t;

typedef struct orc_partition orc_partition_t;

/* src/model.cpp:159-168 */
orc_partition_t *orc_partition_create(unsigned int tips,
                                      unsigned int clv_buffers,
                                      unsigned int states,
                                      unsigned int sites,
                                      unsigned int rate_matrices,
                                      unsigned int prob_matrices,
                                      unsigned int rate_cats,
                                      unsigned int scale_buffers,
                                      unsigned int attributes);
void orc_partition_destroy(orc_partition_t *p); /* src/model.cpp:180 */

/* src/model.cpp:310 ; map = 256-entry char -> state bitmask */
int orc_set_tip_states(orc_partition_t *p,
                       unsigned int     tip_index,
                       const uint64_t  *map,
                       const char      *sequence);
/* direct tip CLV upload (K doubles per site), for >64-state-free generic use */
void orc_set_tip_clv(orc_partition_t *p, unsigned int tip_index,
                     const double *clv_site_state);
void orc_set_pattern_weights(orc_partition_t *p, const unsigned int *w);
void orc_set_subst_params(orc_partition_t *p, unsigned int idx,
                          const double *params);
void orc_set_frequencies(orc_partition_t *p, unsigned int idx,
                         const double *freqs);
void orc_set_category_rates(orc_partition_t *p, const double *rates);
void orc_set_category_weights(orc_partition_t *p, const double *weights);
double *orc_msa_empirical_frequencies(orc_partition_t *p); /* malloc'd */
int orc_compute_gamma_cats(double alpha, unsigned int cats, double *out,
                           int mode);

/* src/model.cpp:367,:432,:842 */
int orc_update_prob_matrices(orc_partition_t    *p,
                             const unsigned int *params_indices,
                             const unsigned int *matrix_indices,
                             const double       *branch_lengths,
                             unsigned int        count);
/* src/model.cpp:402,:440,:461,:851 */
void orc_update_clvs(orc_partition_t *p, const orc_operation_t *ops,
                     unsigned int count);
/* The same update through 256-bit vectors (4-state data; what the reference's
 * coraxlib build selects for nucleotides, src/model.cpp:145-155), bit-identical
 * to orc_update_clvs.  ORC_FAILURE for other state counts or without AVX2. */
int orc_update_clvs_avx2(orc_partition_t *p, const orc_operation_t *ops, unsigned int count);
/* src/model.cpp:406,:441,:466 */
double orc_compute_root_loglikelihood(orc_partition_t    *p,
                                      unsigned int        clv_index,
                                      int                 scaler_index,
                                      const unsigned int *freqs_indices,
                                      double             *persite_lnl);

/* The same traversal WITH subtree site repeats (the reference sets
 * CORAX_ATTRIB_SITE_REPEATS for 4-state data, src/model.cpp:145-149): every node's
 * CLV is computed once per class of alignment columns that agree at all tips below
 * it (libpll-2's pll_update_repeats scheme, restated in rd_oracle.c).  Values and
 * log-likelihood are bit-identical to the plain calls'; the buffers hold per-class
 * data afterwards, so a traversal uses either these two calls or the plain ones.
 * avx2 != 0: the 256-bit-vector inner loop for 4 states.  bench.py's honest CPU
 * comparator (cpu_baseline.with_site_repeats). */
void orc_update_clvs_repeats(orc_partition_t *p, const orc_operation_t *ops, unsigned int count, int avx2);
double orc_compute_root_loglikelihood_repeats(orc_partition_t *p, unsigned int clv_index, int scaler_index,
                                              const unsigned int *freqs_indices);
/* classes computed / columns a plain loop would have computed so far (1.0: no repeats found) */
double orc_repeats_ratio(const orc_partition_t *p);
/* bytes the site-repeats traversals of this partition have moved so far: out[0] written (class
 * CLVs, scalers, class arrays), out[1] read at least once (compulsory: the children's class arrays,
 * every class CLV of an inner child once), out[2] read when no child class is found in a cache */
void orc_repeats_bytes(const orc_partition_t *p, double out[3]);
/* STREAM triad over `threads` pinned threads (cpus may be NULL), `doubles` elements per array and
 * thread, for `seconds`: aggregate GB/s by STREAM's count (24 bytes per element) */
double orc_stream_triad(int threads, const int *cpus, size_t doubles, double seconds);

/* raw views for parity tests */
const double       *orc_get_clv(const orc_partition_t *p, unsigned int idx);
const unsigned int *orc_get_scaler(const orc_partition_t *p, unsigned int idx);
const double       *orc_get_pmatrix(const orc_partition_t *p, unsigned int idx);
void orc_get_qmatrix(const orc_partition_t *p, unsigned int idx, double *out);
unsigned int orc_states(const orc_partition_t *p);
unsigned int orc_rate_cats(const orc_partition_t *p);
unsigned int orc_sites(const orc_partition_t *p);

/* dense expm (row-major n*n), exposed so tests can pin it against SciPy */
void orc_expm(const double *a, unsigned int n, double *out);

extern const uint64_t orc_map_nt[256];

#ifdef __cplusplus
}
#endif
#endif
