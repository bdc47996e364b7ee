#define _GNU_SOURCE   /* (pthread_setaffinity_np, CPU_SET: the STREAM triad at the end) */
/*
 * rd_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See rd_oracle.h for status ("parity unpinned" vs coraxlib absolute lnL;
 * pinned vs closed forms + SciPy goldens + the reference's property tests).
 *
 * Layouts follow SURVEY.md Appendix A:
 *   CLV      [site][rate][state] doubles            (A3)
 *   P-matrix [rate][parent state][child state]      (A2)
 *   scaler   [site] unsigned, per-site 2^256 rule   (A4)
 */
#include "rd_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* 2^256 exactly, and its reciprocal (the libpll/coraxlib scaling constants) */
#define ORC_SCALE_FACTOR 115792089237316195423570985008687907853269984665640564039457584007913129639936.0
#define ORC_SCALE_THRESHOLD (1.0 / ORC_SCALE_FACTOR)

struct orc_partition {
  unsigned int tips, clv_buffers, states, sites, rate_matrices, prob_matrices,
      rate_cats, scale_buffers, attributes;
  double      **clv;      /* tips + clv_buffers, each [S][R][K] */
  unsigned int **scaler;  /* scale_buffers, each [S] */
  double      **pmatrix;  /* prob_matrices, each [R][K][K] */
  double      **subst;    /* rate_matrices, each [K*K-K] */
  double      **freqs;    /* rate_matrices, each [K] */
  double       *rates;    /* [R] */
  double       *rate_weights; /* [R] */
  double       *prop_invar;   /* rate_matrices (always 0, src/model.cpp:297) */
  unsigned int *pattern_weights; /* [S] */
  struct orc_repeats *rep;       /* site repeats (orc_update_clvs_repeats), lazily built */
};

const uint64_t orc_map_nt[256] = {
    ['A'] = 1,  ['a'] = 1,  ['C'] = 2,  ['c'] = 2,  ['G'] = 4,  ['g'] = 4,
    ['T'] = 8,  ['t'] = 8,  ['U'] = 8,  ['u'] = 8,  ['R'] = 5,  ['r'] = 5,
    ['Y'] = 10, ['y'] = 10, ['S'] = 6,  ['s'] = 6,  ['W'] = 9,  ['w'] = 9,
    ['K'] = 12, ['k'] = 12, ['M'] = 3,  ['m'] = 3,  ['B'] = 14, ['b'] = 14,
    ['D'] = 13, ['d'] = 13, ['H'] = 11, ['h'] = 11, ['V'] = 7,  ['v'] = 7,
    ['N'] = 15, ['n'] = 15, ['O'] = 15, ['o'] = 15, ['X'] = 15, ['x'] = 15,
    ['-'] = 15, ['?'] = 15,
};

orc_partition_t *orc_partition_create(unsigned int tips,
                                      unsigned int clv_buffers,
                                      unsigned int states,
                                      unsigned int sites,
                                      unsigned int rate_matrices,
                                      unsigned int prob_matrices,
                                      unsigned int rate_cats,
                                      unsigned int scale_buffers,
                                      unsigned int attributes) {
  orc_partition_t *p = (orc_partition_t *)calloc(1, sizeof(*p));
  if (!p) return NULL;
  p->tips = tips; p->clv_buffers = clv_buffers; p->states = states;
  p->sites = sites; p->rate_matrices = rate_matrices;
  p->prob_matrices = prob_matrices; p->rate_cats = rate_cats;
  p->scale_buffers = scale_buffers; p->attributes = attributes;
  size_t clv_len = (size_t)sites * rate_cats * states;
  p->clv = (double **)calloc(tips + clv_buffers, sizeof(double *));
  for (unsigned int i = 0; i < tips + clv_buffers; ++i)
    p->clv[i] = (double *)calloc(clv_len ? clv_len : 1, sizeof(double));
  p->scaler = (unsigned int **)calloc(scale_buffers ? scale_buffers : 1,
                                      sizeof(unsigned int *));
  for (unsigned int i = 0; i < scale_buffers; ++i)
    p->scaler[i] = (unsigned int *)calloc(sites ? sites : 1, sizeof(unsigned int));
  p->pmatrix = (double **)calloc(prob_matrices ? prob_matrices : 1, sizeof(double *));
  for (unsigned int i = 0; i < prob_matrices; ++i)
    p->pmatrix[i] = (double *)calloc((size_t)rate_cats * states * states, sizeof(double));
  p->subst = (double **)calloc(rate_matrices, sizeof(double *));
  p->freqs = (double **)calloc(rate_matrices, sizeof(double *));
  for (unsigned int i = 0; i < rate_matrices; ++i) {
    p->subst[i] = (double *)calloc((size_t)states * states - states, sizeof(double));
    p->freqs[i] = (double *)calloc(states, sizeof(double));
    for (unsigned int k = 0; k < states * states - states; ++k) p->subst[i][k] = 1.0;
    for (unsigned int k = 0; k < states; ++k) p->freqs[i][k] = 1.0 / states;
  }
  p->rates        = (double *)calloc(rate_cats, sizeof(double));
  p->rate_weights = (double *)calloc(rate_cats, sizeof(double));
  for (unsigned int k = 0; k < rate_cats; ++k) {
    p->rates[k] = 1.0;
    p->rate_weights[k] = 1.0 / rate_cats;
  }
  p->prop_invar = (double *)calloc(rate_matrices, sizeof(double));
  p->pattern_weights = (unsigned int *)calloc(sites ? sites : 1, sizeof(unsigned int));
  for (unsigned int s = 0; s < sites; ++s) p->pattern_weights[s] = 1;
  return p;
}

static void orc_repeats_free(orc_partition_t *p);

void orc_partition_destroy(orc_partition_t *p) {
  if (!p) return;
  for (unsigned int i = 0; i < p->tips + p->clv_buffers; ++i) free(p->clv[i]);
  free(p->clv);
  for (unsigned int i = 0; i < p->scale_buffers; ++i) free(p->scaler[i]);
  free(p->scaler);
  for (unsigned int i = 0; i < p->prob_matrices; ++i) free(p->pmatrix[i]);
  free(p->pmatrix);
  for (unsigned int i = 0; i < p->rate_matrices; ++i) {
    free(p->subst[i]);
    free(p->freqs[i]);
  }
  free(p->subst); free(p->freqs); free(p->rates); free(p->rate_weights);
  free(p->prop_invar); free(p->pattern_weights);
  orc_repeats_free(p);
  free(p);
}

/* Appendix A3: a tip is a full 0/1 CLV (the reference sets no PATTERN_TIP
 * attribute, src/model.cpp:145-157); entry j = bit j of map[char]. */
int orc_set_tip_states(orc_partition_t *p, unsigned int tip_index,
                       const uint64_t *map, const char *sequence) {
  double *clv = p->clv[tip_index];
  orc_repeats_free(p);   /* (tip classes are derived from the tip CLVs) */
  for (unsigned int s = 0; s < p->sites; ++s) {
    uint64_t st = map[(unsigned char)sequence[s]];
    if (!st) return ORC_FAILURE;
    for (unsigned int r = 0; r < p->rate_cats; ++r)
      for (unsigned int j = 0; j < p->states; ++j)
        clv[((size_t)s * p->rate_cats + r) * p->states + j] =
            (double)((st >> j) & 1u);
  }
  return ORC_SUCCESS;
}

void orc_set_tip_clv(orc_partition_t *p, unsigned int tip_index,
                     const double *v) {
  double *clv = p->clv[tip_index];
  for (unsigned int s = 0; s < p->sites; ++s)
    for (unsigned int r = 0; r < p->rate_cats; ++r)
      for (unsigned int j = 0; j < p->states; ++j)
        clv[((size_t)s * p->rate_cats + r) * p->states + j] =
            v[(size_t)s * p->states + j];
}

void orc_set_pattern_weights(orc_partition_t *p, const unsigned int *w) {
  memcpy(p->pattern_weights, w, sizeof(unsigned int) * p->sites);
}
void orc_set_subst_params(orc_partition_t *p, unsigned int idx, const double *v) {
  memcpy(p->subst[idx], v, sizeof(double) * (p->states * p->states - p->states));
}
void orc_set_frequencies(orc_partition_t *p, unsigned int idx, const double *f) {
  memcpy(p->freqs[idx], f, sizeof(double) * p->states);
}
void orc_set_category_rates(orc_partition_t *p, const double *r) {
  memcpy(p->rates, r, sizeof(double) * p->rate_cats);
}
void orc_set_category_weights(orc_partition_t *p, const double *w) {
  memcpy(p->rate_weights, w, sizeof(double) * p->rate_cats);
}

/* corax_msa_empirical_frequencies (src/model.cpp:329): each tip character
 * spreads one (weighted) count evenly over the states it is compatible with. */
double *orc_msa_empirical_frequencies(orc_partition_t *p) {
  unsigned int K = p->states;
  double *f = (double *)calloc(K, sizeof(double));
  double total = 0.0;
  for (unsigned int s = 0; s < p->sites; ++s) total += p->pattern_weights[s];
  for (unsigned int t = 0; t < p->tips; ++t) {
    const double *clv = p->clv[t];
    for (unsigned int s = 0; s < p->sites; ++s) {
      const double *c = clv + (size_t)s * p->rate_cats * K;
      double sum = 0.0;
      for (unsigned int j = 0; j < K; ++j) sum += c[j];
      for (unsigned int j = 0; j < K; ++j)
        f[j] += p->pattern_weights[s] * c[j] / sum;
    }
  }
  for (unsigned int j = 0; j < K; ++j) f[j] /= total * p->tips;
  return f;
}

/* ---- discrete gamma (Appendix A6; Yang 1994) ----------------------------- */

/* regularised lower incomplete gamma P(a, x) */
static double inc_gamma_p(double a, double x) {
  if (x <= 0.0) return 0.0;
  double lg = lgamma(a);
  if (x < a + 1.0) { /* series */
    double ap = a, sum = 1.0 / a, del = sum;
    for (int n = 0; n < 2000; ++n) {
      ap += 1.0;
      del *= x / ap;
      sum += del;
      if (fabs(del) < fabs(sum) * 1e-17) break;
    }
    return sum * exp(-x + a * log(x) - lg);
  }
  /* Lentz continued fraction for Q(a,x) */
  double tiny = 1e-300;
  double b = x + 1.0 - a, c = 1.0 / tiny, d = 1.0 / b, h = d;
  for (int i = 1; i < 2000; ++i) {
    double an = -i * (i - a);
    b += 2.0;
    d = an * d + b; if (fabs(d) < tiny) d = tiny;
    c = b + an / c; if (fabs(c) < tiny) c = tiny;
    d = 1.0 / d;
    double del = d * c;
    h *= del;
    if (fabs(del - 1.0) < 1e-17) break;
  }
  return 1.0 - exp(-x + a * log(x) - lg) * h;
}

/* quantile of Gamma(shape a, rate b): bisection + Newton polish */
static double gamma_quantile(double prob, double a, double b) {
  double lo = 0.0, hi = a > 1.0 ? a : 1.0;
  while (inc_gamma_p(a, hi) < prob) hi *= 2.0;
  for (int i = 0; i < 200; ++i) {
    double mid = 0.5 * (lo + hi);
    if (inc_gamma_p(a, mid) < prob) lo = mid; else hi = mid;
    if (hi - lo <= 1e-16 * hi) break;
  }
  double x = 0.5 * (lo + hi);
  for (int i = 0; i < 3; ++i) { /* Newton on P(a,x) - prob */
    double pdf = exp(-x + (a - 1.0) * log(x) - lgamma(a));
    if (pdf <= 0.0) break;
    double nx = x - (inc_gamma_p(a, x) - prob) / pdf;
    if (nx > lo && nx < hi) x = nx;
  }
  return x / b;
}

int orc_compute_gamma_cats(double alpha, unsigned int cats, double *out,
                           int mode) {
  if (alpha <= 0.0 || cats < 1) return ORC_FAILURE;
  if (cats == 1) { out[0] = 1.0; return ORC_SUCCESS; }
  double beta = alpha, factor = alpha / beta * cats;
  if (mode == ORC_GAMMA_RATES_MEDIAN) {
    double t = 0.0;
    for (unsigned int i = 0; i < cats; ++i) {
      out[i] = gamma_quantile((2.0 * i + 1.0) / (2.0 * cats), alpha, beta);
      t += out[i];
    }
    for (unsigned int i = 0; i < cats; ++i) out[i] *= factor / t;
  } else {
    double *g = (double *)malloc(sizeof(double) * cats);
    for (unsigned int i = 0; i + 1 < cats; ++i)
      g[i] = gamma_quantile((i + 1.0) / cats, alpha, beta);
    for (unsigned int i = 0; i + 1 < cats; ++i)
      g[i] = inc_gamma_p(alpha + 1.0, g[i] * beta);
    out[0] = g[0] * factor;
    out[cats - 1] = (1.0 - g[cats - 2]) * factor;
    for (unsigned int i = 1; i + 1 < cats; ++i) out[i] = (g[i] - g[i - 1]) * factor;
    free(g);
  }
  return ORC_SUCCESS;
}

/* ---- rate matrix and P(t) ------------------------------------------------ */

/* Appendix A1 (UNVERIFIED vs coraxlib's CORAX_ATTRIB_NONREV build): the K*K-K
 * parameters are the off-diagonals in row-major order; Q_ij = s_ij * pi_j;
 * rows sum to zero; Q is scaled to one expected substitution per unit time
 * under pi.  Isolated here so the convention is a one-function swap. */
static void build_q(const orc_partition_t *p, unsigned int idx, double *q) {
  unsigned int K = p->states, k = 0;
  const double *s = p->subst[idx], *f = p->freqs[idx];
  for (unsigned int i = 0; i < K; ++i) {
    double row = 0.0;
    for (unsigned int j = 0; j < K; ++j) {
      if (i == j) continue;
      q[i * K + j] = s[k++] * f[j];
      row += q[i * K + j];
    }
    q[i * K + i] = -row;
  }
  double mean = 0.0;
  for (unsigned int i = 0; i < K; ++i) mean -= f[i] * q[i * K + i];
  for (unsigned int i = 0; i < K * K; ++i) q[i] /= mean;
}

void orc_get_qmatrix(const orc_partition_t *p, unsigned int idx, double *out) {
  build_q(p, idx, out);
}

static void matmul(const double *a, const double *b, double *c, unsigned int n) {
  for (unsigned int i = 0; i < n; ++i)
    for (unsigned int j = 0; j < n; ++j) {
      double s = 0.0;
      for (unsigned int k = 0; k < n; ++k) s += a[i * n + k] * b[k * n + j];
      c[i * n + j] = s;
    }
}

/* exp(A): scale to ||A||_1 <= 1/4, Taylor to convergence, square back. */
void orc_expm(const double *a, unsigned int n, double *out) {
  size_t nn = (size_t)n * n;
  double *x = (double *)malloc(sizeof(double) * nn * 3);
  double *term = x + nn, *tmp = x + 2 * nn;
  double norm = 0.0;
  for (unsigned int j = 0; j < n; ++j) {
    double cs = 0.0;
    for (unsigned int i = 0; i < n; ++i) cs += fabs(a[i * n + j]);
    if (cs > norm) norm = cs;
  }
  int s = 0;
  double scale = 1.0;
  while (norm * scale > 0.25) { scale *= 0.5; ++s; }
  for (size_t i = 0; i < nn; ++i) x[i] = a[i] * scale;
  for (size_t i = 0; i < nn; ++i) { out[i] = 0.0; term[i] = 0.0; }
  for (unsigned int i = 0; i < n; ++i) { out[i * n + i] = 1.0; term[i * n + i] = 1.0; }
  for (int k = 1; k <= 40; ++k) {
    matmul(term, x, tmp, n);
    double tn = 0.0;
    for (size_t i = 0; i < nn; ++i) {
      term[i] = tmp[i] / k;
      out[i] += term[i];
      if (fabs(term[i]) > tn) tn = fabs(term[i]);
    }
    if (tn < 1e-30) break;
  }
  for (int k = 0; k < s; ++k) {
    matmul(out, out, tmp, n);
    memcpy(out, tmp, sizeof(double) * nn);
  }
  free(x);
}

/* Appendix A2: P_r(t) = exp(Q * rate_r * t / (1 - p_inv)); p_inv is always 0
 * in the reference (src/model.cpp:292-300). Negative round-off is clamped. */
int orc_update_prob_matrices(orc_partition_t *p,
                             const unsigned int *params_indices,
                             const unsigned int *matrix_indices,
                             const double *branch_lengths,
                             unsigned int count) {
  unsigned int K = p->states;
  double *q = (double *)malloc(sizeof(double) * K * K * 2);
  double *a = q + K * K;
  for (unsigned int m = 0; m < count; ++m) {
    if (matrix_indices[m] >= p->prob_matrices || !(branch_lengths[m] >= 0.0)) {
      free(q);
      return ORC_FAILURE;
    }
    double *pm = p->pmatrix[matrix_indices[m]];
    for (unsigned int r = 0; r < p->rate_cats; ++r) {
      unsigned int pi = params_indices[r];
      build_q(p, pi, q);
      double f = p->rates[r] * branch_lengths[m] / (1.0 - p->prop_invar[pi]);
      for (unsigned int i = 0; i < K * K; ++i) a[i] = q[i] * f;
      double *out = pm + (size_t)r * K * K;
      orc_expm(a, K, out);
      for (unsigned int i = 0; i < K * K; ++i)
        if (out[i] < 0.0) out[i] = 0.0;
    }
  }
  free(q);
  return ORC_SUCCESS;
}

/* ---- CLV update (row a1 of SURVEY 8a; Appendix A4) ----------------------- */

static void update_one(orc_partition_t *p, const orc_operation_t *op) {
  unsigned int K = p->states, R = p->rate_cats, S = p->sites;
  double *parent = p->clv[op->parent_clv_index];
  const double *left = p->clv[op->child1_clv_index];
  const double *right = p->clv[op->child2_clv_index];
  const double *lm = p->pmatrix[op->child1_matrix_index];
  const double *rm = p->pmatrix[op->child2_matrix_index];
  unsigned int *psc = op->parent_scaler_index == ORC_SCALE_BUFFER_NONE
                          ? NULL : p->scaler[op->parent_scaler_index];
  const unsigned int *lsc = op->child1_scaler_index == ORC_SCALE_BUFFER_NONE
                                ? NULL : p->scaler[op->child1_scaler_index];
  const unsigned int *rsc = op->child2_scaler_index == ORC_SCALE_BUFFER_NONE
                                ? NULL : p->scaler[op->child2_scaler_index];
  /* parent scaler starts as the sum of the children's */
  if (psc)
    for (unsigned int s = 0; s < S; ++s)
      psc[s] = (lsc ? lsc[s] : 0u) + (rsc ? rsc[s] : 0u);

  size_t span = (size_t)R * K;
  for (unsigned int s = 0; s < S; ++s) {
    double *pc = parent + s * span;
    const double *lc = left + s * span, *rc = right + s * span;
    int scaling = psc ? 1 : 0;
    for (unsigned int r = 0; r < R; ++r) {
      const double *lmat = lm + (size_t)r * K * K, *rmat = rm + (size_t)r * K * K;
      for (unsigned int i = 0; i < K; ++i) {
        double ta = 0.0, tb = 0.0;
        for (unsigned int j = 0; j < K; ++j) {
          ta += lmat[i * K + j] * lc[r * K + j];
          tb += rmat[i * K + j] * rc[r * K + j];
        }
        double v = ta * tb;
        pc[r * K + i] = v;
        scaling = scaling && (v < ORC_SCALE_THRESHOLD);
      }
    }
    if (scaling) {
      for (size_t i = 0; i < span; ++i) pc[i] *= ORC_SCALE_FACTOR;
      psc[s] += 1;
    }
  }
}

void orc_update_clvs(orc_partition_t *p, const orc_operation_t *ops,
                     unsigned int count) {
  for (unsigned int i = 0; i < count; ++i) update_one(p, &ops[i]);
}

/* ---- the same update with 256-bit vectors, 4 states only ------------------
 * What the reference links for nucleotide data is coraxlib's AVX2 kernel
 * (src/model.cpp:145-155 selects CORAX_ATTRIB_ARCH_AVX2 at run time); bench.py's
 * cpu_baseline leg times THIS loop so that the stated baseline is a vectorised
 * one.  One vector = the four parent states; separate multiply and add in the
 * scalar loop's order (no FMA: built with -ffp-contract=off), so every CLV entry
 * and scaler is bit-identical to update_one's (tests/test_oracle_golden.py). */
#if defined(__AVX2__)
#include <immintrin.h>
static void update_one_avx2(orc_partition_t *p, const orc_operation_t *op) {
  unsigned int R = p->rate_cats, S = p->sites;
  double *parent = p->clv[op->parent_clv_index];
  const double *left = p->clv[op->child1_clv_index];
  const double *right = p->clv[op->child2_clv_index];
  const double *lm = p->pmatrix[op->child1_matrix_index];
  const double *rm = p->pmatrix[op->child2_matrix_index];
  unsigned int *psc = op->parent_scaler_index == ORC_SCALE_BUFFER_NONE
                          ? NULL : p->scaler[op->parent_scaler_index];
  const unsigned int *lsc = op->child1_scaler_index == ORC_SCALE_BUFFER_NONE
                                ? NULL : p->scaler[op->child1_scaler_index];
  const unsigned int *rsc = op->child2_scaler_index == ORC_SCALE_BUFFER_NONE
                                ? NULL : p->scaler[op->child2_scaler_index];
  /* columns of the P-matrices: lcol[r][j] = (P[0][j], P[1][j], P[2][j], P[3][j]) */
  __m256d *lcol = (__m256d *)aligned_alloc(32, sizeof(__m256d) * 8 * R);
  __m256d *rcol = lcol + 4 * R;
  for (unsigned int r = 0; r < R; ++r)
    for (unsigned int j = 0; j < 4; ++j) {
      const double *a = lm + (size_t)r * 16, *b = rm + (size_t)r * 16;
      lcol[r * 4 + j] = _mm256_set_pd(a[12 + j], a[8 + j], a[4 + j], a[j]);
      rcol[r * 4 + j] = _mm256_set_pd(b[12 + j], b[8 + j], b[4 + j], b[j]);
    }
  const __m256d thr = _mm256_set1_pd(ORC_SCALE_THRESHOLD), fac = _mm256_set1_pd(ORC_SCALE_FACTOR);
  size_t span = (size_t)R * 4;
  for (unsigned int s = 0; s < S; ++s) {
    double *pc = parent + s * span;
    const double *lc = left + s * span, *rc = right + s * span;
    int all_small = 0xF;
    for (unsigned int r = 0; r < R; ++r) {
      /* ((0 + m0 c0) + m1 c1) + m2 c2) + m3 c3, as the scalar loop adds them */
      __m256d ta = _mm256_setzero_pd(), tb = _mm256_setzero_pd();
      for (unsigned int j = 0; j < 4; ++j) {
        ta = _mm256_add_pd(ta, _mm256_mul_pd(lcol[r * 4 + j], _mm256_broadcast_sd(lc + r * 4 + j)));
        tb = _mm256_add_pd(tb, _mm256_mul_pd(rcol[r * 4 + j], _mm256_broadcast_sd(rc + r * 4 + j)));
      }
      const __m256d v = _mm256_mul_pd(ta, tb);
      _mm256_storeu_pd(pc + r * 4, v);
      all_small &= _mm256_movemask_pd(_mm256_cmp_pd(v, thr, _CMP_LT_OQ));
    }
    if (psc) {
      unsigned int sc = (lsc ? lsc[s] : 0u) + (rsc ? rsc[s] : 0u);
      if (all_small == 0xF) {
        for (unsigned int r = 0; r < R; ++r)
          _mm256_storeu_pd(pc + r * 4, _mm256_mul_pd(_mm256_loadu_pd(pc + r * 4), fac));
        sc += 1;
      }
      psc[s] = sc;
    }
  }
  free(lcol);
}
#endif

int orc_update_clvs_avx2(orc_partition_t *p, const orc_operation_t *ops, unsigned int count) {
#if defined(__AVX2__)
  if (p->states != 4) return ORC_FAILURE;
  for (unsigned int i = 0; i < count; ++i) update_one_avx2(p, &ops[i]);
  return ORC_SUCCESS;
#else
  (void)p; (void)ops; (void)count;
  return ORC_FAILURE;
#endif
}


/* ---- subtree site repeats (CORAX_ATTRIB_SITE_REPEATS) ----------------------
 * The reference switches coraxlib's site repeats on for every 4-state partition
 * (src/model.cpp:145-149): two alignment columns that show the same characters at
 * all tips below a node have the same CLV there, so the node's CLV is computed
 * once per CLASS of columns and the columns index into it.  This restates the
 * published scheme (Kobert, Stamatakis, Flouri 2017, "Efficient detection of
 * repeating sites to accelerate phylogenetic likelihood calculations"; libpll-2's
 * pll_update_repeats): a node's class of a column is the pair (class at child 1,
 * class at child 2), numbered in order of first appearance through a lookup table
 * of n1 x n2 entries; a node whose table would pass PLL_REPEATS_LOOKUP_SIZE
 * (2 000 000 entries) keeps one class per column.  CLVs and scalers are stored
 * per class (at the front of the ordinary buffers); tips get a small per-code
 * table.  The arithmetic per class is update_one's, so every value -- and the
 * log-likelihood -- is bit-identical to the plain loop's (tests/test_oracle_golden.py).
 *
 * This is bench.py's HONEST CPU comparator (cpu_baseline.with_site_repeats): what
 * one `rd` rank really executes per evaluation, class bookkeeping included.  A
 * partition driven through these two calls holds per-class buffers: do not mix
 * them with the plain calls inside one traversal. */
#define ORC_REPEATS_LOOKUP_SIZE 2000000u

typedef struct {
  unsigned int *site_id;       /* [S] class of every column */
  unsigned int *rep1, *rep2;   /* [n] per class: the children's classes (inner nodes) */
  unsigned int n;              /* classes */
} orc_rep_node;

struct orc_repeats {
  orc_rep_node *node;          /* tips + clv_buffers */
  double      **tipclv;        /* [tips] per-class tip CLVs [n][R][K] */
  unsigned int *lookup;        /* ORC_REPEATS_LOOKUP_SIZE entries, all 0 between calls */
  unsigned long long class_ops, plain_ops;   /* classes computed / columns a plain loop would have */
  /* memory the traversal moves (orc_repeats_bytes): what it WRITES (the parent's class CLVs and
   * scalers, its column -> class array, the two class -> child-class arrays); what it MUST read at
   * least once (the children's column -> class arrays, every class CLV of an inner child once);
   * what it reads if no child class is ever found in a cache (two child CLVs per parent class) */
  unsigned long long bytes_written, bytes_read_once, bytes_read_every;
};

static void orc_repeats_free(orc_partition_t *p) {
  struct orc_repeats *r = p->rep;
  if (!r) return;
  for (unsigned int i = 0; i < p->tips + p->clv_buffers; ++i) {
    free(r->node[i].site_id); free(r->node[i].rep1); free(r->node[i].rep2);
  }
  for (unsigned int i = 0; i < p->tips; ++i) free(r->tipclv[i]);
  free(r->node); free(r->tipclv); free(r->lookup); free(r);
  p->rep = NULL;
}

static struct orc_repeats *orc_repeats_get(orc_partition_t *p) {
  if (p->rep) return p->rep;
  const unsigned int K = p->states, R = p->rate_cats, S = p->sites, N = p->tips + p->clv_buffers;
  struct orc_repeats *r = (struct orc_repeats *)calloc(1, sizeof(*r));
  r->node = (orc_rep_node *)calloc(N, sizeof(orc_rep_node));
  r->tipclv = (double **)calloc(p->tips ? p->tips : 1, sizeof(double *));
  r->lookup = (unsigned int *)calloc(ORC_REPEATS_LOOKUP_SIZE, sizeof(unsigned int));
  for (unsigned int i = 0; i < N; ++i) {
    r->node[i].site_id = (unsigned int *)calloc(S ? S : 1, sizeof(unsigned int));
    if (i >= p->tips) {
      r->node[i].rep1 = (unsigned int *)calloc(S ? S : 1, sizeof(unsigned int));
      r->node[i].rep2 = (unsigned int *)calloc(S ? S : 1, sizeof(unsigned int));
    }
  }
  /* tips: one class per distinct state set (the 0/1 vector of rate 0 read as a bit mask);
   * more than 2^20 distinct sets (K > 20 with wild data) cannot happen for K <= 20 */
  const size_t span = (size_t)R * K;
  for (unsigned int t = 0; t < p->tips; ++t) {
    orc_rep_node *nd = &r->node[t];
    unsigned int *first = (unsigned int *)calloc(S ? S : 1, sizeof(unsigned int));   /* class -> a column that shows it */
    unsigned long long *mask_of = (unsigned long long *)calloc(S ? S : 1, sizeof(unsigned long long));
    nd->n = 0;
    for (unsigned int s = 0; s < S; ++s) {
      unsigned long long mask = 0;
      for (unsigned int j = 0; j < K && j < 64; ++j)
        if (p->clv[t][s * span + j] != 0.0) mask |= 1ull << j;
      unsigned int c = 0;
      for (; c < nd->n; ++c)   /* (a handful of classes: linear search) */
        if (mask_of[c] == mask) break;
      if (c == nd->n) { mask_of[c] = mask; first[c] = s; ++nd->n; }
      nd->site_id[s] = c;
    }
    r->tipclv[t] = (double *)calloc((size_t)(nd->n ? nd->n : 1) * span, sizeof(double));
    for (unsigned int c = 0; c < nd->n; ++c)
      memcpy(r->tipclv[t] + c * span, p->clv[t] + (size_t)first[c] * span, span * sizeof(double));
    free(first); free(mask_of);
  }
  p->rep = r;
  return r;
}

/* classes of the parent from its children's (libpll-2: pll_update_repeats) */
static void repeats_classes(orc_partition_t *p, struct orc_repeats *r, const orc_operation_t *op) {
  const unsigned int S = p->sites;
  orc_rep_node *pn = &r->node[op->parent_clv_index];
  const orc_rep_node *a = &r->node[op->child1_clv_index], *b = &r->node[op->child2_clv_index];
  if ((unsigned long long)a->n * b->n > ORC_REPEATS_LOOKUP_SIZE) {   /* no table: one class per column */
    pn->n = S;
    for (unsigned int s = 0; s < S; ++s) {
      pn->site_id[s] = s; pn->rep1[s] = a->site_id[s]; pn->rep2[s] = b->site_id[s];
    }
    return;
  }
  unsigned int n = 0;
  for (unsigned int s = 0; s < S; ++s) {
    const unsigned int key = a->site_id[s] * b->n + b->site_id[s];
    unsigned int id = r->lookup[key];
    if (!id) {
      pn->rep1[n] = a->site_id[s]; pn->rep2[n] = b->site_id[s];
      id = r->lookup[key] = ++n;
    }
    pn->site_id[s] = id - 1;
  }
  for (unsigned int c = 0; c < n; ++c) r->lookup[pn->rep1[c] * b->n + pn->rep2[c]] = 0;   /* leave the table clean */
  pn->n = n;
}

static void update_one_repeats(orc_partition_t *p, struct orc_repeats *r, const orc_operation_t *op, int avx2) {
  const unsigned int K = p->states, R = p->rate_cats;
  repeats_classes(p, r, op);
  const orc_rep_node *pn = &r->node[op->parent_clv_index];
  double *parent = p->clv[op->parent_clv_index];
  const double *left = op->child1_clv_index < p->tips ? r->tipclv[op->child1_clv_index] : p->clv[op->child1_clv_index];
  const double *right = op->child2_clv_index < p->tips ? r->tipclv[op->child2_clv_index] : p->clv[op->child2_clv_index];
  const double *lm = p->pmatrix[op->child1_matrix_index];
  const double *rm = p->pmatrix[op->child2_matrix_index];
  unsigned int *psc = op->parent_scaler_index == ORC_SCALE_BUFFER_NONE ? NULL : p->scaler[op->parent_scaler_index];
  const unsigned int *lsc = op->child1_scaler_index == ORC_SCALE_BUFFER_NONE ? NULL : p->scaler[op->child1_scaler_index];
  const unsigned int *rsc = op->child2_scaler_index == ORC_SCALE_BUFFER_NONE ? NULL : p->scaler[op->child2_scaler_index];
  const size_t span = (size_t)R * K;
  r->class_ops += pn->n;
  r->plain_ops += p->sites;
  {
    const unsigned long long clv_b = (unsigned long long)R * K * sizeof(double), S_ = p->sites;
    const unsigned long long n1 = op->child1_clv_index < p->tips ? 0 : r->node[op->child1_clv_index].n;
    const unsigned long long n2 = op->child2_clv_index < p->tips ? 0 : r->node[op->child2_clv_index].n;
    r->bytes_written += pn->n * (clv_b + (psc ? 4u : 0u) + 8u) + 4u * S_;
    r->bytes_read_once += 8u * S_ + (n1 + n2) * (clv_b + 4u);
    r->bytes_read_every += 8u * S_ + (unsigned long long)pn->n * 2u * (clv_b + 4u);
  }
#if defined(__AVX2__)
  if (avx2 && K == 4) {
    __m256d *lcol = (__m256d *)aligned_alloc(32, sizeof(__m256d) * 8 * R);
    __m256d *rcol = lcol + 4 * R;
    for (unsigned int q = 0; q < R; ++q)
      for (unsigned int j = 0; j < 4; ++j) {
        const double *a = lm + (size_t)q * 16, *b = rm + (size_t)q * 16;
        lcol[q * 4 + j] = _mm256_set_pd(a[12 + j], a[8 + j], a[4 + j], a[j]);
        rcol[q * 4 + j] = _mm256_set_pd(b[12 + j], b[8 + j], b[4 + j], b[j]);
      }
    const __m256d thr = _mm256_set1_pd(ORC_SCALE_THRESHOLD), fac = _mm256_set1_pd(ORC_SCALE_FACTOR);
    for (unsigned int c = 0; c < pn->n; ++c) {
      double *pc = parent + c * span;
      const double *lc = left + pn->rep1[c] * span, *rc = right + pn->rep2[c] * span;
      int all_small = 0xF;
      for (unsigned int q = 0; q < R; ++q) {
        __m256d ta = _mm256_setzero_pd(), tb = _mm256_setzero_pd();
        for (unsigned int j = 0; j < 4; ++j) {
          ta = _mm256_add_pd(ta, _mm256_mul_pd(lcol[q * 4 + j], _mm256_broadcast_sd(lc + q * 4 + j)));
          tb = _mm256_add_pd(tb, _mm256_mul_pd(rcol[q * 4 + j], _mm256_broadcast_sd(rc + q * 4 + j)));
        }
        const __m256d v = _mm256_mul_pd(ta, tb);
        _mm256_storeu_pd(pc + q * 4, v);
        all_small &= _mm256_movemask_pd(_mm256_cmp_pd(v, thr, _CMP_LT_OQ));
      }
      if (psc) {
        unsigned int sc = (lsc ? lsc[pn->rep1[c]] : 0u) + (rsc ? rsc[pn->rep2[c]] : 0u);
        if (all_small == 0xF) {
          for (unsigned int q = 0; q < R; ++q)
            _mm256_storeu_pd(pc + q * 4, _mm256_mul_pd(_mm256_loadu_pd(pc + q * 4), fac));
          sc += 1;
        }
        psc[c] = sc;
      }
    }
    free(lcol);
    return;
  }
#else
  (void)avx2;
#endif
  for (unsigned int c = 0; c < pn->n; ++c) {   /* update_one's loop body, per class */
    double *pc = parent + c * span;
    const double *lc = left + pn->rep1[c] * span, *rc = right + pn->rep2[c] * span;
    int scaling = psc ? 1 : 0;
    if (psc) psc[c] = (lsc ? lsc[pn->rep1[c]] : 0u) + (rsc ? rsc[pn->rep2[c]] : 0u);
    for (unsigned int q = 0; q < R; ++q) {
      const double *lmat = lm + (size_t)q * K * K, *rmat = rm + (size_t)q * K * K;
      for (unsigned int i = 0; i < K; ++i) {
        double ta = 0.0, tb = 0.0;
        for (unsigned int j = 0; j < K; ++j) {
          ta += lmat[i * K + j] * lc[q * K + j];
          tb += rmat[i * K + j] * rc[q * K + j];
        }
        const double v = ta * tb;
        pc[q * K + i] = v;
        scaling = scaling && (v < ORC_SCALE_THRESHOLD);
      }
    }
    if (scaling) {
      for (size_t i = 0; i < span; ++i) pc[i] *= ORC_SCALE_FACTOR;
      psc[c] += 1;
    }
  }
}

void orc_update_clvs_repeats(orc_partition_t *p, const orc_operation_t *ops, unsigned int count, int avx2) {
  struct orc_repeats *r = orc_repeats_get(p);
  for (unsigned int i = 0; i < count; ++i) update_one_repeats(p, r, &ops[i], avx2);
}

/* orc_compute_root_loglikelihood on the per-class buffers orc_update_clvs_repeats left:
 * the same terms in the same (column) order */
double orc_compute_root_loglikelihood_repeats(orc_partition_t *p, unsigned int clv_index, int scaler_index,
                                              const unsigned int *freqs_indices) {
  struct orc_repeats *r = orc_repeats_get(p);
  const unsigned int K = p->states, R = p->rate_cats;
  const orc_rep_node *nd = &r->node[clv_index];
  const double *base = p->clv[clv_index];
  const unsigned int *sc = scaler_index == ORC_SCALE_BUFFER_NONE ? NULL : p->scaler[scaler_index];
  const double log_thr = log(ORC_SCALE_THRESHOLD);
  const size_t span = (size_t)R * K;
  double logl = 0.0;
  for (unsigned int s = 0; s < p->sites; ++s) {
    const unsigned int c = nd->site_id[s];
    const double *clv = base + c * span;
    double term = 0.0;
    for (unsigned int q = 0; q < R; ++q) {
      const double *f = p->freqs[freqs_indices[q]];
      double tr = 0.0;
      for (unsigned int k = 0; k < K; ++k) tr += clv[k] * f[k];
      term += tr * p->rate_weights[q];
      clv += K;
    }
    term = log(term);
    if (sc && sc[c]) term += sc[c] * log_thr;
    term *= p->pattern_weights[s];
    logl += term;
  }
  return logl;
}

/* bytes the site-repeats traversals have moved since the partition was created:
 * out[0] written, out[1] read at least once (compulsory), out[2] read if nothing is ever cached */
void orc_repeats_bytes(const orc_partition_t *p, double out[3]) {
  out[0] = p->rep ? (double)p->rep->bytes_written : 0.0;
  out[1] = p->rep ? (double)p->rep->bytes_read_once : 0.0;
  out[2] = p->rep ? (double)p->rep->bytes_read_every : 0.0;
}

/* classes computed / columns a plain loop would have computed, since the partition was created */
double orc_repeats_ratio(const orc_partition_t *p) {
  return p->rep && p->rep->plain_ops ? (double)p->rep->class_ops / (double)p->rep->plain_ops : 1.0;
}

/* ---- root log-likelihood (row a3; Appendix A5) --------------------------- */

double orc_compute_root_loglikelihood(orc_partition_t *p,
                                      unsigned int clv_index,
                                      int scaler_index,
                                      const unsigned int *freqs_indices,
                                      double *persite_lnl) {
  unsigned int K = p->states, R = p->rate_cats;
  const double *clv = p->clv[clv_index];
  const unsigned int *sc =
      scaler_index == ORC_SCALE_BUFFER_NONE ? NULL : p->scaler[scaler_index];
  double logl = 0.0;
  const double log_thr = log(ORC_SCALE_THRESHOLD);
  for (unsigned int s = 0; s < p->sites; ++s) {
    double term = 0.0;
    for (unsigned int r = 0; r < R; ++r) {
      const double *f = p->freqs[freqs_indices[r]];
      double tr = 0.0;
      for (unsigned int k = 0; k < K; ++k) tr += clv[k] * f[k];
      term += tr * p->rate_weights[r];
      clv += K;
    }
    term = log(term);
    if (sc && sc[s]) term += sc[s] * log_thr;
    term *= p->pattern_weights[s];
    if (persite_lnl) persite_lnl[s] = term;
    logl += term;
  }
  return logl;
}

const double *orc_get_clv(const orc_partition_t *p, unsigned int i) { return p->clv[i]; }
const unsigned int *orc_get_scaler(const orc_partition_t *p, unsigned int i) { return p->scaler[i]; }
const double *orc_get_pmatrix(const orc_partition_t *p, unsigned int i) { return p->pmatrix[i]; }
unsigned int orc_states(const orc_partition_t *p) { return p->states; }
unsigned int orc_rate_cats(const orc_partition_t *p) { return p->rate_cats; }
unsigned int orc_sites(const orc_partition_t *p) { return p->sites; }

/* ---- STREAM triad on the host's cores (bench.py: cpu_baseline.one_socket_bandwidth_bound) ----
 * a[i] = b[i] + s * c[i] over three arrays of `doubles` elements per thread, each thread pinned to
 * cpus[t] (if cpus is not NULL) and touching its own arrays first; all threads run for `seconds`
 * between two barriers.  Returns GB/s by STREAM's count (24 bytes per element: two reads and a
 * write; the write-allocate read is not counted), summed over the threads. */
#include <pthread.h>
#include <sched.h>
#include <time.h>
struct orc_triad_arg {
  int cpu;
  size_t n;
  double seconds, *a, *b, *c;
  unsigned long long passes;
  double elapsed;
  pthread_barrier_t *bar;
};
static double orc_now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static void *orc_triad_thread(void *arg_) {
  struct orc_triad_arg *x = (struct orc_triad_arg *)arg_;
  if (x->cpu >= 0) {
    cpu_set_t set;
    CPU_ZERO(&set);
    CPU_SET(x->cpu, &set);
    (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set);
  }
  x->a = (double *)aligned_alloc(64, x->n * sizeof(double));
  x->b = (double *)aligned_alloc(64, x->n * sizeof(double));
  x->c = (double *)aligned_alloc(64, x->n * sizeof(double));
  if (!x->a || !x->b || !x->c) { x->passes = 0; pthread_barrier_wait(x->bar); return NULL; }
  for (size_t i = 0; i < x->n; ++i) { x->a[i] = 0.0; x->b[i] = 1.0; x->c[i] = 2.0; }
  pthread_barrier_wait(x->bar);
  const double t0 = orc_now();
  double s = 3.0;
  do {
    double *restrict a = x->a;
    const double *restrict b = x->b, *restrict c = x->c;
    for (size_t i = 0; i < x->n; ++i) a[i] = b[i] + s * c[i];
    s += 1e-9 * a[x->n / 2];   /* (a dependence the compiler cannot drop) */
    ++x->passes;
    x->elapsed = orc_now() - t0;
  } while (x->elapsed < x->seconds);
  free(x->a); free(x->b); free(x->c);
  return NULL;
}
double orc_stream_triad(int threads, const int *cpus, size_t doubles, double seconds) {
  if (threads < 1 || threads > 1024) return 0.0;
  pthread_t th[1024];
  struct orc_triad_arg arg[1024];
  pthread_barrier_t bar;
  pthread_barrier_init(&bar, NULL, (unsigned)threads);
  for (int t = 0; t < threads; ++t) {
    arg[t].cpu = cpus ? cpus[t] : -1; arg[t].n = doubles; arg[t].seconds = seconds;
    arg[t].passes = 0; arg[t].elapsed = 0.0; arg[t].bar = &bar;
    pthread_create(&th[t], NULL, orc_triad_thread, &arg[t]);
  }
  double gbs = 0.0;
  for (int t = 0; t < threads; ++t) {
    pthread_join(th[t], NULL);
    if (arg[t].elapsed > 0.0) gbs += 24.0 * (double)doubles * (double)arg[t].passes / arg[t].elapsed / 1e9;
  }
  pthread_barrier_destroy(&bar);
  return gbs;
}
