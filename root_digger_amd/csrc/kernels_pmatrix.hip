// Transition-probability matrices on the device.
//
// Replaces corax_update_prob_matrices (called at /root/reference/src/model.cpp
// :367, :432, :842): P[m][r] = exp(Q * rate_r * t_m), layout [m][r][i][j] with
// i = parent state, j = child state.  The reference
// build is non-reversible (CORAX_ATTRIB_NONREV, src/model.cpp:157) so Q may
// have complex eigenvalues; instead of an eigendecomposition the whole branch
// list is exponentiated in one launch by scaling-and-squaring with a fixed
// 16-term Taylor core (||A/2^s||_1 <= 1/4, truncation < 1e-19).
//
// The same launch also fills the tip lookup tables
//   tiptab[m][r][code][i] = sum_j P[m][r][i][j] * bit_j(mask(code))
// which the fused evaluator, the fused root step and the generic K-state kernel
// read instead of expanding a tip into a 0/1 vector (the 4-state traversal
// kernel expands in registers: it is bound by its store stream, not by FMAs).
#include "common.hpp"
#include "expm_k4.hpp"

namespace rdamd {

void build_q_host(unsigned K, const double *subst, const double *freqs, double *q) {
  // off-diagonals row-major, Q_ij = s_ij * pi_j, zero row
  // sums, normalised to one expected substitution per unit time under pi.
  unsigned k = 0;
  for (unsigned i = 0; i < K; ++i) {
    double row = 0.0;
    for (unsigned j = 0; j < K; ++j) {
      if (i == j) continue;
      q[i * K + j] = subst[k++] * freqs[j];
      row += q[i * K + j];
    }
    q[i * K + i] = -row;
  }
  double mean = 0.0;
  for (unsigned i = 0; i < K; ++i) mean -= freqs[i] * q[i * K + i];
  for (unsigned i = 0; i < K * K; ++i) q[i] /= mean;
}


// ---- K = 4: one thread per (matrix, rate), everything in registers ---------
__global__ void __launch_bounds__(64)
pmatrix_k4_kernel(const double *__restrict__ q, const double *__restrict__ rates,
                  const unsigned *__restrict__ params_idx,
                  const unsigned *__restrict__ mat_idx,
                  const double *__restrict__ brlen, unsigned count, unsigned R,
                  double *__restrict__ pmat, double *__restrict__ tiptab,
                  const uint64_t *__restrict__ codemask, unsigned ncodes_cap) {
  unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= count * R) return;
  unsigned m = gid / R, r = gid % R;
  double t = brlen[m] * rates[r];
  const double *qq = q + (size_t)params_idx[r] * 16;
  double out[16];
  expm_k4(qq, t, out);
  size_t slot = (size_t)mat_idx[m] * R + r;
  double *pm = pmat + slot * 16;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    out[i] = out[i] <= 0.0 ? 0.0 : out[i];   // (<=: a -0.0 becomes +0.0 -- the rescale tests read high words)
    pm[i] = out[i];
  }
  double *tt = tiptab + slot * ncodes_cap * 4;
  for (unsigned c = 0; c < ncodes_cap; ++c) {
    uint64_t mask = codemask[c];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      double a = 0.0;
#pragma unroll
      for (int j = 0; j < 4; ++j) a += ((mask >> j) & 1) ? out[i * 4 + j] : 0.0;
      tt[c * 4 + i] = a;
    }
  }
}

// ---- any K <= 64: one workgroup per (matrix, rate), matrices in LDS ---------
__global__ void __launch_bounds__(256)
pmatrix_generic_kernel(const double *__restrict__ q, const double *__restrict__ rates,
                       const unsigned *__restrict__ params_idx,
                       const unsigned *__restrict__ mat_idx,
                       const double *__restrict__ brlen, unsigned R, unsigned K,
                       double *__restrict__ pmat, double *__restrict__ tiptab,
                       const uint64_t *__restrict__ codemask, unsigned ncodes_cap, unsigned operand_rows) {
  extern __shared__ double sm[];
  const unsigned KK = K * K;
  double *x = sm, *term = sm + KK, *out = sm + 2 * KK, *tmp = sm + 3 * KK;
  double *red = sm + 4 * KK;  // [K]
  unsigned m = blockIdx.x / R, r = blockIdx.x % R;
  unsigned tid = threadIdx.x, nt = blockDim.x;
  double t = brlen[m] * rates[r];
  const double *qq = q + (size_t)params_idx[r] * KK;
  for (unsigned e = tid; e < KK; e += nt) x[e] = qq[e] * t;
  __syncthreads();
  for (unsigned j = tid; j < K; j += nt) {
    double cs = 0.0;
    for (unsigned i = 0; i < K; ++i) cs += fabs(x[i * K + j]);
    red[j] = cs;
  }
  __syncthreads();
  double norm = 0.0;
  for (unsigned j = 0; j < K; ++j) norm = fmax(norm, red[j]);
  int s = 0;
  double scale = 1.0;
  while (norm * scale > 0.25 && s < 60) { scale *= 0.5; ++s; }
  __syncthreads();
  for (unsigned e = tid; e < KK; e += nt) {
    x[e] *= scale;
    term[e] = out[e] = (e / K == e % K) ? 1.0 : 0.0;
  }
  __syncthreads();
  for (int k = 1; k <= kTaylorTerms; ++k) {
    double inv = 1.0 / (double)k;
    for (unsigned e = tid; e < KK; e += nt) {
      unsigned i = e / K, j = e % K;
      double a = 0.0;
      for (unsigned l = 0; l < K; ++l) a += term[i * K + l] * x[l * K + j];
      tmp[e] = a * inv;
    }
    __syncthreads();
    for (unsigned e = tid; e < KK; e += nt) { term[e] = tmp[e]; out[e] += tmp[e]; }
    __syncthreads();
  }
  for (int k = 0; k < s; ++k) {
    for (unsigned e = tid; e < KK; e += nt) {
      unsigned i = e / K, j = e % K;
      double a = 0.0;
      for (unsigned l = 0; l < K; ++l) a += out[i * K + l] * out[l * K + j];
      tmp[e] = a;
    }
    __syncthreads();
    for (unsigned e = tid; e < KK; e += nt) out[e] = tmp[e];
    __syncthreads();
  }
  size_t slot = (size_t)mat_idx[m] * R + r;
  double *pm = pmat + slot * KK;
  for (unsigned e = tid; e < KK; e += nt) {
    double v = out[e] <= 0.0 ? 0.0 : out[e];   // (<=: no -0.0 -- the rescale tests read high words)
    out[e] = v;
    pm[e] = v;
  }
  __syncthreads();
  double *tt = tiptab + slot * ncodes_cap * K;
  for (unsigned e = tid; e < ncodes_cap * K; e += nt) {
    unsigned c = e / K, i = operand_rows ? k20_row_state(e % K) : e % K;   // (row order: common.hpp)
    uint64_t mask = codemask[c];
    double a = 0.0;
    for (unsigned j = 0; j < K; ++j) a += ((mask >> j) & 1) ? out[i * K + j] : 0.0;
    tt[e] = a;
  }
}

// Recompute every tip table from the P-matrices already in HBM (used when a
// new ambiguity code shows up after P-matrices were last updated).
__global__ void tiptab_all_kernel(const double *__restrict__ pmat,
                                  double *__restrict__ tiptab,
                                  const uint64_t *__restrict__ codemask,
                                  unsigned K, unsigned ncodes_cap, size_t total, unsigned operand_rows) {
  size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  unsigned i = operand_rows ? k20_row_state((unsigned)(e % K)) : (unsigned)(e % K);
  unsigned c = (e / K) % ncodes_cap;
  size_t slot = e / ((size_t)K * ncodes_cap);
  const double *pm = pmat + slot * K * K + (size_t)i * K;
  uint64_t mask = codemask[c];
  double a = 0.0;
  for (unsigned j = 0; j < K; ++j) a += ((mask >> j) & 1) ? pm[j] : 0.0;
  tiptab[e] = a;
}

hipError_t launch_pmatrix(rdamd_partition *p, const unsigned *d_params_indices,
                          const unsigned *d_matrix_indices,
                          const double *d_branch_lengths, unsigned count) {
  if (count == 0) return hipSuccess;
  unsigned R = p->rate_cats, K = p->states;
  if (K == 4) {
    unsigned total = count * R;
    pmatrix_k4_kernel<<<(total + 63) / 64, 64, 0, p->stream>>>(
        p->d_q, p->d_rates, d_params_indices, d_matrix_indices, d_branch_lengths,
        count, R, p->d_pmat, p->d_tiptab, p->d_codemask, p->ncodes_cap);
  } else {
    size_t lds = (4 * (size_t)K * K + K) * sizeof(double);
    pmatrix_generic_kernel<<<count * R, 256, lds, p->stream>>>(
        p->d_q, p->d_rates, d_params_indices, d_matrix_indices, d_branch_lengths,
        R, K, p->d_pmat, p->d_tiptab, p->d_codemask, p->ncodes_cap, p->mfma_layout ? 1u : 0u);
  }
  return hipGetLastError();
}

hipError_t launch_tiptab_all(rdamd_partition *p) {
  size_t total = (size_t)p->prob_matrices * p->rate_cats * p->ncodes_cap * p->states;
  if (!total) return hipSuccess;
  tiptab_all_kernel<<<(unsigned)((total + 255) / 256), 256, 0, p->stream>>>(
      p->d_pmat, p->d_tiptab, p->d_codemask, p->states, p->ncodes_cap, total, p->mfma_layout ? 1u : 0u);
  return hipGetLastError();
}

}  // namespace rdamd
