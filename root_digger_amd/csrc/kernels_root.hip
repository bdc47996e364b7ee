// Root log-likelihood reduction.
//
// Replaces corax_compute_root_loglikelihood (called at
// /root/reference/src/model.cpp:406, :441, :466):
//   lnL = sum_s w_s * [ log( sum_r omega_r * sum_k pi_k * root[s][r][k] )
//                       + scaler_s * log(2^-256) ]
// (the invariant-sites branch is never active in the
// reference, src/model.cpp:292-300).  The reduction has a fixed shape
// (per-lane strided sums -> wave shuffle tree -> LDS -> one finishing
// workgroup), so a repeated call returns the bit-identical value the
// reference's test demands (test/src/model.cpp:73); no floating-point atomics.
#include "common.hpp"

namespace rdamd {

constexpr unsigned kRootBlocks = 1024;

__device__ inline double block_sum_256(double v, double *lds) {
  // wave tree
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) lds[w] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) r = ((lds[0] + lds[1]) + lds[2]) + lds[3];
  return r;
}

// one lane per (site, rate); R in {1,2,4,8,16}
template <int R>
__global__ void __launch_bounds__(256)
root_lnl_group_kernel(const double *__restrict__ clv, const unsigned *__restrict__ scaler,
                      const double *__restrict__ freqs, const unsigned *__restrict__ fidx,
                      const double *__restrict__ rate_w, const unsigned *__restrict__ pw,
                      unsigned S, unsigned K, double *__restrict__ persite,
                      double *__restrict__ partials) {
  __shared__ double lds[4];
  const size_t total = (size_t)S * R;
  const size_t stride = (size_t)gridDim.x * 256;
  double acc = 0.0;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += stride) {
    const unsigned s = (unsigned)(idx / R), r = (unsigned)(idx % R);
    const double *c = clv + idx * K;
    const double *f = freqs + (size_t)fidx[r] * K;
    double tr = 0.0;
    for (unsigned k = 0; k < K; ++k) tr += c[k] * f[k];
    tr *= rate_w[r];
    // sum the R rate terms in rate order (same order as the reference loop)
    double term = 0.0;
    const int base = (int)(threadIdx.x & 63) & ~(R - 1);
#pragma unroll
    for (int q = 0; q < R; ++q) term += __shfl(tr, base + q);
    if (r == 0) {
      double l = log(term);
      if (scaler) {
        unsigned sc = scaler[s];
        if (sc) l += (double)sc * kLogScaleThreshold;
      }
      l *= (double)pw[s];
      if (persite) persite[s] = l;
      acc += l;
    }
  }
  double b = block_sum_256(acc, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = b;
}

// one lane per site (any R, K)
__global__ void __launch_bounds__(256)
root_lnl_site_kernel(const double *__restrict__ clv, const unsigned *__restrict__ scaler,
                     const double *__restrict__ freqs, const unsigned *__restrict__ fidx,
                     const double *__restrict__ rate_w, const unsigned *__restrict__ pw,
                     unsigned S, unsigned R, unsigned K, double *__restrict__ persite,
                     double *__restrict__ partials) {
  __shared__ double lds[4];
  double acc = 0.0;
  for (unsigned s = blockIdx.x * 256 + threadIdx.x; s < S; s += gridDim.x * 256) {
    const double *c = clv + (size_t)s * R * K;
    double term = 0.0;
    for (unsigned r = 0; r < R; ++r) {
      const double *f = freqs + (size_t)fidx[r] * K;
      double tr = 0.0;
      for (unsigned k = 0; k < K; ++k) tr += c[(size_t)r * K + k] * f[k];
      term += tr * rate_w[r];
    }
    double l = log(term);
    if (scaler) {
      unsigned sc = scaler[s];
      if (sc) l += (double)sc * kLogScaleThreshold;
    }
    l *= (double)pw[s];
    if (persite) persite[s] = l;
    acc += l;
  }
  double b = block_sum_256(acc, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = b;
}

// fixed-order finish: 256 lanes stride over the partials, then the block tree
__global__ void __launch_bounds__(256)
finish_sum_kernel(const double *__restrict__ partials, unsigned n, double *__restrict__ out) {
  __shared__ double lds[4];
  double acc = 0.0;
  for (unsigned i = threadIdx.x; i < n; i += 256) acc += partials[i];
  double b = block_sum_256(acc, lds);
  if (threadIdx.x == 0) *out = b;
}

hipError_t launch_root_lnl(rdamd_partition *p, unsigned clv_index, int scaler_index,
                           const unsigned *d_fidx, double *d_persite, double *d_out) {
  const unsigned S = p->sites, R = p->rate_cats, K = p->states;
  const double *clv = p->d_clv + (size_t)(clv_index - p->tips) * S * R * K;
  const unsigned *sc = scaler_index >= 0 ? p->d_scaler + (size_t)scaler_index * S : nullptr;
  unsigned blocks;
  const bool group = (R == 1 || R == 2 || R == 4 || R == 8 || R == 16);
  if (group) {
    size_t total = (size_t)S * R;
    blocks = (unsigned)((total + 255) / 256);
  } else {
    blocks = (S + 255) / 256;
  }
  if (blocks > kRootBlocks) blocks = kRootBlocks;
  if (blocks == 0) blocks = 1;
#define RDAMD_ROOT_ARGS clv, sc, p->d_freqs, d_fidx, p->d_rate_weights, p->d_pattern_weights
  if (group) {
    switch (R) {
      case 1: root_lnl_group_kernel<1><<<blocks, 256, 0, p->stream>>>(RDAMD_ROOT_ARGS, S, K, d_persite, p->d_partials); break;
      case 2: root_lnl_group_kernel<2><<<blocks, 256, 0, p->stream>>>(RDAMD_ROOT_ARGS, S, K, d_persite, p->d_partials); break;
      case 4: root_lnl_group_kernel<4><<<blocks, 256, 0, p->stream>>>(RDAMD_ROOT_ARGS, S, K, d_persite, p->d_partials); break;
      case 8: root_lnl_group_kernel<8><<<blocks, 256, 0, p->stream>>>(RDAMD_ROOT_ARGS, S, K, d_persite, p->d_partials); break;
      default: root_lnl_group_kernel<16><<<blocks, 256, 0, p->stream>>>(RDAMD_ROOT_ARGS, S, K, d_persite, p->d_partials); break;
    }
  } else {
    root_lnl_site_kernel<<<blocks, 256, 0, p->stream>>>(RDAMD_ROOT_ARGS, S, R, K, d_persite, p->d_partials);
  }
#undef RDAMD_ROOT_ARGS
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  finish_sum_kernel<<<1, 256, 0, p->stream>>>(p->d_partials, blocks, d_out);
  return hipGetLastError();
}

}  // namespace rdamd
