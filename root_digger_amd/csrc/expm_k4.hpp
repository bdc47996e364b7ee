// exp(Q t) for one 4x4 rate matrix, everything in registers: scaling and
// squaring with a fixed 16-term Taylor core (||A/2^s||_1 <= 1/4, truncation
// < 1e-19).  ONE definition shared by every kernel that needs a 4-state
// P-matrix (pmatrix_k4_kernel, fused_pmatrix_k4_kernel's twin in
// kernels_fused.hip keeps its own copy of the same steps, root_single_dna_kernel),
// so they all produce the same bits.  `out` is NOT clamped at zero here.
#pragma once

namespace rdamd {

constexpr int kTaylorTerms = 16;

__device__ __forceinline__ void expm_k4(const double *__restrict__ qq, double t, double (&out)[16]) {
  double x[16], term[16], tmp[16];
  double norm = 0.0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    double cs = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      x[i * 4 + j] = qq[i * 4 + j] * t;
      cs += fabs(x[i * 4 + j]);
    }
    norm = fmax(norm, cs);
  }
  int s = 0;
  double scale = 1.0;
  while (norm * scale > 0.25 && s < 60) { scale *= 0.5; ++s; }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    x[i] *= scale;
    term[i] = out[i] = (i % 5 == 0) ? 1.0 : 0.0;
  }
  for (int k = 1; k <= kTaylorTerms; ++k) {
    double inv = 1.0 / (double)k;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        double a = 0.0;
#pragma unroll
        for (int l = 0; l < 4; ++l) a += term[i * 4 + l] * x[l * 4 + j];
        tmp[i * 4 + j] = a * inv;
      }
#pragma unroll
    for (int i = 0; i < 16; ++i) { term[i] = tmp[i]; out[i] += tmp[i]; }
  }
  for (int k = 0; k < s; ++k) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        double a = 0.0;
#pragma unroll
        for (int l = 0; l < 4; ++l) a += out[i * 4 + l] * out[l * 4 + j];
        tmp[i * 4 + j] = a;
      }
#pragma unroll
    for (int i = 0; i < 16; ++i) out[i] = tmp[i];
  }
}

}  // namespace rdamd
