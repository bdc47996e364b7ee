// exp(Q t) for one 4x4 rate matrix: scaling and squaring with a fixed 16-term Taylor core
// (||A/2^s||_1 <= 1/4, truncation < 1e-19).  ONE definition of the arithmetic shared by every
// kernel that needs a 4-state P-matrix (pmatrix_k4_kernel, fused_pmatrix_k4_kernel,
// root_single_dna_kernel), so they all produce the same bits.  Every multiply-add is an
// explicit fma: what the compiler's contraction would pick must not decide whether two
// kernels agree.  `out` is NOT clamped at zero here.
//
// Two forms of the same arithmetic: expm_k4 -- one lane, everything in registers (64 doubles:
// 128 VGPRs) --, and expm_k4_coop16 -- 16 lanes per matrix, a lane per element, rows and
// columns exchanged through lane shuffles (~20 VGPRs) -- for kernels that cannot afford the
// registers: the fused root kernels run beside the fused evaluator and must fit the wave
// slots its waves leave (kernels_root.hip).  Each element goes through the same operations
// in the same order in both.
#pragma once

namespace rdamd {

constexpr int kTaylorTerms = 16;

// squarings for a matrix of 1-norm `norm`: the smallest s with norm / 2^s <= 1/4 (at most 60)
__device__ __forceinline__ int expm_k4_squarings(double norm, double *scale_out) {
  int s = 0;
  double scale = 1.0;
  while (norm * scale > 0.25 && s < 60) { scale *= 0.5; ++s; }
  *scale_out = scale;
  return s;
}

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
  double scale;
  const int s = expm_k4_squarings(norm, &scale);
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
        for (int l = 0; l < 4; ++l) a = __builtin_fma(term[i * 4 + l], x[l * 4 + j], a);
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
        for (int l = 0; l < 4; ++l) a = __builtin_fma(out[i * 4 + l], out[l * 4 + j], a);
        tmp[i * 4 + j] = a;
      }
#pragma unroll
    for (int i = 0; i < 16; ++i) out[i] = tmp[i];
  }
}

// The same for the 16 lanes [16 g, 16 g + 16) of a wave together: lane 16 g + 4 i + j returns
// element (i, j).  qq and t are the group's (every lane of a group passes the same); all 64
// lanes must call (groups may work on different matrices; a group without one passes t = 0).
__device__ __forceinline__ double expm_k4_coop16(const double *__restrict__ qq, double t) {
  const int lane = (int)(threadIdx.x & 63u), base = lane & ~15, e = lane & 15, i = e >> 2, j = e & 3;
  // my column j of x = Q t (the in-register form computes x[i][j] = qq[i][j] * t, then * scale)
  double xc[4];
#pragma unroll
  for (int l = 0; l < 4; ++l) xc[l] = qq[l * 4 + j] * t;
  // column sums in the in-register order (rows 0..3), then their maximum over j = 0..3
  double cs = 0.0;
#pragma unroll
  for (int l = 0; l < 4; ++l) cs += fabs(xc[l]);
  double norm = 0.0;
#pragma unroll
  for (int c = 0; c < 4; ++c) norm = fmax(norm, __shfl(cs, base + c));   // (lane base + c holds column c's sum)
  double scale;
  const int s = expm_k4_squarings(norm, &scale);
#pragma unroll
  for (int l = 0; l < 4; ++l) xc[l] *= scale;
  double term = i == j ? 1.0 : 0.0, out = term;
  for (int k = 1; k <= kTaylorTerms; ++k) {
    const double inv = 1.0 / (double)k;
    double a = 0.0;
#pragma unroll
    for (int l = 0; l < 4; ++l) a = __builtin_fma(__shfl(term, base + i * 4 + l), xc[l], a);
    term = a * inv;
    out += term;
  }
  // (groups of one wave may need different numbers of squarings: every lane walks the wave's
  // maximum -- a shuffle needs its source lane active -- and keeps its value once it is done)
  int smax = s;
#pragma unroll
  for (int off = 32; off >= 16; off >>= 1) smax = max(smax, __shfl_xor(smax, off));
  for (int k = 0; k < smax; ++k) {
    double a = 0.0;
#pragma unroll
    for (int l = 0; l < 4; ++l)
      a = __builtin_fma(__shfl(out, base + i * 4 + l), __shfl(out, base + l * 4 + j), a);
    if (k < s) out = a;
  }
  return out;
}

}  // namespace rdamd
