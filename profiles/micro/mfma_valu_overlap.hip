// Microbenchmark: do FP64 VALU FMAs and v_mfma_f64_4x4x4_4b_f64 run side by side on a gfx950 SIMD?
// Each wave runs independent chains of both kinds; time(both) ~ max(parts) means the pipes overlap,
// ~ sum(parts) means the matrix instruction occupies the vector issue.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NM, int NV>
__global__ void __launch_bounds__(256) k(double *out, int iters) {
  double m[8], f[8];
  for (int c = 0; c < 8; ++c) { m[c] = 0.0; f[c] = threadIdx.x * 1e-3 + c; }
  const double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (c < NM) m[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, m[c], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NV; ++q) f[(c + q) & 7] = __builtin_fma(f[(c + q) & 7], b, a);
    }
  }
  double s = 0;
  for (int c = 0; c < 8; ++c) s += m[c] + f[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NM, int NV> void run(int waves_per_simd) {
  double *d; (void)hipMalloc(&d, 256 * 256 * 8 * sizeof(double));
  const int blocks = 256 * waves_per_simd, iters = 20000;
  k<NM, NV><<<blocks, 256>>>(d, 100);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0); k<NM, NV><<<blocks, 256>>>(d, iters); (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * waves_per_simd);
  printf("per iteration and wave: %d MFMA + %d DFMA, %d waves/SIMD: %.1f cycles (%.1f per MFMA-equivalent slot)\n",
         NM, 8 * NV, waves_per_simd, cyc, cyc / 8);
  (void)hipFree(d);
}
int main() {
  run<8, 0>(1); run<0, 4>(1); run<8, 4>(1);
  run<8, 0>(2); run<0, 4>(2); run<8, 4>(2);
  run<8, 0>(4); run<0, 4>(4); run<8, 4>(4);
  return 0;
}
