// Microbenchmark: v_mfma_f64_16x16x4_f64 and v_mfma_f64_4x4x4_4b_f64 issue rates on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int CHAINS>
__global__ void __launch_bounds__(256) k(double *out, int iters) {
  v4d acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = v4d{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
  }
  double s = 0;
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS> void run(int waves_per_simd) {
  double *d; (void)hipMalloc(&d, 256 * 256 * 8 * sizeof(double));
  int blocks = 256 * waves_per_simd, iters = 20000;
  k<CHAINS><<<blocks, 256>>>(d, 100);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0); k<CHAINS><<<blocks, 256>>>(d, iters); (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double mfmas_per_simd = (double)iters * CHAINS * waves_per_simd;
  double flops = (double)iters * CHAINS * blocks * 4 * 2048.0;
  printf("chains=%d waves/SIMD=%d: %.3f ms, %.1f TFLOP/s, %.1f cycles/MFMA/SIMD (at 2.4 GHz nominal)\n",
         CHAINS, waves_per_simd, ms, flops / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / mfmas_per_simd);
  (void)hipFree(d);
}
// the 4x4x4 (4 blocks) form: 512 flops per instruction, one D element per lane
template <int CHAINS>
__global__ void __launch_bounds__(256) k4(double *out, int iters) {
  double acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = 0.0;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[c], 0, 0, 0);
  }
  double s = 0;
  for (int c = 0; c < CHAINS; ++c) s += acc[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS> void run4(int waves_per_simd) {
  double *d; (void)hipMalloc(&d, 256 * 256 * 8 * sizeof(double));
  int blocks = 256 * waves_per_simd, iters = 20000;
  k4<CHAINS><<<blocks, 256>>>(d, 100);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0); k4<CHAINS><<<blocks, 256>>>(d, iters); (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double mfmas_per_simd = (double)iters * CHAINS * waves_per_simd;
  double flops = (double)iters * CHAINS * blocks * 4 * 512.0;
  printf("4x4x4_4b chains=%d waves/SIMD=%d: %.3f ms, %.1f TFLOP/s, %.1f cycles/MFMA/SIMD (at 2.4 GHz nominal)\n",
         CHAINS, waves_per_simd, ms, flops / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / mfmas_per_simd);
  (void)hipFree(d);
}
int main() {
  run<1>(1); run<2>(1); run<4>(1); run<4>(2); run<8>(1);
  run4<1>(1); run4<2>(1); run4<4>(1); run4<8>(1); run4<8>(2); run4<16>(1);
  return 0;
}
