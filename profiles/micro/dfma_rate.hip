// Microbenchmark: FP64 FMA issue rate and sustained clock on gfx950.
// hipcc --offload-arch=gfx950 -O3 dfma_rate.hip -o dfma_rate && ./dfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void __launch_bounds__(256) k(double *out, int iters, double a, double b) {
  double x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3 + i;
  unsigned u[8];
  for (int i = 0; i < 8; ++i) u[i] = threadIdx.x + i;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) x[i] = __builtin_fma(x[i], a, b);
      if (MODE == 1) x[i] = x[i] * a;
      if (MODE == 2) u[i] = u[i] * 3u + 1u;                 // 32-bit int mad
      if (MODE == 3) u[i] = max(u[i], (unsigned)it) ^ 5u;   // 2 cheap int ops
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0; unsigned su = 0;
  for (int i = 0; i < 8; ++i) { s += x[i]; su += u[i]; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + su;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0);
}
template <int MODE> void run(const char *name, int waves_per_simd, double ops_per_iter) {
  double *d; hipMalloc(&d, 256 * 1024 * 8 * sizeof(double));
  int blocks = 256 * waves_per_simd;  // 4 waves per block -> waves_per_simd per SIMD
  int iters = 200000;
  k<MODE><<<blocks, 256>>>(d, 1000, 1.0000001, 1e-9);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); k<MODE><<<blocks, 256>>>(d, iters, 1.0000001, 1e-9); hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double cyc; hipMemcpy(&cyc, d, 8, hipMemcpyDeviceToHost);
  double insts = (double)iters * 8 * ops_per_iter;          // per wave
  double waves = blocks * 4.0;
  printf("%-10s waves/SIMD=%d  %.3f ms  %.2f T inst-lanes/s  cycles(memtime)=%.0f -> %.2f cyc/inst/wave, clock~%.2f GHz, SIMD cyc/inst=%.2f\n",
         name, waves_per_simd, ms, insts * waves * 64 / (ms * 1e-3) / 1e12, cyc, cyc / insts,
         cyc / (ms * 1e-3) / 1e9 * 0 + 0.0, (ms * 1e-3) * 2.4e9 / (insts * waves_per_simd));
  hipFree(d);
}
int main() {
  for (int w : {1, 2, 4, 8}) run<0>("dfma", w, 1);
  for (int w : {1, 4}) run<1>("dmul", w, 1);
  for (int w : {1, 4}) run<2>("imad32", w, 1);
  for (int w : {1, 4}) run<3>("int2", w, 2);
  return 0;
}
