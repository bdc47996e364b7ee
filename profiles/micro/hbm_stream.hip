// Measured HBM ceilings on MI355X for the access shapes the traversal kernel
// uses: a pure 16-byte-per-lane store stream (what its loop issues), a pure
// load stream, and a copy.  Build: hipcc -O3 --offload-arch=gfx950 hbm_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(256) write_k(double2 *p, size_t n, double x) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    p[i] = make_double2(x, x + 1.0);
}
__global__ void __launch_bounds__(256) read_k(const double2 *p, size_t n, double *out) {
  double acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const double2 v = p[i];
    acc += v.x + v.y;
  }
  if (acc == 1.2345) out[0] = acc;
}
__global__ void __launch_bounds__(256) copy_k(const double2 *a, double2 *b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    b[i] = a[i];
}

int main() {
  const size_t bytes = (size_t)1 << 30;   // 1 GiB per buffer, far beyond the 256 MB MALL
  double2 *a, *b;
  double *out;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&out, 8);
  hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
  const size_t n = bytes / 16;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {1024, 4096, 16384, 65536}) {
    float ms[3];
    for (int kind = 0; kind < 3; ++kind) {
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int it = 0; it < 5; ++it) {
          if (kind == 0) write_k<<<blocks, 256>>>(a, n, 1.0 + it);
          if (kind == 1) read_k<<<blocks, 256>>>(a, n, out);
          if (kind == 2) copy_k<<<blocks, 256>>>(a, b, n);
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[kind], e0, e1);
      }
      ms[kind] /= 5;
    }
    printf("blocks %6d: write %.0f GB/s  read %.0f GB/s  copy %.0f GB/s (read+write)\n", blocks,
           bytes / ms[0] / 1e6, bytes / ms[1] / 1e6, 2.0 * bytes / ms[2] / 1e6);
  }
  return 0;
}
