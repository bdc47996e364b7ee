// What does an LDS-DMA lane whose buffer offset is out of range do to LDS: write zeros, or
// nothing?  (gfx950; decides how a 16-row table may share the 64-row table's load instruction)
//   hipcc --offload-arch=gfx950 -O2 lds_dma_oob.hip -o lds_dma_oob && ./lds_dma_oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void *lds_void_ptr;
__global__ void k(const double *g, double *out) {
  extern __shared__ double lds[];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = -1.0;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)g, 0, 4096, 0x00020000);
  const int vo = threadIdx.x < 16 ? (int)threadIdx.x * 16 : 0x40000000;   // lanes 16.. out of range
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_ptr)(size_t)0, 16, vo, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}
int main() {
  std::vector<double> h(512);
  for (int i = 0; i < 512; ++i) h[i] = 100 + i;
  double *g, *o;
  hipMalloc(&g, 4096); hipMalloc(&o, 256 * 8);
  hipMemcpy(g, h.data(), 4096, hipMemcpyHostToDevice);
  k<<<1, 64, 4096>>>(g, o);
  std::vector<double> r(256);
  hipMemcpy(r.data(), o, 256 * 8, hipMemcpyDeviceToHost);
  int data = 0, zeros = 0, untouched = 0;
  for (int i = 0; i < 128; ++i) { data += r[i] >= 100; zeros += r[i] == 0.0; untouched += r[i] == -1.0; }
  std::printf("first 1 KB of LDS after the load: %d doubles of data, %d zeros, %d untouched\n", data, zeros, untouched);
  std::printf("%s\n", zeros ? "out-of-range lanes WRITE ZEROS to their LDS slots" : "out-of-range lanes leave LDS alone");
  return 0;
}
