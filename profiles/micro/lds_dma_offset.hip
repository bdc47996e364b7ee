// Where does `buffer_load_dwordx4 ... offset:N lds` put its data?  (gfx950)
// Answer printed by this program: LDS address = M0 + inst_offset + lane * 16, i.e. the
// instruction offset moves the LDS destination as well as the global source.  The fused
// evaluator's 64-row table loads rely on it (kernels_fused.hip, RDAMD_LOAD_TAB).
//   hipcc --offload-arch=gfx950 -O2 lds_dma_offset.hip -o lds_dma_offset && ./lds_dma_offset
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void *lds_void_ptr;
__global__ void k(const double *g, double *out) {
  extern __shared__ double lds[];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = -1.0;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)g, 0, 1 << 20, 0x00020000);
  // LDS pointer 2048, instruction offset 1024: data of global bytes [1024, 2048)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_ptr)(size_t)2048, 16, threadIdx.x * 16, 0, 1024, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}
int main() {
  std::vector<double> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = i;
  double *g, *o;
  hipMalloc(&g, 4096 * 8); hipMalloc(&o, 1024 * 8);
  hipMemcpy(g, h.data(), 4096 * 8, hipMemcpyHostToDevice);
  k<<<1, 64, 8192>>>(g, o);
  std::vector<double> r(1024);
  hipMemcpy(r.data(), o, 1024 * 8, hipMemcpyDeviceToHost);
  int first = -1, last = -1;
  for (int i = 0; i < 1024; ++i) if (r[i] >= 0) { if (first < 0) first = i; last = i; }
  std::printf("data landed at LDS doubles [%d, %d] = bytes [%d, %d); first value %g (global double index)\n",
              first, last, first * 8, (last + 1) * 8, first >= 0 ? r[first] : -1.0);
  std::printf("%s\n", first * 8 == 2048 + 1024 ? "LDS address = pointer + inst_offset + lane*16"
                      : first * 8 == 2048 ? "LDS address = pointer + lane*16 (offset applies to the source only)" : "unexpected");
  return 0;
}
