// What bounds the store stream of the drop-in traversal (clv_dna_traversal_kernel, c2: 99 operations x
// 50 000 sites x 4 rates x 32 B = 634 MB in 172 us = 3.7 TB/s)?  Pure stores in the kernel's pattern --
// every lane owns one (site, rate) and writes its 32-byte record of EVERY operation, one after the
// other -- against the same bytes in other shapes:
//   op-major   [op][site][rate][state]         the partition's layout: a wave's consecutive stores
//                                              are 6.4 MB apart
//   tile-major [tile of 64 sites][op][...]     a workgroup's stores of the whole list fall into
//                                              one contiguous 800 KB region
//   8 B / lane                                 one lane per (site, rate, state): 4 x the waves
//   stream                                     a grid-stride store stream over the same 634 MB
// Build: hipcc -O3 --offload-arch=gfx950 clv_store_pattern.hip -o clv_store_pattern.bin
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr unsigned kOps = 99, kSites = 50000, kRates = 4;
constexpr size_t kLanes = (size_t)kSites * kRates;          // 200 000 records per operation

template <bool TILE>
__global__ void __launch_bounds__(256) rec32(double2 *p, double x, unsigned spacing) {
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= kLanes) return;
  for (unsigned op = 0; op < kOps; ++op) {
    const size_t rec = TILE ? ((size_t)blockIdx.x * kOps + op) * 256 + threadIdx.x : (size_t)op * kLanes + g;
    p[rec * 2] = make_double2(x, x + op);
    p[rec * 2 + 1] = make_double2(x + 2, x + 3);
    for (unsigned k = 0; k < spacing; ++k) __builtin_amdgcn_s_sleep(8);   // (the kernel's arithmetic between stores)
  }
}
template <bool TILE>
__global__ void __launch_bounds__(256) rec8(double *p, double x) {
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;   // one lane per (site, rate, state)
  if (g >= kLanes * 4) return;
  for (unsigned op = 0; op < kOps; ++op) {
    const size_t at = TILE ? ((size_t)blockIdx.x * kOps + op) * 256 + threadIdx.x : (size_t)op * kLanes * 4 + g;
    p[at] = x + op;
  }
}
__global__ void __launch_bounds__(256) stream(double2 *p, size_t n, double x) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = make_double2(x, x + 1);
}

int main() {
  const size_t blocks32 = (kLanes + 255) / 256, blocks8 = (kLanes * 4 + 255) / 256;
  const size_t bytes = blocks32 * 256 * 32 * kOps + (1 << 20);
  double2 *p;
  if (hipMalloc(&p, bytes) != hipSuccess) return 1;
  hipMemset(p, 0, bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const double gb = (double)kLanes * 32 * kOps / 1e9;
  auto time = [&](const char *name, auto launch) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      launch();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep && ms < best) best = ms;
    }
    printf("%-44s %8.1f us  %5.2f TB/s\n", name, best * 1e3, gb / best);
  };
  time("32 B / lane, op-major (the kernel's pattern)", [&] { rec32<false><<<blocks32, 256>>>(p, 1.0, 0); });
  time("32 B / lane, tile-major", [&] { rec32<true><<<blocks32, 256>>>(p, 1.0, 0); });
  for (unsigned sp : {1u, 2u, 4u}) {
    char name[96];
    snprintf(name, sizeof name, "32 B / lane, op-major, %u x s_sleep 8 between", sp);
    time(name, [&] { rec32<false><<<blocks32, 256>>>(p, 1.0, sp); });
    snprintf(name, sizeof name, "32 B / lane, tile-major, %u x s_sleep 8 between", sp);
    time(name, [&] { rec32<true><<<blocks32, 256>>>(p, 1.0, sp); });
  }
  time("8 B / lane, op-major", [&] { rec8<false><<<blocks8, 256>>>((double *)p, 1.0); });
  time("8 B / lane, tile-major", [&] { rec8<true><<<blocks8, 256>>>((double *)p, 1.0); });
  for (int blocks : {1024, 4096, 16384})  {
    char name[64];
    snprintf(name, sizeof name, "grid-stride stream, %d blocks", blocks);
    time(name, [&] { stream<<<blocks, 256>>>(p, (size_t)(gb * 1e9 / 16), 1.0); });
  }
  return 0;
}
