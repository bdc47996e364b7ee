// Microbenchmark: how long does a SHORT kernel take beside a LONG one that fills every wave slot?
//
// The lock-stepped root search runs small kernels (root-only steps, P-matrices, clade tables)
// beside the fused evaluator, whose one-wave workgroups (128 VGPRs, 8.7 KB LDS: 16 per CU) hold
// every slot of the device for a millisecond per launch.  This program reproduces that: a
// background kernel of 40 000 one-wave workgroups that spin ~250 us each, and a foreground
// kernel of 3 128 short workgroups launched 150 us later on another stream.  It prints the
// foreground kernel's duration for: workgroup size (64 / 256 lanes), VGPRs (fits one freed
// slot or not), LDS (more or less than one background wave frees), stream priority, and a CU
// mask that keeps the background kernel off 16 CUs and the foreground kernel on them.
// hipcc --offload-arch=gfx950 -O3 side_kernel_latency.hip -o side_kernel_latency && ./side_kernel_latency
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// ~128 VGPRs, `lds_bytes` of dynamic LDS, spins for `cycles` shader clocks
__global__ void __launch_bounds__(64) background(double *out, long long cycles) {
  extern __shared__ double lds[];
  double x[60];
#pragma unroll
  for (int i = 0; i < 60; ++i) x[i] = threadIdx.x * 1e-3 + i;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {
#pragma unroll
    for (int i = 0; i < 60; ++i) x[i] = __builtin_fma(x[i], 1.0000001, 1e-9);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 60; ++i) s += x[i];
  lds[threadIdx.x] = s;
  out[(size_t)blockIdx.x * 64 + threadIdx.x] = lds[threadIdx.x];
}

// NV doubles of live registers per lane (VGPRs ~ 2 NV + 10), spins `cycles`
template <int NV, int THREADS>
__global__ void __launch_bounds__(THREADS) foreground(double *out, long long cycles) {
  extern __shared__ double lds[];
  double x[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) x[i] = threadIdx.x * 1e-3 + i;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {
#pragma unroll
    for (int i = 0; i < NV; ++i) x[i] = __builtin_fma(x[i], 1.0000001, 1e-9);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += x[i];
  lds[threadIdx.x] = s;
  out[(size_t)blockIdx.x * THREADS + threadIdx.x] = lds[threadIdx.x];
}

struct Streams { hipStream_t bg, fg; };

static Streams make_streams(int bg_prio, int fg_prio, int side_cus) {
  Streams s;
  if (side_cus > 0) {
    // bit i of the mask: XCD i % 8, then round-robin over its shader engines (KFD's symmetric
    // mapping): the lowest `side_cus` bits are side_cus / 8 CUs on every XCD
    std::vector<uint32_t> fg(8, 0), bg(8, 0xffffffffu);
    for (int i = 0; i < side_cus; ++i) { fg[i / 32] |= 1u << (i % 32); bg[i / 32] &= ~(1u << (i % 32)); }
    CHECK(hipExtStreamCreateWithCUMask(&s.bg, 8, bg.data()));
    CHECK(hipExtStreamCreateWithCUMask(&s.fg, 8, fg.data()));
  } else {
    CHECK(hipStreamCreateWithPriority(&s.bg, hipStreamNonBlocking, bg_prio));
    CHECK(hipStreamCreateWithPriority(&s.fg, hipStreamNonBlocking, fg_prio));
  }
  return s;
}

template <int NV, int THREADS>
static void run(const char *name, int bg_prio, int fg_prio, int side_cus, size_t fg_lds, double *d_bg, double *d_fg,
                double clock_ghz) {
  Streams s = make_streams(bg_prio, fg_prio, side_cus);
  hipEvent_t b0, b1, f0, f1;
  CHECK(hipEventCreate(&b0)); CHECK(hipEventCreate(&b1)); CHECK(hipEventCreate(&f0)); CHECK(hipEventCreate(&f1));
  const long long bg_cycles = (long long)(250e-6 * 100e6);   // wall_clock64 ticks at 100 MHz
  const long long fg_cycles = (long long)(8e-6 * 100e6);
  const int fg_blocks = 3128 * 64 / THREADS;
  float fg_alone = 0, fg_beside = 0, bg_alone = 0, bg_beside = 0;
  for (int rep = 0; rep < 3; ++rep) {   // alone
    CHECK(hipEventRecord(f0, s.fg));
    foreground<NV, THREADS><<<fg_blocks, THREADS, fg_lds, s.fg>>>(d_fg, fg_cycles);
    CHECK(hipEventRecord(f1, s.fg));
    CHECK(hipStreamSynchronize(s.fg));
    CHECK(hipEventElapsedTime(&fg_alone, f0, f1));
  }
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipEventRecord(b0, s.bg));
    background<<<40000, 64, 8704, s.bg>>>(d_bg, bg_cycles);
    CHECK(hipEventRecord(b1, s.bg));
    CHECK(hipStreamSynchronize(s.bg));
    CHECK(hipEventElapsedTime(&bg_alone, b0, b1));
  }
  float worst = 0, sum = 0;
  const int reps = 5;
  for (int rep = 0; rep < reps; ++rep) {
    CHECK(hipEventRecord(b0, s.bg));
    background<<<40000, 64, 8704, s.bg>>>(d_bg, bg_cycles);
    CHECK(hipEventRecord(b1, s.bg));
    // wait ~150 us on the host, then the foreground kernel
    timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
    do { clock_gettime(CLOCK_MONOTONIC, &t1); } while ((t1.tv_sec - t0.tv_sec) * 1e9 + (t1.tv_nsec - t0.tv_nsec) < 150e3);
    CHECK(hipEventRecord(f0, s.fg));
    foreground<NV, THREADS><<<fg_blocks, THREADS, fg_lds, s.fg>>>(d_fg, fg_cycles);
    CHECK(hipEventRecord(f1, s.fg));
    CHECK(hipStreamSynchronize(s.fg));
    CHECK(hipStreamSynchronize(s.bg));
    CHECK(hipEventElapsedTime(&fg_beside, f0, f1));
    CHECK(hipEventElapsedTime(&bg_beside, b0, b1));
    sum += fg_beside; if (fg_beside > worst) worst = fg_beside;
  }
  printf("%-44s fg alone %7.1f us  beside %7.1f us (worst %7.1f)   bg alone %7.1f us  beside %7.1f us\n", name,
         fg_alone * 1e3, sum / reps * 1e3, worst * 1e3, bg_alone * 1e3, bg_beside * 1e3);
  CHECK(hipStreamDestroy(s.bg)); CHECK(hipStreamDestroy(s.fg));
  (void)clock_ghz;
}

int main() {
  double *d_bg, *d_fg;
  CHECK(hipMalloc(&d_bg, (size_t)40000 * 64 * 8));
  CHECK(hipMalloc(&d_fg, (size_t)3128 * 64 * 8));
  int least, greatest;
  CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
  printf("stream priorities: least %d greatest %d\n", least, greatest);
  const int N = (least + greatest) / 2;
  // VGPRs: NV = 45 -> ~100 (fits one freed 128-register slot), NV = 60 -> ~130 (does not)
  run<45, 64>("64 lanes, ~100 VGPR, 6 KB LDS, equal prio", N, N, 0, 6144, d_bg, d_fg, 2.4);
  run<45, 64>("64 lanes, ~100 VGPR, 6 KB LDS, fg high", least, greatest, 0, 6144, d_bg, d_fg, 2.4);
  run<45, 64>("64 lanes, ~100 VGPR, 10 KB LDS, fg high", least, greatest, 0, 10240, d_bg, d_fg, 2.4);
  run<45, 64>("64 lanes, ~100 VGPR, 0.5 KB LDS, fg high", least, greatest, 0, 512, d_bg, d_fg, 2.4);
  run<64, 64>("64 lanes, ~135 VGPR, 6 KB LDS, fg high", least, greatest, 0, 6144, d_bg, d_fg, 2.4);
  run<20, 64>("64 lanes, ~50 VGPR, 6 KB LDS, fg high", least, greatest, 0, 6144, d_bg, d_fg, 2.4);
  run<45, 256>("256 lanes, ~100 VGPR, 6 KB LDS, fg high", least, greatest, 0, 6144, d_bg, d_fg, 2.4);
  run<45, 64>("64 lanes, ~100 VGPR, 6 KB LDS, bg low fg normal", least, N, 0, 6144, d_bg, d_fg, 2.4);
  run<45, 64>("64 lanes, ~100 VGPR, 10 KB LDS, 16 CUs masked", N, N, 16, 10240, d_bg, d_fg, 2.4);
  run<45, 64>("64 lanes, ~100 VGPR, 10 KB LDS, 32 CUs masked", N, N, 32, 10240, d_bg, d_fg, 2.4);
  run<45, 256>("256 lanes, ~100 VGPR, 10 KB LDS, 16 CUs masked", N, N, 16, 10240, d_bg, d_fg, 2.4);
  return 0;
}
