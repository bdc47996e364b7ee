// Probe: operand / result lane layout of v_mfma_f64_4x4x4_4b_f64 on gfx950.
// Each lane supplies one A and one B element and receives one D element; the
// host tries the candidate index maps and reports the one that reproduces D.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__global__ void probe(const double *a, const double *b, double *d) {
  const int l = threadIdx.x;
  d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}
int main() {
  double ha[64], hb[64], hd[64];
  for (int l = 0; l < 64; ++l) { ha[l] = 1.0 + 0.37 * l + 0.011 * l * l; hb[l] = 2.0 - 0.21 * l + 0.007 * l * l; }
  double *da, *db, *dd;
  (void)hipMalloc(&da, 512); (void)hipMalloc(&db, 512); (void)hipMalloc(&dd, 512);
  (void)hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(da, db, dd);
  (void)hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
  // candidate maps: the lane number is three 2-bit fields; try every assignment of
  // (block, row/col, k) to the fields for A, B and D
  const int perms[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
  auto lane_of = [&](const int *pm, int blk, int minor, int k) {
    const int v[3] = {blk, minor, k};   // field f of the lane holds v[pm[f]]
    return v[pm[0]] | (v[pm[1]] << 2) | (v[pm[2]] << 4);
  };
  const char *what[3] = {"block", "minor", "k"};
  for (int pa = 0; pa < 6; ++pa) for (int pb = 0; pb < 6; ++pb) for (int pd = 0; pd < 6; ++pd) {
    double worst = 0;
    for (int blk = 0; blk < 4; ++blk) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
      double acc = 0;
      for (int k = 0; k < 4; ++k) acc += ha[lane_of(perms[pa], blk, i, k)] * hb[lane_of(perms[pb], blk, j, k)];
      // D: fields are (block, j as "minor", i as "k")
      worst = fmax(worst, fabs(acc - hd[lane_of(perms[pd], blk, j, i)]) / fabs(acc));
    }
    if (worst < 1e-12)
      printf("MATCH  A lane bits[1:0,3:2,5:4] = (%s,%s,%s) with minor=i;  B = (%s,%s,%s) with minor=j;  "
             "D = (%s,%s,%s) with minor=j, k=i\n",
             what[perms[pa][0]], what[perms[pa][1]], what[perms[pa][2]],
             what[perms[pb][0]], what[perms[pb][1]], what[perms[pb][2]],
             what[perms[pd][0]], what[perms[pd][1]], what[perms[pd][2]]);
  }
  printf("probe done\n");
  return 0;
}
