// Which lane holds what in v_mfma_f64_4x4x4_4b_f64 (4 blocks of D = A * B + C, A 4x4 (i, k), B 4x4 (k, j), one double per lane
// for each of A, B, C/D)?  Feeds distinct values, brute-forces the digit assignment of lane -> (block, row/col, k) on the host.
// build: hipcc -O3 --offload-arch=gfx950 mfma_f64_4x4_layout.hip -o mfma_f64_4x4_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double *a, const double *b, double *d) {
  const int l = threadIdx.x;
  d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}
int main() {
  double ha[64], hb[64], hd[64], *da, *db, *dd;
  for (int l = 0; l < 64; l++) { ha[l] = 1.0 + l; hb[l] = 100.0 + 7.0 * l; }
  hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 512);
  hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd);
  if (hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  // a lane number has three digits: d0 = l % 4, d1 = (l / 4) % 4, d2 = l / 16; a hypothesis assigns them to (block, i|j, k)
  const char *names[3] = {"l%4", "(l/4)%4", "l/16"};
  int found = 0;
  for (int pa = 0; pa < 6; pa++) for (int pb = 0; pb < 6; pb++) for (int pd = 0; pd < 6; pd++) {
    static const int perm[6][3] = {{0,1,2},{0,2,1},{1,0,2},{1,2,0},{2,0,1},{2,1,0}};
    // perm[p] = which digit gives (block, row-or-col, k) -- for D: (block, i, j)
    auto lane_of = [&](const int *p, int blk, int x, int y) { int dg[3]; dg[p[0]] = blk; dg[p[1]] = x; dg[p[2]] = y; return dg[0] + 4 * dg[1] + 16 * dg[2]; };
    bool ok = true;
    for (int blk = 0; blk < 4 && ok; blk++) for (int i = 0; i < 4 && ok; i++) for (int j = 0; j < 4 && ok; j++) {
      double s = 0;
      for (int kk = 0; kk < 4; kk++) s += ha[lane_of(perm[pa], blk, i, kk)] * hb[lane_of(perm[pb], blk, j, kk)];
      if (fabs(s - hd[lane_of(perm[pd], blk, i, j)]) > 1e-9 * fabs(s)) ok = false;
    }
    if (ok) {
      found++;
      printf("A: block=%s i=%s k=%s | B: block=%s j=%s k=%s | D: block=%s i=%s j=%s\n", names[perm[pa][0]], names[perm[pa][1]], names[perm[pa][2]],
             names[perm[pb][0]], names[perm[pb][1]], names[perm[pb][2]], names[perm[pd][0]], names[perm[pd][1]], names[perm[pd][2]]);
    }
  }
  printf("%d layout(s) match\n", found);
  for (int l = 0; l < 8; l++) printf("d[%d] = %.1f\n", l, hd[l]);
  return found == 1 ? 0 : 3;
}
