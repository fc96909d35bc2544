// Read-only and copy bandwidth of the box, for comparison with the certifying pass: a grid of G workgroups x 256 threads streams
// N bytes with 16-byte loads (and optionally stores them).  build: hipcc -O3 --offload-arch=gfx950 read_bw.hip -o read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_read(const f4 *__restrict__ x, size_t n, float *out, int nt) {
  f4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const f4 v = nt ? __builtin_nontemporal_load(&x[i]) : x[i];
    acc += v;
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}
__global__ __launch_bounds__(256) void k_copy(const f4 *__restrict__ x, f4 *__restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = x[i];
}
int main(int argc, char **argv) {
  const size_t bytes = (size_t)(argc > 1 ? atof(argv[1]) : 2e9);
  const size_t n = bytes / 16;
  f4 *x, *y; float *out;
  hipMalloc(&x, n * 16); hipMalloc(&y, n * 16); hipMalloc(&out, 4);
  hipMemset(x, 0, n * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {1024, 2048, 4096, 8192}) {
    for (int mode = 0; mode < 3; mode++) {
      float best = 1e9;
      for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0);
        if (mode < 2) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, x, n, out, mode);
        else hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, x, y, n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      printf("grid %5d %-14s %.3f ms  %.2f TB/s\n", grid, mode == 0 ? "read" : mode == 1 ? "read (nt)" : "copy (r+w)", best,
             (mode < 2 ? 1.0 : 2.0) * n * 16 / (best * 1e-3) / 1e12);
    }
  }
  return 0;
}
