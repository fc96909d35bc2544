// How many vector-memory lane requests does a CU's L1 take per cycle?  Every wavefront re-reads a small buffer that stays in
// the L1 with loads of 4, 8 or 16 bytes per lane, lanes contiguous or each lane in a line of its own; 16 wavefronts per CU as
// in the pass kernels.  Reports loads per cycle and CU and bytes per cycle and CU (2.4 GHz assumed for the cycle count).
// build: hipcc -O3 --offload-arch=gfx950 l1_rate.hip -o l1_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <typename T, int STRIDE_BYTES>
__global__ __launch_bounds__(256, 4) void k(const char *buf, int iters, float *out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // 8 KB per wavefront region when lanes are spread (64 lanes x 128 B), 16 distinct offsets per iteration batch
  const char *base = buf + (size_t)wave * 8192 + (size_t)lane * STRIDE_BYTES;
  float acc = 0.f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const char *pp = base + (u & 3) * sizeof(T) * (STRIDE_BYTES == (int)sizeof(T) ? 64 : 1);
      asm volatile("" : "+v"(pp));  // (the address is opaque: the load is issued every time)
      const T v = *reinterpret_cast<const T *>(pp);
      if constexpr (sizeof(T) == 4) acc += v; else if constexpr (sizeof(T) == 8) acc += v.x + v.y; else acc += v.x + v.y + v.z + v.w;
    }
  }
  if (acc == 1234.5f) out[0] = acc;
}
template <typename T, int STRIDE>
void run(const char *name, const char *buf, float *out) {
  const int iters = 4000, grid = 1024;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<T, STRIDE>), dim3(grid), dim3(256), 0, 0, buf, iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double instr_per_cu = (double)grid * 4 / 256 * iters * 8;  // wavefront-level load instructions per CU
  const double cycles = best * 1e-3 * 2.4e9;
  printf("%-34s %.3f ms  %.3f wavefront loads / cycle / CU = %.1f lane requests, %.0f bytes / cycle / CU\n", name, best, instr_per_cu / cycles,
         instr_per_cu * 64 / cycles, instr_per_cu * 64 * sizeof(T) / cycles);
}
int main() {
  char *buf; float *out;
  hipMalloc(&buf, 1 << 20); hipMemset(buf, 0, 1 << 20); hipMalloc(&out, 4);
  run<float, 4>("4 B per lane, contiguous", buf, out);
  run<f2, 8>("8 B per lane, contiguous", buf, out);
  run<f4, 16>("16 B per lane, contiguous", buf, out);
  run<float, 128>("4 B per lane, a line per lane", buf, out);
  run<f2, 128>("8 B per lane, a line per lane", buf, out);
  run<f4, 128>("16 B per lane, a line per lane", buf, out);
  run<f4, 48>("16 B per lane, 48-byte stride", buf, out);
  run<f4, 64>("16 B per lane, 64-byte stride", buf, out);
  return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}
