// Benchmark scaffolding, no reference counterpart: the synthetic registration scene of open3d_amd/synthetic.py
// generated on the device.  BASELINE config 5 registers 4096 pairs of 200 k points -- 59 GB of clouds and normals, which
// would take longer to push over PCIe than to register; the counter-based generator (splitmix64) makes every point a
// function of (seed, index), so a kernel produces the same clouds as the numpy code up to the last digits of the device's
// sin / cos / log (the CPU baseline of a batch run downloads the pairs it checks).
#include <math.h>

#include "o3d_internal.h"

namespace {

__device__ __forceinline__ double u01(uint64_t seed, uint64_t e) {  // uniform01(seed, 0, ...)[e]
  uint64_t z = seed + (e + 1ull) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

__device__ __forceinline__ double height(double x, double y) {
  return 0.30 * sin(1.3 * x) * cos(0.9 * y) + 0.12 * sin(3.1 * x + 0.7) + 0.08 * cos(2.3 * y - 0.4);
}

struct SceneArgs {
  int64_t n_src, n_tgt;
  uint64_t seed;
  double L, noise;  // noise: absolute standard deviation
  double Ti[16];    // inverse of the true transformation, column-major
  double *src, *tgt, *nrm;
};

__global__ __launch_bounds__(O3D_BLOCK) void k_synthetic_pair(SceneArgs a) {
  const int64_t n = a.n_src > a.n_tgt ? a.n_src : a.n_tgt;
  for (int64_t i = (int64_t)blockIdx.x * O3D_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * O3D_BLOCK) {
    if (i < a.n_tgt) {
      const double x = a.L * u01(a.seed, 2 * (uint64_t)i), y = a.L * u01(a.seed, 2 * (uint64_t)i + 1);
      a.tgt[3 * i] = x; a.tgt[3 * i + 1] = y; a.tgt[3 * i + 2] = height(x, y);
      if (a.nrm) {  // analytic unit normal of the height field
        const double zx = 0.30 * 1.3 * cos(1.3 * x) * cos(0.9 * y) + 0.12 * 3.1 * cos(3.1 * x + 0.7);
        const double zy = -0.30 * 0.9 * sin(1.3 * x) * sin(0.9 * y) - 0.08 * 2.3 * sin(2.3 * y - 0.4);
        const double inv = 1.0 / sqrt(zx * zx + zy * zy + 1.0);
        a.nrm[3 * i] = -zx * inv; a.nrm[3 * i + 1] = -zy * inv; a.nrm[3 * i + 2] = inv;
      }
    }
    if (i < a.n_src) {
      const uint64_t off = 2 * (uint64_t)a.n_tgt;
      const double x = a.L * u01(a.seed, off + 2 * (uint64_t)i), y = a.L * u01(a.seed, off + 2 * (uint64_t)i + 1);
      double p[3] = {x, y, height(x, y)};
      if (a.noise > 0.0) {  // Box-Muller: pair q of the noise stream gives g[2q] = rad cos, g[2q + 1] = rad sin
        const uint64_t noff = off + 2 * (uint64_t)a.n_src;
#pragma unroll
        for (int d = 0; d < 3; d++) {
          const uint64_t k = 3 * (uint64_t)i + d, q = k >> 1;
          const double u1 = 1.0 - u01(a.seed, noff + 2 * q), u2 = u01(a.seed, noff + 2 * q + 1);
          const double rad = sqrt(-2.0 * log(u1)), ang = 2.0 * 3.141592653589793 * u2;
          p[d] += a.noise * rad * ((k & 1) ? sin(ang) : cos(ang));
        }
      }
#pragma unroll
      for (int r = 0; r < 3; r++)
        a.src[3 * i + r] = ((p[0] * a.Ti[r] + p[1] * a.Ti[4 + r]) + p[2] * a.Ti[8 + r]) + a.Ti[12 + r];
    }
  }
}

}  // namespace

extern "C" int o3d_synthetic_pair(o3d_ctx *ctx, int64_t n_src, int64_t n_tgt, uint64_t seed, double L, double noise,
                                  const double inv_T_colmajor[16], double *d_src, double *d_tgt, double *d_tgt_nrm) {
  if (!ctx || n_src < 0 || n_tgt < 0 || !inv_T_colmajor || (n_src > 0 && !d_src) || (n_tgt > 0 && !d_tgt))
    return ctx ? o3d_fail(ctx, O3D_ERR_INVALID_ARG, "o3d_synthetic_pair: bad arguments") : O3D_ERR_INVALID_ARG;
  const int64_t n = n_src > n_tgt ? n_src : n_tgt;
  if (n == 0) return O3D_OK;
  int prev = -1;
  if (hipGetDevice(&prev) != hipSuccess) prev = -1;
  (void)hipSetDevice(ctx->device);
  SceneArgs a;
  a.n_src = n_src; a.n_tgt = n_tgt; a.seed = seed; a.L = L; a.noise = noise;
  for (int i = 0; i < 16; i++) a.Ti[i] = inv_T_colmajor[i];
  a.src = d_src; a.tgt = d_tgt; a.nrm = d_tgt_nrm;
  int64_t g = (n + O3D_BLOCK - 1) / O3D_BLOCK;
  if (g > O3D_MAX_GRID) g = O3D_MAX_GRID;
  hipLaunchKernelGGL(k_synthetic_pair, dim3((int)g), dim3(O3D_BLOCK), 0, ctx->stream, a);
  const hipError_t e = hipGetLastError();
  if (prev >= 0) (void)hipSetDevice(prev);
  if (e != hipSuccess) return o3d_fail(ctx, O3D_ERR_HIP, "o3d_synthetic_pair: %s", hipGetErrorString(e));
  return O3D_OK;
}
