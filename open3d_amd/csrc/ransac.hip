// Feature-space nearest neighbour for RegistrationRANSACBasedOnFeatureMatching (pipelines/registration/Registration.cpp:223,237).
//
// The reference builds a FLANN KD-tree over the target Feature (dim = 33 for FPFH) and asks for the nearest column of each
// sampled source column.  In 33 dimensions a KD-tree prunes next to nothing, so the device path is the exhaustive scan:
// one lane per query with the query column in registers, the target columns read through wave-uniform (scalar) loads, the
// distance in flann::L2<double>'s own order -- groups of four `result += d0*d0 + d1*d1 + d2*d2 + d3*d3`, then the tail
// one by one, no fused multiply-add -- so that the winner is the exact argmin over (d2, index).
// Bound: fp64 vector ALU (3 * dim operations per pair), not HBM: the target table (n_tgt * dim * 8 B) is re-read from L2
// by every wavefront.  The target range is split over blockIdx.y so that a few queries still fill the chip; a second small
// kernel merges the per-chunk winners in chunk order (ties -> lowest index).
#include <math.h>

#include <algorithm>

#include "o3d_internal.h"

namespace {

constexpr int kFeatMaxDim = 128;  // generic path: the workgroup's query columns live in LDS (64 KiB at 128)

template <int DIM>
__device__ __forceinline__ double flann_l2(const double (&f)[DIM], const double *__restrict__ tp) {
  double r = 0.0;
  int k = 0;
#pragma unroll
  for (; k + 3 < DIM; k += 4) {
    const double d0 = f[k] - tp[k], d1 = f[k + 1] - tp[k + 1], d2 = f[k + 2] - tp[k + 2], d3 = f[k + 3] - tp[k + 3];
    r = r + (((d0 * d0 + d1 * d1) + d2 * d2) + d3 * d3);
  }
#pragma unroll
  for (; k < DIM; k++) {
    const double d = f[k] - tp[k];
    r = r + d * d;
  }
  return r;
}

// part_*: [chunk][query].  DIM > 0: compile-time dimension, query in registers.
template <int DIM>
__global__ __launch_bounds__(O3D_WAVE) void k_feature_nn(const double *__restrict__ sf, int64_t n_src, const double *__restrict__ tf,
                                                          int64_t n_tgt, const int32_t *__restrict__ qids, int64_t n_q,
                                                          int64_t per_chunk, double *__restrict__ part_d2,
                                                          int32_t *__restrict__ part_idx) {
  const int64_t q = (int64_t)blockIdx.x * O3D_WAVE + threadIdx.x;
  const bool valid = q < n_q;
  int64_t id = valid ? (qids ? (int64_t)qids[q] : q) : 0;
  const bool in_range = valid && id >= 0 && id < n_src;
  if (!in_range) id = 0;
  double f[DIM];
#pragma unroll
  for (int k = 0; k < DIM; k++) f[k] = n_src > 0 ? sf[(size_t)id * DIM + k] : 0.0;
  const int64_t t0 = (int64_t)blockIdx.y * per_chunk;
  const int64_t t1 = t0 + per_chunk < n_tgt ? t0 + per_chunk : n_tgt;
  double best = INFINITY;
  int32_t bi = -1;
  for (int64_t t = t0; t < t1; t++) {  // wave-uniform: the target column comes through the scalar unit
    const double r = flann_l2<DIM>(f, tf + (size_t)t * DIM);
    if (r < best) { best = r; bi = (int32_t)t; }
  }
  if (valid) {
    part_d2[(size_t)blockIdx.y * n_q + q] = in_range ? best : INFINITY;
    part_idx[(size_t)blockIdx.y * n_q + q] = in_range ? bi : -1;
  }
}

// Any dimension up to kFeatMaxDim: the 64 query columns of the workgroup in LDS, [k][lane].
__global__ __launch_bounds__(O3D_WAVE) void k_feature_nn_generic(const double *__restrict__ sf, int64_t n_src,
                                                                  const double *__restrict__ tf, int64_t n_tgt, int dim,
                                                                  const int32_t *__restrict__ qids, int64_t n_q, int64_t per_chunk,
                                                                  double *__restrict__ part_d2, int32_t *__restrict__ part_idx) {
  extern __shared__ double fq[];  // [dim][64]
  const int lane = threadIdx.x;
  const int64_t q = (int64_t)blockIdx.x * O3D_WAVE + lane;
  const bool valid = q < n_q;
  int64_t id = valid ? (qids ? (int64_t)qids[q] : q) : 0;
  const bool in_range = valid && id >= 0 && id < n_src;
  if (!in_range) id = 0;
  for (int k = 0; k < dim; k++) fq[k * O3D_WAVE + lane] = n_src > 0 ? sf[(size_t)id * dim + k] : 0.0;
  const int64_t t0 = (int64_t)blockIdx.y * per_chunk;
  const int64_t t1 = t0 + per_chunk < n_tgt ? t0 + per_chunk : n_tgt;
  double best = INFINITY;
  int32_t bi = -1;
  for (int64_t t = t0; t < t1; t++) {
    const double *__restrict__ tp = tf + (size_t)t * dim;
    double r = 0.0;
    int k = 0;
    for (; k + 3 < dim; k += 4) {
      const double d0 = fq[k * O3D_WAVE + lane] - tp[k], d1 = fq[(k + 1) * O3D_WAVE + lane] - tp[k + 1];
      const double d2 = fq[(k + 2) * O3D_WAVE + lane] - tp[k + 2], d3 = fq[(k + 3) * O3D_WAVE + lane] - tp[k + 3];
      r = r + (((d0 * d0 + d1 * d1) + d2 * d2) + d3 * d3);
    }
    for (; k < dim; k++) {
      const double d = fq[k * O3D_WAVE + lane] - tp[k];
      r = r + d * d;
    }
    if (r < best) { best = r; bi = (int32_t)t; }
  }
  if (valid) {
    part_d2[(size_t)blockIdx.y * n_q + q] = in_range ? best : INFINITY;
    part_idx[(size_t)blockIdx.y * n_q + q] = in_range ? bi : -1;
  }
}

__global__ __launch_bounds__(O3D_BLOCK) void k_feature_nn_merge(const double *__restrict__ part_d2, const int32_t *__restrict__ part_idx,
                                                                 int64_t n_q, int n_chunks, int32_t *__restrict__ out_idx,
                                                                 double *__restrict__ out_d2) {
  const int64_t q = (int64_t)blockIdx.x * O3D_BLOCK + threadIdx.x;
  if (q >= n_q) return;
  double best = INFINITY;
  int32_t bi = -1;
  for (int c = 0; c < n_chunks; c++) {  // ascending target ranges: a strict comparison keeps the lowest index on ties
    const double d = part_d2[(size_t)c * n_q + q];
    const int32_t i = part_idx[(size_t)c * n_q + q];
    if (i >= 0 && (bi < 0 || d < best)) { best = d; bi = i; }
  }
  out_idx[q] = bi;
  if (out_d2) out_d2[q] = bi >= 0 ? best : 0.0;
}

// EvaluateRANSACBasedOnCorrespondence, Registration.cpp:93-116, for a batch of transformations: workgroup (h, c) takes
// hypothesis h and the c-th slice of the correspondence list.  part: [h][c][2] = {good, error2}.  The source is moved by
// T as PointCloud::Transform does (Geometry3D.cpp:77-82: 4x4 * (p, 1), divided by w unless the last row is (0,0,0,1)).
__device__ __forceinline__ bool corr_pair_d2(const double *__restrict__ T, const double *__restrict__ src, const double *__restrict__ tgt,
                                             const int32_t *__restrict__ c, int64_t n_src, int64_t n_tgt, double &d2) {
  const int32_t i = c[0], j = c[1];
  if (i < 0 || i >= n_src || j < 0 || j >= n_tgt) return false;
  const double x = src[3 * (size_t)i], y = src[3 * (size_t)i + 1], z = src[3 * (size_t)i + 2];
  double v[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    double t = T[r] * x;
    t = t + T[4 + r] * y;
    t = t + T[8 + r] * z;
    v[r] = t + T[12 + r];
  }
  if (!(T[3] == 0.0 && T[7] == 0.0 && T[11] == 0.0 && T[15] == 1.0)) { v[0] /= v[3]; v[1] /= v[3]; v[2] /= v[3]; }
  const double dx = v[0] - tgt[3 * (size_t)j], dy = v[1] - tgt[3 * (size_t)j + 1], dz = v[2] - tgt[3 * (size_t)j + 2];
  d2 = dx * dx;
  d2 = d2 + dy * dy;
  d2 = d2 + dz * dz;
  return true;
}

__global__ __launch_bounds__(O3D_BLOCK) void k_ransac_corr_eval(const double *__restrict__ Ts, const double *__restrict__ src,
                                                                 int64_t n_src, const double *__restrict__ tgt, int64_t n_tgt,
                                                                 const int32_t *__restrict__ corres, int64_t n_c, double max_d2,
                                                                 double *__restrict__ part) {
  __shared__ double red[O3D_BLOCK / O3D_WAVE][2];
  const double *T = Ts + 16 * (size_t)blockIdx.x;
  const int64_t per = (n_c + gridDim.y - 1) / gridDim.y;
  const int64_t k0 = (int64_t)blockIdx.y * per, k1 = k0 + per < n_c ? k0 + per : n_c;
  double good = 0.0, err = 0.0;
  for (int64_t k = k0 + threadIdx.x; k < k1; k += O3D_BLOCK) {
    double d2;
    if (corr_pair_d2(T, src, tgt, corres + 2 * k, n_src, n_tgt, d2) && d2 < max_d2) { good += 1.0; err += d2; }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    good += __shfl_xor(good, d, O3D_WAVE);
    err += __shfl_xor(err, d, O3D_WAVE);
  }
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = good; red[threadIdx.x >> 6][1] = err; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double g = red[0][0], e = red[0][1];
    for (int w = 1; w < O3D_BLOCK / O3D_WAVE; w++) { g += red[w][0]; e += red[w][1]; }
    part[2 * ((size_t)blockIdx.x * gridDim.y + blockIdx.y)] = g;
    part[2 * ((size_t)blockIdx.x * gridDim.y + blockIdx.y) + 1] = e;
  }
}

// out: [h][2] = {good, error2}, slices added in order (run-to-run deterministic)
__global__ __launch_bounds__(O3D_BLOCK) void k_ransac_corr_finish(const double *__restrict__ part, int n_hyp, int n_slices,
                                                                   double *__restrict__ out) {
  const int h = blockIdx.x * O3D_BLOCK + threadIdx.x;
  if (h >= n_hyp) return;
  double g = 0.0, e = 0.0;
  for (int c = 0; c < n_slices; c++) { g += part[2 * ((size_t)h * n_slices + c)]; e += part[2 * ((size_t)h * n_slices + c) + 1]; }
  out[2 * (size_t)h] = g;
  out[2 * (size_t)h + 1] = e;
}

__global__ __launch_bounds__(O3D_BLOCK) void k_ransac_corr_mask(const double *__restrict__ T, const double *__restrict__ src, int64_t n_src,
                                                                 const double *__restrict__ tgt, int64_t n_tgt,
                                                                 const int32_t *__restrict__ corres, int64_t n_c, double max_d2,
                                                                 int32_t *__restrict__ mask) {
  for (int64_t k = (int64_t)blockIdx.x * O3D_BLOCK + threadIdx.x; k < n_c; k += (int64_t)gridDim.x * O3D_BLOCK) {
    double d2;
    mask[k] = (corr_pair_d2(T, src, tgt, corres + 2 * k, n_src, n_tgt, d2) && d2 < max_d2) ? 1 : 0;
  }
}

}  // namespace

// d_out: n_hyp x {good, error2} (device).  All pointers device pointers.
int o3d_ransac_corr_eval_device(o3d_ctx *ctx, const double *d_Ts, int n_hyp, const double *d_src, int64_t n_src, const double *d_tgt,
                                int64_t n_tgt, const int32_t *d_corres, int64_t n_c, double max_dist, double *d_out) {
  if (n_hyp <= 0) return O3D_OK;
  int64_t slices = (O3D_NUM_CU * 8 + n_hyp - 1) / n_hyp;
  slices = std::max<int64_t>(1, std::min<int64_t>(slices, (n_c + 4 * O3D_BLOCK - 1) / (4 * O3D_BLOCK)));
  slices = std::min<int64_t>(slices, 65535);
  void *part = nullptr;
  O3D_TRY(o3d_scratch(ctx, 1, sizeof(double) * 2 * (size_t)n_hyp * (size_t)slices, &part));
  hipLaunchKernelGGL(k_ransac_corr_eval, dim3((unsigned)n_hyp, (unsigned)slices), dim3(O3D_BLOCK), 0, ctx->stream, d_Ts, d_src, n_src,
                     d_tgt, n_tgt, d_corres, n_c, max_dist * max_dist, (double *)part);
  hipLaunchKernelGGL(k_ransac_corr_finish, dim3((unsigned)((n_hyp + O3D_BLOCK - 1) / O3D_BLOCK)), dim3(O3D_BLOCK), 0, ctx->stream,
                     (const double *)part, n_hyp, (int)slices, d_out);
  O3D_HIP(ctx, hipGetLastError());
  return O3D_OK;
}

int o3d_ransac_corr_mask_device(o3d_ctx *ctx, const double *d_T, const double *d_src, int64_t n_src, const double *d_tgt, int64_t n_tgt,
                                const int32_t *d_corres, int64_t n_c, double max_dist, int32_t *d_mask) {
  if (n_c <= 0) return O3D_OK;
  const int64_t g = std::max<int64_t>(1, std::min<int64_t>((n_c + O3D_BLOCK - 1) / O3D_BLOCK, O3D_MAX_GRID));
  hipLaunchKernelGGL(k_ransac_corr_mask, dim3((unsigned)g), dim3(O3D_BLOCK), 0, ctx->stream, d_T, d_src, n_src, d_tgt, n_tgt, d_corres,
                     n_c, max_dist * max_dist, d_mask);
  O3D_HIP(ctx, hipGetLastError());
  return O3D_OK;
}

// All pointers are device pointers.  d_query_ids may be null (queries = all source columns, n_q = n_src).
int o3d_feature_nn_device(o3d_ctx *ctx, const double *d_src_feat, int64_t n_src, const double *d_tgt_feat, int64_t n_tgt, int dim,
                          const int32_t *d_query_ids, int64_t n_q, int32_t *d_out_index, double *d_out_d2) {
  if (dim < 1 || dim > kFeatMaxDim)
    return o3d_fail(ctx, O3D_ERR_UNSUPPORTED, "feature matching: dimension %d is outside [1, %d]", dim, kFeatMaxDim);
  if (n_tgt >= ((int64_t)1 << 31) || n_src >= ((int64_t)1 << 31))
    return o3d_fail(ctx, O3D_ERR_UNSUPPORTED, "feature matching: more than 2^31-1 columns");
  if (n_q <= 0) return O3D_OK;
  const int64_t qblocks = (n_q + O3D_WAVE - 1) / O3D_WAVE;
  // enough workgroups to fill 256 CUs several times over, at least 256 target columns per chunk
  int64_t n_chunks = (O3D_NUM_CU * 16 + qblocks - 1) / qblocks;
  n_chunks = std::max<int64_t>(1, std::min<int64_t>(n_chunks, (n_tgt + 255) / 256));
  n_chunks = std::min<int64_t>(n_chunks, 65535);
  const int64_t per_chunk = n_tgt > 0 ? (n_tgt + n_chunks - 1) / n_chunks : 1;
  void *pd = nullptr, *pi = nullptr;
  O3D_TRY(o3d_scratch(ctx, 1, sizeof(double) * (size_t)n_chunks * (size_t)n_q, &pd));
  O3D_TRY(o3d_scratch(ctx, 3, sizeof(int32_t) * (size_t)n_chunks * (size_t)n_q, &pi));
  const dim3 grid((unsigned)qblocks, (unsigned)n_chunks);
  if (dim == 33) {
    hipLaunchKernelGGL(k_feature_nn<33>, grid, dim3(O3D_WAVE), 0, ctx->stream, d_src_feat, n_src, d_tgt_feat, n_tgt, d_query_ids, n_q,
                       per_chunk, (double *)pd, (int32_t *)pi);
  } else {
    const size_t lds = sizeof(double) * O3D_WAVE * (size_t)dim;
    hipLaunchKernelGGL(k_feature_nn_generic, grid, dim3(O3D_WAVE), lds, ctx->stream, d_src_feat, n_src, d_tgt_feat, n_tgt, dim,
                       d_query_ids, n_q, per_chunk, (double *)pd, (int32_t *)pi);
  }
  hipLaunchKernelGGL(k_feature_nn_merge, dim3((unsigned)((n_q + O3D_BLOCK - 1) / O3D_BLOCK)), dim3(O3D_BLOCK), 0, ctx->stream,
                     (const double *)pd, (const int32_t *)pi, n_q, (int)n_chunks, d_out_index, d_out_d2);
  O3D_HIP(ctx, hipGetLastError());
  return O3D_OK;
}
