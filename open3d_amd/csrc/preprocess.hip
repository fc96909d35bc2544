// Preprocessing kernels of the ICP path: PointCloud::VoxelDownSample (reference geometry/PointCloud.cpp:273-306) and
// PointCloud::EstimateNormals (geometry/EstimateNormals.cpp:320-351 with ComputeNormal :215-252 and FastEigen3x3
// :130-213).  Both are HBM-bound integer / gather work on the same sort + scan + cell-grid machinery as the ICP pass.
#include <math.h>

#include <algorithm>

#include <cstdlib>

#include "o3d_internal.h"

namespace {

int grid_for(int64_t n) {
  int64_t g = (n + O3D_BLOCK - 1) / O3D_BLOCK;
  if (g < 1) g = 1;
  if (g > O3D_MAX_GRID) g = O3D_MAX_GRID;
  return (int)g;
}
int ceil_log2(uint64_t v) {
  int b = 0;
  while (b < 63 && (1ull << b) < v) b++;
  return b;
}

// =====================================================================================================================
// VoxelDownSample
// =====================================================================================================================
struct VoxelParams {
  double vmin[3];  // GetMinBound() - voxel_size / 2  (PointCloud.cpp:279)
  double voxel;
  int bx, by;      // key = (iz << (bx + by)) | (iy << bx) | ix
};

// voxel_index = int(floor((p - voxel_min_bound) / voxel_size)) per axis, true division (PointCloud.cpp:289-290).
// Non-finite points have no voxel: sentinel key, sorted behind every real voxel.
__global__ __launch_bounds__(O3D_BLOCK) void k_voxel_keys(const double *__restrict__ xyz, int64_t n, VoxelParams vp,
                                                           uint64_t *__restrict__ keys, uint32_t *__restrict__ vals) {
  for (int64_t i = (int64_t)blockIdx.x * O3D_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * O3D_BLOCK) {
    const double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    uint64_t key = ~0ull;
    if (isfinite(x) && isfinite(y) && isfinite(z)) {
      const uint64_t ix = (uint64_t)(int)floor((x - vp.vmin[0]) / vp.voxel);
      const uint64_t iy = (uint64_t)(int)floor((y - vp.vmin[1]) / vp.voxel);
      const uint64_t iz = (uint64_t)(int)floor((z - vp.vmin[2]) / vp.voxel);
      key = (iz << (vp.bx + vp.by)) | (iy << vp.bx) | ix;
    }
    keys[i] = key;
    vals[i] = (uint32_t)i;
  }
}

__global__ __launch_bounds__(O3D_BLOCK) void k_segment_heads(const uint64_t *__restrict__ keys, int64_t n, uint32_t *__restrict__ head) {
  for (int64_t p = (int64_t)blockIdx.x * O3D_BLOCK + threadIdx.x; p < n; p += (int64_t)gridDim.x * O3D_BLOCK)
    head[p] = (p == 0 || keys[p] != keys[p - 1]) ? 1u : 0u;
}

// seg_id = exclusive scan of the head flags (so the head of segment m has seg_id == m); record segment starts.
__global__ __launch_bounds__(O3D_BLOCK) void k_segment_starts(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ seg_id,
                                                               int64_t n, uint32_t *__restrict__ seg_start) {
  for (int64_t p = (int64_t)blockIdx.x * O3D_BLOCK + threadIdx.x; p < n; p += (int64_t)gridDim.x * O3D_BLOCK)
    if (p == 0 || keys[p] != keys[p - 1]) seg_start[seg_id[p]] = (uint32_t)p;
}

// One lane per voxel: AccumulatedPoint::AddPoint in input order (stable sort => ascending original index inside the
// segment), then GetAveragePoint / Normal / Color (PointCloud.cpp:190-210).  Normals with a NaN component are skipped
// in the numerator while the divisor stays the full count, as in the reference.
__global__ __launch_bounds__(O3D_BLOCK) void k_voxel_reduce(const double *__restrict__ xyz, const double *__restrict__ nrm,
                                                             const double *__restrict__ col, const uint64_t *__restrict__ keys,
                                                             const uint32_t *__restrict__ vals, const uint32_t *__restrict__ seg_start,
                                                             int64_t m, int64_t n_finite, VoxelParams vp, double *__restrict__ out_xyz,
                                                             double *__restrict__ out_nrm, double *__restrict__ out_col,
                                                             int32_t *__restrict__ out_vox) {
  for (int64_t v = (int64_t)blockIdx.x * O3D_BLOCK + threadIdx.x; v < m; v += (int64_t)gridDim.x * O3D_BLOCK) {
    const uint32_t s = seg_start[v];
    const uint32_t e = (v + 1 < m) ? seg_start[v + 1] : (uint32_t)n_finite;
    double sp[3] = {0, 0, 0}, sn[3] = {0, 0, 0}, sc[3] = {0, 0, 0};
    for (uint32_t p = s; p < e; p++) {
      const size_t i = vals[p];
      sp[0] += xyz[3 * i]; sp[1] += xyz[3 * i + 1]; sp[2] += xyz[3 * i + 2];
      if (nrm) {
        const double a = nrm[3 * i], b = nrm[3 * i + 1], c = nrm[3 * i + 2];
        if (!isnan(a) && !isnan(b) && !isnan(c)) { sn[0] += a; sn[1] += b; sn[2] += c; }
      }
      if (col) { sc[0] += col[3 * i]; sc[1] += col[3 * i + 1]; sc[2] += col[3 * i + 2]; }
    }
    const double cnt = (double)(e - s);
    for (int d = 0; d < 3; d++) {
      out_xyz[3 * v + d] = sp[d] / cnt;
      if (nrm) out_nrm[3 * v + d] = sn[d] / cnt;
      if (col) out_col[3 * v + d] = sc[d] / cnt;
    }
    if (out_vox) {
      const uint64_t key = keys[s];
      out_vox[3 * v] = (int32_t)(key & ((1ull << vp.bx) - 1));
      out_vox[3 * v + 1] = (int32_t)((key >> vp.bx) & ((1ull << vp.by) - 1));
      out_vox[3 * v + 2] = (int32_t)(key >> (vp.bx + vp.by));
    }
  }
}

// =====================================================================================================================
// EstimateNormals
// =====================================================================================================================
__device__ __forceinline__ void cross3(const double *a, const double *b, double *c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

// ComputeEigenvector0, EstimateNormals.cpp:40-68
__device__ void eigvec0(const double (*A)[3], double eval0, double *out) {
  const double row0[3] = {A[0][0] - eval0, A[0][1], A[0][2]};
  const double row1[3] = {A[0][1], A[1][1] - eval0, A[1][2]};
  const double row2[3] = {A[0][2], A[1][2], A[2][2] - eval0};
  double r0xr1[3], r0xr2[3], r1xr2[3];
  cross3(row0, row1, r0xr1);
  cross3(row0, row2, r0xr2);
  cross3(row1, row2, r1xr2);
  const double d0 = r0xr1[0] * r0xr1[0] + r0xr1[1] * r0xr1[1] + r0xr1[2] * r0xr1[2];
  const double d1 = r0xr2[0] * r0xr2[0] + r0xr2[1] * r0xr2[1] + r0xr2[2] * r0xr2[2];
  const double d2 = r1xr2[0] * r1xr2[0] + r1xr2[1] * r1xr2[1] + r1xr2[2] * r1xr2[2];
  double dmax = d0;
  int imax = 0;
  if (d1 > dmax) { dmax = d1; imax = 1; }
  if (d2 > dmax) { imax = 2; }
  if (imax == 0) { const double s = sqrt(d0); out[0] = r0xr1[0] / s; out[1] = r0xr1[1] / s; out[2] = r0xr1[2] / s; }
  else if (imax == 1) { const double s = sqrt(d1); out[0] = r0xr2[0] / s; out[1] = r0xr2[1] / s; out[2] = r0xr2[2] / s; }
  else { const double s = sqrt(d2); out[0] = r1xr2[0] / s; out[1] = r1xr2[1] / s; out[2] = r1xr2[2] / s; }
}

// ComputeEigenvector1, EstimateNormals.cpp:70-128
__device__ void eigvec1(const double (*A)[3], const double *evec0, double eval1, double *out) {
  double U[3], V[3];
  if (fabs(evec0[0]) > fabs(evec0[1])) {
    const double inv = 1 / sqrt(evec0[0] * evec0[0] + evec0[2] * evec0[2]);
    U[0] = -evec0[2] * inv; U[1] = 0; U[2] = evec0[0] * inv;
  } else {
    const double inv = 1 / sqrt(evec0[1] * evec0[1] + evec0[2] * evec0[2]);
    U[0] = 0; U[1] = evec0[2] * inv; U[2] = -evec0[1] * inv;
  }
  cross3(evec0, U, V);
  const double AU[3] = {A[0][0] * U[0] + A[0][1] * U[1] + A[0][2] * U[2], A[0][1] * U[0] + A[1][1] * U[1] + A[1][2] * U[2],
                        A[0][2] * U[0] + A[1][2] * U[1] + A[2][2] * U[2]};
  const double AV[3] = {A[0][0] * V[0] + A[0][1] * V[1] + A[0][2] * V[2], A[0][1] * V[0] + A[1][1] * V[1] + A[1][2] * V[2],
                        A[0][2] * V[0] + A[1][2] * V[1] + A[2][2] * V[2]};
  double m00 = U[0] * AU[0] + U[1] * AU[1] + U[2] * AU[2] - eval1;
  double m01 = U[0] * AV[0] + U[1] * AV[1] + U[2] * AV[2];
  double m11 = V[0] * AV[0] + V[1] * AV[1] + V[2] * AV[2] - eval1;
  const double a00 = fabs(m00), a01 = fabs(m01), a11 = fabs(m11);
  if (a00 >= a11) {
    if (fmax(a00, a01) > 0) {
      if (a00 >= a01) { m01 /= m00; m00 = 1 / sqrt(1 + m01 * m01); m01 *= m00; }
      else { m00 /= m01; m01 = 1 / sqrt(1 + m00 * m00); m00 *= m01; }
      for (int d = 0; d < 3; d++) out[d] = m01 * U[d] - m00 * V[d];
    } else {
      for (int d = 0; d < 3; d++) out[d] = U[d];
    }
  } else {
    if (fmax(a11, a01) > 0) {
      if (a11 >= a01) { m01 /= m11; m11 = 1 / sqrt(1 + m01 * m01); m01 *= m11; }
      else { m11 /= m01; m01 = 1 / sqrt(1 + m11 * m11); m11 *= m01; }
      for (int d = 0; d < 3; d++) out[d] = m11 * U[d] - m01 * V[d];
    } else {
      for (int d = 0; d < 3; d++) out[d] = U[d];
    }
  }
}

// FastEigen3x3, EstimateNormals.cpp:130-213 (A.maxCoeff() is the SIGNED maximum, :137)
__device__ void fast_eigen3x3(const double *cov, double *normal) {
  double A[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) A[i][j] = cov[i * 3 + j];
  double max_coeff = A[0][0];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      if (A[i][j] > max_coeff) max_coeff = A[i][j];
  if (max_coeff == 0) { normal[0] = normal[1] = normal[2] = 0; return; }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) A[i][j] /= max_coeff;
  const double norm = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
  if (norm > 0) {
    double eval[3], evec0[3], evec1[3], evec2[3];
    const double q = (A[0][0] + A[1][1] + A[2][2]) / 3;
    const double b00 = A[0][0] - q, b11 = A[1][1] - q, b22 = A[2][2] - q;
    const double p = sqrt((b00 * b00 + b11 * b11 + b22 * b22 + norm * 2) / 6);
    const double c00 = b11 * b22 - A[1][2] * A[1][2];
    const double c01 = A[0][1] * b22 - A[1][2] * A[0][2];
    const double c02 = A[0][1] * A[1][2] - b11 * A[0][2];
    const double det = (b00 * c00 - A[0][1] * c01 + A[0][2] * c02) / (p * p * p);
    double half_det = det * 0.5;
    half_det = fmin(fmax(half_det, -1.0), 1.0);
    const double angle = acos(half_det) / (double)3;
    const double two_thirds_pi = 2.09439510239319549;
    const double beta2 = cos(angle) * 2;
    const double beta0 = cos(angle + two_thirds_pi) * 2;
    const double beta1 = -(beta0 + beta2);
    eval[0] = q + p * beta0;
    eval[1] = q + p * beta1;
    eval[2] = q + p * beta2;
    if (half_det >= 0) {
      eigvec0(A, eval[2], evec2);
      if (eval[2] < eval[0] && eval[2] < eval[1]) { for (int d = 0; d < 3; d++) normal[d] = evec2[d]; return; }
      eigvec1(A, evec2, eval[1], evec1);
      if (eval[1] < eval[0] && eval[1] < eval[2]) { for (int d = 0; d < 3; d++) normal[d] = evec1[d]; return; }
      cross3(evec1, evec2, evec0);
      for (int d = 0; d < 3; d++) normal[d] = evec0[d];
    } else {
      eigvec0(A, eval[0], evec0);
      if (eval[0] < eval[1] && eval[0] < eval[2]) { for (int d = 0; d < 3; d++) normal[d] = evec0[d]; return; }
      eigvec1(A, evec0, eval[1], evec1);
      if (eval[1] < eval[0] && eval[1] < eval[2]) { for (int d = 0; d < 3; d++) normal[d] = evec1[d]; return; }
      cross3(evec0, evec1, evec2);
      for (int d = 0; d < 3; d++) normal[d] = evec2[d];
    }
  } else {
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) A[i][j] *= max_coeff;
    if (A[0][0] < A[1][1] && A[0][0] < A[2][2]) { normal[0] = 1; normal[1] = 0; normal[2] = 0; }
    else if (A[1][1] < A[0][0] && A[1][1] < A[2][2]) { normal[0] = 0; normal[1] = 1; normal[2] = 0; }
    else { normal[0] = 0; normal[1] = 0; normal[2] = 1; }
  }
}

// fast_normal_computation = false (EstimateNormals.cpp:247-251): column 0 of
// Eigen::SelfAdjointEigenSolver<Matrix3d>::compute(cov, ComputeEigenvectors).eigenvectors() -- the lower triangle scaled by
// its largest magnitude, the closed-form 3x3 tridiagonalisation, implicit symmetric QR steps with Wilkinson shifts (at most
// 90), eigenvalues sorted ascending with their columns.  Operation for operation what oracle/o3d_oracle.c
// (orc_self_adjoint_eigen3x3) restates from Eigen 3.4: the SIGN of the column is whatever that sequence leaves.
__device__ __forceinline__ void make_givens(double p, double q, double &c, double &s) {
  if (q == 0.0) { c = p < 0.0 ? -1.0 : 1.0; s = 0.0; }
  else if (p == 0.0) { c = 0.0; s = q < 0.0 ? 1.0 : -1.0; }
  else if (fabs(p) > fabs(q)) {
    const double t = q / p;
    double u = sqrt(1.0 + t * t);
    if (p < 0.0) u = -u;
    c = 1.0 / u;
    s = -t * c;
  } else {
    const double t = p / q;
    double u = sqrt(1.0 + t * t);
    if (q < 0.0) u = -u;
    s = -1.0 / u;
    c = -t * s;
  }
}
__device__ __forceinline__ double eigen_hypot(double x, double y) {
  x = fabs(x); y = fabs(y);
  if (isinf(x) || isinf(y)) return INFINITY;
  if (isnan(x) || isnan(y)) return NAN;
  const double p = x > y ? x : y;
  if (p == 0.0) return 0.0;
  const double qp = (y < x ? y : x) / p;
  return p * sqrt(1.0 + qp * qp);
}
__device__ void self_adjoint_eigen3_col0(const double *cov, double *normal) {
  double m00 = cov[0], m10 = cov[3], m11 = cov[4], m20 = cov[6], m21 = cov[7], m22 = cov[8];
  double scale = fmax(fmax(fmax(fabs(m00), fabs(m10)), fmax(fabs(m20), fabs(m11))), fmax(fabs(m21), fabs(m22)));
  if (scale == 0.0) scale = 1.0;
  m00 /= scale; m10 /= scale; m20 /= scale; m11 /= scale; m21 /= scale; m22 /= scale;
  double d0 = m00, d1, d2, e0, e1;  // diagonal and sub-diagonal
  double Q[9];                      // column-major
  const double v1norm2 = m20 * m20;
  if (v1norm2 <= 2.2250738585072014e-308) {
    d1 = m11; d2 = m22; e0 = m10; e1 = m21;
#pragma unroll
    for (int i = 0; i < 9; i++) Q[i] = (i % 4 == 0) ? 1.0 : 0.0;
  } else {
    const double beta = sqrt(m10 * m10 + v1norm2);
    const double inv_beta = 1.0 / beta;
    const double m01 = m10 * inv_beta, m02 = m20 * inv_beta;
    const double q = 2.0 * m01 * m21 + m02 * (m22 - m11);
    d1 = m11 + m02 * q;
    d2 = m22 - m02 * q;
    e0 = beta;
    e1 = m21 - m01 * q;
    Q[0] = 1.0; Q[1] = 0.0; Q[2] = 0.0;
    Q[3] = 0.0; Q[4] = m01; Q[5] = m02;
    Q[6] = 0.0; Q[7] = m02; Q[8] = -m01;
  }
  double diag[3] = {d0, d1, d2}, sub[2] = {e0, e1};
  int end = 2, start = 0, iter = 0;
  const double precision_inv = 1.0 / 2.220446049250313e-16;
  while (end > 0) {
    for (int i = start; i < end; i++) {
      if (fabs(sub[i]) < 2.2250738585072014e-308) sub[i] = 0.0;
      else {
        const double scaled = precision_inv * sub[i];
        if (scaled * scaled <= fabs(diag[i]) + fabs(diag[i + 1])) sub[i] = 0.0;
      }
    }
    while (end > 0 && sub[end - 1] == 0.0) end--;
    if (end <= 0) break;
    iter++;
    if (iter > 90) break;
    start = end - 1;
    while (start > 0 && sub[start - 1] != 0.0) start--;
    const double td = (diag[end - 1] - diag[end]) * 0.5;
    const double e = sub[end - 1];
    double mu = diag[end];
    if (td == 0.0) mu -= fabs(e);
    else if (e != 0.0) {
      const double e2 = e * e;
      const double h = eigen_hypot(td, e);
      if (e2 == 0.0) mu -= e / ((td + (td > 0.0 ? h : -h)) / e);
      else mu -= e2 / (td + (td > 0.0 ? h : -h));
    }
    double x = diag[start] - mu;
    double z = sub[start];
    for (int k = start; k < end && z != 0.0; k++) {
      double c, s;
      make_givens(x, z, c, s);
      const double sdk = s * diag[k] + c * sub[k];
      const double dkp1 = s * sub[k] + c * diag[k + 1];
      diag[k] = c * (c * diag[k] - s * sub[k]) - s * (c * sub[k] - s * diag[k + 1]);
      diag[k + 1] = s * sdk + c * dkp1;
      sub[k] = c * sdk - s * dkp1;
      if (k > start) sub[k - 1] = c * sub[k - 1] - s * z;
      x = sub[k];
      if (k < end - 1) {
        z = -s * sub[k + 1];
        sub[k + 1] = c * sub[k + 1];
      }
      if (!(c == 1.0 && s == 0.0)) {
        for (int r = 0; r < 3; r++) {
          const double xi = Q[3 * k + r], yi = Q[3 * (k + 1) + r];
          Q[3 * k + r] = c * xi - s * yi;
          Q[3 * (k + 1) + r] = s * xi + c * yi;
        }
      }
    }
  }
  int col = 0;  // the column the selection sort would bring to the front: the first minimum
  if (iter <= 90) {
    if (diag[1] < diag[col]) col = 1;
    if (diag[2] < diag[col]) col = 2;
  }
  normal[0] = Q[3 * col]; normal[1] = Q[3 * col + 1]; normal[2] = Q[3 * col + 2];
}

constexpr int kNrmBlock = 64;  // one wavefront per workgroup: the k-best lists live in LDS, [slot][lane]
constexpr int kNrmLoads = 12;  // the LDS lists leave ~1 wavefront per SIMD: memory-level parallelism has to come from the lane
constexpr int kMaxKnn = 256;  // 8 B per list entry and lane: 128 KiB of the CU's 160 KiB LDS for one wavefront

// Dynamic LDS beyond 64 KiB has to be allowed per kernel.
template <typename K>
int allow_lds(o3d_ctx *ctx, K kernel, size_t lds) {
  if (lds > 65536)
    O3D_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  return O3D_OK;
}

// k-best list of one lane.  While candidates arrive it is a binary MAX-heap on (d2, index) -- the worst kept entry is the
// root, an insertion moves at most log2(k) entries (a sorted list moves k/2 on average and, lanes inserting at different
// places, nearly k per wavefront step) -- and klist_sort turns it in place into the ascending order FLANN returns and
// ComputeNormal sums in.  The lists are what limits the wavefronts per CU, so an entry is 8 bytes of LDS: the distance
// ROUNDED TO FLOAT and the position in the cell-ordered cloud.  Rounding is monotone, so two different keys order their
// entries exactly as the fp64 distances would; two EQUAL keys say nothing, and then -- only then -- both distances are
// recomputed in fp64 from the cloud and the query (the arithmetic of the visit: ((qx-x)^2 + (qy-y)^2) + (qz-z)^2) and
// ties between those go to the lower original index (pts[pos].w).  The kept set and its order are the exact ones.
struct KList {
  float *key;     // LDS, stride kNrmBlock: (float)d2
  uint32_t *pos;  // LDS, stride kNrmBlock: position in the cell-ordered cloud
  const double4 *pts;  // the cell-ordered cloud (w = original index)
  double qx, qy, qz;   // the query
  int cap, count;
  double thr;     // strict bound on d2 (INFINITY for pure KNN)
};
constexpr size_t kListEntryBytes = sizeof(float) + sizeof(uint32_t);

__device__ __forceinline__ void klist_bind(KList &l, unsigned char *smem, int lane, int knn, const double4 *pts) {
  l.key = reinterpret_cast<float *>(smem) + lane;
  l.pos = reinterpret_cast<uint32_t *>(smem + sizeof(float) * kNrmBlock * (size_t)knn) + lane;
  l.pts = pts;
  l.cap = knn;
}
__device__ __forceinline__ void klist_query(KList &l, double qx, double qy, double qz) {
  l.qx = qx; l.qy = qy; l.qz = qz;
  l.count = 0;
}

__device__ __forceinline__ uint32_t kl_index(const KList &l, uint32_t pos) {
  return (uint32_t)__double_as_longlong(l.pts[pos].w);
}
// the fp64 distance of the entry at `pos`, as the visit computed it
__device__ __forceinline__ double kl_exact(const KList &l, uint32_t pos, uint32_t *idx = nullptr) {
  const double4 t = l.pts[pos];
  const double ddx = l.qx - t.x, ddy = l.qy - t.y, ddz = l.qz - t.z;
  double d2 = ddx * ddx;
  d2 = d2 + ddy * ddy;
  d2 = d2 + ddz * ddz;
  if (idx) *idx = (uint32_t)__double_as_longlong(t.w);
  return d2;
}
// entry 1 < entry 2 in (d2, index) order.  The equal-key path is rare and kept out of line: inlined at every comparison of
// the heap code it pushed the kernels into scratch memory.
__device__ __noinline__ bool kl_less_exact(const double4 *pts, double qx, double qy, double qz, uint32_t p1, uint32_t p2) {
  const double4 a = pts[p1], b = pts[p2];
  double ax = qx - a.x, ay = qy - a.y, az = qz - a.z;
  double e1 = ax * ax;
  e1 = e1 + ay * ay;
  e1 = e1 + az * az;
  double bx = qx - b.x, by = qy - b.y, bz = qz - b.z;
  double e2 = bx * bx;
  e2 = e2 + by * by;
  e2 = e2 + bz * bz;
  return e1 < e2 || (e1 == e2 && (uint32_t)__double_as_longlong(a.w) < (uint32_t)__double_as_longlong(b.w));
}
__device__ __forceinline__ bool kl_less(const KList &l, float k1, uint32_t p1, float k2, uint32_t p2) {
  if (k1 != k2) return k1 < k2;
  if (p1 == p2) return false;
  return kl_less_exact(l.pts, l.qx, l.qy, l.qz, p1, p2);
}

// Can a candidate at distance d2 still enter the list?  (No: it is farther than the worst kept entry -- a larger key
// means a larger distance -- or outside the radius.  Yes may still be refused by klist_add when the keys are equal.)
__device__ __forceinline__ bool klist_may_take(const KList &l, double d2) {
  return d2 < l.thr && (l.count < l.cap || (float)d2 <= l.key[0]);
}
// An upper bound of the k-th distance of a full list (the key may have been rounded down by half an ulp).
__device__ __forceinline__ double klist_kth_upper(const KList &l) {
  return (double)nextafterf(l.key[0], INFINITY);
}

// Put (key, pos) into the hole at the root of the heap of `n` entries and let it sink.
__device__ __forceinline__ void klist_sink(KList &l, int n, float d, uint32_t pos) {
  int i = 0;
  for (;;) {
    int c = 2 * i + 1;
    if (c >= n) break;
    float cd = l.key[c * kNrmBlock];
    uint32_t cp = l.pos[c * kNrmBlock];
    if (c + 1 < n) {
      const float rd = l.key[(c + 1) * kNrmBlock];
      const uint32_t rp = l.pos[(c + 1) * kNrmBlock];
      if (kl_less(l, cd, cp, rd, rp)) { c++; cd = rd; cp = rp; }
    }
    if (!kl_less(l, d, pos, cd, cp)) break;
    l.key[i * kNrmBlock] = cd;
    l.pos[i * kNrmBlock] = cp;
    i = c;
  }
  l.key[i * kNrmBlock] = d;
  l.pos[i * kNrmBlock] = pos;
}

__device__ __forceinline__ void klist_add(KList &l, double d2, uint32_t pos) {
  if (!(d2 < l.thr)) return;
  const float d = (float)d2;
  if (l.count == l.cap) {  // replace the worst entry if the candidate is better
    if (!kl_less(l, d, pos, l.key[0], l.pos[0])) return;
    klist_sink(l, l.cap, d, pos);
    return;
  }
  int i = l.count++;
  while (i > 0) {  // rise
    const int p = (i - 1) >> 1;
    const float pd = l.key[p * kNrmBlock];
    const uint32_t pp = l.pos[p * kNrmBlock];
    if (!kl_less(l, pd, pp, d, pos)) break;
    l.key[i * kNrmBlock] = pd;
    l.pos[i * kNrmBlock] = pp;
    i = p;
  }
  l.key[i * kNrmBlock] = d;
  l.pos[i * kNrmBlock] = pos;
}

// Heap -> ascending by (d2, index), in place.  Call once, after the last klist_add and before reading the entries in order.
__device__ __forceinline__ void klist_sort(KList &l) {
  for (int end = l.count - 1; end > 0; end--) {
    const float d = l.key[end * kNrmBlock];
    const uint32_t pos = l.pos[end * kNrmBlock];
    l.key[end * kNrmBlock] = l.key[0];
    l.pos[end * kNrmBlock] = l.pos[0];
    klist_sink(l, end, d, pos);
  }
}

// Shell 1 of the cube around cell (cx, cy, cz), the whole 3x3x3 block: nine rows (dz outer, dy inner), each ONE run of the
// cell-ordered cloud.  The nine pairs of bounds are loaded together and the runs are walked as one flattened sequence,
// kNrmLoads candidates in flight at a time across row ends -- at ~1 wavefront per SIMD every dependent round trip is
// exposed, and row by row there were two per row (bounds, candidates) plus one per further kNrmLoads candidates.
// visit(point, position, real): real = 0 marks a padding slot of the last batch (`pad` is any valid position).
// The nine runs (dz outer, dy inner) of the 3x3x3 block around cell (cx, cy, cz) in the cell-ordered cloud.  Bricked
// table (sparse scenes, grid.hip): the rows of a cell inside the grid lie in its brick at fixed strides; a cell outside
// the grid is a cell side from every point (the margin) and sees nothing.
__device__ __forceinline__ void block3_bounds(const uint32_t *__restrict__ cell_start, const GridParams &gp, int cx, int cy, int cz,
                                              uint32_t (&rs)[9], uint32_t (&re)[9]) {
  if (gp.brick_bits) {
    const bool inside = cx >= 0 && cx < gp.nx && cy >= 0 && cy < gp.ny && cz >= 0 && cz < gp.nz;
    const uint32_t rowc = inside ? o3d_row_index(gp, cx, cy, cz) : 0u;
    const int sy = o3d_row_stride_y(gp), sz = o3d_row_stride_z(gp);
#pragma unroll
    for (int r = 0; r < 9; r++) {
      const uint32_t at = rowc + (uint32_t)((r % 3 - 1) * sy + (r / 3 - 1) * sz);
      rs[r] = inside ? cell_start[at] : 0u;
      re[r] = inside ? cell_start[at + 3] : 0u;
    }
    return;
  }
  const int xa = max(cx - 1, -1), xb = min(cx + 1, gp.nx);
#pragma unroll
  for (int r = 0; r < 9; r++) {
    const int zz = cz + r / 3 - 1, yy = cy + r % 3 - 1;
    const bool ok = zz >= 0 && zz < gp.nz && yy >= 0 && yy < gp.ny;
    const uint32_t row = ok ? ((uint32_t)zz * (uint32_t)gp.ny + (uint32_t)yy) * (uint32_t)gp.nxp : 0u;
    const uint32_t s0 = cell_start[row + (uint32_t)(xa + 1)], e0 = cell_start[row + (uint32_t)(xb + 2)];
    rs[r] = ok ? s0 : 0u;
    re[r] = ok ? e0 : 0u;
  }
}

template <typename F>
__device__ __forceinline__ void walk_block3(const double4 *__restrict__ pts, const uint32_t *__restrict__ cell_start,
                                            const GridParams &gp, int cx, int cy, int cz, uint32_t pad, F &&visit) {
  uint32_t rs[9], re[9];
  block3_bounds(cell_start, gp, cx, cy, cz, rs, re);
  int r = 0;
  uint32_t j = rs[0], e = re[0];
  auto next_row = [&]() {  // (static indices only: the bounds stay in registers)
    while (j >= e && r < 8) {
      r++;
#pragma unroll
      for (int i = 1; i < 9; i++) {
        j = r == i ? rs[i] : j;
        e = r == i ? re[i] : e;
      }
    }
  };
  next_row();
  while (j < e) {
    uint32_t jj[kNrmLoads];
    bool real[kNrmLoads];
#pragma unroll
    for (int u = 0; u < kNrmLoads; u++) {
      real[u] = j < e;
      jj[u] = real[u] ? j : pad;
      if (real[u]) { j++; next_row(); }
    }
    double4 tq[kNrmLoads];
#pragma unroll
    for (int u = 0; u < kNrmLoads; u++) tq[u] = pts[jj[u]];
#pragma unroll
    for (int u = 0; u < kNrmLoads; u++) visit(tq[u], jj[u], real[u]);
  }
}

// Shell R >= 2 of that cube (the cells at Chebyshev distance exactly R), row by row.
template <typename F>
__device__ __forceinline__ void walk_shell(const double4 *__restrict__ pts, const uint32_t *__restrict__ cell_start,
                                           const GridParams &gp, int cx, int cy, int cz, int R, F &&visit) {
  for (int dz = -R; dz <= R; dz++) {
    const int zz = cz + dz;
    if (zz < 0 || zz >= gp.nz) continue;
    for (int dy = -R; dy <= R; dy++) {
      const int yy = cy + dy;
      if (yy < 0 || yy >= gp.ny) continue;
      const uint32_t row = ((uint32_t)zz * (uint32_t)gp.ny + (uint32_t)yy) * (uint32_t)gp.nxp;
      const bool face = dz == -R || dz == R || dy == -R || dy == R;
      const int nseg = face ? 1 : 2;
      for (int sgi = 0; sgi < nseg; sgi++) {
        int xa, xb;  // unpadded cell range of this piece of the row
        if (face) { xa = cx - R; xb = cx + R; }
        else if (sgi == 0) { xa = xb = cx - R; }
        else { xa = xb = cx + R; }
        xa = max(xa, -1); xb = min(xb, gp.nx);
        if (xa > xb) continue;
        const uint32_t s = cell_start[row + (uint32_t)(xa + 1)], e = cell_start[row + (uint32_t)(xb + 2)];
        for (uint32_t j0 = s; j0 < e; j0 += kNrmLoads) {  // candidate loads in flight; processed in run order
          double4 tq[kNrmLoads];
#pragma unroll
          for (int u = 0; u < kNrmLoads; u++) tq[u] = pts[min(j0 + (uint32_t)u, e - 1)];
#pragma unroll
          for (int u = 0; u < kNrmLoads; u++) visit(tq[u], j0 + (uint32_t)u, j0 + (uint32_t)u < e);
        }
      }
    }
  }
}

// Per point: neighbours by KNN / radius / hybrid search over the cloud's own cell grid, cumulants in neighbour order,
// covariance, FastEigen3x3, fallbacks and orientation against an existing normal (EstimateNormals.cpp:330-350).
//   MODE 0 KNN: shells of cells are added until the k-th distance is inside the searched cube.
//   MODE 1 Radius: every point with d2 < thr in the 3x3x3 block; cumulants summed in traversal order (the reference
//          sums in ascending distance: same set, rounding-level difference only).
//   MODE 2 Hybrid: the max_nn nearest with d2 < thr.
//   Positions [p_lo, p_hi) of the cell-ordered cloud are processed (a shard of it when the cloud is split over GPUs);
//   by_pos = 0: the normal of a point goes to normals[original index]; 1: to normals[position - p_lo] (shard output).
//   The orientation reference (has_normals) is read from `ref` by original index (ref may alias normals when by_pos = 0).
template <int MODE>
__global__ __launch_bounds__(kNrmBlock) void k_estimate_normals(const double4 *__restrict__ pts, int64_t p_lo, int64_t p_hi,
                                                                 const uint32_t *__restrict__ cell_start, GridParams gp, int knn,
                                                                 double thr, double *normals, int has_normals, int by_pos,
                                                                 const double *ref, const uint32_t *__restrict__ list,
                                                                 const uint32_t *__restrict__ list_count, double *__restrict__ cov_out) {
  extern __shared__ unsigned char smem[];
  const int lane = threadIdx.x;
  KList l;
  klist_bind(l, smem, lane, knn, pts);
  l.thr = (MODE == 0) ? INFINITY : thr;
  const int max_dim = max(max(gp.nx, gp.ny), gp.nz);
  // list != null: only the positions k_estimate_normals_hist left behind
  const int64_t n_items = list ? (int64_t)*list_count : p_hi - p_lo;

  for (int64_t it = (int64_t)blockIdx.x * kNrmBlock + lane; it < n_items; it += (int64_t)gridDim.x * kNrmBlock) {
    const int64_t p = list ? (int64_t)list[it] : p_lo + it;
    const double4 q = pts[p];
    const size_t qi = (size_t)__double_as_longlong(q.w);
    const size_t oi = by_pos ? (size_t)(p - p_lo) : qi;
    const int cx = min(max(o3d_cell_coord(q.x, gp.gmin[0], gp.inv_h, gp.nx), 0), gp.nx - 1);
    const int cy = min(max(o3d_cell_coord(q.y, gp.gmin[1], gp.inv_h, gp.ny), 0), gp.ny - 1);
    const int cz = min(max(o3d_cell_coord(q.z, gp.gmin[2], gp.inv_h, gp.nz), 0), gp.nz - 1);
    klist_query(l, q.x, q.y, q.z);
    int count = 0;                                  // MODE 1
    double c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};      // cumulants
    auto visit = [&](const double4 &t, uint32_t j, bool real) {
      const double ddx = q.x - t.x, ddy = q.y - t.y, ddz = q.z - t.z;
      double d2 = ddx * ddx;
      d2 = d2 + ddy * ddy;
      d2 = d2 + ddz * ddz;
      if (!real) d2 = INFINITY;  // a padding slot of the batch: rejected by every mode below
      if (MODE == 1) {
        if (d2 < thr) {
          count++;
          c[0] += t.x; c[1] += t.y; c[2] += t.z;
          c[3] += t.x * t.x; c[4] += t.x * t.y; c[5] += t.x * t.z;
          c[6] += t.y * t.y; c[7] += t.y * t.z; c[8] += t.z * t.z;
        }
      } else if (klist_may_take(l, d2)) {
        klist_add(l, d2, j);
      }
    };
    walk_block3(pts, cell_start, gp, cx, cy, cz, (uint32_t)p, visit);
    for (int R = 1; MODE == 0; R++) {  // KNN: further shells of the cube around the query's cell
      if (R > 1) walk_shell(pts, cell_start, gp, cx, cy, cz, R, visit);
      // every unsearched point is at least R*h away (minus rounding); done once the k-th distance is inside
      if (R >= max_dim) break;
      if (l.count == l.cap) {
        const double reach = (double)R * gp.h * (1.0 - 1e-6);
        if (klist_kth_upper(l) <= reach * reach) break;
      }
    }
    int k = count;
    if (MODE != 1) {
      klist_sort(l);
      k = l.count;
      for (int i0 = 0; i0 < k; i0 += kNrmLoads) {  // ComputeNormal :222-233, in the neighbour order FLANN returns
        double4 tq[kNrmLoads];
#pragma unroll
        for (int u = 0; u < kNrmLoads; u++) tq[u] = pts[l.pos[min(i0 + u, k - 1) * kNrmBlock]];
#pragma unroll
        for (int u = 0; u < kNrmLoads; u++) {
          if (i0 + u < k) {
            const double4 t = tq[u];
            c[0] += t.x; c[1] += t.y; c[2] += t.z;
            c[3] += t.x * t.x; c[4] += t.x * t.y; c[5] += t.x * t.z;
            c[6] += t.y * t.y; c[7] += t.y * t.z; c[8] += t.z * t.z;
          }
        }
      }
    }
    double nv[3] = {0.0, 0.0, 1.0};
    if (cov_out && k < 3) cov_out[7 * oi] = 0.0;  // (k_normals_from_cov writes the (0, 0, 1))
    if (k >= 3) {
      for (int i = 0; i < 9; i++) c[i] /= (double)k;
      double cov[9];
      cov[0] = c[3] - c[0] * c[0];
      cov[4] = c[6] - c[1] * c[1];
      cov[8] = c[8] - c[2] * c[2];
      cov[1] = cov[3] = c[4] - c[0] * c[1];
      cov[2] = cov[6] = c[5] - c[0] * c[2];
      cov[5] = cov[7] = c[7] - c[1] * c[2];
      if (cov_out) {  // fast_normal_computation = false: the covariance is handed to k_normals_from_cov
        double *o = cov_out + 7 * oi;
        o[0] = 1.0; o[1] = cov[0]; o[2] = cov[1]; o[3] = cov[2]; o[4] = cov[4]; o[5] = cov[5]; o[6] = cov[8];
        continue;
      }
      fast_eigen3x3(cov, nv);
      if (sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]) == 0.0) {
        if (has_normals) { nv[0] = ref[3 * qi]; nv[1] = ref[3 * qi + 1]; nv[2] = ref[3 * qi + 2]; }
        else { nv[0] = 0.0; nv[1] = 0.0; nv[2] = 1.0; }
      }
      if (has_normals && nv[0] * ref[3 * qi] + nv[1] * ref[3 * qi + 1] + nv[2] * ref[3 * qi + 2] < 0.0) {
        nv[0] *= -1.0; nv[1] *= -1.0; nv[2] *= -1.0;
      }
    }
    if (cov_out) continue;
    normals[3 * oi] = nv[0]; normals[3 * oi + 1] = nv[1]; normals[3 * oi + 2] = nv[2];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same per-point computation for hybrid search with a selection that every lane runs in lock step.  The heap of
// k_estimate_normals makes a wavefront execute an insertion whenever ANY of its lanes has one -- for nearly every candidate
// -- at 38 % lane utilisation, with more scalar (branch, mask) than vector instructions (profiles/r02_c_*).  Here the
// k nearest of the candidates inside the radius are found by counting:
//   walk    every candidate inside the radius is counted into one of kHistBins equal slices of [0, r^2) by its distance
//           (a per-lane histogram of byte counters in LDS) and noted down -- (key, position) -- in the lane's column of a
//           buffer in global memory;
//   prefix  the slice b* in which the k-th nearest lies; every candidate of a lower slice is among the k nearest;
//   place   the lane's notes are read back: those of lower slices go straight to their slice's place in the list
//           (slices are ordered, so the list is ordered up to the order inside a slice), those of slice b* to a short
//           boundary list.  (Rounds 1-2 walked the block a second time here and computed every distance twice: 43 % of
//           the kernel's wave cycles; with the notes the kernel takes 6.8 ms instead of 7.9 at 10 M points.)
//   finish  the boundary list is sorted by a fixed network and its first k - (number below) entries appended; the order
//           inside the slices is repaired by odd-even transposition rounds until a round moves nothing (two or three).
// Keys are the distances rounded to float as in the heap; equal keys are ordered by the exact fp64 distances, then by
// original index (kl_less) -- the kept set and its order are the exact ones, the cumulants are summed in that order.
// A lane this does not fit (more than kBoundaryCap candidates in slice b*, more than kInsideCap - 1 inside the radius)
// puts its point on a list that k_estimate_normals works off afterwards.
//
// What was measured on the way (round 3, Hybrid(4 s, 30) at 10 M points; profiles/r03_d_*): the wavefront is alone or
// nearly alone on its SIMD (the lists take 21 KiB of LDS), so the kernel runs at the speed of one wavefront's instruction
// stream -- VALU busy 26 % of its cycles, 42 % spent in s_waitcnt.  Issuing the next batch's loads before the current batch is
// worked through: 9.9 -> 7.9 ms.  Sharing a batch between the lanes of one cell through LDS (8 x fewer loads): no gain, so
// the load path is not the limit even though its data-return unit reads 77 % busy.  Batching the LDS atomics of the
// placing, batching the transposition's pairs, an insertion sort in its place, wave-compacted notes with cooperative
// placing, deferring the notes' stores: each between 0 and 6 % slower.
constexpr int kHistBins = 32;
constexpr int kBoundaryCap = 8;
constexpr int kHistMaxKnn = 64;
constexpr int kHistLoads = 8;  // (4: +5 %, 6: +0.2 % on the 10 M-point scene)
constexpr int kInsideCap = 255;  // candidates inside the radius that a lane can note down; one more and the heap kernel takes the point

// The workgroups are persistent (grid-stride over the wavefront-sized chunks of the cloud): as many as the LDS lets run at
// once, so that the notes of the walk -- kInsideCap entries per lane -- are a buffer of ~200 MB of which a fifth is touched
// and re-used chunk after chunk, most of it out of L2 and the Infinity Cache.
static int64_t hist_grid(size_t lds_bytes) {
  return (int64_t)O3D_NUM_CU * std::max<int64_t>(1, std::min<int64_t>(8, (int64_t)(160 * 1024 / lds_bytes)));
}
static size_t hist_inside_bytes(int64_t grid) { return sizeof(uint2) * (size_t)grid * 255 * 64; }

__host__ __device__ constexpr size_t hist_lds_bytes(int knn) {
  return (size_t)kNrmBlock * (8 * (size_t)knn + kHistBins + 8 * kBoundaryCap);
}

struct RowBatch {
  double x[kHistLoads], y[kHistLoads], z[kHistLoads];
  uint32_t j, e;  // candidate u is position j + u, and real iff j + u < e
};

__device__ __forceinline__ void batch_load(RowBatch &b, const double4 *__restrict__ pts, uint32_t j, uint32_t e) {
  const double4 *__restrict__ pp = pts + j;
#pragma unroll
  for (int u = 0; u < kHistLoads; u++) {
    b.x[u] = pp[u].x; b.y[u] = pp[u].y; b.z[u] = pp[u].z;
  }
  b.j = j;
  b.e = e;
}

// The state of a lane's walk over the nine rows of its block: batches of kHistLoads consecutive candidates of the current
// row, the row bookkeeping once per batch (walk_block3 does it per candidate, which costs more instructions than the
// candidate itself; walking row by row in lock step makes every row cost its longest lane: 12.6 ms against 11.5).
struct RowWalk {
  int r;
  uint32_t j, e;
  __device__ __forceinline__ void start(const uint32_t (&rs)[9], const uint32_t (&re)[9]) {
    r = 0; j = rs[0]; e = re[0];
    settle(rs, re);
  }
  __device__ __forceinline__ void settle(const uint32_t (&rs)[9], const uint32_t (&re)[9]) {  // (static indices only: the bounds stay in registers)
    while (j >= e && r < 8) {
      r++;
#pragma unroll
      for (int i = 1; i < 9; i++) {
        j = r == i ? rs[i] : j;
        e = r == i ? re[i] : e;
      }
    }
  }
  __device__ __forceinline__ void advance(const uint32_t (&rs)[9], const uint32_t (&re)[9]) {
    j += kHistLoads;
    settle(rs, re);
  }
  __device__ __forceinline__ bool more() const { return j < e; }
};

// Every lane loads its own candidates: one address and 2 * kHistLoads loads at immediate offsets per batch; the slots past
// the row's end read the next cells' points (the array carries kHistLoads entries of padding) and come with real = false.
// Two batches are kept in registers: the loads of the next one are issued before the current one is worked through, so that
// a wavefront -- there are fewer than two per SIMD, the lists live in LDS -- does not wait out a cache miss per batch.
template <typename F>
__device__ __forceinline__ void walk_rows(const double4 *__restrict__ pts, const uint32_t (&rs)[9], const uint32_t (&re)[9], F &&visit) {
  RowWalk w;
  w.start(rs, re);
  if (!w.more()) return;
  RowBatch b0, b1;
  batch_load(b0, pts, w.j, w.e);
  for (;;) {
    w.advance(rs, re);
    const bool more1 = w.more();
    batch_load(b1, pts, more1 ? w.j : b0.j, more1 ? w.e : 0u);  // (nothing left: a batch of padding at an address known to be valid)
    visit(b0);
    if (!more1) break;
    w.advance(rs, re);
    const bool more0 = w.more();
    batch_load(b0, pts, more0 ? w.j : b1.j, more0 ? w.e : 0u);
    visit(b1);
    if (!more0) break;
  }
}

__global__ __launch_bounds__(kNrmBlock) void k_estimate_normals_hist(const double4 *__restrict__ pts, int64_t p_lo, int64_t p_hi,
                                                                      const uint32_t *__restrict__ cell_start, GridParams gp, int knn,
                                                                      double thr, double *normals, int has_normals, int by_pos,
                                                                      const double *ref, uint32_t *__restrict__ fb_list,
                                                                      uint32_t *__restrict__ fb_count, int knn_only,
                                                                      double *__restrict__ cov_out, uint2 *__restrict__ inside) {
  extern __shared__ unsigned char smem[];
  const int lane = threadIdx.x;
  // the workgroup's part of `inside`: [kInsideCap][lane] (key, position) of the candidates the walk found inside the radius
  // (a wave-uniform base and a 32-bit index: one address instruction per access)
  uint2 *const mine_base = inside + (size_t)blockIdx.x * kInsideCap * kNrmBlock;
  const uint32_t lane_bytes = (uint32_t)lane * (uint32_t)sizeof(uint2);
  auto mine = [&](int slot) -> uint2 & {  // (a 32-bit byte offset from a scalar base)
    return *reinterpret_cast<uint2 *>(reinterpret_cast<unsigned char *>(mine_base) + ((uint32_t)slot * (uint32_t)(kNrmBlock * sizeof(uint2)) + lane_bytes));
  };
  // [slot][lane] planes: keys, positions, histogram words (after the prefix: the slices' list places), boundary keys, boundary
  // positions
  float *lkey = reinterpret_cast<float *>(smem) + lane;
  uint32_t *lpos = reinterpret_cast<uint32_t *>(smem + sizeof(float) * kNrmBlock * (size_t)knn) + lane;
  uint32_t *hist = reinterpret_cast<uint32_t *>(smem + 8 * (size_t)kNrmBlock * (size_t)knn) + lane;
  uint32_t *offs = hist;
  float *bkey = reinterpret_cast<float *>(hist + (kHistBins / 4) * kNrmBlock);
  uint32_t *bpos = reinterpret_cast<uint32_t *>(bkey + kBoundaryCap * kNrmBlock);
  const float thrf = (float)thr;
  // slices of equal width in the key; the factor below 1 keeps key * scale under kHistBins for every key up to thrf, the
  // rounding of the three float operations included, so the slice number needs no clamp (any monotone slicing will do)
  const float scale = ((float)kHistBins / thrf) * (1.0f - 1.0f / 1048576.0f);
  KList l;  // only for the exact tie-break of kl_less
  l.pts = pts;

  for (int64_t p = p_lo + (int64_t)blockIdx.x * kNrmBlock + lane; p < p_hi; p += (int64_t)gridDim.x * kNrmBlock) {
    const double4 q = pts[p];
    const size_t qi = (size_t)__double_as_longlong(q.w);
    const size_t oi = by_pos ? (size_t)(p - p_lo) : qi;
    const int cx = min(max(o3d_cell_coord(q.x, gp.gmin[0], gp.inv_h, gp.nx), 0), gp.nx - 1);
    const int cy = min(max(o3d_cell_coord(q.y, gp.gmin[1], gp.inv_h, gp.ny), 0), gp.ny - 1);
    const int cz = min(max(o3d_cell_coord(q.z, gp.gmin[2], gp.inv_h, gp.nz), 0), gp.nz - 1);
    l.qx = q.x; l.qy = q.y; l.qz = q.z;
    uint32_t rs[9], re[9];
    block3_bounds(cell_start, gp, cx, cy, cz, rs, re);
#pragma unroll
    for (int w = 0; w < kHistBins / 4; w++) hist[w * kNrmBlock] = 0u;
    // walk: what is found inside the radius is counted into the slices and noted down.  At most kInsideCap candidates are
    // counted and noted, so no byte counter overflows; a point with more inside the radius goes to the heap kernel below
    // (fb), and what it does until then it does on a consistent part of them (counting on while the notes stop would leave
    // list places unwritten: stale LDS as positions).
    int n_in = 0;
    walk_rows(pts, rs, re, [&](const RowBatch &b) {
#pragma unroll
      for (int u = 0; u < kHistLoads; u++) {
        const double ddx = q.x - b.x[u], ddy = q.y - b.y[u], ddz = q.z - b.z[u];
        double d2 = ddx * ddx;
        d2 = d2 + ddy * ddy;
        d2 = d2 + ddz * ddz;
        const bool in = b.j + u < b.e && d2 < thr;
        if (in && n_in < kInsideCap) {  // (what is counted is what is noted: the slices and the list places stay consistent)
          const float kf = (float)d2;
          const int bin = (int)(kf * scale);
          atomicAdd(&hist[(bin >> 2) * kNrmBlock], 1u << ((bin & 3) * 8));
          mine(n_in) = make_uint2(__float_as_uint(kf), b.j + u);
        }
        n_in += in ? 1 : 0;
      }
    });
    // KNN search (thr = the cell side squared, what the 3x3x3 block is guaranteed to cover): with fewer than k points that
    // near, the k-th neighbour lies outside the block
    bool fb = n_in >= kInsideCap || (knn_only && n_in < knn);
    const int k_eff = min(knn, n_in);
    // prefix: slice of the k-th nearest, number of candidates below it, first list place of every slice
    int bstar = kHistBins, c_lo = 0;
    {
      int cum = 0;
#pragma unroll
      for (int w = 0; w < kHistBins / 4; w++) {
        const uint32_t hw = hist[w * kNrmBlock];
        uint32_t ow = 0;
#pragma unroll
        for (int b4 = 0; b4 < 4; b4++) {
          const int c = (int)((hw >> (8 * b4)) & 255u);
          ow |= (uint32_t)min(cum, 255) << (8 * b4);
          if (bstar == kHistBins && cum + c >= k_eff && k_eff > 0) { bstar = 4 * w + b4; c_lo = cum; }
          cum += c;
        }
        offs[w * kNrmBlock] = ow;
      }
    }
    // placing: the notes are read back -- a lane's own stores, in order -- and go by their slice either to their place in
    // the list or to the boundary list
    int nb = 0;
    {
      const int n_read = min(n_in, kInsideCap);
      for (int i0 = 0; i0 < n_read; i0 += kHistLoads) {
        uint2 en[kHistLoads];
#pragma unroll
        for (int u = 0; u < kHistLoads; u++) en[u] = mine(min(i0 + u, kInsideCap - 1));
#pragma unroll
        for (int u = 0; u < kHistLoads; u++) {
          if (i0 + u < n_read) {
            const float kf = __uint_as_float(en[u].x);
            const int bin = (int)(kf * scale);
            if (bin < bstar) {
              const uint32_t old = atomicAdd(&offs[(bin >> 2) * kNrmBlock], 1u << ((bin & 3) * 8));
              const int dst = (int)((old >> ((bin & 3) * 8)) & 255u);
              if (dst < knn) { lkey[dst * kNrmBlock] = kf; lpos[dst * kNrmBlock] = en[u].y; }
            } else if (bin == bstar) {
              if (nb < kBoundaryCap) { bkey[nb * kNrmBlock] = kf; bpos[nb * kNrmBlock] = en[u].y; }
              nb++;
            }
          }
        }
      }
    }
    fb = fb || nb > kBoundaryCap;
    // boundary slice: sort (fixed 19-comparator network for 8), append the k_eff - c_lo nearest
    {
      float bk[kBoundaryCap];
      uint32_t bp[kBoundaryCap];
#pragma unroll
      for (int i = 0; i < kBoundaryCap; i++) {
        const bool have = i < nb;
        bk[i] = have ? bkey[i * kNrmBlock] : INFINITY;
        bp[i] = have ? bpos[i * kNrmBlock] : 0xFFFFFFFFu;
      }
      auto ce = [&](int a, int b) {
        const bool swap = bk[b] < bk[a] || (bk[b] == bk[a] && bp[a] != bp[b] && bk[a] != INFINITY && kl_less(l, bk[b], bp[b], bk[a], bp[a]));
        const float ka = swap ? bk[b] : bk[a], kb = swap ? bk[a] : bk[b];
        const uint32_t pa = swap ? bp[b] : bp[a], pb = swap ? bp[a] : bp[b];
        bk[a] = ka; bk[b] = kb; bp[a] = pa; bp[b] = pb;
      };
      ce(0, 1); ce(2, 3); ce(4, 5); ce(6, 7);
      ce(0, 2); ce(1, 3); ce(4, 6); ce(5, 7);
      ce(1, 2); ce(5, 6); ce(0, 4); ce(3, 7);
      ce(1, 5); ce(2, 6);
      ce(1, 4); ce(3, 6);
      ce(2, 4); ce(3, 5);
      ce(3, 4);
      const int need = k_eff - c_lo;
#pragma unroll
      for (int i = 0; i < kBoundaryCap; i++)
        if (i < need && c_lo + i < knn) { lkey[(c_lo + i) * kNrmBlock] = bk[i]; lpos[(c_lo + i) * kNrmBlock] = bp[i]; }
    }
    // order inside the slices below b*: odd-even transposition until nothing moves (wave-uniform loop)
    {
      bool moved = true;
      while (__any(moved)) {
        moved = false;
        for (int par = 0; par < 2; par++)
          for (int i = par; __any(i + 1 < c_lo); i += 2) {
            if (i + 1 < c_lo) {
              const float ka = lkey[i * kNrmBlock], kb = lkey[(i + 1) * kNrmBlock];
              if (kb <= ka) {
                uint32_t pa = lpos[i * kNrmBlock], pb = lpos[(i + 1) * kNrmBlock];
                if (kb < ka || kl_less(l, kb, pb, ka, pa)) {
                  lkey[i * kNrmBlock] = kb; lkey[(i + 1) * kNrmBlock] = ka;
                  lpos[i * kNrmBlock] = pb; lpos[(i + 1) * kNrmBlock] = pa;
                  moved = true;
                }
              }
            }
          }
      }
    }
    if (fb) {  // not representable here: the heap kernel takes this point
      fb_list[atomicAdd(fb_count, 1u)] = (uint32_t)p;
      continue;
    }
    const int k = k_eff;
    double c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i0 = 0; i0 < k; i0 += kHistLoads) {  // ComputeNormal :222-233, in the neighbour order FLANN returns
      double4 tq[kHistLoads];
#pragma unroll
      for (int u = 0; u < kHistLoads; u++) {
        uint32_t at = lpos[min(i0 + u, k - 1) * kNrmBlock];
        tq[u] = pts[at];
      }
#pragma unroll
      for (int u = 0; u < kHistLoads; u++) {
        if (i0 + u < k) {
          const double4 t = tq[u];
          c[0] += t.x; c[1] += t.y; c[2] += t.z;
          c[3] += t.x * t.x; c[4] += t.x * t.y; c[5] += t.x * t.z;
          c[6] += t.y * t.y; c[7] += t.y * t.z; c[8] += t.z * t.z;
        }
      }
    }
    double nv[3] = {0.0, 0.0, 1.0};
    if (cov_out && k < 3) cov_out[7 * oi] = 0.0;  // (k_normals_from_cov writes the (0, 0, 1))
    if (k >= 3) {
      for (int i = 0; i < 9; i++) c[i] /= (double)k;
      double cov[9];
      cov[0] = c[3] - c[0] * c[0];
      cov[4] = c[6] - c[1] * c[1];
      cov[8] = c[8] - c[2] * c[2];
      cov[1] = cov[3] = c[4] - c[0] * c[1];
      cov[2] = cov[6] = c[5] - c[0] * c[2];
      cov[5] = cov[7] = c[7] - c[1] * c[2];
      if (cov_out) {  // fast_normal_computation = false: the covariance is handed to k_normals_from_cov
        double *o = cov_out + 7 * oi;
        o[0] = 1.0; o[1] = cov[0]; o[2] = cov[1]; o[3] = cov[2]; o[4] = cov[4]; o[5] = cov[5]; o[6] = cov[8];
        continue;
      }
      fast_eigen3x3(cov, nv);
      if (sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]) == 0.0) {
        if (has_normals) { nv[0] = ref[3 * qi]; nv[1] = ref[3 * qi + 1]; nv[2] = ref[3 * qi + 2]; }
        else { nv[0] = 0.0; nv[1] = 0.0; nv[2] = 1.0; }
      }
      if (has_normals && nv[0] * ref[3 * qi] + nv[1] * ref[3 * qi + 1] + nv[2] * ref[3 * qi + 2] < 0.0) {
        nv[0] *= -1.0; nv[1] *= -1.0; nv[2] *= -1.0;
      }
    }
    if (cov_out) continue;
    normals[3 * oi] = nv[0]; normals[3 * oi + 1] = nv[1]; normals[3 * oi + 2] = nv[2];
  }
}

// fast_normal_computation = false: the search kernels above leave, per output row, a flag (3 or more neighbours) and the six
// covariance entries; this finishes the point as they would have (EstimateNormals.cpp:336-349) with Eigen's iterative solver.
__global__ __launch_bounds__(O3D_BLOCK) void k_normals_from_cov(const double4 *__restrict__ pts, int64_t p_lo, int64_t p_hi,
                                                                 const double *__restrict__ cov7, double *normals, int has_normals,
                                                                 int by_pos, const double *ref) {
  for (int64_t p = p_lo + (int64_t)blockIdx.x * O3D_BLOCK + threadIdx.x; p < p_hi; p += (int64_t)gridDim.x * O3D_BLOCK) {
    const size_t qi = (size_t)__double_as_longlong(pts[p].w);
    const size_t oi = by_pos ? (size_t)(p - p_lo) : qi;
    const double *c = cov7 + 7 * oi;
    double nv[3] = {0.0, 0.0, 1.0};
    if (c[0] != 0.0) {
      const double cov[9] = {c[1], c[2], c[3], c[2], c[4], c[5], c[3], c[5], c[6]};
      self_adjoint_eigen3_col0(cov, nv);
      if (sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]) == 0.0) {
        if (has_normals) { nv[0] = ref[3 * qi]; nv[1] = ref[3 * qi + 1]; nv[2] = ref[3 * qi + 2]; }
        else { nv[0] = 0.0; nv[1] = 0.0; nv[2] = 1.0; }
      }
      if (has_normals && nv[0] * ref[3 * qi] + nv[1] * ref[3 * qi + 1] + nv[2] * ref[3 * qi + 2] < 0.0) {
        nv[0] *= -1.0; nv[1] *= -1.0; nv[2] *= -1.0;
      }
    }
    normals[3 * oi] = nv[0]; normals[3 * oi + 1] = nv[1]; normals[3 * oi + 2] = nv[2];
  }
}

// Shard outputs (cell order, rank-major) -> the caller's order.
__global__ __launch_bounds__(O3D_BLOCK) void k_normals_scatter(const double4 *__restrict__ pts, int64_t n_grid,
                                                                const double *__restrict__ parts, double *__restrict__ normals) {
  for (int64_t p = (int64_t)blockIdx.x * O3D_BLOCK + threadIdx.x; p < n_grid; p += (int64_t)gridDim.x * O3D_BLOCK) {
    const size_t qi = (size_t)__double_as_longlong(pts[p].w);
    normals[3 * qi] = parts[3 * p]; normals[3 * qi + 1] = parts[3 * p + 1]; normals[3 * qi + 2] = parts[3 * p + 2];
  }
}

// 3 x 3 symmetric solve with Eigen's LDLT (pivot on the largest |diagonal|, zero pivots give a zero component):
// SolveLinearSystemPSD without checks, utility/Eigen.cpp:85.
__device__ void ldlt_solve3(const double *A, const double *b, double *x) {
  double M[9];
  int tr[3];
  for (int i = 0; i < 9; i++) M[i] = A[i];
#define MM(i, j) M[(i) * 3 + (j)]
  for (int k = 0; k < 3; k++) {
    int p = k;
    double big = fabs(MM(k, k));
    for (int i = k + 1; i < 3; i++)
      if (fabs(MM(i, i)) > big) { big = fabs(MM(i, i)); p = i; }
    tr[k] = p;
    if (p != k) {
      for (int j = 0; j < k; j++) { const double t = MM(k, j); MM(k, j) = MM(p, j); MM(p, j) = t; }
      for (int i = p + 1; i < 3; i++) { const double t = MM(i, k); MM(i, k) = MM(i, p); MM(i, p) = t; }
      for (int i = k + 1; i < p; i++) { const double t = MM(i, k); MM(i, k) = MM(p, i); MM(p, i) = t; }
      { const double t = MM(k, k); MM(k, k) = MM(p, p); MM(p, p) = t; }
    }
    if (k > 0) {
      double temp[3];
      for (int j = 0; j < k; j++) temp[j] = MM(j, j) * MM(k, j);
      double s = 0;
      for (int j = 0; j < k; j++) s += MM(k, j) * temp[j];
      MM(k, k) -= s;
      for (int i = k + 1; i < 3; i++) {
        double a = 0;
        for (int j = 0; j < k; j++) a += MM(i, j) * temp[j];
        MM(i, k) -= a;
      }
    }
    const double akk = MM(k, k);
    if (fabs(akk) > 0)
      for (int i = k + 1; i < 3; i++) MM(i, k) /= akk;
  }
  double y[3];
  for (int i = 0; i < 3; i++) y[i] = b[i];
  for (int k = 0; k < 3; k++)
    if (tr[k] != k) { const double t = y[k]; y[k] = y[tr[k]]; y[tr[k]] = t; }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < i; j++) y[i] -= MM(i, j) * y[j];
  for (int i = 0; i < 3; i++) {
    if (fabs(MM(i, i)) > 2.2250738585072014e-308) y[i] /= MM(i, i);
    else y[i] = 0;
  }
  for (int i = 2; i >= 0; i--)
    for (int j = i + 1; j < 3; j++) y[i] -= MM(j, i) * y[j];
  for (int k = 2; k >= 0; k--)
    if (tr[k] != k) { const double t = y[k]; y[k] = y[tr[k]]; y[tr[k]] = t; }
  for (int i = 0; i < 3; i++) x[i] = y[i];
#undef MM
}

// InitializePointCloudForColoredICP, ColoredICP.cpp:68-123: per point the hybrid neighbourhood (the max_nn nearest with
// d2 < thr, ascending by (d2, index) -- the query itself comes first), the least-squares colour gradient in the tangent
// plane with the orthogonality row (:108-112), 3 x 3 normal equations solved by LDLT.  Fewer than 4 neighbours: zero.
// normals / colours / grad are in the caller's order (3 doubles per point).
__global__ __launch_bounds__(kNrmBlock) void k_color_gradient(const double4 *__restrict__ pts, int64_t n_grid,
                                                               const uint32_t *__restrict__ cell_start, GridParams gp, int knn, double thr,
                                                               const double *__restrict__ normals, const double *__restrict__ colors,
                                                               double *__restrict__ grad) {
  extern __shared__ unsigned char smem[];
  const int lane = threadIdx.x;
  KList l;
  klist_bind(l, smem, lane, knn, pts);
  l.thr = thr;
  for (int64_t p = (int64_t)blockIdx.x * kNrmBlock + lane; p < n_grid; p += (int64_t)gridDim.x * kNrmBlock) {
    const double4 q = pts[p];
    const size_t qi = (size_t)__double_as_longlong(q.w);
    const int cx = min(max(o3d_cell_coord(q.x, gp.gmin[0], gp.inv_h, gp.nx), 0), gp.nx - 1);
    const int cy = min(max(o3d_cell_coord(q.y, gp.gmin[1], gp.inv_h, gp.ny), 0), gp.ny - 1);
    const int cz = min(max(o3d_cell_coord(q.z, gp.gmin[2], gp.inv_h, gp.nz), 0), gp.nz - 1);
    klist_query(l, q.x, q.y, q.z);
    walk_block3(pts, cell_start, gp, cx, cy, cz, (uint32_t)p, [&](const double4 &t, uint32_t j, bool real) {
      const double ddx = q.x - t.x, ddy = q.y - t.y, ddz = q.z - t.z;
      double d2 = ddx * ddx;
      d2 = d2 + ddy * ddy;
      d2 = d2 + ddz * ddz;
      if (real && klist_may_take(l, d2)) klist_add(l, d2, j);
    });
    double g[3] = {0.0, 0.0, 0.0};
    klist_sort(l);
    const int nn = l.count;
    if (nn >= 4) {
      const double n0 = normals[3 * qi], n1 = normals[3 * qi + 1], n2 = normals[3 * qi + 2];
      const double it = (colors[3 * qi] + colors[3 * qi + 1] + colors[3 * qi + 2]) / 3.0;
      double AtA[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, Atb[3] = {0, 0, 0};
      for (int i = 1; i <= nn; i++) {
        double a[3], b;
        if (i < nn) {
          const double4 t = pts[l.pos[i * kNrmBlock]];
          const size_t ai = (size_t)__double_as_longlong(t.w);
          const double dot = ((t.x - q.x) * n0 + (t.y - q.y) * n1) + (t.z - q.z) * n2;
          a[0] = (t.x - dot * n0) - q.x; a[1] = (t.y - dot * n1) - q.y; a[2] = (t.z - dot * n2) - q.z;
          b = (colors[3 * ai] + colors[3 * ai + 1] + colors[3 * ai + 2]) / 3.0 - it;
        } else {
          a[0] = (double)(nn - 1) * n0; a[1] = (double)(nn - 1) * n1; a[2] = (double)(nn - 1) * n2;
          b = 0.0;
        }
        for (int r = 0; r < 3; r++) {
          for (int c = 0; c < 3; c++) AtA[r * 3 + c] += a[r] * a[c];
          Atb[r] += a[r] * b;
        }
      }
      ldlt_solve3(AtA, Atb, g);
    }
    grad[3 * qi] = g[0]; grad[3 * qi + 1] = g[1]; grad[3 * qi + 2] = g[2];
  }
}

// The other KDTreeFlann consumers on point clouds (geometry/PointCloud.cpp:112-132, 394-468, 521-544): per query point
//   MODE 0  sqrt of the knn-th smallest d2 over the grid's cloud, 0 when fewer exist (ComputePointCloudDistance: knn 1 on
//           another cloud; ComputeNearestNeighborDistance: knn 2 on the cloud itself)
//   MODE 1  mean of sqrt(d2) over the knn nearest in ascending order, -1 when there is none (RemoveStatisticalOutliers)
//   MODE 2  number of points with d2 < thr (RemoveRadiusOutliers; the grid's cell size is the radius)
// Queries are 3 doubles per point in the caller's order; out[i] belongs to query i.
template <int MODE>
__global__ __launch_bounds__(kNrmBlock) void k_neighbor_query(const double4 *__restrict__ pts, const uint32_t *__restrict__ cell_start,
                                                               GridParams gp, const double *__restrict__ q_xyz, int64_t n_q, int knn,
                                                               double thr, double *__restrict__ out, int32_t *__restrict__ out_count) {
  extern __shared__ unsigned char smem[];
  const int lane = threadIdx.x;
  KList l;
  klist_bind(l, smem, lane, knn, pts);
  l.thr = INFINITY;
  const int max_dim = max(max(gp.nx, gp.ny), gp.nz);
  for (int64_t i = (int64_t)blockIdx.x * kNrmBlock + lane; i < n_q; i += (int64_t)gridDim.x * kNrmBlock) {
    const double qx = q_xyz[3 * i], qy = q_xyz[3 * i + 1], qz = q_xyz[3 * i + 2];
    // a query outside the grid searches from the nearest cell: every point beyond shell R of that cell is still at
    // least R cells away from the query, which lies on the far side of the cell
    const int cx = min(max(o3d_cell_coord(qx, gp.gmin[0], gp.inv_h, gp.nx), 0), gp.nx - 1);
    const int cy = min(max(o3d_cell_coord(qy, gp.gmin[1], gp.inv_h, gp.ny), 0), gp.ny - 1);
    const int cz = min(max(o3d_cell_coord(qz, gp.gmin[2], gp.inv_h, gp.nz), 0), gp.nz - 1);
    klist_query(l, qx, qy, qz);
    int count = 0;
    auto visit = [&](const double4 &t, uint32_t j, bool real) {
      const double ddx = qx - t.x, ddy = qy - t.y, ddz = qz - t.z;
      double d2 = ddx * ddx;
      d2 = d2 + ddy * ddy;
      d2 = d2 + ddz * ddz;
      if (!real) return;
      if (MODE == 2) {
        if (d2 < thr) count++;
      } else if (klist_may_take(l, d2)) {
        klist_add(l, d2, j);
      }
    };
    for (int R = 1;; R++) {
      if (R == 1) walk_block3(pts, cell_start, gp, cx, cy, cz, 0u, visit);
      else walk_shell(pts, cell_start, gp, cx, cy, cz, R, visit);
      if (MODE == 2) break;  // h >= radius: the 3x3x3 block is complete
      if (R >= max_dim) break;
      if (l.count == l.cap) {
        const double reach = (double)R * gp.h * (1.0 - 1e-6);
        if (klist_kth_upper(l) <= reach * reach) break;
      }
    }
    if (MODE == 0) {  // the k-th distance is the worst kept entry: the heap's root
      out[i] = (l.count >= knn) ? sqrt(kl_exact(l, l.pos[0])) : 0.0;
    } else if (MODE == 1) {
      double mean = -1.0;
      if (l.count > 0) {
        klist_sort(l);
        double sum = 0.0;  // std::accumulate over the ascending distances (PointCloud.cpp:445-446)
        for (int k = 0; k < l.count; k++) sum += sqrt(kl_exact(l, l.pos[k * kNrmBlock]));
        mean = sum / (double)l.count;
      }
      out[i] = mean;
    } else {
      out_count[i] = count;
    }
  }
}

// ---- FPFH, pipelines/registration/Feature.cpp:39-159 -------------------------------------------------------------------

// The neighbour list of one query (KDTreeFlann::Search with a KNN or hybrid parameter): the l.cap nearest with d2 < l.thr,
// ascending by (d2, index).  KNN (l.thr = inf) grows the searched cube shell by shell; hybrid needs the 3x3x3 block only
// (cell size >= radius).
__device__ __forceinline__ void collect_neighbors(KList &l, const double4 *__restrict__ pts, const uint32_t *__restrict__ cell_start,
                                                  const GridParams &gp, double qx, double qy, double qz, bool knn_mode) {
  const int cx = min(max(o3d_cell_coord(qx, gp.gmin[0], gp.inv_h, gp.nx), 0), gp.nx - 1);
  const int cy = min(max(o3d_cell_coord(qy, gp.gmin[1], gp.inv_h, gp.ny), 0), gp.ny - 1);
  const int cz = min(max(o3d_cell_coord(qz, gp.gmin[2], gp.inv_h, gp.nz), 0), gp.nz - 1);
  const int max_dim = max(max(gp.nx, gp.ny), gp.nz);
  klist_query(l, qx, qy, qz);
  auto visit = [&](const double4 &t, uint32_t j, bool real) {
    const double ddx = qx - t.x, ddy = qy - t.y, ddz = qz - t.z;
    double d2 = ddx * ddx;
    d2 = d2 + ddy * ddy;
    d2 = d2 + ddz * ddz;
    if (real && klist_may_take(l, d2)) klist_add(l, d2, j);
  };
  for (int R = 1;; R++) {
    if (R == 1) walk_block3(pts, cell_start, gp, cx, cy, cz, 0u, visit);
    else walk_shell(pts, cell_start, gp, cx, cy, cz, R, visit);
    if (!knn_mode) break;
    if (R >= max_dim) break;
    if (l.count == l.cap) {
      const double reach = (double)R * gp.h * (1.0 - 1e-6);
      if (klist_kth_upper(l) <= reach * reach) break;
    }
  }
  klist_sort(l);
}

// ComputePairFeatures, Feature.cpp:39-69
__device__ void pair_features(const double *p1, const double *n1, const double *p2, const double *n2, double *f) {
  double d[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  f[0] = f[1] = f[2] = f[3] = 0.0;
  const double len = sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
  if (len == 0.0) return;
  double a[3] = {n1[0], n1[1], n1[2]}, b[3] = {n2[0], n2[1], n2[2]};
  const double angle1 = ((a[0] * d[0] + a[1] * d[1]) + a[2] * d[2]) / len;
  const double angle2 = ((b[0] * d[0] + b[1] * d[1]) + b[2] * d[2]) / len;
  double f2;
  if (acos(fabs(angle1)) > acos(fabs(angle2))) {
    for (int k = 0; k < 3; k++) { a[k] = n2[k]; b[k] = n1[k]; d[k] *= -1.0; }
    f2 = -angle2;
  } else {
    f2 = angle1;
  }
  double v[3];
  cross3(d, a, v);
  const double vn = sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
  if (vn == 0.0) return;
  for (int k = 0; k < 3; k++) v[k] /= vn;
  double w[3];
  cross3(a, v, w);
  f[3] = len;
  f[2] = f2;
  f[1] = (v[0] * b[0] + v[1] * b[1]) + v[2] * b[2];
  f[0] = atan2((w[0] * b[0] + w[1] * b[1]) + w[2] * b[2], (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]);
}

__device__ __forceinline__ int fpfh_bin(double v) {
  int h = (int)floor(v);
  return h < 0 ? 0 : (h >= 11 ? 10 : h);
}

// PASS 0: ComputeSPFHFeature (Feature.cpp:71-111) -> spfh[33 * i + j].  PASS 1: the weighting of ComputeFPFHFeature
// (:127-157) -> out[33 * i + j], reading spfh.  The reference searches the neighbours again in the second pass; here pass 0
// leaves them behind (index and distance of neighbour k of the point at position p at [k * n_grid + p], the count at
// nbr_cnt[p]) when the host found room for that table (nbr_cnt != nullptr), and pass 1 then needs neither a search nor LDS.
// Features are 33 doubles per point (Feature::data_ is a column-major 33 x n Eigen matrix), in the caller's point order.
template <int PASS>
__global__ __launch_bounds__(kNrmBlock) void k_fpfh(const double4 *__restrict__ pts, int64_t n_grid, const uint32_t *__restrict__ cell_start,
                                                     GridParams gp, int knn, double thr, int knn_mode, const double *__restrict__ normals,
                                                     const double *__restrict__ spfh, double *__restrict__ out,
                                                     uint32_t *__restrict__ nbr_idx, double *__restrict__ nbr_d2,
                                                     int32_t *__restrict__ nbr_cnt) {
  extern __shared__ unsigned char smem[];
  const int lane = threadIdx.x;
  const bool cached = nbr_cnt != nullptr;
  KList l;
  klist_bind(l, smem, lane, knn, pts);
  l.thr = knn_mode ? INFINITY : thr;
  l.count = 0;
  for (int64_t p = (int64_t)blockIdx.x * kNrmBlock + lane; p < n_grid; p += (int64_t)gridDim.x * kNrmBlock) {
    const double4 q = pts[p];
    const size_t qi = (size_t)__double_as_longlong(q.w);
    if (PASS == 0 || !cached) collect_neighbors(l, pts, cell_start, gp, q.x, q.y, q.z, knn_mode != 0);
    double *o = out + 33 * qi;
    if (PASS == 0) {
      if (cached) nbr_cnt[p] = l.count;
      // the row was zeroed by the host; one lane owns it
      if (l.count > 1) {
        const double hist_incr = 100.0 / (double)(l.count - 1);
        const double p1[3] = {q.x, q.y, q.z};
        const double n1[3] = {normals[3 * qi], normals[3 * qi + 1], normals[3 * qi + 2]};
        for (int k = 1; k < l.count; k++) {
          const double4 t = pts[l.pos[k * kNrmBlock]];
          const size_t ti = (size_t)__double_as_longlong(t.w);
          if (cached) {
            nbr_idx[(size_t)k * (size_t)n_grid + (size_t)p] = (uint32_t)ti;
            nbr_d2[(size_t)k * (size_t)n_grid + (size_t)p] = kl_exact(l, l.pos[k * kNrmBlock]);
          }
          const double p2[3] = {t.x, t.y, t.z};
          const double n2[3] = {normals[3 * ti], normals[3 * ti + 1], normals[3 * ti + 2]};
          double f[4];
          pair_features(p1, n1, p2, n2, f);
          o[fpfh_bin(11 * (f[0] + M_PI) / (2.0 * M_PI))] += hist_incr;
          o[fpfh_bin(11 * (f[1] + 1.0) * 0.5) + 11] += hist_incr;
          o[fpfh_bin(11 * (f[2] + 1.0) * 0.5) + 22] += hist_incr;
        }
      }
    } else {
      const int cnt = cached ? nbr_cnt[p] : l.count;
      if (cnt > 1) {
        double acc[33], sum[3] = {0.0, 0.0, 0.0};  // (the row was zeroed by the host: the same additions, in registers)
#pragma unroll
        for (int j = 0; j < 33; j++) acc[j] = 0.0;
        for (int k = 1; k < cnt; k++) {
          double dist;
          size_t ni;
          if (cached) {
            dist = nbr_d2[(size_t)k * (size_t)n_grid + (size_t)p];
            ni = nbr_idx[(size_t)k * (size_t)n_grid + (size_t)p];
          } else {
            dist = kl_exact(l, l.pos[k * kNrmBlock]);
            ni = kl_index(l, l.pos[k * kNrmBlock]);
          }
          if (dist == 0.0) continue;
          const double *sp = spfh + 33 * ni;
#pragma unroll
          for (int j = 0; j < 33; j++) {
            const double val = sp[j] / dist;
            sum[j / 11] += val;
            acc[j] += val;
          }
        }
        for (int j = 0; j < 3; j++)
          if (sum[j] != 0.0) sum[j] = 100.0 / sum[j];
        const double *self = spfh + 33 * qi;
#pragma unroll
        for (int j = 0; j < 33; j++) {
          double v = acc[j] * sum[j / 11];
          v += self[j];
          o[j] = v;
        }
      }
    }
  }
}

// ComputeFPFHFeature with a radius search parameter (KDTreeSearchParamRadius: EVERY point with d2 < thr, Feature.cpp:83,131):
// no list -- the 3x3x3 block is walked once per pass and every neighbour is used as it comes.
//   PASS 0: the neighbours' three bins are COUNTED (per-lane counters in LDS, [bin][lane]); a histogram entry of the
//           reference is hist_incr added once per neighbour of the bin, whatever their order, so it is rebuilt by that many
//           additions afterwards.  The reference leaves out element 0 of the sorted list: the query itself, or a duplicate of
//           it with a lower index -- the same thing to ComputePairFeatures (distance 0 -> all zero) -- so the query's own
//           position is the one left out here.  nbr_cnt[p] keeps the number of neighbours.
//   PASS 1: the weighted sum over the neighbours in the order of the walk (the reference adds them by ascending distance:
//           the same terms, rounding-level difference in the sum).
template <int PASS>
__global__ __launch_bounds__(kNrmBlock) void k_fpfh_radius(const double4 *__restrict__ pts, int64_t n_grid,
                                                            const uint32_t *__restrict__ cell_start, GridParams gp, double thr,
                                                            const double *__restrict__ normals, const double *__restrict__ spfh,
                                                            double *__restrict__ out) {
  __shared__ uint32_t bins[33][kNrmBlock];
  const int lane = threadIdx.x;
  for (int64_t p = (int64_t)blockIdx.x * kNrmBlock + lane; p < n_grid; p += (int64_t)gridDim.x * kNrmBlock) {
    const double4 q = pts[p];
    const size_t qi = (size_t)__double_as_longlong(q.w);
    const int cx = min(max(o3d_cell_coord(q.x, gp.gmin[0], gp.inv_h, gp.nx), 0), gp.nx - 1);
    const int cy = min(max(o3d_cell_coord(q.y, gp.gmin[1], gp.inv_h, gp.ny), 0), gp.ny - 1);
    const int cz = min(max(o3d_cell_coord(q.z, gp.gmin[2], gp.inv_h, gp.nz), 0), gp.nz - 1);
    double *o = out + 33 * qi;
    if (PASS == 0) {
#pragma unroll
      for (int b = 0; b < 33; b++) bins[b][lane] = 0u;
      const double p1[3] = {q.x, q.y, q.z};
      const double n1[3] = {normals[3 * qi], normals[3 * qi + 1], normals[3 * qi + 2]};
      uint32_t others = 0;
      auto visit = [&](const double4 &t, uint32_t j, bool real) {
        const double ddx = q.x - t.x, ddy = q.y - t.y, ddz = q.z - t.z;
        double d2 = ddx * ddx;
        d2 = d2 + ddy * ddy;
        d2 = d2 + ddz * ddz;
        if (!real || j == (uint32_t)p || !(d2 < thr)) return;
        others++;
        const size_t ti = (size_t)__double_as_longlong(t.w);
        const double p2[3] = {t.x, t.y, t.z};
        const double n2[3] = {normals[3 * ti], normals[3 * ti + 1], normals[3 * ti + 2]};
        double f[4];
        pair_features(p1, n1, p2, n2, f);
        bins[fpfh_bin(11 * (f[0] + M_PI) / (2.0 * M_PI))][lane]++;
        bins[fpfh_bin(11 * (f[1] + 1.0) * 0.5) + 11][lane]++;
        bins[fpfh_bin(11 * (f[2] + 1.0) * 0.5) + 22][lane]++;
      };
      walk_block3(pts, cell_start, gp, cx, cy, cz, (uint32_t)p, visit);
      if (others > 0) {  // (the row was zeroed by the host)
        const double hist_incr = 100.0 / (double)others;
        for (int b = 0; b < 33; b++) {
          double v = 0.0;
          for (uint32_t c = bins[b][lane]; c > 0; c--) v += hist_incr;
          o[b] = v;
        }
      }
    } else {
      double acc[33], sum[3] = {0.0, 0.0, 0.0};
#pragma unroll
      for (int j = 0; j < 33; j++) acc[j] = 0.0;
      uint32_t others = 0;
      auto visit = [&](const double4 &t, uint32_t j, bool real) {
        const double ddx = q.x - t.x, ddy = q.y - t.y, ddz = q.z - t.z;
        double d2 = ddx * ddx;
        d2 = d2 + ddy * ddy;
        d2 = d2 + ddz * ddz;
        if (!real || j == (uint32_t)p || !(d2 < thr)) return;
        others++;
        if (d2 == 0.0) return;
        const double *sp = spfh + 33 * (size_t)__double_as_longlong(t.w);
#pragma unroll
        for (int b = 0; b < 33; b++) {
          const double val = sp[b] / d2;
          sum[b / 11] += val;
          acc[b] += val;
        }
      };
      walk_block3(pts, cell_start, gp, cx, cy, cz, (uint32_t)p, visit);
      if (others > 0) {
        for (int j = 0; j < 3; j++)
          if (sum[j] != 0.0) sum[j] = 100.0 / sum[j];
        const double *self = spfh + 33 * qi;
#pragma unroll
        for (int j = 0; j < 33; j++) {
          double v = acc[j] * sum[j / 11];
          v += self[j];
          o[j] = v;
        }
      }
    }
  }
}

// Non-finite points are not in the grid: FLANN finds no finite-distance neighbour for them -> fewer than 3 -> (0,0,1).
__global__ __launch_bounds__(O3D_BLOCK) void k_normals_nonfinite(const double *__restrict__ xyz, int64_t n, double *__restrict__ normals) {
  for (int64_t i = (int64_t)blockIdx.x * O3D_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * O3D_BLOCK) {
    const double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    if (!(isfinite(x) && isfinite(y) && isfinite(z))) { normals[3 * i] = 0.0; normals[3 * i + 1] = 0.0; normals[3 * i + 2] = 1.0; }
  }
}

__global__ __launch_bounds__(O3D_BLOCK) void k_count_heads(const uint32_t *__restrict__ keys, int64_t n, uint32_t sentinel,
                                                            unsigned long long *__restrict__ out) {
  unsigned long long c = 0;
  for (int64_t p = (int64_t)blockIdx.x * O3D_BLOCK + threadIdx.x; p < n; p += (int64_t)gridDim.x * O3D_BLOCK)
    if (keys[p] != sentinel && (p == 0 || keys[p] != keys[p - 1])) c++;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, O3D_WAVE);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

}  // namespace

// keys of the last o3d_build_target_grid call stay sorted in scratch slot 1: number of occupied cells
static int count_occupied_cells(o3d_ctx *ctx, int64_t n, int64_t *out) {
  void *cnt;
  O3D_TRY(o3d_scratch(ctx, 7, 256, &cnt));
  O3D_HIP(ctx, hipMemsetAsync(cnt, 0, 8, ctx->stream));
  hipLaunchKernelGGL(k_count_heads, dim3(grid_for(n)), dim3(O3D_BLOCK), 0, ctx->stream, (const uint32_t *)ctx->scratch[1], n,
                     ctx->gp.ncell, (unsigned long long *)cnt);
  unsigned long long *h = (unsigned long long *)ctx->pinned;
  O3D_HIP(ctx, o3d_d2h(ctx, h, cnt, 8));
  O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *out = (int64_t)*h;
  return O3D_OK;
}

// Grid for k-nearest-neighbour queries over a cloud: cell size from the cloud's density, aiming at ~8 points per
// occupied cell (at most four trial builds).  Leaves the grid built; *out_h is the cell size asked for last.
static int build_knn_grid(o3d_ctx *ctx, const double *d_xyz, int64_t n, double *out_h, double want = 8.0, double at_least = 4.0) {
  double bb[6];
  int64_t nf = 0;
  O3D_TRY(o3d_bounds(ctx, d_xyz, n, bb, &nf));
  const double ext = std::max(std::max(bb[3] - bb[0], bb[4] - bb[1]), std::max(bb[5] - bb[2], 1e-300));
  double h = ext * sqrt(want / (double)std::max<int64_t>(nf, 1));
  if (!(h > 0.0)) h = 1.0;
  for (int trial = 0; trial < 4 && nf > 0; trial++) {
    O3D_TRY(o3d_build_target_grid(ctx, d_xyz, nullptr, n, h));
    ctx->knn_grid_h = h;  // what the grid in the context was built for
    int64_t occ = 0;
    O3D_TRY(count_occupied_cells(ctx, n, &occ));
    const double pop = (double)nf / (double)std::max<int64_t>(occ, 1);
    if (pop >= at_least && pop <= 2.0 * want) break;
    if (ctx->gp.h > h * 1.01 && pop > 2.0 * want) break;  // the dense-table budget already coarsened the grid
    h = ctx->gp.h / (1.0 + 1e-5) * pow(want / pop, 1.0 / 2.5);
  }
  *out_h = h;
  return O3D_OK;
}

// shard < 0: the whole cloud, normals written in place (the drop-in call).  shard >= 0 of n_shards: positions
// [shard * rows, min(ng, (shard + 1) * rows)) of the cell-ordered cloud, rows = ceil(ng / n_shards), written in that order to
// out_part (rows x 3, `mem` space); `normals` is then only the orientation reference (has_normals).  *out_ng = ng.
static int estimate_normals_run(o3d_ctx *ctx, const double *xyz, int64_t n, int search_type, int knn, double radius, int mem,
                                double *normals, int has_normals, int shard, int n_shards, double *out_part, int64_t *out_ng) {
  if (n < 0 || (n > 0 && (!xyz || (!normals && (shard < 0 || has_normals)))))
    return o3d_fail(ctx, O3D_ERR_INVALID_ARG, "EstimateNormals: null buffer");
  if (search_type < 0 || search_type > 2) return o3d_fail(ctx, O3D_ERR_INVALID_ARG, "EstimateNormals: unknown search type %d", search_type);
  if (n >= ((int64_t)1 << 31)) return o3d_fail(ctx, O3D_ERR_UNSUPPORTED, "EstimateNormals: more than 2^31-1 points");
  if (shard >= 0 && (n_shards < 1 || shard >= n_shards || !out_part || !out_ng))
    return o3d_fail(ctx, O3D_ERR_INVALID_ARG, "EstimateNormals: bad shard arguments");
  if (out_ng) *out_ng = 0;
  if (n == 0) return O3D_OK;
  ctx->icp_ready = 0;  // the context's target grid is rebuilt for this cloud
  const bool sharded = shard >= 0;
  const size_t bytes = sizeof(double) * 3 * (size_t)n;
  const double *d_xyz = xyz;
  double *d_nrm = normals;
  void *sx = nullptr, *sn = nullptr, *sp = nullptr, *sf = nullptr, *sc = nullptr, *si = nullptr;
  auto cleanup = [&]() {
    for (void *b : {sx, sn, sp, sf, sc, si})
      if (b) o3d_tmp_free(ctx, b);
  };
  auto run = [&]() -> int {
    if (mem == O3D_MEM_HOST) {
      O3D_HIP(ctx, o3d_tmp_alloc(ctx, &sx, bytes));
      O3D_HIP(ctx, o3d_h2d(ctx, sx, xyz, bytes));
      d_xyz = (const double *)sx;
      if (!sharded || has_normals) {
        O3D_HIP(ctx, o3d_tmp_alloc(ctx, &sn, bytes));
        if (has_normals) O3D_HIP(ctx, o3d_h2d(ctx, sn, normals, bytes));
        d_nrm = (double *)sn;
      }
    }
    const bool degenerate = (search_type != O3D_SEARCH_RADIUS && knn < 3) || (search_type != O3D_SEARCH_KNN && !(radius > 0.0));
    if (search_type != O3D_SEARCH_RADIUS && knn > kMaxKnn)
      return o3d_fail(ctx, O3D_ERR_UNSUPPORTED, "EstimateNormals: knn/max_nn > %d is not on the device path", kMaxKnn);
    double h;
    // KNN by counting (k_estimate_normals_hist): the 3x3x3 block is guaranteed to hold everything nearer than one cell side,
    // so the cells are sized for the k nearest of a typical point to lie that near -- on a surface a disc of radius h holds
    // pi times a cell's population -- and the points whose k-th neighbour is farther go to the heap kernel's shell walk
    ctx->knn_grid_h = -1.0;
    const bool knn_hist = search_type == O3D_SEARCH_KNN && knn >= 3 && knn <= kHistMaxKnn && !getenv("O3D_NORMALS_HEAP");
    if (knn_hist) {
      O3D_TRY(build_knn_grid(ctx, d_xyz, n, &h, std::max(8.0, 0.6 * knn), std::max(4.0, 0.45 * knn)));
    } else if (search_type == O3D_SEARCH_KNN) {
      O3D_TRY(build_knn_grid(ctx, d_xyz, n, &h));
    } else {
      h = radius;
    }
    // degenerate: knn < 3 or an empty radius -- no point can collect 3 neighbours, every normal becomes (0, 0, 1)
    // (:347-349); run the hybrid kernel with an empty acceptance threshold
    // (radius and hybrid searches read the 3x3x3 block only: they can use the bricked table of a sparse scene)
    if (!(search_type == O3D_SEARCH_KNN && !degenerate && ctx->knn_grid_h == h && ctx->n_tgt == n))  // (the trial build already is it)
      O3D_TRY(o3d_build_target_grid(ctx, d_xyz, nullptr, n, degenerate ? 1.0 : h, search_type != O3D_SEARCH_KNN && !degenerate));
    const int64_t ng = ctx->n_tgt_grid;
    if (out_ng) *out_ng = ng;
    int64_t p_lo = 0, p_hi = ng, rows = 0;
    double *d_out = d_nrm;
    if (sharded) {
      rows = (ng + n_shards - 1) / n_shards;
      p_lo = std::min<int64_t>(ng, (int64_t)shard * rows);
      p_hi = std::min<int64_t>(ng, p_lo + rows);
      d_out = out_part;
      if (mem == O3D_MEM_HOST) {
        O3D_HIP(ctx, o3d_tmp_alloc(ctx, &sp, sizeof(double) * 3 * (size_t)std::max<int64_t>(rows, 1)));
        d_out = (double *)sp;
      }
      if (rows > 0) O3D_HIP(ctx, hipMemsetAsync(d_out, 0, sizeof(double) * 3 * (size_t)rows, ctx->stream));  // padding rows
    }
    const double thr = (double)(float)(radius * radius);  // KDTreeFlann.cpp:126,148
    const int kk = degenerate ? 1 : (search_type == O3D_SEARCH_RADIUS ? 1 : knn);
    const size_t lds = kListEntryBytes * kNrmBlock * (size_t)kk;
    int64_t g = (p_hi - p_lo + kNrmBlock - 1) / kNrmBlock;
    g = std::max<int64_t>(1, std::min<int64_t>(g, O3D_NUM_CU * 24));
    const int by_pos = sharded ? 1 : 0;
    const double *ref = d_nrm;
    double *d_cov = nullptr;  // fast_normal_computation = false: covariances per output row for k_normals_from_cov
    if (!ctx->normals_fast && p_hi > p_lo) {
      const size_t out_rows = sharded ? (size_t)rows : (size_t)n;
      O3D_HIP(ctx, o3d_tmp_alloc(ctx, &sc, sizeof(double) * 7 * std::max<size_t>(out_rows, 1)));
      d_cov = (double *)sc;
    }
    if (p_hi > p_lo) {
      if (degenerate) {
        O3D_TRY(allow_lds(ctx, k_estimate_normals<2>, lds));
        hipLaunchKernelGGL(k_estimate_normals<2>, dim3((int)g), dim3(kNrmBlock), lds, ctx->stream, ctx->tgt_sorted, p_lo, p_hi,
                           ctx->cell_start, ctx->gp, 1, 0.0, d_out, has_normals, by_pos, ref, nullptr, nullptr, d_cov);
      } else if (knn_hist && ctx->gp.h > 1e-15) {  // (the slice number is unclamped: keep key * scale finite)
        O3D_HIP(ctx, o3d_tmp_alloc(ctx, &sf, sizeof(uint32_t) * ((size_t)(p_hi - p_lo) + 64)));
        uint32_t *fb_list = (uint32_t *)sf + 64, *fb_count = (uint32_t *)sf;
        O3D_HIP(ctx, hipMemsetAsync(fb_count, 0, sizeof(uint32_t), ctx->stream));
        const double reach = ctx->gp.h / (1.0 + 1e-5);  // (the grid was built for this "radius": cell side = reach * slack)
        const double thr_knn = (double)(float)(reach * reach);
        const size_t hlds = hist_lds_bytes(kk);
        O3D_TRY(allow_lds(ctx, k_estimate_normals_hist, hlds));
        const int64_t gh = std::min<int64_t>(g, hist_grid(hlds));
        O3D_HIP(ctx, o3d_tmp_alloc(ctx, &si, hist_inside_bytes(gh)));
        hipLaunchKernelGGL(k_estimate_normals_hist, dim3((int)gh), dim3(kNrmBlock), hlds, ctx->stream, ctx->tgt_sorted, p_lo, p_hi,
                           ctx->cell_start, ctx->gp, kk, thr_knn, d_out, has_normals, by_pos, ref, fb_list, fb_count, 1, d_cov, (uint2 *)si);
        O3D_TRY(allow_lds(ctx, k_estimate_normals<0>, lds));
        hipLaunchKernelGGL(k_estimate_normals<0>, dim3((int)std::min<int64_t>(g, O3D_NUM_CU * 4)), dim3(kNrmBlock), lds, ctx->stream,
                           ctx->tgt_sorted, p_lo, p_hi, ctx->cell_start, ctx->gp, kk, thr, d_out, has_normals, by_pos, ref,
                           (const uint32_t *)fb_list, (const uint32_t *)fb_count, d_cov);
      } else if (search_type == O3D_SEARCH_KNN) {
        O3D_TRY(allow_lds(ctx, k_estimate_normals<0>, lds));
        hipLaunchKernelGGL(k_estimate_normals<0>, dim3((int)g), dim3(kNrmBlock), lds, ctx->stream, ctx->tgt_sorted, p_lo, p_hi,
                           ctx->cell_start, ctx->gp, kk, thr, d_out, has_normals, by_pos, ref, nullptr, nullptr, d_cov);
      } else if (search_type == O3D_SEARCH_RADIUS) {
        O3D_TRY(allow_lds(ctx, k_estimate_normals<1>, lds));
        hipLaunchKernelGGL(k_estimate_normals<1>, dim3((int)g), dim3(kNrmBlock), lds, ctx->stream, ctx->tgt_sorted, p_lo, p_hi,
                           ctx->cell_start, ctx->gp, kk, thr, d_out, has_normals, by_pos, ref, nullptr, nullptr, d_cov);
      } else if (kk <= kHistMaxKnn && !getenv("O3D_NORMALS_HEAP") && radius > 1e-15) {
        // hybrid: selection by counting (k_estimate_normals_hist); what does not fit it is left on a list for the heap kernel
        O3D_HIP(ctx, o3d_tmp_alloc(ctx, &sf, sizeof(uint32_t) * ((size_t)(p_hi - p_lo) + 64)));
        uint32_t *fb_list = (uint32_t *)sf + 64, *fb_count = (uint32_t *)sf;
        O3D_HIP(ctx, hipMemsetAsync(fb_count, 0, sizeof(uint32_t), ctx->stream));
        const size_t hlds = hist_lds_bytes(kk);
        O3D_TRY(allow_lds(ctx, k_estimate_normals_hist, hlds));
        const int64_t gh = std::min<int64_t>(g, hist_grid(hlds));
        O3D_HIP(ctx, o3d_tmp_alloc(ctx, &si, hist_inside_bytes(gh)));
        hipLaunchKernelGGL(k_estimate_normals_hist, dim3((int)gh), dim3(kNrmBlock), hlds, ctx->stream, ctx->tgt_sorted, p_lo, p_hi,
                           ctx->cell_start, ctx->gp, kk, thr, d_out, has_normals, by_pos, ref, fb_list, fb_count, 0, d_cov, (uint2 *)si);
        O3D_TRY(allow_lds(ctx, k_estimate_normals<2>, lds));
        hipLaunchKernelGGL(k_estimate_normals<2>, dim3((int)std::min<int64_t>(g, O3D_NUM_CU * 2)), dim3(kNrmBlock), lds, ctx->stream,
                           ctx->tgt_sorted, p_lo, p_hi, ctx->cell_start, ctx->gp, kk, thr, d_out, has_normals, by_pos, ref,
                           (const uint32_t *)fb_list, (const uint32_t *)fb_count, d_cov);
      } else {
        O3D_TRY(allow_lds(ctx, k_estimate_normals<2>, lds));
        hipLaunchKernelGGL(k_estimate_normals<2>, dim3((int)g), dim3(kNrmBlock), lds, ctx->stream, ctx->tgt_sorted, p_lo, p_hi,
                           ctx->cell_start, ctx->gp, kk, thr, d_out, has_normals, by_pos, ref, nullptr, nullptr, d_cov);
      }
    }
    if (d_cov)
      hipLaunchKernelGGL(k_normals_from_cov, dim3(grid_for(p_hi - p_lo)), dim3(O3D_BLOCK), 0, ctx->stream, ctx->tgt_sorted, p_lo, p_hi,
                         (const double *)d_cov, d_out, has_normals, by_pos, ref);
    if (!sharded && ng < n) hipLaunchKernelGGL(k_normals_nonfinite, dim3(grid_for(n)), dim3(O3D_BLOCK), 0, ctx->stream, d_xyz, n, d_nrm);
    O3D_HIP(ctx, hipGetLastError());
    if (mem == O3D_MEM_HOST) {
      if (sharded) {
        if (rows > 0) O3D_HIP(ctx, o3d_d2h(ctx, out_part, d_out, sizeof(double) * 3 * (size_t)rows));
      } else {
        O3D_HIP(ctx, o3d_d2h(ctx, normals, d_nrm, bytes));
      }
    }
    O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return O3D_OK;
  };
  const int st = run();
  cleanup();
  return st;
}

int o3d_estimate_normals_impl(o3d_ctx *ctx, const double *xyz, int64_t n, int search_type, int knn, double radius, int mem,
                              double *normals, int has_normals) {
  return estimate_normals_run(ctx, xyz, n, search_type, knn, radius, mem, normals, has_normals, -1, 1, nullptr, nullptr);
}

int o3d_estimate_normals_shard_impl(o3d_ctx *ctx, const double *xyz, int64_t n, int search_type, int knn, double radius, int mem,
                                    const double *oriented, int shard, int n_shards, double *out_part, int64_t *out_ng) {
  return estimate_normals_run(ctx, xyz, n, search_type, knn, radius, mem, const_cast<double *>(oriented), oriented ? 1 : 0, shard,
                              n_shards, out_part, out_ng);
}

// all_parts: n_shards x rows x 3 in rank order (rows = ceil(ng / n_shards)) = the normals of the cell-ordered cloud.  Uses the
// grid the preceding o3d_estimate_normals_shard left in the context.  xyz: the same cloud (non-finite points get (0, 0, 1)).
int o3d_estimate_normals_assemble_impl(o3d_ctx *ctx, const double *xyz, int64_t n, const double *all_parts, int n_shards, int mem,
                                       double *normals) {
  if (n < 0 || n_shards < 1 || (n > 0 && (!xyz || !all_parts || !normals)))
    return o3d_fail(ctx, O3D_ERR_INVALID_ARG, "EstimateNormals (assemble): bad arguments");
  if (n == 0) return O3D_OK;
  if (ctx->n_tgt != n || !ctx->tgt_sorted)
    return o3d_fail(ctx, O3D_ERR_STATE, "o3d_estimate_normals_assemble must follow o3d_estimate_normals_shard on the same cloud");
  const int64_t ng = ctx->n_tgt_grid;
  const int64_t rows = (ng + n_shards - 1) / n_shards;
  const size_t bytes = sizeof(double) * 3 * (size_t)n, pbytes = sizeof(double) * 3 * (size_t)std::max<int64_t>(rows * n_shards, 1);
  const double *d_xyz = xyz, *d_parts = all_parts;
  double *d_nrm = normals;
  void *sx = nullptr, *sn = nullptr, *sp = nullptr;
  auto cleanup = [&]() {
    for (void *b : {sx, sn, sp})
      if (b) o3d_tmp_free(ctx, b);
  };
  auto run = [&]() -> int {
    if (mem == O3D_MEM_HOST) {
      O3D_HIP(ctx, o3d_tmp_alloc(ctx, &sx, bytes));
      O3D_HIP(ctx, o3d_tmp_alloc(ctx, &sn, bytes));
      O3D_HIP(ctx, o3d_tmp_alloc(ctx, &sp, pbytes));
      O3D_HIP(ctx, o3d_h2d(ctx, sx, xyz, bytes));
      O3D_HIP(ctx, o3d_h2d(ctx, sp, all_parts, sizeof(double) * 3 * (size_t)(rows * n_shards)));
      d_xyz = (const double *)sx; d_nrm = (double *)sn; d_parts = (const double *)sp;
    }
    if (ng > 0)
      hipLaunchKernelGGL(k_normals_scatter, dim3(grid_for(ng)), dim3(O3D_BLOCK), 0, ctx->stream, ctx->tgt_sorted, ng, d_parts, d_nrm);
    if (ng < n) hipLaunchKernelGGL(k_normals_nonfinite, dim3(grid_for(n)), dim3(O3D_BLOCK), 0, ctx->stream, d_xyz, n, d_nrm);
    O3D_HIP(ctx, hipGetLastError());
    if (mem == O3D_MEM_HOST) O3D_HIP(ctx, o3d_d2h(ctx, normals, d_nrm, bytes));
    O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return O3D_OK;
  };
  const int st = run();
  cleanup();
  return st;
}

// own_out (device memory only): instead of writing into caller buffers of n entries, the outputs are allocated here with
// hipMalloc at their exact size -- the voxel count is known on the host before they are written -- and handed over in own_out[0..2]
// (points, normals, colours; the caller frees them).
int o3d_voxel_down_sample_impl(o3d_ctx *ctx, const double *xyz, const double *nrm, const double *col, int64_t n, double voxel,
                               int mem, double *out_xyz, double *out_nrm, double *out_col, int32_t *out_vox, int64_t *out_m,
                               double **own_out) {
  if (out_m) *out_m = 0;
  if (own_out) own_out[0] = own_out[1] = own_out[2] = nullptr;
  if (!(voxel > 0.0)) return o3d_fail(ctx, O3D_ERR_VOXEL_SIZE, "%s", o3d_status_string(O3D_ERR_VOXEL_SIZE));  // :275-277
  if (own_out && mem != O3D_MEM_DEVICE) return o3d_fail(ctx, O3D_ERR_INVALID_ARG, "VoxelDownSample: library-owned outputs are device memory");
  if (n < 0 || (n > 0 && (!xyz || (!out_xyz && !own_out))) || (nrm && !out_nrm && !own_out) || (col && !out_col && !own_out) || !out_m)
    return o3d_fail(ctx, O3D_ERR_INVALID_ARG, "VoxelDownSample: null buffer");
  if (n >= ((int64_t)1 << 31)) return o3d_fail(ctx, O3D_ERR_UNSUPPORTED, "VoxelDownSample: more than 2^31-1 points");
  if (n == 0) return O3D_OK;
  const size_t bytes = sizeof(double) * 3 * (size_t)n;
  void *stage[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  auto cleanup = [&]() {
    for (void *p : stage)
      if (p) o3d_tmp_free(ctx, p);
  };
  auto run = [&]() -> int {
    const double *d_xyz = xyz, *d_nrm = nrm, *d_col = col;
    double *o_xyz = out_xyz, *o_nrm = out_nrm, *o_col = out_col;
    int32_t *o_vox = out_vox;
    if (mem == O3D_MEM_HOST) {
      O3D_HIP(ctx, o3d_tmp_alloc(ctx, &stage[0], bytes));
      O3D_HIP(ctx, o3d_h2d(ctx, stage[0], xyz, bytes));
      d_xyz = (const double *)stage[0];
      if (nrm) {
        O3D_HIP(ctx, o3d_tmp_alloc(ctx, &stage[1], bytes));
        O3D_HIP(ctx, o3d_h2d(ctx, stage[1], nrm, bytes));
        d_nrm = (const double *)stage[1];
      }
      if (col) {
        O3D_HIP(ctx, o3d_tmp_alloc(ctx, &stage[2], bytes));
        O3D_HIP(ctx, o3d_h2d(ctx, stage[2], col, bytes));
        d_col = (const double *)stage[2];
      }
    }
    double bb[6];
    int64_t nf = 0;
    O3D_TRY(o3d_bounds(ctx, d_xyz, n, bb, &nf));
    if (nf == 0) return O3D_OK;
    VoxelParams vp;
    double ext = 0.0;
    uint64_t maxi[3];
    for (int d = 0; d < 3; d++) {
      vp.vmin[d] = bb[d] - voxel * 0.5;                       // :279
      const double vmax = bb[3 + d] + voxel * 0.5;            // :280
      ext = std::max(ext, vmax - vp.vmin[d]);
      maxi[d] = (uint64_t)floor((bb[3 + d] - vp.vmin[d]) / voxel) + 1;
    }
    if (voxel * (double)INT32_MAX < ext) return o3d_fail(ctx, O3D_ERR_VOXEL_TOO_SMALL, "%s", o3d_status_string(O3D_ERR_VOXEL_TOO_SMALL));
    vp.voxel = voxel;
    vp.bx = ceil_log2(maxi[0] + 1);
    vp.by = ceil_log2(maxi[1] + 1);
    const int bz = ceil_log2(maxi[2] + 1);
    if (vp.bx + vp.by + bz > 63)
      return o3d_fail(ctx, O3D_ERR_UNSUPPORTED, "VoxelDownSample: voxel lattice needs %d key bits (> 63)", vp.bx + vp.by + bz);

    const size_t ntiles = o3d_sort_num_tiles(n);
    const size_t nh = 256 * ntiles;
    void *keys, *vals, *tkeys, *tvals, *hist, *head, *bsum;
    O3D_TRY(o3d_scratch(ctx, 1, sizeof(uint64_t) * (size_t)n, &keys));
    O3D_TRY(o3d_scratch(ctx, 2, sizeof(uint32_t) * (size_t)n, &vals));
    O3D_TRY(o3d_scratch(ctx, 3, sizeof(uint64_t) * (size_t)n, &tkeys));
    O3D_TRY(o3d_scratch(ctx, 4, sizeof(uint32_t) * (size_t)n, &tvals));
    O3D_TRY(o3d_scratch(ctx, 5, sizeof(uint32_t) * (nh + o3d_scan_num_blocks((int64_t)nh) + 264), &hist));
    O3D_TRY(o3d_scratch(ctx, 6, sizeof(uint32_t) * (o3d_scan_num_blocks(n + 1) + 8), &bsum));
    O3D_TRY(o3d_scratch(ctx, 0, sizeof(uint32_t) * ((size_t)n + 8), &head));
    const int g = grid_for(n);
    hipLaunchKernelGGL(k_voxel_keys, dim3(g), dim3(O3D_BLOCK), 0, ctx->stream, d_xyz, n, vp, (uint64_t *)keys, (uint32_t *)vals);
    // 64 key bits so that the sentinel (all ones) of non-finite points sorts last
    O3D_TRY(o3d_radix_sort_u64(ctx, (uint64_t *)keys, (uint32_t *)vals, (uint64_t *)tkeys, (uint32_t *)tvals, (uint32_t *)hist, n,
                               nf < n ? 64 : vp.bx + vp.by + bz));
    // segments over the finite prefix [0, nf)
    hipLaunchKernelGGL(k_segment_heads, dim3(grid_for(nf)), dim3(O3D_BLOCK), 0, ctx->stream, (const uint64_t *)keys, nf, (uint32_t *)head);
    O3D_HIP(ctx, hipMemsetAsync((uint32_t *)head + nf, 0, sizeof(uint32_t), ctx->stream));
    O3D_TRY(o3d_exclusive_scan_u32(ctx, (uint32_t *)head, (uint32_t *)bsum, nf + 1));
    uint32_t *hm = (uint32_t *)ctx->pinned;
    O3D_HIP(ctx, o3d_d2h(ctx, hm, (uint32_t *)head + nf, sizeof(uint32_t)));
    O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int64_t m = (int64_t)*hm;
    // segment starts reuse the tmp-vals buffer (free after the sort)
    uint32_t *seg_start = (uint32_t *)tvals;
    hipLaunchKernelGGL(k_segment_starts, dim3(grid_for(nf)), dim3(O3D_BLOCK), 0, ctx->stream, (const uint64_t *)keys,
                       (const uint32_t *)head, nf, seg_start);
    if (own_out) {
      const size_t ob = sizeof(double) * 3 * (size_t)(m > 0 ? m : 1);
      O3D_HIP(ctx, hipMalloc((void **)&own_out[0], ob));
      o_xyz = own_out[0];
      if (nrm) { O3D_HIP(ctx, hipMalloc((void **)&own_out[1], ob)); o_nrm = own_out[1]; }
      if (col) { O3D_HIP(ctx, hipMalloc((void **)&own_out[2], ob)); o_col = own_out[2]; }
    }
    if (mem == O3D_MEM_HOST) {
      const size_t ob = sizeof(double) * 3 * (size_t)m;
      O3D_HIP(ctx, o3d_tmp_alloc(ctx, &stage[3], ob));
      o_xyz = (double *)stage[3];
      if (nrm) { O3D_HIP(ctx, o3d_tmp_alloc(ctx, &stage[4], ob)); o_nrm = (double *)stage[4]; }
      if (col) { O3D_HIP(ctx, o3d_tmp_alloc(ctx, &stage[5], ob)); o_col = (double *)stage[5]; }
      if (out_vox) { O3D_HIP(ctx, o3d_tmp_alloc(ctx, &stage[6], sizeof(int32_t) * 3 * (size_t)m)); o_vox = (int32_t *)stage[6]; }
    }
    hipLaunchKernelGGL(k_voxel_reduce, dim3(grid_for(m)), dim3(O3D_BLOCK), 0, ctx->stream, d_xyz, d_nrm, d_col, (const uint64_t *)keys,
                       (const uint32_t *)vals, (const uint32_t *)seg_start, m, nf, vp, o_xyz, nrm ? o_nrm : nullptr,
                       col ? o_col : nullptr, o_vox);
    O3D_HIP(ctx, hipGetLastError());
    if (mem == O3D_MEM_HOST) {
      const size_t ob = sizeof(double) * 3 * (size_t)m;
      O3D_HIP(ctx, o3d_d2h(ctx, out_xyz, o_xyz, ob));
      if (nrm) O3D_HIP(ctx, o3d_d2h(ctx, out_nrm, o_nrm, ob));
      if (col) O3D_HIP(ctx, o3d_d2h(ctx, out_col, o_col, ob));
      if (out_vox) O3D_HIP(ctx, o3d_d2h(ctx, out_vox, o_vox, sizeof(int32_t) * 3 * (size_t)m));
    }
    O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *out_m = m;
    return O3D_OK;
  };
  const int st = run();
  cleanup();
  if (st != O3D_OK && own_out)
    for (int k = 0; k < 3; k++) {
      if (own_out[k]) (void)hipFree(own_out[k]);
      own_out[k] = nullptr;
    }
  return st;
}

// Colour gradients of a cloud on the device (all pointers device memory; d_grad: n x 3, zero for points outside the grid).
// Rebuilds the context's grid over the cloud with cell size = radius.
int o3d_color_gradient_impl(o3d_ctx *ctx, const double *d_xyz, const double *d_nrm, const double *d_col, int64_t n, double radius,
                            int max_nn, double *d_grad) {
  if (n <= 0) return O3D_OK;
  if (max_nn > kMaxKnn) return o3d_fail(ctx, O3D_ERR_UNSUPPORTED, "colour gradient: max_nn > %d is not on the device path", kMaxKnn);
  ctx->icp_ready = 0;
  O3D_HIP(ctx, hipMemsetAsync(d_grad, 0, sizeof(double) * 3 * (size_t)n, ctx->stream));
  if (!(radius > 0.0) || max_nn < 4) return O3D_OK;  // no point can collect 4 neighbours
  O3D_TRY(o3d_build_target_grid(ctx, d_xyz, nullptr, n, radius, 1));
  const int64_t ng = ctx->n_tgt_grid;
  if (ng <= 0) return O3D_OK;
  const double thr = (double)(float)(radius * radius);  // KDTreeFlann.cpp:148
  const size_t lds = kListEntryBytes * kNrmBlock * (size_t)max_nn;
  int64_t g = (ng + kNrmBlock - 1) / kNrmBlock;
  g = std::max<int64_t>(1, std::min<int64_t>(g, O3D_NUM_CU * 24));
  O3D_TRY(allow_lds(ctx, k_color_gradient, lds));
  hipLaunchKernelGGL(k_color_gradient, dim3((int)g), dim3(kNrmBlock), lds, ctx->stream, ctx->tgt_sorted, ng, ctx->cell_start, ctx->gp,
                     max_nn, thr, d_nrm, d_col, d_grad);
  O3D_HIP(ctx, hipGetLastError());
  return O3D_OK;
}

// mode 0: knn-th neighbour distance, 1: mean knn distance, 2: radius count (see k_neighbor_query).  cloud = the points
// searched (n_cloud), q = the queries (n_q; may be the same buffer).  Outputs in `mem` space.
int o3d_neighbor_query_impl(o3d_ctx *ctx, int mode, const double *cloud_xyz, int64_t n_cloud, const double *q_xyz, int64_t n_q,
                            int knn, double radius, int mem, double *out, int32_t *out_count) {
  if (n_cloud < 0 || n_q < 0 || (n_cloud > 0 && !cloud_xyz) || (n_q > 0 && (!q_xyz || (mode == 2 ? !out_count : !out))))
    return o3d_fail(ctx, O3D_ERR_INVALID_ARG, "neighbour query: bad sizes or null buffers");
  if (n_cloud >= ((int64_t)1 << 31) || n_q >= ((int64_t)1 << 31)) return o3d_fail(ctx, O3D_ERR_UNSUPPORTED, "more than 2^31-1 points");
  if (mode != 2 && (knn < 1 || knn > kMaxKnn))
    return o3d_fail(ctx, O3D_ERR_UNSUPPORTED, "neighbour query: knn must be in [1, %d] on the device path", kMaxKnn);
  if (mode == 2 && !(radius > 0.0)) return o3d_fail(ctx, O3D_ERR_INVALID_ARG, "neighbour query: radius must be positive");
  if (n_q == 0) return O3D_OK;
  ctx->icp_ready = 0;  // the context's grid is rebuilt
  const bool same = (q_xyz == cloud_xyz);
  const double *d_cloud = cloud_xyz, *d_q = q_xyz;
  void *sc = nullptr, *sq = nullptr, *so = nullptr;
  const size_t out_bytes = (mode == 2 ? sizeof(int32_t) : sizeof(double)) * (size_t)n_q;
  auto cleanup = [&]() {
    if (sc) o3d_tmp_free(ctx, sc);
    if (sq) o3d_tmp_free(ctx, sq);
    if (so) o3d_tmp_free(ctx, so);
  };
  auto run = [&]() -> int {
    void *d_out = (mode == 2) ? (void *)out_count : (void *)out;
    if (mem == O3D_MEM_HOST) {
      if (n_cloud > 0) {
        O3D_HIP(ctx, o3d_tmp_alloc(ctx, &sc, sizeof(double) * 3 * (size_t)n_cloud));
        O3D_HIP(ctx, o3d_h2d(ctx, sc, cloud_xyz, sizeof(double) * 3 * (size_t)n_cloud));
        d_cloud = (const double *)sc;
      }
      if (same) {
        d_q = d_cloud;
      } else {
        O3D_HIP(ctx, o3d_tmp_alloc(ctx, &sq, sizeof(double) * 3 * (size_t)n_q));
        O3D_HIP(ctx, o3d_h2d(ctx, sq, q_xyz, sizeof(double) * 3 * (size_t)n_q));
        d_q = (const double *)sq;
      }
      O3D_HIP(ctx, o3d_tmp_alloc(ctx, &so, out_bytes));
      d_out = so;
    }
    double h = radius;
    if (mode == 2) O3D_TRY(o3d_build_target_grid(ctx, d_cloud, nullptr, n_cloud, radius, 1));
    else if (n_cloud > 0) O3D_TRY(build_knn_grid(ctx, d_cloud, n_cloud, &h));
    else O3D_TRY(o3d_build_target_grid(ctx, d_cloud, nullptr, 0, 1.0));
    const int kk = (mode == 2) ? 1 : knn;
    const size_t lds = kListEntryBytes * kNrmBlock * (size_t)kk;
    int64_t g = (n_q + kNrmBlock - 1) / kNrmBlock;
    g = std::max<int64_t>(1, std::min<int64_t>(g, O3D_NUM_CU * 24));
    const double thr = (double)(float)(radius * radius);  // KDTreeFlann.cpp:126
    O3D_TRY(allow_lds(ctx, k_neighbor_query<0>, lds));
    O3D_TRY(allow_lds(ctx, k_neighbor_query<1>, lds));
    O3D_TRY(allow_lds(ctx, k_neighbor_query<2>, lds));
    if (mode == 0)
      hipLaunchKernelGGL(k_neighbor_query<0>, dim3((int)g), dim3(kNrmBlock), lds, ctx->stream, ctx->tgt_sorted, ctx->cell_start, ctx->gp,
                         d_q, n_q, kk, thr, (double *)d_out, (int32_t *)nullptr);
    else if (mode == 1)
      hipLaunchKernelGGL(k_neighbor_query<1>, dim3((int)g), dim3(kNrmBlock), lds, ctx->stream, ctx->tgt_sorted, ctx->cell_start, ctx->gp,
                         d_q, n_q, kk, thr, (double *)d_out, (int32_t *)nullptr);
    else
      hipLaunchKernelGGL(k_neighbor_query<2>, dim3((int)g), dim3(kNrmBlock), lds, ctx->stream, ctx->tgt_sorted, ctx->cell_start, ctx->gp,
                         d_q, n_q, kk, thr, (double *)nullptr, (int32_t *)d_out);
    O3D_HIP(ctx, hipGetLastError());
    if (mem == O3D_MEM_HOST)
      O3D_HIP(ctx, o3d_d2h(ctx, mode == 2 ? (void *)out_count : (void *)out, so, out_bytes));
    O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return O3D_OK;
  };
  const int st = run();
  cleanup();
  return st;
}

// ComputeFPFHFeature, pipelines/registration/Feature.cpp:113-159.  out_feature: 33 doubles per point (`mem` space).
int o3d_fpfh_impl(o3d_ctx *ctx, const double *xyz, const double *nrm, int64_t n, int search_type, int knn, double radius, int mem,
                  double *out_feature) {
  if (n < 0 || (n > 0 && (!xyz || !out_feature))) return o3d_fail(ctx, O3D_ERR_INVALID_ARG, "ComputeFPFHFeature: null buffer");
  if (!nrm && n > 0)
    return o3d_fail(ctx, O3D_ERR_NEED_NORMALS, "[Open3D ERROR] [ComputeFPFHFeature] Failed because input point cloud has no normal.");
  if (search_type != O3D_SEARCH_KNN && search_type != O3D_SEARCH_HYBRID && search_type != O3D_SEARCH_RADIUS)
    return o3d_fail(ctx, O3D_ERR_INVALID_ARG, "ComputeFPFHFeature: unknown search parameter type %d", search_type);
  const bool radius_mode = search_type == O3D_SEARCH_RADIUS;
  if (!radius_mode && knn > kMaxKnn) return o3d_fail(ctx, O3D_ERR_UNSUPPORTED, "ComputeFPFHFeature: knn/max_nn > %d is not on the device path", kMaxKnn);
  if (n >= ((int64_t)1 << 31)) return o3d_fail(ctx, O3D_ERR_UNSUPPORTED, "ComputeFPFHFeature: more than 2^31-1 points");
  if (n == 0) return O3D_OK;
  ctx->icp_ready = 0;
  const size_t pbytes = sizeof(double) * 3 * (size_t)n, fbytes = sizeof(double) * 33 * (size_t)n;
  const double *d_xyz = xyz, *d_nrm = nrm;
  double *d_out = out_feature;
  void *sx = nullptr, *sn = nullptr, *so = nullptr, *sp = nullptr, *sd = nullptr, *si = nullptr, *sc = nullptr;
  auto cleanup = [&]() {
    for (void *b : {sx, sn, so, sp, sd, si, sc})
      if (b) o3d_tmp_free(ctx, b);
  };
  auto run = [&]() -> int {
    if (mem == O3D_MEM_HOST) {
      O3D_HIP(ctx, o3d_tmp_alloc(ctx, &sx, pbytes));
      O3D_HIP(ctx, o3d_tmp_alloc(ctx, &sn, pbytes));
      O3D_HIP(ctx, o3d_tmp_alloc(ctx, &so, fbytes));
      O3D_HIP(ctx, o3d_h2d(ctx, sx, xyz, pbytes));
      O3D_HIP(ctx, o3d_h2d(ctx, sn, nrm, pbytes));
      d_xyz = (const double *)sx; d_nrm = (const double *)sn; d_out = (double *)so;
    }
    O3D_HIP(ctx, o3d_tmp_alloc(ctx, &sp, fbytes));
    O3D_HIP(ctx, hipMemsetAsync(sp, 0, fbytes, ctx->stream));
    O3D_HIP(ctx, hipMemsetAsync(d_out, 0, fbytes, ctx->stream));
    const bool knn_mode = search_type == O3D_SEARCH_KNN;
    if ((!radius_mode && knn < 1) || (!knn_mode && !(radius > 0.0))) {  // no point has a neighbour besides (at most) itself: all zero
      if (mem == O3D_MEM_HOST) O3D_HIP(ctx, o3d_d2h(ctx, out_feature, d_out, fbytes));
      O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
      return O3D_OK;
    }
    double h = radius;
    if (knn_mode) O3D_TRY(build_knn_grid(ctx, d_xyz, n, &h));
    else O3D_TRY(o3d_build_target_grid(ctx, d_xyz, nullptr, n, radius, 1));
    const int64_t ng = ctx->n_tgt_grid;
    if (ng > 0 && radius_mode) {
      const double thr = (double)(float)(radius * radius);  // KDTreeFlann.cpp:126
      int64_t g = (ng + kNrmBlock - 1) / kNrmBlock;
      g = std::max<int64_t>(1, std::min<int64_t>(g, O3D_NUM_CU * 24));
      hipLaunchKernelGGL(k_fpfh_radius<0>, dim3((int)g), dim3(kNrmBlock), 0, ctx->stream, ctx->tgt_sorted, ng, ctx->cell_start, ctx->gp,
                         thr, d_nrm, (const double *)nullptr, (double *)sp);
      hipLaunchKernelGGL(k_fpfh_radius<1>, dim3((int)g), dim3(kNrmBlock), 0, ctx->stream, ctx->tgt_sorted, ng, ctx->cell_start, ctx->gp,
                         thr, d_nrm, (const double *)sp, d_out);
      O3D_HIP(ctx, hipGetLastError());
    } else if (ng > 0) {
      const double thr = (double)(float)(radius * radius);
      const size_t lds = kListEntryBytes * kNrmBlock * (size_t)knn;
      int64_t g = (ng + kNrmBlock - 1) / kNrmBlock;
      g = std::max<int64_t>(1, std::min<int64_t>(g, O3D_NUM_CU * 24));
      // the neighbour table between the passes: 12 B per neighbour slot, kept when it is at most 8 GiB
      const size_t slots = (size_t)ng * (size_t)knn;
      uint32_t *nbr_idx = nullptr;
      double *nbr_d2 = nullptr;
      int32_t *nbr_cnt = nullptr;
      if (slots * 12 <= ((size_t)8 << 30) && !getenv("O3D_FPFH_NO_TABLE")) {  // (the switch: both paths are tested)
        O3D_HIP(ctx, o3d_tmp_alloc(ctx, &sd, sizeof(double) * slots));
        O3D_HIP(ctx, o3d_tmp_alloc(ctx, &si, sizeof(uint32_t) * slots));
        O3D_HIP(ctx, o3d_tmp_alloc(ctx, &sc, sizeof(int32_t) * (size_t)ng));
        nbr_d2 = (double *)sd; nbr_idx = (uint32_t *)si; nbr_cnt = (int32_t *)sc;
      }
      O3D_TRY(allow_lds(ctx, k_fpfh<0>, lds));
      hipLaunchKernelGGL(k_fpfh<0>, dim3((int)g), dim3(kNrmBlock), lds, ctx->stream, ctx->tgt_sorted, ng, ctx->cell_start, ctx->gp, knn,
                         thr, knn_mode ? 1 : 0, d_nrm, (const double *)nullptr, (double *)sp, nbr_idx, nbr_d2, nbr_cnt);
      const size_t lds1 = nbr_cnt ? 0 : lds;
      O3D_TRY(allow_lds(ctx, k_fpfh<1>, lds1));
      hipLaunchKernelGGL(k_fpfh<1>, dim3((int)g), dim3(kNrmBlock), lds1, ctx->stream, ctx->tgt_sorted, ng, ctx->cell_start, ctx->gp, knn,
                         thr, knn_mode ? 1 : 0, d_nrm, (const double *)sp, d_out, nbr_idx, nbr_d2, nbr_cnt);
      O3D_HIP(ctx, hipGetLastError());
    }
    if (mem == O3D_MEM_HOST) O3D_HIP(ctx, o3d_d2h(ctx, out_feature, d_out, fbytes));
    O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return O3D_OK;
  };
  const int st = run();
  cleanup();
  return st;
}
