// The ICP iteration on the device.
//
//   k_icp_search_coop  ONE kernel per pass: pending update applied in place (Geometry3D.cpp:77-82), exact radius-bounded
//                   nearest neighbour over the cell grid (KDTreeFlann::SearchHybrid with max_nn = 1,
//                   geometry/KDTreeFlann.cpp:132-152, as used by Registration.cpp:60-68) -- fp32 prefilter over a
//                   wave-cooperative work queue, fp64 confirmation, exact fp64 resolution of what fp32 cannot decide --
//                   correspondence index out, and the NEXT estimate's sums (utility/Eigen.cpp:180-185 with the lambda of
//                   TransformationEstimation.cpp:79-86, the Umeyama moments of :53-59, or the two rows of
//                   ColoredICP.cpp:136-165) as fp64 MFMA rank-4 updates, one 32-double partial row per workgroup.
//   k_icp_search + k_icp_accumulate_mfma   the alternative path: per-lane fp64 search, then the sums.
//   k_icp_finalize  fixed-order sum of the workgroup partials -> this shard's 32 sums (run-to-run deterministic).
//   k_icp_solve     one lane: sum over ranks in rank order, fitness / rmse (Registration.cpp:82-89), convergence test
//                   (:162-165), 6x6 pivoted LDLT (utility/Eigen.cpp:85,116-130) or Umeyama + 3x3 Jacobi SVD, Rz*Ry*Rx
//                   (utility/Eigen.cpp:89-97), transformation = update * transformation (:158).
//
// Bit-exactness of the correspondence search: d2 = ((dx*dx)+dy*dy)+dz*dz with separate IEEE multiplies and adds (this
// file is compiled with -ffp-contract=off), accept iff d2 < (double)(float)(r*r), argmin over (d2, target index).
#include <math.h>
#include <string.h>

#include <mutex>

#include "o3d_internal.h"

#ifndef O3D_SRCH_PROF
#define O3D_SRCH_PROF 0
#endif
#ifndef O3D_COUNT_STATIC
#define O3D_COUNT_STATIC 0
#endif
#ifndef O3D_PASS_PROF
#define O3D_PASS_PROF 0
#endif
#if O3D_PASS_PROF  // profile build only (scripts/pass_phases.sh): when every wavefront of k_icp_pass passes its milestones (100 MHz
                   // clock; plain stores to the wavefront's own slots -- atomics on shared counters cost more than the pass)
#define O3D_PASS_MARKS 24
__device__ unsigned long long g_pass_t[4 * O3D_MAX_GRID][O3D_PASS_MARKS];
#define PASS_MARK(i) { if ((threadIdx.x & 63) == 0) g_pass_t[blockIdx.x * 4 + (threadIdx.x >> 6)][i] = wall_clock64(); }
// shader-clock cycles per phase of the certificate loop, summed over the wavefront's trips (slots 8..)
#define LOOP_T_DECL long long t_loop_prev = clock64(); long long t_loop_sum[6] = {0, 0, 0, 0, 0, 0};
#define LOOP_T(i) { const long long t_now = clock64(); t_loop_sum[i] += t_now - t_loop_prev; t_loop_prev = t_now; }
#define LOOP_T_STORE { if ((threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 6; i_++) g_pass_t[blockIdx.x * 4 + (threadIdx.x >> 6)][8 + i_] = (unsigned long long)t_loop_sum[i_]; }
#else
#define PASS_MARK(i)
#define LOOP_T_DECL
#define LOOP_T(i)
#define LOOP_T_STORE
#endif
#if O3D_SRCH_PROF  // profile build only (scripts/search_phases.sh): wave cycles per phase of a search trip
__device__ unsigned long long g_srch_prof[8];
#define SRCH_T(i) { const long long t_now = clock64(); if (lane == 0) atomicAdd(&g_srch_prof[i], (unsigned long long)(t_now - t_prev)); t_prev = clock64(); }
#else
#define SRCH_T(i)
#endif

#ifndef O3D_CENTRE_MIN
#define O3D_CENTRE_MIN 64
#endif

namespace {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, O3D_WAVE);
  return v;
}

// Running argmin over (d2, original index); ties -> lowest target index.
struct Best {
  double d2;
  long long idx;
  uint32_t pos;
};

__device__ __forceinline__ void consider(const double4 &p, uint32_t j, double x, double y, double z, Best &b) {
  const double ddx = x - p.x, ddy = y - p.y, ddz = z - p.z;
  double d2 = ddx * ddx;
  d2 = d2 + ddy * ddy;
  d2 = d2 + ddz * ddz;
  const long long id = __double_as_longlong(p.w);
  if (d2 < b.d2 || (d2 == b.d2 && id < b.idx)) {
    b.d2 = d2; b.idx = id; b.pos = j;
  }
}

// Scan tgt[s, e).  Four candidate loads are kept in flight per lane; indices past the end are clamped to the last
// element (re-considering a candidate never changes the running argmin).
__device__ __forceinline__ void scan_run(const double4 *__restrict__ tgt, uint32_t s, uint32_t e, double x, double y, double z,
                                         Best &b) {
  if (s >= e) return;
  const uint32_t last = e - 1;
  for (uint32_t j = s; j < e; j += 4) {
    const uint32_t j1 = min(j + 1, last), j2 = min(j + 2, last), j3 = min(j + 3, last);
    const double4 p0 = tgt[j], p1 = tgt[j1], p2 = tgt[j2], p3 = tgt[j3];
    consider(p0, j, x, y, z, b);
    consider(p1, j1, x, y, z, b);
    consider(p2, j2, x, y, z, b);
    consider(p3, j3, x, y, z, b);
  }
}

// Exact nearest neighbour with d2 < thr over the 3x3x3 cell block around the query.
// Inside the grid the block is searched centre cell first; every other cell (or x-run of cells) is visited only if a
// conservative lower bound of its distance to the query does not exceed the best distance so far.  The bound is the
// per-axis distance to the neighbouring cell's face, shrunk by a margin that dominates the rounding of the cell
// assignment, so pruning never changes the result.  All cell-table reads of the block are issued up front so that
// they overlap (one memory latency instead of nine).
__device__ __forceinline__ void nn_search(const double4 *__restrict__ tgt, const uint32_t *__restrict__ cell_start,
                                          const GridParams &gp, double x, double y, double z, Best &b) {
  const int cx = o3d_cell_coord(x, gp.gmin[0], gp.inv_h, gp.nx);
  const int cy = o3d_cell_coord(y, gp.gmin[1], gp.inv_h, gp.ny);
  const int cz = o3d_cell_coord(z, gp.gmin[2], gp.inv_h, gp.nz);
  const bool inside = cx >= 0 && cx < gp.nx && cy >= 0 && cy < gp.ny && cz >= 0 && cz < gp.nz;
  if (inside) {
    const uint32_t rowc = o3d_row_index(gp, cx, cy, cz);
    const uint32_t s0 = cell_start[rowc], s1 = cell_start[rowc + 1], s2 = cell_start[rowc + 2], s3 = cell_start[rowc + 3];
    // centre column of the eight neighbouring rows (dense table: rows outside the grid read a clamped, valid address;
    // bricks: the halo holds them, empty)
    uint32_t rs1[8], rs2[8], rrow[8];
    {
      int k = 0;
#pragma unroll
      for (int dz = -1; dz <= 1; dz++)
#pragma unroll
        for (int dy = -1; dy <= 1; dy++) {
          if (dz == 0 && dy == 0) continue;
          const int zz = min(max(cz + dz, 0), gp.nz - 1), yy = min(max(cy + dy, 0), gp.ny - 1);
          rrow[k] = gp.brick_bits ? rowc + (uint32_t)(dy * o3d_row_stride_y(gp) + dz * o3d_row_stride_z(gp))
                                  : ((uint32_t)zz * (uint32_t)gp.ny + (uint32_t)yy) * (uint32_t)gp.nxp + (uint32_t)cx;
          rs1[k] = cell_start[rrow[k] + 1];
          rs2[k] = cell_start[rrow[k] + 2];
          k++;
        }
    }
    scan_run(tgt, s1, s2, x, y, z, b);
    const double h = gp.h;
    const double rx = x - gp.gmin[0], ry = y - gp.gmin[1], rz = z - gp.gmin[2];
    const double fx = rx - (double)cx * h, fy = ry - (double)cy * h, fz = rz - (double)cz * h;
    const double mx = 1e-6 * h + 1e-14 * fabs(rx), my = 1e-6 * h + 1e-14 * fabs(ry), mz = 1e-6 * h + 1e-14 * fabs(rz);
    double t;
    t = fmax(fx - mx, 0.0); const double axm = t * t;
    t = fmax(h - fx - mx, 0.0); const double axp = t * t;
    t = fmax(fy - my, 0.0); const double aym = t * t;
    t = fmax(h - fy - my, 0.0); const double ayp = t * t;
    t = fmax(fz - mz, 0.0); const double azm = t * t;
    t = fmax(h - fz - mz, 0.0); const double azp = t * t;
    if (axm <= b.d2) scan_run(tgt, s0, s1, x, y, z, b);
    if (axp <= b.d2) scan_run(tgt, s2, s3, x, y, z, b);
    {
      int k = 0;
#pragma unroll
      for (int dz = -1; dz <= 1; dz++) {
        const double bz = dz < 0 ? azm : (dz > 0 ? azp : 0.0);
        const int zz = cz + dz;
#pragma unroll
        for (int dy = -1; dy <= 1; dy++) {
          if (dz == 0 && dy == 0) continue;
          const double byz = bz + (dy < 0 ? aym : (dy > 0 ? ayp : 0.0));
          const int yy = cy + dy;
          if (byz <= b.d2 && zz >= 0 && zz < gp.nz && yy >= 0 && yy < gp.ny) {
            uint32_t s = rs1[k], e = rs2[k];
            if (axm + byz <= b.d2) s = cell_start[rrow[k]];      // corner reach: widen the run to the left cell
            if (axp + byz <= b.d2) e = cell_start[rrow[k] + 3];  // ... and to the right cell
            scan_run(tgt, s, e, x, y, z, b);
          }
          k++;
        }
      }
    }
  } else if (!gp.brick_bits) {  // (a bricked table always has the margin: a query outside is a cell from every point)
    // query outside the target's bounding grid (or NaN): plain 27-cell search, x-neighbours as one run
    const int lo = max(cx - 1, -1), hi = min(cx + 1, gp.nx);
    if (lo <= hi) {
      for (int dz = -1; dz <= 1; dz++) {
        const int zz = cz + dz;
        if (zz < 0 || zz >= gp.nz) continue;
        for (int dy = -1; dy <= 1; dy++) {
          const int yy = cy + dy;
          if (yy < 0 || yy >= gp.ny) continue;
          const uint32_t row = ((uint32_t)zz * (uint32_t)gp.ny + (uint32_t)yy) * (uint32_t)gp.nxp;
          scan_run(tgt, cell_start[row + (uint32_t)(lo + 1)], cell_start[row + (uint32_t)(hi + 2)], x, y, z, b);
        }
      }
    }
  }
}

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Per-lane running sums of the next estimate for one accepted correspondence.
template <int EST>
__device__ __forceinline__ void accumulate(double (&acc)[30], double x, double y, double z, const Best &b,
                                           const double4 &p, double n0, double n1, double n2, double cx0, double cy0,
                                           double cz0) {
  if (EST != O3D_EST_POINT_TO_POINT) {
    double r = (x - p.x) * n0;
    r = r + (y - p.y) * n1;
    r = r + (z - p.z) * n2;
    double J[6];
    J[0] = y * n2 - z * n1;
    J[1] = z * n0 - x * n2;
    J[2] = x * n1 - y * n0;
    J[3] = n0; J[4] = n1; J[5] = n2;
    int k = 0;
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
      for (int c = a; c < 6; c++) acc[k++] += J[a] * J[c];
#pragma unroll
    for (int a = 0; a < 6; a++) acc[21 + a] += J[a] * r;
    acc[27] += r * r;
  } else {
    const double sx = x - cx0, sy = y - cy0, sz = z - cz0;
    const double tx = p.x - cx0, ty = p.y - cy0, tz = p.z - cz0;
    acc[0] += sx; acc[1] += sy; acc[2] += sz;
    acc[3] += tx; acc[4] += ty; acc[5] += tz;
    acc[6] += tx * sx; acc[7] += tx * sy; acc[8] += tx * sz;
    acc[9] += ty * sx; acc[10] += ty * sy; acc[11] += ty * sz;
    acc[12] += tz * sx; acc[13] += tz * sy; acc[14] += tz * sz;
    acc[15] += (sx * sx + sy * sy) + sz * sz;
  }
  acc[O3D_SUM_D2] += b.d2;
  acc[O3D_SUM_NC] += 1.0;
}

// The two rows of one correspondence of TransformationEstimationForColoredICP::ComputeTransformation
// (ColoredICP.cpp:136-165): Jg / rg geometric (point-to-plane scaled by sqrt(lambda)), Jp / rp photometric.
// tc = {target intensity, colour gradient}, is = source intensity.
__device__ __forceinline__ void colored_rows(double x, double y, double z, const double4 &p, double n0, double n1, double n2,
                                             const double4 &tc, double is, double slg, double slp, double (&Jg)[6], double &rg,
                                             double (&Jp)[6], double &rp) {
  double dot = (x - p.x) * n0;
  dot = dot + (y - p.y) * n1;
  dot = dot + (z - p.z) * n2;
  Jg[0] = slg * (y * n2 - z * n1);
  Jg[1] = slg * (z * n0 - x * n2);
  Jg[2] = slg * (x * n1 - y * n0);
  Jg[3] = slg * n0; Jg[4] = slg * n1; Jg[5] = slg * n2;
  rg = slg * dot;
  const double vpx = x - dot * n0, vpy = y - dot * n1, vpz = z - dot * n2;  // vs projected into vt's tangent plane
  double is0 = tc.y * (vpx - p.x);
  is0 = is0 + tc.z * (vpy - p.y);
  is0 = is0 + tc.w * (vpz - p.z);
  is0 = is0 + tc.x;
  // -dit^T (I - n n^T)
  const double nn[3] = {n0, n1, n2};
  double m[3];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    double a = (-tc.y) * ((c == 0 ? 1.0 : 0.0) - n0 * nn[c]);
    a = a + (-tc.z) * ((c == 1 ? 1.0 : 0.0) - n1 * nn[c]);
    a = a + (-tc.w) * ((c == 2 ? 1.0 : 0.0) - n2 * nn[c]);
    m[c] = a;
  }
  Jp[0] = slp * (y * m[2] - z * m[1]);
  Jp[1] = slp * (z * m[0] - x * m[2]);
  Jp[2] = slp * (x * m[1] - y * m[0]);
  Jp[3] = slp * m[0]; Jp[4] = slp * m[1]; Jp[5] = slp * m[2];
  rp = slp * (is - is0);
}

// per-lane form of the coloured sums: same layout as point-to-plane
__device__ __forceinline__ void accumulate_colored(double (&acc)[30], double x, double y, double z, const Best &b, const double4 &p,
                                                   double n0, double n1, double n2, const double4 &tc, double is, double slg,
                                                   double slp) {
  double J[2][6], r[2];
  colored_rows(x, y, z, p, n0, n1, n2, tc, is, slg, slp, J[0], r[0], J[1], r[1]);
#pragma unroll
  for (int j = 0; j < 2; j++) {
    int k = 0;
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
      for (int c = a; c < 6; c++) acc[k++] += J[j][a] * J[j][c];
#pragma unroll
    for (int a = 0; a < 6; a++) acc[21 + a] += J[j][a] * r[j];
    acc[27] += r[j] * r[j];
  }
  acc[O3D_SUM_D2] += b.d2;
  acc[O3D_SUM_NC] += 1.0;
}

// Device-scope accesses: written through to / read from where every XCD sees them (see pass_tail).
__device__ __forceinline__ double load_device_coherent(const double *p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void store_device_coherent(double *p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}

// THE way to write an entry of a partial row (o3d_ctx::partials, PassArgs::partials).  INVARIANT: every store to a partial
// row, a group row (gsums) or a ticket, and every load of one by pass_tail, is a device-scope access.  pass_tail reads the
// rows inside the kernel that wrote them, from whichever XCD its last workgroup runs on; the only ordering it has is
// relaxed agent-scope atomics on the tickets, a workgroup-scope release and s_waitcnt -- no device-wide fence (28 -> 55 us
// per iteration at 100 k points when tried).  ONE plain store to a row leaves the value in the writer's XCD's L2 and the
// tail adds what memory held before: the previous registration's row, silently (this happened with the zero row of a
// certificate half that stands aside).  Rows that only a LATER kernel reads need nothing of this, but go through here as
// well, so that a writer that moves into a kernel with a tail cannot bring the bug with it.  O3D_TAIL_CHECK=1
// (o3d_launch_pass) adds every pass's rows once more with k_finalize_plain and compares bit for bit.
__device__ __forceinline__ void store_partial_row(double *partials, size_t row, int k, double v) {
  store_device_coherent(partials + row * O3D_NSUMS + k, v);
}

// Wavefront shuffle reduction, then a fixed-order LDS combine across the four waves -> one partial row per chunk.
__device__ __forceinline__ void block_reduce_store(double (&acc)[30], double (*red)[O3D_NSUMS], double *__restrict__ partials,
                                                   unsigned chunk) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 30; k++) {
    double v = wave_sum(acc[k]);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < O3D_NSUMS) {
    double v = 0.0;
    if (threadIdx.x < 30) v = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    store_partial_row(partials, chunk, (int)threadIdx.x, v);
  }
}

// Per-context counters in device memory (o3d_ctx::counters, PassArgs::ctr): contexts that run concurrently on one GPU
// (batch workers, shards rehearsed on one device) never see each other's.
//   [0..7] optional statistics of the cooperative search (tests and tuning): [0] wave trips, [1] dense item trips, [2] runs
//          listed, [3] queries resolved by the exact fp64 search, [4] items (four candidates each), [5] valid lanes,
//          [6] queries whose previous match was certified without a search
//   [8]    number of fp32 winners whose fp64 confirmation failed (must stay 0: see k_icp_search_coop)

struct PassArgs {
  int stats;
  unsigned long long *ctr;  // the context's counters (O3D_CTR_*)
  double *src;           // source in processing order: transformed in place (written by the search kernels)
  const double *src_in;  // where the search kernels read the source from: src, or the untouched copy (hypothesis mode)
  int64_t n_src;
  const double4 *tgt;
  const float4 *tgt_local;
  const double *nrm;
  int stream;            // k_icp_cert: streaming loads and stores for the source, the match positions and the bounds
  const double2 *tn;     // point and normal side by side (three double2 per sorted target point: 48 bytes), or null
  const uint32_t *tidx;  // original index of every sorted target point (what tgt[].w holds, as a dense 4-byte array)
  const uint32_t *cell_start;
  const IcpState *st;
  int32_t *corr;   // per processing position: target index, or -1 none
  uint32_t *pos;   // per processing position: position of the match in tgt_sorted (O3D_POS_NONE: none)
  int seed;        // k_icp_search_coop: use the previous pass's match as the first pruning bound
  int listed;      // k_icp_search_coop: search the work list of k_icp_cert only (unless IcpState::skip_cert)
  float gap;       // k_icp_search_coop: scan this far beyond the best distance, so that the bounds have slack (absolute length)
  float *lb;       // per processing position: lower bound on the distance to every target point but the match
  uint32_t *work;        // work list: per wavefront, from the start of its range, the positions k_icp_cert could not certify
  uint32_t *work_count;  // per segment
  int64_t seg_len;       // positions per segment (a multiple of 64)
  int cert_row0;   // first partial row of k_icp_cert (the rows of k_icp_search_coop come first)
  double *partials;
  double thr;
  // the pass's last kernel ends the pass itself (pass_tail): 0 = no, 1 = sum the partial rows, 2 = ... and run the estimate
  int tail;
  int grid_x;            // workgroups of this pass (a batched launch over several registrations may bring more: they leave)
  int nparts;            // partial rows of the pass
  int has_normals;
  int log_rows;
  unsigned *tickets;     // 33 arrival counters, all zero between launches
  double *gsums;         // 32 rows of O3D_NSUMS: one per group of workgroups
  double *sums_out;      // the pass's 32 sums
  double *log;
  IcpState *st_rw;
  // coloured ICP only (ColoredICP.cpp:125-177)
  const double4 *tcol;  // per sorted target point: {intensity, colour gradient}
  const double *sint;   // per processing position: source intensity
  double slg, slp;      // sqrt(lambda_geometric), sqrt(1 - lambda_geometric)
};
#define O3D_CORR_NONE (-1)
#define O3D_POS_NONE 0xFFFFFFFFu
#define O3D_CORR_PENDING (-2)
#define O3D_CORR_CANDIDATE (-3)

// A value that is the same in every lane, moved to scalar registers (the compiler keeps loads from the loop state in
// vector registers otherwise: 32 of them for the update matrix alone).
__device__ __forceinline__ double uniform_f64(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

// Defined behind the estimate's algebra below.  Every thread of the workgroup calls it once the workgroup's partial row
// has been stored.
__device__ __forceinline__ void pass_tail(const PassArgs &a, double *lds, unsigned chunk, unsigned gdim);

__device__ __forceinline__ void load_update(const IcpState *st, double (&T)[12], double (&w)[4]) {
#pragma unroll
  for (int c = 0; c < 4; c++) {
#pragma unroll
    for (int r = 0; r < 3; r++) T[c * 3 + r] = uniform_f64(st->update[c * 4 + r]);  // element (r, c) at T[c * 3 + r]
    w[c] = uniform_f64(st->update[c * 4 + 3]);
  }
}

// Eigen 4x4 * [p;1], then divide by w (Geometry3D.cpp:79-80)
__device__ __forceinline__ void apply_update(const double (&T)[12], const double (&w)[4], double &x, double &y, double &z) {
  const double nx_ = ((T[0] * x + T[3] * y) + T[6] * z) + T[9];
  const double ny_ = ((T[1] * x + T[4] * y) + T[7] * z) + T[10];
  const double nz_ = ((T[2] * x + T[5] * y) + T[8] * z) + T[11];
  // Last row (0, 0, 0, 1) -- every rigid or similarity update: w is exactly 1 for finite input and the division is the
  // identity, so it is skipped (wave-uniform branch).  For a non-finite coordinate the reference's 0 * x term makes w,
  // and with it all three results, NaN; here the affected components alone turn non-finite: either way the point
  // falls outside the grid and gets no correspondence.
  if (w[0] == 0.0 && w[1] == 0.0 && w[2] == 0.0 && w[3] == 1.0) {
    x = nx_; y = ny_; z = nz_;
    return;
  }
  const double nw_ = ((w[0] * x + w[1] * y) + w[2] * z) + w[3];
  x = nx_ / nw_; y = ny_ / nw_; z = nz_ / nw_;
}

// Generic search pass: every lane searches its own 3x3x3 cell block in fp64 (any grid, any source distribution).
// Applies the pending update in place, writes the correspondence and its position in tgt_sorted.
__global__ __launch_bounds__(O3D_BLOCK) void k_icp_search(PassArgs a, GridParams gp) {
  if (a.st->done) return;
  const int apply = a.st->apply_update;
  double T[12], w[4];
  load_update(a.st, T, w);
  // Workgroups b and b + 8 share an XCD (round-robin dispatch): give each XCD one contiguous eighth of the curve
  // order so that neighbouring patches meet in the same L2.  The launch grid is a multiple of 8.
  const int64_t per = ((a.n_src + gridDim.x - 1) / gridDim.x + O3D_BLOCK - 1) / O3D_BLOCK * O3D_BLOCK;
  const unsigned chunk = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int64_t beg = (int64_t)chunk * per;
  const int64_t end = beg + per < a.n_src ? beg + per : a.n_src;
  for (int64_t i = beg + threadIdx.x; i < end; i += O3D_BLOCK) {
    double x = a.src_in[3 * i], y = a.src_in[3 * i + 1], z = a.src_in[3 * i + 2];
    if (apply) {
      apply_update(T, w, x, y, z);
      a.src[3 * i] = x; a.src[3 * i + 1] = y; a.src[3 * i + 2] = z;
    }
    Best b;
    b.d2 = a.thr; b.idx = -1; b.pos = 0;
    nn_search(a.tgt, a.cell_start, gp, x, y, z, b);
    a.corr[i] = (int32_t)b.idx;
    a.pos[i] = b.idx >= 0 ? b.pos : O3D_POS_NONE;
    a.lb[i] = 0.f;  // this search leaves no bound behind: a cooperative pass that follows searches everything
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same sums on the matrix cores.  Every sum of the estimate is an entry of  S = sum_p a(p) b(p)^T  for two short
// per-correspondence vectors that differ in their last component only:
//   point-to-plane  a = (J0..J5, r, 1), b = (J0..J5, r, d2):  JTJ = S[0..5][0..5], JTr = S[c][6], sum r^2 = S[6][6], sum d2 = S[7][7]
//   point-to-point  a = b = (s - c, t - c, 1, d2): sum (s-c) = S[0..2][6], sum (t-c) = S[3..5][6], sum (t-c)(s-c)^T = S[b][3+a],
//                                                  sum |s-c|^2 = S[0][0] + S[1][1] + S[2][2], sum d2 = S[6][7]
// and the number of correspondences is counted with the wavefront's ballot (an integer, exact).  Only the upper triangle of
// 4 x 4 tiles is needed -- (0,0), (0,1), (1,1) -- and v_mfma_f64_4x4x4_4b_f64 computes four independent 4 x 4 x 4 products
// at once: a wavefront stages the 64 vector pairs of one trip in LDS (component-major) and issues 12 of them, each adding
// 4 points to one tile in each of its four blocks (the blocks' partial tiles are added when the pass ends).  Lane l
// supplies A[block (l >> 2) & 3][row l & 3][k = l >> 4] and B[block][column l & 3][k] and holds
// D[block][row l >> 4][column l & 3] (scripts/micro/mfma_f64_4x4_layout.hip asks the hardware): one double per lane and
// tile, three in all, instead of 30 per-lane accumulators -- which is what lets the accumulation ride along in the search
// kernel (k_icp_search_coop) without costing it its occupancy.
// Rounds 1-3 used 16 v_mfma_f64_16x16x4_f64 per trip on 9-component vectors: 16384 multiply-adds for the 3072 needed, and on
// MI355X, whose fp64 matrix rate equals its fp64 vector rate, 1024 of the ~2800 cycles a SIMD spent per wavefront and trip
// of the certifying pass (profiles/r03_m_*: that pass is bound by the SIMDs' fp64 issue, not by memory).  The 4 x 4 form
// issues exactly the 3072: 192 cycles.
// fp64 throughout; products are fused into the accumulation (the per-lane form rounds them first), which moves the
// sums by a few ulp -- far inside the 1e-6 parity bar of the estimate, and the correspondences are not affected.
typedef double v2f64 __attribute__((ext_vector_type(2)));
constexpr int kVComp = 9;     // staged rows: a0..a7 and b7
constexpr int kVStride = 80;  // doubles per component row: 64 points + padding; 80 puts rows c and c + 1 on different halves of the banks
struct VBuf {
  double v[kVComp][kVStride];
};
struct Sums {
  double t00, t01, t11;  // the lane's entries of the tiles (0,0), (0,1), (1,1)
  uint32_t n;            // correspondences (wave-uniform)
};

template <int EST>
__device__ __forceinline__ void make_v(double (&v)[kVComp], bool use, double x, double y, double z, const double4 &p, double n0,
                                       double n1, double n2, double d2, double cx0, double cy0, double cz0) {
  if (EST != O3D_EST_POINT_TO_POINT) {
    double r = (x - p.x) * n0;
    r = r + (y - p.y) * n1;
    r = r + (z - p.z) * n2;
    v[0] = y * n2 - z * n1;
    v[1] = z * n0 - x * n2;
    v[2] = x * n1 - y * n0;
    v[3] = n0; v[4] = n1; v[5] = n2;
    v[6] = r; v[7] = 1.0; v[8] = d2;
  } else {
    v[0] = x - cx0; v[1] = y - cy0; v[2] = z - cz0;
    v[3] = p.x - cx0; v[4] = p.y - cy0; v[5] = p.z - cz0;
    v[6] = 1.0; v[7] = d2; v[8] = d2;
  }
  if (!use) {
#pragma unroll
    for (int c = 0; c < kVComp; c++) v[c] = 0.0;
  }
}

// S += sum over the wavefront's 64 lanes of a b^T (v = a0..a7, b7).  All 64 lanes must call it (lanes without a
// correspondence bring zeros).
__device__ __forceinline__ void mfma_rank64(VBuf &vb, int lane, const double (&v)[kVComp], Sums &S) {
  wave_lds_sync();  // the buffer may alias LDS the wavefront was reading until now
#pragma unroll
  for (int c = 0; c < kVComp; c++) vb.v[c][lane] = v[c];
  wave_lds_sync();
  const int c4 = lane & 3, pt = 4 * ((lane >> 2) & 3) + (lane >> 4);  // the lane's row / column of a tile; its point of a group of 16
  const double *lo = &vb.v[c4][pt];                       // components 0..3
  const double *hi_a = &vb.v[4 + c4][pt];                 // components 4..7 of a
  const double *hi_b = &vb.v[c4 == 3 ? 8 : 4 + c4][pt];   // components 4..7 of b
#pragma unroll
  for (int g = 0; g < 4; g++) {  // points 16 g .. 16 g + 15: four per block
    const double l = lo[16 * g], ha = hi_a[16 * g], hb = hi_b[16 * g];
    S.t00 = __builtin_amdgcn_mfma_f64_4x4x4f64(l, l, S.t00, 0, 0, 0);
    S.t01 = __builtin_amdgcn_mfma_f64_4x4x4f64(l, hb, S.t01, 0, 0, 0);
    S.t11 = __builtin_amdgcn_mfma_f64_4x4x4f64(ha, hb, S.t11, 0, 0, 0);
  }
  wave_lds_sync();
}

// One correspondence's contribution to S: one vector pair for point-to-point / point-to-plane, two (geometric, photometric)
// for coloured ICP.  Every lane of the wavefront must call it.
template <int EST>
__device__ __forceinline__ void push_sums(VBuf &vb, int lane, Sums &S, bool use, double x, double y, double z, const double4 &p,
                                          double n0, double n1, double n2, double d2, double cx0, double cy0, double cz0,
                                          const double4 &tc, double is, double slg, double slp) {
  double v[kVComp];
  S.n += (uint32_t)__popcll(__ballot(use));
  if (EST == O3D_EST_COLORED_ICP) {
    double Jg[6], Jp[6], rg, rp;
    colored_rows(x, y, z, p, n0, n1, n2, tc, is, slg, slp, Jg, rg, Jp, rp);
#pragma unroll
    for (int c = 0; c < 6; c++) v[c] = use ? Jg[c] : 0.0;
    v[6] = use ? rg : 0.0; v[7] = use ? 1.0 : 0.0; v[8] = use ? d2 : 0.0;
    mfma_rank64(vb, lane, v, S);
#pragma unroll
    for (int c = 0; c < 6; c++) v[c] = use ? Jp[c] : 0.0;
    v[6] = use ? rp : 0.0; v[7] = 0.0; v[8] = 0.0;
    mfma_rank64(vb, lane, v, S);
  } else {
    make_v<EST>(v, use, x, y, z, p, n0, n1, n2, d2, cx0, cy0, cz0);
    mfma_rank64(vb, lane, v, S);
  }
}

// Entry (row, col), row <= col, of S that sum k of the partial row comes from (k < 30, not the count); point-to-point sum 15
// is the trace term.  (15, 15) stands for "always zero".
template <int EST>
__device__ __forceinline__ void sum_source(int k, int &row, int &col) {
  row = 15; col = 15;
  if (EST != O3D_EST_POINT_TO_POINT) {
    if (k < 21) {
      int a = 0, rem = k;
      while (rem >= 6 - a) { rem -= 6 - a; a++; }
      row = a; col = a + rem;
    } else if (k < 27) { row = k - 21; col = 6; }
    else if (k == 27) { row = 6; col = 6; }
    else if (k == O3D_SUM_D2) { row = 7; col = 7; }
  } else {
    if (k < 6) { row = k; col = 6; }
    else if (k < 15) { row = (k - 6) % 3; col = 3 + (k - 6) / 3; }  // (t-c)_a (s-c)_b = S[b][3 + a]: a = b is symmetric
    else if (k == O3D_SUM_D2) { row = 6; col = 7; }
  }
}

// Combine the blocks of every tile and the workgroup's wavefronts in a fixed order and write one 32-double partial row.
constexpr int kRedDoubles = 3 * 64;  // per wavefront; the four counts follow the four wavefronts' tiles
__device__ __forceinline__ double red_entry(const double *red, int row, int col) {
  if (row > 7 || col > 7) return 0.0;
  const int tile = (row >> 2) + (col >> 2), at = 16 * (row & 3) + (col & 3);  // D[block][row][col] sits in lane 16 row + 4 block + col
  double v = 0.0;
#pragma unroll
  for (int w = 0; w < O3D_BLOCK / O3D_WAVE; w++) {
    const double *t = red + w * kRedDoubles + tile * 64 + at;
    const double s = ((t[0] + t[4]) + t[8]) + t[12];
    v = w == 0 ? s : v + s;
  }
  return v;
}
template <int EST>
__device__ __forceinline__ void mfma_reduce_store(const Sums &S, double *red /* 4 * kRedDoubles + 4 doubles of LDS */, double *__restrict__ partials,
                                                  unsigned chunk) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  red[wave * kRedDoubles + lane] = S.t00;
  red[wave * kRedDoubles + 64 + lane] = S.t01;
  red[wave * kRedDoubles + 128 + lane] = S.t11;
  if (lane == 0) red[(O3D_BLOCK / O3D_WAVE) * kRedDoubles + wave] = (double)S.n;
  __syncthreads();
  if (threadIdx.x < O3D_NSUMS) {
    const int k = threadIdx.x;
    double v = 0.0;
    if (k == O3D_SUM_NC) {
      const double *c = red + (O3D_BLOCK / O3D_WAVE) * kRedDoubles;
      v = ((c[0] + c[1]) + c[2]) + c[3];
    } else if (k < 30) {
      int row, col;
      sum_source<EST>(k, row, col);
      v = red_entry(red, row, col);
      if (EST == O3D_EST_POINT_TO_POINT && k == 15) {
        v = 0.0;
        for (int d = 0; d < 3; d++) v += red_entry(red, d, d);
      }
    }
    store_partial_row(partials, chunk, k, v);  // pass_tail may read it from another XCD
  }
}

// Accumulation pass of the alternative path (after the per-lane fp64 search k_icp_search): streams the source with its
// correspondences, gathers the matched target point (and normal) by position, sums on the matrix cores.
template <int EST>
__global__ __launch_bounds__(O3D_BLOCK) void k_icp_accumulate_mfma(PassArgs a) {
  if (a.st->done) return;
  // the staging buffers and the final cross-wavefront combine share LDS (the combine starts after the last trip)
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[sizeof(VBuf) * (O3D_BLOCK / O3D_WAVE)];
  static_assert(sizeof(VBuf) * (O3D_BLOCK / O3D_WAVE) >= sizeof(double) * ((O3D_BLOCK / O3D_WAVE) * kRedDoubles + O3D_BLOCK / O3D_WAVE), "combine buffer must fit");
  VBuf *vbs = reinterpret_cast<VBuf *>(lds_raw);
  double *red = reinterpret_cast<double *>(lds_raw);
  const int lane = threadIdx.x & 63;
  VBuf &vb = vbs[threadIdx.x >> 6];
  const double cx0 = a.st->centre[0], cy0 = a.st->centre[1], cz0 = a.st->centre[2];
  Sums S = {0.0, 0.0, 0.0, 0u};
  const int64_t per = ((a.n_src + gridDim.x - 1) / gridDim.x + O3D_BLOCK - 1) / O3D_BLOCK * O3D_BLOCK;
  const unsigned chunk = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int64_t beg = (int64_t)chunk * per;
  const int64_t end = beg + per < a.n_src ? beg + per : a.n_src;
  for (int64_t base = beg; base < end; base += O3D_BLOCK) {  // block-uniform: every wavefront runs every trip
    const int64_t i = base + threadIdx.x;
    int32_t c = O3D_CORR_NONE;
    if (i < end) c = a.corr[i];
    const bool use = c >= 0;
    double x = 0.0, y = 0.0, z = 0.0, n0 = 0.0, n1 = 0.0, n2 = 0.0, d2 = 0.0, is = 0.0;
    double4 p = make_double4(0.0, 0.0, 0.0, 0.0), tc = make_double4(0.0, 0.0, 0.0, 0.0);
    if (use) {
      const uint32_t pos = a.pos[i];
      x = a.src[3 * i]; y = a.src[3 * i + 1]; z = a.src[3 * i + 2];
      p = a.tgt[pos];
      if (EST != O3D_EST_POINT_TO_POINT) {
        if (a.tn) {  // (the set-up that wrote the 48-byte records wrote no separate normals)
          const double2 r1 = a.tn[3 * (size_t)pos + 1], r2 = a.tn[3 * (size_t)pos + 2];
          n0 = r1.y; n1 = r2.x; n2 = r2.y;
        } else {
          n0 = a.nrm[3 * (size_t)pos]; n1 = a.nrm[3 * (size_t)pos + 1]; n2 = a.nrm[3 * (size_t)pos + 2];
        }
      }
      if (EST == O3D_EST_COLORED_ICP) { tc = a.tcol[pos]; is = a.sint[i]; }
      const double ddx = x - p.x, ddy = y - p.y, ddz = z - p.z;
      d2 = ddx * ddx;
      d2 = d2 + ddy * ddy;
      d2 = d2 + ddz * ddz;
    }
    push_sums<EST>(vb, lane, S, use, x, y, z, p, n0, n1, n2, d2, cx0, cy0, cz0, tc, is, a.slg, a.slp);
  }
  __syncthreads();  // every wavefront is done with its staging buffer
  mfma_reduce_store<EST>(S, red, a.partials, chunk);
}

// ---------------------------------------------------------------------------------------------------------------
// fp32 prefilter search (the fast path).  Same 3x3x3 cell block per lane as the generic search, but
//   * candidates are read as 16-byte fp32 records (tgt_local: position relative to the candidate's own cell origin,
//     w = padded x cell index) and compared in fp32 relative to the QUERY's cell origin -- 11 full-rate VALU
//     instructions and half the bytes per candidate instead of an fp64 evaluation;
//   * the centre cell is scanned first; the up-to-ten other runs that survive the face-distance pruning are written to
//     a per-lane list in LDS and walked by ONE flattened loop, so a wavefront iterates max-over-lanes of the per-lane
//     work instead of the sum over ten divergent scan sites;
//   * the lane keeps the best and the second-best fp32 distance.  The winner is handed to k_icp_accumulate, which
//     confirms it in fp64 with the arithmetic of the exact search.  When the fp32 result is not decisive -- runner-up
//     within the fp32 error bound of the winner, winner within the bound of the acceptance threshold, query outside
//     the grid -- the point is searched again exactly in fp64 (wave_exact_search).  The result is therefore always the exact
//     argmin over (d2, index).
// fp32 error budget (h = cell size): candidate and query coordinates are formed relative to the query's cell origin with
// magnitude <= 2h; each is within 4e-7 h of its exact value, so |d2_fp32 - d2| <= 1.6e-6 h^2 for every candidate of
// the block (d2 <= 12 h^2 is irrelevant beyond ~h^2).  kF32Err = 4e-6 (x h^2) is used as the bound e: "decisive" means
// second - best > 2e and |best - thr| > e; runs are pruned only when their face distance exceeds best + 4e.
constexpr float kF32Err = 4e-6f;
constexpr float kSeedMax = 0.9f;   // a seed farther than sqrt(kSeedMax) cells prunes worse than the centre-cell phase
constexpr uint32_t kCentrePhaseMin = O3D_CENTRE_MIN;  // items (of four candidates) per wavefront that make a cooperative round of the own cells worth its set-up
constexpr int kMaxRuns = 11;  // rest of the own cell, left cell, right cell, eight neighbouring rows

// Four consecutive cell-table entries in one 16-byte load (the table is only 4-byte aligned at an arbitrary cell:
// gfx950 global loads take dword-aligned dwordx4 addresses).  One L1 access instead of four.
struct __attribute__((packed, aligned(4))) CellQuad {
  uint32_t v[4];
};

struct F32State {
  float best, second;
  uint32_t bpos;
};

// Up to four candidates of tgt_local[j, e): one address, four 16-byte loads at immediate offsets.  Slots past the end
// read the next cell's points (the array carries four entries of padding) and are neutralised (d2 = +inf), so that
// `second` never sees them.  qy / qz are relative to the origin of the run's row (the caller shifts them per run).
__device__ __forceinline__ void scan4_f32(const float4 *__restrict__ tl, uint32_t j, uint32_t e, float qx, float qy, float qz,
                                          int cxp, float hf, F32State &st) {
  const float4 *__restrict__ p = tl + j;
  const float4 c[4] = {p[0], p[1], p[2], p[3]};
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const float dx = qx - (c[u].x + (float)(__float_as_int(c[u].w) - cxp) * hf);
    const float dy = qy - c[u].y;
    const float dz = qz - c[u].z;
    float d2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
    if (u > 0 && j + u >= e) d2 = INFINITY;
    st.second = __builtin_amdgcn_fmed3f(d2, st.best, st.second);  // best <= second always: the median is the new runner-up
    st.bpos = d2 < st.best ? j + u : st.bpos;
    st.best = fminf(st.best, d2);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Cooperative form of the fp32 prefilter search.  The per-lane walk above makes a wavefront iterate the MAXIMUM over its
// 64 lanes of the per-query work (measured on the headline scene: 8.8 four-candidate trips against a mean of 3.7).  Here
// the work of a wavefront's 64 queries is cut into items of four consecutive candidates, the items are queued in LDS
// (wave prefix sum over the lanes' item counts), and the 64 lanes process the queue densely: lane k takes item k, k + 64,
// ... whoever owns it.  An item's result is merged into its owner's (best, second-best) pair with LDS atomics; the
// streaming top-2 rule -- the larger of (incoming, previous best) goes to `second` -- is independent of the order in
// which the atomics land.  What is not decisive in fp32 is searched again exactly in fp64 on the spot.
constexpr int kQCap = 512;            // items per wavefront and round; more rounds when a trip's demand overflows (mean ~260).
                                      // 512 is what four workgroups per CU leave room for in LDS; against 256 the first
                                      // pass of a 10 M registration takes 0.87 instead of 1.03 ms, thirteen point-to-point
                                      // passes 7.64 instead of 7.95 ms (A/B on one box); 128 is slower still
// an item carries 22 bits of four-candidate chunk index: longer runs go to the exact wave search
constexpr uint32_t kRunMaxLen = 1u << 24;
struct CoopWave {
  union {
    uint2 runs[kMaxRuns][O3D_WAVE];   // [slot][lane]: the lane's run list
    VBuf vb;                          // after the search: staging of the estimate's vectors for the matrix cores
  };
  float4 q[O3D_WAVE];                 // query: position inside its cell, w = bits(padded x cell index)
  unsigned long long best[O3D_WAVE];  // bits(d2) << 32 | position in tgt_local
  uint32_t second[O3D_WAVE];          // bits(second-best d2)
  uint32_t items[kQCap];              // owner lane | run slot << 6 | chunk index << 10
};

// A run of candidates: x = first position in tgt_local, y = length << 4 | (dy + 1) | (dz + 1) << 2.
__device__ __forceinline__ uint2 make_run(uint32_t s, uint32_t e, int dy, int dz) {
  return make_uint2(s, ((e - s) << 4) | (uint32_t)((dy + 1) | ((dz + 1) << 2)));
}

// Inclusive prefix sum over the wavefront's 64 lanes in six DPP additions (row shifts inside the rows of 16 lanes, then the
// two row broadcasts of gfx9), instead of six ds_bpermute round trips through the LDS crossbar.  Lanes without a source
// lane -- and rows outside a broadcast's row mask -- take the `old` operand, 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_add(uint32_t v) {
  return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v) {
  v = dpp_add<0x111, 0xf>(v);  // row_shr:1
  v = dpp_add<0x112, 0xf>(v);  // row_shr:2
  v = dpp_add<0x114, 0xf>(v);  // row_shr:4
  v = dpp_add<0x118, 0xf>(v);  // row_shr:8
  v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
  return v;
}

// One phase: every lane brings runs in consecutive slots of its LDS column starting at `first`, `tot` items in all.
__device__ __forceinline__ unsigned coop_phase(CoopWave &wl, int lane, int first, uint32_t tot, const float4 *__restrict__ tl,
                                               float hf) {
  int li = first - 1;
  uint32_t ch = 0, nch = 0;
  unsigned trips = 0;
  for (;;) {
    const uint32_t incl = wave_inclusive_sum(tot);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, O3D_WAVE - 1);
    if (total == 0) break;
    const uint32_t off = incl - tot;
    uint32_t quota = off >= (uint32_t)kQCap ? 0u : min(tot, (uint32_t)kQCap - off);
    tot -= quota;
    for (uint32_t k = off; quota > 0; quota--, k++) {  // divergent, a few instructions per item
      if (ch >= nch) {
        li++;
        ch = 0;
        nch = ((wl.runs[li][lane].y >> 4) + 3u) >> 2;
      }
      wl.items[k] = (uint32_t)lane | ((uint32_t)li << 6) | (ch << 10);
      ch++;
    }
    wave_lds_sync();
    const uint32_t nitems = min(total, (uint32_t)kQCap);
    for (uint32_t t0 = 0; t0 < nitems; t0 += O3D_WAVE) {  // wave-uniform
      trips++;
      const uint32_t t = t0 + lane;
      if (t < nitems) {
        const uint32_t it = wl.items[t];
        const int owner = it & 63;
        const uint2 run = wl.runs[(it >> 6) & 15][owner];
        const uint32_t e = run.x + (run.y >> 4);
        const uint32_t j = run.x + ((it >> 10) << 2);
        const float4 q = wl.q[owner];
        const float qyr = q.y - (float)((int)(run.y & 3u) - 1) * hf, qzr = q.z - (float)((int)((run.y >> 2) & 3u) - 1) * hf;
        F32State st;
        st.best = INFINITY; st.second = INFINITY; st.bpos = j;
        scan4_f32(tl, j, e, q.x, qyr, qzr, __float_as_int(q.w), hf, st);
        const unsigned long long v = ((unsigned long long)__float_as_uint(st.best) << 32) | st.bpos;
        const unsigned long long old = atomicMin(&wl.best[owner], v);
        const unsigned long long disp = old > v ? old : v;
        atomicMin(&wl.second[owner], min((uint32_t)(disp >> 32), __float_as_uint(st.second)));
      }
    }
    wave_lds_sync();
    if (total <= (uint32_t)kQCap) break;  // everything fitted into this round
  }
  return trips;
}

// Exact fp64 search of one query's 3x3x3 block, for the few queries the fp32 prefilter cannot decide (near ties, winner
// within the error bound of the threshold): argmin over (d2, original index) among d2 < thr, with the reference's
// arithmetic.  rowp points at the table entry of the cell left of the query's; the grid's margin makes all nine rows
// valid.  Rare (about 1e-4 of the queries on the headline scene), but a lane scanning ~40 candidates alone would stall its
// wavefront for as many memory latencies, so the WHOLE wavefront serves one undecided query at a time: seven lanes per
// row of the block, lexicographic wave reduction at the end.  Every lane must call it with the same (wave-uniform) query.
__device__ __forceinline__ bool wave_exact_search(const double4 *__restrict__ tgt, const uint32_t *rowp, int row_y, int row_z,
                                                  double x, double y, double z, double thr, int lane, uint32_t &best_pos) {
  const int r = lane / 7, k = lane % 7;
  uint32_t s = 0, e = 0;
  if (r < 9) {
    const uint32_t *rp = rowp + ((r % 3 - 1) * row_y + (r / 3 - 1) * row_z);
    s = rp[0]; e = rp[3];
  }
  int found = 0;
  double bd = 0.0;
  long long bi = 0;
  uint32_t bp = 0;
  for (uint32_t j = s + (uint32_t)k; j < e; j += 7) {
    const double4 p = tgt[j];
    const double dx = x - p.x, dy = y - p.y, dz = z - p.z;
    double d2 = dx * dx;
    d2 = d2 + dy * dy;
    d2 = d2 + dz * dz;
    const long long id = __double_as_longlong(p.w);
    if (d2 < thr && (!found || d2 < bd || (d2 == bd && id < bi))) { found = 1; bd = d2; bi = id; bp = j; }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const int of = __shfl_xor(found, d, O3D_WAVE);
    const double od = __shfl_xor(bd, d, O3D_WAVE);
    const long long oi = __shfl_xor(bi, d, O3D_WAVE);
    const uint32_t op = (uint32_t)__shfl_xor((int)bp, d, O3D_WAVE);
    if (of && (!found || od < bd || (od == bd && oi < bi))) { found = 1; bd = od; bi = oi; bp = op; }
  }
  best_pos = bp;
  return found != 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Certificates.  Next to its match m every query keeps lb, a lower bound on its distance to every target point OTHER
// than m (to every target point when it has no match), valid at the position the bound was last written for.  A pass
// moves the query by |dq|, so lb - |dq| bounds the same distances at the new position; while the new distance to m stays
// below that, m is still the unique nearest point and no search is needed -- the exact fp64 distance to m, the radius test
// and the sums are computed as always, so the result is what a search would give, bit for bit.  Near convergence this is
// nearly every query of nearly every pass.  lb comes out of a search for free: the runner-up among the scanned candidates
// and the pruning reach (every cell the walk left out is farther than that), which PassArgs::gap widens beyond the
// winner's distance so that a bound has room to shrink.  All bounds are fp32, rounded towards "certificate fails" with a
// relative slack of 1e-6 -- nine orders of magnitude above the rounding of the fp64 distances they stand for -- so equal
// distances (ties) can never be certified.
//
// A pass with certificates is two halves -- one launch, k_icp_pass (below), or two:
//   k_icp_cert          streams the source: update in place, certificate, sums of the certified matches on the matrix
//                       cores; the queries it cannot certify go to a work list (one segment per wavefront, in order, so
//                       that every sum is added in the same order run after run)
//   k_icp_search_coop   the cooperative search over the work list only.
// IcpState::skip_cert (set by the estimate that produced the pending update, from the distance that update moves the
// cloud) turns the first launch into a no-op when the bounds cannot hold anyway; the search then runs over everything,
// as it does in the first pass.

// The wavefront's share of the source.  The grid is dealt round-robin over the 8 XCDs (workgroups b and b + 8 share one, and an
// L2), and ALL its wavefronts sweep the processing order together: trip t covers the window of 8 * W consecutive blocks of 64
// positions that starts at block t * 8 * W (W wavefronts per XCD: 4096 blocks, 262144 positions, with the product's grid), and
// inside the window XCD x takes the x-th eighth -- its w-th wavefront block x * W + w.  The wavefronts that run at the same
// time are then neighbours on the Hilbert curve: a candidate line, table row or matched record one of them has fetched is in
// the XCD's L2 when the next one wants it, and the eight XCDs read eight adjoining stretches of every array.  History, all
// measured (profiles/r03_l_*):
//   one contiguous segment per wavefront   the 512 wavefronts of an XCD worked at 512 places far apart; the L2 -- 4 MB --
//                                          could hold none of what neighbouring segments share, and a pass that searches
//                                          everything fetched 2.5 GB for the 1.1 GB it needs;
//   one eighth of the ORDER per XCD        (swept by that XCD's wavefronts together) eight fronts 1.25 M positions apart
//                                          instead of one: the same fabric bytes, but every pass 3-13 % slower than with the
//                                          single window;
//   shorter runs inside the window         (8 ... 64 consecutive blocks per XCD, interleaved) as fast as the eighths of the
//                                          window, but neighbouring runs meet in different L2s: 1.77 GB instead of 1.37 GB
//                                          fetched by a pass that searches everything.
// k_icp_cert and k_icp_search_coop run on the same grid and walk the same blocks; the work list and its counters are per
// wavefront: segment number = chunk * 4 + wavefront, PassArgs::seg_len list places each (a multiple of 64, at least what the
// wavefront's blocks hold).
struct WaveRange {
  unsigned chunk, seg;
  int64_t first, stride, end;  // positions first + t * stride + lane, t = 0, 1, ..., while below end
  int64_t count;               // positions in all
  int64_t list;                // first place of the wavefront's work list
};
__device__ __forceinline__ WaveRange wave_range(int64_t n_src, int64_t seg_len, unsigned gdim) {
  WaveRange r;
  const unsigned xcd = blockIdx.x & 7u, wg = blockIdx.x >> 3;
  const unsigned per_xcd = gdim >> 3;  // workgroups per XCD (the launch grids are multiples of 8)
  r.chunk = xcd * per_xcd + wg;        // partial row of the workgroup
  r.seg = r.chunk * 4u + (threadIdx.x >> 6);
  const int64_t w = (int64_t)wg * 4 + (threadIdx.x >> 6), W = (int64_t)per_xcd * 4;
  r.first = ((int64_t)xcd * W + w) * O3D_WAVE;  // = seg * 64: wavefront number seg takes block seg of every window
  r.stride = W * 8 * O3D_WAVE;
  r.end = n_src;
  if (r.first > r.end) r.first = r.end;
  // (the last block of the cloud may be short)
  const int64_t span = r.end - r.first;
  const int64_t full = span > 0 ? (span - 1) / r.stride : 0;  // trips before the last
  r.count = span > 0 ? full * O3D_WAVE + (span - full * r.stride < O3D_WAVE ? span - full * r.stride : O3D_WAVE) : 0;
  r.list = (int64_t)r.seg * seg_len;
  return r;
}

constexpr size_t kCertLdsBytes = sizeof(VBuf) * (O3D_BLOCK / O3D_WAVE);
static_assert(kCertLdsBytes >= sizeof(double) * ((O3D_BLOCK / O3D_WAVE) * kRedDoubles + O3D_BLOCK / O3D_WAVE), "combine buffer must fit");
template <int EST, bool STREAM, bool TN /* the target has the 48-byte point + normal records (PassArgs::tn) */>
__device__ __forceinline__ void cert_body(const PassArgs &a, unsigned char *lds_raw /* kCertLdsBytes, 16-byte aligned */) {
  if (a.st->done) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const unsigned gdim = (unsigned)a.grid_x;
  if (blockIdx.x >= gdim) return;
  const WaveRange wr = wave_range(a.n_src, a.seg_len, gdim);
  if (a.st->skip_cert) {  // the pending update moves the cloud farther than any bound reaches: everything is searched
    // (a device-scope store like every other partial row: in the one-launch form of the pass the row is read by pass_tail
    // in the same kernel, possibly from another XCD, which would find last registration's row behind a plain store)
    if (tid < O3D_NSUMS) store_partial_row(a.partials, (size_t)a.cert_row0 + wr.chunk, tid, 0.0);
    return;
  }
  VBuf &vb = reinterpret_cast<VBuf *>(lds_raw)[tid >> 6];
  const double cx0 = uniform_f64(a.st->centre[0]), cy0 = uniform_f64(a.st->centre[1]), cz0 = uniform_f64(a.st->centre[2]);
  const int apply = a.st->apply_update;
  double T[12], w[4];
  load_update(a.st, T, w);
  const float thrf = (float)a.thr;
  Sums S = {0.0, 0.0, 0.0, 0u};
  unsigned long long ncert = 0;
  uint32_t nwork = 0;  // wave-uniform
  constexpr bool stream = STREAM;  // (a run-time flag was merged into one path by the compiler, which dropped the hint)
  // The loop is software-pipelined over three trips: while trip t is worked through, the matched records of trip t + 1 are
  // being gathered (their positions arrived a trip ago) and the points, positions and bounds of trip t + 2 are being read.
  // A wavefront otherwise waits out two dependent trips to memory per trip (point -> record) with only the other three
  // wavefronts of its SIMD to cover them.  Every load is issued unconditionally, at a clamped index where the lane has
  // nothing to read: a load inside a branch makes the wait for an OLDER load a wait for everything in flight (one counter,
  // and the compiler cannot know how many younger loads were issued).
  struct StageA {  // what is streamed: the point, the position of its previous match, its bound
    double x, y, z;
    uint32_t pprev;
    float lbv;
  };
  struct StageB {  // what is gathered: the matched record -- point and normal as three 16-byte pieces (p.x p.y | p.z n0 | n1 n2)
    v2f64 q0, q1, q2;
    double4 p4;  // (point-to-point: the 32-byte point alone)
    double4 tc;
    double is;
  };
  const int64_t i_last = wr.end - 1;  // (only used when the wavefront has a trip: wr.first < wr.end)
  auto load_a = [&](int64_t base) {
    StageA s;
    int64_t i = base + lane;
    if (i > i_last) i = i_last;
    if constexpr (stream) {
      s.pprev = __builtin_nontemporal_load(&a.pos[i]);
      s.lbv = __builtin_nontemporal_load(&a.lb[i]);
      s.x = __builtin_nontemporal_load(&a.src[3 * i]); s.y = __builtin_nontemporal_load(&a.src[3 * i + 1]); s.z = __builtin_nontemporal_load(&a.src[3 * i + 2]);
    } else {
      s.pprev = a.pos[i];
      s.lbv = a.lb[i];
      s.x = a.src[3 * i]; s.y = a.src[3 * i + 1]; s.z = a.src[3 * i + 2];
    }
    return s;
  };
  auto load_b = [&](const StageA &s, int64_t base) {
    StageB r = {};
    int64_t i = base + lane;
    if (i > i_last) i = i_last;
    const size_t at = s.pprev != O3D_POS_NONE ? (size_t)s.pprev : 0;  // (no previous match: record 0, not used)
    if (EST != O3D_EST_POINT_TO_POINT && TN) {
      // one 48-byte record instead of two gathers of 32 and 24 bytes, read as exactly three 16-byte loads (as HIP's double2
      // the compiler took the record apart into one 16-byte and four 8-byte loads; the L1 counts 64 accesses for a
      // wavefront's 8-byte load as for its 16-byte load -- TCP_TOTAL_ACCESSES, scripts/micro -- so that was 320 accesses per
      // trip for the 192 needed, of 550 in all: 1.5 % of the pass)
      const v2f64 *rec = reinterpret_cast<const v2f64 *>(a.tn) + 3 * at;
      r.q0 = rec[0]; r.q1 = rec[1]; r.q2 = rec[2];
    } else if (EST == O3D_EST_POINT_TO_POINT) {
      r.p4 = a.tgt[at];
    } else {
      const double4 p = a.tgt[at];
      r.q0 = v2f64{p.x, p.y};
      r.q1 = v2f64{p.z, 0.0};
      r.q2 = v2f64{0.0, 0.0};
      if (EST != O3D_EST_POINT_TO_POINT) {
        r.q1.y = a.nrm[3 * at]; r.q2.x = a.nrm[3 * at + 1]; r.q2.y = a.nrm[3 * at + 2];
      }
    }
    if (EST == O3D_EST_COLORED_ICP) { r.tc = a.tcol[at]; r.is = a.sint[i]; }
    return r;
  };
  StageA a0, a1, a2;
  StageB r0, r1;
  if (wr.first < wr.end) {
    a0 = load_a(wr.first);
    a1 = load_a(wr.first + wr.stride);
    r0 = load_b(a0, wr.first);
  }
  PASS_MARK(1)
  LOOP_T_DECL
  for (int64_t base = wr.first; base < wr.end; base += wr.stride) {
    r1 = load_b(a1, base + wr.stride);
    a2 = load_a(base + 2 * wr.stride);
    LOOP_T(0)  // waited for the positions read a trip ago; gathers and stream loads issued
    const int64_t i = base + lane;
    const bool valid = i < wr.end;
    double x = a0.x, y = a0.y, z = a0.z;
    const uint32_t pprev = valid ? a0.pprev : O3D_POS_NONE;
    const float lbv = valid ? a0.lbv : 0.f;
    float moved = 0.f;
    const bool has_prev = valid && pprev != O3D_POS_NONE;
    // (p.w, the index, is not needed: a certified entry is not rewritten)
    const double4 p = EST == O3D_EST_POINT_TO_POINT ? r0.p4 : make_double4(r0.q0.x, r0.q0.y, r0.q1.x, 0.0), tc = r0.tc;
    const double n0 = r0.q1.y, n1 = r0.q2.x, n2 = r0.q2.y, is = r0.is;
    a0 = a1; a1 = a2; r0 = r1;
    // A point the update leaves where it is, bit for bit, is not written back, and its bound stays as it stands (it was
    // written for exactly this position).  Point-to-plane ICP converges quadratically: on the headline scene the updates are
    // 6e-3, 3e-6, 4e-10 and from then on ~1e-16 -- below half an ulp of most coordinates -- so from the fifth pass on
    // 70-90 % of the points and 60-85 % of the 64-point blocks do not move at all (scripts/update_decay.py,
    // scripts/static_points.py), and a pass writes 28 bytes less for each of them: 0.29 -> 0.04 GB written per pass.
    // (x != x for a non-finite point: written and never certified, as before.)
    bool stays = true;
    if (valid && apply) {
      const double xo = x, yo = y, zo = z;
      apply_update(T, w, x, y, z);
      stays = x == xo && y == yo && z == zo;
#if O3D_COUNT_STATIC  // (scripts/static_points.py: counter 7 = points that stayed + wavefront blocks without a moving point << 40)
      {
        const unsigned long long m_all = __ballot(stays);
        if (lane == 0 && a.stats) atomicAdd(&a.ctr[7], (unsigned long long)__popcll(m_all) + (m_all == ~0ull ? (1ull << 40) : 0ull));
      }
#endif
      if (!stays) {
        if constexpr (stream) {
          __builtin_nontemporal_store(x, &a.src[3 * i]); __builtin_nontemporal_store(y, &a.src[3 * i + 1]); __builtin_nontemporal_store(z, &a.src[3 * i + 2]);
        } else {
          a.src[3 * i] = x; a.src[3 * i + 1] = y; a.src[3 * i + 2] = z;
        }
        const double mx = x - xo, my = y - yo, mz = z - zo;
        moved = sqrtf((float)(mx * mx + my * my + mz * mz)) * (1.0f + 1e-6f);  // NaN for a non-finite point: never certified
      }
    }
    LOOP_T(1)  // update applied (waited for the point read two trips ago), point written back
    // what the stored bound is still worth at the new position (<= 0, or NaN: nothing)
    const float lbc = stays ? lbv : lbv * (1.0f - 1e-6f) - moved;
    double d2 = 0.0;
    if (has_prev) {
      const double ddx = x - p.x, ddy = y - p.y, ddz = z - p.z;
      d2 = ddx * ddx;
      d2 = d2 + ddy * ddy;
      d2 = d2 + ddz * ddz;
    }
    LOOP_T(2)  // waited for the record gathered a trip ago; distance
    bool certified = false;
    if (valid && lbc > 0.f) {
      const float lb2 = lbc * lbc * (1.0f - 1e-6f);
      certified = has_prev ? (float)d2 * (1.0f + 1e-6f) < lb2   // m is strictly nearer than everything else
                           : lb2 > thrf * (1.0f + 1e-6f);        // nothing within the radius
    }
    // A point the pending update leaves where it is, bit for bit, needs no bound at all: its entry is the exact result of the
    // pass before at this very position (every pass leaves exact results behind: searched or certified), so it stands.  Near
    // convergence this takes most of the points no bound can certify -- ties and near-ties, which a search resolves in fp64
    // one at a time -- off the work list: they rest like everybody else.
    if (valid && apply && stays) certified = true;
    // A certified query keeps its correspondence entry as it is -- the entry already names m (or nothing, without a match):
    // the search that found m wrote it, and no later pass touched it -- so the entry is not rewritten here (4 of the 124
    // bytes a certified point costs).  The one case that would change it, m still the nearest point but now beyond the
    // radius (no correspondence, Registration.cpp:64-66), is left to the search, which clears entry and match together.
    if (certified && has_prev && !(d2 < a.thr)) certified = false;
    const bool use = certified && has_prev;
    if (certified && !stays) {
      if constexpr (stream) __builtin_nontemporal_store(lbc, &a.lb[i]);
      else a.lb[i] = lbc;
    }
    const unsigned long long need = __ballot(valid && !certified);
    if (valid && !certified) a.work[wr.list + nwork + __popcll(need & ((1ull << lane) - 1ull))] = (uint32_t)i;
    nwork += (uint32_t)__popcll(need);
    if (a.stats) ncert += __popcll(__ballot(certified));
    LOOP_T(3)  // certificate test, bound and work list written
    push_sums<EST>(vb, lane, S, use, x, y, z, p, n0, n1, n2, d2, cx0, cy0, cz0, tc, is, a.slg, a.slp);
    LOOP_T(4)  // sums
  }
  LOOP_T_STORE
  PASS_MARK(2)
  if (lane == 0) {
    a.work_count[wr.seg] = nwork;
    if (a.stats && ncert) atomicAdd(&a.ctr[O3D_CTR_CERTIFIED], ncert);
  }
  __syncthreads();  // every wavefront is done with its staging buffer
  PASS_MARK(3)
  mfma_reduce_store<EST>(S, reinterpret_cast<double *>(lds_raw), a.partials, a.cert_row0 + wr.chunk);
}

// The arguments of a pass's kernels as they sit in device memory for the lock-step form of the loop (o3d_launch_pass_gang):
// blockIdx.y picks the registration's record.
struct PassLaunch {
  PassArgs a;
  GridParams gp;
};

template <int EST>
__global__ __launch_bounds__(O3D_BLOCK, 4) void k_icp_cert(PassArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[kCertLdsBytes];
  if (a.tn) cert_body<EST, false, true>(a, lds_raw);
  else cert_body<EST, false, false>(a, lds_raw);
}
template <int EST>
__global__ __launch_bounds__(O3D_BLOCK, 4) void k_icp_cert_s(PassArgs a) {  // PassArgs::stream
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[kCertLdsBytes];
  if (a.tn) cert_body<EST, true, true>(a, lds_raw);
  else cert_body<EST, true, false>(a, lds_raw);
}
// ... and for several registrations in lock step (o3d_registration_icp_batch): blockIdx.y names the registration
template <int EST>
__global__ __launch_bounds__(O3D_BLOCK, 4) void k_icp_cert_batch(const PassLaunch *__restrict__ pl) {
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[kCertLdsBytes];
  if (pl[blockIdx.y].a.tn) cert_body<EST, false, true>(pl[blockIdx.y].a, lds_raw);
  else cert_body<EST, false, false>(pl[blockIdx.y].a, lds_raw);
}
template <int EST>
__global__ __launch_bounds__(O3D_BLOCK, 4) void k_icp_cert_batch_s(const PassLaunch *__restrict__ pl) {
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[kCertLdsBytes];
  if (pl[blockIdx.y].a.tn) cert_body<EST, true, true>(pl[blockIdx.y].a, lds_raw);
  else cert_body<EST, true, false>(pl[blockIdx.y].a, lds_raw);
}

// The cooperative search (see above CoopWave).  PassArgs::listed: the queries are the work list k_icp_cert left behind
// (already moved by the pending update), unless IcpState::skip_cert says that k_icp_cert stood aside.
// Requires gp.margin (empty outer cell layers): a query with all 26 neighbour cells in the table needs no bounds checks,
// any other query is at least a cell size away from every target point and has no correspondence.
template <int EST, bool TN>
__device__ __forceinline__ void search_coop_body(const PassArgs &a, const GridParams &gp, CoopWave *wls /* one per wavefront */) {
  const unsigned gdim = (unsigned)a.grid_x;
  if (blockIdx.x >= gdim || a.st->done) return;
  static_assert(sizeof(CoopWave) * (O3D_BLOCK / O3D_WAVE) >= sizeof(double) * ((O3D_BLOCK / O3D_WAVE) * kRedDoubles + O3D_BLOCK / O3D_WAVE), "the final combine reuses this LDS");
  const int tid = threadIdx.x, lane = tid & 63;
  CoopWave &wl = wls[tid >> 6];
  const double cx0 = uniform_f64(a.st->centre[0]), cy0 = uniform_f64(a.st->centre[1]), cz0 = uniform_f64(a.st->centre[2]);
  Sums S = {0.0, 0.0, 0.0, 0u};
  const bool listed = a.listed && !a.st->skip_cert;
  const int apply = listed ? 0 : a.st->apply_update;
  double T[12], w[4];
  load_update(a.st, T, w);
  const WaveRange wr = wave_range(a.n_src, a.seg_len, gdim);
  const int64_t nq = listed ? (int64_t)a.work_count[wr.seg] : wr.count;  // wave-uniform
  const float hf = (float)gp.h;
  const float err = kF32Err * hf * hf;
  const float thrf = (float)a.thr;
  const float margin = 1e-5f * hf;
  // (no seeds at all after a large step, IcpState::skip_seed: folded into the threshold, which costs no register)
  const float seed_max = (a.seed && !a.st->skip_seed) ? kSeedMax * hf * hf : -1.0f;
  const float cap = hf * (1.0f - 1e-6f);  // every point outside the 3x3x3 block, or around a query outside the grid interior
  const float capsq = cap * cap;
  const float4 *__restrict__ tl = a.tgt_local;
  const uint32_t *__restrict__ cs = a.cell_start;
  const int row_y = o3d_row_stride_y(gp), row_z = o3d_row_stride_z(gp);  // table strides of one cell in y / z (wave-uniform)

  // The fp32 winner is confirmed in fp64 with the arithmetic of the exact search (it then IS the exact result: every
  // other candidate is farther by more than the fp32 error bound), and its terms of the next estimate go to the matrix
  // cores.  What is not decisive in fp32 is searched again in fp64 on the spot (wave_exact_search).  The gathers of the
  // matched target point and normal are issued at the end of a trip and consumed in the next one, behind that trip's own
  // source loads, so that their latency is not exposed.
  bool use_p = false;  // carried from the previous trip
  int64_t i_p = 0;
  double4 p_p = make_double4(0.0, 0.0, 0.0, 0.0), tc_p = make_double4(0.0, 0.0, 0.0, 0.0);
  double n0_p = 0.0, n1_p = 0.0, n2_p = 0.0, xs = 0.0, ys = 0.0, zs = 0.0, is_p = 0.0;
  auto consume = [&]() {
    double d2 = 0.0;
    const bool had_p = use_p;  // the lane brought a candidate: its correspondence entry is written here
    if (use_p) {
      const double ddx = xs - p_p.x, ddy = ys - p_p.y, ddz = zs - p_p.z;
      d2 = ddx * ddx;
      d2 = d2 + ddy * ddy;
      d2 = d2 + ddz * ddz;
    }
    // A confirmation that fails would mean that the fp32 error bound does not hold (winners near the threshold were
    // resolved exactly before they got here).  It is counted; o3d_icp_finish / o3d_icp_fetch then fail loudly instead of
    // returning a result that might not be exact.
    if (use_p && !(d2 < a.thr)) {
      atomicAdd(&a.ctr[O3D_CTR_VIOLATIONS], 1ull);
      use_p = false;
    }
    if (had_p) a.corr[i_p] = use_p ? (int32_t)__double_as_longlong(p_p.w) : O3D_CORR_NONE;
    // begins and ends with a wavefront LDS sync
    push_sums<EST>(wl.vb, lane, S, use_p, xs, ys, zs, p_p, n0_p, n1_p, n2_p, d2, cx0, cy0, cz0, tc_p, is_p, a.slg, a.slp);
  };

  for (int64_t k0 = 0; k0 < nq; k0 += O3D_WAVE) {  // wave-uniform trip count
#if O3D_SRCH_PROF
    long long t_prev = clock64();
#endif
    const bool valid = k0 + lane < nq;
    int64_t i = wr.first + (k0 / O3D_WAVE) * wr.stride + lane;
    if (listed && valid) i = a.work[wr.list + k0 + lane];
    double x = 0, y = 0, z = 0;
    uint32_t pprev = O3D_POS_NONE;
    if (valid) {
      x = a.src_in[3 * i]; y = a.src_in[3 * i + 1]; z = a.src_in[3 * i + 2];
      if (seed_max > 0.f) pprev = a.pos[i];
    }
    if (valid && apply) {
      apply_update(T, w, x, y, z);
      a.src[3 * i] = x; a.src[3 * i + 1] = y; a.src[3 * i + 2] = z;
    }
    const int cx = o3d_cell_coord(x, gp.gmin[0], gp.inv_h, gp.nx);
    const int cy = o3d_cell_coord(y, gp.gmin[1], gp.inv_h, gp.ny);
    const int cz = o3d_cell_coord(z, gp.gmin[2], gp.inv_h, gp.nz);
    const bool inside = valid && cx >= 0 && cx < gp.nx && cy >= 1 && cy < gp.ny - 1 && cz >= 1 && cz < gp.nz - 1;
    int32_t verdict = O3D_CORR_NONE;
    float qx = 0, qy = 0, qz = 0;
    uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    const uint32_t *rowp = cs;
    if (inside) {
      qx = (float)((x - gp.gmin[0]) - (double)cx * gp.h);
      qy = (float)((y - gp.gmin[1]) - (double)cy * gp.h);
      qz = (float)((z - gp.gmin[2]) - (double)cz * gp.h);
      rowp = cs + o3d_row_index(gp, cx, cy, cz);
      const CellQuad qc = *reinterpret_cast<const CellQuad *>(rowp);
      s0 = qc.v[0]; s1 = qc.v[1]; s2 = qc.v[2]; s3 = qc.v[3];
    }
    // Seed: the target point this query matched in the previous pass.  Any target point bounds the nearest-neighbour
    // distance from above; near convergence the previous match is (nearly) the nearest, so its exact distance prunes
    // the block from the start and the centre-cell phase is skipped.  Only a bound: every cell it leaves is scanned and
    // the winner comes out of the scan as always.
    const bool has_seed = inside && pprev != O3D_POS_NONE;
    double4 ps = make_double4(0.0, 0.0, 0.0, 0.0);
    if (has_seed) ps = a.tgt[pprev];
    SRCH_T(0)
    if (k0 != 0) consume();  // the previous trip's matches, behind this trip's source, table and seed loads
    SRCH_T(1)
    bool too_long = s3 - s0 >= kRunMaxLen;
    float seedb = INFINITY;
    if (has_seed) {
      const double ddx = x - ps.x, ddy = y - ps.y, ddz = z - ps.z;
      seedb = (float)(ddx * ddx + ddy * ddy + ddz * ddz) * 1.000001f + err;  // >= the exact distance, in fp32
    }
    const bool seeded = seedb <= seed_max && !too_long;
    // First bound for a query without a usable seed: the first four candidates of its own cell, scanned by the lane itself --
    // one address, four loads, no queue (most centre cells hold no more; what they hold beyond goes to the queue below
    // with the neighbouring cells).  A cooperative phase of its own for the centre cells cost a prefix sum, a queue and two
    // rounds of LDS atomics per trip for 1.5 dense rounds of work.
    F32State st0;
    st0.best = INFINITY; st0.second = INFINITY; st0.bpos = 0;
    const bool own = inside && !too_long && !seeded && s1 < s2;
    if (own) scan4_f32(tl, s1, s2, qx, qy, qz, cx + 1, hf, st0);
    wl.q[lane] = make_float4(qx, qy, qz, __int_as_float(cx + 1));
    wl.best[lane] = ((unsigned long long)__float_as_uint(st0.best) << 32) | st0.bpos;
    wl.second[lane] = __float_as_uint(st0.second);
    // Dense cells (a search radius of several point spacings): where the wavefront's own cells hold a round of work beyond
    // their first chunks, those are scanned cooperatively before the listing, so that the bound the neighbouring cells are
    // pruned with is the whole cell's; otherwise they ride along with the neighbours' runs.
    SRCH_T(2)
    const uint32_t rem = (own && s2 - s1 > 4u) ? (s2 - s1 - 1u) >> 2 : 0u;
    const bool centre_phase = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_sum(rem), O3D_WAVE - 1) >= kCentrePhaseMin;
    unsigned trips = 0;
    if (centre_phase) {  // wave-uniform
      wl.runs[0][lane] = make_run(s1 + 4u, rem ? s2 : s1 + 4u, 0, 0);
      wave_lds_sync();
      trips = coop_phase(wl, lane, 0, rem, tl, hf);
    }
    const float best1 = seeded ? seedb : (centre_phase ? __uint_as_float((uint32_t)(wl.best[lane] >> 32)) : st0.best);
    // phase 2: the runs that survive the face-distance pruning against the centre result.  The table rows are loaded
    // only now and only where the bound lets them matter (one masked 16-byte load per needed row, all in flight
    // together), instead of eight loads per query up front.
    int nruns = 0;
    uint32_t tot = 0;
    float reach = 0.f;
    if (inside && !too_long) {
      float t;
      t = fmaxf(qx - margin, 0.f); const float axm = t * t;
      t = fmaxf(hf - qx - margin, 0.f); const float axp = t * t;
      t = fmaxf(qy - margin, 0.f); const float aym = t * t;
      t = fmaxf(hf - qy - margin, 0.f); const float ayp = t * t;
      t = fmaxf(qz - margin, 0.f); const float azm = t * t;
      t = fmaxf(hf - qz - margin, 0.f); const float azp = t * t;
      // everything nearer than the best so far can still win; gap beyond that is scanned for the certificates' sake
      const float bnd = fminf(best1, thrf + err);
      reach = bnd + 4.0f * err + a.gap * (2.0f * sqrtf(bnd) + a.gap);  // (sqrt(bnd) + gap)^2, gap = 0 without certificates
      CellQuad rowq[9];
#pragma unroll
      for (int r = 0; r < 9; r++) {
        rowq[r].v[0] = rowq[r].v[1] = rowq[r].v[2] = rowq[r].v[3] = 0;
        if (r == 4) continue;
        const int dy = r % 3 - 1, dz = r / 3 - 1;
        const float byz = (dz < 0 ? azm : (dz > 0 ? azp : 0.f)) + (dy < 0 ? aym : (dy > 0 ? ayp : 0.f));
        if (byz <= reach) rowq[r] = *reinterpret_cast<const CellQuad *>(rowp + (dy * row_y + dz * row_z));
      }
      if (seeded) {  // centre row in one run: centre cell, and the left / right cell where the bound reaches them
        const uint32_t s = axm <= reach ? s0 : s1, e = axp <= reach ? s3 : s2;
        if (s < e) { wl.runs[nruns][lane] = make_run(s, e, 0, 0); tot += (e - s + 3u) >> 2; nruns++; }
      } else {
        if (rem && !centre_phase) { wl.runs[nruns][lane] = make_run(s1 + 4u, s2, 0, 0); tot += rem; nruns++; }  // the rest of the own cell
        if (axm <= reach && s0 < s1) { wl.runs[nruns][lane] = make_run(s0, s1, 0, 0); tot += (s1 - s0 + 3u) >> 2; nruns++; }
        if (axp <= reach && s2 < s3) { wl.runs[nruns][lane] = make_run(s2, s3, 0, 0); tot += (s3 - s2 + 3u) >> 2; nruns++; }
      }
#pragma unroll
      for (int r = 0; r < 9; r++) {
        if (r == 4) continue;
        const int dy = r % 3 - 1, dz = r / 3 - 1;
        const float byz = (dz < 0 ? azm : (dz > 0 ? azp : 0.f)) + (dy < 0 ? aym : (dy > 0 ? ayp : 0.f));
        // rows that were not loaded read as empty (all zero); corner reach widens the run to the left / right cell
        const uint32_t s = (axm + byz <= reach) ? rowq[r].v[0] : rowq[r].v[1];
        const uint32_t e = (axp + byz <= reach) ? rowq[r].v[3] : rowq[r].v[2];
        if (s < e) {
          too_long = too_long || e - s >= kRunMaxLen;
          wl.runs[nruns][lane] = make_run(s, e, dy, dz);
          tot += (e - s + 3u) >> 2;
          nruns++;
        }
      }
      if (too_long) tot = 0;
    }
    wave_lds_sync();
    SRCH_T(3)
    trips += coop_phase(wl, lane, 0, tot, tl, hf);
    SRCH_T(4)
    uint32_t bpos = O3D_POS_NONE;
    float lbn = (valid && !inside) ? cap : 0.f;  // the bound left for the certificates; outside: a cell from every target point
    if (inside) {
      const unsigned long long bb = wl.best[lane];
      const float best = __uint_as_float((uint32_t)(bb >> 32)), second = __uint_as_float(wl.second[lane]);
      bpos = (uint32_t)bb;
      const bool near_tie = (second - best) <= 2.0f * err;  // inf - inf = NaN -> false
      const bool near_thr = fabsf(best - thrf) <= err + 1e-6f * thrf;
      if (too_long || (best <= thrf + err && (near_tie || near_thr))) verdict = O3D_CORR_PENDING;
      else if (best < thrf) verdict = O3D_CORR_CANDIDATE;
      if (verdict != O3D_CORR_PENDING) {
        // decided in fp32.  Every other point (every point, without a match) is farther than: the runner-up of the scan
        // (the winner itself without a match), the reach the unscanned cells lie beyond, the edge of the 3x3x3 block
        const float other = (verdict == O3D_CORR_CANDIDATE ? second : best) - err;
        const float lb2 = fminf(fminf(other, reach), capsq) * (1.0f - 1e-5f);
        lbn = lb2 > 0.f ? sqrtf(lb2) * (1.0f - 1e-6f) : 0.f;
      }
    }
    const unsigned long long undecided = __ballot(verdict == O3D_CORR_PENDING);
    for (unsigned long long m = undecided; m; m &= m - 1) {  // wave-uniform, rare: one undecided query at a time
      const int srcl = __ffsll((long long)m) - 1;
      const double qx64 = __shfl(x, srcl, O3D_WAVE), qy64 = __shfl(y, srcl, O3D_WAVE), qz64 = __shfl(z, srcl, O3D_WAVE);
      const unsigned long long rpb = (unsigned long long)(uintptr_t)rowp;
      const unsigned long long rq = ((unsigned long long)(uint32_t)__shfl((int)(rpb >> 32), srcl, O3D_WAVE) << 32) |
                                    (uint32_t)__shfl((int)(uint32_t)rpb, srcl, O3D_WAVE);
      uint32_t ep = 0;
      const bool ok = wave_exact_search(a.tgt, (const uint32_t *)(uintptr_t)rq, row_y, row_z, qx64, qy64, qz64, a.thr, lane, ep);
      if (lane == srcl) { verdict = ok ? O3D_CORR_CANDIDATE : O3D_CORR_NONE; bpos = ep; }  // lbn stays 0: searched again next pass
    }
    SRCH_T(5)
    // issue the gathers for this trip's candidates; they are consumed in the next trip (or after the loop)
    use_p = verdict == O3D_CORR_CANDIDATE;
    i_p = i;
    xs = x; ys = y; zs = z;
    if (use_p) {
      if (TN) {  // (never with point-to-point: no normals)
        const double2 r0 = a.tn[3 * (size_t)bpos], r1 = a.tn[3 * (size_t)bpos + 1], r2 = a.tn[3 * (size_t)bpos + 2];
        // the original index from its own dense array: neighbouring winners share its sectors (out of tgt_sorted, 8 bytes at
        // a stride of 32, every index cost a sector of its own)
        p_p = make_double4(r0.x, r0.y, r1.x, __longlong_as_double((long long)a.tidx[bpos]));
        n0_p = r1.y; n1_p = r2.x; n2_p = r2.y;
      } else {
        p_p = a.tgt[bpos];
        if (EST != O3D_EST_POINT_TO_POINT) {
          n0_p = a.nrm[3 * (size_t)bpos]; n1_p = a.nrm[3 * (size_t)bpos + 1]; n2_p = a.nrm[3 * (size_t)bpos + 2];
        }
      }
      if (EST == O3D_EST_COLORED_ICP) { tc_p = a.tcol[bpos]; is_p = a.sint[i]; }
    }
    if (a.stats) {
      const unsigned long long np = __popcll(undecided), nv = __popcll(__ballot(valid));
      unsigned long long rsum = (unsigned long long)nruns, isum = tot + (own ? 1u : 0u);
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        rsum += __shfl_xor(rsum, d, O3D_WAVE);
        isum += __shfl_xor(isum, d, O3D_WAVE);
      }
      if (lane == 0) {
        atomicAdd(&a.ctr[0], 1ull);
        atomicAdd(&a.ctr[1], (unsigned long long)trips);
        atomicAdd(&a.ctr[2], rsum);
        atomicAdd(&a.ctr[3], np);
        atomicAdd(&a.ctr[4], isum);
        atomicAdd(&a.ctr[5], nv);
      }
    }
    if (valid) {
      if (!use_p) a.corr[i] = verdict;  // a candidate's entry is written when it is confirmed
      a.pos[i] = use_p ? bpos : O3D_POS_NONE;
      a.lb[i] = lbn;
    }
    wave_lds_sync();  // the next iteration rewrites the wavefront's LDS state
    SRCH_T(6)
  }
  if (nq > 0) consume();
  __syncthreads();
  mfma_reduce_store<EST>(S, reinterpret_cast<double *>(wls), a.partials, wr.chunk);
  pass_tail(a, reinterpret_cast<double *>(wls), wr.chunk, gdim);
}

template <int EST, bool TN>
__global__ __launch_bounds__(O3D_BLOCK, 4) void k_icp_search_coop(PassArgs a, GridParams gp) {
  __shared__ CoopWave wls[O3D_BLOCK / O3D_WAVE];
  search_coop_body<EST, TN>(a, gp, wls);
}
template <int EST, bool TN>
__global__ __launch_bounds__(O3D_BLOCK, 4) void k_icp_search_coop_batch(const PassLaunch *__restrict__ pl) {
  __shared__ CoopWave wls[O3D_BLOCK / O3D_WAVE];
  search_coop_body<EST, TN>(pl[blockIdx.y].a, pl[blockIdx.y].gp, wls);
}
static_assert(sizeof(CoopWave) * (O3D_BLOCK / O3D_WAVE) >= kCertLdsBytes, "the certificate half stages its vectors in the search's LDS");
// A pass with certificates as ONE launch: every wavefront certifies its segment (cert_body) and goes straight on to search
// what it could not certify (search_coop_body over its own work list) -- the two kernels walk the same segments on the same
// grid, so nothing of the second half waits for another workgroup's first half; the partial rows, their order and the end
// of the pass (pass_tail) are those of the two launches, and so is every bit of the result
// (test_one_launch_pass_equals_two_launches).  What it saves is the second launch with its prologue.  Both halves share the
// workgroup's LDS (the search's queues are the larger).  Two things the two-launch form got for free from the kernel
// boundary: a workgroup's certificate row must be complete -- and visible device-wide -- when pass_tail adds its group's
// rows, so the certificate rows start at a multiple of 32 (plan_pass: a row's group is row & 31, and a group is added when
// its own last workgroup has arrived) and EVERY store to a partial row is a device-scope store, the zero row of a
// certificate half that stands aside included (as a plain store it sat in its XCD's L2 while another XCD's pass_tail read
// the previous registration's row from memory: wrong by one registration's sums from the third registration on a context on).
template <int EST, bool TN>
__device__ __forceinline__ void pass_body(const PassArgs &a, const GridParams &gp, CoopWave *wls) {
  PASS_MARK(0)
#if O3D_PASS_PROF
  if ((threadIdx.x & 63) == 0)  // where the wavefront runs: HW_ID (wave, SIMD, CU, shader array, shader engine) and XCC_ID
    g_pass_t[blockIdx.x * 4 + (threadIdx.x >> 6)][6] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                                                       ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
#endif
  cert_body<EST, true, TN>(a, reinterpret_cast<unsigned char *>(wls));
  PASS_MARK(4)
  __syncthreads();  // (every return of cert_body is taken by the whole workgroup)
  search_coop_body<EST, TN>(a, gp, wls);
  PASS_MARK(5)
}
template <int EST, bool TN>
__global__ __launch_bounds__(O3D_BLOCK, 4) void k_icp_pass(PassArgs a, GridParams gp) {
  __shared__ CoopWave wls[O3D_BLOCK / O3D_WAVE];
  pass_body<EST, TN>(a, gp, wls);
}
template <int EST, bool TN>
__global__ __launch_bounds__(O3D_BLOCK, 4) void k_icp_pass_batch(const PassLaunch *__restrict__ pl) {
  __shared__ CoopWave wls[O3D_BLOCK / O3D_WAVE];
  pass_body<EST, TN>(pl[blockIdx.y].a, pl[blockIdx.y].gp, wls);
}

// Fixed-order reduction of the workgroup partials (run-to-run deterministic).  1024 threads: thread (g, k) sums
// partials[b][k] for b = g, g + 32, ... (coalesced over k, 32 rows in flight), the 32 group sums are then added in
// group order.
#define O3D_FIN_THREADS 1024
__device__ __forceinline__ void finalize_sums(const double *__restrict__ partials, int nparts, double (*s)[O3D_NSUMS]) {
  const int k = threadIdx.x & 31, g = threadIdx.x >> 5;
  double acc = 0.0;  // 32 independent row loads per step (the order of the additions is fixed by the indices alone)
  for (int b0 = g; b0 < nparts; b0 += 32 * 32) {
    double v[32];
#pragma unroll
    for (int u = 0; u < 32; u++) {
      const int b = b0 + 32 * u;
      v[u] = b < nparts ? partials[(size_t)b * O3D_NSUMS + k] : 0.0;
    }
#pragma unroll
    for (int w = 1; w < 32; w <<= 1)
#pragma unroll
      for (int u = 0; u < 32; u += 2 * w) v[u] += v[u + w];
    acc += v[0];
  }
  s[g][k] = acc;
  __syncthreads();
  if (threadIdx.x < O3D_NSUMS) {
    double t = s[0][k];
#pragma unroll
    for (int q = 1; q < 32; q++) t += s[q][k];
    s[0][k] = t;
  }
  __syncthreads();
}

__global__ __launch_bounds__(O3D_FIN_THREADS) void k_icp_finalize(const double *__restrict__ partials, int nparts,
                                                                   const IcpState *__restrict__ st, double *__restrict__ sums) {
  if (st->done) return;
  __shared__ double s[32][O3D_NSUMS];
  finalize_sums(partials, nparts, s);
  if (threadIdx.x < O3D_NSUMS) sums[threadIdx.x] = s[0][threadIdx.x];
}

// ------------------------------------------------------------------ small dense algebra, one lane
#define TM(T, r, c) ((T)[(c) * 4 + (r)])

__device__ __forceinline__ void mat4_identity(double *T) {
  for (int i = 0; i < 16; i++) T[i] = 0.0;
  T[0] = T[5] = T[10] = T[15] = 1.0;
}

__device__ __forceinline__ void mat4_mul(const double *A, const double *B, double *C) {
  double R[16];
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 4; r++) {
      double a = TM(A, r, 0) * TM(B, 0, c);
      a = a + TM(A, r, 1) * TM(B, 1, c);
      a = a + TM(A, r, 2) * TM(B, 2, c);
      a = a + TM(A, r, 3) * TM(B, 3, c);
      R[c * 4 + r] = a;
    }
  for (int i = 0; i < 16; i++) C[i] = R[i];
}

// x = A^-1 b by LDL^T with symmetric pivoting on the largest |diagonal| (Eigen::LDLT semantics, utility/Eigen.cpp:85):
// lower triangle only, zero pivot -> zero component.  Everything lives in registers: the step index K is a template
// parameter and the pivot row P, the one dynamic index, is turned into a static one by a switch over its few possible
// values (arrays indexed dynamically would go to scratch memory, a microsecond per dependent access on the one lane that
// runs the estimate; an earlier version kept them in LDS).
#define MR(i, j) M[(i) * 6 + (j)]
template <int K, int P>
__device__ __forceinline__ void ldlt_swap(double (&M)[36]) {
  double t;
#pragma unroll
  for (int j = 0; j < K; j++) { t = MR(K, j); MR(K, j) = MR(P, j); MR(P, j) = t; }
#pragma unroll
  for (int i = P + 1; i < 6; i++) { t = MR(i, K); MR(i, K) = MR(i, P); MR(i, P) = t; }
#pragma unroll
  for (int i = K + 1; i < P; i++) { t = MR(i, K); MR(i, K) = MR(P, i); MR(P, i) = t; }
}
template <int K>
__device__ __forceinline__ void ldlt_swap_dyn(double (&M)[36], int p) {
  switch (p) {
    case 1: if constexpr (K < 1) ldlt_swap<K, 1>(M); break;
    case 2: if constexpr (K < 2) ldlt_swap<K, 2>(M); break;
    case 3: if constexpr (K < 3) ldlt_swap<K, 3>(M); break;
    case 4: if constexpr (K < 4) ldlt_swap<K, 4>(M); break;
    case 5: if constexpr (K < 5) ldlt_swap<K, 5>(M); break;
    default: break;
  }
  // the diagonal pair by selects on the values: written as one more swap inside the cases, the compiler merges the cases
  // into one swap at a dynamic address and the diagonal goes to scratch memory after all
  const double dk = MR(K, K);
  double dp = dk;
#pragma unroll
  for (int i = K + 1; i < 6; i++) dp = p == i ? MR(i, i) : dp;
#pragma unroll
  for (int i = K + 1; i < 6; i++) MR(i, i) = p == i ? dk : MR(i, i);
  MR(K, K) = dp;
}
template <int K>
__device__ __forceinline__ void vec_swap_dyn(double (&y)[6], int p) {  // by selects, for the same reason
  const double yk = y[K];
  double yp = yk;
#pragma unroll
  for (int i = K + 1; i < 6; i++) yp = p == i ? y[i] : yp;
#pragma unroll
  for (int i = K + 1; i < 6; i++) y[i] = p == i ? yk : y[i];
  y[K] = yp;
}
template <int K>
__device__ __forceinline__ void ldlt_step(double (&M)[36], int (&tr)[6]) {
  int p = K;
  double big = fabs(MR(K, K));
#pragma unroll
  for (int i = K + 1; i < 6; i++)
    if (fabs(MR(i, i)) > big) { big = fabs(MR(i, i)); p = i; }
  tr[K] = p;
  if (p != K) ldlt_swap_dyn<K>(M, p);
  if constexpr (K > 0) {
    double temp[K];
#pragma unroll
    for (int j = 0; j < K; j++) temp[j] = MR(j, j) * MR(K, j);
    double sdot = 0;
#pragma unroll
    for (int j = 0; j < K; j++) sdot += MR(K, j) * temp[j];
    MR(K, K) -= sdot;
#pragma unroll
    for (int i = K + 1; i < 6; i++) {
      double a = 0;
#pragma unroll
      for (int j = 0; j < K; j++) a += MR(i, j) * temp[j];
      MR(i, K) -= a;
    }
  }
  const double akk = MR(K, K);
  if (fabs(akk) > 0) {
#pragma unroll
    for (int i = K + 1; i < 6; i++) MR(i, K) /= akk;
  }
}
__device__ void ldlt_solve6(const double *A, const double *b, double *x) {
  double M[36];
  int tr[6];
#pragma unroll
  for (int i = 0; i < 36; i++) M[i] = A[i];
  ldlt_step<0>(M, tr); ldlt_step<1>(M, tr); ldlt_step<2>(M, tr);
  ldlt_step<3>(M, tr); ldlt_step<4>(M, tr); ldlt_step<5>(M, tr);
  double y[6];
#pragma unroll
  for (int i = 0; i < 6; i++) y[i] = b[i];
  if (tr[0] != 0) vec_swap_dyn<0>(y, tr[0]);
  if (tr[1] != 1) vec_swap_dyn<1>(y, tr[1]);
  if (tr[2] != 2) vec_swap_dyn<2>(y, tr[2]);
  if (tr[3] != 3) vec_swap_dyn<3>(y, tr[3]);
  if (tr[4] != 4) vec_swap_dyn<4>(y, tr[4]);
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = 0; j < i; j++) y[i] -= MR(i, j) * y[j];
#pragma unroll
  for (int i = 0; i < 6; i++) {
    if (fabs(MR(i, i)) > 2.2250738585072014e-308) y[i] /= MR(i, i);
    else y[i] = 0;
  }
#pragma unroll
  for (int i = 5; i >= 0; i--)
#pragma unroll
    for (int j = i + 1; j < 6; j++) y[i] -= MR(j, i) * y[j];
  if (tr[4] != 4) vec_swap_dyn<4>(y, tr[4]);
  if (tr[3] != 3) vec_swap_dyn<3>(y, tr[3]);
  if (tr[2] != 2) vec_swap_dyn<2>(y, tr[2]);
  if (tr[1] != 1) vec_swap_dyn<1>(y, tr[1]);
  if (tr[0] != 0) vec_swap_dyn<0>(y, tr[0]);
#pragma unroll
  for (int i = 0; i < 6; i++) x[i] = y[i];
}
#undef MR

__device__ __forceinline__ void vec6_to_mat4(const double *v, double *T) {
  double sx, cx, sy, cy, sz, cz;  // one argument reduction per angle
  sincos(v[0], &sx, &cx);
  sincos(v[1], &sy, &cy);
  sincos(v[2], &sz, &cz);
  mat4_identity(T);
  TM(T, 0, 0) = cy * cz;  TM(T, 0, 1) = sx * sy * cz - cx * sz;  TM(T, 0, 2) = cx * sy * cz + sx * sz;
  TM(T, 1, 0) = cy * sz;  TM(T, 1, 1) = sx * sy * sz + cx * cz;  TM(T, 1, 2) = cx * sy * sz - sx * cz;
  TM(T, 2, 0) = -sy;      TM(T, 2, 1) = sx * cy;                 TM(T, 2, 2) = cx * cy;
  TM(T, 0, 3) = v[3]; TM(T, 1, 3) = v[4]; TM(T, 2, 3) = v[5];
}

__device__ __forceinline__ void cross3(const double *a, const double *b, double *c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ double det3(const double *M) {
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// A = U diag(S) V^T (row-major 3x3), one-sided Jacobi; singular values descending.  Every index is static (the rotations
// are instantiated per column pair, the ordering swaps whole columns by selects): the arrays stay in registers -- with
// loop-variable indices they lived in scratch memory, a dependent memory access per element on the one lane that runs the
// estimate (5 us of the 27 us of a 100 k-point point-to-point iteration).  The arithmetic is unchanged.
template <int P, int Q>
__device__ __forceinline__ void jacobi_rotate(double (&W)[9], double (&Vm)[9], double &off) {
  double alpha = 0, beta = 0, gamma = 0;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    alpha += W[i * 3 + P] * W[i * 3 + P];
    beta += W[i * 3 + Q] * W[i * 3 + Q];
    gamma += W[i * 3 + P] * W[i * 3 + Q];
  }
  if (gamma == 0.0) return;
  if (gamma * gamma <= 1e-32 * alpha * beta) return;
  off = fmax(off, fabs(gamma) / sqrt(alpha * beta));
  const double zeta = (beta - alpha) / (2.0 * gamma);
  const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
  const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const double wp = W[i * 3 + P], wq = W[i * 3 + Q];
    W[i * 3 + P] = c * wp - sn * wq;
    W[i * 3 + Q] = sn * wp + c * wq;
    const double vp = Vm[i * 3 + P], vq = Vm[i * 3 + Q];
    Vm[i * 3 + P] = c * vp - sn * vq;
    Vm[i * 3 + Q] = sn * vp + c * vq;
  }
}

// columns A and B of W and Vm and their singular values change places when sv[B] > sv[A]
template <int A, int B>
__device__ __forceinline__ void svd_order(double (&W)[9], double (&Vm)[9], double (&sv)[3]) {
  const bool sw = sv[B] > sv[A];
  double t;
  t = sv[A]; sv[A] = sw ? sv[B] : t; sv[B] = sw ? t : sv[B];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    t = W[i * 3 + A]; W[i * 3 + A] = sw ? W[i * 3 + B] : t; W[i * 3 + B] = sw ? t : W[i * 3 + B];
    t = Vm[i * 3 + A]; Vm[i * 3 + A] = sw ? Vm[i * 3 + B] : t; Vm[i * 3 + B] = sw ? t : Vm[i * 3 + B];
  }
}

__device__ __forceinline__ void svd3(const double *A, double *U, double *S, double *V) {
  double W[9], Vm[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
#pragma unroll
  for (int i = 0; i < 9; i++) W[i] = A[i];
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0;
    jacobi_rotate<0, 1>(W, Vm, off);
    jacobi_rotate<0, 2>(W, Vm, off);
    jacobi_rotate<1, 2>(W, Vm, off);
    if (off < 1e-15) break;
  }
  double sv[3];
#pragma unroll
  for (int j = 0; j < 3; j++) sv[j] = sqrt(W[j] * W[j] + W[3 + j] * W[3 + j] + W[6 + j] * W[6 + j]);
  // the selection sort of the column order (largest first, a swap only on strictly greater), applied to the columns
  svd_order<0, 1>(W, Vm, sv);
  svd_order<0, 2>(W, Vm, sv);
  svd_order<1, 2>(W, Vm, sv);
  double Uc[3][3];
  bool have[3];
#pragma unroll
  for (int j = 0; j < 3; j++) {
    S[j] = sv[j];
#pragma unroll
    for (int i = 0; i < 3; i++) V[i * 3 + j] = Vm[i * 3 + j];
    have[j] = sv[j] > 1e-300 && sv[j] > sv[0] * 1e-14;
#pragma unroll
    for (int i = 0; i < 3; i++) Uc[j][i] = have[j] ? W[i * 3 + j] / sv[j] : 0.0;
  }
  if (!have[0]) { Uc[0][0] = 1; Uc[0][1] = 0; Uc[0][2] = 0; }
  if (!have[1]) {
    const double a0 = fabs(Uc[0][0]), a1 = fabs(Uc[0][1]), a2 = fabs(Uc[0][2]);
    const int m = a0 < a1 ? (a0 < a2 ? 0 : 2) : (a1 < a2 ? 1 : 2);
    const double e[3] = {m == 0 ? 1.0 : 0.0, m == 1 ? 1.0 : 0.0, m == 2 ? 1.0 : 0.0};
    double c[3];
    cross3(Uc[0], e, c);
    const double nrm = sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
#pragma unroll
    for (int i = 0; i < 3; i++) Uc[1][i] = c[i] / nrm;
  }
  if (!have[2]) {
    double c[3];
    cross3(Uc[0], Uc[1], c);
    const double nrm = sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
#pragma unroll
    for (int i = 0; i < 3; i++) Uc[2][i] = c[i] / nrm;
  }
#pragma unroll
  for (int j = 0; j < 3; j++)
#pragma unroll
    for (int i = 0; i < 3; i++) U[i * 3 + j] = Uc[j][i];
}

// Eigen::umeyama from single-pass moments about a fixed origin c (TransformationEstimation.cpp:59):
// mu_s = c + S_s/n, mu_d = c + S_d/n, sigma = S_ds/n - (S_d/n)(S_s/n)^T, var_s = S_ss/n - |S_s/n|^2.
__device__ __forceinline__ void umeyama_from_sums(const double *q, const double *centre, int with_scaling, double *T) {
  const double n = q[O3D_SUM_NC];
  const double inv_n = 1.0 / n;
  double ms[3], md[3];
  for (int d = 0; d < 3; d++) { ms[d] = q[d] * inv_n; md[d] = q[3 + d] * inv_n; }
  double sig[9];
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) sig[a * 3 + b] = q[6 + a * 3 + b] * inv_n - md[a] * ms[b];
  const double var = q[15] * inv_n - (ms[0] * ms[0] + ms[1] * ms[1] + ms[2] * ms[2]);
  double U[9], S[3], V[9];
  svd3(sig, U, S, V);
  double sgn[3] = {1, 1, 1};
  if (det3(U) * det3(V) < 0) sgn[2] = -1;
  double R[9];
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) {
      double acc = 0;
      for (int k = 0; k < 3; k++) acc += U[a * 3 + k] * sgn[k] * V[b * 3 + k];
      R[a * 3 + b] = acc;
    }
  double c = 1.0;
  if (with_scaling) c = 1.0 / var * (S[0] * sgn[0] + S[1] * sgn[1] + S[2] * sgn[2]);
  mat4_identity(T);
  for (int a = 0; a < 3; a++) {
    const double mus[3] = {ms[0] + centre[0], ms[1] + centre[1], ms[2] + centre[2]};
    const double rm = R[a * 3] * mus[0] + R[a * 3 + 1] * mus[1] + R[a * 3 + 2] * mus[2];
    TM(T, a, 3) = (md[a] + centre[a]) - c * rm;
    for (int b = 0; b < 3; b++) TM(T, a, b) = c * R[a * 3 + b];
  }
}

// One lane: everything between two passes of Registration.cpp:155-166.
// One row of the per-pass history, entry e by lane e (52 lanes in parallel instead of 52 stores on the solving lane):
// {fitness, rmse, n_corr, sum d2, T before this pass's update (16), the 32 sums}.  Reads the state BEFORE solve_body
// changes it: the callers put a workgroup barrier between the two.
__device__ __forceinline__ void write_log_row(const IcpState *st, const double *q, double *__restrict__ log, int log_rows, int e) {
  const int pass = st->passes;
  if (!log || pass >= log_rows || e >= O3D_LOG_STRIDE) return;
  const double nc = q[O3D_SUM_NC];
  double v;
  if (e == 0) v = nc > 0.0 ? nc / st->n_src_global : 0.0;
  else if (e == 1) v = nc > 0.0 ? sqrt(q[O3D_SUM_D2] / nc) : 0.0;
  else if (e == 2) v = nc;
  else if (e == 3) v = q[O3D_SUM_D2];
  else if (e < 20) v = st->T[e - 4];
  else v = q[e - 20];
  log[(size_t)pass * O3D_LOG_STRIDE + e] = v;
}

// How far the pending update moves a point of the cloud at most (points within `radius` of `centre`): when that exceeds
// what the certificates' slack can absorb, the next pass does not try them (a prediction for speed only -- a certificate
// that is tried is checked point by point against the distance the point really moved).
__device__ __forceinline__ void update_skips_certificates(IcpState *st, const double *upd) {
  double f2 = 0.0, c[3];
  for (int r = 0; r < 3; r++) {
    double v = TM(upd, r, 3) - st->centre[r];
    for (int k = 0; k < 3; k++) {
      const double e = TM(upd, r, k) - (r == k ? 1.0 : 0.0);
      f2 += e * e;
      v += TM(upd, r, k) * st->centre[k];
    }
    c[r] = v;
  }
  const double moved = sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]) + sqrt(f2) * st->radius;
  st->skip_cert = !(moved <= st->cert_skip_above);  // NaN: skip
  // The previous match as the search's first bound pays while the cloud moves by a fraction of the point spacing; after a
  // large step it prunes worse than the centre-cell phase it replaces (second pass of the 10 M headline: 0.785 ms with
  // seeds, 0.64 ms without).
  st->skip_seed = !(moved <= st->seed_skip_above);
}

__device__ __forceinline__ void solve_body(IcpState *__restrict__ st, const double *q, int has_normals, double *__restrict__ log, int log_rows) {
  const double nc = q[O3D_SUM_NC];
  double fit = 0.0, rmse = 0.0;
  if (nc > 0.0) {  // Registration.cpp:82-89
    fit = nc / st->n_src_global;
    rmse = sqrt(q[O3D_SUM_D2] / nc);
  }
  const int pass = st->passes;
  const double pf = st->fitness, pr = st->rmse;
  st->fitness = fit; st->rmse = rmse; st->n_corr = nc; st->sum_d2 = q[O3D_SUM_D2];
  st->passes = pass + 1;
  if (pass >= 1) {
    st->iters = pass;
    if (fabs(pf - fit) < st->rel_fitness && fabs(pr - rmse) < st->rel_rmse) { st->done = 1; return; }  // :162-165
  }
  if (pass >= st->max_iter) { st->done = 1; return; }

  PASS_MARK(16)
  double upd[16];
  mat4_identity(upd);
  if (nc > 0.0) {
    if (st->estimation != O3D_EST_POINT_TO_POINT) {  // point-to-plane and coloured ICP: 6x6 normal equations
      if (has_normals) {
        double A[36], b[6], x[6];
        int k = 0;
        for (int a = 0; a < 6; a++)
          for (int c = a; c < 6; c++) { A[a * 6 + c] = q[k]; A[c * 6 + a] = q[k]; k++; }
        for (int a = 0; a < 6; a++) b[a] = -q[21 + a];
        ldlt_solve6(A, b, x);
        PASS_MARK(17)
        vec6_to_mat4(x, upd);
      }
    } else {
      umeyama_from_sums(q, st->centre, st->with_scaling, upd);
    }
  }
  PASS_MARK(18)
  for (int i = 0; i < 16; i++) st->update[i] = upd[i];
  mat4_mul(upd, st->T, st->T);
  st->apply_update = 1;
  PASS_MARK(19)
  update_skips_certificates(st, upd);
  PASS_MARK(20)
}

// The loop state is staged in LDS for the one lane that runs the estimate (dozens of dependent global accesses otherwise)
// and written back by the workgroup.
static_assert(sizeof(IcpState) % 4 == 0, "IcpState is copied word by word");
__device__ __forceinline__ void state_copy(uint32_t *dst, const uint32_t *src) {
  for (unsigned k = threadIdx.x; k < sizeof(IcpState) / 4; k += blockDim.x) dst[k] = src[k];
}

// Sharded loop: launched with one wavefront; sums the ranks' rows in rank order, then the estimate.
__global__ void k_icp_solve(IcpState *__restrict__ st, const double *__restrict__ all_sums, int n_ranks, int has_normals,
                            double *__restrict__ log, int log_rows) {
  if (blockIdx.x != 0) return;
  if (st->done) return;
  __shared__ IcpState ls;
  __shared__ double q[O3D_NSUMS];
  state_copy(reinterpret_cast<uint32_t *>(&ls), reinterpret_cast<const uint32_t *>(st));
  for (int k = threadIdx.x; k < O3D_NSUMS; k += blockDim.x) {
    double v = all_sums[k];
    for (int r = 1; r < n_ranks; r++) v += all_sums[(size_t)r * O3D_NSUMS + k];
    q[k] = v;
  }
  __syncthreads();
  write_log_row(&ls, q, log, log_rows, threadIdx.x);
  __syncthreads();
  if (threadIdx.x == 0) solve_body(&ls, q, has_normals, log, log_rows);
  __syncthreads();
  state_copy(reinterpret_cast<uint32_t *>(st), reinterpret_cast<const uint32_t *>(&ls));
}

// Unsharded loop: partial reduction and estimate in one launch.
__global__ __launch_bounds__(O3D_FIN_THREADS) void k_icp_finalize_solve(const double *__restrict__ partials, int nparts,
                                                                         IcpState *__restrict__ st, double *__restrict__ sums,
                                                                         int has_normals, double *__restrict__ log, int log_rows) {
  if (st->done) return;
  __shared__ double s[32][O3D_NSUMS];
  __shared__ IcpState ls;
  state_copy(reinterpret_cast<uint32_t *>(&ls), reinterpret_cast<const uint32_t *>(st));
  finalize_sums(partials, nparts, s);
  if (threadIdx.x < O3D_NSUMS) sums[threadIdx.x] = s[0][threadIdx.x];
  if (threadIdx.x >= O3D_WAVE && threadIdx.x < 2 * O3D_WAVE) write_log_row(&ls, s[0], log, log_rows, threadIdx.x - O3D_WAVE);
  __syncthreads();
  if (threadIdx.x == 0) solve_body(&ls, s[0], has_normals, log, log_rows);
  __syncthreads();
  state_copy(reinterpret_cast<uint32_t *>(st), reinterpret_cast<const uint32_t *>(&ls));
}

// The end of a pass without a launch of its own.  The workgroups of the pass's last kernel are dealt into 32 groups by
// partial row (chunk & 31); the LAST workgroup of a group to store its row adds the group's rows, and the last GROUP to
// finish adds the 32 group rows and, in the device-resident loop, runs the estimate.  "Last" is decided by an arrival
// counter; nobody waits for anybody.  WHO does the adding varies from run to run, WHAT is added in which order does not:
// group g owns rows g + 32 u + 1024 j and adds them exactly as finalize_sums does (pairwise over u, then over j; then the
// groups in order), so the sums equal k_icp_finalize's bit for bit.
// Visibility between workgroups on different XCDs (one L2 each) without device-wide fences -- a release fence writes back
// the whole L2 and an acquire fence invalidates it, tens of microseconds over a pass's workgroups (measured: 28 -> 55 us per
// iteration at 100 k points): the rows are stored and loaded with device-scope accesses (write-through / read-through,
// sc1), a workgroup arrives only after its stores have been acknowledged (s_waitcnt), and the loads of the last one depend
// on the counter value it drew.
__device__ __forceinline__ bool last_to_arrive(unsigned *ticket, unsigned expected) {
  __shared__ int s_last;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);  // this wavefront's stores have been acknowledged
  __syncthreads();
  if (threadIdx.x == 0) {
    const bool last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == expected - 1u;
    if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // nobody else touches it before the next launch
    s_last = last;
  }
  __syncthreads();
  return s_last != 0;
}

__device__ __forceinline__ void pass_tail(const PassArgs &a, double *lds, unsigned chunk, unsigned gdim) {
  if (!a.tail) return;
  const int tid = threadIdx.x, k = tid & 31, uq = tid >> 5;
  const unsigned g = chunk & 31u;
  const unsigned ngroups = gdim < 32u ? gdim : 32u;
  if (!last_to_arrive(a.tickets + g, (gdim - g + 31u) / 32u)) return;
  double (*s8)[O3D_NSUMS] = reinterpret_cast<double (*)[O3D_NSUMS]>(lds);
  double acc = 0.0;
  for (int b0 = (int)g; b0 < a.nparts; b0 += 32 * 32) {  // workgroup-uniform
    double v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int b = b0 + 32 * (4 * uq + u);
      v[u] = b < a.nparts ? load_device_coherent(a.partials + (size_t)b * O3D_NSUMS + k) : 0.0;
    }
    v[0] += v[1]; v[2] += v[3]; v[0] += v[2];
    __syncthreads();
    s8[uq][k] = v[0];
    __syncthreads();
    if (tid < O3D_NSUMS) {
      double t[8];
#pragma unroll
      for (int q = 0; q < 8; q++) t[q] = s8[q][k];
      t[0] += t[1]; t[2] += t[3]; t[4] += t[5]; t[6] += t[7];
      t[0] += t[2]; t[4] += t[6];
      t[0] += t[4];
      acc += t[0];
    }
  }
  if (tid < O3D_NSUMS) store_device_coherent(a.gsums + g * O3D_NSUMS + k, acc);
  if (!last_to_arrive(a.tickets + 32, ngroups)) return;
  PASS_MARK(14)  // (only the workgroup that ends the pass gets here)
  __shared__ double q[O3D_NSUMS];
  if (tid < O3D_NSUMS) {
    double v[32];
#pragma unroll
    for (int r = 0; r < 32; r++) {
      if ((unsigned)r < ngroups) {
        v[r] = load_device_coherent(a.gsums + r * O3D_NSUMS + k);
      } else {
        // a grid of fewer than 32 workgroups: group r has no member; its rows -- at most r and r + 32, both rows of
        // k_icp_cert, complete before this kernel began -- are added here, again as finalize_sums would
        const double x0 = r < a.nparts ? load_device_coherent(a.partials + (size_t)r * O3D_NSUMS + k) : 0.0;
        const double x1 = r + 32 < a.nparts ? load_device_coherent(a.partials + (size_t)(r + 32) * O3D_NSUMS + k) : 0.0;
        v[r] = 0.0 + (x0 + x1);
      }
    }
    double t = v[0];
#pragma unroll
    for (int r = 1; r < 32; r++) t += v[r];
    q[k] = t;
    a.sums_out[k] = t;
  }
  if (a.tail < 2) return;
  __shared__ IcpState ls;
  state_copy(reinterpret_cast<uint32_t *>(&ls), reinterpret_cast<const uint32_t *>(a.st_rw));
  __syncthreads();
  PASS_MARK(15)
  if (tid >= O3D_WAVE && tid < 2 * O3D_WAVE) write_log_row(&ls, q, a.log, a.log_rows, tid - O3D_WAVE);
  __syncthreads();
  if (tid == 0) solve_body(&ls, q, a.has_normals, a.log, a.log_rows);
  __syncthreads();
  PASS_MARK(7)
  state_copy(reinterpret_cast<uint32_t *>(a.st_rw), reinterpret_cast<const uint32_t *>(&ls));
}

__global__ __launch_bounds__(O3D_BLOCK) void k_scatter_corr(const int32_t *__restrict__ corr_sorted, const uint32_t *__restrict__ perm,
                                                             int64_t n, int32_t *__restrict__ out) {
  for (int64_t p = (int64_t)blockIdx.x * O3D_BLOCK + threadIdx.x; p < n; p += (int64_t)gridDim.x * O3D_BLOCK)
    out[perm[p]] = corr_sorted[p];
}

// ------------------------------------------------------------------ companions of the loop

// Same fixed-order reduction as k_icp_finalize, without the loop's done flag (used outside the ICP loop).
__global__ __launch_bounds__(O3D_FIN_THREADS) void k_finalize_plain(const double *__restrict__ partials, int nparts,
                                                                     double *__restrict__ sums) {
  __shared__ double s[32][O3D_NSUMS];
  finalize_sums(partials, nparts, s);
  if (threadIdx.x < O3D_NSUMS) sums[threadIdx.x] = s[0][threadIdx.x];
}

// GetInformationMatrixFromPointClouds, Registration.cpp:326-346: the three rank-one terms per correspondence only
// involve the matched target point, so GTG is a function of ten moments {x, y, z, xx, yy, zz, xy, xz, yz, 1}.
__global__ __launch_bounds__(O3D_BLOCK) void k_info_accumulate(PassArgs a) {
  __shared__ double red[O3D_BLOCK / O3D_WAVE][O3D_NSUMS];
  double acc[30];
#pragma unroll
  for (int k = 0; k < 30; k++) acc[k] = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * O3D_BLOCK + threadIdx.x; i < a.n_src; i += (int64_t)gridDim.x * O3D_BLOCK) {
    if (a.corr[i] < 0) continue;
    const double4 p = a.tgt[a.pos[i]];
    acc[0] += p.x; acc[1] += p.y; acc[2] += p.z;
    acc[3] += p.x * p.x; acc[4] += p.y * p.y; acc[5] += p.z * p.z;
    acc[6] += p.x * p.y; acc[7] += p.x * p.z; acc[8] += p.y * p.z;
    acc[9] += 1.0;
  }
  block_reduce_store(acc, red, a.partials, blockIdx.x);
}

// TransformationEstimation::ComputeTransformation / ComputeRMSE over a caller-given CorrespondenceSet
// (TransformationEstimation.cpp:38-98): pairs = int32 (source index, target index), any order, repeats allowed.
struct PairArgs {
  const double *src;
  int64_t n_src;
  const double *tgt;
  const double *nrm;
  int64_t n_tgt;
  const int32_t *pairs;
  int64_t n_pairs;
  double *partials;
  int *bad;  // number of pairs with an index out of range
};

// Fixed origin of the Umeyama moments: the target point of the first pair (any value is valid; it only has to be
// the same in the accumulation and in the estimate).
__device__ __forceinline__ void pairs_centre(const PairArgs &a, double (&c)[3]) {
  c[0] = c[1] = c[2] = 0.0;
  if (a.n_pairs > 0) {
    const int32_t j = a.pairs[1];
    if (j >= 0 && j < a.n_tgt)
      for (int d = 0; d < 3; d++) {
        const double v = a.tgt[3 * (size_t)j + d];
        c[d] = (v == v && fabs(v) < 1e300) ? v : 0.0;
      }
  }
}

template <int EST>
__global__ __launch_bounds__(O3D_BLOCK) void k_pairs_accumulate(PairArgs a) {
  __shared__ double red[O3D_BLOCK / O3D_WAVE][O3D_NSUMS];
  double c0[3];
  pairs_centre(a, c0);
  double acc[30];
#pragma unroll
  for (int k = 0; k < 30; k++) acc[k] = 0.0;
  int bad = 0;
  for (int64_t k = (int64_t)blockIdx.x * O3D_BLOCK + threadIdx.x; k < a.n_pairs; k += (int64_t)gridDim.x * O3D_BLOCK) {
    const int32_t i = a.pairs[2 * k], j = a.pairs[2 * k + 1];
    if (i < 0 || i >= a.n_src || j < 0 || j >= a.n_tgt) { bad++; continue; }
    const double x = a.src[3 * (size_t)i], y = a.src[3 * (size_t)i + 1], z = a.src[3 * (size_t)i + 2];
    const double4 p = make_double4(a.tgt[3 * (size_t)j], a.tgt[3 * (size_t)j + 1], a.tgt[3 * (size_t)j + 2], 0.0);
    Best b;
    const double dx = x - p.x, dy = y - p.y, dz = z - p.z;
    double d2 = dx * dx;
    d2 = d2 + dy * dy;
    d2 = d2 + dz * dz;
    b.d2 = d2; b.idx = j; b.pos = (uint32_t)j;
    double n0 = 0.0, n1 = 0.0, n2 = 0.0;
    if (EST == O3D_EST_POINT_TO_PLANE) {
      n0 = a.nrm[3 * (size_t)j]; n1 = a.nrm[3 * (size_t)j + 1]; n2 = a.nrm[3 * (size_t)j + 2];
    }
    accumulate<EST>(acc, x, y, z, b, p, n0, n1, n2, c0[0], c0[1], c0[2]);
  }
  if (bad) atomicAdd(a.bad, bad);
  block_reduce_store(acc, red, a.partials, blockIdx.x);
}

// out[0..15] = ComputeTransformation (column-major), out[16] = point-to-point RMSE, out[17] = point-to-plane RMSE,
// out[18] = number of pairs used.
__global__ __launch_bounds__(O3D_FIN_THREADS) void k_pairs_estimate(PairArgs a, int nparts, int estimation, int with_scaling,
                                                                     double *__restrict__ out, int write_bad) {
  __shared__ double s[32][O3D_NSUMS];
  finalize_sums(a.partials, nparts, s);
  if (threadIdx.x != 0) return;
  double q[O3D_NSUMS];
  for (int k = 0; k < O3D_NSUMS; k++) q[k] = s[0][k];
  double c0[3];
  pairs_centre(a, c0);
  double T[16];
  mat4_identity(T);
  const double nc = q[O3D_SUM_NC];
  if (nc > 0.0) {
    if (estimation == O3D_EST_POINT_TO_PLANE) {
      if (a.nrm) {
        double A[36], b[6], x[6];
        int k = 0;
        for (int r = 0; r < 6; r++)
          for (int c = r; c < 6; c++) { A[r * 6 + c] = q[k]; A[c * 6 + r] = q[k]; k++; }
        for (int r = 0; r < 6; r++) b[r] = -q[21 + r];
        ldlt_solve6(A, b, x);
        vec6_to_mat4(x, T);
      }
    } else {
      umeyama_from_sums(q, c0, with_scaling, T);
    }
  }
  for (int i = 0; i < 16; i++) out[i] = T[i];
  out[16] = nc > 0.0 ? sqrt(q[O3D_SUM_D2] / nc) : 0.0;
  out[17] = (nc > 0.0 && a.nrm) ? sqrt(q[27] / nc) : 0.0;
  out[18] = nc;
  if (write_bad) out[19] = (double)*a.bad;  // o3d_compute_transformation_noalloc: the caller has no other way to see it
}

// RegistrationRANSACBasedOnFeatureMatching, Registration.cpp:250-268, one lane per candidate correspondence set: the
// checkers of CorrespondenceChecker.cpp:38-82 and estimation.ComputeTransformation over the ransac_n sampled pairs.
__global__ __launch_bounds__(O3D_WAVE) void k_ransac_hypotheses(RansacArgs a) {
  const int h = blockIdx.x * O3D_WAVE + threadIdx.x;
  if (h >= a.n_hyp) return;
  const int32_t *c = a.corres + 2 * (size_t)a.ransac_n * h;
  double T[16];
  mat4_identity(T);
  int ok = 1;
  for (int j = 0; j < a.ransac_n; j++)
    if (c[2 * j] < 0 || c[2 * j] >= a.n_src || c[2 * j + 1] < 0 || c[2 * j + 1] >= a.n_tgt) ok = 0;
  if (ok && a.use_edge) {  // CorrespondenceCheckerBasedOnEdgeLength::Check, :38-51
    for (int i = 0; i < a.ransac_n && ok; i++)
      for (int j = i + 1; j < a.ransac_n; j++) {
        const double *si = a.src + 3 * (size_t)c[2 * i], *sj = a.src + 3 * (size_t)c[2 * j];
        const double *ti = a.tgt + 3 * (size_t)c[2 * i + 1], *tj = a.tgt + 3 * (size_t)c[2 * j + 1];
        const double sx = si[0] - sj[0], sy = si[1] - sj[1], sz = si[2] - sj[2];
        const double tx = ti[0] - tj[0], ty = ti[1] - tj[1], tz = ti[2] - tj[2];
        const double ds = sqrt((sx * sx + sy * sy) + sz * sz), dt = sqrt((tx * tx + ty * ty) + tz * tz);
        if (ds < dt * a.edge_sim || dt < ds * a.edge_sim) { ok = 0; break; }
      }
  }
  if (ok) {  // ComputeTransformation over the sampled pairs (an identity for point-to-plane without target normals, :76-77)
    const double *t0 = a.tgt + 3 * (size_t)c[1];
    const double c0[3] = {t0[0], t0[1], t0[2]};
    double acc[30];
    for (int k = 0; k < 30; k++) acc[k] = 0.0;
    const bool plane = a.estimation == O3D_EST_POINT_TO_PLANE;
    for (int j = 0; j < a.ransac_n; j++) {
      const double *sp = a.src + 3 * (size_t)c[2 * j], *tp = a.tgt + 3 * (size_t)c[2 * j + 1];
      const double4 p = make_double4(tp[0], tp[1], tp[2], 0.0);
      Best b;
      const double dx = sp[0] - p.x, dy = sp[1] - p.y, dz = sp[2] - p.z;
      double d2 = dx * dx;
      d2 = d2 + dy * dy;
      d2 = d2 + dz * dz;
      b.d2 = d2; b.idx = c[2 * j + 1]; b.pos = 0;
      if (plane) {
        if (a.tnrm) {
          const double *np = a.tnrm + 3 * (size_t)c[2 * j + 1];
          accumulate<O3D_EST_POINT_TO_PLANE>(acc, sp[0], sp[1], sp[2], b, p, np[0], np[1], np[2], c0[0], c0[1], c0[2]);
        }
      } else {
        accumulate<O3D_EST_POINT_TO_POINT>(acc, sp[0], sp[1], sp[2], b, p, 0.0, 0.0, 0.0, c0[0], c0[1], c0[2]);
      }
    }
    double q[O3D_NSUMS];
    for (int k = 0; k < 30; k++) q[k] = acc[k];
    q[30] = q[31] = 0.0;
    if (q[O3D_SUM_NC] > 0.0) {
      if (plane) {
        double A[36], bb[6], x[6];
        int k = 0;
        for (int r = 0; r < 6; r++)
          for (int cc = r; cc < 6; cc++) { A[r * 6 + cc] = q[k]; A[cc * 6 + r] = q[k]; k++; }
        for (int r = 0; r < 6; r++) bb[r] = -q[21 + r];
        ldlt_solve6(A, bb, x);
        vec6_to_mat4(x, T);
      } else {
        umeyama_from_sums(q, c0, a.with_scaling, T);
      }
    }
  }
  if (ok && a.use_dist) {  // CorrespondenceCheckerBasedOnDistance::Check, :53-63
    for (int j = 0; j < a.ransac_n; j++) {
      const double *sp = a.src + 3 * (size_t)c[2 * j], *tp = a.tgt + 3 * (size_t)c[2 * j + 1];
      double d[3];
      for (int r = 0; r < 3; r++) {
        double v = TM(T, r, 0) * sp[0];
        v = v + TM(T, r, 1) * sp[1];
        v = v + TM(T, r, 2) * sp[2];
        v = v + TM(T, r, 3);
        d[r] = tp[r] - v;
      }
      if (sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]) > a.dist_thr) { ok = 0; break; }
    }
  }
  if (ok && a.use_normal && a.snrm && a.tnrm) {  // CorrespondenceCheckerBasedOnNormal::Check, :65-82
    for (int j = 0; j < a.ransac_n; j++) {
      const double *sn = a.snrm + 3 * (size_t)c[2 * j], *tn = a.tnrm + 3 * (size_t)c[2 * j + 1];
      double dot = 0.0;
      for (int r = 0; r < 3; r++) {
        double v = TM(T, r, 0) * sn[0];
        v = v + TM(T, r, 1) * sn[1];
        v = v + TM(T, r, 2) * sn[2];
        dot = (r == 0) ? tn[0] * v : dot + tn[r] * v;
      }
      if (dot < a.cos_normal) { ok = 0; break; }
    }
  }
  for (int i = 0; i < 16; i++) a.out_T[16 * (size_t)h + i] = T[i];
  a.out_ok[h] = ok;
}

__global__ __launch_bounds__(O3D_BLOCK) void k_gather_tcol(const double4 *__restrict__ tgt_sorted, const double *__restrict__ col,
                                                            const double *__restrict__ grad, int64_t n, double4 *__restrict__ out) {
  for (int64_t p = (int64_t)blockIdx.x * O3D_BLOCK + threadIdx.x; p < n; p += (int64_t)gridDim.x * O3D_BLOCK) {
    const size_t i = (size_t)__double_as_longlong(tgt_sorted[p].w);
    const double it = (col[3 * i] + col[3 * i + 1] + col[3 * i + 2]) / 3.0;
    out[p] = make_double4(it, grad[3 * i], grad[3 * i + 1], grad[3 * i + 2]);
  }
}

__global__ __launch_bounds__(O3D_BLOCK) void k_gather_sint(const uint32_t *__restrict__ perm, const double *__restrict__ col, int64_t n,
                                                            double *__restrict__ out) {
  for (int64_t p = (int64_t)blockIdx.x * O3D_BLOCK + threadIdx.x; p < n; p += (int64_t)gridDim.x * O3D_BLOCK) {
    const size_t i = perm[p];
    out[p] = (col[3 * i] + col[3 * i + 1] + col[3 * i + 2]) / 3.0;
  }
}

// Caller-computed update (estimations that stay on the host, e.g. ColoredICP or a user subclass): what
// Registration.cpp:157-159 does with the result of the virtual ComputeTransformation.
struct Mat16 {
  double m[16];
};
__global__ void k_icp_set_update(IcpState *__restrict__ st, Mat16 u) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  for (int i = 0; i < 16; i++) st->update[i] = u.m[i];
  mat4_mul(u.m, st->T, st->T);
  st->apply_update = 1;
  update_skips_certificates(st, u.m);
  st->iters = st->iters + 1;
}

// Hypothesis mode (RANSAC validation, Registration.cpp:269-271): the next pass transforms the untouched source by T.
__global__ void k_icp_set_hypothesis(IcpState *__restrict__ st, Mat16 u) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  for (int i = 0; i < 16; i++) { st->update[i] = u.m[i]; st->T[i] = u.m[i]; }
  st->apply_update = 1;
  st->skip_seed = 1;  // hypotheses are unrelated: the matches under the previous one bound nothing useful
}

}  // namespace

int o3d_launch_set_hypothesis(o3d_ctx *ctx, const double T[16]) {
  Mat16 u;
  for (int i = 0; i < 16; i++) u.m[i] = T[i];
  hipLaunchKernelGGL(k_icp_set_hypothesis, dim3(1), dim3(64), 0, ctx->stream, ctx->state, u);
  O3D_HIP(ctx, hipGetLastError());
  return O3D_OK;
}

int o3d_launch_info(o3d_ctx *ctx, double *dev_sums_out) {
  PassArgs pa{};
  pa.n_src = ctx->n_src; pa.tgt = ctx->tgt_sorted; pa.corr = ctx->corr_sorted; pa.pos = ctx->pos_sorted;
  pa.partials = ctx->partials;
  const int g = ctx->pass_grid;
  hipLaunchKernelGGL(k_info_accumulate, dim3(g), dim3(O3D_BLOCK), 0, ctx->stream, pa);
  hipLaunchKernelGGL(k_finalize_plain, dim3(1), dim3(O3D_FIN_THREADS), 0, ctx->stream, ctx->partials, g, dev_sums_out);
  O3D_HIP(ctx, hipGetLastError());
  return O3D_OK;
}

// dev_out: 19 doubles (see k_pairs_estimate), 20 with write_bad; dev_bad: one int, zeroed here.  Allocates nothing.
static hipError_t launch_pairs_on(hipStream_t stream, double *partials, const double *d_src, int64_t n_src, const double *d_tgt,
                                  const double *d_nrm, int64_t n_tgt, const int32_t *d_pairs, int64_t n_pairs, int estimation,
                                  int with_scaling, double *dev_out, int *dev_bad, int write_bad) {
  PairArgs a;
  a.src = d_src; a.n_src = n_src; a.tgt = d_tgt; a.nrm = d_nrm; a.n_tgt = n_tgt; a.pairs = d_pairs; a.n_pairs = n_pairs;
  a.partials = partials; a.bad = dev_bad;
  int64_t g = (n_pairs + O3D_BLOCK - 1) / O3D_BLOCK;
  if (g < 1) g = 1;
  if (g > O3D_MAX_GRID) g = O3D_MAX_GRID;
  hipError_t e = hipMemsetAsync(dev_bad, 0, sizeof(int), stream);
  if (e != hipSuccess) return e;
  if (estimation == O3D_EST_POINT_TO_PLANE && d_nrm)
    hipLaunchKernelGGL(k_pairs_accumulate<O3D_EST_POINT_TO_PLANE>, dim3((int)g), dim3(O3D_BLOCK), 0, stream, a);
  else
    hipLaunchKernelGGL(k_pairs_accumulate<O3D_EST_POINT_TO_POINT>, dim3((int)g), dim3(O3D_BLOCK), 0, stream, a);
  hipLaunchKernelGGL(k_pairs_estimate, dim3(1), dim3(O3D_FIN_THREADS), 0, stream, a, (int)g, estimation, with_scaling, dev_out,
                     write_bad);
  return hipGetLastError();
}

int o3d_launch_pairs(o3d_ctx *ctx, const double *d_src, int64_t n_src, const double *d_tgt, const double *d_nrm, int64_t n_tgt,
                     const int32_t *d_pairs, int64_t n_pairs, int estimation, int with_scaling, double *dev_out, int *dev_bad) {
  O3D_HIP(ctx, launch_pairs_on(ctx->stream, ctx->partials, d_src, n_src, d_tgt, d_nrm, n_tgt, d_pairs, n_pairs, estimation,
                               with_scaling, dev_out, dev_bad, 0));
  return O3D_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// o3d_compute_transformation_noalloc: the same two kernels for callers that cannot allocate (code under HIP graph capture,
// serving loops, harnesses that forbid hipMalloc).  Their scratch -- one 32-double partial row per workgroup and the count
// of out-of-range pairs -- lives in this file-scope device array instead of a context's buffers.  Every GPU the library
// runs on holds its own copy (512.25 KiB) at its own address until the process ends; the address is looked up once per
// device with hipGetSymbolAddress and host code only ever uses that pointer, never the array's name.
constexpr size_t kNoAllocPartialBytes = sizeof(double) * O3D_NSUMS * O3D_MAX_GRID;
constexpr size_t kNoAllocBytes = kNoAllocPartialBytes + 256;
static __device__ __align__(256) unsigned char scratch[kNoAllocBytes];
constexpr int kNoAllocMaxDevices = 64;
static std::mutex g_noalloc_mu;
static void *g_noalloc_ptr[kNoAllocMaxDevices] = {};

// `device` must be the current device.  Returns the device's scratch address, or nullptr with *err set.
static void *noalloc_scratch_of(int device, hipError_t *err) {
  *err = hipSuccess;
  if (device < 0 || device >= kNoAllocMaxDevices) { *err = hipErrorInvalidDevice; return nullptr; }
  std::lock_guard<std::mutex> lk(g_noalloc_mu);
  if (!g_noalloc_ptr[device]) {
    void *p = nullptr;
    const hipError_t e = hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !p) { *err = e != hipSuccess ? e : hipErrorInvalidSymbol; return nullptr; }
    g_noalloc_ptr[device] = p;
  }
  return g_noalloc_ptr[device];
}

// Called by o3d_ctx_create with the context's device current, so that a later call under graph capture finds the
// address already there.
int o3d_noalloc_prepare_device(int device) {
  hipError_t e;
  return noalloc_scratch_of(device, &e) ? O3D_OK : O3D_ERR_HIP;
}

extern "C" int o3d_compute_transformation_noalloc(int device_id, void *hip_stream, const double *d_src_xyz, int64_t n_src,
                                                  const double *d_tgt_xyz, const double *d_tgt_nrm, int64_t n_tgt,
                                                  const int32_t *d_corres_pairs, int64_t n_corres, int estimation, int with_scaling,
                                                  double *d_out20) {
  if (estimation != O3D_EST_POINT_TO_POINT && estimation != O3D_EST_POINT_TO_PLANE)
    return o3d_fail(nullptr, O3D_ERR_UNSUPPORTED, "o3d_compute_transformation_noalloc: estimation type %d is not on the device path", estimation);
  if (!d_out20 || n_src < 0 || n_tgt < 0 || n_corres < 0 || (n_corres > 0 && (!d_corres_pairs || !d_src_xyz || !d_tgt_xyz)))
    return o3d_fail(nullptr, O3D_ERR_INVALID_ARG, "o3d_compute_transformation_noalloc: bad sizes or null buffers");
  if (n_corres > O3D_NOALLOC_MAX_CORRES || n_src > O3D_NOALLOC_MAX_CORRES || n_tgt > O3D_NOALLOC_MAX_CORRES)
    return o3d_fail(nullptr, O3D_ERR_UNSUPPORTED, "o3d_compute_transformation_noalloc: %lld correspondences over %lld x %lld points "
                    "exceed the entry point's maximum of %lld each", (long long)n_corres, (long long)n_src, (long long)n_tgt,
                    (long long)O3D_NOALLOC_MAX_CORRES);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev)
    return o3d_fail(nullptr, O3D_ERR_HIP, "o3d_compute_transformation_noalloc: no HIP device %d (libo3d_amd has no CPU fallback)", device_id);
  int prev = -1;
  if (hipGetDevice(&prev) != hipSuccess) prev = -1;
  if (prev != device_id && hipSetDevice(device_id) != hipSuccess)
    return o3d_fail(nullptr, O3D_ERR_HIP, "o3d_compute_transformation_noalloc: hipSetDevice(%d) failed", device_id);
  hipError_t e;
  unsigned char *sp = static_cast<unsigned char *>(noalloc_scratch_of(device_id, &e));
  int st = O3D_OK;
  if (!sp) {
    st = o3d_fail(nullptr, O3D_ERR_HIP, "o3d_compute_transformation_noalloc: hipGetSymbolAddress(scratch): %s", hipGetErrorString(e));
  } else {
    // the array keeps the previous call's contents: the bad-pair count is cleared on the stream, every partial row the
    // estimate reads is written by this call's accumulation first
    e = launch_pairs_on((hipStream_t)hip_stream, reinterpret_cast<double *>(sp), d_src_xyz, n_src, d_tgt_xyz, d_tgt_nrm, n_tgt,
                        d_corres_pairs, n_corres, estimation, with_scaling, d_out20, reinterpret_cast<int *>(sp + kNoAllocPartialBytes), 1);
    if (e != hipSuccess) st = o3d_fail(nullptr, O3D_ERR_HIP, "o3d_compute_transformation_noalloc: launch: %s", hipGetErrorString(e));
  }
  if (prev >= 0 && prev != device_id) (void)hipSetDevice(prev);
  return st;
}

int o3d_launch_ransac_hypotheses(o3d_ctx *ctx, const RansacArgs &a) {
  if (a.n_hyp <= 0) return O3D_OK;
  hipLaunchKernelGGL(k_ransac_hypotheses, dim3((a.n_hyp + O3D_WAVE - 1) / O3D_WAVE), dim3(O3D_WAVE), 0, ctx->stream, a);
  O3D_HIP(ctx, hipGetLastError());
  return O3D_OK;
}

int o3d_launch_set_update(o3d_ctx *ctx, const double update[16]) {
  Mat16 u;
  for (int i = 0; i < 16; i++) u.m[i] = update[i];
  hipLaunchKernelGGL(k_icp_set_update, dim3(1), dim3(64), 0, ctx->stream, ctx->state, u);
  O3D_HIP(ctx, hipGetLastError());
  return O3D_OK;
}

// The arguments of one pass.  listed_pass: a pass behind the first (certificates, where they are in use).
struct PassPlan {
  PassArgs pa;
  int grows;    // workgroups of the pass kernels
  bool fused;   // k_icp_search_coop (else k_icp_search + k_icp_accumulate_mfma)
  bool tail;    // the search kernel ends the pass itself
  bool one_launch;  // k_icp_pass: certificates and search of a listed pass in one launch
  int nparts;
};

static PassPlan plan_pass(o3d_ctx *ctx, double *dev_sums_out, int fuse_solve, bool listed_pass) {
  PassPlan pl;
  PassArgs &pa = pl.pa;
  pa.stats = ctx->tile_stats;
  pa.ctr = ctx->counters;
  pa.src = ctx->src_sorted; pa.src_in = ctx->hyp_mode ? ctx->src_orig : ctx->src_sorted; pa.n_src = ctx->n_src; pa.tgt = ctx->tgt_sorted; pa.tgt_local = ctx->tgt_local;
  pa.tn = ctx->tn_valid ? ctx->tn_sorted : nullptr;
  pa.tidx = ctx->tgt_index;
  pa.stream = ctx->use_stream;
  pa.nrm = ctx->tn_valid ? nullptr : ctx->nrm_sorted; pa.cell_start = ctx->cell_start; pa.st = ctx->state; pa.corr = ctx->corr_sorted;
  pa.partials = ctx->partials; pa.thr = ctx->thr;
  pa.pos = ctx->pos_sorted;
  pa.seed = ctx->use_seed;
  pa.lb = ctx->lb_sorted;
  pa.work = ctx->work; pa.work_count = ctx->work_count;
  pa.tcol = ctx->tcol_sorted; pa.sint = ctx->sint_sorted; pa.slg = ctx->slg; pa.slp = ctx->slp;
  // Default: ONE kernel per pass -- the cooperative fp32-prefilter search with the exact fp64 resolution of what fp32
  // cannot decide and the estimate's sums fused in.  Alternative (grids too large for the fp32 error bound, or forced by
  // o3d_debug_set): the per-lane fp64 search followed by the accumulation kernel.  One partial row per workgroup.
  pl.fused = ctx->tile_ok && ctx->use_tile && ctx->gp.margin;
  pl.grows = pl.fused ? ctx->search_grid : ctx->pass_grid;
  // Certificates: the bounds refer to the position of the previous pass -- not in hypothesis mode, where every pass starts
  // from the untouched source, and not in the first pass, which has nothing to certify.  The work list is laid out for the
  // search kernel's chunks, so k_icp_cert runs on the same grid.
  const bool cert = pl.fused && ctx->use_cert && !ctx->hyp_mode;
  pa.gap = cert ? (float)(ctx->cert_gap * ctx->gp.h) : 0.f;
  pa.listed = (cert && listed_pass) ? 1 : 0;
  pa.cert_row0 = pl.grows;
  {  // one segment per wavefront, the same for both kernels
    const int64_t nseg = 4 * (int64_t)pl.grows;
    // (a wavefront takes one block of 64 out of every window of nseg blocks: see wave_range)
    const int64_t blocks = (ctx->n_src + O3D_WAVE - 1) / O3D_WAVE;
    pa.seg_len = ((blocks + nseg - 1) / nseg) * O3D_WAVE;
    if (pa.seg_len < O3D_WAVE) pa.seg_len = O3D_WAVE;
  }
  pa.grid_x = pl.grows;
  pl.nparts = pa.listed ? 2 * pl.grows : pl.grows;
  // The search kernel's last workgroups add the partial rows and run the estimate themselves (pass_tail).
  pl.tail = pl.fused && ctx->use_tail;
  // (see k_icp_pass for the two conditions on the rows)
  pl.one_launch = pl.fused && pl.tail && pa.listed && pa.stream && pa.cert_row0 % 32 == 0 && ctx->use_one_launch;
  pa.tail = pl.tail ? (fuse_solve ? 2 : 1) : 0;
  pa.nparts = pl.nparts; pa.has_normals = ctx->has_tgt_normals; pa.log_rows = ctx->log_rows; pa.log = ctx->log;
  pa.tickets = ctx->tickets; pa.gsums = ctx->gsums; pa.sums_out = dev_sums_out; pa.st_rw = ctx->state;
  return pl;
}

// O3D_TAIL_CHECK=1: what pass_tail added inside the pass's kernel against the separate-launch reduction over the same partial
// rows, bit for bit (the invariant at store_partial_row: a row written with a plain store shows up here as a difference).
// `was_done`: the loop had ended before this pass, whose kernels then returned at once.  Synchronises; debugging only.
static int o3d_tail_check_before(o3d_ctx *ctx, hipStream_t stream, int *was_done) {
  IcpState hs;
  O3D_HIP(ctx, hipMemcpyAsync(&hs, ctx->state, sizeof(IcpState), hipMemcpyDeviceToHost, stream));
  O3D_HIP(ctx, hipStreamSynchronize(stream));
  *was_done = hs.done;
  return O3D_OK;
}
static int o3d_tail_check_after(o3d_ctx *ctx, hipStream_t stream, int nparts, const double *dev_sums, const char *what) {
  if (!ctx->tail_check_buf) O3D_HIP(ctx, hipMalloc((void **)&ctx->tail_check_buf, sizeof(double) * O3D_NSUMS));
  hipLaunchKernelGGL(k_finalize_plain, dim3(1), dim3(O3D_FIN_THREADS), 0, stream, (const double *)ctx->partials, nparts, ctx->tail_check_buf);
  double h[2 * O3D_NSUMS];
  O3D_HIP(ctx, hipMemcpyAsync(h, ctx->tail_check_buf, sizeof(double) * O3D_NSUMS, hipMemcpyDeviceToHost, stream));
  O3D_HIP(ctx, hipMemcpyAsync(h + O3D_NSUMS, dev_sums, sizeof(double) * O3D_NSUMS, hipMemcpyDeviceToHost, stream));
  O3D_HIP(ctx, hipStreamSynchronize(stream));
  for (int k = 0; k < O3D_NSUMS; k++)
    if (memcmp(&h[k], &h[O3D_NSUMS + k], sizeof(double)) != 0)
      return o3d_fail(ctx, O3D_ERR_STATE, "O3D_TAIL_CHECK: %s: sum %d of the pass is %a in the kernel's own tail and %a over the same %d partial rows "
                      "added by a separate launch", what, k, h[O3D_NSUMS + k], h[k], nparts);
  if (ctx->tail_check > 1) fprintf(stderr, "O3D_TAIL_CHECK: %s: %d partial rows, n = %.0f: equal\n", what, nparts, h[O3D_SUM_NC]);
  return O3D_OK;
}

int o3d_launch_pass(o3d_ctx *ctx, double *dev_sums_out, int fuse_solve) {
  int check_done = 0;
  if (ctx->tail_check && ctx->use_tail) O3D_TRY(o3d_tail_check_before(ctx, ctx->stream, &check_done));
  const bool prof = ctx->prof_on && ctx->prof_n < 340;
  if (prof) {  // three events per pass: before the search kernel, after it, after the accumulation kernel (alternative path)
    for (int k = 0; k < 3; k++)
      if (!ctx->prof_ev[3 * ctx->prof_n + k]) O3D_HIP(ctx, hipEventCreate(&ctx->prof_ev[3 * ctx->prof_n + k]));
    O3D_HIP(ctx, hipEventRecord(ctx->prof_ev[3 * ctx->prof_n], ctx->stream));
  }
  const PassPlan pl = plan_pass(ctx, dev_sums_out, fuse_solve, ctx->enq_passes > 1);  // enq_passes counts this pass already
  const PassArgs &pa = pl.pa;
  const int grows = pl.grows, nparts = pl.nparts;
  const int est = ctx->estimation;
#define O3D_LAUNCH_EST(KERNEL, GRID, ...)                                                                                   \
  do {                                                                                                                     \
    if (est == O3D_EST_POINT_TO_PLANE)                                                                                     \
      hipLaunchKernelGGL(KERNEL<O3D_EST_POINT_TO_PLANE>, dim3(GRID), dim3(O3D_BLOCK), 0, ctx->stream, __VA_ARGS__);        \
    else if (est == O3D_EST_COLORED_ICP)                                                                                   \
      hipLaunchKernelGGL(KERNEL<O3D_EST_COLORED_ICP>, dim3(GRID), dim3(O3D_BLOCK), 0, ctx->stream, __VA_ARGS__);           \
    else                                                                                                                   \
      hipLaunchKernelGGL(KERNEL<O3D_EST_POINT_TO_POINT>, dim3(GRID), dim3(O3D_BLOCK), 0, ctx->stream, __VA_ARGS__);        \
  } while (0)
  if (pl.one_launch) {  // k_icp_pass: both halves of the pass in one launch (the profile's first interval stays empty)
    if (prof) O3D_HIP(ctx, hipEventRecord(ctx->prof_ev[3 * ctx->prof_n + 1], ctx->stream));
    if (pa.tn && est == O3D_EST_POINT_TO_PLANE)
      hipLaunchKernelGGL((k_icp_pass<O3D_EST_POINT_TO_PLANE, true>), dim3(grows), dim3(O3D_BLOCK), 0, ctx->stream, pa, ctx->gp);
    else if (pa.tn && est == O3D_EST_COLORED_ICP)
      hipLaunchKernelGGL((k_icp_pass<O3D_EST_COLORED_ICP, true>), dim3(grows), dim3(O3D_BLOCK), 0, ctx->stream, pa, ctx->gp);
    else if (est == O3D_EST_POINT_TO_PLANE)
      hipLaunchKernelGGL((k_icp_pass<O3D_EST_POINT_TO_PLANE, false>), dim3(grows), dim3(O3D_BLOCK), 0, ctx->stream, pa, ctx->gp);
    else if (est == O3D_EST_COLORED_ICP)
      hipLaunchKernelGGL((k_icp_pass<O3D_EST_COLORED_ICP, false>), dim3(grows), dim3(O3D_BLOCK), 0, ctx->stream, pa, ctx->gp);
    else
      hipLaunchKernelGGL((k_icp_pass<O3D_EST_POINT_TO_POINT, false>), dim3(grows), dim3(O3D_BLOCK), 0, ctx->stream, pa, ctx->gp);
  } else if (pa.listed && pa.stream) O3D_LAUNCH_EST(k_icp_cert_s, grows, pa);
  else if (pa.listed) O3D_LAUNCH_EST(k_icp_cert, grows, pa);
  if (pl.one_launch) {
  } else if (pl.fused) {  // profile: [0] = k_icp_cert (nothing in the first pass), [1] = k_icp_search_coop
    if (prof) O3D_HIP(ctx, hipEventRecord(ctx->prof_ev[3 * ctx->prof_n + 1], ctx->stream));
    // (two instantiations rather than a run-time branch: with both gathers compiled in, the kernel spilled six registers)
    if (pa.tn && est == O3D_EST_POINT_TO_PLANE)
      hipLaunchKernelGGL((k_icp_search_coop<O3D_EST_POINT_TO_PLANE, true>), dim3(grows), dim3(O3D_BLOCK), 0, ctx->stream, pa, ctx->gp);
    else if (pa.tn && est == O3D_EST_COLORED_ICP)
      hipLaunchKernelGGL((k_icp_search_coop<O3D_EST_COLORED_ICP, true>), dim3(grows), dim3(O3D_BLOCK), 0, ctx->stream, pa, ctx->gp);
    else if (est == O3D_EST_POINT_TO_PLANE)
      hipLaunchKernelGGL((k_icp_search_coop<O3D_EST_POINT_TO_PLANE, false>), dim3(grows), dim3(O3D_BLOCK), 0, ctx->stream, pa, ctx->gp);
    else if (est == O3D_EST_COLORED_ICP)
      hipLaunchKernelGGL((k_icp_search_coop<O3D_EST_COLORED_ICP, false>), dim3(grows), dim3(O3D_BLOCK), 0, ctx->stream, pa, ctx->gp);
    else
      hipLaunchKernelGGL((k_icp_search_coop<O3D_EST_POINT_TO_POINT, false>), dim3(grows), dim3(O3D_BLOCK), 0, ctx->stream, pa, ctx->gp);
  } else {
    hipLaunchKernelGGL(k_icp_search, dim3(grows), dim3(O3D_BLOCK), 0, ctx->stream, pa, ctx->gp);
    if (prof) O3D_HIP(ctx, hipEventRecord(ctx->prof_ev[3 * ctx->prof_n + 1], ctx->stream));
    O3D_LAUNCH_EST(k_icp_accumulate_mfma, grows, pa);
  }
  if (prof) {
    O3D_HIP(ctx, hipEventRecord(ctx->prof_ev[3 * ctx->prof_n + 2], ctx->stream));
    ctx->prof_n++;
  }
  if (pl.tail) {
  } else if (fuse_solve)
    hipLaunchKernelGGL(k_icp_finalize_solve, dim3(1), dim3(O3D_FIN_THREADS), 0, ctx->stream, ctx->partials, nparts, ctx->state,
                       dev_sums_out, ctx->has_tgt_normals, ctx->log, ctx->log_rows);
  else
    hipLaunchKernelGGL(k_icp_finalize, dim3(1), dim3(O3D_FIN_THREADS), 0, ctx->stream, ctx->partials, nparts, ctx->state,
                       dev_sums_out);
#undef O3D_LAUNCH_EST
  O3D_HIP(ctx, hipGetLastError());
  if (ctx->tail_check && pl.tail && !check_done) O3D_TRY(o3d_tail_check_after(ctx, ctx->stream, nparts, dev_sums_out, "pass"));
#if O3D_PASS_PROF
  if (pl.one_launch) {
    static unsigned long long ht[4 * O3D_MAX_GRID][O3D_PASS_MARKS];
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipMemcpyFromSymbol(ht, HIP_SYMBOL(g_pass_t), sizeof(ht));
    const int nw = 4 * pl.grows;
    unsigned long long t0 = ~0ull;
    for (int w = 0; w < nw; w++) if (ht[w][0] && ht[w][0] < t0) t0 = ht[w][0];
    // per milestone: mean and latest time after the first wavefront came in; then per XCD the latest "loop done" and "all done"
    double mean[6] = {0}, last[6] = {0}, xl[8] = {0}, xe[8] = {0}, xm[8] = {0};
    for (int w = 0; w < nw; w++) {
      const int xcd = (w / 4) & 7;
      for (int i = 0; i < 6; i++) {
        const double t = ht[w][i] >= t0 ? (ht[w][i] - t0) / 100.0 : 0.0;
        mean[i] += t / nw;
        if (t > last[i]) last[i] = t;
        if (i == 2) { xm[xcd] += t / (nw / 8); if (t > xl[xcd]) xl[xcd] = t; }
        if (i == 5 && t > xe[xcd]) xe[xcd] = t;
      }
    }
    fprintf(stderr, "pass milestones, us after the first wavefront came in (mean / latest wavefront): in %.1f/%.1f loop starts %.1f/%.1f loop done %.1f/%.1f "
            "workgroup's loops done %.1f/%.1f certificate half done %.1f/%.1f all done %.1f/%.1f | per XCD loop done mean/latest, all done:",
            mean[0], last[0], mean[1], last[1], mean[2], last[2], mean[3], last[3], mean[4], last[4], mean[5], last[5]);
    for (int x = 0; x < 8; x++) fprintf(stderr, " [%.0f/%.0f %.0f]", xm[x], xl[x], xe[x]);
    fprintf(stderr, "\n");
    if (const char *dump = getenv("O3D_PASS_PROF_DUMP")) {  // raw marks of this pass, appended
      if (FILE *f = fopen(dump, "ab")) { fwrite(ht, sizeof(ht[0]), nw, f); fclose(f); }
    }
    memset(ht, 0, sizeof(ht));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pass_t), ht, sizeof(ht));
  }
#endif
#if O3D_SRCH_PROF
  {
    unsigned long long hp[8];
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_srch_prof), sizeof(hp));
    fprintf(stderr, "search phases (M wave-cycles): head %.1f consume %.1f own-scan %.1f listing %.1f queue %.1f verdict %.1f tail %.1f\n", hp[0] * 1e-6,
            hp[1] * 1e-6, hp[2] * 1e-6, hp[3] * 1e-6, hp[4] * 1e-6, hp[5] * 1e-6, hp[6] * 1e-6);
    for (auto &v : hp) v = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_srch_prof), hp, sizeof(hp));
  }
#endif
  return O3D_OK;
}

// ---- several registrations in lock step ---------------------------------------------------------------------------------
// A batch of small registrations (BASELINE config 5) one by one is a stream of 25 us kernels, each filling three quarters
// of the GPU for one latency chain, with the gaps of dependent launches in between (profiles/r02_n_*).  Here `n` slots --
// contexts that have been set up for one registration each, all on `lead`'s stream -- advance together: ONE launch per
// pass kernel over a grid of (workgroups of the largest slot) x n, the slot's arguments read from device memory by
// blockIdx.y.  Everything a slot computes is what it would compute alone (its own state, partial rows, counters).
bool o3d_gang_eligible(o3d_ctx *ctx) {
  if (ctx->prof_on || ctx->hyp_mode) return false;
  const bool fused = ctx->tile_ok && ctx->use_tile && ctx->gp.margin;
  return fused && ctx->use_tail;
}

// upload: the slots have just been set up (first pass of the gang's loop).  first: the unlisted first pass.
int o3d_launch_pass_gang(o3d_ctx *lead, o3d_ctx *const *slots, int n, bool upload, bool first) {
  if (n < 1 || n > O3D_GANG_MAX) return o3d_fail(lead, O3D_ERR_INVALID_ARG, "gang of %d registrations", n);
  if (!lead->d_gang) {
    O3D_HIP(lead, hipMalloc(&lead->d_gang, 2 * O3D_GANG_MAX * sizeof(PassLaunch)));
    O3D_HIP(lead, hipHostMalloc(&lead->pinned_gang, 2 * O3D_GANG_MAX * sizeof(PassLaunch), hipHostMallocDefault));
  }
  const int est = slots[0]->estimation;
  if (upload) {
    PassLaunch *h = reinterpret_cast<PassLaunch *>(lead->pinned_gang);
    int gx = 0;
    bool cert = false, tn = false;
    for (int k = 0; k < n; k++) {
      o3d_ctx *c = slots[k];
      const PassPlan p0 = plan_pass(c, c->sums, 1, false), p1 = plan_pass(c, c->sums, 1, true);
      h[k].a = p0.pa; h[k].gp = c->gp;
      h[O3D_GANG_MAX + k].a = p1.pa; h[O3D_GANG_MAX + k].gp = c->gp;
      const bool ck = p1.pa.listed != 0, tk = p0.pa.tn != nullptr && est != O3D_EST_POINT_TO_POINT;
      if (k == 0) lead->gang_one_launch = p1.one_launch ? 1 : 0;
      if (k > 0 && (ck != cert || tk != tn || c->estimation != est || (p1.one_launch ? 1 : 0) != lead->gang_one_launch))
        return o3d_fail(lead, O3D_ERR_STATE, "gang: the slots do not run the same kernels");
      cert = ck; tn = tk;
      gx = p0.grows > gx ? p0.grows : gx;
    }
    O3D_HIP(lead, hipMemcpyAsync(lead->d_gang, h, 2 * O3D_GANG_MAX * sizeof(PassLaunch), hipMemcpyHostToDevice, lead->stream));
    lead->gang_grid = gx; lead->gang_cert = cert ? 1 : 0; lead->gang_tn = tn ? 1 : 0; lead->gang_stream = h[O3D_GANG_MAX].a.stream;
  }
  const PassLaunch *d = reinterpret_cast<const PassLaunch *>(lead->d_gang) + (first ? 0 : O3D_GANG_MAX);
  const dim3 grid((unsigned)lead->gang_grid, (unsigned)n), block(O3D_BLOCK);
  hipStream_t st = lead->stream;
  int check_done[O3D_GANG_MAX];
  const bool check = slots[0]->tail_check != 0;  // (gang-eligible slots end their passes in the kernel: o3d_gang_eligible)
  if (check)
    for (int k = 0; k < n; k++) O3D_TRY(o3d_tail_check_before(slots[k], st, &check_done[k]));
  auto checked = [&]() -> int {  // O3D_TAIL_CHECK=1: every slot's rows once more
    if (!check) return O3D_OK;
    for (int k = 0; k < n; k++) {
      if (check_done[k]) continue;
      const PassPlan pk = plan_pass(slots[k], slots[k]->sums, 1, !first);
      const int stt = o3d_tail_check_after(slots[k], st, pk.nparts, slots[k]->sums, "pass of a gang");
      if (stt != O3D_OK) return o3d_fail(lead, stt, "%s (slot %d of %d)", o3d_last_error(slots[k]), k, n);
    }
    return O3D_OK;
  };
  if (!first && lead->gang_cert && lead->gang_one_launch) {
    if (est == O3D_EST_POINT_TO_PLANE && lead->gang_tn) hipLaunchKernelGGL((k_icp_pass_batch<O3D_EST_POINT_TO_PLANE, true>), grid, block, 0, st, d);
    else if (est == O3D_EST_POINT_TO_PLANE) hipLaunchKernelGGL((k_icp_pass_batch<O3D_EST_POINT_TO_PLANE, false>), grid, block, 0, st, d);
    else if (est == O3D_EST_COLORED_ICP && lead->gang_tn) hipLaunchKernelGGL((k_icp_pass_batch<O3D_EST_COLORED_ICP, true>), grid, block, 0, st, d);
    else if (est == O3D_EST_COLORED_ICP) hipLaunchKernelGGL((k_icp_pass_batch<O3D_EST_COLORED_ICP, false>), grid, block, 0, st, d);
    else hipLaunchKernelGGL((k_icp_pass_batch<O3D_EST_POINT_TO_POINT, false>), grid, block, 0, st, d);
    O3D_HIP(lead, hipGetLastError());
    return checked();
  }
  if (!first && lead->gang_cert) {
    if (lead->gang_stream) {
      if (est == O3D_EST_POINT_TO_PLANE) hipLaunchKernelGGL(k_icp_cert_batch_s<O3D_EST_POINT_TO_PLANE>, grid, block, 0, st, d);
      else if (est == O3D_EST_COLORED_ICP) hipLaunchKernelGGL(k_icp_cert_batch_s<O3D_EST_COLORED_ICP>, grid, block, 0, st, d);
      else hipLaunchKernelGGL(k_icp_cert_batch_s<O3D_EST_POINT_TO_POINT>, grid, block, 0, st, d);
    } else if (est == O3D_EST_POINT_TO_PLANE) hipLaunchKernelGGL(k_icp_cert_batch<O3D_EST_POINT_TO_PLANE>, grid, block, 0, st, d);
    else if (est == O3D_EST_COLORED_ICP) hipLaunchKernelGGL(k_icp_cert_batch<O3D_EST_COLORED_ICP>, grid, block, 0, st, d);
    else hipLaunchKernelGGL(k_icp_cert_batch<O3D_EST_POINT_TO_POINT>, grid, block, 0, st, d);
  }
  if (est == O3D_EST_POINT_TO_PLANE && lead->gang_tn) hipLaunchKernelGGL((k_icp_search_coop_batch<O3D_EST_POINT_TO_PLANE, true>), grid, block, 0, st, d);
  else if (est == O3D_EST_POINT_TO_PLANE) hipLaunchKernelGGL((k_icp_search_coop_batch<O3D_EST_POINT_TO_PLANE, false>), grid, block, 0, st, d);
  else if (est == O3D_EST_COLORED_ICP && lead->gang_tn) hipLaunchKernelGGL((k_icp_search_coop_batch<O3D_EST_COLORED_ICP, true>), grid, block, 0, st, d);
  else if (est == O3D_EST_COLORED_ICP) hipLaunchKernelGGL((k_icp_search_coop_batch<O3D_EST_COLORED_ICP, false>), grid, block, 0, st, d);
  else hipLaunchKernelGGL((k_icp_search_coop_batch<O3D_EST_POINT_TO_POINT, false>), grid, block, 0, st, d);
  O3D_HIP(lead, hipGetLastError());
  return checked();
}

// Coloured ICP: per sorted target point {intensity, colour gradient}, per processing position the source intensity
// ((r + g + b) / 3, ColoredICP.cpp:86,152-153).
int o3d_launch_color_gather(o3d_ctx *ctx, const double *d_tgt_col, const double *d_grad, const double *d_src_col) {
  if (ctx->n_tgt_grid > 0) {
    int64_t g = (ctx->n_tgt_grid + O3D_BLOCK - 1) / O3D_BLOCK;
    if (g > O3D_MAX_GRID) g = O3D_MAX_GRID;
    hipLaunchKernelGGL(k_gather_tcol, dim3((int)g), dim3(O3D_BLOCK), 0, ctx->stream, ctx->tgt_sorted, d_tgt_col, d_grad,
                       ctx->n_tgt_grid, ctx->tcol_sorted);
  }
  if (ctx->n_src > 0) {
    int64_t g = (ctx->n_src + O3D_BLOCK - 1) / O3D_BLOCK;
    if (g > O3D_MAX_GRID) g = O3D_MAX_GRID;
    hipLaunchKernelGGL(k_gather_sint, dim3((int)g), dim3(O3D_BLOCK), 0, ctx->stream, ctx->src_perm, d_src_col, ctx->n_src,
                       ctx->sint_sorted);
  }
  O3D_HIP(ctx, hipGetLastError());
  return O3D_OK;
}

int o3d_launch_solve(o3d_ctx *ctx, const double *dev_all_sums, int n_ranks) {
  hipLaunchKernelGGL(k_icp_solve, dim3(1), dim3(64), 0, ctx->stream, ctx->state, dev_all_sums, n_ranks,
                     ctx->has_tgt_normals, ctx->log, ctx->log_rows);
  O3D_HIP(ctx, hipGetLastError());
  return O3D_OK;
}

int o3d_launch_scatter_corr(o3d_ctx *ctx, int32_t *dev_out) {
  if (ctx->n_src <= 0) return O3D_OK;
  int64_t g = (ctx->n_src + O3D_BLOCK - 1) / O3D_BLOCK;
  if (g > O3D_MAX_GRID) g = O3D_MAX_GRID;
  hipLaunchKernelGGL(k_scatter_corr, dim3((int)g), dim3(O3D_BLOCK), 0, ctx->stream, ctx->corr_sorted, ctx->src_perm, ctx->n_src,
                     dev_out);
  O3D_HIP(ctx, hipGetLastError());
  return O3D_OK;
}

