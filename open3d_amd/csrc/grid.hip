// Uniform cell grid over the target cloud: the replacement for KDTreeFlann::SetGeometry
// (reference geometry/KDTreeFlann.cpp:58-73,154-167) on the ICP path.
//
// Layout in HBM
//   tgt_sorted[n]  : double4 {x, y, z, bits(original index)} ordered by cell key (x fastest), stable
//   nrm_sorted[n]  : 3 doubles per sorted target point (point-to-plane only)
//   cell_start[k]  : number of target points with key < k, for k in [0, ncell]  (dense, uint32)
// A cell is a cube of side h >= max_correspondence_distance*(1+1e-5), so every target point within the search
// radius of a query lies in the 3x3x3 block around the query's cell.  Each x-row is padded by one empty cell on
// both ends, which makes the three x-neighbours of any cell ONE contiguous run of tgt_sorted:
// [cell_start[k-1], cell_start[k+2]) -- 9 run look-ups per query instead of 27.
#include <math.h>

#include <algorithm>

#include "o3d_internal.h"

namespace {

constexpr double kCellSlack = 1.0 + 1e-5;      // h = r * slack: covers float(r*r) round-up and floor() rounding
constexpr uint64_t kMaxDenseCells = 1ull << 31;  // uint32 keys; dense table <= 8 GiB
constexpr double kCellsPerPointBudget = 64.0;   // coarsen the grid when the dense table would dwarf the cloud
constexpr int kMinBrickBits = 3, kMaxBrickBits = 6;  // bricks of 8^3 .. 64^3 cells
constexpr uint64_t kMaxBrickDir = 1ull << 27;        // directory entries (one per brick of the bounding box): 512 MiB
constexpr double kBrickEntriesPerPoint = 512.0;      // ... and the bricks' own budget: 2 KiB of table per point, 512 MiB at least
constexpr double kMaxBrickAxis = 1048000.0;          // cells per axis (the fp32 prefilter wants cell indices below 2^20)

struct Bounds6 {
  double mn[3], mx[3];
  unsigned long long finite;
};

__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fmin(v, __shfl_xor(v, d, O3D_WAVE));
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fmax(v, __shfl_xor(v, d, O3D_WAVE));
  return v;
}

// K8: AABB of the finite points (Geometry3D::ComputeMinBound/MaxBound, geometry/Geometry3D.cpp:39-53).
__global__ __launch_bounds__(O3D_BLOCK) void k_bounds_partial(const double *__restrict__ xyz, int64_t n, Bounds6 *__restrict__ part) {
  __shared__ double s[O3D_BLOCK / O3D_WAVE][6];
  __shared__ unsigned long long sc[O3D_BLOCK / O3D_WAVE];
  double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  unsigned long long cnt = 0;
  for (int64_t i = (int64_t)blockIdx.x * O3D_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * O3D_BLOCK) {
    double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    if (isfinite(x) && isfinite(y) && isfinite(z)) {
      mn[0] = fmin(mn[0], x); mn[1] = fmin(mn[1], y); mn[2] = fmin(mn[2], z);
      mx[0] = fmax(mx[0], x); mx[1] = fmax(mx[1], y); mx[2] = fmax(mx[2], z);
      cnt++;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int d = 0; d < 3; d++) {
    mn[d] = wave_min(mn[d]);
    mx[d] = wave_max(mx[d]);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d, O3D_WAVE);
  if (lane == 0) {
    for (int d = 0; d < 3; d++) { s[wave][d] = mn[d]; s[wave][3 + d] = mx[d]; }
    sc[wave] = cnt;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    Bounds6 b;
    for (int d = 0; d < 3; d++) { b.mn[d] = s[0][d]; b.mx[d] = s[0][3 + d]; }
    b.finite = sc[0];
    for (int w = 1; w < O3D_BLOCK / O3D_WAVE; w++) {
      for (int d = 0; d < 3; d++) { b.mn[d] = fmin(b.mn[d], s[w][d]); b.mx[d] = fmax(b.mx[d], s[w][3 + d]); }
      b.finite += sc[w];
    }
    part[blockIdx.x] = b;
  }
}

__global__ __launch_bounds__(O3D_BLOCK) void k_bounds_final(const Bounds6 *__restrict__ part, int nparts, Bounds6 *__restrict__ out) {
  __shared__ double s[O3D_BLOCK / O3D_WAVE][6];
  __shared__ unsigned long long sc[O3D_BLOCK / O3D_WAVE];
  double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  unsigned long long cnt = 0;
  for (int p = threadIdx.x; p < nparts; p += O3D_BLOCK) {
    for (int d = 0; d < 3; d++) { mn[d] = fmin(mn[d], part[p].mn[d]); mx[d] = fmax(mx[d], part[p].mx[d]); }
    cnt += part[p].finite;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int d = 0; d < 3; d++) {
    mn[d] = wave_min(mn[d]);
    mx[d] = wave_max(mx[d]);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d, O3D_WAVE);
  if (lane == 0) {
    for (int d = 0; d < 3; d++) { s[wave][d] = mn[d]; s[wave][3 + d] = mx[d]; }
    sc[wave] = cnt;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    Bounds6 b;
    for (int d = 0; d < 3; d++) { b.mn[d] = s[0][d]; b.mx[d] = s[0][3 + d]; }
    b.finite = sc[0];
    for (int w = 1; w < O3D_BLOCK / O3D_WAVE; w++) {
      for (int d = 0; d < 3; d++) { b.mn[d] = fmin(b.mn[d], s[w][d]); b.mx[d] = fmax(b.mx[d], s[w][3 + d]); }
      b.finite += sc[w];
    }
    *out = b;
  }
}

// K1a: cell key per target point, + per-cell population count (histogram) in the dense table.
// Non-finite points can never be a nearest neighbour (d2 is NaN or inf): they get the sentinel key ncell and
// sort behind every real cell.
__global__ __launch_bounds__(O3D_BLOCK) void k_target_keys(const double *__restrict__ xyz, int64_t n, GridParams gp,
                                                            uint32_t *__restrict__ keys, uint32_t *__restrict__ vals, uint32_t tag) {
  for (int64_t i = (int64_t)blockIdx.x * O3D_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * O3D_BLOCK) {
    double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    uint32_t key = gp.ncell;
    if (isfinite(x) && isfinite(y) && isfinite(z)) {
      int cx = min(max(o3d_cell_coord(x, gp.gmin[0], gp.inv_h, gp.nx), 0), gp.nx - 1);
      int cy = min(max(o3d_cell_coord(y, gp.gmin[1], gp.inv_h, gp.ny), 0), gp.ny - 1);
      int cz = min(max(o3d_cell_coord(z, gp.gmin[2], gp.inv_h, gp.nz), 0), gp.nz - 1);
      key = ((uint32_t)cz * (uint32_t)gp.ny + (uint32_t)cy) * (uint32_t)gp.nxp + (uint32_t)(cx + 1);
    }
    keys[i] = key | tag;  // (tag: the job number of a gang's common sort, above every key bit; else 0)
    vals[i] = (uint32_t)i;
  }
}

// K1b: the dense prefix table from the SORTED keys: cell_start[k] = number of keys < k, for every k in [0, ntab).
// One workgroup per kCtCells consecutive cells: the range of sorted keys that falls into them is found by a 64-ary search
// (one wavefront, 64 probes per step), counted in LDS, scanned in the workgroup and written out once, coalesced -- the
// table is written exactly once and never read (the earlier version cleared it, built a histogram with one global atomic
// per point on a table far larger than any cache, and ran a three-kernel scan over all of it).
constexpr int kCtCells = 8192;      // cells per workgroup: tables of millions of cells
constexpr int kCtCellsSmall = 1024;  // small tables: more workgroups, a short count loop each
__device__ __forceinline__ uint32_t wave_lower_bound(const uint32_t *__restrict__ keys, uint32_t n, uint64_t target, int lane,
                                                     uint32_t mask) {
  uint32_t lo = 0, hi = n;
  while (hi > lo) {
    const uint32_t step = (hi - lo + 63u) / 64u;
    const uint64_t idx = (uint64_t)lo + (uint64_t)lane * step;
    const bool pred = idx < hi && (uint64_t)(keys[idx] & mask) < target;
    const int k = __popcll(__ballot(pred));  // the keys are sorted: pred holds for lanes 0 .. k - 1
    if (k == 0) { hi = lo; break; }
    const uint32_t nlo = lo + (uint32_t)(k - 1) * step + 1u;
    const uint64_t nhi = (uint64_t)lo + (uint64_t)k * step;
    hi = nhi < hi ? (uint32_t)nhi : hi;
    lo = nlo;
  }
  return lo;
}

template <int kCells>
__global__ __launch_bounds__(O3D_BLOCK) void k_cell_table(const uint32_t *__restrict__ keys, uint32_t n, uint32_t *__restrict__ cell_start,
                                                           uint64_t ntab, uint32_t mask) {
  __shared__ __align__(16) uint32_t cnt[kCells];
  __shared__ uint32_t bounds[2];
  __shared__ uint32_t wsum[O3D_BLOCK / O3D_WAVE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint64_t c0 = (uint64_t)blockIdx.x * kCells;
  if (blockIdx.x == 0 && tid < 8) cell_start[ntab + tid] = 0u;  // the padding entries behind the table (one memset less)
  if (wave < 2) {  // the two ends of the key range, one wavefront each
    const uint32_t a = wave_lower_bound(keys, n, c0 + (uint64_t)wave * kCells, lane, mask);
    if (lane == 0) bounds[wave] = a;
  }
  constexpr int kQuads = kCells / (4 * O3D_BLOCK);  // uint4 stores per thread
  uint4 *cnt4 = reinterpret_cast<uint4 *>(cnt);
#pragma unroll
  for (int q = 0; q < kQuads; q++) cnt4[q * O3D_BLOCK + tid] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  const uint32_t p0 = bounds[0], p1 = bounds[1];
  if (p1 > p0) {  // (most workgroups of a sparse table own no point and skip this)
    for (uint32_t p = p0 + tid; p < p1; p += O3D_BLOCK) atomicAdd(&cnt[(uint32_t)((uint64_t)(keys[p] & mask) - c0)], 1u);
    __syncthreads();
    // exclusive scan in place: each wavefront over its own quarter, 256 cells (one uint4 per lane) at a time
    constexpr int kWaveCells = kCells / (O3D_BLOCK / O3D_WAVE);
    uint32_t carry = 0;
#pragma unroll 2
    for (int it = 0; it < kWaveCells / (4 * O3D_WAVE); it++) {
      const int qi = (wave * kWaveCells) / 4 + it * O3D_WAVE + lane;
      const uint4 v = cnt4[qi];
      const uint32_t sum = v.x + v.y + v.z + v.w;
      uint32_t incl = sum;
#pragma unroll
      for (int d = 1; d < O3D_WAVE; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d, O3D_WAVE);
        if (lane >= d) incl += o;
      }
      const uint32_t e = carry + incl - sum;
      cnt4[qi] = make_uint4(e, e + v.x, e + v.x + v.y, e + v.x + v.y + v.z);
      carry += __shfl(incl, O3D_WAVE - 1, O3D_WAVE);
    }
    if (lane == 0) wsum[wave] = carry;
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < kQuads; q++) {
    const int qi = q * O3D_BLOCK + tid;
    uint4 v = make_uint4(p0, p0, p0, p0);
    if (p1 > p0) {
      constexpr int kWaveQuads = kCells / (O3D_BLOCK / O3D_WAVE) / 4;
      uint32_t base = p0;
      for (int w = 0; w < qi / kWaveQuads; w++) base += wsum[w];
      const uint4 e = cnt4[qi];
      v = make_uint4(base + e.x, base + e.y, base + e.z, base + e.w);
    }
    const uint64_t c = c0 + 4ull * (uint64_t)qi;
    if (c + 3 < ntab) {
      // a table far larger than the caches (hundreds of MB: written once, read by later kernels) leaves with streaming
      // stores -- 210 -> 160 us for the 844 MB table of the 10 M headline; smaller ones stay in the cache for their readers
      if (ntab > (64ull << 20)) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 nv = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(nv, reinterpret_cast<u32x4 *>(cell_start + c));
      } else {
        *reinterpret_cast<uint4 *>(cell_start + c) = v;
      }
    } else {
      if (c < ntab) cell_start[c] = v.x;
      if (c + 1 < ntab) cell_start[c + 1] = v.y;
      if (c + 2 < ntab) cell_start[c + 2] = v.z;
    }
  }
}

// K1c: gather the target into cell order (stable: ascending original index inside a cell).
__global__ __launch_bounds__(O3D_BLOCK) void k_gather_target(const double *__restrict__ xyz, const double *__restrict__ nrm,
                                                              const uint32_t *__restrict__ vals, int64_t n_grid, GridParams gp,
                                                              double4 *__restrict__ tgt_sorted, double *__restrict__ nrm_sorted,
                                                              float4 *__restrict__ tgt_local, double2 *__restrict__ tn_sorted,
                                                              uint32_t *__restrict__ tgt_index, int64_t n_pad) {
  if (blockIdx.x == 0 && threadIdx.x < 4) tgt_local[n_pad + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);  // the read-ahead padding
  for (int64_t p = (int64_t)blockIdx.x * O3D_BLOCK + threadIdx.x; p < n_grid; p += (int64_t)gridDim.x * O3D_BLOCK) {
    const uint32_t i = vals[p];
    double4 q;
    q.x = xyz[3 * (size_t)i];
    q.y = xyz[3 * (size_t)i + 1];
    q.z = xyz[3 * (size_t)i + 2];
    q.w = __longlong_as_double((long long)i);
    tgt_sorted[p] = q;
    tgt_index[p] = i;
    {  // fp32 copy relative to the origin of the cell the point was filed under (same arithmetic as k_target_keys)
      const int cx = min(max(o3d_cell_coord(q.x, gp.gmin[0], gp.inv_h, gp.nx), 0), gp.nx - 1);
      const int cy = min(max(o3d_cell_coord(q.y, gp.gmin[1], gp.inv_h, gp.ny), 0), gp.ny - 1);
      const int cz = min(max(o3d_cell_coord(q.z, gp.gmin[2], gp.inv_h, gp.nz), 0), gp.nz - 1);
      float4 l;
      l.x = (float)((q.x - gp.gmin[0]) - (double)cx * gp.h);
      l.y = (float)((q.y - gp.gmin[1]) - (double)cy * gp.h);
      l.z = (float)((q.z - gp.gmin[2]) - (double)cz * gp.h);
      l.w = __int_as_float(cx + 1);
      tgt_local[p] = l;
    }
    if (nrm) {  // one form or the other: the 48-byte records of the ICP loop, or plain normals
      const double n0 = nrm[3 * (size_t)i], n1 = nrm[3 * (size_t)i + 1], n2 = nrm[3 * (size_t)i + 2];
      if (tn_sorted) {
        tn_sorted[3 * p] = make_double2(q.x, q.y);
        tn_sorted[3 * p + 1] = make_double2(q.z, n0);
        tn_sorted[3 * p + 2] = make_double2(n1, n2);
      } else {
        nrm_sorted[3 * p] = n0;
        nrm_sorted[3 * p + 1] = n1;
        nrm_sorted[3 * p + 2] = n2;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Bricked table (sparse scenes).  The dense table costs one entry per cell of the bounding box; a scene that is mostly
// empty space -- clusters of surface far apart -- would need more cells than the budget allows, and coarser cells put
// hundreds of points into each.  Here the cells keep the side max_dist and the table holds only bricks of B^3 cells that
// some query can see a target point from (the cell of a point and its 26 neighbours).  A brick carries a halo -- (B+2) x
// (B+2) rows of B+3 entries -- so that the rows and entries a query in one of its cells reads (cells cx-1 .. cx+1 and the
// end of cx+1, rows cy-1 .. cy+1, cz-1 .. cz+1) lie inside it at fixed strides: the search kernels change in ONE place,
// the address of the query's row (o3d_row_index: one more dependent load, the directory entry).  Every entry is what the
// dense table would hold at that cell -- the number of target points whose row-major cell key is smaller -- so runs
// along x stay contiguous in the cell-ordered target whatever brick they are read from.  Bricks nobody can see a point
// from share one empty brick (all zero: every run empty).
__global__ __launch_bounds__(O3D_BLOCK) void k_target_keys64(const double *__restrict__ xyz, int64_t n, GridParams gp,
                                                              uint64_t *__restrict__ keys, uint32_t *__restrict__ vals) {
  for (int64_t i = (int64_t)blockIdx.x * O3D_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * O3D_BLOCK) {
    double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    uint64_t key = ~0ull;  // non-finite points sort behind every cell
    if (isfinite(x) && isfinite(y) && isfinite(z)) {
      int cx = min(max(o3d_cell_coord(x, gp.gmin[0], gp.inv_h, gp.nx), 0), gp.nx - 1);
      int cy = min(max(o3d_cell_coord(y, gp.gmin[1], gp.inv_h, gp.ny), 0), gp.ny - 1);
      int cz = min(max(o3d_cell_coord(z, gp.gmin[2], gp.inv_h, gp.nz), 0), gp.nz - 1);
      key = ((uint64_t)cz * (uint64_t)gp.ny + (uint64_t)cy) * (uint64_t)gp.nxp + (uint64_t)(cx + 1);
    }
    keys[i] = key;
    vals[i] = (uint32_t)i;
  }
}

// One flag per brick that a query can see a point from: for the first point of every occupied cell, the bricks of the 27
// cells around it.
__global__ __launch_bounds__(O3D_BLOCK) void k_brick_mark(const uint64_t *__restrict__ keys, int64_t n_grid, GridParams gp,
                                                           uint32_t *__restrict__ flags) {
  for (int64_t p = (int64_t)blockIdx.x * O3D_BLOCK + threadIdx.x; p < n_grid; p += (int64_t)gridDim.x * O3D_BLOCK) {
    const uint64_t key = keys[p];
    if (p > 0 && keys[p - 1] == key) continue;
    const int cx = (int)(key % (uint64_t)gp.nxp) - 1;
    const uint64_t row = key / (uint64_t)gp.nxp;
    const int cy = (int)(row % (uint64_t)gp.ny), cz = (int)(row / (uint64_t)gp.ny);
    const int b = gp.brick_bits;
    for (int dz = -1; dz <= 1; dz++)
      for (int dy = -1; dy <= 1; dy++)
        for (int dx = -1; dx <= 1; dx++) {
          const int qx = cx + dx, qy = cy + dy, qz = cz + dz;
          if (qx < 0 || qx >= gp.nx || qy < 0 || qy >= gp.ny || qz < 0 || qz >= gp.nz) continue;
          flags[((size_t)(qz >> b) * (size_t)gp.nby + (size_t)(qy >> b)) * (size_t)gp.nbx + (size_t)(qx >> b)] = 1u;
        }
  }
}

// flags + their exclusive scan -> the directory (first entry of the brick's table; brick 0 is the shared empty one) and
// the list of the bricks that exist.
__global__ __launch_bounds__(O3D_BLOCK) void k_brick_dir(uint32_t *__restrict__ flags_dir, const uint32_t *__restrict__ rank, int64_t ndir,
                                                          uint32_t brick_entries, uint32_t *__restrict__ list) {
  for (int64_t i = (int64_t)blockIdx.x * O3D_BLOCK + threadIdx.x; i < ndir; i += (int64_t)gridDim.x * O3D_BLOCK) {
    const uint32_t f = flags_dir[i];
    if (f) {
      const uint32_t r = rank[i];
      list[r] = (uint32_t)i;
      flags_dir[i] = (r + 1u) * brick_entries;
    }
  }
}

__device__ __forceinline__ uint32_t lower_bound_u64(const uint64_t *__restrict__ keys, uint32_t lo, uint32_t hi, uint64_t target) {
  while (lo < hi) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (keys[mid] < target) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// The entries of the bricks that exist: one thread per row (two searches over all keys for the ends of the row's stretch
// of B+3 cells, then a search inside that stretch -- usually empty -- per entry).
__global__ __launch_bounds__(O3D_BLOCK) void k_brick_fill(const uint64_t *__restrict__ keys, uint32_t n_grid, GridParams gp,
                                                           const uint32_t *__restrict__ list, int64_t n_bricks,
                                                           uint32_t *__restrict__ table) {
  const int b = gp.brick_bits, B = 1 << b, rows = (B + 2) * (B + 2), ne = B + 3;
  const int64_t total = n_bricks * rows;
  for (int64_t t = (int64_t)blockIdx.x * O3D_BLOCK + threadIdx.x; t < total; t += (int64_t)gridDim.x * O3D_BLOCK) {
    const int64_t k = t / rows;
    const int r = (int)(t - k * rows);
    const uint32_t id = list[k];
    const int bx = (int)(id % (uint32_t)gp.nbx), by = (int)((id / (uint32_t)gp.nbx) % (uint32_t)gp.nby),
              bz = (int)(id / ((uint32_t)gp.nbx * (uint32_t)gp.nby));
    const int gy = by * B + r % (B + 2) - 1, gz = bz * B + r / (B + 2) - 1;
    uint32_t *__restrict__ out = table + (size_t)(k + 1) * (size_t)(rows * ne) + (size_t)r * ne;
    if (gy < 0 || gy >= gp.ny || gz < 0 || gz >= gp.nz) {  // a row outside the grid: empty
      for (int e = 0; e < ne; e++) out[e] = 0u;
      continue;
    }
    const uint64_t k0 = ((uint64_t)gz * (uint64_t)gp.ny + (uint64_t)gy) * (uint64_t)gp.nxp + (uint64_t)(bx * B);
    const uint32_t lo = lower_bound_u64(keys, 0u, n_grid, k0), hi = lower_bound_u64(keys, lo, n_grid, k0 + (uint64_t)(ne - 1));
    out[0] = lo;
    uint32_t at = lo;
    for (int e = 1; e < ne - 1; e++) {
      at = lower_bound_u64(keys, at, hi, k0 + (uint64_t)e);
      out[e] = at;
    }
    out[ne - 1] = hi;
  }
}

// Spread the low 21 bits of v so that bit i lands at bit 3*i.
__device__ __forceinline__ uint64_t spread3(uint32_t v) {
  uint64_t x = v & 0x1fffffu;
  x = (x | x << 32) & 0x1f00000000ffffull;
  x = (x | x << 16) & 0x1f0000ff0000ffull;
  x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}

// 3-D Hilbert index of a cell (Skilling's transpose form, `bits` bits per axis): consecutive indices are face-adjacent
// cells, so a run of consecutive source points stays inside a small box even where a Z-order curve would jump.
__device__ __forceinline__ uint64_t hilbert3(uint32_t x0, uint32_t x1, uint32_t x2, int bits) {
  uint32_t X[3] = {x0, x1, x2};
  const uint32_t M = 1u << (bits - 1);
  for (uint32_t Q = M; Q > 1; Q >>= 1) {
    const uint32_t P = Q - 1;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      if (X[i] & Q) {
        X[0] ^= P;
      } else {
        const uint32_t t = (X[0] ^ X[i]) & P;
        X[0] ^= t;
        X[i] ^= t;
      }
    }
  }
  X[1] ^= X[0];
  X[2] ^= X[1];
  uint32_t t = 0;
  for (uint32_t Q = M; Q > 1; Q >>= 1)
    if (X[2] & Q) t ^= Q - 1;
  X[0] ^= t; X[1] ^= t; X[2] ^= t;
  return (spread3(X[0]) << 2) | (spread3(X[1]) << 1) | spread3(X[2]);
}

// Source ordering key: Hilbert index of the target-grid cell of the (init-transformed) source point, clamped into the
// grid.  Only the processing ORDER depends on it -- consecutive points form compact patches of cells, which is what
// the tile pass stages into LDS; results are scattered back by original index.
struct InitT {  // the initial transformation, column-major, as a kernel argument (no upload of its own)
  double m[16];
};
template <typename KeyT>  // (uint32_t with a job tag on top: the common sort of a gang)
__global__ __launch_bounds__(O3D_BLOCK) void k_source_keys(const double *__restrict__ xyz, int64_t n, GridParams gp,
                                                            InitT Tm, int apply, int bits, int drop,
                                                            KeyT *__restrict__ keys, uint32_t *__restrict__ vals, KeyT tag) {
  const double *T = Tm.m;
  for (int64_t i = (int64_t)blockIdx.x * O3D_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * O3D_BLOCK) {
    double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    if (apply) {
      double tx = ((T[0] * x + T[4] * y) + T[8] * z) + T[12];
      double ty = ((T[1] * x + T[5] * y) + T[9] * z) + T[13];
      double tz = ((T[2] * x + T[6] * y) + T[10] * z) + T[14];
      x = tx; y = ty; z = tz;
    }
    int cx = min(max(o3d_cell_coord(x, gp.gmin[0], gp.inv_h, gp.nx), 0), gp.nx - 1);
    int cy = min(max(o3d_cell_coord(y, gp.gmin[1], gp.inv_h, gp.ny), 0), gp.ny - 1);
    int cz = min(max(o3d_cell_coord(z, gp.gmin[2], gp.inv_h, gp.nz), 0), gp.nz - 1);
    keys[i] = (KeyT)(hilbert3((uint32_t)cx, (uint32_t)cy, (uint32_t)cz, bits) >> drop) | tag;
    vals[i] = (uint32_t)i;
  }
}

__global__ __launch_bounds__(O3D_BLOCK) void k_gather_source(const double *__restrict__ xyz, const uint32_t *__restrict__ vals,
                                                              int64_t n, double *__restrict__ src_sorted,
                                                              uint32_t *__restrict__ pos_sorted, float *__restrict__ lb_sorted,
                                                              uint32_t *__restrict__ perm_out) {
  for (int64_t p = (int64_t)blockIdx.x * O3D_BLOCK + threadIdx.x; p < n; p += (int64_t)gridDim.x * O3D_BLOCK) {
    const size_t i = vals[p];
    if (perm_out) perm_out[p] = (uint32_t)i;  // (the permutation came out of a gang's common sort: keep a copy of our own)
    pos_sorted[p] = 0xFFFFFFFFu;  // no previous match ...
    lb_sorted[p] = 0.f;           // ... and no bound (two memsets less per set-up)
    src_sorted[3 * p] = xyz[3 * i];
    src_sorted[3 * p + 1] = xyz[3 * i + 1];
    src_sorted[3 * p + 2] = xyz[3 * i + 2];
  }
}

int grid_for(int64_t n) {
  int64_t g = (n + O3D_BLOCK - 1) / O3D_BLOCK;
  if (g < 1) g = 1;
  if (g > O3D_MAX_GRID) g = O3D_MAX_GRID;
  return (int)g;
}

int ceil_log2(uint64_t v) {
  int b = 0;
  while ((1ull << b) < v) b++;
  return b;
}

}  // namespace

// The bounding box ahead of time (o3d_registration_icp_batch): kernels and the copy to the host are enqueued, nobody waits.
// The o3d_bounds call for the same cloud then finds the result in the pinned page -- a gang of set-ups pays one wait for the
// GPU instead of one each (a wait is a trip through the queue behind other threads' pass kernels: 1 ms per set-up with eight
// worker threads).
static constexpr size_t kBoundsAheadSlot = 768;  // byte offset in ctx->pinned
int o3d_bounds_enqueue(o3d_ctx *ctx, const double *dev_xyz, int64_t n) {
  ctx->bounds_for = nullptr;
  if (n <= 0 || !dev_xyz) return O3D_OK;
  const int g = grid_for(n);
  void *buf;
  O3D_TRY(o3d_scratch(ctx, 0, sizeof(Bounds6) * (size_t)(g + 1), &buf));
  Bounds6 *part = (Bounds6 *)buf;
  hipLaunchKernelGGL(k_bounds_partial, dim3(g), dim3(O3D_BLOCK), 0, ctx->stream, dev_xyz, n, part);
  hipLaunchKernelGGL(k_bounds_final, dim3(1), dim3(O3D_BLOCK), 0, ctx->stream, part, g, part + g);
  O3D_HIP(ctx, hipMemcpyAsync((char *)ctx->pinned + kBoundsAheadSlot, part + g, sizeof(Bounds6), hipMemcpyDeviceToHost, ctx->stream));
  ctx->bounds_for = dev_xyz;
  ctx->bounds_n = n;
  return O3D_OK;
}

int o3d_bounds(o3d_ctx *ctx, const double *dev_xyz, int64_t n, double out6[6], int64_t *out_finite) {
  if (ctx->bounds_for && ctx->bounds_for == dev_xyz && ctx->bounds_n == n) {  // enqueued ahead, on this stream
    ctx->bounds_for = nullptr;
    O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const Bounds6 *h = (const Bounds6 *)((const char *)ctx->pinned + kBoundsAheadSlot);
    for (int d = 0; d < 3; d++) { out6[d] = h->mn[d]; out6[3 + d] = h->mx[d]; }
    *out_finite = (int64_t)h->finite;
    return O3D_OK;
  }
  ctx->bounds_for = nullptr;
  const int g = grid_for(n);
  void *buf;
  O3D_TRY(o3d_scratch(ctx, 0, sizeof(Bounds6) * (size_t)(g + 1), &buf));
  Bounds6 *part = (Bounds6 *)buf;
  hipLaunchKernelGGL(k_bounds_partial, dim3(g), dim3(O3D_BLOCK), 0, ctx->stream, dev_xyz, n, part);
  hipLaunchKernelGGL(k_bounds_final, dim3(1), dim3(O3D_BLOCK), 0, ctx->stream, part, g, part + g);
  Bounds6 *h = (Bounds6 *)ctx->pinned;
  O3D_HIP(ctx, hipMemcpyAsync(h, part + g, sizeof(Bounds6), hipMemcpyDeviceToHost, ctx->stream));
  O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int d = 0; d < 3; d++) { out6[d] = h->mn[d]; out6[3 + d] = h->mx[d]; }
  *out_finite = (int64_t)h->finite;
  return O3D_OK;
}

// Cell size and table form for a target with bounding box bb: cells of side max_dist in a dense table when that fits the
// budget; else, where the caller can read it, bricks (the smallest whose directory stays small); else coarser cells
// (h *= 1.25 until the dense table fits -- still exact, only slower).
static void choose_grid(const double bb[6], int64_t nfinite, double max_dist, bool bricks, GridParams &gp, uint64_t &ndir) {
  double h = max_dist * kCellSlack;
  const double ext[3] = {bb[3] - bb[0], bb[4] - bb[1], bb[5] - bb[2]};
  const double budget = std::max(kCellsPerPointBudget * (double)nfinite, 16777216.0);
  ndir = 0;
  gp.brick_bits = 0;
  for (bool first = true;; first = false) {
    double c[3];
    double total = 1.0;
    // occupied cells per axis, plus a margin of empty cells: one either side in x, two either side in y and z.  A
    // query in the outermost y / z layer, or outside the grid, is then at least one cell size (>= the radius) from
    // every target point, and every other query has all 26 neighbour cells inside the table (no bounds checks).
    for (int d = 0; d < 3; d++) c[d] = floor(ext[d] / h) + 1.0 + (d == 0 ? 2.0 : 4.0);
    total = (c[0] + 2.0) * c[1] * c[2];
    static const bool force_bricks = getenv("O3D_FORCE_BRICKS") != nullptr;  // tuning aid
    if (total < (double)kMaxDenseCells && total <= budget && !(force_bricks && first && bricks)) {
      gp.nx = (int)c[0]; gp.ny = (int)c[1]; gp.nz = (int)c[2];
      break;
    }
    if (first && bricks && c[0] < kMaxBrickAxis && c[1] < kMaxBrickAxis && c[2] < kMaxBrickAxis) {
      int b = kMinBrickBits;
      for (; b <= kMaxBrickBits; b++) {
        const double B = (double)(1 << b);
        if ((floor(c[0] / B) + 1.0) * (floor(c[1] / B) + 1.0) * (floor(c[2] / B) + 1.0) <= (double)kMaxBrickDir) break;
      }
      if (b <= kMaxBrickBits) {
        gp.nx = (int)c[0]; gp.ny = (int)c[1]; gp.nz = (int)c[2];
        gp.brick_bits = b;
        gp.nbx = (gp.nx >> b) + 1; gp.nby = (gp.ny >> b) + 1;
        ndir = (uint64_t)gp.nbx * (uint64_t)gp.nby * (uint64_t)((gp.nz >> b) + 1);
        break;
      }
    }
    h *= 1.25;
  }
  gp.h = h;
  gp.inv_h = 1.0 / h;
  for (int d = 0; d < 3; d++) gp.gmin[d] = bb[d] - (d == 0 ? 1.0 : 2.0) * h;
  gp.margin = 1;
  gp.nxp = gp.nx + 2;
  gp.ncell = gp.brick_bits ? 0u : (uint32_t)((uint64_t)gp.nxp * (uint64_t)gp.ny * (uint64_t)gp.nz);
}

int o3d_build_target_grid(o3d_ctx *ctx, const double *dev_xyz, const double *dev_nrm, int64_t n, double max_dist, int allow_bricks) {
  ctx->n_tgt = n;
  ctx->n_tgt_grid = 0;
  ctx->n_bricks = 0;
  ctx->tn_valid = 0;
  GridParams gp{};
  gp.nx = gp.ny = gp.nz = 1;
  gp.nxp = 3;
  gp.h = max_dist * kCellSlack;
  gp.inv_h = 1.0 / gp.h;
  gp.ncell = 3;
  double bb[6] = {0, 0, 0, 0, 0, 0};
  int64_t nfinite = 0;
  if (n > 0) O3D_TRY(o3d_bounds(ctx, dev_xyz, n, bb, &nfinite));
  uint64_t ndir = 0;
  ctx->use_bricks = getenv("O3D_NO_BRICKS") ? 0 : 1;  // tuning aid
  if (nfinite > 0) choose_grid(bb, nfinite, max_dist, allow_bricks && ctx->use_bricks, gp, ndir);
  // the fp32 prefilter's error bound assumes fp64 cell origins are exact to well below fp32 resolution
  ctx->tile_ok = (gp.nx < (1 << 20) && gp.ny < (1 << 20) && gp.nz < (1 << 20)) ? 1 : 0;

  bool want_tn = false;
  if (n > 0) {
    O3D_TRY(o3d_reserve(ctx, (void **)&ctx->tgt_sorted, &ctx->cap_tgt, sizeof(double4) * ((size_t)n + 8)));  // (+8: k_estimate_normals_hist reads whole batches)
    // four entries of padding: k_icp_search_f32 reads candidates four at a time without clamping
    O3D_TRY(o3d_reserve(ctx, (void **)&ctx->tgt_local, &ctx->cap_local, sizeof(float4) * ((size_t)n + 4)));
    O3D_TRY(o3d_reserve(ctx, (void **)&ctx->tgt_index, &ctx->cap_tidx, sizeof(uint32_t) * (size_t)n));
    want_tn = dev_nrm && allow_bricks && !getenv("O3D_NO_TN");  // (the ICP set-up; the switch is a tuning aid)
    if (want_tn) O3D_TRY(o3d_reserve(ctx, (void **)&ctx->tn_sorted, &ctx->cap_tn, sizeof(double2) * 3 * (size_t)n));
    else if (dev_nrm) O3D_TRY(o3d_reserve(ctx, (void **)&ctx->nrm_sorted, &ctx->cap_nrm, sizeof(double) * 3 * (size_t)n));
  }
  const size_t ntiles = o3d_sort_num_tiles(n > 0 ? n : 1);
  const size_t nh = 256 * ntiles;
  const int g = grid_for(n > 0 ? n : 1);
  void *keys, *vals, *tkeys, *tvals, *hist;

  bool bricks_fit = true;
  if (gp.brick_bits) {
    const int B = 1 << gp.brick_bits;
    const uint32_t brick_entries = (uint32_t)((B + 2) * (B + 2) * (B + 3));
    O3D_TRY(o3d_reserve(ctx, (void **)&ctx->brick_dir, &ctx->cap_dir, sizeof(uint32_t) * (size_t)ndir));
    O3D_HIP(ctx, hipMemsetAsync(ctx->brick_dir, 0, sizeof(uint32_t) * (size_t)ndir, ctx->stream));
    gp.dir = ctx->brick_dir;
    O3D_TRY(o3d_scratch(ctx, 1, sizeof(uint64_t) * (size_t)n, &keys));
    O3D_TRY(o3d_scratch(ctx, 2, sizeof(uint32_t) * (size_t)n, &vals));
    O3D_TRY(o3d_scratch(ctx, 3, sizeof(uint64_t) * (size_t)n, &tkeys));
    O3D_TRY(o3d_scratch(ctx, 4, sizeof(uint32_t) * (size_t)n, &tvals));
    O3D_TRY(o3d_scratch(ctx, 5, sizeof(uint32_t) * (nh + o3d_scan_num_blocks((int64_t)nh) + 264), &hist));
    {
      const int bits = ceil_log2((uint64_t)gp.nxp * (uint64_t)gp.ny * (uint64_t)gp.nz + 1);
      const bool odd = o3d_radix_sort_odd(bits);
      uint64_t *k0 = (uint64_t *)(odd ? tkeys : keys), *k1 = (uint64_t *)(odd ? keys : tkeys);
      uint32_t *v0 = (uint32_t *)(odd ? tvals : vals), *v1 = (uint32_t *)(odd ? vals : tvals);
      hipLaunchKernelGGL(k_target_keys64, dim3(g), dim3(O3D_BLOCK), 0, ctx->stream, dev_xyz, n, gp, k0, v0);
      O3D_TRY(o3d_radix_sort_u64(ctx, k0, v0, k1, v1, (uint32_t *)hist, n, bits, false));
    }
    hipLaunchKernelGGL(k_brick_mark, dim3(grid_for(nfinite)), dim3(O3D_BLOCK), 0, ctx->stream, (const uint64_t *)keys, nfinite, gp,
                       ctx->brick_dir);
    // rank of every brick that exists (a copy of the flags, scanned), and their number
    void *rank, *bsums, *list;
    O3D_TRY(o3d_scratch(ctx, 6, sizeof(uint32_t) * ((size_t)ndir + 1), &rank));
    O3D_TRY(o3d_scratch(ctx, 7, sizeof(uint32_t) * (o3d_scan_num_blocks((int64_t)ndir + 1) + 8), &bsums));
    O3D_HIP(ctx, hipMemcpyAsync(rank, ctx->brick_dir, sizeof(uint32_t) * (size_t)ndir, hipMemcpyDeviceToDevice, ctx->stream));
    O3D_HIP(ctx, hipMemsetAsync((uint32_t *)rank + ndir, 0, sizeof(uint32_t), ctx->stream));
    O3D_TRY(o3d_exclusive_scan_u32(ctx, (uint32_t *)rank, (uint32_t *)bsums, (int64_t)ndir + 1));
    uint32_t *hcount = (uint32_t *)((char *)ctx->pinned + 512);
    O3D_HIP(ctx, hipMemcpyAsync(hcount, (uint32_t *)rank + ndir, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int64_t n_bricks = (int64_t)*hcount;
    // Points scattered thinly through the volume put nearly every point into a brick of its own: when the bricks would
    // hold more entries than the budget, the scene is not locally dense either, and coarser cells in a dense table serve.
    const double brick_budget = std::max(kBrickEntriesPerPoint * (double)nfinite, 134217728.0);
    if ((double)(n_bricks + 1) * (double)brick_entries > brick_budget || (uint64_t)(n_bricks + 1) * (uint64_t)brick_entries >= (1ull << 32)) {
      choose_grid(bb, nfinite, max_dist, false, gp, ndir);
      gp.dir = nullptr;
      ctx->tile_ok = (gp.nx < (1 << 20) && gp.ny < (1 << 20) && gp.nz < (1 << 20)) ? 1 : 0;
      bricks_fit = false;
    } else {
    ctx->n_bricks = n_bricks;
    O3D_TRY(o3d_scratch(ctx, 0, sizeof(uint32_t) * (size_t)(n_bricks + 1), &list));
    const size_t ntab = (size_t)(n_bricks + 1) * (size_t)brick_entries;
    O3D_TRY(o3d_reserve(ctx, (void **)&ctx->cell_start, &ctx->cap_cells, sizeof(uint32_t) * (ntab + 8)));
    O3D_HIP(ctx, hipMemsetAsync(ctx->cell_start, 0, sizeof(uint32_t) * (size_t)brick_entries, ctx->stream));  // the empty brick
    O3D_HIP(ctx, hipMemsetAsync(ctx->cell_start + ntab, 0, sizeof(uint32_t) * 8, ctx->stream));
    hipLaunchKernelGGL(k_brick_dir, dim3(grid_for((int64_t)ndir)), dim3(O3D_BLOCK), 0, ctx->stream, ctx->brick_dir, (const uint32_t *)rank,
                       (int64_t)ndir, brick_entries, (uint32_t *)list);
    if (n_bricks > 0)
      hipLaunchKernelGGL(k_brick_fill, dim3(grid_for(n_bricks * (B + 2) * (B + 2))), dim3(O3D_BLOCK), 0, ctx->stream, (const uint64_t *)keys,
                         (uint32_t)nfinite, gp, (const uint32_t *)list, n_bricks, ctx->cell_start);
    ctx->gp = gp;
    }
  }
  if (!gp.brick_bits || !bricks_fit) {
    ctx->gp = gp;
    // dense table: ncell + 1 prefix entries (+1 sentinel bucket for non-finite points), padded for uint4 access
    const size_t ntab = (size_t)gp.ncell + 2;
    O3D_TRY(o3d_reserve(ctx, (void **)&ctx->cell_start, &ctx->cap_cells, sizeof(uint32_t) * (ntab + 8)));
    if (n <= 0) {
      O3D_HIP(ctx, hipMemsetAsync(ctx->cell_start, 0, sizeof(uint32_t) * (ntab + 8), ctx->stream));
      return O3D_OK;
    }
    // (the eight padding entries behind the table are written by k_cell_table)
    O3D_TRY(o3d_scratch(ctx, 1, sizeof(uint32_t) * (size_t)n, &keys));
    O3D_TRY(o3d_scratch(ctx, 2, sizeof(uint32_t) * (size_t)n, &vals));
    O3D_TRY(o3d_scratch(ctx, 3, sizeof(uint32_t) * (size_t)n, &tkeys));
    O3D_TRY(o3d_scratch(ctx, 4, sizeof(uint32_t) * (size_t)n, &tvals));
    O3D_TRY(o3d_scratch(ctx, 5, sizeof(uint32_t) * (nh + o3d_scan_num_blocks((int64_t)nh) + 264), &hist));
    const int key_bits = ceil_log2((uint64_t)gp.ncell + 1);
    if (o3d_gang_sort *gs = ctx->gang_sort; gs && nfinite > 0 && key_bits <= O3D_GANG_KEY_BITS && gs->jobs.size() < 64 &&
                                            gs->used + n <= (int64_t)gs->cap) {
      // the gang sorts for us: keys with our job number on top into its buffer; table and gather wait behind its sort
      const int64_t off = gs->used;
      const uint32_t tag = (uint32_t)gs->jobs.size() << O3D_GANG_KEY_BITS;
      gs->used += n;
      hipLaunchKernelGGL(k_target_keys, dim3(g), dim3(O3D_BLOCK), 0, ctx->stream, dev_xyz, n, gp, gs->keys + off, gs->vals + off, tag);
      O3D_HIP(ctx, hipGetLastError());
      const size_t ntab_c = ntab;
      const bool tn_c = want_tn;
      gs->jobs.push_back({ctx, off, n, [=](const uint32_t *skeys, const uint32_t *svals) -> int {
        const uint32_t mask = (1u << O3D_GANG_KEY_BITS) - 1u;
        if (ntab_c > (4u << 20))
          hipLaunchKernelGGL(k_cell_table<kCtCells>, dim3((unsigned)((ntab_c + kCtCells - 1) / kCtCells)), dim3(O3D_BLOCK), 0, ctx->stream,
                             skeys, (uint32_t)n, ctx->cell_start, (uint64_t)ntab_c, mask);
        else
          hipLaunchKernelGGL(k_cell_table<kCtCellsSmall>, dim3((unsigned)((ntab_c + kCtCellsSmall - 1) / kCtCellsSmall)), dim3(O3D_BLOCK),
                             0, ctx->stream, skeys, (uint32_t)n, ctx->cell_start, (uint64_t)ntab_c, mask);
        hipLaunchKernelGGL(k_gather_target, dim3(grid_for(nfinite)), dim3(O3D_BLOCK), 0, ctx->stream, dev_xyz, dev_nrm, svals, nfinite, gp,
                           ctx->tgt_sorted, dev_nrm ? ctx->nrm_sorted : nullptr, ctx->tgt_local, tn_c ? ctx->tn_sorted : nullptr, ctx->tgt_index, n);
        O3D_HIP(ctx, hipGetLastError());
        return O3D_OK;
      }});
      ctx->n_tgt_grid = nfinite;
      ctx->tn_valid = want_tn ? 1 : 0;
      return O3D_OK;
    }
    {  // the keys start in whichever pair of buffers makes the sorted result end in (keys, vals): no copy home
      const int bits = key_bits;
      const bool odd = o3d_radix_sort_odd(bits);
      uint32_t *k0 = (uint32_t *)(odd ? tkeys : keys), *v0 = (uint32_t *)(odd ? tvals : vals);
      uint32_t *k1 = (uint32_t *)(odd ? keys : tkeys), *v1 = (uint32_t *)(odd ? vals : tvals);
      hipLaunchKernelGGL(k_target_keys, dim3(g), dim3(O3D_BLOCK), 0, ctx->stream, dev_xyz, n, gp, k0, v0, 0u);
      O3D_TRY(o3d_radix_sort_u32(ctx, k0, v0, k1, v1, (uint32_t *)hist, n, bits, false));
    }
    if (ntab > (4u << 20))
      hipLaunchKernelGGL(k_cell_table<kCtCells>, dim3((unsigned)((ntab + kCtCells - 1) / kCtCells)), dim3(O3D_BLOCK), 0, ctx->stream,
                         (const uint32_t *)keys, (uint32_t)n, ctx->cell_start, (uint64_t)ntab, 0xFFFFFFFFu);
    else
      hipLaunchKernelGGL(k_cell_table<kCtCellsSmall>, dim3((unsigned)((ntab + kCtCellsSmall - 1) / kCtCellsSmall)), dim3(O3D_BLOCK), 0,
                         ctx->stream, (const uint32_t *)keys, (uint32_t)n, ctx->cell_start, (uint64_t)ntab, 0xFFFFFFFFu);
  }
  ctx->n_tgt_grid = nfinite;
  if (nfinite > 0)
    hipLaunchKernelGGL(k_gather_target, dim3(grid_for(nfinite)), dim3(O3D_BLOCK), 0, ctx->stream, dev_xyz, dev_nrm,
                       (const uint32_t *)vals, nfinite, gp, ctx->tgt_sorted, dev_nrm ? ctx->nrm_sorted : nullptr,
                       ctx->tgt_local, want_tn ? ctx->tn_sorted : nullptr, ctx->tgt_index, n);
  ctx->tn_valid = (want_tn && nfinite > 0) ? 1 : 0;
  O3D_HIP(ctx, hipGetLastError());
  return O3D_OK;
}

int o3d_sort_source(o3d_ctx *ctx, const double *dev_xyz, int64_t n, const double *init_colmajor, int apply_init) {
  ctx->n_src = n;
  const size_t nn = (size_t)(n > 0 ? n : 1);
  if (nn > ctx->cap_src) {
    if (ctx->src_sorted) (void)hipFree(ctx->src_sorted);
    if (ctx->src_perm) (void)hipFree(ctx->src_perm);
    if (ctx->corr_sorted) (void)hipFree(ctx->corr_sorted);
    if (ctx->corr_out) (void)hipFree(ctx->corr_out);
    if (ctx->pos_sorted) (void)hipFree(ctx->pos_sorted);
    if (ctx->lb_sorted) (void)hipFree(ctx->lb_sorted);
    if (ctx->work) (void)hipFree(ctx->work);
    ctx->pos_sorted = nullptr;
    ctx->lb_sorted = nullptr;
    ctx->work = nullptr;
    ctx->src_sorted = nullptr; ctx->src_perm = nullptr; ctx->corr_sorted = nullptr; ctx->corr_out = nullptr;
    ctx->cap_src = 0;
    O3D_HIP(ctx, hipMalloc((void **)&ctx->src_sorted, sizeof(double) * 3 * nn));
    O3D_HIP(ctx, hipMalloc((void **)&ctx->src_perm, sizeof(uint32_t) * nn));
    O3D_HIP(ctx, hipMalloc((void **)&ctx->corr_sorted, sizeof(int32_t) * nn));
    O3D_HIP(ctx, hipMalloc((void **)&ctx->corr_out, sizeof(int32_t) * nn));
    O3D_HIP(ctx, hipMalloc((void **)&ctx->pos_sorted, sizeof(uint32_t) * nn));
    O3D_HIP(ctx, hipMalloc((void **)&ctx->lb_sorted, sizeof(float) * nn));
    // k_icp_cert's chunks are rounded up to whole wavefronts: a segment may start (not reach) beyond n
    // (one list per wavefront of the largest grid, each rounded up to whole blocks of 64: icp.hip, plan_pass)
    O3D_HIP(ctx, hipMalloc((void **)&ctx->work, sizeof(uint32_t) * (nn + (size_t)O3D_WAVE * (4 * O3D_MAX_GRID + 8))));
    ctx->cap_src = nn;
  }
  if (n <= 0) return O3D_OK;
  const size_t ntiles = o3d_sort_num_tiles(n);
  const size_t nh = 256 * ntiles;
  void *keys, *tkeys, *tvals, *hist;
  O3D_TRY(o3d_scratch(ctx, 1, sizeof(uint64_t) * (size_t)n, &keys));
  O3D_TRY(o3d_scratch(ctx, 3, sizeof(uint64_t) * (size_t)n, &tkeys));
  O3D_TRY(o3d_scratch(ctx, 4, sizeof(uint32_t) * (size_t)n, &tvals));
  O3D_TRY(o3d_scratch(ctx, 5, sizeof(uint32_t) * (nh + o3d_scan_num_blocks((int64_t)nh) + 264), &hist));
  InitT Tm;
  for (int i = 0; i < 16; i++) Tm.m[i] = init_colmajor[i];
  const int g = grid_for(n);
  int axis_bits = ceil_log2((uint64_t)std::max(std::max(ctx->gp.nx, ctx->gp.ny), ctx->gp.nz));
  axis_bits = std::min(std::max(axis_bits, 1), 21);
  // One or two key bits over a multiple of the 8-bit digit would cost a whole radix pass: they are dropped (two or four
  // neighbouring cells of the curve share a key and keep their given order; only the processing ORDER depends on the key).
  const int over = (3 * axis_bits) % 8;
  int drop = (3 * axis_bits > 8 && over >= 1 && over <= 2) ? over : 0;
  if (const char *e = getenv("O3D_SRC_KEY_BITS")) {  // tuning aid: a coarser processing order for fewer radix passes
    const int want = atoi(e);
    if (want >= 8 && 3 * axis_bits > want) drop = 3 * axis_bits - want;
  }
  if (o3d_gang_sort *gs = ctx->gang_sort; gs && 3 * axis_bits - drop <= O3D_GANG_KEY_BITS && gs->jobs.size() < 64 &&
                                          gs->used + n <= (int64_t)gs->cap) {
    const int64_t off = gs->used;
    const uint32_t tag = (uint32_t)gs->jobs.size() << O3D_GANG_KEY_BITS;
    gs->used += n;
    hipLaunchKernelGGL(k_source_keys<uint32_t>, dim3(g), dim3(O3D_BLOCK), 0, ctx->stream, dev_xyz, n, ctx->gp, Tm, apply_init, axis_bits,
                       drop, gs->keys + off, gs->vals + off, tag);
    O3D_HIP(ctx, hipGetLastError());
    gs->jobs.push_back({ctx, off, n, [=](const uint32_t *, const uint32_t *svals) -> int {
      hipLaunchKernelGGL(k_gather_source, dim3(g), dim3(O3D_BLOCK), 0, ctx->stream, dev_xyz, svals, n, ctx->src_sorted, ctx->pos_sorted,
                         ctx->lb_sorted, ctx->src_perm);
      O3D_HIP(ctx, hipGetLastError());
      return O3D_OK;
    }});
    return O3D_OK;
  }
  if (3 * axis_bits - drop <= 32) {  // (the common case: 32-bit keys, two thirds of the bytes per radix pass)
    const int bits = 3 * axis_bits - drop;
    const bool odd = o3d_radix_sort_odd(bits);
    uint32_t *k0 = (uint32_t *)(odd ? tkeys : keys), *k1 = (uint32_t *)(odd ? keys : tkeys);
    uint32_t *v0 = odd ? (uint32_t *)tvals : ctx->src_perm, *v1 = odd ? ctx->src_perm : (uint32_t *)tvals;
    hipLaunchKernelGGL(k_source_keys<uint32_t>, dim3(g), dim3(O3D_BLOCK), 0, ctx->stream, dev_xyz, n, ctx->gp, Tm, apply_init, axis_bits,
                       drop, k0, v0, 0u);
    O3D_TRY(o3d_radix_sort_u32(ctx, k0, v0, k1, v1, (uint32_t *)hist, n, bits, false));
  } else {  // the permutation must END in src_perm: with an odd number of radix passes it starts in the other buffer (no copy home)
    const int bits = 3 * axis_bits - drop;
    const bool odd = o3d_radix_sort_odd(bits);
    uint64_t *k0 = (uint64_t *)(odd ? tkeys : keys), *k1 = (uint64_t *)(odd ? keys : tkeys);
    uint32_t *v0 = odd ? (uint32_t *)tvals : ctx->src_perm, *v1 = odd ? ctx->src_perm : (uint32_t *)tvals;
    hipLaunchKernelGGL(k_source_keys<uint64_t>, dim3(g), dim3(O3D_BLOCK), 0, ctx->stream, dev_xyz, n, ctx->gp, Tm, apply_init, axis_bits,
                       drop, k0, v0, (uint64_t)0);
    O3D_TRY(o3d_radix_sort_u64(ctx, k0, v0, k1, v1, (uint32_t *)hist, n, bits, false));
  }
  hipLaunchKernelGGL(k_gather_source, dim3(g), dim3(O3D_BLOCK), 0, ctx->stream, dev_xyz, (const uint32_t *)ctx->src_perm, n,
                     ctx->src_sorted, ctx->pos_sorted, ctx->lb_sorted, (uint32_t *)nullptr);
  O3D_HIP(ctx, hipGetLastError());
  return O3D_OK;
}
