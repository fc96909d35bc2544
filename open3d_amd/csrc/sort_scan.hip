// LDS-staged stable radix sort (8-bit digits) and device-wide exclusive scan for gfx950.
//
// Both are HBM-bound integer kernels.  The sort ranks a 4096-key tile per 256-thread workgroup:
// each wave owns a contiguous 1024-key slice, ranks its rows of 64 keys with 8 ballots per row
// (wave64 match-any) against wave-private digit counters in LDS, and the four waves' counters are
// combined in LDS to give each key its stable destination.  The per-tile digit counts are laid out
// digit-major so that one exclusive scan over the table yields every (digit, tile) base.
#include "o3d_internal.h"

namespace {

constexpr int kSortItems = 16;
constexpr int kSortTile = O3D_BLOCK * kSortItems;  // 4096 keys per workgroup
constexpr int kWaves = O3D_BLOCK / O3D_WAVE;       // 4
constexpr int kScanVec = 4;
constexpr int kScanRounds = 8;
constexpr int kScanTile = O3D_BLOCK * kScanVec * kScanRounds;  // 8192

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int lane) {
#pragma unroll
  for (int d = 1; d < O3D_WAVE; d <<= 1) {
    uint32_t o = __shfl_up(v, d, O3D_WAVE);
    if (lane >= d) v += o;
  }
  return v;
}

// Exclusive scan of one value per thread over a 256-thread block; returns the exclusive prefix and the block total.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *wave_tot /* LDS[kWaves] */, uint32_t *total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = wave_inclusive_scan(v, lane);
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kWaves; w++) {
    uint32_t t = wave_tot[w];
    if (w < wave) base += t;
    tot += t;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

// ---------------------------------------------------------------- radix sort
template <typename KeyT>
__global__ __launch_bounds__(O3D_BLOCK) void k_sort_hist(const KeyT *__restrict__ keys, uint32_t *__restrict__ hist,
                                                          int64_t n, int shift, uint32_t ntiles) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kSortTile;
#pragma unroll
  for (int it = 0; it < kSortItems; it++) {
    int64_t i = base + it * O3D_BLOCK + threadIdx.x;
    if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  hist[(size_t)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

// The digit-major table of per-tile counts, one workgroup per digit row: exclusive scan of the row in place, the row's total
// to row_tot[digit].  The scatter kernel adds the totals of the lower digits itself (a 256-entry scan in LDS), so a radix
// pass is three launches -- histogram, this, scatter -- instead of five (the device-wide scan was three by itself).
__global__ __launch_bounds__(O3D_BLOCK) void k_sort_rowscan(uint32_t *__restrict__ hist, uint32_t *__restrict__ row_tot, uint32_t ntiles) {
  __shared__ uint32_t wt[kWaves];
  uint32_t *row = hist + (size_t)blockIdx.x * ntiles;
  uint32_t carry = 0;
  for (uint32_t base = 0; base < ntiles; base += O3D_BLOCK * 4) {  // four consecutive tiles per thread
    const uint32_t i = base + threadIdx.x * 4;
    uint32_t v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = i + j < ntiles ? row[i + j] : 0u;
    uint32_t tot;
    const uint32_t ex = carry + block_exclusive_scan(v[0] + v[1] + v[2] + v[3], wt, &tot);
    uint32_t run = ex;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (i + j < ntiles) row[i + j] = run;
      run += v[j];
    }
    carry += tot;
  }
  if (threadIdx.x == 0) row_tot[blockIdx.x] = carry;
}

// FUSED (few tiles): the table holds the raw per-tile counts and every workgroup derives its 256 digit bases from them
// itself (a row sum and a partial row sum per digit, one workgroup scan) -- three scan launches fewer per pass, which is
// most of the time of a pass over a small cloud.
constexpr uint32_t kFusedTiles = 64;
template <typename KeyT, bool FUSED>
__global__ __launch_bounds__(O3D_BLOCK) void k_sort_scatter(const KeyT *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                             KeyT *__restrict__ out_keys, uint32_t *__restrict__ out_vals,
                                                             const uint32_t *__restrict__ hist_scanned, int64_t n, int shift,
                                                             uint32_t ntiles) {
  __shared__ uint32_t wcnt[kWaves][256];  // per-wave digit counters, then exclusive-over-waves offsets
  __shared__ uint32_t gbase[256];
  __shared__ uint32_t dbase[256];         // first place of a digit in the tile's own sorted order
  __shared__ uint32_t wtot[kWaves];
  __shared__ KeyT skey[kSortTile];        // the tile in digit order: neighbouring lanes then write neighbouring words
  __shared__ uint16_t sidx[kSortTile];    // where in the tile the key came from (its value is fetched from there)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int w = 0; w < kWaves; w++) wcnt[w][threadIdx.x] = 0;
  if (FUSED) {
    __shared__ uint32_t wt[kWaves];
    const uint32_t *row = hist_scanned + (size_t)threadIdx.x * ntiles;
    uint32_t total = 0, mine = 0;
#pragma unroll 8
    for (uint32_t t = 0; t < ntiles; t++) {
      const uint32_t c = row[t];
      total += c;
      mine += t < blockIdx.x ? c : 0u;
    }
    uint32_t all;
    gbase[threadIdx.x] = block_exclusive_scan(total, wt, &all) + mine;  // (ends with a workgroup barrier)
  } else {  // the row's own prefix (k_sort_rowscan) + the totals of the lower digits (behind the table)
    __shared__ uint32_t wt2[kWaves];
    uint32_t all;
    const uint32_t lower = block_exclusive_scan(hist_scanned[(size_t)256 * ntiles + threadIdx.x], wt2, &all);
    gbase[threadIdx.x] = lower + hist_scanned[(size_t)threadIdx.x * ntiles + blockIdx.x];
  }
  __syncthreads();

  const int64_t wbase = (int64_t)blockIdx.x * kSortTile + (int64_t)wave * (O3D_WAVE * kSortItems);
  KeyT k[kSortItems];
  uint32_t rank[kSortItems];
  const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int it = 0; it < kSortItems; it++) {
    const int64_t i = wbase + it * O3D_WAVE + lane;
    const bool valid = i < n;
    k[it] = valid ? keys[i] : (KeyT)0;
    const uint32_t d = (uint32_t)(k[it] >> shift) & 255u;
    // lanes of this row holding the same digit (wave64 match-any by 8 ballots); invalid lanes match nobody
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const uint64_t bal = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? bal : ~bal;
    }
    uint32_t before = 0;
    if (valid) {
      before = wcnt[wave][d];
    }
    __builtin_amdgcn_wave_barrier();
    if (valid && (peers & lt_mask) == 0) wcnt[wave][d] = before + (uint32_t)__popcll(peers);
    __builtin_amdgcn_wave_barrier();
    rank[it] = before + (uint32_t)__popcll(peers & lt_mask);
  }
  __syncthreads();
  {  // exclusive prefix over the four waves, per digit (thread t owns digit t), and over the digits of the tile
    uint32_t run = 0;
#pragma unroll
    for (int w = 0; w < kWaves; w++) {
      uint32_t c = wcnt[w][threadIdx.x];
      wcnt[w][threadIdx.x] = run;
      run += c;
    }
    uint32_t all;
    dbase[threadIdx.x] = block_exclusive_scan(run, wtot, &all);  // (ends with a workgroup barrier)
  }
  __syncthreads();
  // Straight from the registers, neighbouring lanes would write to 64 different digit segments of the output (a pass over
  // the low digits of unordered keys ran at a quarter of the speed of one over already ordered digits).  The tile is put
  // into digit order in LDS first; a digit's run then leaves as consecutive words.
#pragma unroll
  for (int it = 0; it < kSortItems; it++) {
    const int64_t i = wbase + it * O3D_WAVE + lane;
    if (i < n) {
      const uint32_t d = (uint32_t)(k[it] >> shift) & 255u;
      const uint32_t lp = dbase[d] + wcnt[wave][d] + rank[it];
      skey[lp] = k[it];
      sidx[lp] = (uint16_t)(wave * (O3D_WAVE * kSortItems) + it * O3D_WAVE + lane);
    }
  }
  __syncthreads();
  const int64_t tile0 = (int64_t)blockIdx.x * kSortTile;
  const uint32_t count = (uint32_t)(n - tile0 < (int64_t)kSortTile ? n - tile0 : (int64_t)kSortTile);
#pragma unroll
  for (int it = 0; it < kSortItems; it++) {
    const uint32_t j = (uint32_t)it * O3D_BLOCK + threadIdx.x;
    if (j < count) {
      const KeyT key = skey[j];
      const uint32_t d = (uint32_t)(key >> shift) & 255u;
      const uint32_t pos = gbase[d] + (j - dbase[d]);
      out_keys[pos] = key;
      out_vals[pos] = vals[tile0 + sidx[j]];
    }
  }
}

// ---------------------------------------------------------------- exclusive scan
__global__ __launch_bounds__(O3D_BLOCK) void k_scan_reduce(const uint32_t *__restrict__ data, uint32_t *__restrict__ block_sums,
                                                            int64_t n) {
  __shared__ uint32_t wt[kWaves];
  const int64_t base = (int64_t)blockIdx.x * kScanTile;
  uint32_t s = 0;
#pragma unroll
  for (int r = 0; r < kScanRounds; r++) {
    int64_t i = base + ((int64_t)r * O3D_BLOCK + threadIdx.x) * kScanVec;
    if (i + kScanVec <= n) {
      uint4 v = *reinterpret_cast<const uint4 *>(data + i);
      s += v.x + v.y + v.z + v.w;
    } else {
      for (int j = 0; j < kScanVec; j++)
        if (i + j < n) s += data[i + j];
    }
  }
  uint32_t tot;
  block_exclusive_scan(s, wt, &tot);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// One workgroup scans the block sums in place (exclusive), chunk by chunk with a running carry.
__global__ __launch_bounds__(O3D_BLOCK) void k_scan_block_sums(uint32_t *__restrict__ block_sums, uint32_t nblocks) {
  __shared__ uint32_t wt[kWaves];
  uint32_t carry = 0;
  for (uint32_t base = 0; base < nblocks; base += O3D_BLOCK) {
    uint32_t i = base + threadIdx.x;
    uint32_t v = i < nblocks ? block_sums[i] : 0u;
    uint32_t tot;
    uint32_t ex = block_exclusive_scan(v, wt, &tot);
    if (i < nblocks) block_sums[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) block_sums[nblocks] = carry;  // grand total
}

__global__ __launch_bounds__(O3D_BLOCK) void k_scan_apply(uint32_t *__restrict__ data, const uint32_t *__restrict__ block_sums,
                                                           int64_t n) {
  __shared__ uint32_t wt[kWaves];
  const int64_t base = (int64_t)blockIdx.x * kScanTile;
  uint32_t carry = block_sums[blockIdx.x];
#pragma unroll
  for (int r = 0; r < kScanRounds; r++) {
    int64_t i = base + ((int64_t)r * O3D_BLOCK + threadIdx.x) * kScanVec;
    uint32_t v[kScanVec] = {0, 0, 0, 0};
    const bool full = i + kScanVec <= n;
    if (full) {
      uint4 q = *reinterpret_cast<const uint4 *>(data + i);
      v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
      for (int j = 0; j < kScanVec; j++)
        if (i + j < n) v[j] = data[i + j];
    }
    uint32_t s = v[0] + v[1] + v[2] + v[3];
    uint32_t tot;
    uint32_t ex = carry + block_exclusive_scan(s, wt, &tot);
    uint32_t o0 = ex, o1 = o0 + v[0], o2 = o1 + v[1], o3 = o2 + v[2];
    if (full) {
      *reinterpret_cast<uint4 *>(data + i) = make_uint4(o0, o1, o2, o3);
    } else {
      uint32_t o[kScanVec] = {o0, o1, o2, o3};
      for (int j = 0; j < kScanVec; j++)
        if (i + j < n) data[i + j] = o[j];
    }
    carry += tot;
  }
}

template <typename KeyT>
int radix_sort_impl(o3d_ctx *ctx, KeyT *keys, uint32_t *vals, KeyT *tmp_keys, uint32_t *tmp_vals, uint32_t *hist, int64_t n,
                    int bits, bool copy_home) {
  if (n <= 0) return O3D_OK;
  if (n >= (int64_t)1 << 31) return o3d_fail(ctx, O3D_ERR_UNSUPPORTED, "radix sort: n=%lld exceeds 2^31-1", (long long)n);
  const uint32_t ntiles = (uint32_t)o3d_sort_num_tiles(n);
  const int64_t nh = (int64_t)256 * ntiles;
  uint32_t *bsums = hist + nh;  // the 256 row totals: callers reserve o3d_scan_num_blocks(nh) + 264 words behind the table
  KeyT *ka = keys, *kb = tmp_keys;
  uint32_t *va = vals, *vb = tmp_vals;
  int passes = (bits + 7) / 8;
  if (passes < 1) passes = 1;
  for (int p = 0; p < passes; p++) {
    const int shift = 8 * p;
    hipLaunchKernelGGL(k_sort_hist<KeyT>, dim3(ntiles), dim3(O3D_BLOCK), 0, ctx->stream, ka, hist, n, shift, ntiles);
    if (ntiles <= kFusedTiles) {
      hipLaunchKernelGGL((k_sort_scatter<KeyT, true>), dim3(ntiles), dim3(O3D_BLOCK), 0, ctx->stream, ka, va, kb, vb, hist, n,
                         shift, ntiles);
    } else {
      hipLaunchKernelGGL(k_sort_rowscan, dim3(256), dim3(O3D_BLOCK), 0, ctx->stream, hist, bsums, ntiles);  // bsums = hist + 256 * ntiles
      hipLaunchKernelGGL((k_sort_scatter<KeyT, false>), dim3(ntiles), dim3(O3D_BLOCK), 0, ctx->stream, ka, va, kb, vb, hist, n,
                         shift, ntiles);
    }
    KeyT *tk = ka; ka = kb; kb = tk;
    uint32_t *tv = va; va = vb; vb = tv;
  }
  if (ka != keys && copy_home) {  // odd number of passes: bring the result home (or the caller started in the other pair)
    O3D_HIP(ctx, hipMemcpyAsync(keys, ka, sizeof(KeyT) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
    O3D_HIP(ctx, hipMemcpyAsync(vals, va, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
  }
  O3D_HIP(ctx, hipGetLastError());
  return O3D_OK;
}

}  // namespace

size_t o3d_sort_num_tiles(int64_t n) { return (size_t)((n + kSortTile - 1) / kSortTile); }
size_t o3d_scan_num_blocks(int64_t n) { return (size_t)((n + kScanTile - 1) / kScanTile); }

int o3d_exclusive_scan_u32(o3d_ctx *ctx, uint32_t *data, uint32_t *block_sums, int64_t n) {
  if (n <= 0) return O3D_OK;
  const uint32_t nb = (uint32_t)o3d_scan_num_blocks(n);
  hipLaunchKernelGGL(k_scan_reduce, dim3(nb), dim3(O3D_BLOCK), 0, ctx->stream, data, block_sums, n);
  hipLaunchKernelGGL(k_scan_block_sums, dim3(1), dim3(O3D_BLOCK), 0, ctx->stream, block_sums, nb);
  hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(O3D_BLOCK), 0, ctx->stream, data, block_sums, n);
  O3D_HIP(ctx, hipGetLastError());
  return O3D_OK;
}

int o3d_radix_sort_u32(o3d_ctx *ctx, uint32_t *keys, uint32_t *vals, uint32_t *tmp_keys, uint32_t *tmp_vals, uint32_t *hist,
                       int64_t n, int bits, bool copy_home) {
  return radix_sort_impl<uint32_t>(ctx, keys, vals, tmp_keys, tmp_vals, hist, n, bits, copy_home);
}
int o3d_radix_sort_u64(o3d_ctx *ctx, uint64_t *keys, uint32_t *vals, uint64_t *tmp_keys, uint32_t *tmp_vals, uint32_t *hist,
                       int64_t n, int bits, bool copy_home) {
  return radix_sort_impl<uint64_t>(ctx, keys, vals, tmp_keys, tmp_vals, hist, n, bits, copy_home);
}
// With an odd number of passes the result lands in the OTHER pair of buffers: a caller that can put its input there to
// begin with (o3d_radix_sort_odd) and passes copy_home = false saves the two copies that bring it home.
bool o3d_radix_sort_odd(int bits) {
  int passes = (bits + 7) / 8;
  if (passes < 1) passes = 1;
  return (passes & 1) != 0;
}
