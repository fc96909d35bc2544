// Internal declarations shared by the HIP translation units of libo3d_amd.so.
// gfx950 (MI355X / CDNA4) only: wave = 64, 256 CUs in 8 XCDs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <functional>
#include <string>
#include <vector>

#include "../../include/o3d_icp.h"

#define O3D_WAVE 64
#define O3D_BLOCK 256
#define O3D_NUM_CU 256
#define O3D_MAX_GRID (O3D_NUM_CU * 8)  // memory-bound kernels: cap the grid and stride the rest
#define O3D_NSUMS 32
#define O3D_GANG_MAX 32  // registrations advancing in lock step (o3d_launch_pass_gang)

// ---- sums layout (per pass, fp64) -------------------------------------------
// point-to-plane: [0..20] JTJ upper triangle row-major, [21..26] JTr, [27] sum r^2
// point-to-point: [0..2] sum (s-c), [3..5] sum (t-c), [6..14] sum (t-c)(s-c)^T row-major, [15] sum |s-c|^2
// both:           [28] sum d^2, [29] number of correspondences, [30..31] unused
#define O3D_SUM_D2 28
#define O3D_SUM_NC 29

// per-context device counters (o3d_ctx::counters): [0..7] search statistics, [8] failed fp64 confirmations
#define O3D_NCOUNTERS 16
#define O3D_CTR_CERTIFIED 6
#define O3D_CTR_VIOLATIONS 8

struct GridParams {
  double gmin[3];  // grid origin (min of the finite target points minus the margin of empty cells)
  double inv_h;    // 1 / cell size
  double h;
  int nx, ny, nz;  // cells per axis (unpadded)
  int nxp;         // padded row length = nx + 2 (one empty cell either side of every x-row)
  uint32_t ncell;  // nxp * ny * nz
  int margin;      // 1: the outer cell layers (1 in x, 2 in y and z) hold no target point (see o3d_build_target_grid)
  // Sparse scenes (grid.hip, "bricks"): the table holds only the bricks of B^3 cells, B = 1 << brick_bits, that a query
  // can see a target point from, each with a halo (B+2 rows in y and z, B+3 entries in x) so that the 3x3x3 block of any
  // query routed to the brick lies inside it; dir[(bz * nby + by) * nbx + bx] is the brick's first entry (0: the shared
  // empty brick).  brick_bits = 0: the dense table.
  int brick_bits;
  int nbx, nby;
  const uint32_t *dir;
};

// Device-resident ICP state: everything the loop of Registration.cpp:146-167 carries between iterations.
struct IcpState {
  double T[16];       // cumulative transformation, column-major
  double update[16];  // pending update for the next pass, column-major
  double fitness, rmse, n_corr, sum_d2;
  double rel_fitness, rel_rmse;
  double centre[3];   // fixed origin subtracted inside the Umeyama sums
  double n_src_global;
  double radius;            // half diagonal of the target's bounding box about `centre`
  double cert_skip_above;   // a pending update that moves the cloud farther than this makes k_icp_cert stand aside
  double seed_skip_above;   // ... and farther than this, the previous matches are poor first bounds for the search
  int max_iter;
  int passes;        // LOOP-A passes completed
  int iters;         // estimation updates applied
  int done;          // loop ended (break or max_iteration)
  int apply_update;  // next pass transforms the source by `update`
  int estimation;
  int with_scaling;
  int skip_cert;     // the pending update is too large for the certificates (k_icp_cert returns at once, everything is searched)
  int skip_seed;     // ... and for the previous matches to seed the search
  int pad_;
};

// One radix sort for the set-ups of a whole gang of registrations (o3d_registration_icp_batch): instead of sorting their
// cell keys themselves -- six short launches per sort, two sorts per registration, most of a small registration's set-up --
// the slots write their keys, tagged with a job number in the top bits, into one buffer; the gang sorts it once and then
// runs what each slot had queued behind its sort (cell table and gathers).  A stable sort with the tag on top leaves every
// job's keys in a segment of their own, in exactly the order its own sort would have produced.
#define O3D_GANG_KEY_BITS 26  // key bits below the job tag (6 bits: 64 jobs)
struct o3d_gang_sort {
  uint32_t *keys = nullptr, *vals = nullptr, *tkeys = nullptr, *tvals = nullptr, *hist = nullptr;
  size_t cap = 0, hist_cap = 0;  // elements
  int64_t used = 0;
  struct Job {
    struct o3d_ctx *owner;
    int64_t off, n;
    std::function<int(const uint32_t *keys, const uint32_t *vals)> then;  // segment pointers of the sorted buffer
  };
  std::vector<Job> jobs;
};

struct o3d_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipStream_t own_stream_or_given = nullptr;  // multi_registration_icp: the stream to return to after sharing a leader's
  std::string err;

  // target grid
  GridParams gp{};
  int64_t n_tgt = 0;        // points given
  int64_t n_tgt_grid = 0;   // finite points placed in the grid
  double4 *tgt_sorted = nullptr;  // {x, y, z, bits(original index)} in cell order
  double *nrm_sorted = nullptr;   // 3 doubles per sorted target point (nullable)
  double2 *tn_sorted = nullptr;   // point and normal of a sorted target point side by side: {x, y}, {z, nx}, {ny, nz} -- what the
                                  // estimate's sums gather per correspondence, in ONE 48-byte record (three 16-byte loads)
  size_t cap_tn = 0;
  uint32_t *tgt_index = nullptr;  // original index of every sorted target point (tgt_sorted[p].w as a dense array)
  size_t cap_tidx = 0;
  int tn_valid = 0;               // tn_sorted belongs to the grid in the context
  float4 *tgt_local = nullptr;    // fp32 {x, y, z} relative to the point's own cell origin, w = bits(padded x cell index)
  size_t cap_local = 0;
  int tile_ok = 0;                // fp32 prefilter usable on this grid (cell origins exact to fp32 resolution)
  int use_tile = 1;               // cooperative fp32 prefilter search (k_icp_search_coop) when tile_ok; 0: per-lane fp64 search
  int tile_stats = 0;             // collect g_tile_stats
  uint32_t *cell_start = nullptr; // ncell + 1 entries + pad
  size_t cap_tgt = 0, cap_nrm = 0, cap_cells = 0;
  uint32_t *brick_dir = nullptr;  // GridParams::dir of a bricked table
  size_t cap_dir = 0;
  int64_t n_bricks = 0;
  // several registrations in lock step (icp.hip, o3d_launch_pass_gang): the lead context's argument table
  void *d_gang = nullptr, *pinned_gang = nullptr;
  int gang_grid = 0, gang_cert = 0, gang_tn = 0, gang_stream = 0, gang_one_launch = 0;
  int gang_slots = 0;  // > 1: this context is one slot of a gang of that many (set-up sizes its share of the grid)
  const double *bounds_for = nullptr;      // o3d_bounds_enqueue: the cloud whose bounding box is on its way to the pinned page
  int64_t bounds_n = 0;
  int normals_fast = 1;                    // EstimateNormals: FastEigen3x3 (1, the reference default) or Eigen's iterative solver (0)
  o3d_gang_sort *gang_sort = nullptr;      // where this slot's set-up leaves its sorts (not owned), or null
  o3d_gang_sort *gang_sort_own = nullptr;  // the gang's sorter, owned by the gang's first context
  double knn_grid_h = -1.0;  // build_knn_grid: the cell size its last trial build was made for

  // source shard
  int64_t n_src = 0;
  int64_t n_src_global = 0;        // fitness denominator (all shards)
  double *src_sorted = nullptr;    // 3 doubles per point, processing order
  uint32_t *src_perm = nullptr;    // processing position -> caller's index
  int32_t *corr_sorted = nullptr;  // per processing position
  int32_t *corr_out = nullptr;     // caller's order (device staging)
  uint32_t *pos_sorted = nullptr;  // per processing position: position of the match in tgt_sorted
  float *lb_sorted = nullptr;      // per processing position: lower bound on the distance to every OTHER target point
  uint32_t *work = nullptr;        // work list of the search kernel (k_icp_cert), one segment per wavefront
  uint32_t *work_count = nullptr;  // one counter per segment (4 per workgroup)
  double *gsums = nullptr;         // pass_tail: 32 group rows of O3D_NSUMS, then (same allocation) ...
  unsigned *tickets = nullptr;     // ... 64 arrival counters, zero between launches
  size_t cap_src = 0;
  double *src_orig = nullptr;      // hypothesis mode: the source in processing order, never transformed
  size_t cap_src_orig = 0;
  int hyp_mode = 0;
  // coloured ICP
  double4 *tcol_sorted = nullptr;  // per sorted target point {intensity, colour gradient}
  double *sint_sorted = nullptr;   // per processing position: source intensity
  size_t cap_tcol = 0, cap_sint = 0;
  double slg = 0.0, slp = 0.0;     // sqrt(lambda_geometric), sqrt(1 - lambda_geometric)

  // loop
  IcpState *state = nullptr;   // device
  unsigned long long *counters = nullptr;  // device, O3D_NCOUNTERS entries, cleared by o3d_icp_setup
  double *partials = nullptr;  // [grid][32]
  double *sums = nullptr;      // [32] local
  double *log = nullptr;       // [(max_iter+1)][O3D_LOG_STRIDE]
  int log_rows = 0;
  int pass_grid = 0;    // workgroups of the accumulate / per-lane search kernels (one partial row each)
  int search_grid = 0;  // workgroups of k_icp_search_coop
  int use_seed = 1;     // k_icp_search_coop: previous match as the first pruning bound (O3D_NO_SEED=1 turns it off)
  int use_bricks = 1;   // sparse scenes get the bricked cell table (O3D_NO_BRICKS=1: coarser cells in a dense table instead)
  int use_tail = 1;     // the search kernel's last workgroups end the pass (O3D_NO_TAIL=1: separate finalize / solve launch)
  int tail_check = 0;   // O3D_TAIL_CHECK=1 (debugging aid): every pass's partial rows are added once more by k_finalize_plain and compared
  double *tail_check_buf = nullptr;  // 32 doubles
  int use_cert = 1;     // k_icp_search_coop: certified previous matches skip the search (O3D_NO_CERT=1 turns it off)
  int use_stream = 1;   // certificate half with streaming loads and stores (O3D_STREAM=0: ordinary ones; read per set-up)
  int use_one_launch = 1;  // certificates and search of a pass in one launch, k_icp_pass (O3D_ONE_LAUNCH=0: two launches)
  double cert_gap = 0.0625;  // extra search reach for the certificates' slack, in cells (O3D_CERT_GAP)
  int icp_ready = 0;
  int max_iter = 0;
  int enq_passes = 0;
  int sums_valid = 0;  // ctx->sums holds the sums of the last pass (o3d_icp_fetch)
  int estimation = 0;
  double thr = 0;  // (double)(float)(r*r)
  hipEvent_t setup_ev[3] = {nullptr, nullptr, nullptr};  // start, grid built, source sorted (o3d_icp_get_setup_ms)
  int has_tgt_normals = 0;

  // optional HIP-event profile of k_icp_pass
  int prof_on = 0, prof_n = 0;
  hipEvent_t prof_ev[1024] = {};

  // generic scratch (sort buffers etc.)
  void *scratch[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t scratch_cap[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  void *pinned = nullptr;  // 4 KiB pinned host staging
  void *xfer = nullptr;    // staged-copy state (xfer.hip), created by the first large host <-> device copy
  hipEvent_t poll_ev = nullptr;  // run_icp_loop: the copy of IcpState::done behind a chunk of iterations
  double rel_fitness = 0.0, rel_rmse = 0.0;  // the criteria of the last o3d_icp_setup
  // per-call device temporaries (copies of host inputs, staged outputs): blocks are kept between calls, see o3d_tmp_alloc
  struct TmpBlock { void *p; size_t cap; };
  std::vector<TmpBlock> tmp_idle, tmp_live;
  size_t tmp_idle_bytes = 0;
};

// ---- helpers ----------------------------------------------------------------
int o3d_fail(o3d_ctx *ctx, int status, const char *fmt, ...);
#define O3D_HIP(ctx, call)                                                                        \
  do {                                                                                            \
    hipError_t e__ = (call);                                                                      \
    if (e__ != hipSuccess)                                                                        \
      return o3d_fail(ctx, e__ == hipErrorOutOfMemory ? O3D_ERR_OOM : O3D_ERR_HIP, "%s: %s (%s:%d)", #call, \
                      hipGetErrorString(e__), __FILE__, __LINE__);                                \
  } while (0)
#define O3D_TRY(expr)          \
  do {                         \
    int s__ = (expr);          \
    if (s__ != O3D_OK) return s__; \
  } while (0)

// grow-only device buffer
int o3d_reserve(o3d_ctx *ctx, void **ptr, size_t *cap, size_t bytes);
// Device memory for the duration of one call.  hipMalloc + hipFree of a few hundred MB cost milliseconds per call (the free
// synchronises the device); a call that is repeated gets the blocks of the previous one back instead.  Blocks up to twice
// the requested size are reused; at most kTmpIdleMax bytes stay cached; o3d_tmp_release_all gives everything back.
hipError_t o3d_tmp_alloc(o3d_ctx *ctx, void **out, size_t bytes);
void o3d_tmp_free(o3d_ctx *ctx, void *p);
void o3d_tmp_release_all(o3d_ctx *ctx);
int o3d_scratch(o3d_ctx *ctx, int slot, size_t bytes, void **out);

// ---- xfer.hip: copies of caller (pageable) buffers, ordered with ctx->stream -------
hipError_t o3d_h2d(o3d_ctx *ctx, void *dst, const void *src, size_t bytes);
hipError_t o3d_d2h(o3d_ctx *ctx, void *dst, const void *src, size_t bytes);
void o3d_xfer_destroy(o3d_ctx *ctx);

// ---- sort_scan.hip ------------------------------------------------------------
// Stable LSD radix sort of (key, value) pairs, 8 bits per pass over the low `bits` bits of the key.
// keys/vals are overwritten with the sorted sequence (ping-pong through tmp buffers of the same size).
// KeyT = uint32_t or uint64_t.  hist needs 256 * ntiles(n) uint32.
size_t o3d_sort_num_tiles(int64_t n);
int o3d_radix_sort_u32(o3d_ctx *ctx, uint32_t *keys, uint32_t *vals, uint32_t *tmp_keys, uint32_t *tmp_vals,
                       uint32_t *hist, int64_t n, int bits, bool copy_home = true);
int o3d_radix_sort_u64(o3d_ctx *ctx, uint64_t *keys, uint32_t *vals, uint64_t *tmp_keys, uint32_t *tmp_vals,
                       uint32_t *hist, int64_t n, int bits, bool copy_home = true);
bool o3d_radix_sort_odd(int bits);  // the sort of `bits` key bits ends in the tmp pair (see sort_scan.hip)
// In-place exclusive prefix sum over n uint32 (n up to 2^31); block_sums needs ceil(n / 8192) + 1 uint32.
size_t o3d_scan_num_blocks(int64_t n);
int o3d_exclusive_scan_u32(o3d_ctx *ctx, uint32_t *data, uint32_t *block_sums, int64_t n);

// ---- grid.hip -----------------------------------------------------------------
// allow_bricks: a scene whose dense table of cells of side max_dist would exceed the budget gets the bricked table instead
// of coarser cells (only the ICP search kernels read it; every other caller passes 0).
int o3d_build_target_grid(o3d_ctx *ctx, const double *dev_xyz, const double *dev_nrm, int64_t n, double max_dist, int allow_bricks = 0);
int o3d_bounds_enqueue(o3d_ctx *ctx, const double *dev_xyz, int64_t n);
int o3d_sort_source(o3d_ctx *ctx, const double *dev_xyz, int64_t n, const double *init_colmajor, int apply_init);
// AABB over finite points: out6 (host) = {minx,miny,minz,maxx,maxy,maxz}; *out_finite = number of finite points.
int o3d_bounds(o3d_ctx *ctx, const double *dev_xyz, int64_t n, double out6[6], int64_t *out_finite);

// ---- RANSAC (icp.hip: hypotheses; ransac.hip: feature matching and the driver) ---
struct RansacArgs {
  const double *src, *snrm, *tgt, *tnrm;  // caller's order, 3 doubles per point; normals nullable
  int64_t n_src, n_tgt;
  const int32_t *corres;  // n_hyp x ransac_n x {source index, target index}
  int n_hyp, ransac_n, estimation, with_scaling;
  int use_edge, use_dist, use_normal;
  double edge_sim, dist_thr, cos_normal;
  double *out_T;      // n_hyp x 16, column-major
  int32_t *out_ok;    // n_hyp
};
int o3d_launch_ransac_hypotheses(o3d_ctx *ctx, const RansacArgs &a);
int o3d_ransac_corr_eval_device(o3d_ctx *ctx, const double *d_Ts, int n_hyp, const double *d_src, int64_t n_src, const double *d_tgt,
                                int64_t n_tgt, const int32_t *d_corres, int64_t n_c, double max_dist, double *d_out);
int o3d_ransac_corr_mask_device(o3d_ctx *ctx, const double *d_T, const double *d_src, int64_t n_src, const double *d_tgt, int64_t n_tgt,
                                const int32_t *d_corres, int64_t n_c, double max_dist, int32_t *d_mask);
int o3d_feature_nn_device(o3d_ctx *ctx, const double *d_src_feat, int64_t n_src, const double *d_tgt_feat, int64_t n_tgt, int dim,
                          const int32_t *d_query_ids, int64_t n_q, int32_t *d_out_index, double *d_out_d2);

// ---- device helpers -------------------------------------------------------------
__device__ __forceinline__ int o3d_cell_coord(double x, double gmin, double inv_h, int n) {
  // monotone in x; NaN -> -2 (no neighbours); clamped to [-2, n+1]
  double f = floor((x - gmin) * inv_h);
  return (f >= -2.0) ? ((f <= (double)(n + 1)) ? (int)f : n + 1) : -2;
}

// Index into the cell table of the entry LEFT of cell (cx, cy, cz) -- the four entries from there are the starts of the
// cells cx-1, cx, cx+1 and the end of cx+1 in the cell-ordered target -- for a cell inside the grid.  The rows (cy + dy,
// cz + dz), |dy|, |dz| <= 1, are o3d_row_stride_y / _z entries away.
__device__ __forceinline__ int o3d_row_stride_y(const GridParams &gp) { return gp.brick_bits ? (1 << gp.brick_bits) + 3 : gp.nxp; }
__device__ __forceinline__ int o3d_row_stride_z(const GridParams &gp) {
  return gp.brick_bits ? ((1 << gp.brick_bits) + 2) * ((1 << gp.brick_bits) + 3) : gp.ny * gp.nxp;
}
__device__ __forceinline__ uint32_t o3d_row_index(const GridParams &gp, int cx, int cy, int cz) {
  if (gp.brick_bits) {
    const int b = gp.brick_bits, m = (1 << b) - 1;
    const uint32_t base = gp.dir[((uint32_t)(cz >> b) * (uint32_t)gp.nby + (uint32_t)(cy >> b)) * (uint32_t)gp.nbx + (uint32_t)(cx >> b)];
    return base + (uint32_t)(((cz & m) + 1) * o3d_row_stride_z(gp) + ((cy & m) + 1) * o3d_row_stride_y(gp) + (cx & m));
  }
  return ((uint32_t)cz * (uint32_t)gp.ny + (uint32_t)cy) * (uint32_t)gp.nxp + (uint32_t)cx;
}
