/*
 * o3d_icp.h -- C ABI of the MI355X-native ICP registration hot path.
 *
 * Drop-in boundary for PolyCam/open3d's legacy registration pipeline.  The
 * reference has no FFI layer of its own (its boundary is ordinary C++,
 * SURVEY.md section 8b); each entry point below names the reference function
 * (file:line under /root/reference/open3d/) whose work it replaces.  The C++
 * adapter with the reference's own signatures lives in
 * open3d_amd/cpp/open3d_amd_registration.h; INTEGRATION.md shows the patch a
 * maintainer of the reference would apply.
 *
 * Conventions
 *  - points / normals: contiguous double[3] per point (std::vector<Eigen::Vector3d>
 *    storage, geometry/PointCloud.h:330-334); 4x4 matrices: 16 doubles COLUMN-major
 *    (Eigen::Matrix4d default storage).
 *  - correspondences: int32 corr[n_src], corr[i] = target index or -1; the
 *    reference's CorrespondenceSet is the ascending-i list of (i, corr[i]) >= 0
 *    (pipelines/registration/Registration.cpp:60-68 run serially).
 *  - every function returns an o3d_status (0 = ok) and never throws; the text for
 *    the last failure on a context is available from o3d_last_error().
 *  - `mem` says where the caller's bulk buffers live (host or device).  Scalars and
 *    4x4 matrices are always host memory.
 *  - a context owns one GPU, one stream and all scratch buffers; one call in flight
 *    per context, any number of contexts per process.
 *  - there is NO CPU fallback: without a usable HIP device every call fails with
 *    O3D_ERR_HIP.
 */
#ifndef O3D_ICP_H_
#define O3D_ICP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define O3D_API __attribute__((visibility("default")))
#else
#define O3D_API
#endif

typedef struct o3d_ctx o3d_ctx;

typedef enum {
  O3D_OK = 0,
  O3D_ERR_INVALID_DISTANCE = 1, /* Registration.cpp:134-136 "Invalid max_correspondence_distance." */
  O3D_ERR_NEED_NORMALS = 2,     /* Registration.cpp:137-144 point-to-plane without target normals */
  O3D_ERR_VOXEL_SIZE = 3,       /* geometry/PointCloud.cpp:275-277 "voxel_size <= 0." */
  O3D_ERR_VOXEL_TOO_SMALL = 4,  /* geometry/PointCloud.cpp:281-283 "voxel_size is too small." */
  O3D_ERR_INVALID_ARG = 5,
  O3D_ERR_HIP = 6,              /* HIP runtime failure or no device */
  O3D_ERR_OOM = 7,
  O3D_ERR_STATE = 8,            /* stepwise API called out of order */
  O3D_ERR_UNSUPPORTED = 9
} o3d_status;

typedef enum { O3D_MEM_HOST = 0, O3D_MEM_DEVICE = 1 } o3d_mem;

/* TransformationEstimationType, pipelines/registration/TransformationEstimation.h:45-50 */
typedef enum { O3D_EST_POINT_TO_POINT = 1, O3D_EST_POINT_TO_PLANE = 2, O3D_EST_COLORED_ICP = 3 } o3d_estimation;

/* KDTreeSearchParam::SearchType, geometry/KDTreeSearchParam.h:35-107 */
typedef enum { O3D_SEARCH_KNN = 0, O3D_SEARCH_RADIUS = 1, O3D_SEARCH_HYBRID = 2 } o3d_search;

/* ---- context -------------------------------------------------------------- */
/* One GPU, one stream, and the device memory of the calls made on it: grid, sorted source and sort buffers (grow-only) and,
 * for calls on host buffers, up to 16 GiB of idle device blocks kept from earlier calls for reuse (hipMalloc / hipFree cost
 * milliseconds per call).  Everything is returned by o3d_ctx_destroy; nothing is freed before that. */
O3D_API int o3d_ctx_create(int device_id, o3d_ctx **out_ctx);
/* One context over several GPUs of the node (no reference counterpart: the reference is single-process CPU code; SURVEY.md
 * 8e).  The context is rank 0 (device_ids[0]) and behaves like an ordinary context in every entry point; in addition
 * o3d_registration_icp -- and with it the C++ RegistrationICP of the adapter -- splits the source into n_devices contiguous
 * shards, builds the target grid on every device and exchanges 32 fp64 sums per rank and iteration through pinned host
 * memory, every rank adding the table in rank order: the same transformation on every rank, the correspondence array the
 * concatenation of the shards.  The same device may be listed more than once (a rehearsal of the sharded path on one GPU).
 * o3d_estimate_normals splits its per-point loop the same way (every device builds the grid, computes the normals of its
 * range of the cell-ordered cloud, device 0 assembles them: bit-identical to the one-device call).
 * With mem = O3D_MEM_DEVICE the caller's buffers must be readable from every listed device (peer access is enabled here
 * where the hardware offers it).  n_devices = 1 is o3d_ctx_create. */
O3D_API int o3d_ctx_create_multi(int n_devices, const int *device_ids, o3d_ctx **out_ctx);
O3D_API int o3d_ctx_num_devices(const o3d_ctx *ctx);
O3D_API void o3d_ctx_destroy(o3d_ctx *ctx);
/* Launch everything on an existing hipStream_t (e.g. torch's current stream).  The handle is used as given: NULL is
 * the HIP null stream (torch's default stream), so that work enqueued by the caller on that stream -- an RCCL
 * all-gather issued through torch.distributed, say -- is ordered with the library's kernels.
 * o3d_ctx_use_own_stream returns to the private non-blocking stream the context was created with. */
O3D_API int o3d_ctx_set_stream(o3d_ctx *ctx, void *hip_stream);
O3D_API int o3d_ctx_use_own_stream(o3d_ctx *ctx);
O3D_API const char *o3d_last_error(const o3d_ctx *ctx);
O3D_API const char *o3d_status_string(int status);
/* 1 when the library was built for and sees a gfx950 device */
O3D_API int o3d_device_count(void);

/* ---- whole-call drop-ins ---------------------------------------------------- */

/* RegistrationICP, pipelines/registration/Registration.cpp:129-168 (driver), with
 * GetRegistrationResultAndCorrespondences :41-91, KDTreeFlann::SearchHybrid
 * geometry/KDTreeFlann.cpp:132-152, PointCloud::Transform geometry/Geometry3D.cpp:77-89,
 * TransformationEstimationPointToPoint/PointToPlane::ComputeTransformation
 * TransformationEstimation.cpp:49-60,74-98 and utility::ComputeJTJandJTr
 * utility/Eigen.cpp:159-201 fused on the device.
 * tgt_nrm may be NULL for point-to-point.  out_corr may be NULL. */
O3D_API int o3d_registration_icp(o3d_ctx *ctx, const double *src_xyz, int64_t n_src, const double *tgt_xyz, const double *tgt_nrm,
                         int64_t n_tgt, double max_correspondence_distance, const double init_colmajor[16], int estimation,
                         int with_scaling, double relative_fitness, double relative_rmse, int max_iteration, int mem,
                         double out_transformation[16], int32_t *out_corr, int64_t *out_n_corr, double *out_fitness,
                         double *out_inlier_rmse, int *out_iterations);

/* EvaluateRegistration, pipelines/registration/Registration.cpp:118-127 */
O3D_API int o3d_evaluate_registration(o3d_ctx *ctx, const double *src_xyz, int64_t n_src, const double *tgt_xyz, int64_t n_tgt,
                              double max_correspondence_distance, const double transformation_colmajor[16], int mem,
                              int32_t *out_corr, int64_t *out_n_corr, double *out_fitness, double *out_inlier_rmse);

/* The validation step of RegistrationRANSACBasedOnFeatureMatching, Registration.cpp:269-275, for a batch of hypotheses:
 * for each of the n_hyp transformations (16 doubles each, column-major) the source is transformed (always, :269-270) and
 * GetRegistrationResultAndCorrespondences (:41-91) gives out_fitness[k] / out_inlier_rmse[k] / out_n_corr[k].  The target
 * grid is built and the source uploaded once.  *out_best = the hypothesis RANSAC keeps (:272-275: highest fitness, then
 * lowest rmse, the earliest on a tie; -1 when no hypothesis has a correspondence) and out_corr_best (nullable, n_src) its
 * correspondences. */
O3D_API int o3d_evaluate_hypotheses(o3d_ctx *ctx, const double *src_xyz, int64_t n_src, const double *tgt_xyz, int64_t n_tgt,
                            double max_correspondence_distance, const double *transformations_colmajor, int n_hyp, int mem,
                            double *out_fitness, double *out_inlier_rmse, int64_t *out_n_corr, int *out_best,
                            int32_t *out_corr_best);

/* RegistrationColoredICP, pipelines/registration/ColoredICP.cpp:205-211: InitializePointCloudForColoredICP (:68-123, the
 * per-target-point colour gradient over KDTreeSearchParamHybrid(2 * max_distance, 30)) followed by RegistrationICP with
 * TransformationEstimationForColoredICP::ComputeTransformation (:125-177: a geometric and a photometric row per
 * correspondence through the multi-row utility::ComputeJTJandJTr, utility/Eigen.cpp:203-249).  Colours are 3 doubles per
 * point (PointCloud::colors_).  lambda_geometric outside [0, 1] becomes 0.968 (:50-53).  Without source or target
 * colours every update is the identity (:127-128), as in the reference.  out_corr / out_gradients (n_tgt x 3) may be NULL. */
O3D_API int o3d_registration_colored_icp(o3d_ctx *ctx, const double *src_xyz, const double *src_col, int64_t n_src,
                                 const double *tgt_xyz, const double *tgt_nrm, const double *tgt_col, int64_t n_tgt,
                                 double max_distance, const double init_colmajor[16], double lambda_geometric,
                                 double relative_fitness, double relative_rmse, int max_iteration, int mem,
                                 double out_transformation[16], int32_t *out_corr, int64_t *out_n_corr, double *out_fitness,
                                 double *out_inlier_rmse, int *out_iterations, double *out_gradients);

/* RegistrationICP over a batch of independent pairs (BASELINE config 5: thousands of small fragment pairs; the reference
 * would call Registration.cpp:129-168 in a loop).  Pair p uses src_xyz[p] (n_src[p] points), tgt_xyz[p], tgt_nrm[p]
 * (n_tgt[p] points; tgt_nrm may be NULL for point-to-point), init_colmajor + 16 p (init_colmajor NULL: identity) and
 * writes out_transformation + 16 p, out_fitness[p], out_inlier_rmse[p], out_iterations[p], out_n_corr[p] and, when
 * out_corr and out_corr[p] are non-NULL, n_src[p] correspondences.  The pairs are spread over `n_workers` host threads
 * (0: default 7) on ctx's device; each thread advances a GANG of registrations in lock step -- as many as put about six
 * million source points into a launch, 32 at most, one for pairs that fill the GPU by themselves -- with one launch per
 * pass kernel for the whole gang, so that small pairs do not pay a launch each, and the set-up of one gang overlaps the
 * loop of another.  The context keeps one worker context (stream, grid and source buffers of a pair) per thread and gang
 * slot until it is destroyed.  A pair's correspondences are those of its single call; its transformation agrees to
 * rounding (the gang shares the GPU's resident round, so a pair's sums are added over fewer, longer workgroups).
 * out_status[p] (may be NULL) receives the pair's status; the return value is the first non-zero one. */
O3D_API int o3d_registration_icp_batch(o3d_ctx *ctx, int64_t n_pairs, const double *const *src_xyz, const int64_t *n_src,
                               const double *const *tgt_xyz, const double *const *tgt_nrm, const int64_t *n_tgt,
                               double max_correspondence_distance, const double *init_colmajor, int estimation,
                               int with_scaling, double relative_fitness, double relative_rmse, int max_iteration, int mem,
                               int n_workers, double *out_transformation, int32_t *const *out_corr, int64_t *out_n_corr,
                               double *out_fitness, double *out_inlier_rmse, int *out_iterations, int *out_status);

/* TransformationEstimationPointToPoint / PointToPlane::ComputeTransformation, TransformationEstimation.cpp:49-60,74-98
 * (with utility::ComputeJTJandJTr utility/Eigen.cpp:159-201 and SolveJacobianSystemAndObtainExtrinsicMatrix :116-130),
 * over a caller-given CorrespondenceSet: corres_pairs = n_corres x int32[2] (source index, target index), the storage of
 * std::vector<Eigen::Vector2i>.  Identity when n_corres == 0 or point-to-plane without target normals (:51-52,76-77). */
O3D_API int o3d_compute_transformation(o3d_ctx *ctx, const double *src_xyz, int64_t n_src, const double *tgt_xyz,
                               const double *tgt_nrm, int64_t n_tgt, const int32_t *corres_pairs, int64_t n_corres,
                               int estimation, int with_scaling, int mem, double out_transformation[16]);
/* No counterpart in the reference.  o3d_compute_transformation (and, through out[16..17], o3d_compute_rmse) for callers that
 * cannot allocate: code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  No context: the
 * kernels' scratch (one 32-double partial row per workgroup + the count of out-of-range pairs, 512.25 KiB) is a file-scope
 * device array that every GPU the library runs on holds until the process ends.  The call allocates and frees nothing, does
 * not synchronise, and only enqueues one memset and two kernels on `hip_stream` (a hipStream_t; NULL = the null stream) of
 * device `device_id`.  Every buffer is DEVICE memory.  d_out20 receives 20 doubles: [0..15] the transformation
 * (column-major), [16] point-to-point RMSE, [17] point-to-plane RMSE (0 without normals), [18] number of pairs used,
 * [19] number of pairs with an index outside the clouds (the result is only valid when it is 0).  Sizes above
 * O3D_NOALLOC_MAX_CORRES are rejected with O3D_ERR_UNSUPPORTED.  All calls on one GPU share that GPU's array: calls that
 * overlap in time on one GPU (different streams or threads) race; serialise them.  The array's address is looked up with
 * hipGetSymbolAddress by o3d_ctx_create or, failing that, by the first call on the device -- make one of the two happen
 * before a graph capture begins. */
#define O3D_NOALLOC_MAX_CORRES 2147483647LL
O3D_API int o3d_compute_transformation_noalloc(int device_id, void *hip_stream, const double *d_src_xyz, int64_t n_src,
                                       const double *d_tgt_xyz, const double *d_tgt_nrm, int64_t n_tgt,
                                       const int32_t *d_corres_pairs, int64_t n_corres, int estimation, int with_scaling,
                                       double *d_out20);
/* TransformationEstimationPointToPoint / PointToPlane::ComputeRMSE, TransformationEstimation.cpp:38-47,62-72. */
O3D_API int o3d_compute_rmse(o3d_ctx *ctx, const double *src_xyz, int64_t n_src, const double *tgt_xyz, const double *tgt_nrm,
                     int64_t n_tgt, const int32_t *corres_pairs, int64_t n_corres, int estimation, int mem, double *out_rmse);

/* GetInformationMatrixFromPointClouds, pipelines/registration/Registration.cpp:303-355.  out_gtg: 36 doubles (6x6,
 * symmetric, so row- and column-major agree). */
O3D_API int o3d_information_matrix(o3d_ctx *ctx, const double *src_xyz, int64_t n_src, const double *tgt_xyz, int64_t n_tgt,
                           double max_correspondence_distance, const double transformation_colmajor[16], int mem,
                           double out_gtg[36]);

/* ---- stepwise, device-resident ICP (same loop, caller owns the per-iteration
 *      exchange; used by the sharded multi-GPU driver and by bench.py) -------- */

/* Builds the target grid (replaces KDTreeFlann::SetGeometry, geometry/KDTreeFlann.cpp:58-73,154-167),
 * uploads/permutes the source shard and stores init and the criteria (Registration.cpp:146-152).
 * n_src_global: number of source points over ALL shards (fitness denominator, Registration.cpp:87); pass
 * n_src when not sharded. */
O3D_API int o3d_icp_setup(o3d_ctx *ctx, const double *src_xyz, int64_t n_src, const double *tgt_xyz, const double *tgt_nrm,
                  int64_t n_tgt, double max_correspondence_distance, const double init_colmajor[16], int estimation,
                  int with_scaling, double relative_fitness, double relative_rmse, int max_iteration, int mem,
                  int64_t n_src_global);
/* Enqueue one fused pass: apply the pending update to the source shard (Registration.cpp:159), find
 * correspondences (:161) and reduce this shard's JTJ/JTr (or Umeyama) sums for the next estimate (:157).
 * The 32 fp64 partial sums land in the device buffer returned by o3d_icp_local_sums(). */
O3D_API int o3d_icp_pass(o3d_ctx *ctx);
O3D_API double *o3d_icp_local_sums(o3d_ctx *ctx); /* device pointer, 32 doubles */
/* Same pass, partial sums written to a caller-owned device buffer of 32 doubles (e.g. a torch tensor that is then
 * all-gathered over RCCL). */
O3D_API int o3d_icp_pass_to(o3d_ctx *ctx, double *dev_sums_out);
/* Enqueue the estimate: sums `n_ranks` x 32 doubles (device pointer, rank-major, added in rank order so every
 * rank gets bit-identical results), then fitness/rmse (Registration.cpp:82-89), the convergence test
 * (:162-165), the 6x6 LDLT solve or Umeyama SVD, update composition (:158).  Pass the pointer from
 * o3d_icp_local_sums() and n_ranks = 1 when not sharded. */
O3D_API int o3d_icp_solve(o3d_ctx *ctx, const double *dev_all_sums, int n_ranks);
/* Unsharded shortcut: o3d_icp_pass + o3d_icp_solve(local sums, 1) with the partial reduction and the estimate fused
 * into one launch. */
O3D_API int o3d_icp_step(o3d_ctx *ctx);
/* Estimations that stay on the host (TransformationEstimationForColoredICP, user subclasses of TransformationEstimation,
 * TransformationEstimation.h:57-80): the caller runs o3d_icp_pass, reads the correspondences and the fitness / rmse of
 * that pass (Registration.cpp:82-89) with o3d_icp_fetch, calls its own ComputeTransformation and hands the update back
 * with o3d_icp_set_update, which composes it into the transformation and applies it to the source in the next pass
 * (Registration.cpp:158-159).  The convergence test (:162-165) is the caller's. */
O3D_API int o3d_icp_fetch(o3d_ctx *ctx, int mem, int32_t *out_corr, int64_t *out_n_corr, double *out_fitness, double *out_inlier_rmse);
O3D_API int o3d_icp_set_update(o3d_ctx *ctx, const double update_colmajor[16]);
/* Synchronise and report whether the loop has ended (converged or max_iteration reached). */
O3D_API int o3d_icp_poll(o3d_ctx *ctx, int *out_done, int *out_passes);
/* Synchronise and fetch the RegistrationResult (Registration.h:107-126).  out_corr (n_src entries, `mem` space, may be
 * NULL) is in the caller's original source order. */
O3D_API int o3d_icp_finish(o3d_ctx *ctx, int mem, double out_transformation[16], int32_t *out_corr, int64_t *out_n_corr,
                   double *out_fitness, double *out_inlier_rmse, int *out_iterations);
/* Per-pass history for parity tests: row p = {fitness, rmse, n_corr, sum_d2, T[16] colmajor, sums[32]} = 52 doubles. */
#define O3D_LOG_STRIDE 52
O3D_API int o3d_icp_get_log(o3d_ctx *ctx, double *host_log, int max_rows, int *out_rows);
/* HIP-event timing of the pass kernels on the context's stream.  enable != 0: every following pass is bracketed by HIP
 * events (at most 340 passes are kept).  o3d_icp_profile_read synchronises, returns the mean duration in milliseconds
 * of the pass kernel(s) over the recorded passes and their number, and clears the record. */
O3D_API int o3d_icp_profile(o3d_ctx *ctx, int enable);
O3D_API int o3d_icp_profile_read(o3d_ctx *ctx, double *out_mean_ms, int *out_count);
/* Same, split into the two kernels of a pass: out_mean_ms[0] = k_icp_cert (certified previous matches; absent, ~0, in the
 * first pass and without certificates), [1] = k_icp_search_coop (the search over what is left).  On the alternative
 * path: [0] = k_icp_search, [1] = k_icp_accumulate_mfma. */
O3D_API int o3d_icp_profile_read2(o3d_ctx *ctx, double out_mean_ms[2], int *out_count);
/* Same record, pass by pass (search + accumulation kernels of each recorded pass, milliseconds); does not clear it. */
O3D_API int o3d_icp_profile_read_all(o3d_ctx *ctx, double *out_ms, int max_count, int *out_count);
/* Test / tuning hooks (no reference counterpart).  force_generic_pass != 0 selects the per-lane fp64 search kernel +
 * accumulation kernel instead of the default cooperative fp32-prefilter search (both give the exact result);
 * collect_tile_stats != 0 makes the cooperative search count {wave trips, dense item trips, runs listed, queries resolved
 * by the exact fp64 search, items, valid lanes, queries certified without a search} (per-context counters in device memory,
 * read by o3d_debug_tile_stats). */
O3D_API int o3d_debug_set(o3d_ctx *ctx, int force_generic_pass, int collect_tile_stats);
O3D_API int o3d_debug_tile_stats(o3d_ctx *ctx, uint64_t out[8], int reset);
/* The cell table of the last grid built on the context: {cell side, nx, ny, nz, padded row length, margin cells, 1 if the
 * cooperative fp32-prefilter search is in use (0: the per-lane fp64 search), points in the grid, cells per brick side
 * (0: dense table), bricks stored}. */
O3D_API int o3d_debug_grid_info(const o3d_ctx *ctx, double out[10]);
/* Test hook, no reference counterpart: sets THIS context's count of fp32 winners whose fp64 confirmation failed, as the
 * search kernel would; the next o3d_icp_finish / o3d_icp_fetch on this context (and on no other) then returns
 * O3D_ERR_STATE.  The counter is cleared by o3d_icp_setup and by the call that reports it. */
O3D_API int o3d_debug_inject_bound_violation(o3d_ctx *ctx, unsigned int count);
/* Processing order of the source shard after o3d_icp_setup: out[p] = caller's index of the p-th processed point. */
O3D_API int o3d_debug_get_source_perm(o3d_ctx *ctx, uint32_t *host_out);
/* Tuning hook, no reference counterpart: out[p] = where the match of the p-th processed point sits in the cell-sorted target
 * (0xFFFFFFFF: none), as the last pass left it -- the addresses the certifying pass gathers from (scripts/micro). */
O3D_API int o3d_debug_get_match_positions(o3d_ctx *ctx, uint32_t *host_out);
/* Timing breakdown of the last setup (milliseconds, HIP events): [0] grid build, [1] source upload+sort. */
O3D_API int o3d_icp_get_setup_ms(o3d_ctx *ctx, double out_ms[2]);

/* ---- preprocessing ---------------------------------------------------------- */

/* PointCloud::VoxelDownSample, geometry/PointCloud.cpp:273-306.  nrm / col may be NULL.  Output buffers must hold
 * n entries (the voxel count M <= n is returned in *out_m).  Output order: ascending voxel index (z, y, x); the
 * reference's order is unordered_map bucket order (unspecified).  out_voxel_index (3 x int32 per voxel) may be NULL. */
O3D_API int o3d_voxel_down_sample(o3d_ctx *ctx, const double *xyz, const double *nrm, const double *col, int64_t n,
                          double voxel_size, int mem, double *out_xyz, double *out_nrm, double *out_col,
                          int32_t *out_voxel_index, int64_t *out_m);

/* PointCloud::EstimateNormals, geometry/EstimateNormals.cpp:320-351 with ComputeNormal :215-252 and FastEigen3x3
 * :130-213 (fast_normal_computation = true, the reference default).  normals is in/out when has_normals != 0.
 * Limit: KNN and hybrid searches keep their k nearest in per-lane lists in LDS (8 bytes per entry and lane), so knn / max_nn
 * may be at most 256; a larger value returns O3D_ERR_UNSUPPORTED (the reference accepts any).  Radius search has no such
 * limit.  The same limit holds for o3d_compute_fpfh_feature and the colour gradients of o3d_registration_colored_icp. */
O3D_API int o3d_estimate_normals(o3d_ctx *ctx, const double *xyz, int64_t n, int search_type, int knn, double radius, int mem,
                         double *normals, int has_normals);

/* ---- point clouds that stay in HBM across calls ------------------------------------------------------------------------
 * The reference hands `const geometry::PointCloud &` from call to call (Registration.h:149-152, PointCloud.h:130,192):
 * VoxelDownSample returns a cloud that EstimateNormals extends and RegistrationICP reads, all in the same memory.  Behind raw
 * host pointers every one of those calls would move its clouds over PCIe again.  An o3d_cloud is the device-resident form of a
 * PointCloud's points_ / normals_ / colors_ (PointCloud.h:330-334), owned by the library, on the device of the context it was
 * created with; each function below is the entry point of the same name on such clouds (same kernels: results identical to
 * the host-buffer call bit for bit).  Inputs are never modified (Registration.cpp:149 works on a copy), except that
 * o3d_cloud_estimate_normals sets the cloud's normals as PointCloud::EstimateNormals does.  Results (transformation,
 * correspondences, ...) go to HOST memory.  A cloud may be used with any context on its device, and with a multi context
 * whose other devices can read that device's memory (o3d_ctx_create_multi). */
typedef struct o3d_cloud o3d_cloud;
/* nrm / col may be NULL.  The caller's buffers (`mem` space) are free when the call returns. */
O3D_API int o3d_cloud_upload(o3d_ctx *ctx, const double *xyz, const double *nrm, const double *col, int64_t n, int mem,
                             o3d_cloud **out_cloud);
O3D_API void o3d_cloud_free(o3d_cloud *cloud);
O3D_API int64_t o3d_cloud_size(const o3d_cloud *cloud);
O3D_API int o3d_cloud_has_normals(const o3d_cloud *cloud);
O3D_API int o3d_cloud_has_colors(const o3d_cloud *cloud);
/* device pointers (3 doubles per point; NULL when the cloud has no such attribute), valid until o3d_cloud_free */
O3D_API const double *o3d_cloud_points(const o3d_cloud *cloud);
O3D_API const double *o3d_cloud_normals(const o3d_cloud *cloud);
O3D_API const double *o3d_cloud_colors(const o3d_cloud *cloud);
/* host buffers of o3d_cloud_size entries each; any may be NULL */
O3D_API int o3d_cloud_download(o3d_ctx *ctx, const o3d_cloud *cloud, double *out_xyz, double *out_nrm, double *out_col);
/* PointCloud::VoxelDownSample, geometry/PointCloud.cpp:273-306: a new cloud (points, and normals / colours where the input
 * has them) of exactly the voxel count, in ascending voxel index like o3d_voxel_down_sample. */
O3D_API int o3d_cloud_voxel_down_sample(o3d_ctx *ctx, const o3d_cloud *in, double voxel_size, o3d_cloud **out_cloud);
/* PointCloud::EstimateNormals, geometry/EstimateNormals.cpp:320-351, in place: existing normals orient the new ones (:343-346). */
O3D_API int o3d_cloud_estimate_normals(o3d_ctx *ctx, o3d_cloud *cloud, int search_type, int knn, double radius);
/* RegistrationICP, Registration.cpp:129-168, as o3d_registration_icp; the target's normals are the cloud's.  out_corr: HOST,
 * o3d_cloud_size(source) entries, may be NULL. */
O3D_API int o3d_cloud_registration_icp(o3d_ctx *ctx, const o3d_cloud *source, const o3d_cloud *target,
                                       double max_correspondence_distance, const double init_colmajor[16], int estimation,
                                       int with_scaling, double relative_fitness, double relative_rmse, int max_iteration,
                                       double out_transformation[16], int32_t *out_corr, int64_t *out_n_corr, double *out_fitness,
                                       double *out_inlier_rmse, int *out_iterations);
/* EvaluateRegistration, Registration.cpp:118-127, as o3d_evaluate_registration. */
O3D_API int o3d_cloud_evaluate_registration(o3d_ctx *ctx, const o3d_cloud *source, const o3d_cloud *target,
                                            double max_correspondence_distance, const double transformation_colmajor[16],
                                            int32_t *out_corr, int64_t *out_n_corr, double *out_fitness, double *out_inlier_rmse);

/* ---- the other KDTreeFlann consumers on point clouds ------ */

/* PointCloud::ComputePointCloudDistance, geometry/PointCloud.cpp:112-132: out_dist[i] = distance from source point i to its
 * nearest target point (SearchKNN k = 1, no radius), 0 when the target is empty. */
O3D_API int o3d_compute_point_cloud_distance(o3d_ctx *ctx, const double *src_xyz, int64_t n_src, const double *tgt_xyz, int64_t n_tgt,
                                     int mem, double *out_dist);
/* PointCloud::ComputeNearestNeighborDistance, geometry/PointCloud.cpp:521-544: distance to the second of SearchKNN(k = 2)
 * on the cloud itself (the first is the point itself), all zero for fewer than 2 points. */
O3D_API int o3d_compute_nearest_neighbor_distance(o3d_ctx *ctx, const double *xyz, int64_t n, int mem, double *out_dist);
/* The per-point part of PointCloud::RemoveRadiusOutliers, geometry/PointCloud.cpp:406-411: out_count[i] = SearchRadius
 * neighbour count (d2 < (double)(float)(r * r), the point itself included); the caller keeps count > nb_points. */
O3D_API int o3d_radius_neighbor_count(o3d_ctx *ctx, const double *xyz, int64_t n, double search_radius, int mem, int32_t *out_count);
/* The per-point part of PointCloud::RemoveStatisticalOutliers, geometry/PointCloud.cpp:438-449: out_mean[i] = mean of the
 * distances to the nb_neighbors nearest points (the point itself included, ascending order), -1 when there is none; the
 * cloud statistics and the threshold (:453-466) are a serial pass over these values on the caller's side. */
O3D_API int o3d_knn_mean_distance(o3d_ctx *ctx, const double *xyz, int64_t n, int nb_neighbors, int mem, double *out_mean);

/* The fast_normal_computation argument of PointCloud::EstimateNormals (PointCloud.h:192, EstimateNormals.cpp:245-251) as a
 * sticky setting of the context: non-zero (the default, the reference's default) = FastEigen3x3; 0 = column 0 of
 * Eigen::SelfAdjointEigenSolver<Matrix3d>::compute(covariance, ComputeEigenvectors).eigenvectors(), :247-251.  Read by
 * o3d_estimate_normals and o3d_estimate_normals_shard on this context (and on every device of a multi context). */
O3D_API int o3d_set_fast_normal_computation(o3d_ctx *ctx, int fast_normal_computation);

/* EstimateNormals split over GPUs (no counterpart in the reference, whose loop EstimateNormals.cpp:330-350 is
 * per point and independent).  Every rank builds the same grid over the whole cloud; shard g of G computes the normals of
 * positions [g * rows, min(ng, (g + 1) * rows)) of the cell-ordered cloud, rows = ceil(ng / G), ng = number of finite points
 * (*out_ng), and writes them in that order to out_part (rows x 3, padding rows zero).  oriented (nullable): existing normals
 * to orient against (EstimateNormals.cpp:343-346).  After the ranks' parts have been all-gathered in rank order,
 * o3d_estimate_normals_assemble -- on the same context, directly afterwards -- scatters them to the caller's order;
 * the result is bit-identical to o3d_estimate_normals. */
O3D_API int o3d_estimate_normals_shard(o3d_ctx *ctx, const double *xyz, int64_t n, int search_type, int knn, double radius, int mem,
                                       const double *oriented, int shard, int n_shards, double *out_part, int64_t *out_ng);
O3D_API int o3d_estimate_normals_assemble(o3d_ctx *ctx, const double *xyz, int64_t n, const double *all_parts, int n_shards, int mem,
                                          double *normals);

/* ComputeFPFHFeature, pipelines/registration/Feature.cpp:113-159 (SPFH :71-111, pair features :39-69).  search_type KNN
 * or HYBRID (knn / max_nn <= 256).  out_feature: 33 doubles per point, the storage of Feature::data_ (33 x n, column-major). */
O3D_API int o3d_compute_fpfh_feature(o3d_ctx *ctx, const double *xyz, const double *normals, int64_t n, int search_type, int knn,
                             double radius, int mem, double *out_feature);

/* The feature-space nearest neighbour of RegistrationRANSACBasedOnFeatureMatching, Registration.cpp:223,237
 * (KDTreeFlann over the target Feature, SearchKNN(column, 1): geometry/KDTreeFlann.cpp:93-110): for every query the target
 * column with the smallest flann::L2<double> distance (4-wide groups, then the tail), ties to the lowest
 * index.  Features are `dim` doubles per point (Feature::data_, dim x n column-major).  query_ids (n_q indices into the
 * source features, same memory space as the data) may be NULL: all n_src columns, n_q = n_src.  out_d2 may be NULL. */
O3D_API int o3d_feature_nearest(o3d_ctx *ctx, const double *src_feature, int64_t n_src, const double *tgt_feature, int64_t n_tgt,
                                int dim, const int32_t *query_ids, int64_t n_q, int mem, int32_t *out_index, double *out_d2);

/* The per-iteration body of RegistrationRANSACBasedOnFeatureMatching between sampling and validation,
 * Registration.cpp:250-268, for n_hyp candidate correspondence sets of ransac_n pairs each (corres: n_hyp x ransac_n x
 * {source index, target index}): the checkers of CorrespondenceChecker.cpp:38-82 (edge_similarity, distance_threshold,
 * normal_angle_threshold; a negative value leaves a checker out; the normal checker passes when either normals pointer is
 * NULL, :67-72) and estimation.ComputeTransformation over the ransac_n pairs (TransformationEstimation.cpp:49-60,74-98).
 * out_T: n_hyp x 16 column-major; out_ok[k] = 1 when every checker passed. */
O3D_API int o3d_ransac_hypotheses(o3d_ctx *ctx, const double *src_xyz, const double *src_normals, int64_t n_src,
                                  const double *tgt_xyz, const double *tgt_normals, int64_t n_tgt, const int32_t *corres,
                                  int n_hyp, int ransac_n, int estimation, int with_scaling, double edge_similarity,
                                  double distance_threshold, double normal_angle_threshold, int mem, double *out_T,
                                  int32_t *out_ok);

/* RegistrationRANSACBasedOnFeatureMatching, Registration.cpp:199-301.  Sampling (utility::UniformRandInt, :234) runs on
 * the host from std::mt19937(seed) (seed < 0: std::random_device, as the reference); feature matching, checkers,
 * ComputeTransformation and the validation of :269-275 run on the device in batches, consumed in iteration order so
 * that max_iteration / max_validation count as in the reference's serial loop.  Empty result (identity, fitness 0) for
 * ransac_n < 3 or max_dist <= 0 (:207-209).  out_corr: per source point the matched target index or -1, for the returned
 * transformation.  out_counts (nullable): {iterations run, validations run}. */
O3D_API int o3d_registration_ransac_feature_matching(o3d_ctx *ctx, const double *src_xyz, const double *src_normals, int64_t n_src,
                                                     const double *tgt_xyz, const double *tgt_normals, int64_t n_tgt,
                                                     const double *src_feature, const double *tgt_feature, int dim,
                                                     double max_dist, int estimation, int with_scaling, int ransac_n,
                                                     double edge_similarity, double distance_threshold,
                                                     double normal_angle_threshold, int max_iteration, int max_validation,
                                                     int64_t seed, int mem, double *out_T, int32_t *out_corr,
                                                     int64_t *out_n_corr, double *out_fitness, double *out_rmse,
                                                     int *out_counts);

/* RegistrationRANSACBasedOnCorrespondence, Registration.cpp:170-197: min(max_iteration, max_validation) rounds of
 * {sample ransac_n of the given correspondences (:184-186), ComputeTransformation (:187), evaluate the transformed source
 * against the whole correspondence list (EvaluateRANSACBasedOnCorrespondence, :93-116)}, all rounds in one batch on the
 * device; sampling from std::mt19937(seed) on the host (seed < 0: std::random_device).  corres: n_corres x {source index,
 * target index}.  out_inlier (n_corres flags, nullable): the correspondences kept in the result's correspondence_set_.
 * Empty result for ransac_n < 3, n_corres < ransac_n or max_dist <= 0 (:176-178). */
O3D_API int o3d_registration_ransac_correspondence(o3d_ctx *ctx, const double *src_xyz, int64_t n_src, const double *tgt_xyz,
                                                   const double *tgt_normals, int64_t n_tgt, const int32_t *corres,
                                                   int64_t n_corres, double max_dist, int estimation, int with_scaling,
                                                   int ransac_n, int max_iteration, int max_validation, int64_t seed, int mem,
                                                   double *out_T, int32_t *out_inlier, int64_t *out_n_inlier,
                                                   double *out_fitness, double *out_rmse);

/* The correspondence_set_ of a RegistrationResult from the dense form the entry points above return (corr[i] = target index
 * of source point i, or -1): {i, corr[i]} for every matched i in ascending i, the order Registration.cpp:60-75 builds it
 * in (per-thread vectors of ascending i, concatenated in thread order under a static schedule).  Host memory only, no
 * context.  out_pairs holds 2 * n int32 at most; returns the number of pairs. */
O3D_API int64_t o3d_expand_correspondences(const int32_t *corr, int64_t n, int32_t *out_pairs);

/* Benchmark scaffolding, no counterpart in the reference (which ships no data): the synthetic registration pair of SURVEY.md
 * 8d -- target and source drawn independently from one analytic height field over [0, L]^2 with a splitmix64 counter stream,
 * Gaussian noise of standard deviation `noise` on the source, the source moved by inv_T (the inverse of the transformation a
 * registration should recover) -- generated into DEVICE buffers (3 doubles per point; d_tgt_nrm, nullable, receives the
 * analytic unit normals of the target).  Asynchronous on the context's stream.  The same clouds as
 * open3d_amd/synthetic.py::make_pair up to the last digits of the device's libm.  For BASELINE config 5 (4096 pairs of
 * 200 k points: 59 GB that would otherwise cross PCIe). */
O3D_API int o3d_synthetic_pair(o3d_ctx *ctx, int64_t n_src, int64_t n_tgt, uint64_t seed, double L, double noise,
                               const double inv_T_colmajor[16], double *d_src, double *d_tgt, double *d_tgt_nrm);

/* Tuning aid, no counterpart in the reference: best-of-reps milliseconds of one copy of `bytes` of pageable host memory to
 * (to_device = 1) or from the device, through the library's staged path (staged = 1) or plain hipMemcpyAsync. */
O3D_API int o3d_debug_copy_bench(o3d_ctx *ctx, size_t bytes, int to_device, int staged, int reps, double *out_ms);

/* Self-check of the library's host <-> device copy path, no counterpart in the reference: pattern-filled buffers of `bytes`
 * go up and down back to back, `rounds` times; *out_mismatches = 8-byte words that came back different (must be 0). */
O3D_API int o3d_debug_copy_check(o3d_ctx *ctx, size_t bytes, int rounds, int64_t *out_mismatches);

#ifdef __cplusplus
}
#endif
#endif /* O3D_ICP_H_ */
