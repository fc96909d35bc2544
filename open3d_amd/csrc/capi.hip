// C ABI of libo3d_amd.so (declared in include/o3d_icp.h): context management and the host-side driver of the
// device-resident ICP loop (reference pipelines/registration/Registration.cpp:118-168).
#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <new>
#include <random>
#include <thread>
#include <vector>

#include "o3d_internal.h"

int o3d_launch_pass(o3d_ctx *ctx, double *dev_sums_out, int fuse_solve);
bool o3d_gang_eligible(o3d_ctx *ctx);
int o3d_launch_pass_gang(o3d_ctx *lead, o3d_ctx *const *slots, int n, bool upload, bool first);
int o3d_launch_solve(o3d_ctx *ctx, const double *dev_all_sums, int n_ranks);
int o3d_launch_scatter_corr(o3d_ctx *ctx, int32_t *dev_out);
int o3d_launch_info(o3d_ctx *ctx, double *dev_sums_out);
int o3d_launch_pairs(o3d_ctx *ctx, const double *d_src, int64_t n_src, const double *d_tgt, const double *d_nrm, int64_t n_tgt,
                     const int32_t *d_pairs, int64_t n_pairs, int estimation, int with_scaling, double *dev_out, int *dev_bad);
int o3d_launch_set_update(o3d_ctx *ctx, const double update[16]);
int o3d_noalloc_prepare_device(int device);
int o3d_launch_set_hypothesis(o3d_ctx *ctx, const double T[16]);
int o3d_launch_color_gather(o3d_ctx *ctx, const double *d_tgt_col, const double *d_grad, const double *d_src_col);
int o3d_color_gradient_impl(o3d_ctx *ctx, const double *d_xyz, const double *d_nrm, const double *d_col, int64_t n, double radius,
                            int max_nn, double *d_grad);
int o3d_voxel_down_sample_impl(o3d_ctx *ctx, const double *xyz, const double *nrm, const double *col, int64_t n, double voxel,
                               int mem, double *out_xyz, double *out_nrm, double *out_col, int32_t *out_vox, int64_t *out_m,
                               double **own_out = nullptr);
int o3d_estimate_normals_shard_impl(o3d_ctx *ctx, const double *xyz, int64_t n, int search_type, int knn, double radius, int mem,
                                    const double *oriented, int shard, int n_shards, double *out_part, int64_t *out_ng);
int o3d_estimate_normals_assemble_impl(o3d_ctx *ctx, const double *xyz, int64_t n, const double *all_parts, int n_shards, int mem,
                                       double *normals);
int o3d_estimate_normals_impl(o3d_ctx *ctx, const double *xyz, int64_t n, int search_type, int knn, double radius, int mem,
                              double *normals, int has_normals);

int o3d_neighbor_query_impl(o3d_ctx *ctx, int mode, const double *cloud_xyz, int64_t n_cloud, const double *q_xyz, int64_t n_q,
                            int knn, double radius, int mem, double *out, int32_t *out_count);

int o3d_fpfh_impl(o3d_ctx *ctx, const double *xyz, const double *nrm, int64_t n, int search_type, int knn, double radius, int mem,
                  double *out_feature);

static thread_local std::string g_err_noctx;

int o3d_fail(o3d_ctx *ctx, int status, const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  else g_err_noctx = buf;
  return status;
}

int o3d_reserve(o3d_ctx *ctx, void **ptr, size_t *cap, size_t bytes) {
  if (bytes <= *cap && *ptr) return O3D_OK;
  if (*ptr) {
    O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(*ptr);
    *ptr = nullptr;
    *cap = 0;
  }
  size_t want = bytes < 256 ? 256 : bytes;
  hipError_t e = hipMalloc(ptr, want);
  if (e == hipErrorOutOfMemory && !ctx->tmp_idle.empty()) {  // the idle per-call blocks go first
    (void)hipGetLastError();
    for (const o3d_ctx::TmpBlock &b : ctx->tmp_idle) (void)hipFree(b.p);
    ctx->tmp_idle.clear();
    ctx->tmp_idle_bytes = 0;
    e = hipMalloc(ptr, want);
  }
  O3D_HIP(ctx, e);
  *cap = want;
  return O3D_OK;
}

static constexpr size_t kTmpIdleMax = (size_t)16 << 30;

hipError_t o3d_tmp_alloc(o3d_ctx *ctx, void **out, size_t bytes) {
  const size_t want = bytes < 256 ? 256 : bytes;
  int best = -1;
  for (int i = 0; i < (int)ctx->tmp_idle.size(); i++) {
    const size_t cap = ctx->tmp_idle[i].cap;
    if (cap >= want && cap <= 2 * want + ((size_t)1 << 20) && (best < 0 || cap < ctx->tmp_idle[best].cap)) best = i;
  }
  if (best >= 0) {
    const o3d_ctx::TmpBlock b = ctx->tmp_idle[best];
    ctx->tmp_idle.erase(ctx->tmp_idle.begin() + best);
    ctx->tmp_idle_bytes -= b.cap;
    ctx->tmp_live.push_back(b);
    *out = b.p;
    return hipSuccess;
  }
  void *p = nullptr;
  hipError_t e = hipMalloc(&p, want);
  if (e == hipErrorOutOfMemory && !ctx->tmp_idle.empty()) {  // give the cache back and try once more
    (void)hipGetLastError();
    for (const o3d_ctx::TmpBlock &b : ctx->tmp_idle) (void)hipFree(b.p);
    ctx->tmp_idle.clear();
    ctx->tmp_idle_bytes = 0;
    e = hipMalloc(&p, want);
  }
  if (e != hipSuccess) return e;
  ctx->tmp_live.push_back({p, want});
  *out = p;
  return hipSuccess;
}

void o3d_tmp_free(o3d_ctx *ctx, void *p) {
  if (!p) return;
  for (size_t i = 0; i < ctx->tmp_live.size(); i++) {
    if (ctx->tmp_live[i].p != p) continue;
    const o3d_ctx::TmpBlock b = ctx->tmp_live[i];
    ctx->tmp_live.erase(ctx->tmp_live.begin() + (long)i);
    // the block may go to another call's copy streams next: nothing of this call may still be using it
    (void)hipStreamSynchronize(ctx->stream);
    if (b.cap > kTmpIdleMax / 2 || ctx->tmp_idle.size() >= 32) { (void)hipFree(b.p); return; }
    ctx->tmp_idle.push_back(b);
    ctx->tmp_idle_bytes += b.cap;
    while (ctx->tmp_idle_bytes > kTmpIdleMax) {  // oldest first
      ctx->tmp_idle_bytes -= ctx->tmp_idle.front().cap;
      (void)hipFree(ctx->tmp_idle.front().p);
      ctx->tmp_idle.erase(ctx->tmp_idle.begin());
    }
    return;
  }
  (void)hipFree(p);  // not one of ours
}

void o3d_tmp_release_all(o3d_ctx *ctx) {
  for (const o3d_ctx::TmpBlock &b : ctx->tmp_idle) (void)hipFree(b.p);
  for (const o3d_ctx::TmpBlock &b : ctx->tmp_live) (void)hipFree(b.p);
  ctx->tmp_idle.clear();
  ctx->tmp_live.clear();
  ctx->tmp_idle_bytes = 0;
}

int o3d_scratch(o3d_ctx *ctx, int slot, size_t bytes, void **out) {
  O3D_TRY(o3d_reserve(ctx, &ctx->scratch[slot], &ctx->scratch_cap[slot], bytes));
  *out = ctx->scratch[slot];
  return O3D_OK;
}

namespace {

struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    (void)hipSetDevice(dev);
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

// Eigen::Matrix4d::isIdentity() with dummy_precision 1e-12 (Registration.cpp:123,150)
bool is_identity(const double *T) {
  const double prec = 1e-12;
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 4; r++) {
      const double v = T[c * 4 + r];
      if (r == c) {
        const double m = fabs(v) < 1.0 ? fabs(v) : 1.0;
        if (!(fabs(v - 1.0) <= prec * m)) return false;
      } else if (!(fabs(v) <= prec)) {
        return false;
      }
    }
  return true;
}

// Bring a caller buffer to the device (no-op for device memory).  slot = scratch slot used for the copy.
int to_device(o3d_ctx *ctx, const void *p, size_t bytes, int mem, void **slot_ptr, size_t *slot_cap, const void **out) {
  if (mem == O3D_MEM_DEVICE || p == nullptr || bytes == 0) { *out = p; return O3D_OK; }
  O3D_TRY(o3d_reserve(ctx, slot_ptr, slot_cap, bytes));
  O3D_HIP(ctx, o3d_h2d(ctx, *slot_ptr, p, bytes));
  *out = *slot_ptr;
  return O3D_OK;
}

}  // namespace

struct o3d_staging {  // host->device staging of caller buffers (kept on the context, grow-only)
  void *p[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // tgt, nrm, src, pairs, tgt colours, src colours, gradients
  size_t cap[7] = {0, 0, 0, 0, 0, 0, 0};
};
static o3d_staging *staging_of(o3d_ctx *ctx);

extern "C" {

int o3d_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char *o3d_status_string(int s) {
  switch (s) {
    case O3D_OK: return "ok";
    case O3D_ERR_INVALID_DISTANCE: return "[Open3D ERROR] Invalid max_correspondence_distance.";
    case O3D_ERR_NEED_NORMALS:
      return "[Open3D ERROR] TransformationEstimationPointToPlane and TransformationEstimationColoredICP require "
             "pre-computed normal vectors for target PointCloud.";
    case O3D_ERR_VOXEL_SIZE: return "[Open3D ERROR] [VoxelDownSample] voxel_size <= 0.";
    case O3D_ERR_VOXEL_TOO_SMALL: return "[Open3D ERROR] [VoxelDownSample] voxel_size is too small.";
    case O3D_ERR_INVALID_ARG: return "invalid argument";
    case O3D_ERR_HIP: return "HIP runtime error";
    case O3D_ERR_OOM: return "out of device memory";
    case O3D_ERR_STATE: return "call out of order";
    case O3D_ERR_UNSUPPORTED: return "unsupported";
  }
  return "unknown status";
}

const char *o3d_last_error(const o3d_ctx *ctx) { return ctx ? ctx->err.c_str() : g_err_noctx.c_str(); }

}  // extern "C"

// RCCL, bound at run time (dlopen): the library has no link-time dependency on it, and a process that has loaded a copy
// already -- torch ships its own -- keeps using that one.  Only what the multi context needs: one communicator per listed
// device in this process (ncclCommInitAll) and the all-gather of the ranks' 32 sums.
namespace {
typedef void *rcclComm_t;
constexpr int kRcclDouble = 8;  // ncclFloat64 / ncclDouble (rccl.h)
struct RcclApi {
  int (*CommInitAll)(rcclComm_t *, int, const int *) = nullptr;
  int (*CommDestroy)(rcclComm_t) = nullptr;
  int (*CommAbort)(rcclComm_t) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, rcclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  bool ok = false;
};
const RcclApi &rccl_api() {
  static RcclApi api = [] {
    RcclApi a;
    void *h = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so"})
      if (!h) h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
      if (!h) h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (!h) return a;
    a.CommInitAll = reinterpret_cast<decltype(a.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    a.CommAbort = reinterpret_cast<decltype(a.CommAbort)>(dlsym(h, "ncclCommAbort"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(h, "ncclAllGather"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    a.ok = a.CommInitAll && a.CommDestroy && a.CommAbort && a.AllGather && a.GetErrorString;
    return a;
  }();
  return api;
}
}  // namespace

struct o3d_ctx_full : o3d_ctx {
  std::recursive_mutex mu;  // one call in flight per context: concurrent callers of the same context are serialised
  o3d_staging stage;
  std::vector<o3d_ctx *> workers;  // contexts of the batch worker threads (created on first use, kept)
  // o3d_ctx_create_multi: this context is rank 0 of `ranks` (ranks[0] == this); empty for an ordinary context
  std::vector<o3d_ctx *> ranks;
  double *mailbox = nullptr;            // pinned host memory mapped into every device: [parity][rank][32 sums]
  std::vector<double *> mailbox_dev;    // the same memory as each rank's device sees it
  std::vector<hipEvent_t> xev;          // [parity * G + rank]: the rank's sums of this parity are in the mailbox
  // distinct devices: the exchange is an RCCL all-gather over xGMI on each rank's stream instead of the mailbox
  std::vector<rcclComm_t> comms;        // one per rank (ncclCommInitAll), empty when the mailbox is in use
  std::vector<double *> xsend, xrecv;   // per rank, on its device: its 32 sums / the G x 32 table
};
static o3d_staging *staging_of(o3d_ctx *ctx) { return &static_cast<o3d_ctx_full *>(ctx)->stage; }
#define O3D_LOCK(ctx) std::lock_guard<std::recursive_mutex> lock__(static_cast<o3d_ctx_full *>(ctx)->mu)

// Device copies of caller buffers for the duration of one call (host memory: upload; device memory: used in place).
namespace {
struct DevBufs {
  std::vector<void *> owned;
  o3d_ctx *owner = nullptr;
  ~DevBufs() {
    for (void *p : owned) o3d_tmp_free(owner, p);
  }
  template <typename T>
  int in(o3d_ctx *ctx, const T *p, size_t count, int mem, const T **out) {
    *out = p;
    if (!p || mem != O3D_MEM_HOST) return O3D_OK;
    void *d = nullptr;
    owner = ctx;
    O3D_HIP(ctx, o3d_tmp_alloc(ctx, &d, sizeof(T) * (count ? count : 1)));
    owned.push_back(d);
    if (count) O3D_HIP(ctx, o3d_h2d(ctx, d, p, sizeof(T) * count));
    *out = (const T *)d;
    return O3D_OK;
  }
  template <typename T>
  int tmp(o3d_ctx *ctx, size_t count, T **out) {
    void *d = nullptr;
    owner = ctx;
    O3D_HIP(ctx, o3d_tmp_alloc(ctx, &d, sizeof(T) * (count ? count : 1)));
    owned.push_back(d);
    *out = (T *)d;
    return O3D_OK;
  }
};
}  // namespace


extern "C" {

int o3d_ctx_create(int device_id, o3d_ctx **out_ctx) {
  if (!out_ctx) return O3D_ERR_INVALID_ARG;
  *out_ctx = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return o3d_fail(nullptr, O3D_ERR_HIP, "no HIP device available (%s); libo3d_amd has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  if (device_id < 0 || device_id >= ndev) return o3d_fail(nullptr, O3D_ERR_INVALID_ARG, "device %d out of range [0,%d)", device_id, ndev);
  o3d_ctx_full *ctx = new (std::nothrow) o3d_ctx_full();
  if (!ctx) return O3D_ERR_OOM;
  ctx->device = device_id;
  DeviceGuard guard(device_id);
  auto bail = [&](hipError_t err, const char *what) {
    o3d_fail(nullptr, O3D_ERR_HIP, "%s: %s", what, hipGetErrorString(err));
    o3d_ctx_destroy(ctx);
    return (int)O3D_ERR_HIP;
  };
  if ((e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
  ctx->stream = ctx->own_stream;
  if ((e = hipHostMalloc(&ctx->pinned, 4096, hipHostMallocDefault)) != hipSuccess) return bail(e, "hipHostMalloc");
  if ((e = hipMalloc((void **)&ctx->state, sizeof(IcpState))) != hipSuccess) return bail(e, "hipMalloc state");
  // rows of the search kernel (or of any other per-workgroup reduction), then the rows of k_icp_cert
  if ((e = hipMalloc((void **)&ctx->partials, sizeof(double) * O3D_NSUMS * 2 * O3D_MAX_GRID)) != hipSuccess) return bail(e, "hipMalloc partials");
  if ((e = hipMalloc((void **)&ctx->sums, sizeof(double) * O3D_NSUMS)) != hipSuccess) return bail(e, "hipMalloc sums");
  if ((e = hipMalloc((void **)&ctx->counters, sizeof(unsigned long long) * O3D_NCOUNTERS)) != hipSuccess) return bail(e, "hipMalloc counters");
  if ((e = hipMalloc((void **)&ctx->work_count, sizeof(uint32_t) * 8 * O3D_MAX_GRID)) != hipSuccess) return bail(e, "hipMalloc work_count");
  if ((e = hipMemsetAsync(ctx->counters, 0, sizeof(unsigned long long) * O3D_NCOUNTERS, ctx->stream)) != hipSuccess) return bail(e, "hipMemsetAsync");
  if ((e = hipMalloc((void **)&ctx->gsums, sizeof(double) * O3D_NSUMS * 32 + sizeof(unsigned) * 64)) != hipSuccess) return bail(e, "hipMalloc gsums");
  ctx->tickets = reinterpret_cast<unsigned *>(ctx->gsums + O3D_NSUMS * 32);
  if ((e = hipMemsetAsync(ctx->tickets, 0, sizeof(unsigned) * 64, ctx->stream)) != hipSuccess) return bail(e, "hipMemsetAsync");
  // the scratch address of o3d_compute_transformation_noalloc on this device, so that its first call may come under capture
  if (o3d_noalloc_prepare_device(device_id) != O3D_OK) return bail(hipErrorInvalidSymbol, "hipGetSymbolAddress(scratch)");
  *out_ctx = ctx;
  return O3D_OK;
}

int o3d_ctx_create_multi(int n_devices, const int *device_ids, o3d_ctx **out_ctx) {
  if (!out_ctx) return O3D_ERR_INVALID_ARG;
  *out_ctx = nullptr;
  if (n_devices < 1 || n_devices > 64 || !device_ids) return o3d_fail(nullptr, O3D_ERR_INVALID_ARG, "o3d_ctx_create_multi: 1..64 devices");
  o3d_ctx *first = nullptr;
  O3D_TRY(o3d_ctx_create(device_ids[0], &first));
  // (O3D_MULTI_ONE_RANK=1: a "multi" context of ONE rank, which runs the whole sharded path -- communicator, all-gather,
  // solve over the gathered table -- on a box with a single GPU; a test hook)
  if (n_devices == 1 && !getenv("O3D_MULTI_ONE_RANK")) { *out_ctx = first; return O3D_OK; }
  o3d_ctx_full *ctx = static_cast<o3d_ctx_full *>(first);
  const int G = n_devices;
  auto bail = [&](int st, const char *what, hipError_t e) {
    o3d_fail(nullptr, st, "o3d_ctx_create_multi: %s%s%s", what, e != hipSuccess ? ": " : "", e != hipSuccess ? hipGetErrorString(e) : "");
    o3d_ctx_destroy(ctx);
    return st;
  };
  ctx->ranks.push_back(ctx);
  for (int r = 1; r < G; r++) {
    o3d_ctx *c = nullptr;
    const int st = o3d_ctx_create(device_ids[r], &c);
    if (st != O3D_OK) { o3d_ctx_destroy(ctx); return st; }  // the message of o3d_ctx_create stands
    ctx->ranks.push_back(c);
  }
  // caller buffers in device memory live on one GPU and are read by the kernels of all of them
  for (int a = 0; a < G; a++)
    for (int b = 0; b < G; b++) {
      const int da = device_ids[a], db = device_ids[b];
      if (da == db) continue;
      DeviceGuard g(da);
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, da, db) != hipSuccess || !can) { (void)hipGetLastError(); continue; }  // host buffers still work
      const hipError_t e = hipDeviceEnablePeerAccess(db, 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return bail(O3D_ERR_HIP, "hipDeviceEnablePeerAccess", e);
      (void)hipGetLastError();
    }
  // The per-iteration exchange: an RCCL all-gather of 32 doubles per rank over xGMI when every rank has a GPU
  // of its own; RCCL refuses a device listed twice, so the one-GPU rehearsal of the sharded path (and a process where RCCL
  // cannot be loaded, O3D_NO_RCCL=1) exchanges through the pinned host mailbox below.
  bool distinct = true;
  for (int a = 0; a < G; a++)
    for (int b = a + 1; b < G; b++)
      if (device_ids[a] == device_ids[b]) distinct = false;
  if (distinct && !getenv("O3D_NO_RCCL") && rccl_api().ok) {
    ctx->comms.assign((size_t)G, nullptr);
    const int rc = rccl_api().CommInitAll(ctx->comms.data(), G, device_ids);
    if (rc != 0) {
      ctx->comms.clear();
      return bail(O3D_ERR_HIP, rccl_api().GetErrorString(rc), hipSuccess);
    }
    ctx->xsend.assign((size_t)G, nullptr);
    ctx->xrecv.assign((size_t)G, nullptr);
    for (int r = 0; r < G; r++) {
      DeviceGuard g(device_ids[r]);
      hipError_t e2 = hipMalloc((void **)&ctx->xsend[(size_t)r], sizeof(double) * O3D_NSUMS);
      if (e2 == hipSuccess) e2 = hipMalloc((void **)&ctx->xrecv[(size_t)r], sizeof(double) * O3D_NSUMS * (size_t)G);
      if (e2 != hipSuccess) return bail(O3D_ERR_HIP, "hipMalloc(exchange buffers)", e2);
    }
    *out_ctx = ctx;
    return O3D_OK;
  }
  hipError_t e = hipHostMalloc((void **)&ctx->mailbox, sizeof(double) * 2 * (size_t)G * O3D_NSUMS, hipHostMallocPortable | hipHostMallocMapped);
  if (e != hipSuccess) return bail(O3D_ERR_HIP, "hipHostMalloc(mailbox)", e);
  memset(ctx->mailbox, 0, sizeof(double) * 2 * (size_t)G * O3D_NSUMS);
  ctx->mailbox_dev.assign((size_t)G, nullptr);
  ctx->xev.assign(2 * (size_t)G, nullptr);
  for (int r = 0; r < G; r++) {
    DeviceGuard g(device_ids[r]);
    void *dp = nullptr;
    if ((e = hipHostGetDevicePointer(&dp, ctx->mailbox, 0)) != hipSuccess) return bail(O3D_ERR_HIP, "hipHostGetDevicePointer(mailbox)", e);
    ctx->mailbox_dev[(size_t)r] = static_cast<double *>(dp);
    for (int par = 0; par < 2; par++)
      if ((e = hipEventCreateWithFlags(&ctx->xev[(size_t)(par * G + r)], hipEventDisableTiming)) != hipSuccess)
        return bail(O3D_ERR_HIP, "hipEventCreate", e);
  }
  *out_ctx = ctx;
  return O3D_OK;
}

int o3d_ctx_num_devices(const o3d_ctx *c) {
  if (!c) return 0;
  const o3d_ctx_full *ctx = static_cast<const o3d_ctx_full *>(c);
  return ctx->ranks.empty() ? 1 : (int)ctx->ranks.size();
}

void o3d_ctx_destroy(o3d_ctx *c) {
  if (!c) return;
  o3d_ctx_full *ctx = static_cast<o3d_ctx_full *>(c);
  for (o3d_ctx *w : ctx->workers) o3d_ctx_destroy(w);
  ctx->workers.clear();
  for (size_t k = 0; k < ctx->xev.size(); k++)
    if (ctx->xev[k]) {
      DeviceGuard g(ctx->ranks[k % ctx->ranks.size()]->device);
      (void)hipEventDestroy(ctx->xev[k]);
    }
  ctx->xev.clear();
  for (size_t r = 0; r < ctx->comms.size(); r++)
    if (ctx->comms[r]) (void)rccl_api().CommDestroy(ctx->comms[r]);
  ctx->comms.clear();
  for (size_t r = 0; r < ctx->xsend.size(); r++) {
    DeviceGuard g(ctx->ranks[r]->device);
    if (ctx->xsend[r]) (void)hipFree(ctx->xsend[r]);
    if (ctx->xrecv[r]) (void)hipFree(ctx->xrecv[r]);
  }
  ctx->xsend.clear();
  ctx->xrecv.clear();
  for (size_t r = 1; r < ctx->ranks.size(); r++) o3d_ctx_destroy(ctx->ranks[r]);
  ctx->ranks.clear();
  if (ctx->mailbox) (void)hipHostFree(ctx->mailbox);
  ctx->mailbox = nullptr;
  DeviceGuard guard(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  void *bufs[] = {ctx->tgt_local, ctx->tgt_sorted, ctx->nrm_sorted, ctx->cell_start, ctx->src_sorted, ctx->src_perm, ctx->corr_sorted,
                  ctx->corr_out,   ctx->pos_sorted, ctx->lb_sorted, ctx->work, ctx->work_count, ctx->state,      ctx->partials,   ctx->sums,       ctx->log,
                  ctx->counters,   ctx->gsums,      ctx->brick_dir,  ctx->tn_sorted,  ctx->d_gang, ctx->tgt_index, ctx->tail_check_buf};
  for (void *b : bufs)
    if (b) (void)hipFree(b);
  for (int i = 0; i < 8; i++)
    if (ctx->scratch[i]) (void)hipFree(ctx->scratch[i]);
  for (int i = 0; i < 7; i++)
    if (ctx->stage.p[i]) (void)hipFree(ctx->stage.p[i]);
  if (ctx->src_orig) (void)hipFree(ctx->src_orig);
  if (ctx->tcol_sorted) (void)hipFree(ctx->tcol_sorted);
  if (ctx->sint_sorted) (void)hipFree(ctx->sint_sorted);
  for (hipEvent_t e : ctx->prof_ev)
    if (e) (void)hipEventDestroy(e);
  if (ctx->poll_ev) (void)hipEventDestroy(ctx->poll_ev);
  for (hipEvent_t e : ctx->setup_ev)
    if (e) (void)hipEventDestroy(e);
  o3d_tmp_release_all(ctx);
  o3d_xfer_destroy(ctx);
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  if (ctx->pinned_gang) (void)hipHostFree(ctx->pinned_gang);
  if (ctx->gang_sort_own) {
    for (uint32_t *b : {ctx->gang_sort_own->keys, ctx->gang_sort_own->vals, ctx->gang_sort_own->tkeys, ctx->gang_sort_own->tvals,
                        ctx->gang_sort_own->hist})
      if (b) (void)hipFree(b);
    delete ctx->gang_sort_own;
    ctx->gang_sort_own = nullptr;
  }
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
}

int o3d_ctx_set_stream(o3d_ctx *ctx, void *hip_stream) {
  if (!ctx) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  DeviceGuard guard(ctx->device);
  O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->stream = (hipStream_t)hip_stream;  // NULL is the HIP null stream, which is what torch's default stream is
  return O3D_OK;
}

int o3d_ctx_use_own_stream(o3d_ctx *ctx) {
  if (!ctx) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  DeviceGuard guard(ctx->device);
  O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->stream = ctx->own_stream;
  return O3D_OK;
}

// The fp32 prefilter's error bound is checked on every winner (k_icp_search_coop); a violation makes the result suspect.
// The counter lives in the context's own device memory and travels with the state copy of o3d_icp_finish / o3d_icp_fetch
// (kViolationSlot of the pinned page): no symbol copy, no synchronisation beyond the one the caller makes anyway.
static constexpr size_t kViolationSlot = 3584;  // byte offset in ctx->pinned
static hipError_t enqueue_violation_read(o3d_ctx *ctx) {
  return hipMemcpyAsync((char *)ctx->pinned + kViolationSlot, ctx->counters + O3D_CTR_VIOLATIONS, sizeof(unsigned long long),
                        hipMemcpyDeviceToHost, ctx->stream);
}
// after the stream has been synchronised
static int check_bound_violations(o3d_ctx *ctx) {
  unsigned long long v = 0;
  memcpy(&v, (char *)ctx->pinned + kViolationSlot, sizeof v);
  if (v) {
    O3D_HIP(ctx, hipMemsetAsync(ctx->counters + O3D_CTR_VIOLATIONS, 0, sizeof(unsigned long long), ctx->stream));
    return o3d_fail(ctx, O3D_ERR_STATE, "internal: %llu fp32 prefilter winner(s) failed their fp64 confirmation; the result is not "
                    "trustworthy (o3d_debug_set(ctx, 1, 0) selects the all-fp64 path)", v);
  }
  return O3D_OK;
}

// o3d_icp_setup, plus the coloured-ICP inputs (src_col / tgt_col / lambda_geometric; out_grad nullable, `mem` space).
static int icp_setup_impl(o3d_ctx *ctx, const double *src_xyz, int64_t n_src, const double *tgt_xyz, const double *tgt_nrm, int64_t n_tgt,
                          double max_dist, const double init[16], int estimation, int with_scaling, double rel_fitness,
                          double rel_rmse, int max_iter, int mem, int64_t n_src_global, const double *src_col, const double *tgt_col,
                          double lambda_geometric, double *out_grad) {
  if (!ctx) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  ctx->icp_ready = 0;
  if (!(max_dist > 0.0)) return o3d_fail(ctx, O3D_ERR_INVALID_DISTANCE, "%s", o3d_status_string(O3D_ERR_INVALID_DISTANCE));
  if (estimation != O3D_EST_POINT_TO_POINT && estimation != O3D_EST_POINT_TO_PLANE && estimation != O3D_EST_COLORED_ICP)
    return o3d_fail(ctx, O3D_ERR_UNSUPPORTED, "estimation type %d is not on the device path", estimation);
  if (estimation != O3D_EST_POINT_TO_POINT && (!tgt_nrm || n_tgt <= 0))
    return o3d_fail(ctx, O3D_ERR_NEED_NORMALS, "%s", o3d_status_string(O3D_ERR_NEED_NORMALS));
  if (estimation == O3D_EST_COLORED_ICP && (!src_col || !tgt_col))
    return o3d_fail(ctx, O3D_ERR_INVALID_ARG, "coloured ICP set up without colours");
  if (n_src < 0 || n_tgt < 0 || !init || (n_src > 0 && !src_xyz) || (n_tgt > 0 && !tgt_xyz) || max_iter < 0)
    return o3d_fail(ctx, O3D_ERR_INVALID_ARG, "bad sizes or null buffers");
  if (n_src >= ((int64_t)1 << 31) || n_tgt >= ((int64_t)1 << 31))
    return o3d_fail(ctx, O3D_ERR_UNSUPPORTED, "clouds are limited to 2^31-1 points per GPU (the reference indexes with int)");
  if (n_src_global < n_src) n_src_global = n_src;
  DeviceGuard guard(ctx->device);
  o3d_staging *sg = staging_of(ctx);

  // three events kept on the context: set-up timing is read lazily by o3d_icp_get_setup_ms (a read here would drain the
  // stream once more per call; the batch workers make thousands of small calls)
  // (not for the slots of a batch's gangs: nobody reads their timing, and every call into the HIP runtime counts there --
  // eight worker threads were spending 1 ms of host time per set-up, most of it waiting for each other in the runtime)
  const bool timed = ctx->gang_slots <= 1;
  hipEvent_t *ev = ctx->setup_ev;
  for (int k = 0; k < 3 && timed; k++)
    if (!ev[k]) O3D_HIP(ctx, hipEventCreate(&ev[k]));
  if (timed) O3D_HIP(ctx, hipEventRecord(ev[0], ctx->stream));
  // the statistics counters start at zero (the violation counter is cleared by whoever reads it non-zero)
  if (timed || ctx->tile_stats)
    O3D_HIP(ctx, hipMemsetAsync(ctx->counters, 0, sizeof(unsigned long long) * O3D_NCOUNTERS, ctx->stream));

  const void *d_tgt, *d_nrm, *d_src;
  O3D_TRY(to_device(ctx, tgt_xyz, sizeof(double) * 3 * (size_t)n_tgt, mem, &sg->p[0], &sg->cap[0], &d_tgt));
  O3D_TRY(to_device(ctx, estimation != O3D_EST_POINT_TO_POINT ? tgt_nrm : nullptr, sizeof(double) * 3 * (size_t)n_tgt, mem,
                    &sg->p[1], &sg->cap[1], &d_nrm));
  const void *d_tcol = nullptr, *d_scol = nullptr;
  if (estimation == O3D_EST_COLORED_ICP) {
    // InitializePointCloudForColoredICP (ColoredICP.cpp:68-123, :209): gradients over Hybrid(2 * max_distance, 30); this
    // uses the context's grid, which is then rebuilt with the cell size of the correspondence search
    O3D_TRY(to_device(ctx, tgt_col, sizeof(double) * 3 * (size_t)n_tgt, mem, &sg->p[4], &sg->cap[4], &d_tcol));
    O3D_TRY(to_device(ctx, src_col, sizeof(double) * 3 * (size_t)n_src, mem, &sg->p[5], &sg->cap[5], &d_scol));
    O3D_TRY(o3d_reserve(ctx, &sg->p[6], &sg->cap[6], sizeof(double) * 3 * (size_t)(n_tgt > 0 ? n_tgt : 1)));
    O3D_TRY(o3d_color_gradient_impl(ctx, (const double *)d_tgt, (const double *)d_nrm, (const double *)d_tcol, n_tgt, max_dist * 2.0, 30,
                                    (double *)sg->p[6]));
    if (out_grad && n_tgt > 0)
      O3D_HIP(ctx, hipMemcpyAsync(out_grad, sg->p[6], sizeof(double) * 3 * (size_t)n_tgt,
                                  mem == O3D_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
    if (lambda_geometric < 0.0 || lambda_geometric > 1.0) lambda_geometric = 0.968;  // ColoredICP.cpp:50-53
    ctx->slg = sqrt(lambda_geometric);
    ctx->slp = sqrt(1.0 - lambda_geometric);
  }
  O3D_TRY(o3d_build_target_grid(ctx, (const double *)d_tgt, (const double *)d_nrm, n_tgt, max_dist, 1));
  if (timed) O3D_HIP(ctx, hipEventRecord(ev[1], ctx->stream));

  const int apply_init = is_identity(init) ? 0 : 1;  // Registration.cpp:150-152
  O3D_TRY(to_device(ctx, src_xyz, sizeof(double) * 3 * (size_t)n_src, mem, &sg->p[2], &sg->cap[2], &d_src));
  O3D_TRY(o3d_sort_source(ctx, (const double *)d_src, n_src, init, apply_init));
  if (estimation == O3D_EST_COLORED_ICP) {
    O3D_TRY(o3d_reserve(ctx, (void **)&ctx->tcol_sorted, &ctx->cap_tcol, sizeof(double4) * (size_t)(n_tgt > 0 ? n_tgt : 1)));
    O3D_TRY(o3d_reserve(ctx, (void **)&ctx->sint_sorted, &ctx->cap_sint, sizeof(double) * (size_t)(n_src > 0 ? n_src : 1)));
    O3D_TRY(o3d_launch_color_gather(ctx, (const double *)d_tcol, (const double *)sg->p[6], (const double *)d_scol));
  }
  if (timed) O3D_HIP(ctx, hipEventRecord(ev[2], ctx->stream));

  // loop state
  ctx->estimation = estimation;
  ctx->has_tgt_normals = (estimation != O3D_EST_POINT_TO_POINT) ? 1 : 0;
  ctx->thr = (double)(float)(max_dist * max_dist);  // geometry/KDTreeFlann.cpp:148
  ctx->max_iter = max_iter;
  int64_t g = (n_src + O3D_BLOCK - 1) / O3D_BLOCK;
  if (g < 1) g = 1;
  if (g > O3D_MAX_GRID) g = O3D_MAX_GRID;
  g = (g + 7) / 8 * 8;  // the pass kernels map workgroup -> chunk per XCD (8 XCDs)
  ctx->pass_grid = (int)g;
  // k_icp_search_coop holds four workgroups per CU: one resident round of 4 x 256 workgroups, no tail
  int64_t gs = g < O3D_NUM_CU * 4 ? g : O3D_NUM_CU * 4;
  if (ctx->gang_slots > 1) {
    // one of several registrations advancing in lock step: the GANG fills the one resident round, every wavefront of a
    // slot walks several trips (a launch of slots x 782 one-trip workgroups spends most of its time in their prologues
    // and epilogues: measured 1850 pairs/s against 2300 one by one)
    int64_t share = (O3D_NUM_CU * 4 / ctx->gang_slots) / 8 * 8;
    if (share < 8) share = 8;
    if (gs > share) gs = share;
  }
  if (const char *e = getenv("O3D_SEARCH_GRID")) {  // tuning aid
    const long v = atol(e);
    if (v >= 8 && v <= O3D_MAX_GRID) gs = (v + 7) / 8 * 8;
  }

  ctx->use_seed = getenv("O3D_NO_SEED") ? 0 : 1;  // tuning aid
  // Certificates cost a pass over the source of their own: with the pass as one launch (k_icp_pass) they cost a small
  // cloud nothing (steady-state iteration, without / with: 100 k points 22.8 / 22.6 us, 200 k 27.6 / 27.0) and pay from there
  // on (300 k: 40.0 / 31.0 us; as a launch of their own they only paid from 400 k points on).  In a gang
  // (o3d_registration_icp_batch) they always pay (4096 pairs of 200 k points: 2000 pairs/s without, 3100 with, at the time).
  ctx->use_cert = getenv("O3D_NO_CERT") ? 0 : (getenv("O3D_CERT") ? 1 : ((n_src >= 200000 || ctx->gang_slots > 1) ? 1 : 0));  // tuning aids
  // with certificates the pass is one launch (k_icp_pass) when its grid is a multiple of 32 workgroups (icp.hip): rounded up
  // (the surplus workgroups own empty segments), or down for a gang's share of the resident round
  if (ctx->use_cert && gs % 32 != 0) gs = (ctx->gang_slots > 1 && gs > 32) ? gs / 32 * 32 : (gs + 31) / 32 * 32;
  ctx->search_grid = (int)gs;
  ctx->use_tail = getenv("O3D_NO_TAIL") ? 0 : 1;  // tuning aid
  {  // tuning aids, read per set-up like the others (a test can change them between registrations)
    const char *e = getenv("O3D_STREAM");
    ctx->use_stream = e ? (atoi(e) ? 1 : 0) : 1;
    e = getenv("O3D_ONE_LAUNCH");
    ctx->use_one_launch = (e && atoi(e) == 0) ? 0 : 1;
    e = getenv("O3D_TAIL_CHECK");
    ctx->tail_check = e ? atoi(e) : 0;  // (2: one line on stderr per check)
  }
  if (const char *e = getenv("O3D_CERT_GAP")) {
    const double v = atof(e);
    if (v >= 0.0 && v <= 1.0) ctx->cert_gap = v;
  }
  const int rows = max_iter + 1;
  if (rows > ctx->log_rows || !ctx->log) {
    if (ctx->log) (void)hipFree(ctx->log);
    ctx->log = nullptr;
    O3D_HIP(ctx, hipMalloc((void **)&ctx->log, sizeof(double) * O3D_LOG_STRIDE * (size_t)rows));
    ctx->log_rows = rows;
  }
  IcpState *hs = reinterpret_cast<IcpState *>((char *)ctx->pinned + 1024);
  memset(hs, 0, sizeof(IcpState));
  for (int i = 0; i < 16; i++) { hs->T[i] = init[i]; hs->update[i] = init[i]; }
  hs->rel_fitness = rel_fitness;
  hs->rel_rmse = rel_rmse;
  ctx->rel_fitness = rel_fitness;
  ctx->rel_rmse = rel_rmse;
  hs->n_src_global = (double)n_src_global;
  ctx->n_src_global = n_src_global;
  hs->max_iter = max_iter;
  hs->apply_update = apply_init;
  hs->estimation = estimation;
  hs->with_scaling = with_scaling;
  // fixed origin for the Umeyama moments: centre of the target's bounding box (cuts cancellation, any value is valid)
  double diag2 = 0.0;
  for (int d = 0; d < 3; d++) {
    const double ext = ctx->gp.h * (double)(d == 0 ? ctx->gp.nx : (d == 1 ? ctx->gp.ny : ctx->gp.nz));
    hs->centre[d] = ctx->gp.gmin[d] + 0.5 * ext;
    diag2 += 0.25 * ext * ext;
  }
  hs->radius = sqrt(diag2);
  // k_icp_cert stands aside while the pending update moves the cloud farther than twice the slack the bounds were given (the
  // search's extra reach): beyond that too few certificates survive to pay for the kernel.  Measured on fresh 30-iteration
  // registrations at 10 M points with factors 0.5 / 1 / 2 / 4 / 16: point-to-point, which keeps sliding, 17.1 / 16.2 / 16.2 /
  // 16.3 / 16.3 ms of kernels (at 0.5 the first nine passes searched everything); point-to-plane 9.2-9.3 ms whatever the factor
  double skip_factor = 2.0;
  if (const char *e = getenv("O3D_CERT_SKIP")) {  // tuning aid
    const double v = atof(e);
    if (v > 0.0) skip_factor = v;
  }
  hs->cert_skip_above = skip_factor * ctx->cert_gap * ctx->gp.h;
  double seed_factor = 0.25;
  if (const char *e = getenv("O3D_SEED_SKIP")) {  // tuning aid
    const double v = atof(e);
    if (v > 0.0) seed_factor = v;
  }
  hs->seed_skip_above = seed_factor * ctx->gp.h;
  // (the staging page is rewritten by the next set-up on this context at the earliest, and every path to it --
  // o3d_icp_finish, o3d_icp_poll, o3d_icp_fetch, an error -- has synchronised the stream by then)
  O3D_HIP(ctx, o3d_h2d(ctx, ctx->state, hs, sizeof(IcpState)));
  ctx->enq_passes = 0;
  ctx->sums_valid = 0;
  ctx->hyp_mode = 0;
  ctx->icp_ready = 1;
  return O3D_OK;
}

int o3d_icp_setup(o3d_ctx *ctx, const double *src_xyz, int64_t n_src, const double *tgt_xyz, const double *tgt_nrm, int64_t n_tgt,
                  double max_dist, const double init[16], int estimation, int with_scaling, double rel_fitness, double rel_rmse,
                  int max_iter, int mem, int64_t n_src_global) {
  if (!ctx) return O3D_ERR_INVALID_ARG;
  if (estimation != O3D_EST_POINT_TO_POINT && estimation != O3D_EST_POINT_TO_PLANE)
    return o3d_fail(ctx, O3D_ERR_UNSUPPORTED, "estimation type %d is not on the device path of o3d_icp_setup", estimation);
  O3D_TRY(icp_setup_impl(ctx, src_xyz, n_src, tgt_xyz, tgt_nrm, n_tgt, max_dist, init, estimation, with_scaling, rel_fitness, rel_rmse,
                         max_iter, mem, n_src_global, nullptr, nullptr, 0.0, nullptr));
  if (mem == O3D_MEM_HOST) {  // the caller's host buffers are free when this returns; device buffers are read in stream order
    DeviceGuard guard(ctx->device);
    O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  return O3D_OK;
}

int o3d_icp_pass(o3d_ctx *ctx) {
  if (!ctx) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  if (!ctx->icp_ready) return o3d_fail(ctx, O3D_ERR_STATE, "o3d_icp_pass before o3d_icp_setup");
  DeviceGuard guard(ctx->device);
  ctx->enq_passes++;
  ctx->sums_valid = 1;
  return o3d_launch_pass(ctx, ctx->sums, 0);
}

int o3d_icp_pass_to(o3d_ctx *ctx, double *dev_sums_out) {
  if (!ctx || !dev_sums_out) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  if (!ctx->icp_ready) return o3d_fail(ctx, O3D_ERR_STATE, "o3d_icp_pass_to before o3d_icp_setup");
  DeviceGuard guard(ctx->device);
  ctx->enq_passes++;
  return o3d_launch_pass(ctx, dev_sums_out, 0);
}

int o3d_icp_step(o3d_ctx *ctx) {
  if (!ctx) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  if (!ctx->icp_ready) return o3d_fail(ctx, O3D_ERR_STATE, "o3d_icp_step before o3d_icp_setup");
  DeviceGuard guard(ctx->device);
  ctx->enq_passes++;
  ctx->sums_valid = 1;
  return o3d_launch_pass(ctx, ctx->sums, 1);
}

int o3d_debug_set(o3d_ctx *ctx, int force_generic_pass, int collect_tile_stats) {
  if (!ctx) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  ctx->use_tile = force_generic_pass ? 0 : 1;
  ctx->tile_stats = collect_tile_stats;
  return O3D_OK;
}

int o3d_debug_inject_bound_violation(o3d_ctx *ctx, unsigned int count) {
  if (!ctx) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  DeviceGuard guard(ctx->device);
  unsigned long long *h = reinterpret_cast<unsigned long long *>((char *)ctx->pinned + kViolationSlot + 192);
  *h = count;
  O3D_HIP(ctx, hipMemcpyAsync(ctx->counters + O3D_CTR_VIOLATIONS, h, sizeof *h, hipMemcpyHostToDevice, ctx->stream));
  O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return O3D_OK;
}

int o3d_debug_get_source_perm(o3d_ctx *ctx, uint32_t *host_out) {
  if (!ctx || !host_out) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  if (!ctx->icp_ready) return o3d_fail(ctx, O3D_ERR_STATE, "no source loaded");
  DeviceGuard guard(ctx->device);
  O3D_HIP(ctx, o3d_d2h(ctx, host_out, ctx->src_perm, sizeof(uint32_t) * (size_t)ctx->n_src));
  O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return O3D_OK;
}

int o3d_debug_get_match_positions(o3d_ctx *ctx, uint32_t *host_out) {
  if (!ctx || !host_out) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  if (!ctx->icp_ready || !ctx->pos_sorted) return o3d_fail(ctx, O3D_ERR_STATE, "no source loaded");
  DeviceGuard guard(ctx->device);
  O3D_HIP(ctx, o3d_d2h(ctx, host_out, ctx->pos_sorted, sizeof(uint32_t) * (size_t)ctx->n_src));
  O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return O3D_OK;
}

int o3d_debug_tile_stats(o3d_ctx *ctx, uint64_t out[8], int reset) {
  if (!ctx || !out) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  DeviceGuard guard(ctx->device);
  unsigned long long *t = reinterpret_cast<unsigned long long *>((char *)ctx->pinned + kViolationSlot + 64);
  O3D_HIP(ctx, hipMemcpyAsync(t, ctx->counters, sizeof(unsigned long long) * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (reset) O3D_HIP(ctx, hipMemsetAsync(ctx->counters, 0, sizeof(unsigned long long) * 8, ctx->stream));
  O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < 8; i++) out[i] = t[i];
  return O3D_OK;
}

int o3d_debug_grid_info(const o3d_ctx *ctx, double out[10]) {
  if (!ctx || !out) return O3D_ERR_INVALID_ARG;
  out[0] = ctx->gp.h; out[1] = ctx->gp.nx; out[2] = ctx->gp.ny; out[3] = ctx->gp.nz; out[4] = ctx->gp.nxp;
  out[8] = ctx->gp.brick_bits ? (double)(1 << ctx->gp.brick_bits) : 0.0; out[9] = (double)ctx->n_bricks;
  out[5] = ctx->gp.margin; out[6] = (ctx->tile_ok && ctx->use_tile && ctx->gp.margin) ? 1.0 : 0.0; out[7] = (double)ctx->n_tgt_grid;
  return O3D_OK;
}

int o3d_icp_profile(o3d_ctx *ctx, int enable) {
  if (!ctx) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  ctx->prof_on = enable ? 1 : 0;
  ctx->prof_n = 0;
  return O3D_OK;
}

int o3d_icp_profile_read2(o3d_ctx *ctx, double out_mean_ms[2], int *out_count);
int o3d_icp_profile_read(o3d_ctx *ctx, double *out_mean_ms, int *out_count) {
  double ms2[2];
  const int st = o3d_icp_profile_read2(ctx, ms2, out_count);
  if (st == O3D_OK && out_mean_ms) *out_mean_ms = ms2[0] + ms2[1];
  return st;
}

int o3d_icp_profile_read2(o3d_ctx *ctx, double out_mean_ms[2], int *out_count) {
  if (!ctx) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  DeviceGuard guard(ctx->device);
  O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
  double tot[2] = {0, 0};
  for (int i = 0; i < ctx->prof_n; i++) {
    float ms = 0;
    O3D_HIP(ctx, hipEventElapsedTime(&ms, ctx->prof_ev[3 * i], ctx->prof_ev[3 * i + 1]));
    tot[0] += ms;
    O3D_HIP(ctx, hipEventElapsedTime(&ms, ctx->prof_ev[3 * i + 1], ctx->prof_ev[3 * i + 2]));
    tot[1] += ms;
  }
  if (out_mean_ms) {
    out_mean_ms[0] = ctx->prof_n ? tot[0] / ctx->prof_n : 0.0;
    out_mean_ms[1] = ctx->prof_n ? tot[1] / ctx->prof_n : 0.0;
  }
  if (out_count) *out_count = ctx->prof_n;
  ctx->prof_n = 0;
  return O3D_OK;
}

int o3d_icp_profile_read_all(o3d_ctx *ctx, double *out_ms, int max_count, int *out_count) {
  if (!ctx || !out_ms || max_count < 0) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  DeviceGuard guard(ctx->device);
  O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const int n = ctx->prof_n < max_count ? ctx->prof_n : max_count;
  for (int i = 0; i < n; i++) {
    float ms = 0;
    O3D_HIP(ctx, hipEventElapsedTime(&ms, ctx->prof_ev[3 * i], ctx->prof_ev[3 * i + 2]));
    out_ms[i] = ms;
  }
  if (out_count) *out_count = n;
  return O3D_OK;
}

double *o3d_icp_local_sums(o3d_ctx *ctx) { return ctx ? ctx->sums : nullptr; }

int o3d_icp_solve(o3d_ctx *ctx, const double *dev_all_sums, int n_ranks) {
  if (!ctx || !dev_all_sums || n_ranks < 1) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  if (!ctx->icp_ready) return o3d_fail(ctx, O3D_ERR_STATE, "o3d_icp_solve before o3d_icp_setup");
  DeviceGuard guard(ctx->device);
  return o3d_launch_solve(ctx, dev_all_sums, n_ranks);
}

int o3d_icp_poll(o3d_ctx *ctx, int *out_done, int *out_passes) {
  if (!ctx) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  if (!ctx->icp_ready) return o3d_fail(ctx, O3D_ERR_STATE, "o3d_icp_poll before o3d_icp_setup");
  DeviceGuard guard(ctx->device);
  IcpState *hs = reinterpret_cast<IcpState *>((char *)ctx->pinned + 1024);
  O3D_HIP(ctx, o3d_d2h(ctx, hs, ctx->state, sizeof(IcpState)));
  O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (out_done) *out_done = hs->done;
  if (out_passes) *out_passes = hs->passes;
  return O3D_OK;
}

// o3d_icp_finish in two halves, so that a gang (o3d_registration_icp_batch) waits for the GPU once, not once per slot:
// the copies are enqueued on the context's stream; after that stream has been synchronised the results are in the pinned page.
static int icp_finish_enqueue(o3d_ctx *ctx, int mem, int32_t *out_corr) {
  if (out_corr && ctx->n_src > 0) {
    if (mem == O3D_MEM_DEVICE) {
      O3D_TRY(o3d_launch_scatter_corr(ctx, out_corr));
    } else {
      O3D_TRY(o3d_launch_scatter_corr(ctx, ctx->corr_out));
      O3D_HIP(ctx, o3d_d2h(ctx, out_corr, ctx->corr_out, sizeof(int32_t) * (size_t)ctx->n_src));
    }
  }
  IcpState *hs = reinterpret_cast<IcpState *>((char *)ctx->pinned + 1024);
  O3D_HIP(ctx, o3d_d2h(ctx, hs, ctx->state, sizeof(IcpState)));
  O3D_HIP(ctx, enqueue_violation_read(ctx));
  return O3D_OK;
}
static int icp_finish_complete(o3d_ctx *ctx, double out_T[16], int64_t *out_n_corr, double *out_fitness, double *out_rmse,
                               int *out_iters) {
  const IcpState *hs = reinterpret_cast<const IcpState *>((const char *)ctx->pinned + 1024);
  O3D_TRY(check_bound_violations(ctx));
  if (out_T) memcpy(out_T, hs->T, sizeof(double) * 16);
  if (out_n_corr) *out_n_corr = (int64_t)hs->n_corr;
  if (out_fitness) *out_fitness = hs->fitness;
  if (out_rmse) *out_rmse = hs->rmse;
  if (out_iters) *out_iters = hs->iters;
  return O3D_OK;
}

int o3d_icp_finish(o3d_ctx *ctx, int mem, double out_T[16], int32_t *out_corr, int64_t *out_n_corr, double *out_fitness,
                   double *out_rmse, int *out_iters) {
  if (!ctx) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  if (!ctx->icp_ready) return o3d_fail(ctx, O3D_ERR_STATE, "o3d_icp_finish before o3d_icp_setup");
  DeviceGuard guard(ctx->device);
  if (out_corr && ctx->n_src > 0) {
    if (mem == O3D_MEM_DEVICE) {
      O3D_TRY(o3d_launch_scatter_corr(ctx, out_corr));
    } else {
      O3D_TRY(o3d_launch_scatter_corr(ctx, ctx->corr_out));
      O3D_HIP(ctx, o3d_d2h(ctx, out_corr, ctx->corr_out, sizeof(int32_t) * (size_t)ctx->n_src));
    }
  }
  IcpState *hs = reinterpret_cast<IcpState *>((char *)ctx->pinned + 1024);
  O3D_HIP(ctx, o3d_d2h(ctx, hs, ctx->state, sizeof(IcpState)));
  O3D_HIP(ctx, enqueue_violation_read(ctx));
  O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
  O3D_TRY(check_bound_violations(ctx));
  if (out_T) memcpy(out_T, hs->T, sizeof(double) * 16);
  if (out_n_corr) *out_n_corr = (int64_t)hs->n_corr;
  if (out_fitness) *out_fitness = hs->fitness;
  if (out_rmse) *out_rmse = hs->rmse;
  if (out_iters) *out_iters = hs->iters;
  return O3D_OK;
}

int o3d_icp_get_log(o3d_ctx *ctx, double *host_log, int max_rows, int *out_rows) {
  if (!ctx || !host_log) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  if (!ctx->icp_ready) return o3d_fail(ctx, O3D_ERR_STATE, "o3d_icp_get_log before o3d_icp_setup");
  DeviceGuard guard(ctx->device);
  int done = 0, passes = 0;
  O3D_TRY(o3d_icp_poll(ctx, &done, &passes));
  int rows = passes < max_rows ? passes : max_rows;
  if (rows > ctx->log_rows) rows = ctx->log_rows;
  if (rows > 0) {
    O3D_HIP(ctx, o3d_d2h(ctx, host_log, ctx->log, sizeof(double) * O3D_LOG_STRIDE * (size_t)rows));
    O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  if (out_rows) *out_rows = rows;
  return O3D_OK;
}

int o3d_icp_get_setup_ms(o3d_ctx *ctx, double out_ms[2]) {
  if (!ctx || !out_ms) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  out_ms[0] = out_ms[1] = 0.0;
  if (ctx->setup_ev[0] && ctx->setup_ev[1] && ctx->setup_ev[2]) {
    DeviceGuard guard(ctx->device);
    O3D_HIP(ctx, hipEventSynchronize(ctx->setup_ev[2]));
    float ms0 = 0, ms1 = 0;
    (void)hipEventElapsedTime(&ms0, ctx->setup_ev[0], ctx->setup_ev[1]);
    (void)hipEventElapsedTime(&ms1, ctx->setup_ev[1], ctx->setup_ev[2]);
    out_ms[0] = ms0;
    out_ms[1] = ms1;
  }
  return O3D_OK;
}

// Registration.cpp:154-166.  The whole loop is enqueued without host round trips; kernels past the break are no-ops
// (IcpState::done).  The host looks at the flag every kChunk iterations to stop enqueueing early.
static int run_icp_loop(o3d_ctx *ctx, int max_iter) {
  const int kChunk = 6;
  DeviceGuard guard(ctx->device);  // the event and the copy below belong to the context's device, whatever the caller's is
  O3D_TRY(o3d_icp_step(ctx));
  // IcpState::done is set by the convergence test (:162-165: |d fitness| < relative_fitness AND |d rmse| < relative_rmse)
  // or at max_iteration: with a threshold that no absolute value is below, there is nothing to look for before the end.
  const bool can_break = ctx->rel_fitness > 0.0 && ctx->rel_rmse > 0.0;
  if (can_break && !ctx->poll_ev) O3D_HIP(ctx, hipEventCreateWithFlags(&ctx->poll_ev, hipEventDisableTiming));
  // The flag of a chunk is copied behind it and read only after the NEXT chunk has been enqueued: the device never waits
  // for the host (a blocking look after every chunk drained the stream five times in a 30-iteration call).  Iterations
  // enqueued past the break are no-ops on the device.
  volatile int *flag = reinterpret_cast<volatile int *>((char *)ctx->pinned + 3072);
  bool pending = false;
  int it = 0;
  while (it < max_iter) {
    const int upto = it + kChunk < max_iter ? it + kChunk : max_iter;
    for (; it < upto; it++) O3D_TRY(o3d_icp_step(ctx));
    if (!can_break || it >= max_iter) continue;
    if (pending) {
      O3D_HIP(ctx, hipEventSynchronize(ctx->poll_ev));
      if (*flag) break;
    }
    O3D_HIP(ctx, hipMemcpyAsync((void *)flag, &ctx->state->done, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    O3D_HIP(ctx, hipEventRecord(ctx->poll_ev, ctx->stream));
    pending = true;
  }
  return O3D_OK;
}

// RegistrationICP over the G devices of a multi context: contiguous shards of the source, the target grid built
// on every device, one exchange of 32 sums per rank and iteration through a pinned host mailbox that every device maps.
// One host thread per rank enqueues, per iteration,
//     pass (k_icp_cert, k_icp_search_coop) -> k_icp_finalize into mailbox[parity][rank] -> event
//     wait for the other ranks' events -> k_icp_solve over mailbox[parity][0..G) in rank order
// so the device never waits for the host and every rank computes bit-identical transformations from the same table; the
// threads meet at a host barrier between "record" and "wait" (an event must have been recorded before another stream is
// told to wait for it).  Two mailbox parities suffice: a rank rewrites a slot two iterations later, behind its own solve
// of the iteration in between, which has waited for every rank's sums of that iteration -- each enqueued behind that
// rank's read of the slot.
namespace {
// The rank threads meet once per iteration, a few microseconds apart: spin (a futex wake-up costs more than the iteration
// of a small shard), yield when the wait drags on.
struct HostBarrier {
  const int n;
  std::atomic<int> waiting{0};
  std::atomic<unsigned> gen{0};
  std::atomic<int> votes[2] = {{0}, {0}};  // by generation parity: the OR of what the arriving threads bring
  int result[2] = {0, 0};
  explicit HostBarrier(int n_) : n(n_) {}
  // Returns the OR of every thread's `vote` for THIS meeting.  A decision that all threads must take alike (stop, give up)
  // has to travel this way: a flag set by a fast thread on its way to the next meeting could be seen by a slow thread still
  // leaving this one, which would then leave the loop one meeting early and strand the others (seen as a rare hang of the
  // multi-context tests in a long process).  A thread can be at most one meeting ahead, so two slots suffice.
  int wait(int vote) {
    const unsigned g = gen.load(std::memory_order_acquire);
    const int slot = (int)(g & 1u);
    if (vote) votes[slot].fetch_or(vote, std::memory_order_acq_rel);
    if (waiting.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
      result[slot] = votes[slot].exchange(0, std::memory_order_acq_rel);
      waiting.store(0, std::memory_order_relaxed);
      const int r = result[slot];
      gen.store(g + 1, std::memory_order_release);
      return r;
    }
    for (unsigned spins = 0; gen.load(std::memory_order_acquire) == g; spins++)
      if (spins > 2000) std::this_thread::yield();
    return result[slot];
  }
};
}  // namespace

static int multi_registration_icp(o3d_ctx_full *m, const double *src_xyz, int64_t n_src, const double *tgt_xyz, const double *tgt_nrm,
                                  int64_t n_tgt, double max_dist, const double init[16], int estimation, int with_scaling,
                                  double rel_fitness, double rel_rmse, int max_iter, int mem, double out_T[16], int32_t *out_corr,
                                  int64_t *out_n_corr, double *out_fitness, double *out_rmse, int *out_iters, int out_mem = -1) {
  const int G = (int)m->ranks.size();
  if (out_mem < 0) out_mem = mem;  // (o3d_cloud_registration_icp: device-resident clouds, results to the host)
  if (n_src < 0 || max_iter < 0) return o3d_fail(m, O3D_ERR_INVALID_ARG, "bad sizes");
  if (!m->comms.empty()) {
    // One GPU per rank: pass -> RCCL all-gather of the 32 sums -> solve over the gathered table, all on the rank's stream.  The
    // collective orders the ranks on the device; the host threads meet only before the loop (did every set-up succeed?) and,
    // with stopping criteria, once per chunk of iterations to look at the loop's flag -- every rank computes bit-identical
    // estimates from the same table, so every rank reads the same flag behind the same pass and takes the same decision.
    for (rcclComm_t cm : m->comms)
      if (!cm) return o3d_fail(m, O3D_ERR_STATE, "the context's communicator was aborted by an earlier failure: create a new context");
    HostBarrier bar(G);
    std::vector<int> status((size_t)G, O3D_OK);
    const bool can_break = rel_fitness > 0.0 && rel_rmse > 0.0;
    const int kChunk = 6;
    auto work = [&](int r) {
      o3d_ctx *c = m->ranks[(size_t)r];
      std::unique_lock<std::recursive_mutex> lock(static_cast<o3d_ctx_full *>(c)->mu);
      DeviceGuard guard(c->device);
      const int64_t lo = n_src * r / G, hi = n_src * (r + 1) / G;
      int st = o3d_icp_setup(c, src_xyz ? src_xyz + 3 * lo : nullptr, hi - lo, tgt_xyz, tgt_nrm, n_tgt, max_dist, init, estimation,
                             with_scaling, rel_fitness, rel_rmse, max_iter, mem, n_src);
      if (st == O3D_OK && can_break && !c->poll_ev && hipEventCreateWithFlags(&c->poll_ev, hipEventDisableTiming) != hipSuccess)
        st = o3d_fail(c, O3D_ERR_HIP, "hipEventCreate");
      bool go = bar.wait(st != O3D_OK ? 1 : 0) == 0;
      volatile int *flag = reinterpret_cast<volatile int *>((char *)c->pinned + 3072);
      bool pending = false, failed_here = false;
      for (int it = 0; go && it <= max_iter; it++) {  // passes 0 .. max_iter (Registration.cpp:154-166)
        if (st == O3D_OK) st = o3d_icp_pass_to(c, m->xsend[(size_t)r]);
        if (st == O3D_OK) {
          const int rc = rccl_api().AllGather(m->xsend[(size_t)r], m->xrecv[(size_t)r], O3D_NSUMS, kRcclDouble, m->comms[(size_t)r], c->stream);
          if (rc != 0) st = o3d_fail(c, O3D_ERR_HIP, "ncclAllGather: %s", rccl_api().GetErrorString(rc));
        }
        if (st == O3D_OK) st = o3d_icp_solve(c, m->xrecv[(size_t)r], G);
        if (it % kChunk == kChunk - 1 && it < max_iter) {
          // a rank that failed has not enqueued its part of a collective: the others must not go on waiting for it
          if (bar.wait(st != O3D_OK ? 1 : 0) != 0) { failed_here = true; break; }
          if (can_break) {
            if (pending) {
              if (hipEventSynchronize(c->poll_ev) != hipSuccess) st = o3d_fail(c, O3D_ERR_HIP, "hipEventSynchronize");
              else if (*flag) break;  // (the same value on every rank)
            }
            if (st == O3D_OK && (hipMemcpyAsync((void *)flag, &c->state->done, sizeof(int), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                                 hipEventRecord(c->poll_ev, c->stream) != hipSuccess))
              st = o3d_fail(c, O3D_ERR_HIP, "flag copy");
            pending = true;
          }
        }
      }
      const bool all_ok = bar.wait(st != O3D_OK ? 1 : 0) == 0 && go && !failed_here;
      if (!all_ok) (void)rccl_api().CommAbort(m->comms[(size_t)r]);  // frees kernels waiting for a rank that never came
      if (st == O3D_OK && all_ok)
        st = o3d_icp_finish(c, out_mem, r == 0 ? out_T : nullptr, out_corr ? out_corr + lo : nullptr, r == 0 ? out_n_corr : nullptr,
                            r == 0 ? out_fitness : nullptr, r == 0 ? out_rmse : nullptr, r == 0 ? out_iters : nullptr);
      else if (c->stream)
        (void)hipStreamSynchronize(c->stream);
      if (!all_ok) m->comms[(size_t)r] = nullptr;  // (aborted: the context exchanges nothing any more)
      status[(size_t)r] = st != O3D_OK ? st : (all_ok ? O3D_OK : O3D_ERR_STATE);
    };
    std::vector<std::thread> threads;
    for (int r = 1; r < G; r++) threads.emplace_back(work, r);
    work(0);
    for (auto &t : threads) t.join();
    for (int r = 0; r < G; r++)
      if (status[(size_t)r] != O3D_OK) {
        if (r == 0 && status[0] != O3D_ERR_STATE) return status[0];
        return o3d_fail(m, status[(size_t)r], "rank %d (device %d): %s", r, m->ranks[(size_t)r]->device, o3d_last_error(m->ranks[(size_t)r]));
      }
    return O3D_OK;
  }
  // Ranks on the SAME device (a rehearsal of the sharded path on one GPU) enqueue on one stream, that of the first of
  // them: behind the host barrier every rank's sums are then ahead of every rank's solve in stream order, and no event is
  // needed between them.  (Cross-stream events between many streams of one device -- they share its four hardware queues
  // -- were seen to stall once in a long test process; between devices every stream has a queue of its own.)
  std::vector<int> leader((size_t)G, 0);
  struct RestoreStreams {  // whatever way this function is left, every rank goes back to the stream of its own
    o3d_ctx_full *m;
    ~RestoreStreams() {
      for (o3d_ctx *c : m->ranks) c->stream = c->own_stream_or_given;
    }
  } restore{m};
  for (o3d_ctx *c : m->ranks) c->own_stream_or_given = c->stream;
  for (int r = 0; r < G; r++) {
    leader[(size_t)r] = r;
    for (int f = 0; f < r; f++)
      if (m->ranks[(size_t)f]->device == m->ranks[(size_t)r]->device) { leader[(size_t)r] = f; break; }
    if (leader[(size_t)r] != r) {
      o3d_ctx *c = m->ranks[(size_t)r];
      DeviceGuard guard(c->device);
      O3D_HIP(m, hipStreamSynchronize(c->stream));
      c->stream = m->ranks[(size_t)leader[(size_t)r]]->stream;
    }
  }
  HostBarrier bar(G);
  std::vector<int> status((size_t)G, O3D_OK);
  const bool can_break = rel_fitness > 0.0 && rel_rmse > 0.0;
  const int kChunk = 6;
  auto work = [&](int r) {
    o3d_ctx *c = m->ranks[(size_t)r];
    std::unique_lock<std::recursive_mutex> lock(static_cast<o3d_ctx_full *>(c)->mu);
    DeviceGuard guard(c->device);
    const int64_t lo = n_src * r / G, hi = n_src * (r + 1) / G;
    int st = o3d_icp_setup(c, src_xyz ? src_xyz + 3 * lo : nullptr, hi - lo, tgt_xyz, tgt_nrm, n_tgt, max_dist, init, estimation,
                           with_scaling, rel_fitness, rel_rmse, max_iter, mem, n_src);
    bool any_failed = false, want_stop = false;
    auto sync_point = [&]() {  // every rank arrives, whatever happened to it; afterwards all agree on whether to go on
      const int v = bar.wait((st != O3D_OK ? 1 : 0) | (want_stop ? 2 : 0));
      if (v & 1) any_failed = true;
      return v == 0;
    };
    double *box = m->mailbox_dev[(size_t)r];
    volatile int *flag = reinterpret_cast<volatile int *>((char *)c->pinned + 3072);
    bool pending = false;
    if (sync_point()) {
      if (can_break && !c->poll_ev && hipEventCreateWithFlags(&c->poll_ev, hipEventDisableTiming) != hipSuccess)
        st = o3d_fail(c, O3D_ERR_HIP, "hipEventCreate");
      for (int it = 0; it <= max_iter; it++) {  // passes 0 .. max_iter (Registration.cpp:154-166)
        const int par = it & 1;
        if (st == O3D_OK) st = o3d_icp_pass_to(c, box + ((size_t)par * G + r) * O3D_NSUMS);
        if (st == O3D_OK && hipEventRecord(m->xev[(size_t)(par * G + r)], c->stream) != hipSuccess) st = o3d_fail(c, O3D_ERR_HIP, "hipEventRecord");
        // rank 0 looks at the flag a chunk of iterations after it was copied (the device never waits for the host)
        if (r == 0 && st == O3D_OK && pending && it % kChunk == 0) {
          if (hipEventSynchronize(c->poll_ev) != hipSuccess) st = o3d_fail(c, O3D_ERR_HIP, "hipEventSynchronize");
          else if (*flag) want_stop = true;
          pending = false;
        }
        if (!sync_point()) break;
        for (int q = 0; q < G && st == O3D_OK; q++)  // the sums of the ranks on other devices (same device: same stream)
          if (m->ranks[(size_t)q]->device != c->device && hipStreamWaitEvent(c->stream, m->xev[(size_t)(par * G + q)], 0) != hipSuccess)
            st = o3d_fail(c, O3D_ERR_HIP, "hipStreamWaitEvent");
        if (st == O3D_OK) st = o3d_icp_solve(c, box + (size_t)par * G * O3D_NSUMS, G);
        if (r == 0 && st == O3D_OK && can_break && it % kChunk == kChunk - 1 && it < max_iter) {
          if (hipMemcpyAsync((void *)flag, &c->state->done, sizeof(int), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
              hipEventRecord(c->poll_ev, c->stream) != hipSuccess)
            st = o3d_fail(c, O3D_ERR_HIP, "flag copy");
          pending = true;
        }
      }
    }
    // results: the transformation / fitness / rmse are the same on every rank; the correspondences are the rank's shard
    if (st == O3D_OK && !any_failed)
      st = o3d_icp_finish(c, out_mem, r == 0 ? out_T : nullptr, out_corr ? out_corr + lo : nullptr, r == 0 ? out_n_corr : nullptr,
                          r == 0 ? out_fitness : nullptr, r == 0 ? out_rmse : nullptr, r == 0 ? out_iters : nullptr);
    else if (c->stream)
      (void)hipStreamSynchronize(c->stream);
    status[(size_t)r] = st;
  };
  std::vector<std::thread> threads;
  for (int r = 1; r < G; r++) threads.emplace_back(work, r);
  work(0);
  for (auto &t : threads) t.join();
  for (int r = 0; r < G; r++)
    if (status[(size_t)r] != O3D_OK) {
      if (r == 0) return status[0];  // its message is already on this context
      return o3d_fail(m, status[(size_t)r], "rank %d (device %d): %s", r, m->ranks[(size_t)r]->device, o3d_last_error(m->ranks[(size_t)r]));
    }
  return O3D_OK;
}

int o3d_registration_icp(o3d_ctx *ctx, const double *src_xyz, int64_t n_src, const double *tgt_xyz, const double *tgt_nrm,
                         int64_t n_tgt, double max_dist, const double init[16], int estimation, int with_scaling,
                         double rel_fitness, double rel_rmse, int max_iter, int mem, double out_T[16], int32_t *out_corr,
                         int64_t *out_n_corr, double *out_fitness, double *out_rmse, int *out_iters) {
  if (!ctx) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);  // held for the whole call
  if (static_cast<o3d_ctx_full *>(ctx)->ranks.size() > 1)
    return multi_registration_icp(static_cast<o3d_ctx_full *>(ctx), src_xyz, n_src, tgt_xyz, tgt_nrm, n_tgt, max_dist, init, estimation,
                                  with_scaling, rel_fitness, rel_rmse, max_iter, mem, out_T, out_corr, out_n_corr, out_fitness,
                                  out_rmse, out_iters);
  DeviceGuard guard(ctx->device);
  O3D_TRY(o3d_icp_setup(ctx, src_xyz, n_src, tgt_xyz, tgt_nrm, n_tgt, max_dist, init, estimation, with_scaling, rel_fitness,
                        rel_rmse, max_iter, mem, n_src));
  O3D_TRY(run_icp_loop(ctx, max_iter));
  return o3d_icp_finish(ctx, mem, out_T, out_corr, out_n_corr, out_fitness, out_rmse, out_iters);
}

int o3d_registration_colored_icp(o3d_ctx *ctx, const double *src_xyz, const double *src_col, int64_t n_src, const double *tgt_xyz,
                                 const double *tgt_nrm, const double *tgt_col, int64_t n_tgt, double max_dist, const double init[16],
                                 double lambda_geometric, double rel_fitness, double rel_rmse, int max_iter, int mem, double out_T[16],
                                 int32_t *out_corr, int64_t *out_n_corr, double *out_fitness, double *out_rmse, int *out_iters,
                                 double *out_gradients) {
  if (!ctx) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);  // held for the whole call
  DeviceGuard guard(ctx->device);
  if (!src_col || !tgt_col) {
    // ColoredICP.cpp:127-128: without colours every update is the identity; the loop of Registration.cpp:154-166 then
    // stops after its first iteration with the result of the initial search (and zero gradients where they exist)
    if (!(max_dist > 0.0)) return o3d_fail(ctx, O3D_ERR_INVALID_DISTANCE, "%s", o3d_status_string(O3D_ERR_INVALID_DISTANCE));
    if (!tgt_nrm) return o3d_fail(ctx, O3D_ERR_NEED_NORMALS, "%s", o3d_status_string(O3D_ERR_NEED_NORMALS));
    if (!init) return O3D_ERR_INVALID_ARG;
    if (out_gradients && n_tgt > 0) {
      DeviceGuard guard(ctx->device);
      if (mem == O3D_MEM_DEVICE) {
        O3D_HIP(ctx, hipMemsetAsync(out_gradients, 0, sizeof(double) * 3 * (size_t)n_tgt, ctx->stream));
        O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
      } else {
        memset(out_gradients, 0, sizeof(double) * 3 * (size_t)n_tgt);
      }
    }
    O3D_TRY(o3d_evaluate_registration(ctx, src_xyz, n_src, tgt_xyz, n_tgt, max_dist, init, mem, out_corr, out_n_corr, out_fitness,
                                      out_rmse));
    if (out_T) memcpy(out_T, init, sizeof(double) * 16);
    if (out_iters) *out_iters = max_iter > 0 ? 1 : 0;
    return O3D_OK;
  }
  O3D_TRY(icp_setup_impl(ctx, src_xyz, n_src, tgt_xyz, tgt_nrm, n_tgt, max_dist, init, O3D_EST_COLORED_ICP, 0, rel_fitness, rel_rmse,
                         max_iter, mem, n_src, src_col, tgt_col, lambda_geometric, out_gradients));
  O3D_TRY(run_icp_loop(ctx, max_iter));
  return o3d_icp_finish(ctx, mem, out_T, out_corr, out_n_corr, out_fitness, out_rmse, out_iters);
}

int o3d_registration_icp_batch(o3d_ctx *c, int64_t n_pairs, const double *const *src_xyz, const int64_t *n_src,
                               const double *const *tgt_xyz, const double *const *tgt_nrm, const int64_t *n_tgt, double max_dist,
                               const double *init, int estimation, int with_scaling, double rel_fitness, double rel_rmse,
                               int max_iter, int mem, int n_workers, double *out_T, int32_t *const *out_corr, int64_t *out_n_corr,
                               double *out_fitness, double *out_rmse, int *out_iters, int *out_status) {
  if (!c) return O3D_ERR_INVALID_ARG;
  o3d_ctx_full *ctx = static_cast<o3d_ctx_full *>(c);
  O3D_LOCK(c);
  if (n_pairs < 0 || (n_pairs > 0 && (!src_xyz || !n_src || !tgt_xyz || !n_tgt)))
    return o3d_fail(ctx, O3D_ERR_INVALID_ARG, "bad sizes or null buffers");
  if (n_pairs == 0) return O3D_OK;
  if (n_workers <= 0) n_workers = 7;  // (config 5: 4450 / 4410 / 4330 pairs/s with 7 / 8 / 10)
  if (n_workers > 16) n_workers = 16;
  if ((int64_t)n_workers > n_pairs) n_workers = (int)n_pairs;
  // Each worker thread advances a gang of registrations in lock step (icp.hip, o3d_launch_pass_gang): one launch per pass
  // kernel for the whole gang.  O3D_BATCH_GANG=1 gives the registrations one by one (the round-1 form).
  // As many registrations per gang as put about six million source points into a launch (a pair of 200 k points: 30;
  // a pair that fills the GPU by itself: 1 -- and every slot holds its pair's buffers, so large pairs must not multiply).
  int64_t largest = 1;
  for (int64_t p = 0; p < n_pairs; p++) largest = std::max<int64_t>(largest, std::max<int64_t>(n_src[p], n_tgt[p]));
  int gang = (int)std::min<int64_t>(O3D_GANG_MAX, std::max<int64_t>(1, 6000000 / largest));
  const bool use_gang_sort = !getenv("O3D_NO_GANG_SORT") && estimation != O3D_EST_COLORED_ICP;  // tuning aid
  if (const char *e = getenv("O3D_BATCH_GANG")) gang = atoi(e);
  if (gang < 1) gang = 1;
  if (gang > O3D_GANG_MAX) gang = O3D_GANG_MAX;
  if ((int64_t)gang * n_workers > n_pairs) gang = (int)((n_pairs + n_workers - 1) / n_workers);
  while ((int)ctx->workers.size() < n_workers * gang) {
    o3d_ctx *w = nullptr;
    const int st = o3d_ctx_create(ctx->device, &w);
    if (st != O3D_OK) return o3d_fail(ctx, st, "batch worker context: %s", o3d_last_error(nullptr));
    ctx->workers.push_back(w);
  }
  static const double kIdentity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  std::atomic<int64_t> next(0);
  std::vector<int> first_status((size_t)n_workers, O3D_OK);
  std::vector<std::string> first_error((size_t)n_workers);
  auto work = [&](int wi) {
    o3d_ctx *const *slot = &ctx->workers[(size_t)wi * (size_t)gang];
    o3d_ctx *lead = slot[0];
    DeviceGuard guard(lead->device);
    for (int k = 0; k < gang; k++) {
      slot[k]->use_tile = ctx->use_tile;
      slot[k]->gang_slots = gang;
    }
    auto fail = [&](int64_t p, int st, o3d_ctx *c) {
      if (out_status) out_status[p] = st;
      if (first_status[(size_t)wi] == O3D_OK) { first_status[(size_t)wi] = st; first_error[(size_t)wi] = o3d_last_error(c); }
    };
    // O3D_BATCH_TIMING=1: where this thread's wall time went (set-up calls / waiting for them and the common sort / passes
    // enqueued / finish calls), printed once per thread
    static const bool timing = getenv("O3D_BATCH_TIMING") != nullptr;
    double t_phase[4] = {0, 0, 0, 0};
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    struct Report {
      const double *t; int wi; bool on;
      ~Report() { if (on) fprintf(stderr, "[batch worker %d] set-up calls %.1f ms, sync + common sort %.1f ms, passes enqueued %.1f ms, finish %.1f ms\n", wi, 1e3 * t[0], 1e3 * t[1], 1e3 * t[2], 1e3 * t[3]); }
    } report{t_phase, wi, timing};
    for (;;) {
      const int64_t base = next.fetch_add(gang);
      if (base >= n_pairs) break;
      double t0 = timing ? now() : 0.0;
      auto lap = [&](int k) { if (timing) { const double t1 = now(); t_phase[k] += t1 - t0; t0 = t1; } };
      const int m = (int)std::min<int64_t>(gang, n_pairs - base);
      o3d_ctx *live[O3D_GANG_MAX];
      int64_t live_pair[O3D_GANG_MAX];
      int nl = 0;
      // the gang's common sorter (o3d_gang_sort): room for every key of these pairs
      o3d_gang_sort *gs = nullptr;
      if (m > 1 && use_gang_sort) {
        if (!lead->gang_sort_own) lead->gang_sort_own = new o3d_gang_sort();
        gs = lead->gang_sort_own;
        int64_t need = 0;
        for (int k = 0; k < m; k++) need += std::max<int64_t>(n_src[base + k], 0) + std::max<int64_t>(n_tgt[base + k], 0);
        const size_t nh = 256 * o3d_sort_num_tiles(need > 0 ? need : 1);
        const size_t hneed = nh + o3d_scan_num_blocks((int64_t)nh) + 264;
        bool ok = need < ((int64_t)1 << 31);
        if (ok && ((size_t)need > gs->cap || hneed > gs->hist_cap)) {
          (void)hipStreamSynchronize(lead->own_stream);
          for (uint32_t **b : {&gs->keys, &gs->vals, &gs->tkeys, &gs->tvals, &gs->hist}) {
            if (*b) (void)hipFree(*b);
            *b = nullptr;
          }
          gs->cap = gs->hist_cap = 0;
          const size_t cap = (size_t)need + (size_t)need / 4 + 1024, hcap = hneed + hneed / 4;
          ok = hipMalloc((void **)&gs->keys, 4 * cap) == hipSuccess && hipMalloc((void **)&gs->vals, 4 * cap) == hipSuccess &&
               hipMalloc((void **)&gs->tkeys, 4 * cap) == hipSuccess && hipMalloc((void **)&gs->tvals, 4 * cap) == hipSuccess &&
               hipMalloc((void **)&gs->hist, 4 * hcap) == hipSuccess;
          if (ok) { gs->cap = cap; gs->hist_cap = hcap; }
        }
        if (!ok) gs = nullptr;  // (the slots then sort for themselves)
        if (gs) { gs->used = 0; gs->jobs.clear(); }
      }
      // The set-ups of a gang share the gang's stream: with the sorts gone to the common sort a set-up is four small kernels,
      // and a stream of its own for every slot (31 per thread) meant 31 places to queue up behind other threads' pass kernels
      // in the four hardware queues (measured on one box, 7 threads: 4035 pairs/s with a stream per slot, 4410 with one).
      // O3D_BATCH_SLOT_STREAMS=1 brings the streams per slot back (tuning aid).
      static const bool one_stream = getenv("O3D_BATCH_SLOT_STREAMS") == nullptr;
      if (mem == O3D_MEM_DEVICE && m > 1)  // every slot's bounding box is on its way before the first set-up waits for one
        for (int k = 0; k < m; k++) {
          slot[k]->stream = one_stream ? lead->own_stream : slot[k]->own_stream;
          (void)o3d_bounds_enqueue(slot[k], tgt_xyz[base + k], n_tgt[base + k]);
        }
      for (int k = 0; k < m; k++) {
        const int64_t p = base + k;
        slot[k]->stream = one_stream ? lead->own_stream : slot[k]->own_stream;
        slot[k]->gang_sort = gs;
        const int st = o3d_icp_setup(slot[k], src_xyz[p], n_src[p], tgt_xyz[p], tgt_nrm ? tgt_nrm[p] : nullptr, n_tgt[p], max_dist,
                                     init ? init + 16 * p : kIdentity, estimation, with_scaling, rel_fitness, rel_rmse, max_iter, mem,
                                     n_src[p]);
        if (st != O3D_OK) { fail(p, st, slot[k]); continue; }  // the other pairs are independent
        live[nl] = slot[k]; live_pair[nl] = p; nl++;
      }
      lap(0);
      for (int k = 0; k < m; k++) {
        slot[k]->gang_sort = nullptr;
        slot[k]->bounds_for = nullptr;  // (a set-up that failed early has not picked its box up)
      }
      // (every slot, the failed ones too: their keys lie in the common buffer and must be at rest when the sort counts them)
      if (!one_stream)  // (one stream: the sort is simply next in line)
        for (int k = 0; k < m; k++) (void)hipStreamSynchronize(slot[k]->own_stream);
      for (int k = 0; k < nl; k++) live[k]->stream = lead->own_stream;
      if (gs && !gs->jobs.empty()) {
        // every key of the gang in one sort (32 bits: job number on top), then what each slot queued behind its sort
        int st = O3D_OK;
        lead->stream = lead->own_stream;
        if (gs->used > 0)
          st = o3d_radix_sort_u32(lead, gs->keys, gs->vals, gs->tkeys, gs->tvals, gs->hist, gs->used, 32, false);
        // (no copy home: an even number of passes, four for 32 bits, leaves the result where the keys were written)
        const bool other = o3d_radix_sort_odd(32);
        const uint32_t *rk = other ? gs->tkeys : gs->keys, *rv = other ? gs->tvals : gs->vals;
        for (auto &job : gs->jobs) {
          bool alive = false;
          for (int k = 0; k < nl; k++) alive = alive || live[k] == job.owner;
          if (!alive || st != O3D_OK) continue;
          const int js = job.then(rk + job.off, rv + job.off);
          if (js != O3D_OK) st = js;
        }
        gs->jobs.clear();
        if (st != O3D_OK) {  // nothing of this gang can be trusted
          (void)hipStreamSynchronize(lead->own_stream);
          for (int k = 0; k < nl; k++) fail(live_pair[k], st, lead);
          nl = 0;
        }
      }
      lap(1);
      bool together = nl > 1;
      for (int k = 0; k < nl && together; k++)
        together = o3d_gang_eligible(live[k]) && live[k]->use_cert == live[0]->use_cert && live[k]->tn_valid == live[0]->tn_valid;
      int loop_status = O3D_OK;
      if (together) {
        for (int pass = 0; pass <= max_iter && loop_status == O3D_OK; pass++) {
          for (int k = 0; k < nl; k++) { live[k]->enq_passes++; live[k]->sums_valid = 1; }
          loop_status = o3d_launch_pass_gang(lead, live, nl, pass == 0, pass == 0);
        }
      }
      lap(2);
      // the slots' results: with the gang in lock step every copy is enqueued first (all on the gang's stream), then ONE wait
      int enq_status[O3D_GANG_MAX];
      if (together && loop_status == O3D_OK) {
        for (int k = 0; k < nl; k++) {
          const int64_t p = live_pair[k];
          enq_status[k] = icp_finish_enqueue(live[k], mem, (out_corr && out_corr[p]) ? out_corr[p] : nullptr);
        }
        if (hipStreamSynchronize(lead->own_stream) != hipSuccess) loop_status = o3d_fail(lead, O3D_ERR_HIP, "hipStreamSynchronize");
      }
      for (int k = 0; k < nl; k++) {
        const int64_t p = live_pair[k];
        int st = loop_status;
        if (together && st != O3D_OK) { fail(p, st, lead); continue; }
        if (!together) st = run_icp_loop(live[k], max_iter);
        double T[16], fit = 0.0, rmse = 0.0;
        int64_t nc = 0;
        int iters = 0;
        if (together) {
          st = enq_status[k];
          if (st == O3D_OK) st = icp_finish_complete(live[k], T, &nc, &fit, &rmse, &iters);
        } else if (st == O3D_OK) {
          st = o3d_icp_finish(live[k], mem, T, (out_corr && out_corr[p]) ? out_corr[p] : nullptr, &nc, &fit, &rmse, &iters);
        }
        if (st != O3D_OK) { fail(p, st, live[k]); continue; }
        if (out_status) out_status[p] = O3D_OK;
        if (out_T) memcpy(out_T + 16 * p, T, sizeof T);
        if (out_n_corr) out_n_corr[p] = nc;
        if (out_fitness) out_fitness[p] = fit;
        if (out_rmse) out_rmse[p] = rmse;
        if (out_iters) out_iters[p] = iters;
      }
      lap(3);
    }
    for (int k = 0; k < gang; k++) {
      slot[k]->stream = slot[k]->own_stream;
      slot[k]->gang_slots = 0;
    }
  };
  std::vector<std::thread> threads;
  for (int wi = 1; wi < n_workers; wi++) threads.emplace_back(work, wi);
  work(0);
  for (auto &t : threads) t.join();
  for (int wi = 0; wi < n_workers; wi++)
    if (first_status[(size_t)wi] != O3D_OK)
      return o3d_fail(ctx, first_status[(size_t)wi], "batch: %s", first_error[(size_t)wi].c_str());
  return O3D_OK;
}

int o3d_evaluate_registration(o3d_ctx *ctx, const double *src_xyz, int64_t n_src, const double *tgt_xyz, int64_t n_tgt,
                              double max_dist, const double T[16], int mem, int32_t *out_corr, int64_t *out_n_corr,
                              double *out_fitness, double *out_rmse) {
  if (!ctx) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  if (!(max_dist > 0.0)) {  // Registration.cpp:45-47: an empty result, not an error
    DeviceGuard guard(ctx->device);
    if (out_corr && n_src > 0) {
      if (mem == O3D_MEM_DEVICE) {
        O3D_HIP(ctx, hipMemsetAsync(out_corr, 0xFF, sizeof(int32_t) * (size_t)n_src, ctx->stream));
        O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
      } else {
        memset(out_corr, 0xFF, sizeof(int32_t) * (size_t)n_src);
      }
    }
    if (out_n_corr) *out_n_corr = 0;
    if (out_fitness) *out_fitness = 0.0;
    if (out_rmse) *out_rmse = 0.0;
    return O3D_OK;
  }
  O3D_TRY(o3d_icp_setup(ctx, src_xyz, n_src, tgt_xyz, nullptr, n_tgt, max_dist, T, O3D_EST_POINT_TO_POINT, 0, 0.0, 0.0, 0, mem, n_src));
  O3D_TRY(o3d_icp_step(ctx));
  return o3d_icp_finish(ctx, mem, nullptr, out_corr, out_n_corr, out_fitness, out_rmse, nullptr);
}


// ---- estimations that stay on the host ---------------------------------------------------------------------------

int o3d_icp_set_update(o3d_ctx *ctx, const double update_colmajor[16]) {
  if (!ctx || !update_colmajor) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  if (!ctx->icp_ready) return o3d_fail(ctx, O3D_ERR_STATE, "o3d_icp_set_update before o3d_icp_setup");
  DeviceGuard guard(ctx->device);
  return o3d_launch_set_update(ctx, update_colmajor);
}

int o3d_icp_fetch(o3d_ctx *ctx, int mem, int32_t *out_corr, int64_t *out_n_corr, double *out_fitness, double *out_rmse) {
  if (!ctx) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  if (!ctx->icp_ready || !ctx->sums_valid)
    return o3d_fail(ctx, O3D_ERR_STATE, "o3d_icp_fetch must directly follow o3d_icp_pass (or o3d_icp_step) on this context");
  DeviceGuard guard(ctx->device);
  if (out_corr && ctx->n_src > 0) {
    if (mem == O3D_MEM_DEVICE) {
      O3D_TRY(o3d_launch_scatter_corr(ctx, out_corr));
    } else {
      O3D_TRY(o3d_launch_scatter_corr(ctx, ctx->corr_out));
      O3D_HIP(ctx, o3d_d2h(ctx, out_corr, ctx->corr_out, sizeof(int32_t) * (size_t)ctx->n_src));
    }
  }
  double *hq = reinterpret_cast<double *>(ctx->pinned);
  O3D_HIP(ctx, o3d_d2h(ctx, hq, ctx->sums, sizeof(double) * O3D_NSUMS));
  O3D_HIP(ctx, enqueue_violation_read(ctx));
  O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
  O3D_TRY(check_bound_violations(ctx));
  const double nc = hq[O3D_SUM_NC];
  if (out_n_corr) *out_n_corr = (int64_t)nc;
  if (out_fitness) *out_fitness = nc > 0.0 ? nc / (double)ctx->n_src_global : 0.0;  // Registration.cpp:82-89
  if (out_rmse) *out_rmse = nc > 0.0 ? sqrt(hq[O3D_SUM_D2] / nc) : 0.0;
  return O3D_OK;
}

// ---- TransformationEstimation members over a given CorrespondenceSet -----------------------------------------------

static int pairs_common(o3d_ctx *ctx, const double *src_xyz, int64_t n_src, const double *tgt_xyz, const double *tgt_nrm, int64_t n_tgt,
                        const int32_t *pairs, int64_t n_pairs, int estimation, int with_scaling, int mem, double host_out[19]) {
  if (!ctx) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  if (estimation != O3D_EST_POINT_TO_POINT && estimation != O3D_EST_POINT_TO_PLANE)
    return o3d_fail(ctx, O3D_ERR_UNSUPPORTED, "estimation type %d is not on the device path", estimation);
  if (n_src < 0 || n_tgt < 0 || n_pairs < 0 || (n_pairs > 0 && (!pairs || !src_xyz || !tgt_xyz)))
    return o3d_fail(ctx, O3D_ERR_INVALID_ARG, "bad sizes or null buffers");
  for (int i = 0; i < 19; i++) host_out[i] = 0.0;
  host_out[0] = host_out[5] = host_out[10] = host_out[15] = 1.0;
  if (n_pairs == 0) return O3D_OK;  // TransformationEstimation.cpp:40-41,51-52,64-65,76-77
  DeviceGuard guard(ctx->device);
  o3d_staging *sg = staging_of(ctx);
  // Safe between the passes of a set-up loop on the same context (the host-estimation path calls it from the virtual
  // ComputeTransformation): the staging buffers are free once o3d_icp_setup has returned; only ctx->sums is taken over.
  ctx->sums_valid = 0;
  const void *d_tgt, *d_nrm, *d_src, *d_pairs;
  O3D_TRY(to_device(ctx, tgt_xyz, sizeof(double) * 3 * (size_t)n_tgt, mem, &sg->p[0], &sg->cap[0], &d_tgt));
  O3D_TRY(to_device(ctx, tgt_nrm, sizeof(double) * 3 * (size_t)n_tgt, mem, &sg->p[1], &sg->cap[1], &d_nrm));
  O3D_TRY(to_device(ctx, src_xyz, sizeof(double) * 3 * (size_t)n_src, mem, &sg->p[2], &sg->cap[2], &d_src));
  O3D_TRY(to_device(ctx, pairs, sizeof(int32_t) * 2 * (size_t)n_pairs, mem, &sg->p[3], &sg->cap[3], &d_pairs));
  int *d_bad = reinterpret_cast<int *>(ctx->sums + 24);
  O3D_TRY(o3d_launch_pairs(ctx, (const double *)d_src, n_src, (const double *)d_tgt, (const double *)d_nrm, n_tgt,
                           (const int32_t *)d_pairs, n_pairs, estimation, with_scaling, ctx->sums, d_bad));
  double *hq = reinterpret_cast<double *>(ctx->pinned);
  O3D_HIP(ctx, o3d_d2h(ctx, hq, ctx->sums, sizeof(double) * O3D_NSUMS));
  O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
  int bad;
  memcpy(&bad, hq + 24, sizeof bad);
  if (bad) return o3d_fail(ctx, O3D_ERR_INVALID_ARG, "%d correspondence(s) index outside the clouds", bad);
  for (int i = 0; i < 19; i++) host_out[i] = hq[i];
  return O3D_OK;
}

int o3d_compute_transformation(o3d_ctx *ctx, const double *src_xyz, int64_t n_src, const double *tgt_xyz, const double *tgt_nrm,
                               int64_t n_tgt, const int32_t *corres_pairs, int64_t n_corres, int estimation, int with_scaling,
                               int mem, double out_transformation[16]) {
  if (!out_transformation) return O3D_ERR_INVALID_ARG;
  double r[19];
  O3D_TRY(pairs_common(ctx, src_xyz, n_src, tgt_xyz, tgt_nrm, n_tgt, corres_pairs, n_corres, estimation, with_scaling, mem, r));
  memcpy(out_transformation, r, sizeof(double) * 16);
  return O3D_OK;
}

int o3d_compute_rmse(o3d_ctx *ctx, const double *src_xyz, int64_t n_src, const double *tgt_xyz, const double *tgt_nrm, int64_t n_tgt,
                     const int32_t *corres_pairs, int64_t n_corres, int estimation, int mem, double *out_rmse) {
  if (!out_rmse) return O3D_ERR_INVALID_ARG;
  double r[19];
  O3D_TRY(pairs_common(ctx, src_xyz, n_src, tgt_xyz, tgt_nrm, n_tgt, corres_pairs, n_corres, estimation, 0, mem, r));
  *out_rmse = estimation == O3D_EST_POINT_TO_PLANE ? r[17] : r[16];
  return O3D_OK;
}

// ---- GetInformationMatrixFromPointClouds, Registration.cpp:303-355 ---------------------------------------------------

int o3d_information_matrix(o3d_ctx *ctx, const double *src_xyz, int64_t n_src, const double *tgt_xyz, int64_t n_tgt,
                           double max_dist, const double T[16], int mem, double out_gtg[36]) {
  if (!ctx || !out_gtg) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  for (int i = 0; i < 36; i++) out_gtg[i] = 0.0;
  if (!(max_dist > 0.0)) return O3D_OK;  // Registration.cpp:45-47: no correspondences, GTG stays zero
  O3D_TRY(o3d_icp_setup(ctx, src_xyz, n_src, tgt_xyz, nullptr, n_tgt, max_dist, T, O3D_EST_POINT_TO_POINT, 0, 0.0, 0.0, 0, mem, n_src));
  O3D_TRY(o3d_icp_pass(ctx));
  DeviceGuard guard(ctx->device);
  O3D_TRY(o3d_launch_info(ctx, ctx->sums));
  double *q = reinterpret_cast<double *>(ctx->pinned);
  O3D_HIP(ctx, o3d_d2h(ctx, q, ctx->sums, sizeof(double) * O3D_NSUMS));
  O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->sums_valid = 0;  // ctx->sums no longer holds the pass sums
  const double sx = q[0], sy = q[1], sz = q[2], xx = q[3], yy = q[4], zz = q[5], xy = q[6], xz = q[7], yz = q[8], n = q[9];
#define G(r, c) out_gtg[(r) * 6 + (c)]
  G(0, 0) = zz + yy; G(0, 1) = -xy;     G(0, 2) = -xz;     G(0, 4) = -sz; G(0, 5) = sy;
  G(1, 1) = zz + xx; G(1, 2) = -yz;     G(1, 3) = sz;      G(1, 5) = -sx;
  G(2, 2) = yy + xx; G(2, 3) = -sy;     G(2, 4) = sx;
  G(3, 3) = n; G(4, 4) = n; G(5, 5) = n;
  for (int r = 0; r < 6; r++)
    for (int c = 0; c < r; c++) G(r, c) = G(c, r);
#undef G
  return O3D_OK;
}

int o3d_evaluate_hypotheses(o3d_ctx *ctx, const double *src_xyz, int64_t n_src, const double *tgt_xyz, int64_t n_tgt, double max_dist,
                            const double *Ts, int n_hyp, int mem, double *out_fitness, double *out_rmse, int64_t *out_n_corr,
                            int *out_best, int32_t *out_corr_best) {
  if (!ctx) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  if (n_hyp < 0 || (n_hyp > 0 && !Ts)) return o3d_fail(ctx, O3D_ERR_INVALID_ARG, "bad hypothesis list");
  if (out_best) *out_best = -1;
  if (n_hyp == 0) return O3D_OK;
  static const double kIdentity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  if (!(max_dist > 0.0)) {  // Registration.cpp:45-47: empty results
    for (int k = 0; k < n_hyp; k++) {
      if (out_fitness) out_fitness[k] = 0.0;
      if (out_rmse) out_rmse[k] = 0.0;
      if (out_n_corr) out_n_corr[k] = 0;
    }
    return O3D_OK;
  }
  O3D_TRY(o3d_icp_setup(ctx, src_xyz, n_src, tgt_xyz, nullptr, n_tgt, max_dist, kIdentity, O3D_EST_POINT_TO_POINT, 0, 0.0, 0.0, 0, mem,
                        n_src));
  DeviceGuard guard(ctx->device);
  const size_t sbytes = sizeof(double) * 3 * (size_t)(n_src > 0 ? n_src : 1);
  O3D_TRY(o3d_reserve(ctx, (void **)&ctx->src_orig, &ctx->cap_src_orig, sbytes));
  if (n_src > 0) O3D_HIP(ctx, hipMemcpyAsync(ctx->src_orig, ctx->src_sorted, sizeof(double) * 3 * (size_t)n_src, hipMemcpyDeviceToDevice, ctx->stream));
  void *d_all;
  O3D_TRY(o3d_scratch(ctx, 7, sizeof(double) * O3D_NSUMS * (size_t)n_hyp, &d_all));
  ctx->hyp_mode = 1;
  auto run = [&]() -> int {
    for (int k = 0; k < n_hyp; k++) {  // everything is enqueued; no host round trip per hypothesis
      O3D_TRY(o3d_launch_set_hypothesis(ctx, Ts + 16 * (size_t)k));
      O3D_TRY(o3d_icp_pass_to(ctx, (double *)d_all + (size_t)k * O3D_NSUMS));
    }
    std::vector<double> h((size_t)n_hyp * O3D_NSUMS);
    O3D_HIP(ctx, o3d_d2h(ctx, h.data(), d_all, sizeof(double) * h.size()));
    O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int best = -1;
    double bf = 0.0, br = 0.0;  // RegistrationResult() starts at fitness 0, rmse 0 (Registration.h:112-113)
    for (int k = 0; k < n_hyp; k++) {
      const double nc = h[(size_t)k * O3D_NSUMS + O3D_SUM_NC];
      const double fit = nc > 0.0 ? nc / (double)n_src : 0.0;
      const double rm = nc > 0.0 ? sqrt(h[(size_t)k * O3D_NSUMS + O3D_SUM_D2] / nc) : 0.0;
      if (out_fitness) out_fitness[k] = fit;
      if (out_rmse) out_rmse[k] = rm;
      if (out_n_corr) out_n_corr[k] = (int64_t)nc;
      if (fit > bf || (fit == bf && rm < br)) { bf = fit; br = rm; best = k; }  // :272-275
    }
    if (out_best) *out_best = best;
    if (out_corr_best && n_src > 0) {
      if (best >= 0) {
        O3D_TRY(o3d_launch_set_hypothesis(ctx, Ts + 16 * (size_t)best));
        O3D_TRY(o3d_icp_pass(ctx));
        O3D_TRY(o3d_icp_fetch(ctx, mem, out_corr_best, nullptr, nullptr, nullptr));
      } else if (mem == O3D_MEM_DEVICE) {
        O3D_HIP(ctx, hipMemsetAsync(out_corr_best, 0xFF, sizeof(int32_t) * (size_t)n_src, ctx->stream));
        O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
      } else {
        memset(out_corr_best, 0xFF, sizeof(int32_t) * (size_t)n_src);
      }
    }
    return O3D_OK;
  };
  const int st = run();
  ctx->hyp_mode = 0;
  ctx->icp_ready = 0;  // the in-place source no longer matches the loop state
  return st;
}

int o3d_voxel_down_sample(o3d_ctx *ctx, const double *xyz, const double *nrm, const double *col, int64_t n, double voxel_size,
                          int mem, double *out_xyz, double *out_nrm, double *out_col, int32_t *out_voxel_index, int64_t *out_m) {
  if (!ctx) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  DeviceGuard guard(ctx->device);
  return o3d_voxel_down_sample_impl(ctx, xyz, nrm, col, n, voxel_size, mem, out_xyz, out_nrm, out_col, out_voxel_index, out_m);
}

}  // extern "C"

// EstimateNormals over the devices of a multi context: every rank builds the grid over the whole cloud and computes the
// normals of its contiguous range of cell-ordered points (o3d_estimate_normals_shard, one host thread per rank, no
// exchange), rank 0 scatters the parts to the caller's order (o3d_estimate_normals_assemble).  Bit-identical to the
// one-device call.
static int multi_estimate_normals(o3d_ctx_full *m, const double *xyz, int64_t n, int search_type, int knn, double radius, int mem,
                                  double *normals, int has_normals) {
  const int G = (int)m->ranks.size();
  const int64_t rows_max = (n + G - 1) / G;
  const size_t part_doubles = 3 * (size_t)rows_max;
  std::vector<double> host_parts;
  double *parts = nullptr, *packed = nullptr;
  DeviceGuard guard0(m->device);
  if (mem == O3D_MEM_HOST) {
    host_parts.resize(part_doubles * (size_t)G);
    parts = host_parts.data();
  } else {
    void *p = nullptr;
    O3D_HIP(m, o3d_tmp_alloc(m, &p, sizeof(double) * part_doubles * (size_t)G));
    parts = static_cast<double *>(p);
  }
  std::vector<int> status((size_t)G, O3D_OK);
  std::vector<int64_t> ng((size_t)G, 0);
  auto work = [&](int r) {
    o3d_ctx *c = m->ranks[(size_t)r];
    std::unique_lock<std::recursive_mutex> lock(static_cast<o3d_ctx_full *>(c)->mu);
    DeviceGuard guard(c->device);
    int st = o3d_estimate_normals_shard_impl(c, xyz, n, search_type, knn, radius, mem, has_normals ? normals : nullptr, r, G,
                                             parts + part_doubles * (size_t)r, &ng[(size_t)r]);
    if (st == O3D_OK && hipStreamSynchronize(c->stream) != hipSuccess) st = o3d_fail(c, O3D_ERR_HIP, "hipStreamSynchronize");
    status[(size_t)r] = st;
  };
  std::vector<std::thread> threads;
  for (int r = 1; r < G; r++) threads.emplace_back(work, r);
  work(0);
  for (auto &t : threads) t.join();
  int st = O3D_OK;
  for (int r = 0; r < G && st == O3D_OK; r++)
    if (status[(size_t)r] != O3D_OK)
      st = r == 0 ? status[0] : o3d_fail(m, status[(size_t)r], "rank %d (device %d): %s", r, m->ranks[(size_t)r]->device,
                                         o3d_last_error(m->ranks[(size_t)r]));
  if (st == O3D_OK) {
    // the parts were written at the stride of the largest possible shard; the assembly wants them at ceil(ng / G) rows
    const int64_t rows = (ng[0] + G - 1) / G;
    const size_t row_doubles = 3 * (size_t)rows;
    if (mem == O3D_MEM_HOST) {
      if (rows != rows_max)
        for (int r = 1; r < G; r++) memmove(parts + row_doubles * (size_t)r, parts + part_doubles * (size_t)r, sizeof(double) * row_doubles);
      st = o3d_estimate_normals_assemble_impl(m, xyz, n, parts, G, mem, normals);
    } else {
      void *p = nullptr;
      hipError_t e = o3d_tmp_alloc(m, &p, sizeof(double) * std::max<size_t>(row_doubles * (size_t)G, 1));
      if (e != hipSuccess) st = o3d_fail(m, O3D_ERR_OOM, "EstimateNormals: %s", hipGetErrorString(e));
      packed = static_cast<double *>(p);
      for (int r = 0; r < G && st == O3D_OK && row_doubles > 0; r++)
        if (hipMemcpyAsync(packed + row_doubles * (size_t)r, parts + part_doubles * (size_t)r, sizeof(double) * row_doubles,
                           hipMemcpyDeviceToDevice, m->stream) != hipSuccess)
          st = o3d_fail(m, O3D_ERR_HIP, "EstimateNormals: packing the parts");
      if (st == O3D_OK) st = o3d_estimate_normals_assemble_impl(m, xyz, n, packed, G, mem, normals);
    }
  }
  if (mem != O3D_MEM_HOST) {
    o3d_tmp_free(m, parts);
    if (packed) o3d_tmp_free(m, packed);
  }
  return st;
}

extern "C" {

int o3d_estimate_normals(o3d_ctx *ctx, const double *xyz, int64_t n, int search_type, int knn, double radius, int mem,
                         double *normals, int has_normals) {
  if (!ctx) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  if (static_cast<o3d_ctx_full *>(ctx)->ranks.size() > 1 && n > 0 && xyz && normals)
    return multi_estimate_normals(static_cast<o3d_ctx_full *>(ctx), xyz, n, search_type, knn, radius, mem, normals, has_normals);
  DeviceGuard guard(ctx->device);
  return o3d_estimate_normals_impl(ctx, xyz, n, search_type, knn, radius, mem, normals, has_normals);
}

int o3d_set_fast_normal_computation(o3d_ctx *ctx, int fast_normal_computation) {
  if (!ctx) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  ctx->normals_fast = fast_normal_computation ? 1 : 0;
  for (o3d_ctx *r : static_cast<o3d_ctx_full *>(ctx)->ranks)
    if (r != ctx) r->normals_fast = ctx->normals_fast;
  return O3D_OK;
}

int o3d_estimate_normals_shard(o3d_ctx *ctx, const double *xyz, int64_t n, int search_type, int knn, double radius, int mem,
                               const double *oriented, int shard, int n_shards, double *out_part, int64_t *out_ng) {
  if (!ctx) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  DeviceGuard guard(ctx->device);
  return o3d_estimate_normals_shard_impl(ctx, xyz, n, search_type, knn, radius, mem, oriented, shard, n_shards, out_part, out_ng);
}

int o3d_estimate_normals_assemble(o3d_ctx *ctx, const double *xyz, int64_t n, const double *all_parts, int n_shards, int mem,
                                  double *normals) {
  if (!ctx) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  DeviceGuard guard(ctx->device);
  return o3d_estimate_normals_assemble_impl(ctx, xyz, n, all_parts, n_shards, mem, normals);
}

int o3d_compute_point_cloud_distance(o3d_ctx *ctx, const double *src_xyz, int64_t n_src, const double *tgt_xyz, int64_t n_tgt, int mem,
                                     double *out_dist) {
  if (!ctx) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  DeviceGuard guard(ctx->device);
  return o3d_neighbor_query_impl(ctx, 0, tgt_xyz, n_tgt, src_xyz, n_src, 1, 0.0, mem, out_dist, nullptr);
}

int o3d_compute_nearest_neighbor_distance(o3d_ctx *ctx, const double *xyz, int64_t n, int mem, double *out_dist) {
  if (!ctx) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  DeviceGuard guard(ctx->device);
  return o3d_neighbor_query_impl(ctx, 0, xyz, n, xyz, n, 2, 0.0, mem, out_dist, nullptr);
}

int o3d_radius_neighbor_count(o3d_ctx *ctx, const double *xyz, int64_t n, double search_radius, int mem, int32_t *out_count) {
  if (!ctx) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  DeviceGuard guard(ctx->device);
  return o3d_neighbor_query_impl(ctx, 2, xyz, n, xyz, n, 1, search_radius, mem, nullptr, out_count);
}

int o3d_knn_mean_distance(o3d_ctx *ctx, const double *xyz, int64_t n, int nb_neighbors, int mem, double *out_mean) {
  if (!ctx) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  DeviceGuard guard(ctx->device);
  return o3d_neighbor_query_impl(ctx, 1, xyz, n, xyz, n, nb_neighbors, 0.0, mem, out_mean, nullptr);
}

int o3d_compute_fpfh_feature(o3d_ctx *ctx, const double *xyz, const double *normals, int64_t n, int search_type, int knn, double radius,
                             int mem, double *out_feature) {
  if (!ctx) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  DeviceGuard guard(ctx->device);
  return o3d_fpfh_impl(ctx, xyz, normals, n, search_type, knn, radius, mem, out_feature);
}

int o3d_feature_nearest(o3d_ctx *ctx, const double *src_feature, int64_t n_src, const double *tgt_feature, int64_t n_tgt, int dim,
                        const int32_t *query_ids, int64_t n_q, int mem, int32_t *out_index, double *out_d2) {
  if (!ctx) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  if (n_src < 0 || n_tgt < 0 || n_q < 0 || dim < 1 || (n_src > 0 && !src_feature) || (n_tgt > 0 && !tgt_feature))
    return o3d_fail(ctx, O3D_ERR_INVALID_ARG, "o3d_feature_nearest: bad arguments");
  if (!query_ids) n_q = n_src;
  if (n_q == 0) return O3D_OK;
  if (!out_index) return o3d_fail(ctx, O3D_ERR_INVALID_ARG, "o3d_feature_nearest: null output");
  DeviceGuard guard(ctx->device);
  ctx->icp_ready = 0;  // the scratch slots of the loop are reused
  DevBufs b;
  const double *d_sf, *d_tf;
  const int32_t *d_q;
  O3D_TRY(b.in(ctx, src_feature, (size_t)n_src * dim, mem, &d_sf));
  O3D_TRY(b.in(ctx, tgt_feature, (size_t)n_tgt * dim, mem, &d_tf));
  O3D_TRY(b.in(ctx, query_ids, (size_t)n_q, mem, &d_q));
  int32_t *d_idx = out_index;
  double *d_d2 = out_d2;
  if (mem == O3D_MEM_HOST) {
    O3D_TRY(b.tmp(ctx, (size_t)n_q, &d_idx));
    if (out_d2) O3D_TRY(b.tmp(ctx, (size_t)n_q, &d_d2));
  }
  O3D_TRY(o3d_feature_nn_device(ctx, d_sf, n_src, d_tf, n_tgt, dim, d_q, n_q, d_idx, d_d2));
  if (mem == O3D_MEM_HOST) {
    O3D_HIP(ctx, o3d_d2h(ctx, out_index, d_idx, sizeof(int32_t) * (size_t)n_q));
    if (out_d2) O3D_HIP(ctx, o3d_d2h(ctx, out_d2, d_d2, sizeof(double) * (size_t)n_q));
  }
  O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return O3D_OK;
}

static void fill_ransac_args(RansacArgs &a, const double *d_src, const double *d_snrm, int64_t n_src, const double *d_tgt,
                             const double *d_tnrm, int64_t n_tgt, int ransac_n, int estimation, int with_scaling,
                             double edge_similarity, double distance_threshold, double normal_angle_threshold) {
  a.src = d_src; a.snrm = d_snrm; a.tgt = d_tgt; a.tnrm = d_tnrm; a.n_src = n_src; a.n_tgt = n_tgt;
  a.ransac_n = ransac_n; a.estimation = estimation; a.with_scaling = with_scaling;
  a.use_edge = edge_similarity >= 0.0; a.use_dist = distance_threshold >= 0.0; a.use_normal = normal_angle_threshold >= 0.0;
  a.edge_sim = edge_similarity; a.dist_thr = distance_threshold;
  a.cos_normal = cos(normal_angle_threshold);  // CorrespondenceChecker.cpp:73
}

int o3d_ransac_hypotheses(o3d_ctx *ctx, const double *src_xyz, const double *src_normals, int64_t n_src, const double *tgt_xyz,
                          const double *tgt_normals, int64_t n_tgt, const int32_t *corres, int n_hyp, int ransac_n, int estimation,
                          int with_scaling, double edge_similarity, double distance_threshold, double normal_angle_threshold, int mem,
                          double *out_T, int32_t *out_ok) {
  if (!ctx) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  if (n_hyp < 0 || ransac_n < 1 || n_src < 0 || n_tgt < 0 || (n_hyp > 0 && (!corres || !out_T || !out_ok)) ||
      (n_src > 0 && !src_xyz) || (n_tgt > 0 && !tgt_xyz))
    return o3d_fail(ctx, O3D_ERR_INVALID_ARG, "o3d_ransac_hypotheses: bad arguments");
  if (estimation != O3D_EST_POINT_TO_POINT && estimation != O3D_EST_POINT_TO_PLANE)
    return o3d_fail(ctx, O3D_ERR_UNSUPPORTED, "o3d_ransac_hypotheses: estimation type %d", estimation);
  if (n_hyp == 0) return O3D_OK;
  DeviceGuard guard(ctx->device);
  DevBufs b;
  const double *d_src, *d_snrm, *d_tgt, *d_tnrm;
  const int32_t *d_c;
  O3D_TRY(b.in(ctx, src_xyz, 3 * (size_t)n_src, mem, &d_src));
  O3D_TRY(b.in(ctx, src_normals, 3 * (size_t)n_src, mem, &d_snrm));
  O3D_TRY(b.in(ctx, tgt_xyz, 3 * (size_t)n_tgt, mem, &d_tgt));
  O3D_TRY(b.in(ctx, tgt_normals, 3 * (size_t)n_tgt, mem, &d_tnrm));
  O3D_TRY(b.in(ctx, corres, 2 * (size_t)ransac_n * (size_t)n_hyp, mem, &d_c));
  double *d_T = out_T;
  int32_t *d_ok = out_ok;
  if (mem == O3D_MEM_HOST) {
    O3D_TRY(b.tmp(ctx, 16 * (size_t)n_hyp, &d_T));
    O3D_TRY(b.tmp(ctx, (size_t)n_hyp, &d_ok));
  }
  RansacArgs a;
  fill_ransac_args(a, d_src, d_snrm, n_src, d_tgt, d_tnrm, n_tgt, ransac_n, estimation, with_scaling, edge_similarity,
                   distance_threshold, normal_angle_threshold);
  a.corres = d_c; a.n_hyp = n_hyp; a.out_T = d_T; a.out_ok = d_ok;
  O3D_TRY(o3d_launch_ransac_hypotheses(ctx, a));
  if (mem == O3D_MEM_HOST) {
    O3D_HIP(ctx, o3d_d2h(ctx, out_T, d_T, sizeof(double) * 16 * (size_t)n_hyp));
    O3D_HIP(ctx, o3d_d2h(ctx, out_ok, d_ok, sizeof(int32_t) * (size_t)n_hyp));
  }
  O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return O3D_OK;
}

int o3d_registration_ransac_feature_matching(o3d_ctx *ctx, const double *src_xyz, const double *src_normals, int64_t n_src,
                                             const double *tgt_xyz, const double *tgt_normals, int64_t n_tgt,
                                             const double *src_feature, const double *tgt_feature, int dim, double max_dist,
                                             int estimation, int with_scaling, int ransac_n, double edge_similarity,
                                             double distance_threshold, double normal_angle_threshold, int max_iteration,
                                             int max_validation, int64_t seed, int mem, double *out_T, int32_t *out_corr,
                                             int64_t *out_n_corr, double *out_fitness, double *out_rmse, int *out_counts) {
  if (!ctx) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  if (n_src < 0 || n_tgt < 0 || dim < 1 || (n_src > 0 && (!src_xyz || !src_feature)) || (n_tgt > 0 && (!tgt_xyz || !tgt_feature)))
    return o3d_fail(ctx, O3D_ERR_INVALID_ARG, "RegistrationRANSACBasedOnFeatureMatching: bad arguments");
  if (estimation != O3D_EST_POINT_TO_POINT && estimation != O3D_EST_POINT_TO_PLANE)
    return o3d_fail(ctx, O3D_ERR_UNSUPPORTED, "RegistrationRANSACBasedOnFeatureMatching: estimation type %d is not on the device path",
                    estimation);
  if (n_src >= ((int64_t)1 << 31) || n_tgt >= ((int64_t)1 << 31))
    return o3d_fail(ctx, O3D_ERR_UNSUPPORTED, "RegistrationRANSACBasedOnFeatureMatching: more than 2^31-1 points");
  DeviceGuard guard(ctx->device);
  static const double kIdentity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  double bestT[16];
  memcpy(bestT, kIdentity, sizeof(bestT));
  double bf = 0.0, br = 0.0;  // RegistrationResult(): fitness 0, rmse 0 (Registration.h:112-113)
  int64_t bnc = 0;
  int itr = 0, validations = 0;
  bool have_best = false;
  auto finish_empty = [&]() -> int {
    if (out_T) memcpy(out_T, bestT, sizeof(bestT));
    if (out_n_corr) *out_n_corr = 0;
    if (out_fitness) *out_fitness = 0.0;
    if (out_rmse) *out_rmse = 0.0;
    if (out_counts) { out_counts[0] = itr; out_counts[1] = validations; }
    if (out_corr && n_src > 0) {
      if (mem == O3D_MEM_DEVICE) {
        O3D_HIP(ctx, hipMemsetAsync(out_corr, 0xFF, sizeof(int32_t) * (size_t)n_src, ctx->stream));
        O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
      } else {
        memset(out_corr, 0xFF, sizeof(int32_t) * (size_t)n_src);
      }
    }
    return O3D_OK;
  };
  if (ransac_n < 3 || !(max_dist > 0.0) || n_src == 0 || n_tgt == 0 || max_iteration <= 0) return finish_empty();  // :207-209

  DevBufs b;
  const double *d_src, *d_snrm, *d_tgt, *d_tnrm, *d_sf, *d_tf;
  O3D_TRY(b.in(ctx, src_xyz, 3 * (size_t)n_src, mem, &d_src));
  O3D_TRY(b.in(ctx, src_normals, 3 * (size_t)n_src, mem, &d_snrm));
  O3D_TRY(b.in(ctx, tgt_xyz, 3 * (size_t)n_tgt, mem, &d_tgt));
  O3D_TRY(b.in(ctx, tgt_normals, 3 * (size_t)n_tgt, mem, &d_tnrm));
  O3D_TRY(b.in(ctx, src_feature, (size_t)dim * (size_t)n_src, mem, &d_sf));
  O3D_TRY(b.in(ctx, tgt_feature, (size_t)dim * (size_t)n_tgt, mem, &d_tf));

  // Hypotheses per round trip: 256 to begin with, doubled (up to kBatch) after every batch in which the checkers let
  // nothing through -- such a batch costs one small kernel and two host round trips, whatever its size.
  const int kBatch = 4096;
  int batch = 256;
  // similar_features (:215, :235-242): the feature-space nearest neighbour of a source point, found when it is first
  // sampled.  When the loop can sample most of the cloud anyway, all of them are found in one launch up front.
  std::vector<int32_t> nn((size_t)n_src, -2);
  int32_t *d_nn, *d_ids, *d_corres, *d_ok;
  double *d_T;
  const size_t max_ids = (size_t)kBatch * (size_t)ransac_n;
  O3D_TRY(b.tmp(ctx, std::max<size_t>((size_t)n_src, max_ids), &d_nn));
  O3D_TRY(b.tmp(ctx, max_ids, &d_ids));
  O3D_TRY(b.tmp(ctx, 2 * max_ids, &d_corres));
  O3D_TRY(b.tmp(ctx, (size_t)kBatch, &d_ok));
  O3D_TRY(b.tmp(ctx, 16 * (size_t)kBatch, &d_T));
  const bool all_up_front = (double)max_iteration * ransac_n >= 0.5 * (double)n_src;
  if (all_up_front) {
    O3D_TRY(o3d_feature_nn_device(ctx, d_sf, n_src, d_tf, n_tgt, dim, nullptr, n_src, d_nn, nullptr));
    O3D_HIP(ctx, o3d_d2h(ctx, nn.data(), d_nn, sizeof(int32_t) * (size_t)n_src));
    O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }

  // validation set-up: target grid, source in processing order with an untouched copy (as o3d_evaluate_hypotheses)
  O3D_TRY(o3d_icp_setup(ctx, d_src, n_src, d_tgt, nullptr, n_tgt, max_dist, kIdentity, O3D_EST_POINT_TO_POINT, 0, 0.0, 0.0, 0,
                        O3D_MEM_DEVICE, n_src));
  O3D_TRY(o3d_reserve(ctx, (void **)&ctx->src_orig, &ctx->cap_src_orig, sizeof(double) * 3 * (size_t)n_src));
  O3D_HIP(ctx, hipMemcpyAsync(ctx->src_orig, ctx->src_sorted, sizeof(double) * 3 * (size_t)n_src, hipMemcpyDeviceToDevice, ctx->stream));
  void *d_all;
  O3D_TRY(o3d_scratch(ctx, 7, sizeof(double) * O3D_NSUMS * (size_t)kBatch, &d_all));
  ctx->hyp_mode = 1;

  std::mt19937 rng;
  if (seed < 0) {
    std::random_device rd;  // utility/Helper.cpp:105-109
    rng.seed(rd());
  } else {
    rng.seed((uint32_t)seed);
  }
  std::uniform_int_distribution<int> pick(0, (int)n_src - 1);
  std::vector<int32_t> ids(max_ids), corres(2 * max_ids), ok(kBatch), pending;
  std::vector<double> Ts(16 * (size_t)kBatch), sums((size_t)kBatch * O3D_NSUMS);
  RansacArgs ra;
  fill_ransac_args(ra, d_src, d_snrm, n_src, d_tgt, d_tnrm, n_tgt, ransac_n, estimation, with_scaling, edge_similarity,
                   distance_threshold, normal_angle_threshold);
  ra.corres = d_corres; ra.out_T = d_T; ra.out_ok = d_ok;

  auto run = [&]() -> int {
    while (itr < max_iteration && validations < max_validation) {
      const int nb = std::min(batch, max_iteration - itr);
      const size_t nid = (size_t)nb * (size_t)ransac_n;
      for (size_t k = 0; k < nid; k++) ids[k] = pick(rng);  // :234
      if (!all_up_front) {
        pending.clear();
        for (size_t k = 0; k < nid; k++)
          if (nn[(size_t)ids[k]] == -2) { nn[(size_t)ids[k]] = -3; pending.push_back(ids[k]); }
        if (!pending.empty()) {
          O3D_HIP(ctx, o3d_h2d(ctx, d_ids, pending.data(), sizeof(int32_t) * pending.size()));
          O3D_TRY(o3d_feature_nn_device(ctx, d_sf, n_src, d_tf, n_tgt, dim, d_ids, (int64_t)pending.size(), d_nn, nullptr));
          std::vector<int32_t> got(pending.size());
          O3D_HIP(ctx, o3d_d2h(ctx, got.data(), d_nn, sizeof(int32_t) * pending.size()));
          O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
          for (size_t k = 0; k < pending.size(); k++) nn[(size_t)pending[k]] = got[k];
        }
      }
      for (size_t k = 0; k < nid; k++) { corres[2 * k] = ids[k]; corres[2 * k + 1] = nn[(size_t)ids[k]]; }  // :243-245
      O3D_HIP(ctx, o3d_h2d(ctx, d_corres, corres.data(), sizeof(int32_t) * 2 * nid));
      ra.n_hyp = nb;
      O3D_TRY(o3d_launch_ransac_hypotheses(ctx, ra));  // :250-268
      O3D_HIP(ctx, o3d_d2h(ctx, ok.data(), d_ok, sizeof(int32_t) * (size_t)nb));
      O3D_HIP(ctx, o3d_d2h(ctx, Ts.data(), d_T, sizeof(double) * 16 * (size_t)nb));
      O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
      // validation (:269-284) of the hypotheses that passed, in iteration order, until max_validation is reached
      int nv = 0, used = nb;  // used: iterations of this batch the reference's loop would have run (:230)
      std::vector<int> which;
      for (int k = 0; k < nb; k++) {
        if (validations + nv >= max_validation) { used = k; break; }
        if (!ok[k]) continue;
        O3D_TRY(o3d_launch_set_hypothesis(ctx, Ts.data() + 16 * (size_t)k));
        O3D_TRY(o3d_icp_pass_to(ctx, (double *)d_all + (size_t)nv * O3D_NSUMS));
        which.push_back(k);
        nv++;
      }
      if (nv > 0) {
        O3D_HIP(ctx, o3d_d2h(ctx, sums.data(), d_all, sizeof(double) * (size_t)nv * O3D_NSUMS));
        O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int v = 0; v < nv; v++) {
          const double nc = sums[(size_t)v * O3D_NSUMS + O3D_SUM_NC];
          const double fit = nc > 0.0 ? nc / (double)n_src : 0.0;
          const double rm = nc > 0.0 ? sqrt(sums[(size_t)v * O3D_NSUMS + O3D_SUM_D2] / nc) : 0.0;
          if (fit > bf || (fit == bf && rm < br)) {  // :272-275
            bf = fit; br = rm; bnc = (int64_t)nc; have_best = true;
            memcpy(bestT, Ts.data() + 16 * (size_t)which[v], sizeof(bestT));
          }
        }
        validations += nv;
      }
      itr += used;
      if (nv == 0) batch = std::min(2 * batch, kBatch);
    }
    if (out_T) memcpy(out_T, bestT, sizeof(bestT));
    if (out_n_corr) *out_n_corr = bnc;
    if (out_fitness) *out_fitness = bf;
    if (out_rmse) *out_rmse = br;
    if (out_counts) { out_counts[0] = itr; out_counts[1] = validations; }
    if (out_corr) {
      if (have_best) {  // the correspondence set of the kept result
        O3D_TRY(o3d_launch_set_hypothesis(ctx, bestT));
        O3D_TRY(o3d_icp_pass(ctx));
        O3D_TRY(o3d_icp_fetch(ctx, mem, out_corr, nullptr, nullptr, nullptr));
      } else if (mem == O3D_MEM_DEVICE) {
        O3D_HIP(ctx, hipMemsetAsync(out_corr, 0xFF, sizeof(int32_t) * (size_t)n_src, ctx->stream));
        O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
      } else {
        memset(out_corr, 0xFF, sizeof(int32_t) * (size_t)n_src);
      }
    }
    return O3D_OK;
  };
  const int st = run();
  ctx->hyp_mode = 0;
  ctx->icp_ready = 0;
  (void)hipStreamSynchronize(ctx->stream);  // nothing of this call may still read the buffers DevBufs frees
  return st;
}

int o3d_registration_ransac_correspondence(o3d_ctx *ctx, const double *src_xyz, int64_t n_src, const double *tgt_xyz,
                                           const double *tgt_normals, int64_t n_tgt, const int32_t *corres, int64_t n_corres,
                                           double max_dist, int estimation, int with_scaling, int ransac_n, int max_iteration,
                                           int max_validation, int64_t seed, int mem, double *out_T, int32_t *out_inlier,
                                           int64_t *out_n_inlier, double *out_fitness, double *out_rmse) {
  if (!ctx) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  if (n_src < 0 || n_tgt < 0 || n_corres < 0 || (n_src > 0 && !src_xyz) || (n_tgt > 0 && !tgt_xyz) || (n_corres > 0 && !corres))
    return o3d_fail(ctx, O3D_ERR_INVALID_ARG, "RegistrationRANSACBasedOnCorrespondence: bad arguments");
  if (estimation != O3D_EST_POINT_TO_POINT && estimation != O3D_EST_POINT_TO_PLANE)
    return o3d_fail(ctx, O3D_ERR_UNSUPPORTED, "RegistrationRANSACBasedOnCorrespondence: estimation type %d is not on the device path",
                    estimation);
  if (n_corres >= ((int64_t)1 << 31)) return o3d_fail(ctx, O3D_ERR_UNSUPPORTED, "more than 2^31-1 correspondences");
  DeviceGuard guard(ctx->device);
  static const double kIdentity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  double bestT[16];
  memcpy(bestT, kIdentity, sizeof(bestT));
  double bf = 0.0, br = 0.0;
  int64_t bgood = 0;
  int best = -1;
  const int n_hyp = std::min(max_iteration, max_validation);  // :183
  auto write_out = [&](const int32_t *d_mask) -> int {
    if (out_T) memcpy(out_T, bestT, sizeof(bestT));
    if (out_n_inlier) *out_n_inlier = bgood;
    if (out_fitness) *out_fitness = bf;
    if (out_rmse) *out_rmse = br;
    if (out_inlier && n_corres > 0) {
      if (d_mask) {
        O3D_HIP(ctx, hipMemcpyAsync(out_inlier, d_mask, sizeof(int32_t) * (size_t)n_corres,
                                    mem == O3D_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, ctx->stream));
      } else if (mem == O3D_MEM_DEVICE) {
        O3D_HIP(ctx, hipMemsetAsync(out_inlier, 0, sizeof(int32_t) * (size_t)n_corres, ctx->stream));
      } else {
        memset(out_inlier, 0, sizeof(int32_t) * (size_t)n_corres);
      }
      O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return O3D_OK;
  };
  if (ransac_n < 3 || n_corres < ransac_n || !(max_dist > 0.0) || n_hyp <= 0) return write_out(nullptr);  // :176-178
  ctx->icp_ready = 0;  // scratch slots of the loop are reused

  DevBufs b;
  const double *d_src, *d_tgt, *d_tnrm;
  const int32_t *d_corres;
  O3D_TRY(b.in(ctx, src_xyz, 3 * (size_t)n_src, mem, &d_src));
  O3D_TRY(b.in(ctx, tgt_xyz, 3 * (size_t)n_tgt, mem, &d_tgt));
  O3D_TRY(b.in(ctx, tgt_normals, 3 * (size_t)n_tgt, mem, &d_tnrm));
  O3D_TRY(b.in(ctx, corres, 2 * (size_t)n_corres, mem, &d_corres));
  std::vector<int32_t> hc;  // host copy of the list for the sampling
  const int32_t *h_corres = corres;
  if (mem == O3D_MEM_DEVICE) {
    hc.resize(2 * (size_t)n_corres);
    O3D_HIP(ctx, o3d_d2h(ctx, hc.data(), corres, sizeof(int32_t) * hc.size()));
    O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    h_corres = hc.data();
  }
  std::mt19937 rng;
  if (seed < 0) {
    std::random_device rd;
    rng.seed(rd());
  } else {
    rng.seed((uint32_t)seed);
  }
  std::uniform_int_distribution<int> pick(0, (int)n_corres - 1);
  std::vector<int32_t> sets(2 * (size_t)ransac_n * (size_t)n_hyp);
  for (size_t k = 0; k < (size_t)ransac_n * (size_t)n_hyp; k++) {  // :184-186
    const int c = pick(rng);
    sets[2 * k] = h_corres[2 * (size_t)c];
    sets[2 * k + 1] = h_corres[2 * (size_t)c + 1];
  }
  int32_t *d_sets, *d_ok, *d_mask;
  double *d_T, *d_res;
  O3D_TRY(b.tmp(ctx, sets.size(), &d_sets));
  O3D_TRY(b.tmp(ctx, (size_t)n_hyp, &d_ok));
  O3D_TRY(b.tmp(ctx, 16 * (size_t)n_hyp, &d_T));
  O3D_TRY(b.tmp(ctx, 2 * (size_t)n_hyp, &d_res));
  O3D_TRY(b.tmp(ctx, (size_t)n_corres, &d_mask));
  O3D_HIP(ctx, o3d_h2d(ctx, d_sets, sets.data(), sizeof(int32_t) * sets.size()));
  RansacArgs ra;
  fill_ransac_args(ra, d_src, nullptr, n_src, d_tgt, d_tnrm, n_tgt, ransac_n, estimation, with_scaling, -1.0, -1.0, -1.0);
  ra.corres = d_sets; ra.n_hyp = n_hyp; ra.out_T = d_T; ra.out_ok = d_ok;
  auto run = [&]() -> int {
    O3D_TRY(o3d_launch_ransac_hypotheses(ctx, ra));  // :187
    O3D_TRY(o3d_ransac_corr_eval_device(ctx, d_T, n_hyp, d_src, n_src, d_tgt, n_tgt, d_corres, n_corres, max_dist, d_res));  // :188-190
    std::vector<double> res(2 * (size_t)n_hyp), Ts(16 * (size_t)n_hyp);
    O3D_HIP(ctx, o3d_d2h(ctx, res.data(), d_res, sizeof(double) * res.size()));
    O3D_HIP(ctx, o3d_d2h(ctx, Ts.data(), d_T, sizeof(double) * Ts.size()));
    O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int h = 0; h < n_hyp; h++) {  // :191-193, in iteration order
      const double good = res[2 * (size_t)h];
      const double fit = good > 0.0 ? good / (double)n_corres : 0.0;
      const double rm = good > 0.0 ? sqrt(res[2 * (size_t)h + 1] / good) : 0.0;
      if (fit > bf || (fit == bf && rm < br)) { bf = fit; br = rm; bgood = (int64_t)good; best = h; }
    }
    if (best < 0) return write_out(nullptr);
    memcpy(bestT, Ts.data() + 16 * (size_t)best, sizeof(bestT));
    O3D_TRY(o3d_ransac_corr_mask_device(ctx, d_T + 16 * (size_t)best, d_src, n_src, d_tgt, n_tgt, d_corres, n_corres, max_dist, d_mask));
    return write_out(d_mask);
  };
  const int st = run();
  (void)hipStreamSynchronize(ctx->stream);
  return st;
}

// Tuning aid (not part of the drop-in surface): milliseconds of one host <-> device copy of `bytes` of pageable memory
// through o3d_h2d / o3d_d2h (staged = 1) or plain hipMemcpyAsync (staged = 0), best of `reps`.
int o3d_debug_copy_bench(o3d_ctx *ctx, size_t bytes, int to_device, int staged, int reps, double *out_ms) {
  if (!ctx || !out_ms || bytes == 0) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  DeviceGuard guard(ctx->device);
  void *d = nullptr;
  O3D_HIP(ctx, hipMalloc(&d, bytes));
  char *h = (char *)malloc(bytes);
  if (!h) { (void)hipFree(d); return o3d_fail(ctx, O3D_ERR_OOM, "host allocation"); }
  memset(h, 1, bytes);
  double best = 1e30;
  int st = O3D_OK;
  for (int r = 0; r < reps + 1 && st == O3D_OK; r++) {
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e;
    if (staged) e = to_device ? o3d_h2d(ctx, d, h, bytes) : o3d_d2h(ctx, h, d, bytes);
    else e = hipMemcpyAsync(to_device ? d : (void *)h, to_device ? (const void *)h : d, bytes,
                            to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) st = o3d_fail(ctx, O3D_ERR_HIP, "copy bench: %s", hipGetErrorString(e));
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (r > 0 && ms < best) best = ms;
  }
  *out_ms = best;
  free(h);
  (void)hipFree(d);
  return st;
}

// Self-check of the staged copy path (xfer.hip): two pattern-filled pageable buffers go up back to back and come down
// back to back, `rounds` times; *out_mismatches = number of 8-byte words that came back different.
int64_t o3d_expand_correspondences(const int32_t *corr, int64_t n, int32_t *out_pairs) {
  if (!corr || !out_pairs || n <= 0) return 0;
  int64_t m = 0;
  for (int64_t i = 0; i < n; i++) {
    if (corr[i] >= 0) {
      out_pairs[2 * m] = (int32_t)i;
      out_pairs[2 * m + 1] = corr[i];
      m++;
    }
  }
  return m;
}

int o3d_debug_copy_check(o3d_ctx *ctx, size_t bytes, int rounds, int64_t *out_mismatches) {
  if (!ctx || !out_mismatches || bytes < 16) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  DeviceGuard guard(ctx->device);
  const size_t nw = bytes / 8;
  void *da = nullptr, *db = nullptr;
  O3D_HIP(ctx, hipMalloc(&da, nw * 8));
  hipError_t e0 = hipMalloc(&db, nw * 8);
  std::vector<uint64_t> a, b, a2, b2;
  int st = O3D_OK;
  int64_t bad = 0;
  try {
    a.resize(nw); b.resize(nw); a2.resize(nw); b2.resize(nw);
  } catch (...) {
    st = o3d_fail(ctx, O3D_ERR_OOM, "host allocation");
  }
  if (e0 != hipSuccess) st = o3d_fail(ctx, O3D_ERR_OOM, "device allocation");
  for (int r = 0; r < rounds && st == O3D_OK; r++) {
    for (size_t k = 0; k < nw; k++) {
      a[k] = (k * 0x9E3779B97F4A7C15ull) ^ (uint64_t)(2 * r + 1);
      b[k] = ~a[k] + k;
    }
    std::fill(a2.begin(), a2.end(), 0);
    std::fill(b2.begin(), b2.end(), 0);
    hipError_t e = o3d_h2d(ctx, da, a.data(), nw * 8);
    if (e == hipSuccess) e = o3d_h2d(ctx, db, b.data(), nw * 8);
    if (e == hipSuccess) e = o3d_d2h(ctx, a2.data(), da, nw * 8);
    if (e == hipSuccess) e = o3d_d2h(ctx, b2.data(), db, nw * 8);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { st = o3d_fail(ctx, O3D_ERR_HIP, "copy check: %s", hipGetErrorString(e)); break; }
    for (size_t k = 0; k < nw; k++) bad += (a2[k] != a[k]) + (b2[k] != b[k]);
  }
  *out_mismatches = bad;
  if (da) (void)hipFree(da);
  if (db) (void)hipFree(db);
  return st;
}

}  // extern "C"

// ---- clouds that stay in HBM across calls --------------------------------------------------------------------------------
// The reference passes `const PointCloud &` from call to call (Registration.h:149-152, PointCloud.h:130,192); behind a C ABI of
// raw host pointers every call would upload its clouds again -- config 3 from host buffers moved 2 x 240 MB three times, config
// 4 three times 2 x 2.4 GB.  An o3d_cloud is a PointCloud's points_ / normals_ / colors_ (PointCloud.h:330-334) in device
// memory owned by the library: uploaded once, down-sampled / given normals / registered without leaving the GPU.  The entry
// points are the existing ones with mem = O3D_MEM_DEVICE underneath, so every result is the host-buffer call's bit for bit.
struct o3d_cloud {
  int device = 0;
  int64_t n = 0;
  double *xyz = nullptr, *nrm = nullptr, *col = nullptr;
};

extern "C" {

void o3d_cloud_free(o3d_cloud *c) {
  if (!c) return;
  DeviceGuard guard(c->device);
  for (double *p : {c->xyz, c->nrm, c->col})
    if (p) (void)hipFree(p);
  delete c;
}

int o3d_cloud_upload(o3d_ctx *ctx, const double *xyz, const double *nrm, const double *col, int64_t n, int mem, o3d_cloud **out) {
  if (!ctx || !out || n < 0 || (n > 0 && !xyz)) return o3d_fail(ctx, O3D_ERR_INVALID_ARG, "o3d_cloud_upload: bad sizes or null buffers");
  *out = nullptr;
  O3D_LOCK(ctx);
  DeviceGuard guard(ctx->device);
  o3d_cloud *c = new (std::nothrow) o3d_cloud();
  if (!c) return O3D_ERR_OOM;
  c->device = ctx->device;
  c->n = n;
  const size_t bytes = sizeof(double) * 3 * (size_t)(n > 0 ? n : 1);
  const double *in[3] = {xyz, nrm, col};
  double **dst[3] = {&c->xyz, &c->nrm, &c->col};
  for (int k = 0; k < 3; k++) {
    if (!in[k] && k > 0) continue;
    hipError_t e = hipMalloc((void **)dst[k], bytes);
    if (e == hipSuccess && n > 0)
      e = mem == O3D_MEM_DEVICE ? hipMemcpyAsync(*dst[k], in[k], sizeof(double) * 3 * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream)
                                : o3d_h2d(ctx, *dst[k], in[k], sizeof(double) * 3 * (size_t)n);
    if (e != hipSuccess) {
      o3d_cloud_free(c);
      return o3d_fail(ctx, e == hipErrorOutOfMemory ? O3D_ERR_OOM : O3D_ERR_HIP, "o3d_cloud_upload: %s", hipGetErrorString(e));
    }
  }
  O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the caller's buffers are free when this returns
  *out = c;
  return O3D_OK;
}

int64_t o3d_cloud_size(const o3d_cloud *c) { return c ? c->n : 0; }
int o3d_cloud_has_normals(const o3d_cloud *c) { return c && c->nrm && c->n > 0; }
int o3d_cloud_has_colors(const o3d_cloud *c) { return c && c->col && c->n > 0; }
const double *o3d_cloud_points(const o3d_cloud *c) { return c ? c->xyz : nullptr; }
const double *o3d_cloud_normals(const o3d_cloud *c) { return c ? c->nrm : nullptr; }
const double *o3d_cloud_colors(const o3d_cloud *c) { return c ? c->col : nullptr; }

int o3d_cloud_download(o3d_ctx *ctx, const o3d_cloud *c, double *out_xyz, double *out_nrm, double *out_col) {
  if (!ctx || !c) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  DeviceGuard guard(ctx->device);
  const size_t bytes = sizeof(double) * 3 * (size_t)c->n;
  if (c->n > 0) {
    if (out_xyz) O3D_HIP(ctx, o3d_d2h(ctx, out_xyz, c->xyz, bytes));
    if (out_nrm && c->nrm) O3D_HIP(ctx, o3d_d2h(ctx, out_nrm, c->nrm, bytes));
    if (out_col && c->col) O3D_HIP(ctx, o3d_d2h(ctx, out_col, c->col, bytes));
  }
  O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return O3D_OK;
}

int o3d_cloud_voxel_down_sample(o3d_ctx *ctx, const o3d_cloud *in, double voxel_size, o3d_cloud **out) {
  if (!ctx || !in || !out) return O3D_ERR_INVALID_ARG;
  *out = nullptr;
  O3D_LOCK(ctx);
  DeviceGuard guard(ctx->device);
  o3d_cloud *c = new (std::nothrow) o3d_cloud();
  if (!c) return O3D_ERR_OOM;
  c->device = ctx->device;
  double *own[3] = {nullptr, nullptr, nullptr};
  int64_t m = 0;
  const int st = o3d_voxel_down_sample_impl(ctx, in->xyz, in->nrm, in->col, in->n, voxel_size, O3D_MEM_DEVICE, nullptr, nullptr, nullptr,
                                            nullptr, &m, own);
  if (st != O3D_OK) { delete c; return st; }
  c->n = m;
  c->xyz = own[0]; c->nrm = own[1]; c->col = own[2];
  if (!c->xyz) {  // an empty cloud (or one without a finite point): still a valid handle
    if (hipMalloc((void **)&c->xyz, 256) != hipSuccess) { delete c; return o3d_fail(ctx, O3D_ERR_OOM, "o3d_cloud_voxel_down_sample"); }
    c->n = 0;
  }
  *out = c;
  return O3D_OK;
}

int o3d_cloud_estimate_normals(o3d_ctx *ctx, o3d_cloud *c, int search_type, int knn, double radius) {
  if (!ctx || !c) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  const int had = c->nrm != nullptr;
  if (!had) {
    DeviceGuard guard(ctx->device);
    O3D_HIP(ctx, hipMalloc((void **)&c->nrm, sizeof(double) * 3 * (size_t)(c->n > 0 ? c->n : 1)));
  }
  if (c->n == 0) return O3D_OK;
  const int st = o3d_estimate_normals(ctx, c->xyz, c->n, search_type, knn, radius, O3D_MEM_DEVICE, c->nrm, had);
  if (st == O3D_OK) {  // (device memory: the call returns with its kernels enqueued)
    DeviceGuard guard(ctx->device);
    O3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  return st;
}

int o3d_cloud_registration_icp(o3d_ctx *ctx, const o3d_cloud *src, const o3d_cloud *tgt, double max_dist, const double init[16],
                               int estimation, int with_scaling, double rel_fitness, double rel_rmse, int max_iter, double out_T[16],
                               int32_t *out_corr, int64_t *out_n_corr, double *out_fitness, double *out_rmse, int *out_iters) {
  if (!ctx || !src || !tgt) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);  // held for the whole call
  if (static_cast<o3d_ctx_full *>(ctx)->ranks.size() > 1)
    return multi_registration_icp(static_cast<o3d_ctx_full *>(ctx), src->xyz, src->n, tgt->xyz, tgt->nrm, tgt->n, max_dist, init, estimation,
                                  with_scaling, rel_fitness, rel_rmse, max_iter, O3D_MEM_DEVICE, out_T, out_corr, out_n_corr, out_fitness,
                                  out_rmse, out_iters, O3D_MEM_HOST);
  DeviceGuard guard(ctx->device);
  O3D_TRY(o3d_icp_setup(ctx, src->xyz, src->n, tgt->xyz, tgt->nrm, tgt->n, max_dist, init, estimation, with_scaling, rel_fitness, rel_rmse,
                        max_iter, O3D_MEM_DEVICE, src->n));
  O3D_TRY(run_icp_loop(ctx, max_iter));
  return o3d_icp_finish(ctx, O3D_MEM_HOST, out_T, out_corr, out_n_corr, out_fitness, out_rmse, out_iters);
}

int o3d_cloud_evaluate_registration(o3d_ctx *ctx, const o3d_cloud *src, const o3d_cloud *tgt, double max_dist, const double T[16],
                                    int32_t *out_corr, int64_t *out_n_corr, double *out_fitness, double *out_rmse) {
  if (!ctx || !src || !tgt || !T) return O3D_ERR_INVALID_ARG;
  O3D_LOCK(ctx);
  if (!(max_dist > 0.0))  // Registration.cpp:45-47: an empty result (host outputs)
    return o3d_evaluate_registration(ctx, nullptr, src->n, nullptr, tgt->n, max_dist, T, O3D_MEM_HOST, out_corr, out_n_corr, out_fitness, out_rmse);
  O3D_TRY(o3d_icp_setup(ctx, src->xyz, src->n, tgt->xyz, nullptr, tgt->n, max_dist, T, O3D_EST_POINT_TO_POINT, 0, 0.0, 0.0, 0, O3D_MEM_DEVICE,
                        src->n));
  O3D_TRY(o3d_icp_step(ctx));
  return o3d_icp_finish(ctx, O3D_MEM_HOST, nullptr, out_corr, out_n_corr, out_fitness, out_rmse, nullptr);
}

}  // extern "C"
