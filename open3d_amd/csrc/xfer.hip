// Host <-> device copies of caller buffers.
//
// The drop-in boundary hands over pageable host memory (std::vector<Eigen::Vector3d>::data()).  hipMemcpyAsync stages such
// copies through pinned memory with ONE host thread, which caps them near 10 GB/s while the DMA engines idle (rocprof of the
// 10M-point pipeline: 700 MB in 13 ms of DMA time, 60 ms of wall clock).  Large copies are therefore staged here by
// kThreads persistent host threads, each with two pinned 4 MiB buffers and its own stream: memcpy into pinned memory, DMA from it,
// the next chunk's memcpy overlapping the DMA.  Ordering: the side streams wait for everything enqueued on ctx->stream
// before touching the device buffer, and ctx->stream waits for them before going on.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "o3d_internal.h"

namespace {

constexpr int kThreads = 4;
constexpr size_t kChunk = (size_t)4 << 20;
// Below these sizes the runtime's own path is as good (scripts/copy_bench.py: H2D 24 MiB 0.62 vs 0.67 ms, 48 MiB 1.06 vs
// 1.65 ms; D2H 16 MiB 0.49 vs 0.93 ms; at 240 MiB 54 / 53 GB/s staged against 30 / 18 GB/s plain).
constexpr size_t kStagedMinH2D = (size_t)32 << 20;
constexpr size_t kStagedMinD2H = (size_t)12 << 20;

struct Xfer {
  void *pinned[kThreads][2] = {};
  hipStream_t s[kThreads] = {};
  hipEvent_t ev[kThreads][2] = {};
  hipEvent_t done[kThreads] = {};
  hipEvent_t start = nullptr;
  // Persistent helper threads (a thread's first HIP call costs milliseconds of runtime set-up: not per copy).
  std::thread th[kThreads];
  std::mutex mu;
  std::condition_variable cv_go, cv_done;
  std::function<void(int)> job;
  unsigned long generation = 0;
  int running = 0;
  bool quit = false;
  bool broken = false;  // helper threads could not be started
  int device = 0;
};

void xfer_worker(Xfer *x, int t) {
  (void)hipSetDevice(x->device);
  unsigned long seen = 0;
  for (;;) {
    std::function<void(int)> job;
    {
      std::unique_lock<std::mutex> lk(x->mu);
      x->cv_go.wait(lk, [&] { return x->quit || x->generation != seen; });
      if (x->quit) return;
      seen = x->generation;
      job = x->job;
    }
    job(t);
    {
      std::lock_guard<std::mutex> lk(x->mu);
      if (--x->running == 0) x->cv_done.notify_all();
    }
  }
}

// Runs job(1) .. job(kThreads - 1) on the helpers and job(0) here; returns when all are done.
void xfer_run(Xfer *x, const std::function<void(int)> &job) {
  {
    std::lock_guard<std::mutex> lk(x->mu);
    x->job = job;
    x->running = kThreads - 1;
    x->generation++;
  }
  x->cv_go.notify_all();
  job(0);
  std::unique_lock<std::mutex> lk(x->mu);
  x->cv_done.wait(lk, [&] { return x->running == 0; });
}

hipError_t xfer_init(o3d_ctx *ctx) {
  if (ctx->xfer) return hipSuccess;
  Xfer *x = new Xfer();
  hipError_t e = hipEventCreateWithFlags(&x->start, hipEventDisableTiming);
  for (int t = 0; t < kThreads && e == hipSuccess; t++) {
    e = hipStreamCreateWithFlags(&x->s[t], hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&x->done[t], hipEventDisableTiming);
    for (int b = 0; b < 2 && e == hipSuccess; b++) {
      e = hipHostMalloc(&x->pinned[t][b], kChunk, hipHostMallocDefault);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&x->ev[t][b], hipEventDisableTiming);
    }
  }
  ctx->xfer = x;  // o3d_xfer_destroy releases whatever was created
  x->device = ctx->device;
  if (e != hipSuccess) {  // half-built state must not be taken for a working one by the next copy
    o3d_xfer_destroy(ctx);
    return e;
  }
  {
    try {
      for (int t = 1; t < kThreads; t++) x->th[t] = std::thread(xfer_worker, x, t);
    } catch (...) {  // no thread to be had: the copy falls back to the runtime's path (see staged_copy)
      x->broken = true;
    }
  }
  return e;
}

// to_device: dst = device, src = host; else dst = host, src = device.
hipError_t staged_copy(o3d_ctx *ctx, void *dst, const void *src, size_t bytes, bool to_device) {
  hipError_t e = xfer_init(ctx);
  if (e != hipSuccess) return e;
  Xfer *x = static_cast<Xfer *>(ctx->xfer);
  if (x->broken)
    return hipMemcpyAsync(dst, src, bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, ctx->stream);
  e = hipEventRecord(x->start, ctx->stream);
  if (e != hipSuccess) return e;
  const size_t nchunks = (bytes + kChunk - 1) / kChunk;
  hipError_t errs[kThreads];
  auto work = [&](int t) {
    hipError_t r = hipStreamWaitEvent(x->s[t], x->start, 0);
    size_t round = 0;
    // D2H: chunk k's DMA is enqueued one round before its memcpy out of pinned memory, so that DMA and memcpy overlap
    size_t pend_off[2] = {0, 0}, pend_len[2] = {0, 0};
    bool pend[2] = {false, false};
    for (size_t k = (size_t)t; k < nchunks && r == hipSuccess; k += kThreads, round++) {
      const int b = (int)(round & 1);
      const size_t off = k * kChunk, len = std::min(kChunk, bytes - off);
      if (to_device) {
        // the DMA that last used this pinned buffer -- in this copy or, for the first two rounds, in the previous one,
        // which returned with its last DMAs still in flight (an event never recorded reads as complete)
        r = hipEventSynchronize(x->ev[t][b]);
        if (r != hipSuccess) break;
        memcpy(x->pinned[t][b], (const char *)src + off, len);
        r = hipMemcpyAsync((char *)dst + off, x->pinned[t][b], len, hipMemcpyHostToDevice, x->s[t]);
        if (r == hipSuccess) r = hipEventRecord(x->ev[t][b], x->s[t]);
      } else {
        r = hipEventSynchronize(x->ev[t][b]);  // this buffer's last DMA, of this copy or of the previous one
        if (r != hipSuccess) break;
        if (pend[b]) {  // drain the chunk that used this buffer two rounds ago
          memcpy((char *)dst + pend_off[b], x->pinned[t][b], pend_len[b]);
          pend[b] = false;
        }
        r = hipMemcpyAsync(x->pinned[t][b], (const char *)src + off, len, hipMemcpyDeviceToHost, x->s[t]);
        if (r == hipSuccess) r = hipEventRecord(x->ev[t][b], x->s[t]);
        pend[b] = true; pend_off[b] = off; pend_len[b] = len;
      }
    }
    if (!to_device && r == hipSuccess) {
      for (int i = 0; i < 2 && r == hipSuccess; i++) {
        const int b = (int)((round + i) & 1);  // oldest first
        if (!pend[b]) continue;
        r = hipEventSynchronize(x->ev[t][b]);
        if (r == hipSuccess) memcpy((char *)dst + pend_off[b], x->pinned[t][b], pend_len[b]);
        pend[b] = false;
      }
    }
    if (r == hipSuccess) r = hipEventRecord(x->done[t], x->s[t]);
    errs[t] = r;
  };
  xfer_run(x, work);
  for (int t = 0; t < kThreads; t++)
    if (errs[t] != hipSuccess) return errs[t];
  for (int t = 0; t < kThreads; t++) {
    e = hipStreamWaitEvent(ctx->stream, x->done[t], 0);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace

static bool staged_enabled() {
  static const bool on = getenv("O3D_NO_STAGED") == nullptr;  // tuning aid
  return on;
}

hipError_t o3d_h2d(o3d_ctx *ctx, void *dst, const void *src, size_t bytes) {
  if (bytes == 0) return hipSuccess;
  if (bytes < kStagedMinH2D || !staged_enabled()) return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream);
  return staged_copy(ctx, dst, src, bytes, true);
}

// Large copies return with the data in `dst`; small ones are asynchronous on ctx->stream like hipMemcpyAsync.
hipError_t o3d_d2h(o3d_ctx *ctx, void *dst, const void *src, size_t bytes) {
  if (bytes == 0) return hipSuccess;
  if (bytes < kStagedMinD2H || !staged_enabled()) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream);
  return staged_copy(ctx, dst, src, bytes, false);
}

void o3d_xfer_destroy(o3d_ctx *ctx) {
  Xfer *x = static_cast<Xfer *>(ctx->xfer);
  if (!x) return;
  {
    std::lock_guard<std::mutex> lk(x->mu);
    x->quit = true;
  }
  x->cv_go.notify_all();
  for (int t = 1; t < kThreads; t++)
    if (x->th[t].joinable()) x->th[t].join();
  for (int t = 0; t < kThreads; t++) {
    if (x->s[t]) (void)hipStreamSynchronize(x->s[t]);
    for (int b = 0; b < 2; b++) {
      if (x->pinned[t][b]) (void)hipHostFree(x->pinned[t][b]);
      if (x->ev[t][b]) (void)hipEventDestroy(x->ev[t][b]);
    }
    if (x->done[t]) (void)hipEventDestroy(x->done[t]);
    if (x->s[t]) (void)hipStreamDestroy(x->s[t]);
  }
  if (x->start) (void)hipEventDestroy(x->start);
  delete x;
  ctx->xfer = nullptr;
}
