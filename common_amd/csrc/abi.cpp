// abi.cpp -- the extern "C" surface declared in include/microscopes_hip.h.
// Host logic only: argument checking, table bookkeeping and kernel launches.
// There is deliberately no CPU arithmetic path here: every score / update is a
// kernel in kernels_*.hip, and context creation fails without a gfx950 device.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <memory>
#include <mutex>
#include <new>

#include "launchers.hpp"

namespace msc {

static thread_local std::string g_last_error;

void set_error(const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
}
int fail(int code, const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

size_t primitive_size(int t) {
  static const size_t sz[MSC_TYPE_NELEMS] = {1, 1, 1, 2, 2, 4, 4, 8, 8, 4, 8};
  return (t >= 0 && t < MSC_TYPE_NELEMS) ? sz[t] : 0;
}

static bool family_ok(int f) { return f >= MSC_BB && f <= MSC_DM; }

template <typename T>
static int dev_alloc(std::vector<void *> &owned, T **out, size_t count) {
  void *p = nullptr;
  const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
  MSC_HIP(hipMalloc(&p, bytes));
  MSC_HIP(hipMemset(p, 0, bytes));
  owned.push_back(p);
  *out = static_cast<T *>(p);
  return MSC_OK;
}

}  // namespace msc

using namespace msc;

// ---------------------------------------------------------------------------
// library / context
// ---------------------------------------------------------------------------
extern "C" int msc_abi_version(void) { return MSC_ABI_VERSION; }
extern "C" const char *msc_last_kernel(int which) { return msc::last_kernel(which); }
extern "C" const char *msc_last_error(void) { return g_last_error.c_str(); }
extern "C" const char *msc_build_info(void) {
  return "microscopes_hip abi " "1" " gfx950 (CDNA4) hipcc " __VERSION__;
}

// ---- device-side errors (device_error.hpp) ----------------------------------------------------------------------------
// One pinned pair of words per device, bound into every kernel translation unit the first time a context is created on
// the device; it lives as long as the library.
static int device_error_word(int device, volatile uint32_t **host_out) {
  static std::mutex mu;
  static uint32_t *words[64] = {};
  MSC_REQUIRE(device >= 0 && device < 64, "device %d: at most 64 devices", device);
  std::lock_guard<std::mutex> lock(mu);
  if (!words[device]) {
    uint32_t *h = nullptr;
    void *d = nullptr;
    MSC_HIP(hipHostMalloc(reinterpret_cast<void **>(&h), 64, hipHostMallocMapped));
    h[0] = h[1] = 0;
    MSC_HIP(hipHostGetDevicePointer(&d, h, 0));
    uint32_t *w = static_cast<uint32_t *>(d);
    if (bind_error_word_score(w) || bind_error_word_sweep(w) || bind_error_word_state(w))
      return fail(MSC_EHIP, "binding the device error word failed: %s", hipGetErrorString(hipGetLastError()));
    words[device] = h;
  }
  *host_out = words[device];
  return MSC_OK;
}
// what the launching and synchronising entry points call first / last: an error a kernel of an earlier call reported
static int device_error_check(msc_context *ctx) {
  volatile uint32_t *w = ctx ? ctx->err_host : nullptr;
  if (!w || w[0] == 0) return MSC_OK;
  std::atomic_thread_fence(std::memory_order_acquire);
  const uint32_t code = w[0], detail = w[1];
  w[0] = 0;
  // (code bits: every kind that was reported since the last check is named; `detail` belongs to the first)
  std::string what;
  auto add = [&](const char *t) { if (!what.empty()) what += "; "; what += t; };
  if (code & 1u) add("a wave-subset barrier of a tile kernel timed out (detail: workgroup): rows of that launch are wrong");
  if (code & 2u) add("msc_entity_op: leave from a group the row is not in / an empty group, or join of an assigned row (detail: group)");
  if (code & 4u) add("msc_relation_slice_scores: a block offset beyond the score row (detail: cell)");
  if (code & ~7u) add("unknown device-side error");
  return fail(MSC_EDEVICE, "reported by an earlier kernel on device %d: %s [detail of the first: %u]; rebuild the affected state's tables",
              ctx->device, what.c_str(), detail);
}

extern "C" int msc_context_create(int device, void *stream, msc_context **out) {
  MSC_REQUIRE(out != nullptr, "msc_context_create: out is null");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return fail(MSC_ENODEVICE, "no HIP device visible (%s); this library has no CPU path",
                e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
  MSC_REQUIRE(device >= 0 && device < ndev, "device %d out of range (have %d)", device, ndev);
  hipDeviceProp_t prop;
  MSC_HIP(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(MSC_ENODEVICE, "device %d is %s; kernels are built for gfx950 only", device,
                prop.gcnArchName);
  MSC_HIP(hipSetDevice(device));
  std::unique_ptr<msc_context> ctx(new (std::nothrow) msc_context());
  if (!ctx) return fail(MSC_ENOMEM, "out of host memory");
  ctx->device = device;
  ctx->stream = static_cast<hipStream_t>(stream);
  ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  ctx->mailbox_bytes = 192 * 1024;       // niw at dim 128: a 66 KB record + a 66 KB hp block
  MSC_HIP(hipHostMalloc(&ctx->mailbox_host, ctx->mailbox_bytes, hipHostMallocMapped));
  MSC_HIP(hipHostGetDevicePointer(&ctx->mailbox_dev, ctx->mailbox_host, 0));
  const char *sw = std::getenv("MSC_SYNC_WORD");           // 0: plain hipStreamSynchronize (A/B knob)
  if (sw && std::atoi(sw) == 0) ctx->sync_word_ok = false;
  if (hipHostMalloc(reinterpret_cast<void **>(&ctx->sync_word_host), 64, hipHostMallocMapped) == hipSuccess) {
    *ctx->sync_word_host = 0;
    if (hipHostGetDevicePointer(&ctx->sync_word_dev, ctx->sync_word_host, 0) != hipSuccess) ctx->sync_word_ok = false;
  } else {
    (void)hipGetLastError();
    ctx->sync_word_host = nullptr;
  }
  if (const int rc = device_error_word(device, &ctx->err_host)) {      // (nothing pinned is left behind)
    if (ctx->mailbox_host) (void)hipHostFree(ctx->mailbox_host);
    if (ctx->sync_word_host) (void)hipHostFree(ctx->sync_word_host);
    return rc;
  }
  *out = ctx.release();
  return MSC_OK;
}

static void vmm_free(msc_context::VmmAlloc &a);

extern "C" int msc_context_destroy(msc_context *ctx) {
  if (!ctx) return MSC_OK;
  (void)hipSetDevice(ctx->device);
  for (auto &a : ctx->vmm) vmm_free(a);
  ctx->vmm.clear();
  if (ctx->mailbox_host) (void)hipHostFree(ctx->mailbox_host);
  if (ctx->sync_word_host) (void)hipHostFree(ctx->sync_word_host);
  if (ctx->record_stream) (void)hipStreamDestroy(ctx->record_stream);
  delete ctx;
  return MSC_OK;
}

extern "C" int msc_context_set_stream(msc_context *ctx, void *stream) {
  MSC_REQUIRE(ctx, "null context");
  ctx->stream = static_cast<hipStream_t>(stream);
  return MSC_OK;
}

extern "C" int msc_context_synchronize(msc_context *ctx) {
  MSC_REQUIRE(ctx, "null context");
  // The per-entity paths (hip::mixture_state: remove, score, add) wait here once per move, for kernels that take a few
  // microseconds: waking up from hipStreamSynchronize costs as much again.  So the stream writes a sequence number
  // into a pinned word when it gets here and the host watches the word; bounded, the stream wait is the fallback.
  if (ctx->sync_word_host && ctx->sync_word_ok) {
    const uint32_t seq = ++ctx->sync_seq;
    if (hipStreamWriteValue32(ctx->stream, ctx->sync_word_dev, seq, 0) == hipSuccess) {
      volatile uint32_t *w = ctx->sync_word_host;
      for (int spin = 0; spin < 400000; spin++)
        if (*w == seq) {
          std::atomic_thread_fence(std::memory_order_acquire);
          return device_error_check(ctx);
        }
    } else {
      (void)hipGetLastError();
      ctx->sync_word_ok = false;
    }
  }
  MSC_HIP(hipStreamSynchronize(ctx->stream));
  return device_error_check(ctx);
}

// One candidate of msc_device_alloc_probed: nbytes of device memory mapped from 32 MiB physical chunks that are created
// one by one (hipMemCreate) and mapped side by side into one reserved VA range.  Measured
// (tools/microbench/placement_stitch.hip, four rounds): such buffers take the C2 store stream at 6.2-7.0 TB/s where
// hipMalloc'ed ones of the same process take it at 5.3-6.2 -- whatever the chunk size (2 / 32 / 256 MiB) and mapping
// order.  Returns false (nothing allocated) when the virtual-memory API is not usable.
static bool vmm_alloc(int device, size_t nbytes, msc_context::VmmAlloc *out) {
  const size_t chunk = 32u << 20;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  size_t gran = 0;
  if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || gran == 0 || chunk % gran != 0) {
    (void)hipGetLastError();
    return false;
  }
  msc_context::VmmAlloc a;
  const size_t n = (std::max<size_t>(nbytes, 1) + chunk - 1) / chunk;
  a.size = n * chunk;
  a.va = nullptr;
  if (hipMemAddressReserve(&a.va, a.size, 2u << 20, nullptr, 0) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  bool ok = true;
  size_t mapped = 0;
  for (size_t i = 0; i < n && ok; i++) {
    hipMemGenericAllocationHandle_t h;
    if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) { ok = false; break; }
    a.handles.push_back(h);
    if (hipMemMap(static_cast<char *>(a.va) + i * chunk, chunk, 0, h, 0) != hipSuccess) { ok = false; break; }
    mapped = i + 1;
  }
  if (ok) {
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    ok = hipMemSetAccess(a.va, a.size, &acc, 1) == hipSuccess;
  }
  if (!ok) {
    (void)hipGetLastError();
    if (mapped) (void)hipMemUnmap(a.va, mapped * chunk);
    for (auto h : a.handles) (void)hipMemRelease(h);
    (void)hipMemAddressFree(a.va, a.size);
    return false;
  }
  *out = a;
  return true;
}
static void vmm_free(msc_context::VmmAlloc &a) {
  (void)hipMemUnmap(a.va, a.size);
  for (auto h : a.handles) (void)hipMemRelease(h);
  (void)hipMemAddressFree(a.va, a.size);
  a.handles.clear();
  a.va = nullptr;
}

// Large buffers (from 64 MiB on: score matrices) are PLACED: the same store stream runs 5.5 or 7.0 TB/s depending on where
// the driver put the pages (profiles/r02_placement_study.txt), no allocator argument selects that, and buffers mapped
// from separately created 32 MiB chunks land in the upper band more often than hipMalloc'ed ones.  So a candidate is
// mapped from chunks, stream-filled a few times with the score kernels' store pattern, and kept when it takes the stream
// at `accept_gbps` or better; otherwise the next candidate is tried while the rejected ones are still held (released
// first, the driver would hand the same pages back), and the fastest wins.  Three things end the search early:
//   * `max_candidates` (msc_device_alloc: 12);
//   * the candidates held side by side would exceed HALF of what the device had free when the call began (the transient
//     footprint is the caller's memory too: torch's caching allocator, other tenants);
//   * a FLAT box: after `flat_after` candidates whose fill rates lie within 6 % of each other and below the mark there is
//     no fast stretch to find here (the round-3 driver's box: 24 of 24 at 5.40-5.67 TB/s) -- stop, keep the best.
// Per-chunk selection was tried and does not work: a chunk's rate inside a fill follows its position in the launch, not
// the chunk (tools/microbench/placement_chunks.hip, profiles/r03_placement_chunks.txt).  Synchronous (about 1 ms per
// candidate and GB).
static int alloc_placed(msc_context *ctx, size_t nbytes, uint32_t max_candidates, float accept_gbps, uint32_t flat_after,
                        void **out, float *rates_gbps, uint32_t *chosen) {
  static const bool no_vmm = std::getenv("MSC_ALLOC_NO_VMM") != nullptr;       // (A/B knob: plain hipMalloc candidates)
  struct Cand { void *p; bool vmm; msc_context::VmmAlloc v; };
  std::vector<Cand> bufs;
  std::vector<float> rate;
  auto release = [&](int keep) {
    for (size_t i = 0; i < bufs.size(); i++) {
      if ((int)i == keep) continue;
      if (bufs[i].vmm) vmm_free(bufs[i].v);
      else (void)hipFree(bufs[i].p);
    }
  };
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return fail(MSC_EHIP, "hipEventCreate failed");
  int rc = MSC_OK;
  const int reps = nbytes >= (256u << 20) ? 5 : 8;
  const bool want_vmm = !no_vmm && nbytes >= (64u << 20);                      // small buffers: not worth 32 MiB chunks
  size_t free_at_start = 0;
  {
    size_t total_b = 0;
    if (hipMemGetInfo(&free_at_start, &total_b) != hipSuccess) {
      (void)hipGetLastError();
      free_at_start = 0;
    }
  }
  for (uint32_t i = 0; i < max_candidates; i++) {
    if (i > 0) {                                                               // never take the device's last memory for a probe,
      size_t free_b = 0, total_b = 0;                                          // nor more than half of what was free for all of them
      if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < 2 * nbytes + (1ull << 30) ||
          (flat_after != 0 && (size_t)(i + 1) * nbytes > free_at_start / 2)) {
        (void)hipGetLastError();
        break;
      }
      if (flat_after != 0 && i >= flat_after && rate.size() == i) {            // a flat box: nothing to find
        const float hi = *std::max_element(rate.begin(), rate.end()), lo = *std::min_element(rate.begin(), rate.end());
        if (hi > 0.f && (hi - lo) <= 0.06f * hi) break;
      }
    }
    Cand c{nullptr, false, {}};
    if (want_vmm && vmm_alloc(ctx->device, nbytes, &c.v)) {
      c.p = c.v.va;
      c.vmm = true;
    } else if (hipMalloc(&c.p, nbytes ? nbytes : 1) != hipSuccess) {           // out of memory: settle for what there is
      (void)hipGetLastError();
      break;
    }
    bufs.push_back(c);
    void *p = c.p;
    hipError_t e = hipMemsetAsync(p, 0, nbytes ? nbytes : 1, ctx->stream);          // (zero-filled, like every msc_device_alloc)
    if (max_candidates > 1) {
      if (e == hipSuccess && launch_stream_fill(ctx->stream, ctx->num_cus, p, nbytes)) e = hipErrorLaunchFailure;
      if (e == hipSuccess) e = hipEventRecord(e0, ctx->stream);
      for (int r = 0; r < reps && e == hipSuccess; r++)
        if (launch_stream_fill(ctx->stream, ctx->num_cus, p, nbytes)) e = hipErrorLaunchFailure;
      if (e == hipSuccess) e = hipEventRecord(e1, ctx->stream);
      if (e == hipSuccess) e = hipEventSynchronize(e1);
      float ms = 0.f;
      if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
      if (e != hipSuccess) { rc = fail(MSC_EHIP, "placement probe failed: %s", hipGetErrorString(e)); break; }
      rate.push_back(ms > 0.f ? (float)((double)nbytes * reps / (ms * 1e-3) / 1e9) : 0.f);
      if (rate.back() >= accept_gbps) break;
    } else {
      if (e != hipSuccess) { rc = fail(MSC_EHIP, "hipMemsetAsync: %s", hipGetErrorString(e)); break; }
      rate.push_back(0.f);
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (rc != MSC_OK || bufs.empty() || rate.size() != bufs.size()) {
    (void)hipStreamSynchronize(ctx->stream);
    release(-1);
    return rc != MSC_OK ? rc : fail(MSC_ENOMEM, "allocation of %zu bytes failed", nbytes);
  }
  int best = 0;
  for (size_t i = 1; i < rate.size(); i++)
    if (rate[i] > rate[best]) best = (int)i;
  if (bufs.size() > 1) (void)hipStreamSynchronize(ctx->stream);
  release(best);
  if (bufs[best].vmm) ctx->vmm.push_back(bufs[best].v);
  ctx->last_alloc_rates = rate;
  ctx->last_alloc_chosen = (uint32_t)best;
  // (the stores' mark follows the accept mark when a caller lowers that one -- MSC_ALLOC_ACCEPT_GBPS: the evidence runs of
  // tools/profile_round.sh trace the non-temporal instantiation on boxes without a fast stretch that way)
  ctx->placed.push_back(msc_context::Placed{bufs[best].p, nbytes, max_candidates > 1 && rate[best] >= std::min(kNtFastGbps, accept_gbps)});
  if (rates_gbps)
    for (uint32_t i = 0; i < max_candidates; i++) rates_gbps[i] = i < rate.size() ? rate[i] : 0.f;
  if (chosen) *chosen = (uint32_t)best;
  *out = bufs[best].p;
  return MSC_OK;
}

// device buffers for callers that hold no HIP headers of their own (the assignment vector, score rows, score matrices)
extern "C" int msc_device_alloc(msc_context *ctx, size_t nbytes, void **out) {
  MSC_REQUIRE(ctx && out, "null argument");
  *out = nullptr;
  MSC_HIP(hipSetDevice(ctx->device));
  // MSC_ALLOC_CANDIDATES (default 12; 1 = no probing, 0 = plain hipMalloc) / MSC_ALLOC_ACCEPT_GBPS (default 6650: a
  // candidate that fills at 6.7 TB/s takes the C2 pass at 0.87-0.88 of the HBM roof, one at 6.5 at 0.83-0.87, one at 6.2
  // at 0.80-0.83; when none reaches the mark the best of all is kept, ~1 ms a candidate and GB); after six candidates
  // whose rates all lie within 6 % the box is taken to have no fast stretch.
  // Candidates held side by side walk through physical memory, and where the fast stretches lie differs from box to
  // box: one box offered one within six candidates in ten processes of ten, another none within twelve in one process
  // of twelve (profiles/r03_alloc_distribution.jsonl), the round-3 driver's none within 24
  static const int cand = std::getenv("MSC_ALLOC_CANDIDATES") ? std::atoi(std::getenv("MSC_ALLOC_CANDIDATES")) : 12;
  static const float accept = std::getenv("MSC_ALLOC_ACCEPT_GBPS") ? (float)std::atof(std::getenv("MSC_ALLOC_ACCEPT_GBPS")) : 6650.f;
  constexpr int flat = 6;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  const bool capturing = hipStreamIsCapturing(ctx->stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
  if (nbytes >= (64u << 20) && cand >= 1 && !capturing)
    return alloc_placed(ctx, nbytes, (uint32_t)std::min(cand, 64), accept, (uint32_t)std::max(flat, 2), out, nullptr, nullptr);
  void *p = nullptr;
  MSC_HIP(hipMalloc(&p, nbytes ? nbytes : 1));
  const hipError_t e = hipMemsetAsync(p, 0, nbytes ? nbytes : 1, ctx->stream);
  if (e != hipSuccess) {
    (void)hipFree(p);
    return fail(MSC_EHIP, "hipMemsetAsync: %s", hipGetErrorString(e));
  }
  *out = p;
  return MSC_OK;
}

extern "C" int msc_device_alloc_probed(msc_context *ctx, size_t nbytes, uint32_t candidates, void **out,
                                       float *rates_gbps, uint32_t *chosen) {
  MSC_REQUIRE(ctx && out, "null argument");
  MSC_REQUIRE(candidates >= 1 && candidates <= 64, "candidates %u outside 1..64", candidates);
  *out = nullptr;
  MSC_HIP(hipSetDevice(ctx->device));
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(ctx->stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
    return fail(MSC_EINVAL, "msc_device_alloc_probed waits for the device: not on a capturing stream");
  return alloc_placed(ctx, nbytes, candidates, 1e30f, 0u, out, rates_gbps, chosen);     // every candidate is probed, the best kept
}
extern "C" int msc_device_alloc_stats(msc_context *ctx, float *rates_gbps, uint32_t capacity, uint32_t *ntried, uint32_t *chosen) {
  MSC_REQUIRE(ctx, "null context");
  const uint32_t n = (uint32_t)ctx->last_alloc_rates.size();
  if (rates_gbps)
    for (uint32_t i = 0; i < capacity && i < n; i++) rates_gbps[i] = ctx->last_alloc_rates[i];
  if (ntried) *ntried = n;
  if (chosen) *chosen = ctx->last_alloc_chosen;
  return MSC_OK;
}
extern "C" int msc_device_free(msc_context *ctx, void *dev) {
  MSC_REQUIRE(ctx, "null context");
  if (!dev) return MSC_OK;
  MSC_HIP(hipSetDevice(ctx->device));
  MSC_HIP(hipStreamSynchronize(ctx->stream));
  for (size_t i = 0; i < ctx->placed.size(); i++)
    if (ctx->placed[i].base == dev) {
      ctx->placed.erase(ctx->placed.begin() + i);
      break;
    }
  for (size_t i = 0; i < ctx->vmm.size(); i++)
    if (ctx->vmm[i].va == dev) {                          // a probed buffer mapped from chunks
      vmm_free(ctx->vmm[i]);
      ctx->vmm.erase(ctx->vmm.begin() + i);
      return MSC_OK;
    }
  MSC_HIP(hipFree(dev));
  return MSC_OK;
}
extern "C" int msc_pinned_alloc(msc_context *ctx, size_t nbytes, void **host, void **dev) {
  MSC_REQUIRE(ctx && host && dev, "null argument");
  *host = *dev = nullptr;
  MSC_HIP(hipSetDevice(ctx->device));
  void *h = nullptr, *d = nullptr;
  MSC_HIP(hipHostMalloc(&h, nbytes ? nbytes : 1, hipHostMallocMapped));
  const hipError_t e = hipHostGetDevicePointer(&d, h, 0);
  if (e != hipSuccess) {
    (void)hipHostFree(h);
    return fail(MSC_EHIP, "hipHostGetDevicePointer: %s", hipGetErrorString(e));
  }
  std::memset(h, 0, nbytes ? nbytes : 1);
  *host = h;
  *dev = d;
  return MSC_OK;
}
extern "C" int msc_pinned_free(msc_context *ctx, void *host) {
  MSC_REQUIRE(ctx, "null context");
  if (!host) return MSC_OK;
  MSC_HIP(hipSetDevice(ctx->device));
  MSC_HIP(hipStreamSynchronize(ctx->stream));
  MSC_HIP(hipHostFree(host));
  return MSC_OK;
}
// both copies are ordered on the context's stream and have completed when the call returns
extern "C" int msc_device_upload(msc_context *ctx, void *dst_dev, const void *src_host, size_t nbytes) {
  MSC_REQUIRE(ctx && (nbytes == 0 || (dst_dev && src_host)), "null argument");
  if (nbytes == 0) return MSC_OK;
  MSC_HIP(hipSetDevice(ctx->device));
  MSC_HIP(hipMemcpyAsync(dst_dev, src_host, nbytes, hipMemcpyHostToDevice, ctx->stream));
  MSC_HIP(hipStreamSynchronize(ctx->stream));
  return MSC_OK;
}
extern "C" int msc_device_download(msc_context *ctx, void *dst_host, const void *src_dev, size_t nbytes) {
  MSC_REQUIRE(ctx && (nbytes == 0 || (dst_host && src_dev)), "null argument");
  if (nbytes == 0) return MSC_OK;
  MSC_HIP(hipSetDevice(ctx->device));
  MSC_HIP(hipMemcpyAsync(dst_host, src_dev, nbytes, hipMemcpyDeviceToHost, ctx->stream));
  MSC_HIP(hipStreamSynchronize(ctx->stream));
  return device_error_check(ctx);
}

// ---------------------------------------------------------------------------
// dataview
// ---------------------------------------------------------------------------
static uint64_t next_view_serial() {
  static std::atomic<uint64_t> n{0};
  return ++n;
}
// The serials of the views that exist.  A state remembers the view it last bound by pointer AND serial and keeps no
// reference to it (the reference's dataviews are borrowed the same way, recarray/_dataview.pxd:24-27); before it looks at
// that view again outside a call that was handed one -- msc_state_set_hp re-plans, msc_sweep_* price kernels by its row
// count -- it asks here whether the view is still there (bound_view_of).  A destroyed view's serial never comes back, so
// a new view at the old address does not pass for it.
static std::mutex g_views_mu;
static std::vector<uint64_t> g_live_views;
static void view_register(uint64_t serial) {
  std::lock_guard<std::mutex> lock(g_views_mu);
  g_live_views.push_back(serial);
}
static void view_unregister(uint64_t serial) {
  std::lock_guard<std::mutex> lock(g_views_mu);
  g_live_views.erase(std::remove(g_live_views.begin(), g_live_views.end(), serial), g_live_views.end());
}
static bool view_alive(uint64_t serial) {
  std::lock_guard<std::mutex> lock(g_views_mu);
  return std::find(g_live_views.begin(), g_live_views.end(), serial) != g_live_views.end();
}
// the view the state last bound, if it still exists and is the one that was bound (msc_dataview_invalidate moves a
// view's serial on); otherwise the binding is dropped: no column pointer of it is dereferenced again, the next call
// that brings a view binds afresh
static const msc_dataview *bound_view_of(msc_state *st) {
  if (st->bound_view == nullptr) return nullptr;
  if (view_alive(st->bound_serial) && st->bound_view->serial == st->bound_serial) return st->bound_view;
  st->bound_view = nullptr;
  st->bound_serial = 0;
  st->bound_cols.clear();
  for (FeatDesc &d : st->desc_host) {
    d.col = nullptr;
    d.mask = nullptr;
    d.col_sentinel = nullptr;
    d.dm_tot = nullptr;
  }
  return nullptr;
}

static void free_all(std::vector<void *> &owned) {
  for (void *p : owned) (void)hipFree(p);
  owned.clear();
}

extern "C" int msc_dataview_from_records(msc_context *ctx, const void *host_records,
                                         const uint8_t *host_mask, uint64_t nrows,
                                         const msc_runtime_type *types, uint32_t ntypes,
                                         const int32_t *col_types, msc_dataview **out) {
  MSC_REQUIRE(ctx && out && types, "null argument");
  MSC_REQUIRE(ntypes > 0, "a dataview needs at least one feature");
  MSC_REQUIRE(nrows == 0 || host_records, "null records");
  *out = nullptr;
  MSC_HIP(hipSetDevice(ctx->device));
  // runtime_type::GetOffsetsAndSize (runtime_type.hpp:123-134)
  std::vector<UnpackFeat> uf(ntypes);
  size_t rowsize = 0, maskrowsize = 0;
  for (uint32_t i = 0; i < ntypes; i++) {
    MSC_REQUIRE(types[i].type >= 0 && types[i].type < MSC_TYPE_NELEMS, "feature %u: bad type %d", i,
                types[i].type);
    MSC_REQUIRE(types[i].count >= 1, "feature %u: count must be >= 1", i);
    const int dst = col_types ? col_types[i] : types[i].type;
    MSC_REQUIRE(dst >= 0 && dst < MSC_TYPE_NELEMS, "feature %u: bad column type %d", i, dst);
    uf[i].offset = (uint32_t)rowsize;
    uf[i].mask_offset = (uint32_t)maskrowsize;
    uf[i].src_type = types[i].type;
    uf[i].dst_type = dst;
    uf[i].count = types[i].count;
    rowsize += primitive_size(types[i].type) * types[i].count;
    maskrowsize += types[i].count;
  }
  std::unique_ptr<msc_dataview> v(new (std::nothrow) msc_dataview());
  if (!v) return fail(MSC_ENOMEM, "out of host memory");
  v->ctx = ctx;
  v->serial = next_view_serial();
  v->nrows = nrows;
  int rc = MSC_OK;
  uint8_t *rec_dev = nullptr, *mask_dev = nullptr;
  UnpackFeat *uf_dev = nullptr;
  std::vector<void *> scratch;
  auto cleanup = [&](int code) { free_all(scratch); if (code != MSC_OK) free_all(v->owned); return code; };
  for (uint32_t i = 0; i < ntypes; i++) {
    uint8_t *col = nullptr;
    rc = dev_alloc(v->owned, &col, (size_t)nrows * types[i].count * primitive_size(uf[i].dst_type));
    if (rc) return cleanup(rc);
    uf[i].dst = col;
    uf[i].dst_mask = nullptr;
    if (host_mask) {
      rc = dev_alloc(v->owned, &uf[i].dst_mask, (size_t)nrows * types[i].count);
      if (rc) return cleanup(rc);
    }
    v->cols.push_back(col);
    v->masks.push_back(uf[i].dst_mask);
    v->types.push_back(msc_runtime_type{uf[i].dst_type, types[i].count});
  }
  v->col_max.assign(ntypes, -1);
  v->dm_max.assign(ntypes, std::vector<uint32_t>());
  v->dm_tot.assign(ntypes, nullptr);
  if (nrows > 0) {
    if ((rc = dev_alloc(scratch, &rec_dev, (size_t)nrows * rowsize))) return cleanup(rc);
    if ((rc = dev_alloc(scratch, &uf_dev, ntypes))) return cleanup(rc);
    if (host_mask && (rc = dev_alloc(scratch, &mask_dev, (size_t)nrows * maskrowsize))) return cleanup(rc);
    hipError_t e = hipMemcpyAsync(rec_dev, host_records, (size_t)nrows * rowsize, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && host_mask)
      e = hipMemcpyAsync(mask_dev, host_mask, (size_t)nrows * maskrowsize, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
      e = hipMemcpyAsync(uf_dev, uf.data(), sizeof(UnpackFeat) * ntypes, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) return cleanup(fail(MSC_EHIP, "upload failed: %s", hipGetErrorString(e)));
    if (launch_unpack(ctx->stream, rec_dev, mask_dev, nrows, (uint32_t)rowsize, (uint32_t)maskrowsize, uf_dev, ntypes))
      return cleanup(fail(MSC_EHIP, "k_unpack launch failed: %s", hipGetErrorString(hipGetLastError())));
    e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return cleanup(fail(MSC_EHIP, "k_unpack failed: %s", hipGetErrorString(e)));
  }
  cleanup(MSC_OK);
  view_register(v->serial);
  *out = v.release();
  return MSC_OK;
}

extern "C" int msc_dataview_from_device_columns(msc_context *ctx, uint64_t nrows,
                                                const msc_runtime_type *types, uint32_t ntypes,
                                                void *const *dev_columns, void *const *dev_masks,
                                                msc_dataview **out) {
  MSC_REQUIRE(ctx && out && types && dev_columns, "null argument");
  MSC_REQUIRE(ntypes > 0, "a dataview needs at least one feature");
  *out = nullptr;
  std::unique_ptr<msc_dataview> v(new (std::nothrow) msc_dataview());
  if (!v) return fail(MSC_ENOMEM, "out of host memory");
  v->ctx = ctx;
  v->serial = next_view_serial();
  v->nrows = nrows;
  for (uint32_t i = 0; i < ntypes; i++) {
    MSC_REQUIRE(types[i].type >= 0 && types[i].type < MSC_TYPE_NELEMS, "feature %u: bad type", i);
    MSC_REQUIRE(types[i].count >= 1, "feature %u: count must be >= 1", i);
    MSC_REQUIRE(nrows == 0 || dev_columns[i], "feature %u: null column", i);
    MSC_REQUIRE(reinterpret_cast<uintptr_t>(dev_columns[i]) % primitive_size(types[i].type) == 0,
                "feature %u: column not aligned to its element size (%u bytes)", i, (unsigned)primitive_size(types[i].type));
    v->types.push_back(types[i]);
    v->cols.push_back(dev_columns[i]);
    v->masks.push_back(dev_masks ? dev_masks[i] : nullptr);
  }
  v->col_max.assign(ntypes, -1);
  v->dm_max.assign(ntypes, std::vector<uint32_t>());
  v->dm_tot.assign(ntypes, nullptr);
  view_register(v->serial);
  *out = v.release();
  return MSC_OK;
}

extern "C" int msc_dataview_destroy(msc_dataview *view) {
  if (!view) return MSC_OK;
  view_unregister(view->serial);                          // (states that had it bound drop the binding: bound_view_of)
  (void)hipSetDevice(view->ctx->device);
  (void)hipStreamSynchronize(view->ctx->stream);
  free_all(view->owned);
  free_all(view->owned_lazy);
  delete view;
  return MSC_OK;
}

// What the library derives from a view's columns and keeps WITH the view -- a column converted to a model's value type
// (column_as), a masked lookup column with the mask folded in (sentinel_column), bool columns packed four to a byte
// (packed_column), the maxima of count columns and dm row totals that size the exact tables -- is a SNAPSHOT of the
// columns' contents.  A view made by msc_dataview_from_records owns its columns and nothing else can write them; a view
// over the caller's device columns (msc_dataview_from_device_columns) must be told when the caller has rewritten them in
// place (the usual minibatch pattern): this call drops every derived copy and moves the view's serial on, so that every
// state binds -- and derives -- afresh at its next call.  Synchronises the context's stream.
extern "C" int msc_dataview_invalidate(msc_dataview *view) {
  MSC_REQUIRE(view, "null view");
  MSC_HIP(hipSetDevice(view->ctx->device));
  MSC_HIP(hipStreamSynchronize(view->ctx->stream));
  view_unregister(view->serial);
  free_all(view->owned_lazy);
  view->converted.clear();
  view->sentinels.clear();
  view->packed_bits.clear();
  view->nich_x.clear();
  view->look_idx.clear();
  view->col_max.assign(view->cols.size(), -1);
  view->dm_max.assign(view->cols.size(), std::vector<uint32_t>());
  view->dm_tot.assign(view->cols.size(), nullptr);
  view->serial = next_view_serial();
  view_register(view->serial);
  return MSC_OK;
}

extern "C" int msc_dataview_size(const msc_dataview *view, uint64_t *nrows, uint32_t *nfeatures) {
  MSC_REQUIRE(view, "null view");
  if (nrows) *nrows = view->nrows;
  if (nfeatures) *nfeatures = (uint32_t)view->types.size();
  return MSC_OK;
}

extern "C" int msc_dataview_column(const msc_dataview *view, uint32_t feature, void **dev_ptr,
                                   msc_runtime_type *type) {
  MSC_REQUIRE(view, "null view");
  MSC_REQUIRE(feature < view->types.size(), "feature %u out of range", feature);
  if (dev_ptr) *dev_ptr = view->cols[feature];
  if (type) *type = view->types[feature];
  return MSC_OK;
}

// ---------------------------------------------------------------------------
// state
// ---------------------------------------------------------------------------
extern "C" size_t msc_hp_floats(int family, uint32_t dim) {
  switch (family) {
    case MSC_BB: return 2;
    case MSC_BBNC: return 2;
    case MSC_GP: return 2;
    case MSC_DD: return dim;
    case MSC_NICH: return 4;
    case MSC_NIW: return 2 + (size_t)dim + (size_t)dim * dim;
    case MSC_BNB: return 3;
    case MSC_DM: return dim;
    default: return 0;
  }
}

extern "C" size_t msc_ss_bytes(int family, uint32_t dim) {
  switch (family) {
    case MSC_BB: return 8;
    case MSC_BBNC: return 12;
    case MSC_GP: return 12;
    case MSC_DD: return 4 * (1 + (size_t)dim);
    case MSC_NICH: return 12;
    case MSC_NIW: return 4 * (1 + (size_t)dim + (size_t)dim * dim);
    case MSC_BNB: return 8;
    case MSC_DM: return 4 * ((size_t)dim + 1);
    default: return 4;
  }
}

static void default_hp(int family, uint32_t dim, std::vector<float> &hp) {
  // microscopes/models.pyx:189,211,223,238,264-269
  hp.assign(msc_hp_floats(family, dim), 0.f);
  switch (family) {
    case MSC_BB: hp[0] = 1; hp[1] = 1; break;
    case MSC_BBNC: hp[0] = 1; hp[1] = 1; break;
    case MSC_GP: hp[0] = 1; hp[1] = 1; break;
    case MSC_DD: std::fill(hp.begin(), hp.end(), 1.f); break;
    case MSC_DM: std::fill(hp.begin(), hp.end(), 1.f); break;     // models.pyx:287 (dd's defaults)
    case MSC_BNB: hp[0] = 1; hp[1] = 1; hp[2] = 1; break;          // models.pyx:200
    case MSC_NICH: hp[0] = 0; hp[1] = 1; hp[2] = 1; hp[3] = 1; break;
    case MSC_NIW:
      hp[0] = 1; hp[1] = (float)dim;
      for (uint32_t i = 0; i < dim; i++) hp[2 + dim + (size_t)i * dim + i] = 1.f;
      break;
    default: break;
  }
}

// The plan of the tile kernels (score_block.hpp).  They walk a reordered copy of the descriptors: first every
// feature except the unmasked nich ones, in the caller's order, then the unmasked nich features, which get a phase
// (and an inner loop) of their own.  (Float addition order follows this plan, not the caller's feature order.)
// Consecutive features are packed into groups whose table blocks fit the LDS slot together: greedy, at most
// kGrpRows rows per group, never across the two phases.
// what the choice of kernels depends on, for one plan
struct PlanFacts {
  bool roles_ok = false, nich_only = false, lookups_only = false, tail_ok = false, tail_masked_nich = false, tail_dm = false;
  uint32_t tail_max_rows = 0, tail_pack_rows = 0;
};

// group packing, lookup kinds and runs of a plan `t` whose first `split` entries are the first phase (`extra`: one more
// table row for a masked lookup column's zero row)
static PlanFacts plan_layout(std::vector<FeatDesc> &t, uint32_t split, const std::vector<uint32_t> &extra) {
  auto rows_of = [](const FeatDesc &d) -> uint32_t {
    if (d.fuse_n >= 2) {
      uint32_t rows = 1;
      for (uint32_t j = 0; j < d.fuse_n; j++) rows *= d.fuse_radix;
      return rows;
    }
    switch (d.family) {
      case MSC_BB:
      case MSC_BBNC: return 2;
      case MSC_NICH: return 6;
      // (a table up to the whole slot is staged -- on its own group if need be: a dd feature of 100 categories with 64 of
      // them staged was a GENERIC feature to every tile kernel, 1.03 ms for four of them beside two nich columns on a
      // million rows whatever K <= 256 is; staged whole they are lookup runs like any other)
      case MSC_DD: return std::min<uint32_t>(d.dim, (uint32_t)kGrpRows);
      case MSC_GP:
      case MSC_BNB: return std::min<uint32_t>(d.vcap, (uint32_t)kGrpRows);
      // dm: all dim + 1 tables or none (small counts: tens of rows).  Read from L2 they cost 2 KiB per row and
      // stage -- 4 x dm(4) on 1M rows ran at the L2's bandwidth, 1.9 ms
      case MSC_DM: return d.dm_meta != nullptr && d.dm_rows <= (uint32_t)kGrpRows ? d.dm_rows : 0;
      default: return 0;
    }
  };
  const uint32_t n = (uint32_t)t.size();
  uint32_t f = 0;
  while (f < n) {
    const uint32_t limit = f < split ? split : n;
    uint32_t used = 0, g = f;
    while (g < limit && used + rows_of(t[g]) + extra[g] <= (uint32_t)kGrpRows) {
      // (a nich block is staged whole: its members' constants are read side by side)
      if (t[g].blk_end > g + 1 && t[g].blk_first == g && used > 0 && used + 6u * (t[g].blk_end - g) > (uint32_t)kGrpRows) break;
      t[g].grp_off = used;
      t[g].grp_rows = rows_of(t[g]) + extra[g];
      used += t[g].grp_rows;
      g++;
    }
    if (g == f) {                                           // (a block larger than the slot: a group of its own, nothing staged)
      t[g].grp_off = 0;
      t[g].grp_rows = 0;
      g++;
    }
    for (uint32_t i = f; i < g; i++) t[i].grp_end = g;
    f = g;
  }
  // Unmasked lookup features whose whole table is staged (so that no row can miss it) take the tight inner
  // loop of the first phase; run_end lets a wave stay in it for a whole run of them.
  for (uint32_t i = 0; i < n; i++) {
    FeatDesc &d = t[i];
    d.kind = MSC_KIND_GENERIC;
    if (d.mask != nullptr || d.col == nullptr || d.grp_rows == 0) continue;
    // (run_clamp: the largest row a value may select -- with the mask folded in that is the zero row)
    if (d.fuse_n >= 2) d.kind = MSC_KIND_LOOKUP_U8, d.run_clamp = d.grp_rows - 1;
    else if (d.family == MSC_BB || d.family == MSC_BBNC) d.kind = MSC_KIND_LOOKUP_U8, d.run_clamp = 1 + extra[i];
    else if ((d.family == MSC_GP || d.family == MSC_BNB) && d.grp_rows >= d.vcap + extra[i]) d.kind = MSC_KIND_LOOKUP_U32, d.run_clamp = d.grp_rows - 1;
    else if (d.family == MSC_DD && d.grp_rows >= d.dim + extra[i]) d.kind = MSC_KIND_LOOKUP_I32, d.run_clamp = d.dim - 1 + extra[i];
  }
  PlanFacts pf;
  bool has_dm = false;
  pf.roles_ok = split > 0 && split < n;
  for (uint32_t i = 0; i < n; i++) {
    has_dm |= t[i].family == MSC_DM;
    if (i < split && t[i].kind == MSC_KIND_GENERIC) pf.roles_ok = false;
  }
  if (has_dm) pf.roles_ok = false;
  // the kernels whose waves are all nich waves (k_score_nich_pack): no first phase at all and two or more plain nich features,
  // or a first phase of at most kPackMaxLookups lookup features beside at least twice as many nich features (the lookups
  // are gathered from L2 there, ~0.03 ms a feature and million rows: 8 bb + 8 nich sweep 0.81 -> 0.70 ms, 2 gp + 12 nich
  // 0.99 -> 0.86; with 16 bb + 4 nich -- four fused lookups, four nich -- the role-split kernels are as good or better)
  pf.nich_only = (split == 0 && n >= 2) || (pf.roles_ok && !has_dm && split <= (uint32_t)kPackMaxLookups && n - split >= 2 * split &&
                                            std::getenv("MSC_NO_PACK_LOOKUPS") == nullptr);
  if (pf.nich_only) pf.roles_ok = false;
  // staged lookup features and nothing else: k_score_lookups / k_sweep_lookups (sixteen lookup waves of 16 sums)
  pf.lookups_only = split == n && n > 0 && !has_dm && std::getenv("MSC_NO_LOOKUPS_KERNEL") == nullptr;
  for (uint32_t i = 0; i < n; i++) pf.lookups_only &= t[i].kind != MSC_KIND_GENERIC;
  // the lane <-> row kernel for a partly filled last tile (k_score_tail_rows): lookup features only in the first phase
  // (what it implements), whatever the second holds of plain nich features
  pf.tail_ok = std::getenv("MSC_NO_NARROW_TAIL") == nullptr;
  // (a masked nich column among them is evaluated like the second phase's features, under the row's mask; a dm feature
  // whose dim + 1 tables are staged whole -- small counts: no row of the bound column is beyond them -- is dim + 1 lookups
  // of (hi, lo) pairs, round 5; the kernel has one instantiation for either, none for both)
  for (uint32_t i = 0; i < split; i++) {
    const bool masked_nich = t[i].family == MSC_NICH && t[i].mask != nullptr && t[i].col != nullptr;
    const bool staged_dm = t[i].family == MSC_DM && t[i].col != nullptr && t[i].dm_meta != nullptr && t[i].grp_rows != 0;
    pf.tail_ok &= t[i].kind != MSC_KIND_GENERIC || masked_nich || staged_dm;
    pf.tail_masked_nich |= masked_nich;
    pf.tail_dm |= staged_dm;
  }
  if (pf.tail_dm && pf.tail_masked_nich) pf.tail_ok = false;
  for (uint32_t i = split; i < n; i++) pf.tail_ok &= t[i].family == MSC_NICH && t[i].mask == nullptr && t[i].grp_rows == 6;
  if (!pf.tail_ok) pf.tail_masked_nich = pf.tail_dm = false;
  if (pf.tail_ok)
    for (uint32_t i = 0; i < split; i++) {
      const uint32_t rows = t[i].family == MSC_DM ? t[i].dm_rows : t[i].kind == MSC_KIND_GENERIC ? 0u : t[i].run_clamp + 1;
      pf.tail_max_rows = std::max(pf.tail_max_rows, rows);
      pf.tail_pack_rows += rows;
    }
  for (uint32_t i = n; i-- > 0;) {
    FeatDesc &d = t[i];
    if (d.kind == MSC_KIND_GENERIC) d.run_end = i;
    else d.run_end = (i + 1 < d.grp_end && t[i + 1].kind != MSC_KIND_GENERIC) ? t[i + 1].run_end : i + 1;
  }
  return pf;
}

// the byte column holding the bits of m bool columns (kept by the view: every state that fuses the same columns shares it)
static int packed_column(const msc_dataview *view, const void *const *cols, int m, uint32_t radix, const void **out) {
  std::vector<const void *> key(cols, cols + m);
  key.push_back(reinterpret_cast<const void *>((uintptr_t)radix));
  for (const auto &e : view->packed_bits)
    if (e.first == key) {
      *out = e.second;
      return MSC_OK;
    }
  void *dst = nullptr;
  MSC_HIP(hipMalloc(&dst, std::max<size_t>(1, (size_t)view->nrows)));
  view->owned_lazy.push_back(dst);
  if (launch_pack_bits(view->ctx->stream, cols, m, radix, view->nrows, dst)) return fail(MSC_EHIP, "k_pack_bits launch failed");
  view->packed_bits.emplace_back(key, dst);
  *out = dst;
  return MSC_OK;
}

// the float columns `cols` (n2 of them; device, one float a row) as one matrix float [nrows][n2p], n2p = n2 rounded up to
// four and the padding zero: what the role-split kernels' nich waves read a row's second-phase values from (msc::NichPos).
// Kept by the view like the packed bool columns; a copy of those columns (64 MB for C3's sixteen on a million rows).
// A view keeps at most kViewPackCap matrices of either kind (one per distinct column list a state bound it with); a
// plan that would need one more, or whose allocation fails, does without -- the kernels that run the phases one after
// the other need neither (the copies are an optimisation, never a reason for a call to fail).
constexpr size_t kViewPackCap = 8;
static bool nich_x_matrix(const msc_dataview *view, const std::vector<const void *> &cols, const float **out) {
  for (const auto &e : view->nich_x)
    if (e.first == cols) {
      *out = e.second;
      return true;
    }
  if (view->nich_x.size() >= kViewPackCap) return false;
  const uint32_t n2 = (uint32_t)cols.size(), n2p = (n2 + 3u) & ~3u;
  void *dst = nullptr, *ptrs = nullptr;
  if (hipMalloc(&dst, std::max<size_t>(16, (size_t)view->nrows * n2p * sizeof(float))) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  bool ok = hipMalloc(&ptrs, sizeof(void *) * n2) == hipSuccess;
  ok = ok && hipMemcpyAsync(ptrs, cols.data(), sizeof(void *) * n2, hipMemcpyHostToDevice, view->ctx->stream) == hipSuccess;
  ok = ok && launch_pack_nich_x(view->ctx->stream, static_cast<const float *const *>(ptrs), n2, n2p, view->nrows, static_cast<float *>(dst)) == 0;
  ok = (hipStreamSynchronize(view->ctx->stream) == hipSuccess) && ok;      // (`cols` is the caller's, `ptrs` this function's)
  if (ptrs != nullptr) (void)hipFree(ptrs);
  if (!ok) {
    (void)hipGetLastError();
    (void)hipFree(dst);
    return false;
  }
  view->owned_lazy.push_back(dst);
  view->nich_x.emplace_back(cols, static_cast<const float *>(dst));
  *out = static_cast<const float *>(dst);
  return true;
}

// The lookup index matrix of a plan's first phase tf[0 .. split) (msc::FeatDesc::lk_idx): a row's record is the groups'
// features side by side, a byte each, every group from a dword boundary on; sets lk_goff at every group's first feature.
// Kept by the view like the x matrix (C3: 36 features in ten groups, 48 bytes a row).  false: no matrix (cap reached, or no
// memory) -- the plan does not take the kernels that read one.
static bool look_idx_matrix(const msc_dataview *view, std::vector<FeatDesc> &tf, uint32_t split, const uint32_t **out, uint32_t *l4_out) {
  std::vector<LookIdxSrc> src(split);
  std::vector<uint64_t> key;
  uint32_t l4 = 0;
  for (uint32_t f0 = 0; f0 < split;) {
    const uint32_t f1 = tf[f0].grp_end;
    tf[f0].lk_goff = l4;
    for (uint32_t i = f0; i < f1; i++) {
      const FeatDesc &d = tf[i];
      if (d.kind == MSC_KIND_GENERIC || d.col == nullptr || d.grp_off + d.run_clamp >= 256u) return false;
      src[i] = LookIdxSrc{d.col, d.kind, d.run_clamp, d.grp_off, 4u * l4 + (i - f0)};
      key.push_back(reinterpret_cast<uint64_t>(d.col));
      key.push_back((uint64_t)d.kind | (uint64_t)d.run_clamp << 8 | (uint64_t)d.grp_off << 32);
      key.push_back(src[i].byte_at);
    }
    l4 += (f1 - f0 + 3u) / 4u;
    f0 = f1;
  }
  *l4_out = l4;
  for (const auto &e : view->look_idx)
    if (e.first == key) {
      *out = e.second;
      return true;
    }
  if (view->look_idx.size() >= kViewPackCap || split == 0) return false;
  void *dst = nullptr, *srcs = nullptr;
  if (hipMalloc(&dst, ((size_t)view->nrows * l4 + 4u) * sizeof(uint32_t)) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  bool ok = hipMalloc(&srcs, sizeof(LookIdxSrc) * split) == hipSuccess;
  ok = ok && hipMemcpyAsync(srcs, src.data(), sizeof(LookIdxSrc) * split, hipMemcpyHostToDevice, view->ctx->stream) == hipSuccess;
  ok = ok && hipMemsetAsync(static_cast<uint32_t *>(dst) + (size_t)view->nrows * l4, 0, 4u * sizeof(uint32_t), view->ctx->stream) == hipSuccess;
  ok = ok && launch_pack_look_idx(view->ctx->stream, static_cast<const LookIdxSrc *>(srcs), split, l4, view->nrows, static_cast<uint32_t *>(dst)) == 0;
  ok = (hipStreamSynchronize(view->ctx->stream) == hipSuccess) && ok;
  if (srcs != nullptr) (void)hipFree(srcs);
  if (!ok) {
    (void)hipGetLastError();
    (void)hipFree(dst);
    return false;
  }
  view->owned_lazy.push_back(dst);
  view->look_idx.emplace_back(key, static_cast<const uint32_t *>(dst));
  *out = static_cast<const uint32_t *>(dst);
  return true;
}

static int plan_groups(msc_state *st) {
  const msc_dataview *const bview = bound_view_of(st);    // (null: no column pointer survives in desc_host either)
  auto nich_tail = [](const FeatDesc &d) { return d.family == MSC_NICH && d.mask == nullptr && d.col != nullptr; };
  std::vector<FeatDesc> &t = st->desc_tile_host;
  t.clear();
  for (FeatDesc &d : st->desc_host) d.blk_first = d.blk_end = 0, d.nich_info = nullptr;
  std::vector<uint32_t> t_src;                              // t[i] is the caller's feature t_src[i]
  for (uint32_t i = 0; i < st->nfeat; i++)
    if (!nich_tail(st->desc_host[i])) t.push_back(st->desc_host[i]), t_src.push_back(i);
  st->tile_split = (uint32_t)t.size();
  // The second phase, in BLOCKS (family_math.hpp "nich BLOCKS"): the plain nich features ordered by their nu prior
  // (stable: the caller's order among equals), runs of an equal nu cut into blocks of at most kNichBlock, as even as they
  // come (five are 3 + 2, not 4 + 1: a feature on its own pays the full logarithm).  Whether a block's c1 really are equal
  // is for the head kernel to say (the counts are the suff-stats', not the plan's).
  {
    std::vector<uint32_t> tail;
    for (uint32_t i = 0; i < st->nfeat; i++) if (nich_tail(st->desc_host[i])) tail.push_back(i);
    auto nu_bits = [&](uint32_t i) { uint32_t b; std::memcpy(&b, &st->feats[i].hp[3], 4); return b; };
    static const bool no_blocks = std::getenv("MSC_NO_NICH_BLOCKS") != nullptr;       // (A/B knob: every feature on its own)
    std::stable_sort(tail.begin(), tail.end(), [&](uint32_t a, uint32_t b) { return nu_bits(a) < nu_bits(b); });
    st->nich_blocks_any = false;
    for (size_t a = 0; a < tail.size();) {
      size_t b = a + 1;
      while (b < tail.size() && nu_bits(tail[b]) == nu_bits(tail[a])) b++;
      const size_t n = b - a, nb = no_blocks ? n : (n + kNichBlock - 1) / kNichBlock, base = n / nb, rem = n % nb;
      for (size_t blk = 0, at = a; blk < nb; blk++) {
        const size_t len = base + (blk < rem ? 1 : 0);
        for (size_t j = 0; j < len; j++) {
          FeatDesc d = st->desc_host[tail[at + j]];
          d.blk_first = (uint32_t)(t.size() - j);
          d.blk_end = d.blk_first + (uint32_t)len;
          d.nich_info = st->nich_info + tail[at + j];
          t.push_back(d);
          t_src.push_back(tail[at + j]);
        }
        st->nich_blocks_any |= len >= 2;
        at += len;
      }
      a = b;
    }
  }
  // masked lookup columns: the tile plan reads the copy with the mask folded in (bind_view) and sees no mask -- a masked
  // value selects the table's zero row, which the feature stages with the others (one more row: `extra`)
  std::vector<uint32_t> extra(t.size(), 0u);
  for (size_t i = 0; i < t.size(); i++) {
    t[i].fuse_n = 0;
    if (t[i].mask != nullptr && t[i].col_sentinel != nullptr) {
      t[i].col = t[i].col_sentinel;
      t[i].mask = nullptr;
      extra[i] = 1;
    }
  }
  const uint32_t n = st->nfeat, split = st->tile_split;
  PlanFacts facts = plan_layout(t, split, extra);
  // The leave-one-out pass (k_loo_own_lds) stages what a row gathers per feature -- the lookup families' "value against
  // the group minus one" tables, nich's twelve doubles per group -- for ALL kpad groups (a row's own group is any of
  // them): consecutive features share the 64 KiB slot while their blocks fit; a feature whose block does not fit, or
  // that is not of a staged kind, reads global memory inside its stage.
  st->loo_staged = 0;
  for (uint32_t i = 0; i < n; i++) {
    FeatDesc &d = t[i];
    d.loo_off = d.loo_rows = 0;
    uint32_t rows = 0;
    if (d.kind != MSC_KIND_GENERIC && (d.loo_tab != nullptr || d.family == MSC_BBNC)) rows = d.run_clamp + 1;
    else if (d.family == MSC_NICH && d.mask == nullptr && d.col != nullptr && d.loo64 != nullptr) rows = 2 * kNlooStride;
    if (rows != 0 && (uint64_t)rows * st->kpad <= kLooSlotFloats) d.loo_rows = rows;
  }
  for (uint32_t f0 = 0; f0 < n;) {
    uint32_t used = 0, g = f0;
    while (g < n && g - f0 < (uint32_t)kLooStageFeats && used + t[g].loo_rows * st->kpad <= kLooSlotFloats) {
      t[g].loo_off = used;
      used += t[g].loo_rows * st->kpad;
      st->loo_staged += t[g].loo_rows != 0;
      g++;
    }
    for (uint32_t i = f0; i < g; i++) t[i].loo_stage_end = g;
    f0 = g;
  }
  // The score / sweep kernels' plan: the unmasked bb / bbnc columns of the first phase come first, fused four (three, two)
  // at a time -- one byte column of the members' bits, one table of 2^m rows (FeatDesc::fuse_*; k_fuse_tables fills the
  // tables at the head of a call) -- then the other first-phase features in the caller's order, then the second phase.
  // A quarter of the LDS reads and additions for the same sum; the order and the association of the terms --
  // ((t0 + t1) + t2) + t3 per fused feature -- are the same in every kernel that walks this plan.  The leave-one-out
  // pass walks the plan above.
  std::vector<FeatDesc> &tf = st->desc_fuse_host;
  std::vector<uint32_t> extra_f;
  tf.clear();
  // (the view is only looked at while it exists: a plan made after its view is gone -- msc_state_set_hp on a dd feature
  // re-plans -- fuses nothing and holds no column; the next call that brings a view binds and plans again)
  if (!st->nich_blocks_any)                                     // (no block of two or more: no records, no head-kernel work, no far rows)
    for (FeatDesc &d : t) d.nich_info = nullptr;
  const bool may_fuse = bview != nullptr && std::getenv("MSC_NO_BB_FUSE") == nullptr;
  // (columns with a mask: their mask-folded copies, three states a value -- 0, 1, masked = the member's zero row --,
  // three at a time against 27 rows)
  std::vector<uint32_t> members[2];                        // [0] unmasked, [1] masked: the fusable features, in plan order
  for (uint32_t i = 0; i < split; i++) {
    const FeatDesc &d = t[i];
    if (may_fuse && (d.family == MSC_BB || d.family == MSC_BBNC) && d.kind == MSC_KIND_LOOKUP_U8 && d.mask == nullptr &&
        d.col != nullptr && d.tab != nullptr)
      members[extra[i] ? 1 : 0].push_back(i);
  }
  struct Fused { uint32_t first, m, radix; };               // (first: index into members[radix - 2])
  std::vector<Fused> quads;
  for (uint32_t cls = 0; cls < 2; cls++) {
    const size_t widest = cls == 0 ? 4 : 3;
    size_t at = 0;
    for (size_t left = members[cls].size(); left >= 2;) {
      const uint32_t m = left == widest + 1 ? (uint32_t)widest - 1 : (uint32_t)std::min(widest, left);    // (never a single one left over)
      quads.push_back(Fused{(uint32_t)at, m, 2u + cls});
      at += m;
      left -= m;
    }
  }
  const size_t need = quads.size() * 32 * (size_t)st->kpad;
  if (st->fuse_tab_floats < need) {
    void *p = nullptr;
    MSC_HIP(hipMalloc(&p, need * sizeof(float)));
    st->owned.push_back(p);
    st->fuse_tab = static_cast<float *>(p);
    st->fuse_tab_floats = need;
  }
  std::vector<bool> taken(n, false);
  for (size_t q = 0; q < quads.size(); q++) {
    const Fused &fq = quads[q];
    const std::vector<uint32_t> &mem = members[fq.radix - 2];
    FeatDesc d = t[mem[fq.first]];
    const void *cols[4] = {nullptr, nullptr, nullptr, nullptr};
    for (uint32_t j = 0; j < 4; j++) d.fuse_src[j] = nullptr;
    for (uint32_t j = 0; j < fq.m; j++) {
      const FeatDesc &mj = t[mem[fq.first + j]];
      cols[j] = mj.col;
      d.fuse_src[j] = mj.tab;
      taken[mem[fq.first + j]] = true;
    }
    MSC_TRY(packed_column(bview, cols, (int)fq.m, fq.radix, &d.col));
    d.fuse_n = fq.m;
    d.fuse_radix = fq.radix;
    d.col_type = MSC_TYPE_U8;
    d.family = MSC_BB;
    d.tab = st->fuse_tab + q * 32 * (size_t)st->kpad;
    d.loo_tab = nullptr;
    tf.push_back(d);
    extra_f.push_back(0u);
  }
  for (uint32_t i = 0; i < n; i++)
    if (!taken[i]) {
      tf.push_back(t[i]);
      extra_f.push_back(extra[i]);
    }
  // the accumulate pass's list: a fused feature reads z and its byte column once for all its members (their additive
  // tables: fuse_acc); everything else as the caller gave it.  (Histograms of 2 m K counters: while they fit LDS.)
  {
    std::vector<FeatDesc> &ta = st->desc_acc_host;
    ta.clear();
    std::vector<bool> covered(n, false);
    for (size_t q = 0; q < quads.size(); q++) {
      const Fused &fq = quads[q];
      if ((size_t)st->K * 8u * fq.m * 4u > 64u * 1024u) continue;
      const std::vector<uint32_t> &mem = members[fq.radix - 2];
      FeatDesc d = tf[q];
      for (uint32_t j = 0; j < 4; j++) d.fuse_acc[j] = nullptr;
      for (uint32_t j = 0; j < fq.m; j++) {
        const uint32_t src = t_src[mem[fq.first + j]];
        d.fuse_acc[j] = st->desc_host[src].acc_i64;
        covered[src] = true;
      }
      ta.push_back(d);
    }
    for (uint32_t i = 0; i < n; i++)
      if (!covered[i]) ta.push_back(st->desc_host[i]);
  }
  const std::vector<Fused> &fused = quads;
  st->fuse_any = !fused.empty();
  st->fuse_nfeat = (uint32_t)tf.size();
  st->fuse_split = split - (n - st->fuse_nfeat);
  for (uint32_t i = st->fuse_split; i < st->fuse_nfeat; i++) {     // (block bounds are indices into the plan they sit in)
    tf[i].blk_first -= n - st->fuse_nfeat;
    tf[i].blk_end -= n - st->fuse_nfeat;
  }
  if (st->fuse_any) facts = plan_layout(tf, st->fuse_split, extra_f);
  // what the role-split kernels' nich waves read (msc::NichPos): positions, the pack (k_fuse_tables fills it at the head
  // of every call), the x matrix -- the last needs the view; without it (a plan made while no view is bound) the plan
  // does not take those kernels, and the next call with a view plans again
  for (FeatDesc &d : tf) d.rn_pack = nullptr, d.rn_pos = nullptr, d.rn_x = nullptr, d.rn_n2 = d.rn_n2p = 0;
  for (FeatDesc &d : t) d.rn_pack = nullptr, d.rn_pos = nullptr, d.rn_x = nullptr, d.rn_n2 = d.rn_n2p = 0;
  static const bool no_roles_pack = std::getenv("MSC_NO_ROLES") != nullptr;      // (A/B knob: the kernels that run the phases one after the other)
  if (bview == nullptr || no_roles_pack) facts.roles_ok = facts.nich_only = false;
  if (facts.roles_ok || facts.nich_only) {
    const uint32_t s0 = st->fuse_split, n2 = st->fuse_nfeat - s0, n2p = (n2 + 3u) & ~3u;
    std::vector<NichPos> pos(n2p);
    std::vector<const void *> xcols(n2);
    for (uint32_t i = 0; i < n2p; i++) {
      NichPos &q = pos[i];
      q.xlim = INFINITY, q.blk_ok = 0u;                       // (the head kernel writes these two for real positions)
      if (i < n2) {
        const FeatDesc &d = tf[s0 + i];
        q.blk = (d.blk_first - s0) | (d.blk_end - d.blk_first) << 16, q.seg_end = d.grp_end - s0;
        xcols[i] = d.col;
      } else q.blk = i | 1u << 16, q.seg_end = n2p;
    }
    const float *xm = nullptr;
    if (nich_x_matrix(bview, xcols, &xm)) {
      MSC_HIP(hipMemcpyAsync(st->rn_pos, pos.data(), sizeof(NichPos) * n2p, hipMemcpyHostToDevice, st->ctx->stream));
      MSC_HIP(hipStreamSynchronize(st->ctx->stream));           // (`pos` is this function's)
      FeatDesc &h = tf[s0];
      h.rn_pack = st->rn_pack, h.rn_pos = st->rn_pos, h.rn_x = xm, h.rn_n2 = n2, h.rn_n2p = n2p;
    } else facts.roles_ok = facts.nich_only = false;           // (no copy: the kernels that stage the nich constants need none)
  }
  // ... and what their lookup waves read (FeatDesc::lk_idx): the first phase's slot rows, row by row
  for (FeatDesc &d : tf) d.lk_idx = nullptr, d.lk_l4 = d.lk_goff = 0;
  for (FeatDesc &d : t) d.lk_idx = nullptr, d.lk_l4 = d.lk_goff = 0;
  if (bview == nullptr) facts.lookups_only = false;
  if (facts.roles_ok || facts.lookups_only) {
    const uint32_t *im = nullptr;
    uint32_t l4 = 0;
    if (look_idx_matrix(bview, tf, st->fuse_split, &im, &l4)) tf[0].lk_idx = im, tf[0].lk_l4 = l4;
    else facts.roles_ok = facts.lookups_only = false;
  }
  st->tile_roles_ok = facts.roles_ok;
  st->tile_nich_only = facts.nich_only;
  st->tile_lookups_only = facts.lookups_only;
  {
    // the prices the kernels are chosen by (launchers.hpp PlanCost): staged lookup features and table rows of the first
    // phase, nich features (the second phase's, and masked ones evaluated in the first)
    double lookups = 0, rows = 0, nich = (double)(st->fuse_nfeat - st->fuse_split);
    for (uint32_t i = 0; i < st->fuse_split; i++) {
      if (tf[i].family == MSC_NICH) nich += 1;
      else if (tf[i].family == MSC_DM) lookups += 2.0 * (tf[i].dim + 1), rows += tf[i].grp_rows;   // (dim + 1 stages of (hi, lo) pairs)
      else lookups += 1, rows += tf[i].grp_rows;
    }
    PlanCost pc;
    const double first = 6.0 + 0.45 * lookups + 0.012 * rows;
    pc.tile_round_us = facts.nich_only ? 6.0 + 1.3 * nich + 3.0 * lookups : facts.roles_ok ? first + 0.75 * nich : facts.lookups_only ? 0.75 * first : first + 1.5 * nich;
    pc.sweep_round_us = pc.tile_round_us + 2.5;              // (+ the draws)
    pc.tail_fixed_us = 2.0;                                  // (per launch and round of 1024 rows a CU: tools/scans/grid_scan.py at 1M rows,
    pc.tail_group_us = 0.5 + 0.03 * lookups + 0.057 * nich;  //  K = 8 against K = 32 for six feature lists; C3: 2 + 2.5 a group)
    st->plan_cost = pc;
  }
  st->tile_narrow_tail_ok = facts.tail_ok;
  st->tail_masked_nich = facts.tail_masked_nich;
  st->tail_dm = facts.tail_dm;
  st->tail_max_rows = facts.tail_max_rows;
  st->tail_pack_rows = facts.tail_pack_rows;
  return MSC_OK;
}

static int upload_desc(msc_state *st) {
  MSC_TRY(plan_groups(st));
  MSC_HIP(hipMemcpyAsync(st->desc_dev, st->desc_host.data(), sizeof(FeatDesc) * st->nfeat,
                         hipMemcpyHostToDevice, st->ctx->stream));
  MSC_HIP(hipMemcpyAsync(st->desc_tile_dev, st->desc_tile_host.data(), sizeof(FeatDesc) * st->nfeat,
                         hipMemcpyHostToDevice, st->ctx->stream));
  MSC_HIP(hipMemcpyAsync(st->desc_fuse_dev, st->desc_fuse_host.data(), sizeof(FeatDesc) * st->fuse_nfeat,
                         hipMemcpyHostToDevice, st->ctx->stream));
  MSC_HIP(hipMemcpyAsync(st->desc_acc_dev, st->desc_acc_host.data(), sizeof(FeatDesc) * st->desc_acc_host.size(),
                         hipMemcpyHostToDevice, st->ctx->stream));
  MSC_HIP(hipStreamSynchronize(st->ctx->stream));     // (the tile copy is rebuilt by the next call)
  return MSC_OK;
}

extern "C" int msc_state_create(msc_context *ctx, const msc_feature_spec *features,
                                uint32_t nfeatures, uint32_t ngroups, msc_state **out) {
  MSC_REQUIRE(ctx && features && out, "null argument");
  MSC_REQUIRE(nfeatures > 0 && nfeatures <= 65535, "a state needs 1..65535 features (got %u)", nfeatures);
  MSC_REQUIRE(ngroups > 0 && ngroups <= (1u << 20), "ngroups %u out of range", ngroups);
  *out = nullptr;
  MSC_HIP(hipSetDevice(ctx->device));
  std::unique_ptr<msc_state> st(new (std::nothrow) msc_state());
  if (!st) return fail(MSC_ENOMEM, "out of host memory");
  st->ctx = ctx;
  st->nfeat = nfeatures;
  st->K = ngroups;
  st->kpad = round_up(ngroups, kGroupTile);
  const size_t kpad = st->kpad;
  st->feats.resize(nfeatures);
  st->desc_host.resize(nfeatures);
  size_t n_i64 = kpad, n_f64 = 0;
  for (uint32_t f = 0; f < nfeatures; f++) {
    msc_feature_host &h = st->feats[f];
    h.family = features[f].family;
    h.dim = features[f].dim;
    MSC_REQUIRE(family_ok(h.family), "feature %u: unknown family %d", f, h.family);
    if (h.family == MSC_DD && (h.dim == 0 || h.dim > kMaxDDDim))
      return fail(MSC_EUNSUPPORTED, "feature %u: dd dim %u outside 1..%u (DirichletDiscrete<128>)", f,
                  h.dim, kMaxDDDim);
    if (h.family == MSC_DM && (h.dim == 0 || h.dim > kMaxDDDim))
      return fail(MSC_EUNSUPPORTED, "feature %u: dm categories %u outside 1..%u", f, h.dim, kMaxDDDim);
    if (h.family == MSC_NIW && (h.dim == 0 || h.dim > kMaxNiwDim))
      return fail(MSC_EUNSUPPORTED, "feature %u: niw dim %u outside 1..%u", f, h.dim, kMaxNiwDim);
    h.i64_off = n_i64;
    h.i64_len = acc_i64_rows(h.family, h.dim) * kpad;
    n_i64 += h.i64_len;
    h.f64_off = n_f64;
    h.f64_len = h.family == MSC_NIW ? (size_t)ngroups * (h.dim + (size_t)h.dim * h.dim)
                                    : acc_f64_rows(h.family) * kpad;
    n_f64 += h.f64_len;
  }
  st->n_i64 = n_i64;
  st->n_f64 = n_f64;
  int rc;
  auto bail = [&](int code) { free_all(st->owned); return code; };
  if ((rc = dev_alloc(st->owned, &st->red_i64, n_i64))) return bail(rc);
  if ((rc = dev_alloc(st->owned, &st->red_f64, n_f64))) return bail(rc);
  if ((rc = dev_alloc(st->owned, &st->cnt_u32, kpad))) return bail(rc);
  if ((rc = dev_alloc(st->owned, &st->logpc, crp_floats(st->kpad)))) return bail(rc);
  if ((rc = dev_alloc(st->owned, &st->desc_dev, nfeatures))) return bail(rc);
  if ((rc = dev_alloc(st->owned, &st->desc_tile_dev, nfeatures))) return bail(rc);
  if ((rc = dev_alloc(st->owned, &st->desc_fuse_dev, nfeatures))) return bail(rc);
  if ((rc = dev_alloc(st->owned, &st->nich_info, nfeatures))) return bail(rc);
  if ((rc = dev_alloc(st->owned, &st->rn_pos, (size_t)((nfeatures + 3u) & ~3u)))) return bail(rc);
  {
    size_t nnich = 0;
    for (uint32_t f = 0; f < nfeatures; f++) nnich += features[f].family == MSC_NICH;
    if ((rc = dev_alloc(st->owned, &st->rn_pack, (1 + kNichPackRows * nnich) * (size_t)kpad))) return bail(rc);
  }
  if ((rc = dev_alloc(st->owned, &st->desc_acc_dev, nfeatures))) return bail(rc);
  if ((rc = dev_alloc(st->owned, &st->rng_dev, 2))) return bail(rc);
  if ((rc = dev_alloc(st->owned, &st->colmax_dev, 1))) return bail(rc);
  for (uint32_t f = 0; f < nfeatures; f++) {
    msc_feature_host &h = st->feats[f];
    default_hp(h.family, h.dim, h.hp);
    if ((rc = dev_alloc(st->owned, &h.hp_dev, h.hp.size()))) return bail(rc);
    if ((rc = dev_alloc(st->owned, &h.tab, (size_t)(tab_rows(h.family, h.dim) + 4) * kpad))) return bail(rc);   // +4: stage_table1 copies 4 rows per instruction
    if ((rc = dev_alloc(st->owned, &h.raw_u32, (size_t)raw_u32_rows(h.family, h.dim) * kpad))) return bail(rc);
    const size_t nf32 = h.family == MSC_NIW ? (size_t)ngroups * (h.dim + (size_t)h.dim * h.dim)
                                            : (size_t)raw_f32_rows(h.family) * kpad;
    if ((rc = dev_alloc(st->owned, &h.raw_f32, nf32))) return bail(rc);
    if (loo_rows(h.family) && (rc = dev_alloc(st->owned, &h.loo64, (size_t)loo_rows(h.family) * kpad))) return bail(rc);
    if (loo_tab_rows(h.family, h.dim) &&
        (rc = dev_alloc(st->owned, &h.loo_tab, (size_t)loo_tab_rows(h.family, h.dim) * kpad))) return bail(rc);
    if (h.family == MSC_DM) {
      if ((rc = dev_alloc(st->owned, &h.dm_meta_dev, 2 * ((size_t)h.dim + 1)))) return bail(rc);
      h.dm_meta.assign(2 * ((size_t)h.dim + 1), 0u);
    }
    if (h.family == MSC_NIW) {
      if (h.dim <= (uint32_t)kNiwPad) {                    // operands of the f32 MFMA kernel (MSC_SCORE_NIW_F32), dim <= 32 only
        if ((rc = dev_alloc(st->owned, &h.niw_w, (size_t)ngroups * kNiwPad * kNiwPad))) return bail(rc);
        if ((rc = dev_alloc(st->owned, &h.niw_b, (size_t)ngroups * 2 * kNiwPad))) return bail(rc);
      }
      if ((rc = dev_alloc(st->owned, &h.niw_w64, (size_t)ngroups * niw_w_stream(h.dim)))) return bail(rc);
      if ((rc = dev_alloc(st->owned, &h.niw_mu64, (size_t)ngroups * niw_b_stream(h.dim)))) return bail(rc);
      if ((rc = dev_alloc(st->owned, &h.niw_c64, (size_t)ngroups * 8))) return bail(rc);
    }
    if (!h.hp.empty()) {
      hipError_t e = hipMemcpy(h.hp_dev, h.hp.data(), h.hp.size() * sizeof(float), hipMemcpyHostToDevice);
      if (e != hipSuccess) return bail(fail(MSC_EHIP, "hp upload failed: %s", hipGetErrorString(e)));
    }
    h.raw_valid = true;
    h.additive_valid = true;   // both all-zero
    h.derived_valid = false;
    FeatDesc &d = st->desc_host[f];
    std::memset(&d, 0, sizeof d);
    d.family = h.family;
    d.dim = h.dim;
    d.col_type = value_type_of(h.family);
    d.hp = h.hp_dev;
    d.tab = h.tab;
    d.raw_u32 = h.raw_u32;
    d.raw_f32 = h.raw_f32;
    d.acc_i64 = st->red_i64 + h.i64_off;
    d.acc_f64 = st->red_f64 + h.f64_off;
    d.niw_w = h.niw_w;
    d.niw_b = h.niw_b;
    d.niw_w64 = h.niw_w64;
    d.niw_mu64 = h.niw_mu64;
    d.niw_c64 = h.niw_c64;
    d.vcap = 32;
    d.dm_meta = nullptr;       // set when a column is bound
    d.loo64 = h.loo64;
    d.loo_tab = h.loo_tab;
    d.aux = 0.0;
    for (float a : h.hp) d.aux += h.family == MSC_DD || h.family == MSC_DM ? (double)a : 0.0;
  }
  st->cnt_additive_valid = true;
  if ((rc = upload_desc(st.get()))) return bail(rc);
  hipError_t e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) return bail(fail(MSC_EHIP, "state init failed: %s", hipGetErrorString(e)));
  *out = st.release();
  return MSC_OK;
}

extern "C" int msc_state_destroy(msc_state *st) {
  if (!st) return MSC_OK;
  (void)hipSetDevice(st->ctx->device);
  (void)hipStreamSynchronize(st->ctx->stream);
  if (st->step_graph.exec) (void)hipGraphExecDestroy(st->step_graph.exec);
  free_all(st->owned);
  delete st;
  return MSC_OK;
}

extern "C" int msc_state_shape(const msc_state *st, uint32_t *nfeatures, uint32_t *ngroups) {
  MSC_REQUIRE(st, "null state");
  if (nfeatures) *nfeatures = st->nfeat;
  if (ngroups) *ngroups = st->K;
  return MSC_OK;
}

extern "C" int msc_state_set_hp(msc_state *st, uint32_t feature, const float *host_hp, size_t nfloats) {
  MSC_REQUIRE(st && host_hp, "null argument");
  MSC_REQUIRE(feature < st->nfeat, "feature %u out of range", feature);
  msc_feature_host &h = st->feats[feature];
  MSC_REQUIRE(nfloats == h.hp.size(), "feature %u: hp block has %zu floats, expected %zu", feature,
              nfloats, h.hp.size());
  std::copy(host_hp, host_hp + nfloats, h.hp.begin());
  if (nfloats) {
    MSC_HIP(hipMemcpyAsync(h.hp_dev, h.hp.data(), nfloats * sizeof(float), hipMemcpyHostToDevice, st->ctx->stream));
    MSC_HIP(hipStreamSynchronize(st->ctx->stream));
  }
  h.derived_valid = false;
  if (h.family == MSC_DD || h.family == MSC_DM) {
    double asum = 0;
    for (float a : h.hp) asum += (double)a;
    st->desc_host[feature].aux = asum;
    MSC_TRY(upload_desc(st));
  }
  return MSC_OK;
}

extern "C" int msc_state_get_hp(const msc_state *st, uint32_t feature, float *host_hp, size_t nfloats) {
  MSC_REQUIRE(st && host_hp, "null argument");
  MSC_REQUIRE(feature < st->nfeat, "feature %u out of range", feature);
  const msc_feature_host &h = st->feats[feature];
  MSC_REQUIRE(nfloats == h.hp.size(), "feature %u: hp block has %zu floats, expected %zu", feature,
              nfloats, h.hp.size());
  std::copy(h.hp.begin(), h.hp.end(), host_hp);
  return MSC_OK;
}

static int launch_commit_all(msc_state *st) {
  if (launch_commit(st->ctx->stream, st->desc_dev, (int)st->nfeat, st->kpad, st->red_i64, st->cnt_u32))
    return fail(MSC_EHIP, "k_commit launch failed");
  for (uint32_t f = 0; f < st->nfeat; f++)
    if (st->feats[f].family == MSC_NIW &&
        launch_niw_commit(st->ctx->stream, st->desc_dev, f, st->feats[f].dim, st->K, st->kpad, 1))
      return fail(MSC_EHIP, "k_niw_commit launch failed");
  return MSC_OK;
}

static int ensure_raw(msc_state *st) {
  bool any = false;
  for (auto &h : st->feats) any |= !h.raw_valid;
  if (!any) return MSC_OK;
  MSC_TRY(launch_commit_all(st));
  for (auto &h : st->feats) { h.raw_valid = true; h.derived_valid = false; }
  st->crp_valid = false;
  return MSC_OK;
}

// u32 rows / f32 rows of one family's packed record, in record order
static void record_layout(int family, uint32_t dim, uint32_t &nu32, uint32_t &nf32) {
  nu32 = raw_u32_rows(family, dim);
  nf32 = raw_f32_rows(family);
}

extern "C" int msc_state_set_ss(msc_state *st, uint32_t feature, uint32_t first_group,
                                uint32_t ngroups, const void *host_records, size_t nbytes) {
  MSC_REQUIRE(st && host_records, "null argument");
  MSC_REQUIRE(feature < st->nfeat, "feature %u out of range", feature);
  msc_feature_host &h = st->feats[feature];
  MSC_REQUIRE(ngroups > 0 && first_group <= st->K && ngroups <= st->K - first_group,      // (no 32-bit wrap of the sum)
              "groups [%u,%llu) outside [0,%u)", first_group, (unsigned long long)first_group + ngroups, st->K);
  const size_t rec = msc_ss_bytes(h.family, h.dim);
  MSC_REQUIRE(nbytes == rec * ngroups, "expected %zu bytes of records, got %zu", rec * ngroups, nbytes);
  MSC_HIP(hipSetDevice(st->ctx->device));
  MSC_TRY(ensure_raw(st));
  if (h.family == MSC_NIW) {
    // record = {u32 count, f32 sum_x[d], f32 sum_xxT[d*d]}; device keeps count as a row and the
    // float block group-major
    const size_t nf = h.dim + (size_t)h.dim * h.dim;
    std::vector<uint32_t> cnt(ngroups);
    std::vector<float> blk(nf * ngroups);
    const uint8_t *p = static_cast<const uint8_t *>(host_records);
    for (uint32_t g = 0; g < ngroups; g++) {
      std::memcpy(&cnt[g], p + (size_t)g * rec, 4);
      std::memcpy(&blk[nf * g], p + (size_t)g * rec + 4, 4 * nf);
    }
    MSC_HIP(hipMemcpyAsync(h.raw_u32 + first_group, cnt.data(), 4 * (size_t)ngroups, hipMemcpyHostToDevice, st->ctx->stream));
    MSC_HIP(hipMemcpyAsync(h.raw_f32 + nf * first_group, blk.data(), 4 * nf * ngroups, hipMemcpyHostToDevice, st->ctx->stream));
    MSC_HIP(hipStreamSynchronize(st->ctx->stream));
    h.additive_valid = false;
    h.derived_valid = false;
    return MSC_OK;
  }
  uint32_t nu32, nf32;
  record_layout(h.family, h.dim, nu32, nf32);
  // AoS records -> SoA rows
  std::vector<uint32_t> su((size_t)nu32 * ngroups);
  std::vector<float> sf((size_t)nf32 * ngroups);
  const uint8_t *p = static_cast<const uint8_t *>(host_records);
  for (uint32_t g = 0; g < ngroups; g++) {
    const uint8_t *r = p + (size_t)g * rec;
    for (uint32_t i = 0; i < nu32; i++) std::memcpy(&su[(size_t)i * ngroups + g], r + 4 * i, 4);
    for (uint32_t i = 0; i < nf32; i++) std::memcpy(&sf[(size_t)i * ngroups + g], r + 4 * (nu32 + i), 4);
  }
  for (uint32_t i = 0; i < nu32; i++)
    MSC_HIP(hipMemcpyAsync(h.raw_u32 + (size_t)i * st->kpad + first_group, &su[(size_t)i * ngroups],
                           4 * (size_t)ngroups, hipMemcpyHostToDevice, st->ctx->stream));
  for (uint32_t i = 0; i < nf32; i++)
    MSC_HIP(hipMemcpyAsync(h.raw_f32 + (size_t)i * st->kpad + first_group, &sf[(size_t)i * ngroups],
                           4 * (size_t)ngroups, hipMemcpyHostToDevice, st->ctx->stream));
  MSC_HIP(hipStreamSynchronize(st->ctx->stream));
  h.additive_valid = false;
  h.derived_valid = false;
  return MSC_OK;
}

extern "C" int msc_state_get_ss(msc_state *st, uint32_t feature, uint32_t first_group,
                                uint32_t ngroups, void *host_records, size_t nbytes) {
  MSC_REQUIRE(st && host_records, "null argument");
  MSC_REQUIRE(feature < st->nfeat, "feature %u out of range", feature);
  msc_feature_host &h = st->feats[feature];
  MSC_REQUIRE(ngroups > 0 && first_group <= st->K && ngroups <= st->K - first_group,      // (no 32-bit wrap of the sum)
              "groups [%u,%llu) outside [0,%u)", first_group, (unsigned long long)first_group + ngroups, st->K);
  const size_t rec = msc_ss_bytes(h.family, h.dim);
  MSC_REQUIRE(nbytes == rec * ngroups, "expected %zu bytes of records, got %zu", rec * ngroups, nbytes);
  MSC_HIP(hipSetDevice(st->ctx->device));
  MSC_TRY(ensure_raw(st));
  if (h.family == MSC_NIW) {
    const size_t nf = h.dim + (size_t)h.dim * h.dim;
    std::vector<uint32_t> cnt(ngroups);
    std::vector<float> blk(nf * ngroups);
    MSC_HIP(hipMemcpyAsync(cnt.data(), h.raw_u32 + first_group, 4 * (size_t)ngroups, hipMemcpyDeviceToHost, st->ctx->stream));
    MSC_HIP(hipMemcpyAsync(blk.data(), h.raw_f32 + nf * first_group, 4 * nf * ngroups, hipMemcpyDeviceToHost, st->ctx->stream));
    MSC_HIP(hipStreamSynchronize(st->ctx->stream));
    uint8_t *p = static_cast<uint8_t *>(host_records);
    for (uint32_t g = 0; g < ngroups; g++) {
      std::memcpy(p + (size_t)g * rec, &cnt[g], 4);
      std::memcpy(p + (size_t)g * rec + 4, &blk[nf * g], 4 * nf);
    }
    return MSC_OK;
  }
  uint32_t nu32, nf32;
  record_layout(h.family, h.dim, nu32, nf32);
  std::vector<uint32_t> su((size_t)nu32 * ngroups);
  std::vector<float> sf((size_t)nf32 * ngroups);
  for (uint32_t i = 0; i < nu32; i++)
    MSC_HIP(hipMemcpyAsync(&su[(size_t)i * ngroups], h.raw_u32 + (size_t)i * st->kpad + first_group,
                           4 * (size_t)ngroups, hipMemcpyDeviceToHost, st->ctx->stream));
  for (uint32_t i = 0; i < nf32; i++)
    MSC_HIP(hipMemcpyAsync(&sf[(size_t)i * ngroups], h.raw_f32 + (size_t)i * st->kpad + first_group,
                           4 * (size_t)ngroups, hipMemcpyDeviceToHost, st->ctx->stream));
  MSC_HIP(hipStreamSynchronize(st->ctx->stream));
  uint8_t *p = static_cast<uint8_t *>(host_records);
  for (uint32_t g = 0; g < ngroups; g++) {
    uint8_t *r = p + (size_t)g * rec;
    for (uint32_t i = 0; i < nu32; i++) std::memcpy(r + 4 * i, &su[(size_t)i * ngroups + g], 4);
    for (uint32_t i = 0; i < nf32; i++) std::memcpy(r + 4 * (nu32 + i), &sf[(size_t)i * ngroups + g], 4);
  }
  return MSC_OK;
}

extern "C" int msc_state_set_alpha(msc_state *st, float alpha) {
  MSC_REQUIRE(st, "null state");
  MSC_REQUIRE(alpha > 0.f, "alpha must be positive (group_manager.hpp:78)");
  st->alpha = alpha;
  st->crp_valid = false;
  return MSC_OK;
}

extern "C" int msc_state_set_group_counts(msc_state *st, const uint32_t *host_counts, uint32_t ngroups) {
  MSC_REQUIRE(st && host_counts, "null argument");
  MSC_REQUIRE(ngroups == st->K, "expected %u counts, got %u", st->K, ngroups);
  MSC_HIP(hipSetDevice(st->ctx->device));
  MSC_TRY(ensure_raw(st));
  MSC_HIP(hipMemcpyAsync(st->cnt_u32, host_counts, 4 * (size_t)ngroups, hipMemcpyHostToDevice, st->ctx->stream));
  MSC_HIP(hipStreamSynchronize(st->ctx->stream));
  st->cnt_additive_valid = false;
  st->crp_valid = false;
  return MSC_OK;
}

extern "C" int msc_state_get_group_counts(msc_state *st, uint32_t *host_counts, uint32_t ngroups) {
  MSC_REQUIRE(st && host_counts, "null argument");
  MSC_REQUIRE(ngroups == st->K, "expected %u counts, got %u", st->K, ngroups);
  MSC_HIP(hipSetDevice(st->ctx->device));
  MSC_TRY(ensure_raw(st));
  MSC_HIP(hipMemcpyAsync(host_counts, st->cnt_u32, 4 * (size_t)ngroups, hipMemcpyDeviceToHost, st->ctx->stream));
  MSC_HIP(hipStreamSynchronize(st->ctx->stream));
  return MSC_OK;
}

// ---------------------------------------------------------------------------
// binding a dataview to the state's features
// ---------------------------------------------------------------------------
// dm: one exact count table per category and one for the row totals, each covering 0..max of what the
// bound column holds (capped at kGpMaxTable); the maxima are found once per view column.
// The column as the model's value type.  The reference converts per value, whatever primitive type the column holds
// (runtime_cast::cast, runtime_type.hpp:145-166 -- its own niw descriptor pairs a float64 numpy dtype with a TYPE_F32[dim]
// model, microscopes/models.pyx:259); here the conversion happens once, when a state first binds the column to a model
// of another value type: a device copy through k_unpack's cast (the column read as one-feature records), kept with
// the view.  make = false: only look the copy up (null when there is none yet).
static int column_as(const msc_dataview *view, uint32_t c, int want, bool make, const void **out) {
  *out = nullptr;
  const msc_runtime_type t = view->types[c];
  if (t.type == want) {
    *out = view->cols[c];
    return MSC_OK;
  }
  if (view->converted.size() < view->cols.size()) view->converted.resize(view->cols.size());
  for (const auto &e : view->converted[c])
    if (e.first == want) {
      *out = e.second;
      return MSC_OK;
    }
  if (!make) return MSC_OK;
  hipStream_t s = view->ctx->stream;
  void *dst = nullptr;
  MSC_HIP(hipMalloc(&dst, std::max<size_t>(1, (size_t)view->nrows * t.count * primitive_size(want))));
  view->owned_lazy.push_back(dst);
  if (view->nrows > 0) {
    UnpackFeat uf{};
    uf.dst = dst;
    uf.dst_mask = nullptr;                              // (the mask column is per element, whatever the value type)
    uf.offset = 0;
    uf.mask_offset = 0;
    uf.src_type = t.type;
    uf.dst_type = want;
    uf.count = t.count;
    void *uf_dev = nullptr;
    MSC_HIP(hipMalloc(&uf_dev, sizeof uf));
    hipError_t e = hipMemcpyAsync(uf_dev, &uf, sizeof uf, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && launch_unpack(s, static_cast<const uint8_t *>(view->cols[c]), nullptr, view->nrows,
                                         (uint32_t)(primitive_size(t.type) * t.count), 0, uf_dev, 1))
      e = hipErrorLaunchFailure;
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(uf_dev);
    if (e != hipSuccess) return fail(MSC_EHIP, "column conversion failed: %s", hipGetErrorString(e));
  }
  view->converted[c].emplace_back(want, dst);
  *out = dst;
  return MSC_OK;
}

// A masked lookup column with the mask folded in (FeatDesc::col_sentinel): masked rows hold `sentinel`.  `col` is the
// column as the model's value type (column_as); made once per (column, element type, sentinel), kept with the view.
static int sentinel_column(const msc_dataview *view, uint32_t c, const void *col, bool bytes, uint32_t sentinel, const void **out) {
  *out = nullptr;
  if (view->sentinels.size() < view->cols.size()) view->sentinels.resize(view->cols.size());
  const std::pair<int, uint32_t> key(bytes ? 1 : 4, sentinel);
  for (const auto &e : view->sentinels[c])
    if (e.first == key) {
      *out = e.second;
      return MSC_OK;
    }
  void *dst = nullptr;
  MSC_HIP(hipMalloc(&dst, std::max<size_t>(1, (size_t)view->nrows * (bytes ? 1 : 4))));
  view->owned_lazy.push_back(dst);
  if (launch_mask_sentinel(view->ctx->stream, col, static_cast<const uint8_t *>(view->masks[c]), view->nrows, bytes, sentinel, dst))
    return fail(MSC_EHIP, "k_mask_sentinel launch failed");
  view->sentinels[c].emplace_back(key, dst);
  *out = dst;
  return MSC_OK;
}

static int bind_dm_column(msc_state *st, uint32_t f, const msc_dataview *view, uint32_t c) {
  msc_feature_host &h = st->feats[f];
  FeatDesc &d = st->desc_host[f];
  const uint32_t nst = h.dim + 1;
  hipStream_t s = st->ctx->stream;
  if (view->dm_max[c].empty()) {
    std::vector<uint32_t> mx(nst, 0u);
    if (view->nrows > 0) {
      void *tmp = nullptr, *tot = nullptr;
      MSC_HIP(hipMalloc(&tmp, sizeof(uint32_t) * nst));
      view->owned_lazy.push_back(tmp);
      MSC_HIP(hipMalloc(&tot, sizeof(uint32_t) * view->nrows));
      view->owned_lazy.push_back(tot);
      view->dm_tot[c] = static_cast<uint32_t *>(tot);
      MSC_HIP(hipMemsetAsync(tmp, 0, sizeof(uint32_t) * nst, s));
      if (launch_dm_stats(s, static_cast<const uint32_t *>(d.col), view->nrows, h.dim,
                          static_cast<uint32_t *>(tmp), view->dm_tot[c]))
        return fail(MSC_EHIP, "k_dm_stats launch failed");
      MSC_HIP(hipMemcpyAsync(mx.data(), tmp, sizeof(uint32_t) * nst, hipMemcpyDeviceToHost, s));
      MSC_HIP(hipStreamSynchronize(s));
    }
    view->dm_max[c] = mx;
  }
  d.dm_tot = view->dm_tot[c];
  std::vector<uint32_t> meta(2 * (size_t)nst);
  uint32_t rows = 0;
  for (uint32_t i = 0; i < nst; i++) {
    const uint32_t vcap = (uint32_t)std::min<unsigned long long>((unsigned long long)view->dm_max[c][i] + 1ull, kGpMaxTable);
    meta[2 * i] = rows;
    meta[2 * i + 1] = vcap;
    rows += 2 * vcap;                                   // (hi, lo) row pairs
  }
  d.dm_rows = rows;
  if (meta != h.dm_meta || d.dm_meta == nullptr) {
    if ((size_t)rows + 4 > h.tab_rows_cap || h.loo64 == nullptr) {   // grow the table buffers (the old ones stay owned until destroy)
      float *t = nullptr;
      MSC_TRY(dev_alloc(st->owned, &t, ((size_t)rows + 4) * st->kpad));
      h.tab = t;
      h.tab_rows_cap = (size_t)rows + 4;
      d.tab = t;
      // the leave-one-out twin of the tables ("count v against the group minus v"), one double per entry: read by
      // k_loo_own only, once per row and stage
      double *l = nullptr;
      MSC_TRY(dev_alloc(st->owned, &l, ((size_t)rows / 2 + 1) * st->kpad));
      h.loo64 = l;
      d.loo64 = l;
    }
    h.dm_meta = meta;
    MSC_HIP(hipMemcpyAsync(h.dm_meta_dev, h.dm_meta.data(), sizeof(uint32_t) * meta.size(), hipMemcpyHostToDevice, s));
    MSC_HIP(hipStreamSynchronize(s));
    d.dm_meta = h.dm_meta_dev;
    h.derived_valid = false;
  }
  return MSC_OK;
}

static int bind_view(msc_state *st, const msc_dataview *view, const uint32_t *cols, uint64_t row0,
                     uint64_t nrows) {
  MSC_REQUIRE(view, "null dataview");
  MSC_REQUIRE(view->ctx->device == st->ctx->device, "dataview and state live on different devices");
  MSC_REQUIRE(row0 + nrows <= view->nrows, "rows [%llu,%llu) outside the view (%llu rows)",
              (unsigned long long)row0, (unsigned long long)(row0 + nrows),
              (unsigned long long)view->nrows);
  (void)bound_view_of(st);                                  // (a binding to a view that is gone, or was invalidated, is dropped first)
  bool same = st->bound_view == view && st->bound_serial == view->serial && st->bound_cols.size() == st->nfeat;
  for (uint32_t f = 0; f < st->nfeat && same; f++) same = st->bound_cols[f] == (cols ? cols[f] : f);
  if (same) {
    for (uint32_t f = 0; f < st->nfeat && same; f++) {
      const void *eff = nullptr;
      if (st->feats[f].family == MSC_NOOP) eff = view->cols[st->bound_cols[f]];
      else MSC_TRY(column_as(view, st->bound_cols[f], value_type_of(st->feats[f].family), false, &eff));
      same = st->desc_host[f].col == eff;
    }
  }
  if (same) return MSC_OK;
  st->bound_cols.resize(st->nfeat);
  for (uint32_t f = 0; f < st->nfeat; f++) {
    const uint32_t c = cols ? cols[f] : f;
    MSC_REQUIRE(c < view->types.size(), "feature %u: column %u outside the view (%zu columns)", f, c,
                view->types.size());
    const msc_feature_host &h = st->feats[f];
    const msc_runtime_type t = view->types[c];
    if (h.family != MSC_NOOP) {
      const uint32_t want_n = h.family == MSC_NIW || h.family == MSC_DM ? h.dim : 1;
      // model::get_runtime_type() fixes the element count (distributions.hpp:398-403, _dataview.pyx:27-44); the element
      // type is converted per value upstream (runtime_cast::cast) and once per column here (column_as)
      MSC_REQUIRE(t.count == want_n, "feature %u: column %u holds %u elements per value but the model wants %u", f, c,
                  t.count, want_n);
    }
    st->bound_cols[f] = c;
    const void *eff = view->cols[c];
    if (h.family != MSC_NOOP) MSC_TRY(column_as(view, c, value_type_of(h.family), true, &eff));
    st->desc_host[f].col = eff;
    st->desc_host[f].mask = static_cast<const uint8_t *>(view->masks[c]);
    st->desc_host[f].col_type = h.family != MSC_NOOP ? value_type_of(h.family) : t.type;
    if (is_count_family(h.family)) {
      // the exact table covers counts 0..max of the bound column (capped): find the max once
      if (view->col_max[c] < 0) {
        uint32_t mx = 0;
        if (view->nrows > 0) {
          MSC_HIP(hipMemsetAsync(st->colmax_dev, 0, 4, st->ctx->stream));
          if (launch_col_max_u32(st->ctx->stream, static_cast<const uint32_t *>(eff), view->nrows, st->colmax_dev))
            return fail(MSC_EHIP, "k_col_max_u32 launch failed");
          MSC_HIP(hipMemcpyAsync(&mx, st->colmax_dev, 4, hipMemcpyDeviceToHost, st->ctx->stream));
          MSC_HIP(hipStreamSynchronize(st->ctx->stream));
        }
        view->col_max[c] = (long long)mx;
      }
      const uint32_t vcap = (uint32_t)std::min<long long>(view->col_max[c] + 1, (long long)kGpMaxTable);
      if (vcap != st->desc_host[f].vcap) {
        st->desc_host[f].vcap = vcap;
        st->feats[f].derived_valid = false;
      }
    }
    if (h.family == MSC_DM) MSC_TRY(bind_dm_column(st, f, view, c));
    // a masked column of a lookup family, for the tile kernels: the mask folded in as the index of the family's zero
    // table row, while that row is among those a feature group stages (plan_groups)
    st->desc_host[f].col_sentinel = nullptr;
    if (view->masks[c] != nullptr) {
      uint32_t sentinel = 0;
      bool ok = false, bytes = false;
      switch (h.family) {
        case MSC_BB:
        case MSC_BBNC: sentinel = 2, ok = true, bytes = true; break;
        case MSC_DD: sentinel = h.dim, ok = h.dim + 1 <= 64; break;
        case MSC_GP:
        case MSC_BNB: sentinel = st->desc_host[f].vcap, ok = sentinel + 1 <= 64; break;
        default: break;
      }
      if (ok) MSC_TRY(sentinel_column(view, c, eff, bytes, sentinel, &st->desc_host[f].col_sentinel));
    }
  }
  st->bound_view = view;
  st->bound_serial = view->serial;
  return upload_desc(st);
}

static uint32_t dm_value_slices(const msc_feature_host &h) {
  uint32_t rows = 1;
  for (size_t i = 1; i < h.dm_meta.size(); i += 2) rows = std::max(rows, h.dm_meta[i]);     // {first_row, vcap} per stage
  return std::min<uint32_t>(64, (rows + 3) / 4);
}

// threads per (feature, group) that share the rows of a count / categorical table in the prepare kernels: ~4 rows each
static uint32_t prepare_value_slices(const msc_state *st) {
  uint32_t rows = 1;
  for (uint32_t f = 0; f < st->nfeat; f++) {
    const int fam = st->feats[f].family;
    if (is_count_family(fam)) rows = std::max(rows, st->desc_host[f].vcap);
    else if (fam == MSC_DD) rows = std::max(rows, st->feats[f].dim);
  }
  return std::min<uint32_t>(64, (rows + 3) / 4);
}

static int ensure_derived(msc_state *st) {
  MSC_TRY(ensure_raw(st));
  bool any = false;
  for (auto &h : st->feats) any |= !h.derived_valid;
  if (!any) return MSC_OK;
  if (launch_prepare(st->ctx->stream, st->desc_dev, st->nfeat, st->kpad, prepare_value_slices(st)))
    return fail(MSC_EHIP, "k_prepare launch failed");
  for (uint32_t f = 0; f < st->nfeat; f++)
    if (st->feats[f].family == MSC_NIW && !st->feats[f].derived_valid &&
        launch_niw_prepare(st->ctx->stream, st->desc_dev, f, st->feats[f].dim, st->K, st->kpad))
      return fail(MSC_EHIP, "k_niw_prepare launch failed");
  for (uint32_t f = 0; f < st->nfeat; f++)
    if (st->feats[f].family == MSC_DM && !st->feats[f].derived_valid && st->desc_host[f].dm_meta != nullptr &&
        launch_dm_prepare(st->ctx->stream, st->desc_dev, (int)f, st->feats[f].dim, st->kpad, dm_value_slices(st->feats[f])))
      return fail(MSC_EHIP, "k_dm_prepare launch failed");
  for (auto &h : st->feats) h.derived_valid = true;
  return MSC_OK;
}

static int ensure_crp(msc_state *st) {
  MSC_TRY(ensure_raw(st));
  if (st->crp_valid) return MSC_OK;
  if (launch_crp_prepare(st->ctx->stream, st->cnt_u32, st->K, st->kpad, st->alpha, st->logpc))
    return fail(MSC_EHIP, "k_crp_prepare launch failed");
  st->crp_valid = true;
  return MSC_OK;
}

// ---------------------------------------------------------------------------
// hot path
// ---------------------------------------------------------------------------
// scalar families go through one fused kernel (scores summed over features in registers);
// every niw feature then adds its MFMA pass on top.
static int ensure_own(msc_state *st, uint64_t nrows) {
  if (st->own_cap >= nrows) return MSC_OK;
  void *p = nullptr;
  MSC_HIP(hipMalloc(&p, nrows * sizeof(float)));
  st->owned.push_back(p);
  st->own = static_cast<float *>(p);
  st->own_cap = nrows;
  return MSC_OK;
}

// the tables of the plan's fused bb runs follow their members' (whatever updated those -- prepare, commit, an entity op):
// rebuilt at the head of every call that scores with the fused plan, one small launch
// ... and so does what the plan's nich blocks go by (NichPlanInfo: is a block's c1 one number per group, how far a value may
// lie before a product of four could overflow) -- the same launch
static int refresh_fused_tables(msc_state *st) {
  const bool packed = st->tile_roles_ok || st->tile_nich_only;
  if (!st->fuse_any && !st->nich_blocks_any && !packed) return MSC_OK;
  if (launch_fuse_tables(st->ctx->stream, st->desc_fuse_dev, (int)st->fuse_split, (st->nich_blocks_any || packed) ? (int)st->fuse_nfeat : (int)st->fuse_split, st->kpad))
    return fail(MSC_EHIP, "k_fuse_tables launch failed");
  return MSC_OK;
}

// what the narrow kernels of a partly filled last tile need (launchers.hpp); the packed-table scratch grows on demand
static int tail_plan(msc_state *st, TailPlan &tp) {
  tp = TailPlan();
  tp.cost = st->plan_cost;
  if (!st->tile_narrow_tail_ok) return MSC_OK;
  const size_t need = (size_t)st->tail_pack_rows * 64;
  if (st->tail_pack_floats < need) {
    void *p = nullptr;
    MSC_HIP(hipMalloc(&p, need * sizeof(float)));
    st->owned.push_back(p);
    st->tail_pack = static_cast<float *>(p);
    st->tail_pack_floats = need;
  }
  tp.ok = true;
  tp.masked_nich = st->tail_masked_nich;
  tp.dm = st->tail_dm;
  tp.max_rows = st->tail_max_rows;
  tp.pack_rows = st->tail_pack_rows;
  tp.pack = st->tail_pack;
  return MSC_OK;
}

// (called inside calls that have just bound their view: bind_view leaves bound_view current)
static bool gp_beyond_table(const msc_state *st, uint32_t f) {
  if (!st->bound_view || f >= st->bound_cols.size()) return false;
  const uint32_t c = st->bound_cols[f];
  if (is_count_family(st->feats[f].family)) return st->bound_view->col_max[c] >= (long long)kGpMaxTable;
  if (st->feats[f].family == MSC_DM)      // a row goes to the double path when its total is beyond the tables
    return !st->bound_view->dm_max[c].empty() && st->bound_view->dm_max[c].back() >= kGpMaxTable;
  return false;
}

// does any row's leave-one-out value need the formula paths (dm always; gp / bnb when the column holds counts beyond the table)?
static bool loo_needs_heavy(const msc_state *st) {
  for (uint32_t f = 0; f < st->nfeat; f++) {
    const int fam = st->feats[f].family;
    // (a count family's table covers 0 .. the bound column's maximum unless that exceeds the cap; dm likewise, by row total)
    if ((is_count_family(fam) || fam == MSC_DM) && gp_beyond_table(st, f)) return true;
  }
  return false;
}

// K <= 64 and nothing but scalar families whose tables all fit 64 KiB of LDS at 4 L groups per row: the narrow tiling
// (kernels_sweep.hip k_narrow).  Returns L = lanes per row (4 / 8 / 16) or 0, and the table rows to stage.
static int narrow_lanes(const msc_state *st, uint32_t *table_rows) {
  static const bool off = std::getenv("MSC_NO_NARROW") != nullptr;        // (A/B knob; the tests run both tilings)
  // (32 lanes per row for K <= 128 was measured and loses to the 256-group tiling: 8 bb at K = 100, 0.52 against 0.20 ms)
  if (off || st->K > 64) return 0;
  // Views of many rows leave it to the lane <-> row kernel (round 3's; by the bound VIEW's rows or the rows of the whole a
  // sharded driver announced, so that a state keeps one kernel for all of a view's rows: the two add a row's features in
  // different orders).  At a million rows that kernel is 1.3-2.7x the faster one (tools/scans/k_monotone.sh, MSC_NO_NARROW:
  // sixteen dd32 columns at K = 32 0.34 -> 0.13 ms, 8 bb + 8 nich at K = 64 0.48 -> 0.26, sixteen nich at K = 32 0.30 -> 0.18);
  // this tiling is for the small problems it was made for (C1: 10k rows, 8 us a pass): views below kNarrowMaxRows rows.
  {
    const uint64_t view_rows = st->sweep_rows_hint ? st->sweep_rows_hint : st->bound_view ? st->bound_view->nrows : 0;
    if (view_rows >= kNarrowMaxRows && st->tile_narrow_tail_ok && std::getenv("MSC_TAIL_MIN_ROWS") == nullptr) return 0;
  }
  const int L = st->K <= 16 ? 4 : st->K <= 32 ? 8 : 16;
  uint32_t rows = 0;
  for (uint32_t f = 0; f < st->nfeat; f++) {
    const msc_feature_host &h = st->feats[f];
    switch (h.family) {
      case MSC_BB: case MSC_BBNC: rows += 2; break;
      case MSC_NICH: rows += 6; break;                                      // (NICH_ROWS)
      case MSC_DD: rows += h.dim; break;
      case MSC_GP: case MSC_BNB:
        if (gp_beyond_table(st, f)) return 0;
        rows += st->desc_host[f].vcap;
        break;
      case MSC_NOOP: break;
      default: return 0;                                                    // niw, dm: their own kernels
    }
  }
  if ((size_t)rows * L * 16 > 64u * 1024u) return 0;
  // ... and only where it is the cheaper tiling for THIS plan (by the plan alone, not the call's rows: the narrow kernel
  // adds the features in the caller's order, the tile kernels in the plan's, so a state keeps one of them for all its rows).
  // Per million rows, us, at 16 lanes a row (tools/scans/grid_scan.py at 100k rows, k_monotone.sh): a bb column 30 (the tile
  // plan fuses four of them into one lookup, this tiling cannot), dd / gp 50, nich 60, half of it at 8 lanes and no less at
  // 4; against the tile kernels' rounds at the plan's price (launchers.hpp PlanCost; up to 128 groups the role-split /
  // nich-only kernels run in PAIR mode).  32 bool columns at K = 64, 1M rows: 0.86 ms here, 0.30 on the tile kernel; 64 at
  // K = 8, 100k rows: 0.10 against 0.04.
  {
    double us = 0;
    for (uint32_t f = 0; f < st->nfeat; f++) {
      const int fam = st->feats[f].family;
      us += fam == MSC_BB || fam == MSC_BBNC ? 30.0 : fam == MSC_NICH ? 60.0 : fam == MSC_NOOP ? 0.0 : 50.0;
    }
    us *= std::max(L, 8) / 16.0;
    const bool pair = pair_mode_ok(st->tile_roles_ok ? MSC_PATH_TILE_ROLES : st->tile_nich_only ? MSC_PATH_NICH_PACK : st->tile_lookups_only ? MSC_PATH_LOOKUPS : MSC_PATH_TILE, st->K, false);
    const double tile = st->plan_cost.tile_round_us * (1.0e6 / 128.0 / st->ctx->num_cus) * (pair ? kPairTileShare : 1.0);
    if (us > 1.25 * tile) return 0;
  }
  *table_rows = rows;
  return L;
}

// k_score_nich1's launch shape for a pass of `nrows` x K into `out`: what msc_score_tune remembered for this very
// buffer, else for this size, else the default.  (profiles/r02_placement_study.txt: the shape moves the rate by <= 4 %
// either way; what decides between 5.6 and 7.0 TB/s is where the driver placed the buffer.)
static int nich1_shape_for(msc_context *ctx, const void *out, uint64_t nrows, uint32_t K) {
  // which stores the pass uses (bit 8 of the launch code: plain): what msc_score_tune found for this very buffer; else
  // non-temporal into a buffer this context placed and probed fast, plain into everything else (msc_context::placed)
  int plain = 1;
  for (const msc_context::Placed &p : ctx->placed)
    if (out >= p.base && static_cast<const char *>(out) < static_cast<const char *>(p.base) + p.size) plain = p.nt_fast ? 0 : 1;
  static const int fixed = [] { const char *e = std::getenv("MSC_NICH1_SHAPE"); return e ? std::atoi(e) : -1; }();
  int shape = -1, by_size = -1;
  for (const msc_context::ShapeEntry &e : ctx->nich1_shapes) {
    if (e.nrows != nrows || e.K != K) continue;
    if (e.out == out) {
      shape = e.shape;
      if (e.plain_stores >= 0) plain = e.plain_stores;
      break;
    }
    if (by_size < 0) by_size = e.shape;
  }
  if (shape < 0) shape = by_size < 0 ? 0 : by_size;
  if (fixed >= 0 && fixed < kNich1NumShapes) shape = fixed;
  return shape | (plain ? 0x100 : 0);
}

static int run_score(msc_state *st, uint64_t row0, uint64_t nrows, const int32_t *z_dev, bool crp,
                     bool niw_f32, float *out_dev, uint64_t ld_out) {
  hipStream_t s = st->ctx->stream;
  if (z_dev) {
    MSC_TRY(ensure_own(st, nrows));
    if (launch_loo_own(s, st->ctx->num_cus, loo_needs_heavy(st), st->loo_staged != 0, st->desc_tile_dev, (int)st->nfeat, st->K, st->kpad, row0, nrows, z_dev, crp ? st->logpc : nullptr, st->own))
      return fail(MSC_EHIP, "k_loo_own launch failed");
  }
  uint32_t n_niw = 0;
  for (auto &h : st->feats) n_niw += h.family == MSC_NIW;
  const bool nich1 = st->nfeat == 1 && st->feats[0].family == MSC_NICH;
  uint32_t narrow_rows = 0;
  if (const int nl = narrow_lanes(st, &narrow_rows)) {
    if (launch_narrow(s, st->ctx->num_cus, nl, narrow_rows, false, st->desc_dev, (int)st->nfeat, st->K, st->kpad, row0, nrows,
                      z_dev, st->own, crp ? st->logpc : nullptr, out_dev, ld_out, 0, nullptr, nullptr, ZeroSpans()))
      return fail(MSC_EHIP, "k_narrow launch failed: %s", hipGetErrorString(hipGetLastError()));
    return MSC_OK;
  }
  bool written = false;
  if (n_niw < st->nfeat || crp) {
    bool has_dm = false;
    for (auto &h : st->feats) has_dm |= h.family == MSC_DM;
    const int path = nich1 ? MSC_PATH_NICH1 : has_dm ? MSC_PATH_TILE_DM : st->tile_roles_ok ? MSC_PATH_TILE_ROLES : st->tile_nich_only ? MSC_PATH_NICH_PACK : st->tile_lookups_only ? MSC_PATH_LOOKUPS : MSC_PATH_TILE;
    const FeatDesc *descs = path == MSC_PATH_NICH1 ? st->desc_dev : st->desc_fuse_dev;
    TailPlan tail;
    tail.cost = st->plan_cost;
    if (path != MSC_PATH_NICH1 && st->K - (st->kpad - kGroupTile) <= kTailMaxGroups)
      MSC_TRY(tail_plan(st, tail));
    if (path != MSC_PATH_NICH1) MSC_TRY(refresh_fused_tables(st));
    auto launch = [&](int shape) {
      return launch_score(s, st->ctx->num_cus, path, tail, shape, descs,
                          (int)st->fuse_nfeat, (int)st->fuse_split, st->K, st->kpad, row0,
                          nrows, z_dev, st->own, crp ? st->logpc : nullptr, out_dev, ld_out);
    };
    // The single-nich pass is bound by the HBM write stream; its launch shape (rows per visit, visits per wave =
    // write fronts) is the default (4 rows, 2 visits) unless msc_score_tune settled another one for passes like this
    // (same buffer first, else same size), or MSC_NICH1_SHAPE fixes it.  Nothing here waits for the device.
    const int shape = path == MSC_PATH_NICH1 ? nich1_shape_for(st->ctx, out_dev, nrows, st->K) : 0;
    if (launch(shape))
      return fail(MSC_EHIP, "score kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
    written = true;
    for (uint32_t f = 0; f < st->nfeat; f++)
      if (gp_beyond_table(st, f) &&
          launch_gp_large_fix(s, st->ctx->num_cus, st->desc_dev, (int)f, st->K, st->kpad, row0, nrows, z_dev, out_dev, ld_out))
        return fail(MSC_EHIP, "k_gp_large_fix launch failed");
  }
  for (uint32_t f = 0; f < st->nfeat; f++) {
    if (st->feats[f].family != MSC_NIW) continue;
    if (z_dev && st->niw_qown_cap < nrows) {            // leave-one-out: the score kernel parks q of the own group here
      void *p = nullptr;
      MSC_HIP(hipMalloc(&p, nrows * sizeof(double)));
      st->owned.push_back(p);
      st->niw_qown = static_cast<double *>(p);
      st->niw_qown_cap = nrows;
    }
    if (launch_niw_score(s, st->ctx->num_cus, st->desc_dev, f, st->feats[f].dim, st->K, st->kpad, row0, nrows, z_dev, written,
                         niw_f32, st->niw_qown, out_dev, ld_out))
      return fail(MSC_EHIP, "k_score_niw launch failed: %s", hipGetErrorString(hipGetLastError()));
    written = true;
  }
  return MSC_OK;
}

extern "C" int msc_score_value(msc_state *st, const msc_dataview *view, const uint32_t *cols,
                               uint64_t row0, uint64_t nrows, const int32_t *z_dev, uint32_t flags,
                               float *out_dev, uint64_t ld_out) {
  MSC_REQUIRE(st && (out_dev || nrows == 0), "null argument");    // an empty row range has no output to point at
  MSC_REQUIRE(ld_out >= st->K, "ld_out %llu < ngroups %u", (unsigned long long)ld_out, st->K);
  MSC_REQUIRE((flags & ~(MSC_SCORE_CRP_PRIOR | MSC_SCORE_NIW_F32)) == 0, "unknown flags 0x%x", flags);
  MSC_HIP(hipSetDevice(st->ctx->device));
  MSC_TRY(device_error_check(st->ctx));
  MSC_TRY(bind_view(st, view, cols, row0, nrows));
  if (nrows == 0) return MSC_OK;
  MSC_TRY(ensure_derived(st));
  const bool crp = (flags & MSC_SCORE_CRP_PRIOR) != 0;
  if (crp) MSC_TRY(ensure_crp(st));
  return run_score(st, row0, nrows, z_dev, crp, (flags & MSC_SCORE_NIW_F32) != 0, out_dev, ld_out);
}

// Settle the launch shape of the single-nich scoring pass for passes of this size: every shape of kNich1Shapes, one
// warm-up and six timed launches each into the caller's buffer (every run writes the same values), the fastest is
// remembered per context for (this buffer, nrows, K) and, as the fallback, for (nrows, K).  SYNCHRONOUS (~10 ms): it
// waits on events, so never call it on a capturing stream.  States that do not take the single-nich kernel return
// MSC_OK and remember nothing.
extern "C" int msc_score_tune(msc_state *st, const msc_dataview *view, const uint32_t *cols, uint64_t row0,
                              uint64_t nrows, float *out_dev, uint64_t ld_out, int *shape_out, float *ms_out) {
  MSC_REQUIRE(st && view && out_dev, "null argument");
  MSC_REQUIRE(ld_out >= st->K, "ld_out %llu < ngroups %u", (unsigned long long)ld_out, st->K);
  if (shape_out) *shape_out = -1;
  if (ms_out) *ms_out = 0.f;
  MSC_HIP(hipSetDevice(st->ctx->device));
  MSC_TRY(bind_view(st, view, cols, row0, nrows));
  uint32_t narrow_rows = 0;
  const bool nich1 = st->nfeat == 1 && st->feats[0].family == MSC_NICH && narrow_lanes(st, &narrow_rows) == 0;
  if (!nich1 || nrows == 0) return MSC_OK;
  MSC_TRY(ensure_derived(st));
  hipStream_t s = st->ctx->stream;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
    return fail(MSC_EINVAL, "msc_score_tune waits for the device: not on a capturing stream");
  auto launch = [&](int shape) {
    return launch_score(s, st->ctx->num_cus, MSC_PATH_NICH1, TailPlan(), shape, st->desc_dev, 1, (int)st->tile_split, st->K, st->kpad, row0,
                        nrows, nullptr, nullptr, nullptr, out_dev, ld_out);
  };
  hipEvent_t e0, e1;
  MSC_HIP(hipEventCreate(&e0));
  MSC_HIP(hipEventCreate(&e1));
  float best = 0.f;
  int shape = -1, rc = MSC_OK;
  for (int cand = 0; cand < kNich1NumShapes && rc == MSC_OK; cand++) {
    if (launch(cand)) { rc = fail(MSC_EHIP, "score kernel launch failed: %s", hipGetErrorString(hipGetLastError())); break; }
    hipError_t e = hipEventRecord(e0, s);
    for (int r = 0; r < 6 && e == hipSuccess; r++) (void)launch(cand);
    if (e == hipSuccess) e = hipEventRecord(e1, s);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e != hipSuccess) { rc = fail(MSC_EHIP, "timing failed: %s", hipGetErrorString(e)); break; }
    if (shape < 0 || ms < best) {
      best = ms;
      shape = cand;
    }
  }
  // ... and, with the winning shape, the stores: non-temporal or plain (which is faster depends on where the buffer lies)
  int plain_stores = 0;
  if (rc == MSC_OK && shape >= 0) {
    float ms_kind[2] = {0.f, 0.f};
    for (int kind = 0; kind < 2 && rc == MSC_OK; kind++) {
      const int code = shape | (kind ? 0x100 : 0);
      if (launch(code)) { rc = fail(MSC_EHIP, "score kernel launch failed: %s", hipGetErrorString(hipGetLastError())); break; }
      hipError_t e = hipEventRecord(e0, s);
      for (int r = 0; r < 6 && e == hipSuccess; r++) (void)launch(code);
      if (e == hipSuccess) e = hipEventRecord(e1, s);
      if (e == hipSuccess) e = hipEventSynchronize(e1);
      if (e == hipSuccess) e = hipEventElapsedTime(&ms_kind[kind], e0, e1);
      if (e != hipSuccess) rc = fail(MSC_EHIP, "timing failed: %s", hipGetErrorString(e));
    }
    plain_stores = ms_kind[1] < ms_kind[0] ? 1 : 0;
    best = ms_kind[plain_stores];
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  MSC_TRY(rc);
  auto &memo = st->ctx->nich1_shapes;
  for (size_t i = 0; i < memo.size();)                      // one entry per (buffer, size); the newest size entry leads
    if (memo[i].out == out_dev && memo[i].nrows == nrows && memo[i].K == st->K) memo.erase(memo.begin() + i);
    else i++;
  memo.insert(memo.begin(), msc_context::ShapeEntry{out_dev, nrows, st->K, shape, plain_stores});
  if (memo.size() > 16) memo.pop_back();
  if (shape_out) *shape_out = shape;
  if (ms_out) *ms_out = best / 6.f;
  return MSC_OK;
}

static int commit(msc_state *st) {
  MSC_TRY(launch_commit_all(st));
  for (auto &h : st->feats) { h.raw_valid = true; h.derived_valid = false; }
  st->crp_valid = false;
  return MSC_OK;
}

// internal flags of accumulate_impl, next to the public MSC_ACC_*: the additive tables are already zero (a fused
// sweep kernel did it); commit and prepare in one launch, which also moves the random stream on (msc_sweep_step)
enum : uint32_t { kAccZeroed = 0x100, kAccThenPrepare = 0x200 };

static int commit_and_prepare(msc_state *st, bool bump_rng = true) {
  hipStream_t s = st->ctx->stream;
  if (launch_commit_prepare(s, st->desc_dev, (int)st->nfeat, st->K, st->kpad, st->red_i64, st->cnt_u32, st->alpha,
                            st->logpc, bump_rng ? st->rng_dev : nullptr, prepare_value_slices(st)))
    return fail(MSC_EHIP, "k_commit_prepare launch failed");
  for (uint32_t f = 0; f < st->nfeat; f++) {
    const msc_feature_host &h = st->feats[f];
    if (h.family == MSC_NIW && (launch_niw_commit(s, st->desc_dev, f, h.dim, st->K, st->kpad, 1) ||
                                launch_niw_prepare(s, st->desc_dev, f, h.dim, st->K, st->kpad)))
      return fail(MSC_EHIP, "niw commit / prepare launch failed");
    if (h.family == MSC_DM && st->desc_host[f].dm_meta != nullptr &&
        launch_dm_prepare(s, st->desc_dev, (int)f, h.dim, st->kpad, dm_value_slices(h)))
      return fail(MSC_EHIP, "k_dm_prepare launch failed");
  }
  for (auto &h : st->feats) { h.raw_valid = true; h.derived_valid = true; }
  st->crp_valid = true;
  return MSC_OK;
}

static int accumulate_impl(msc_state *st, const msc_dataview *view, const uint32_t *cols, uint64_t row0,
                           uint64_t nrows, const int32_t *z_dev, uint32_t flags) {
  MSC_TRY(bind_view(st, view, cols, row0, nrows));
  hipStream_t s = st->ctx->stream;
  if (flags & MSC_ACC_RESET) {
    if (!(flags & kAccZeroed) && launch_zero64(s, st->red_i64, st->n_i64, st->red_f64, st->n_f64))
      return fail(MSC_EHIP, "k_zero64 launch failed");
  } else {
    MSC_TRY(ensure_raw(st));
    for (uint32_t f = 0; f < st->nfeat; f++)
      if (!st->feats[f].additive_valid) {
        const int rc = st->feats[f].family == MSC_NIW
                           ? launch_niw_commit(s, st->desc_dev, f, st->feats[f].dim, st->K, st->kpad, 0)
                           : launch_lift(s, st->desc_dev + f, 1, st->kpad, st->red_i64, st->cnt_u32, 0);
        if (rc) return fail(MSC_EHIP, "k_lift launch failed");
      }
    if (!st->cnt_additive_valid)
      if (launch_lift(s, st->desc_dev, 0, st->kpad, st->red_i64, st->cnt_u32, 1))
        return fail(MSC_EHIP, "k_lift launch failed");
  }
  for (auto &h : st->feats) h.additive_valid = true;
  st->cnt_additive_valid = true;
  if (nrows > 0) {
    const int rc = launch_accumulate(s, st->ctx->num_cus, st->desc_acc_dev, st->desc_acc_host.data(), (int)st->desc_acc_host.size(),
                                     st->K, st->kpad, row0, nrows, z_dev,
                                     (flags & MSC_ACC_SUBTRACT) ? -1 : 1, st->red_i64);
    if (rc == -2) return fail(MSC_EUNSUPPORTED, "accumulate tables for %u groups exceed LDS", st->K);
    if (rc) return fail(MSC_EHIP, "k_accumulate launch failed: %s", hipGetErrorString(hipGetLastError()));
    for (uint32_t f = 0; f < st->nfeat; f++) {
      if (st->feats[f].family != MSC_NIW) continue;
      MSC_REQUIRE(nrows < (1ull << 32), "niw accumulate takes at most 2^32 - 1 rows per call");
      const size_t need = 2 * (size_t)st->K + 1 + (size_t)nrows;
      if (st->niw_scratch_len < need) {
        void *p = nullptr;
        MSC_HIP(hipMalloc(&p, need * sizeof(uint32_t)));
        st->owned.push_back(p);
        st->niw_scratch = static_cast<uint32_t *>(p);
        st->niw_scratch_len = need;
      }
      if (launch_niw_accumulate(s, st->ctx->num_cus, st->desc_dev, f, st->K, row0, nrows, z_dev,
                                (flags & MSC_ACC_SUBTRACT) ? -1 : 1, st->niw_scratch, st->feats[f].dim))
        return fail(MSC_EHIP, "niw accumulate launch failed");
    }
  }
  for (auto &h : st->feats) h.raw_valid = false;
  if (flags & kAccThenPrepare) return commit_and_prepare(st);
  if (!(flags & MSC_ACC_NO_COMMIT)) MSC_TRY(commit(st));
  return MSC_OK;
}

extern "C" int msc_accumulate(msc_state *st, const msc_dataview *view, const uint32_t *cols,
                              uint64_t row0, uint64_t nrows, const int32_t *z_dev, uint32_t flags) {
  MSC_REQUIRE(st && (z_dev || nrows == 0), "null argument");
  MSC_REQUIRE((flags & ~(MSC_ACC_RESET | MSC_ACC_SUBTRACT | MSC_ACC_NO_COMMIT)) == 0, "unknown flags 0x%x", flags);
  MSC_HIP(hipSetDevice(st->ctx->device));
  MSC_TRY(device_error_check(st->ctx));
  return accumulate_impl(st, view, cols, row0, nrows, z_dev, flags);
}

extern "C" int msc_entity_op(msc_state *st, const msc_dataview *view, const uint32_t *cols, uint64_t row, uint32_t group,
                             int sign, int32_t *z_dev) {
  MSC_REQUIRE(st && view, "null argument");
  MSC_REQUIRE(sign != 0, "sign must be +1 (join) or -1 (leave)");
  MSC_REQUIRE(group < st->K, "group %u outside [0,%u)", group, st->K);
  MSC_REQUIRE(!st->rng_bump_pending, "msc_entity_op between msc_sweep_step_begin and msc_state_commit_reduce: the additive "
                                     "tables hold uncommitted sums");
  MSC_TRY(device_error_check(st->ctx));
  MSC_HIP(hipSetDevice(st->ctx->device));
  MSC_TRY(bind_view(st, view, cols, row, 1));
  hipStream_t s = st->ctx->stream;
  bool scalar_only = true;
  for (const auto &h : st->feats) scalar_only &= h.family != MSC_NIW && h.family != MSC_DM;
  if (!scalar_only) {
    // general path: the group travels through a one-entry assignment vector of the state's own
    if (!st->one_z) MSC_TRY(dev_alloc(st->owned, &st->one_z, 1));
    if (launch_set_i32(s, st->one_z, (int32_t)group)) return fail(MSC_EHIP, "k_set_i32 launch failed");
    MSC_TRY(accumulate_impl(st, view, cols, row, 1, st->one_z, sign < 0 ? (uint32_t)MSC_ACC_SUBTRACT : 0u));
    if (z_dev && launch_set_i32(s, z_dev + row, sign > 0 ? (int32_t)group : -1)) return fail(MSC_EHIP, "k_set_i32 launch failed");
    return MSC_OK;
  }
  // every table current before, every table current after
  MSC_TRY(ensure_raw(st));
  for (uint32_t f = 0; f < st->nfeat; f++)
    if (!st->feats[f].additive_valid && launch_lift(s, st->desc_dev + f, 1, st->kpad, st->red_i64, st->cnt_u32, 0))
      return fail(MSC_EHIP, "k_lift launch failed");
  if (!st->cnt_additive_valid && launch_lift(s, st->desc_dev, 0, st->kpad, st->red_i64, st->cnt_u32, 1))
    return fail(MSC_EHIP, "k_lift launch failed");
  for (auto &h : st->feats) h.additive_valid = true;
  st->cnt_additive_valid = true;
  MSC_TRY(ensure_derived(st));
  MSC_TRY(ensure_crp(st));
  if (launch_entity_op(s, st->desc_dev, (int)st->nfeat, st->K, st->kpad, row, group, sign > 0 ? 1 : -1, st->red_i64, st->cnt_u32,
                       st->alpha, st->logpc, z_dev ? z_dev + row : nullptr))
    return fail(MSC_EHIP, "k_entity_op launch failed");
  return MSC_OK;
}

extern "C" int msc_score_data(msc_state *st, float *out_dev) {
  MSC_REQUIRE(st && out_dev, "null argument");
  MSC_HIP(hipSetDevice(st->ctx->device));
  MSC_TRY(ensure_raw(st));
  if (launch_score_data(st->ctx->stream, st->desc_dev, (int)st->nfeat, st->K, st->kpad, out_dev))
    return fail(MSC_EHIP, "k_score_data launch failed");
  bool any_niw = false;
  for (auto &h : st->feats) any_niw |= h.family == MSC_NIW;
  if (any_niw) {
    MSC_TRY(ensure_derived(st));   // ln det Psi_n comes from the prepare step
    for (uint32_t f = 0; f < st->nfeat; f++)
      if (st->feats[f].family == MSC_NIW &&
          launch_niw_score_data(st->ctx->stream, st->desc_dev, f, st->K, st->kpad, out_dev))
        return fail(MSC_EHIP, "k_niw_score_data launch failed");
  }
  return MSC_OK;
}

// score + sample in one kernel; niw and gp-beyond-table features and wide K need the materialised path
// one niw feature of small dimension on few groups: the fused k_sweep_niw1
static bool sweep_is_niw1(const msc_state *st) {
  return st->nfeat == 1 && st->feats[0].family == MSC_NIW && (int)st->K <= sweep_niw1_max_groups(st->feats[0].dim);
}

// the lane <-> row kernel fills the chip from ~260k rows on; with fewer the tile kernels' rounds of 128-row chunks can be
// shorter.  Decided on the bound view's row count, so every row range of it takes the same kernels.
// (priced with the cost model of launchers.hpp: the lane <-> row launches -- plus, beyond 64 groups, the trip through
// 128 floats per row and the row sampler -- against the rounds of the fused tile sweep kernel; for a tail beyond a full
// tile, against one more tile pass, the materialised matrix and the sampler)
// PAIR mode of the role-split sweep kernel (at most 128 groups; kernels_sweep.hip): by the bound view's rows (or the rows
// of the whole a sharded driver announced), never the call's -- its draw associates a row's entries differently from the
// other tile kernels', so every row range of a view takes the same one
static bool sweep_pair_mode(const msc_state *st) {
  const uint64_t rows = st->sweep_rows_hint ? st->sweep_rows_hint : st->bound_view ? st->bound_view->nrows : 0;
  return rows >= kTailMinRows && pair_mode_ok(st->tile_roles_ok ? MSC_PATH_TILE_ROLES : st->tile_nich_only ? MSC_PATH_NICH_PACK : st->tile_lookups_only ? MSC_PATH_LOOKUPS : MSC_PATH_TILE, st->K, false);
}
static bool sweep_rows_pays(const msc_state *st, uint32_t groups) {
  const uint64_t rows = st->sweep_rows_hint ? st->sweep_rows_hint : st->bound_view ? st->bound_view->nrows : 0;
  if (const char *forced = std::getenv("MSC_TAIL_MIN_ROWS")) return rows >= (uint64_t)std::atoll(forced);
  if (rows < kTailMinRows) return false;
  const int cus = st->ctx->num_cus;
  const uint64_t c128 = (rows + 127) / 128;
  const bool tail = groups < st->K;                       // the groups beyond a full first tile
  const double sample_us = [&](uint64_t floats_per_row) { return 30.0 + (double)rows * floats_per_row * 4.0 / 2.0e6; }(tail ? st->K : 128);
  double rows_us = tail_rows_us(groups, false, rows, cus, st->plan_cost), tile_us;
  if (tail) {
    rows_us *= 1.15;                                      // (the tile kernel's instantiation that reads the tail is that much slower)
    tile_us = tile_rounds_us(c128, cus, false, st->plan_cost) + sample_us;                    // one more tile pass, then the sampler over K floats a row
  } else {
    if (groups > 64) rows_us += sample_us;
    // (a fused sweep round = the scoring round -- its PAIR share up to 128 groups -- + the draws, which do not shrink with the
    // plan or the mode: ~2.5 us a round; 8 bb columns at K = 64: 0.18 ms in PAIR mode, 0.10 on the lane <-> row kernel)
    PlanCost pc = st->plan_cost;
    pc.sweep_round_us = pc.tile_round_us * (sweep_pair_mode(st) ? kPairTileShare : 1.0) + 2.5;
    tile_us = tile_rounds_us(c128, cus, true, pc);
  }
  return rows_us < tile_us;
}

static bool sweep_is_fused(const msc_state *st) {
  if (sweep_is_niw1(st)) return true;
  const bool nich1 = st->nfeat == 1 && st->feats[0].family == MSC_NICH;
  for (uint32_t f = 0; f < st->nfeat; f++)
    if (st->feats[f].family == MSC_NIW || gp_beyond_table(st, f)) return false;
  return nich1 ? st->K <= sweep_nich1_rows_max_groups() : st->K <= 256;
}

// the sampling kernels read (seed, sweep) from the state's device pair and the step ends by incrementing the sweep
// index there, so consecutive sweeps need no host -> device traffic (and replay as a graph, msc_sweep_step)
// `zeroed` non-null = part of a sweep step: the fused kernels also empty the additive tables for the accumulate pass
// that follows (*zeroed says whether one did), and the increment of the sweep index is left to that pass's tail.
static int sweep_assign_impl(msc_state *st, const msc_dataview *view, const uint32_t *cols, uint64_t row0,
                             uint64_t nrows, uint64_t row_id0, int32_t *z_dev, uint64_t seed, uint64_t sweep,
                             bool *zeroed = nullptr) {
  MSC_TRY(bind_view(st, view, cols, row0, nrows));
  if (nrows == 0) return MSC_OK;
  if (!st->rng_valid || st->rng_seed != seed || st->rng_sweep != sweep) {
    if (launch_rng_set(st->ctx->stream, st->rng_dev, seed, sweep)) return fail(MSC_EHIP, "k_rng_set launch failed");
    st->rng_valid = true;
    st->rng_seed = seed;
    st->rng_sweep = sweep;
  }
  MSC_TRY(ensure_derived(st));
  MSC_TRY(ensure_crp(st));
  hipStream_t s = st->ctx->stream;
  const int cus = st->ctx->num_cus;
  const bool nich1 = st->nfeat == 1 && st->feats[0].family == MSC_NICH;
  bool has_dm = false;
  for (uint32_t f = 0; f < st->nfeat; f++) has_dm |= st->feats[f].family == MSC_DM;
  int rc = -2;
  bool not_zeroed = false;
  ZeroSpans zero;
  if (zeroed) {
    zero.a = reinterpret_cast<unsigned long long *>(st->red_i64); zero.na = st->n_i64;
    zero.b = reinterpret_cast<unsigned long long *>(st->red_f64); zero.nb = st->n_f64;
  }
  if (sweep_is_niw1(st)) {
    rc = launch_sweep_niw1(s, cus, st->feats[0].dim, st->desc_dev, st->K, st->kpad, row0, nrows, row_id0, z_dev, st->logpc,
                           st->rng_dev, zero);
    if (zeroed) *zeroed = rc == 0;
  } else if (sweep_is_fused(st)) {
    if (!nich1) {                                       // (the single-nich kernel computes the own-group values itself)
      MSC_TRY(ensure_own(st, nrows));
      if (launch_loo_own(s, st->ctx->num_cus, loo_needs_heavy(st), st->loo_staged != 0, st->desc_tile_dev, (int)st->nfeat, st->K, st->kpad, row0, nrows, z_dev, st->logpc, st->own))
        return fail(MSC_EHIP, "k_loo_own launch failed");
    }
    uint32_t narrow_rows = 0;
    const int nl = nich1 ? 0 : narrow_lanes(st, &narrow_rows);
    if (nich1 && st->K > 1024) {                          // beyond the register-resident table: lane <-> row
      if (!st->rows_table) {
        void *p = nullptr;
        MSC_HIP(hipMalloc(&p, sweep_nich1_rows_table_floats(st->kpad) * sizeof(float)));
        st->owned.push_back(p);
        st->rows_table = static_cast<float *>(p);
      }
      rc = launch_sweep_nich1_rows(s, cus, st->desc_dev, st->K, st->kpad, row0, nrows, row_id0, z_dev, st->logpc, st->rng_dev, zero, st->rows_table);
    } else if (nich1) rc = launch_sweep_nich1(s, cus, st->desc_dev, st->K, st->kpad, row0, nrows, row_id0, z_dev, st->own, st->logpc, st->rng_dev, zero);
    else if (nl) rc = launch_narrow(s, cus, nl, narrow_rows, true, st->desc_dev, (int)st->nfeat, st->K, st->kpad, row0, nrows, z_dev,
                                    st->own, st->logpc, nullptr, 0, row_id0, z_dev, st->rng_dev, zero);
    else if (refresh_fused_tables(st)) return MSC_EHIP;    // (everything below walks the fused plan)
    else if (st->tile_narrow_tail_ok && st->K <= kTailMaxGroups && std::getenv("MSC_NO_SWEEP_ROWS") == nullptr &&
             sweep_rows_pays(st, st->K)) {
      // at most 128 groups on a plan of lookup + plain nich features: the lane <-> row kernel, whose cost follows the
      // groups (a tile pass costs what 256 cost).  Up to 64: scores and draw in one launch, a lane draws its own row.
      // Beyond: the scores into 128 floats per row, then the row sampler.  The choice looks at the VIEW's rows, not the
      // call's (a shard draws what the whole draws).
      TailPlan tail;
      MSC_TRY(tail_plan(st, tail));
      tail.exact = false;
      if (st->K <= 64) {
        rc = launch_sweep_rows(s, cus, tail, st->desc_fuse_dev, (int)st->fuse_nfeat, (int)st->fuse_split, st->K, st->kpad, row0, nrows, row_id0,
                               z_dev, st->own, st->logpc, st->rng_dev, zero);
        if (rc == 1) rc = -2;
      } else {
        const size_t need = ((size_t)nrows + 16) * 128;
        if (st->tail_floats < need) {
          void *p = nullptr;
          MSC_HIP(hipMalloc(&p, need * sizeof(float)));
          st->owned.push_back(p);
          st->tail_scores = static_cast<float *>(p);
          st->tail_floats = need;
        }
        rc = launch_score_tail(s, cus, tail, st->desc_fuse_dev, (int)st->fuse_nfeat, (int)st->fuse_split, st->K, st->kpad, 0, row0, nrows,
                               z_dev, st->own, st->logpc, st->tail_scores, 128);
        if (rc == 0) {
          rc = launch_sample_rows(s, cus, st->tail_scores, 128, st->K, nrows, row_id0, z_dev, st->rng_dev);
          not_zeroed = true;                              // (nothing emptied the additive tables on the way)
        } else if (rc == 1) rc = -2;
      }
      if (rc == -2) rc = launch_sweep_mixed(s, cus, has_dm, st->tile_roles_ok, sweep_pair_mode(st), st->tile_nich_only, st->tile_lookups_only, st->desc_fuse_dev, (int)st->fuse_nfeat, (int)st->fuse_split, st->K, st->kpad, row0, nrows, row_id0, z_dev, st->own, st->logpc, st->rng_dev, zero);
    } else rc = launch_sweep_mixed(s, cus, has_dm, st->tile_roles_ok, sweep_pair_mode(st), st->tile_nich_only, st->tile_lookups_only, st->desc_fuse_dev, (int)st->fuse_nfeat, (int)st->fuse_split, st->K, st->kpad, row0, nrows, row_id0, z_dev, st->own, st->logpc, st->rng_dev, zero);
    if (zeroed) *zeroed = rc == 0 && !not_zeroed;
  }
  // 256 < K <= 384 on a role-split state: the groups beyond the tile from the narrow kernel (leave-one-out value and prior
  // included) into 64 or 128 floats per row, then the fused kernel over the tile draws over both -- nothing materialised
  // but that.  Where it pays (sweep_rows_pays: by the view's row count, so that a shard draws from the same bits as the whole).
  if (rc == -2 && !nich1 && !has_dm && (st->tile_roles_ok || st->tile_nich_only || st->tile_lookups_only) && st->tile_narrow_tail_ok && tile_roles_enabled() && st->K > 256 &&
      st->K <= (uint32_t)kGroupTile + kTailMaxGroups && std::getenv("MSC_NO_FUSED_TAIL") == nullptr &&
      sweep_rows_pays(st, st->K - kGroupTile)) {
    const uint64_t tail_ld = st->K <= (uint32_t)kGroupTile + 64 ? 64 : 128;
    bool plain = true;
    for (uint32_t f = 0; f < st->nfeat; f++) plain &= st->feats[f].family != MSC_NIW && !gp_beyond_table(st, f);
    if (plain) {
      MSC_TRY(refresh_fused_tables(st));
      MSC_TRY(ensure_own(st, nrows));
      if (launch_loo_own(s, cus, loo_needs_heavy(st), st->loo_staged != 0, st->desc_tile_dev, (int)st->nfeat, st->K, st->kpad, row0, nrows, z_dev, st->logpc, st->own))
        return fail(MSC_EHIP, "k_loo_own launch failed");
      const size_t need = ((size_t)nrows + 16) * tail_ld;
      if (st->tail_floats < need) {
        void *p = nullptr;
        MSC_HIP(hipMalloc(&p, need * sizeof(float)));
        st->owned.push_back(p);
        st->tail_scores = static_cast<float *>(p);
        st->tail_floats = need;
      }
      // (the kernel stores at out + row * ld + k: handing it tail_scores - 256 puts group 256 + j at column j)
      TailPlan tail;
      MSC_TRY(tail_plan(st, tail));
      tail.exact = false;                                  // (nothing else scores these groups for a draw: one sum per group)
      // (65 .. 128 groups beyond the tile on a role-split plan: ONE pass of the role-split kernel in PAIR mode at tile 1
      // instead of three launches of the lane <-> row kernel -- round 5; by the plan and K alone, so a shard takes what the
      // whole takes)
      int tail_rc = -2;
      if (st->tile_roles_ok && tail_ld == 128 && std::getenv("MSC_NO_PAIR") == nullptr)
        tail_rc = launch_score_pair_tail(s, cus, st->desc_fuse_dev, (int)st->fuse_nfeat, (int)st->fuse_split, st->K, st->kpad, row0, nrows,
                                         z_dev, st->own, st->logpc, st->tail_scores, tail_ld);
      if (tail_rc == -2)
        tail_rc = launch_score_tail(s, cus, tail, st->desc_fuse_dev, (int)st->fuse_nfeat, (int)st->fuse_split, st->K, st->kpad,
                                    kGroupTile, row0, nrows, z_dev, st->own, st->logpc, st->tail_scores - kGroupTile, tail_ld);
      if (tail_rc == 0) {
        rc = launch_sweep_roles_tail(s, cus, st->tile_roles_ok ? 0 : st->tile_nich_only ? 1 : 2, st->desc_fuse_dev, (int)st->fuse_nfeat, (int)st->fuse_split, st->K, st->kpad, row0, nrows, row_id0,
                                     z_dev, st->own, st->logpc, st->rng_dev, zero, st->tail_scores);
        if (zeroed) *zeroed = rc == 0;
      }
    }
  }
  if (rc == -2) {
    // generic shape: score a chunk of rows (leave-one-out + prior) into scratch, then sample it
    // (rows of K rounded up to 64 floats, not of the padded table width: at K = 300 the chunk is 320 wide, not 512 --
    // neither the score kernels nor the sampler touch a row beyond K)
    const uint64_t ld = std::min<uint64_t>(st->kpad, ((uint64_t)st->K + 63) & ~63ull);
    // Up to 4 GiB of scores per chunk (288 GB of HBM: the scratch is not what runs out): fewer and larger launches beat
    // keeping the chunk cache-resident (single nich, 1M rows x 300 groups, 32 / 64 / 128 / 256 / 512 MiB: 1.88 / 1.55 /
    // 1.28 / 1.13 / 1.05 ms), a state with niw features wants >= 4 waves per SIMD on its MFMA kernel, and the
    // leave-one-out pass takes its staged kernel only when the rows fill the chip -- C3's columns at K = 512, 1M rows:
    // 5.39 ms a sweep step with 256 MiB chunks (eight of them), 4.95 with 1 GiB, 4.63 with the whole 2 GiB at once
    static const uint64_t forced_mib = std::getenv("MSC_SWEEP_CHUNK_MIB") ? std::strtoull(std::getenv("MSC_SWEEP_CHUNK_MIB"), nullptr, 10) : 0;
    uint64_t chunk = ((forced_mib ? forced_mib : 4096ull) << 20) / (ld * sizeof(float));
    if (chunk == 0) chunk = 1;
    if (chunk > nrows) chunk = nrows;
    if (st->scratch_floats < chunk * ld) {
      void *p = nullptr;
      MSC_HIP(hipMalloc(&p, chunk * ld * sizeof(float)));
      st->owned.push_back(p);
      st->scratch = static_cast<float *>(p);
      st->scratch_floats = chunk * ld;
    }
    for (uint64_t r = 0; r < nrows; r += chunk) {
      const uint64_t n = std::min<uint64_t>(chunk, nrows - r);
      MSC_TRY(run_score(st, row0 + r, n, z_dev + r, true, false, st->scratch, ld));
      if (launch_sample_rows(s, cus, st->scratch, ld, st->K, n, row_id0 + r, z_dev + r, st->rng_dev))
        return fail(MSC_EHIP, "k_sample_rows launch failed");
    }
    rc = 0;
  }
  if (rc) return fail(MSC_EHIP, "sweep kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
  if (zeroed) return MSC_OK;
  if (launch_rng_bump(s, st->rng_dev)) return fail(MSC_EHIP, "k_rng_bump launch failed");
  st->rng_sweep = sweep + 1;
  return MSC_OK;
}

extern "C" int msc_sweep_assign(msc_state *st, const msc_dataview *view, const uint32_t *cols,
                                uint64_t row0, uint64_t nrows, uint64_t row_id0, int32_t *z_dev,
                                uint64_t seed, uint64_t sweep) {
  MSC_REQUIRE(st && (z_dev || nrows == 0), "null argument");
  MSC_HIP(hipSetDevice(st->ctx->device));
  MSC_TRY(device_error_check(st->ctx));
  return sweep_assign_impl(st, view, cols, row0, nrows, row_id0, z_dev, seed, sweep);
}

// ---------------------------------------------------------------------------
// One whole single-process sweep step: assign, then rebuild the tables from the new assignment
// (msc_sweep_assign + msc_accumulate(RESET), commit included).  On small problems the step is a dozen
// launches of a few microseconds each and the gaps between them are the cost, so its steady state -- same
// view, rows and assignment vector, consecutive sweep indices -- is captured once as a HIP graph and replayed.
// ---------------------------------------------------------------------------
static int accumulate_impl(msc_state *st, const msc_dataview *view, const uint32_t *cols, uint64_t row0,
                           uint64_t nrows, const int32_t *z_dev, uint32_t flags);

// everything the host tracks about which device tables are current; a captured step must leave it as it found it
typedef std::vector<uint8_t> StepFlags;
static StepFlags save_flags(const msc_state *st) {
  StepFlags f;
  for (auto &h : st->feats) f.push_back((uint8_t)(h.raw_valid | (h.additive_valid << 1) | (h.derived_valid << 2)));
  f.push_back((uint8_t)(st->cnt_additive_valid | (st->crp_valid << 1) | (st->rng_valid << 2)));
  return f;
}
static void restore_flags(msc_state *st, const StepFlags &f) {
  for (size_t i = 0; i < st->feats.size(); i++) {
    st->feats[i].raw_valid = f[i] & 1;
    st->feats[i].additive_valid = (f[i] >> 1) & 1;
    st->feats[i].derived_valid = (f[i] >> 2) & 1;
  }
  st->cnt_additive_valid = f.back() & 1;
  st->crp_valid = (f.back() >> 1) & 1;
  st->rng_valid = (f.back() >> 2) & 1;
}

extern "C" int msc_sweep_step(msc_state *st, const msc_dataview *view, const uint32_t *cols, uint64_t row0,
                              uint64_t nrows, uint64_t row_id0, int32_t *z_dev, uint64_t seed, uint64_t sweep) {
  MSC_REQUIRE(st && view && (z_dev || nrows == 0), "null argument");
  MSC_HIP(hipSetDevice(st->ctx->device));
  MSC_TRY(device_error_check(st->ctx));
  hipStream_t s = st->ctx->stream;
  auto eager = [&]() -> int {
    st->step_graph.n_eager++;
    bool zeroed = false;
    MSC_TRY(sweep_assign_impl(st, view, cols, row0, nrows, row_id0, z_dev, seed, sweep, &zeroed));
    st->rng_valid = false;                              // (until the tail has moved the device's pair on)
    MSC_TRY(accumulate_impl(st, view, cols, row0, nrows, z_dev,
                            MSC_ACC_RESET | kAccThenPrepare | (zeroed ? (uint32_t)kAccZeroed : 0u)));
    st->rng_valid = true;
    st->rng_sweep = sweep + 1;
    return MSC_OK;
  };
  MSC_TRY(bind_view(st, view, cols, row0, nrows));       // (uploads and waits if the binding changed: never inside a capture)
  msc_state::StepGraph &g = st->step_graph;
  // Replay is opt-in: on ROCm 7.2 / MI355X a graph launch of the step's 3-5 kernels measured 4-5 us SLOWER than
  // launching them (27 -> 32 us at N = 10k), and instantiation costs milliseconds (sweep step).
  const char *gv = std::getenv("MSC_SWEEP_GRAPH");
  const bool no_graph = gv == nullptr || gv[0] == '0' || gv[0] == 0;
  if (nrows == 0) return accumulate_impl(st, view, cols, row0, 0, z_dev, MSC_ACC_RESET);   // (no row draws anything)
  if (no_graph || g.disabled) return eager();
  std::vector<uint32_t> colv(st->nfeat);
  for (uint32_t f = 0; f < st->nfeat; f++) colv[f] = cols ? cols[f] : f;
  const bool same_key = g.view == view && g.view_serial == view->serial && g.row0 == row0 && g.nrows == nrows &&
                        g.row_id0 == row_id0 && g.z == z_dev && g.alpha == st->alpha && g.cols == colv;
  if (!same_key) {
    if (g.exec) { (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; }
    g.view = view; g.view_serial = view->serial; g.row0 = row0; g.nrows = nrows; g.row_id0 = row_id0; g.z = z_dev;
    g.alpha = st->alpha; g.cols = colv;
    g.seen = 0;
  }
  const bool next_sweep = st->rng_valid && st->rng_seed == seed && st->rng_sweep == sweep;
  const StepFlags entry = save_flags(st);
  if (g.exec) {
    if (!next_sweep || entry != g.flags) return eager();
    MSC_HIP(hipGraphLaunch(g.exec, s));
    g.n_replayed++;
    st->rng_sweep = sweep + 1;      // (the graph ends with k_rng_bump; the flags end as they began)
    return MSC_OK;
  }
  // the first two steps with a key run eagerly -- allocations, bindings and first-use setup happen there and the
  // state reaches its steady flags -- and the third is captured, if it is the fused kind (nothing in it then
  // allocates, copies to the host or waits)
  if (g.seen < 2 || !next_sweep || !sweep_is_fused(st)) {
    g.seen++;
    return eager();
  }
  // record on the library's own stream (the caller's may be the null stream), replay on the caller's
  msc_context *ctx = st->ctx;
  if (!ctx->record_stream && hipStreamCreateWithFlags(&ctx->record_stream, hipStreamNonBlocking) != hipSuccess)
    ctx->record_stream = nullptr;
  if (!ctx->record_stream || hipStreamBeginCapture(ctx->record_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    (void)hipGetLastError();
    g.disabled = true;
    return eager();
  }
  const uint64_t sweep_before = st->rng_sweep;
  ctx->stream = ctx->record_stream;
  const int rc = eager();
  ctx->stream = s;
  g.n_eager--;                      // (recorded, not run)
  hipGraph_t graph = nullptr;
  const hipError_t ec = hipStreamEndCapture(ctx->record_stream, &graph);
  bool ok = rc == MSC_OK && ec == hipSuccess && graph != nullptr && save_flags(st) == entry;
  if (ok && hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0) != hipSuccess) {
    g.exec = nullptr;
    ok = false;
  }
  if (graph) (void)hipGraphDestroy(graph);
  if (!ok) {
    // nothing of the captured step has run: put the host's view of the state back and run the step as it is
    (void)hipGetLastError();
    if (g.exec) { (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; }
    g.disabled = true;
    restore_flags(st, entry);
    st->rng_sweep = sweep_before;
    return eager();
  }
  g.flags = entry;
  MSC_HIP(hipGraphLaunch(g.exec, s));
  g.n_replayed++;
  return MSC_OK;
}

// The row-sharded form of the step: everything up to the exchange.  msc_sweep_assign + msc_accumulate(RESET | NO_COMMIT)
// with the step's fusions (the sweep kernel empties the additive tables; the sweep index moves on in the commit);
// the caller all-reduces msc_state_reduce_buffers and calls msc_state_commit_reduce.
extern "C" int msc_sweep_step_begin(msc_state *st, const msc_dataview *view, const uint32_t *cols, uint64_t row0,
                                    uint64_t nrows, uint64_t row_id0, int32_t *z_dev, uint64_t seed, uint64_t sweep) {
  MSC_REQUIRE(st && view && (z_dev || nrows == 0), "null argument");
  MSC_HIP(hipSetDevice(st->ctx->device));
  MSC_TRY(device_error_check(st->ctx));
  if (nrows == 0) return accumulate_impl(st, view, cols, row0, 0, z_dev, MSC_ACC_RESET | MSC_ACC_NO_COMMIT);
  bool zeroed = false;
  MSC_TRY(sweep_assign_impl(st, view, cols, row0, nrows, row_id0, z_dev, seed, sweep, &zeroed));
  st->rng_valid = false;                                // (until the commit has moved the device's pair on)
  st->rng_bump_pending = true;
  st->rng_next_sweep = sweep + 1;
  return accumulate_impl(st, view, cols, row0, nrows, z_dev,
                         MSC_ACC_RESET | MSC_ACC_NO_COMMIT | (zeroed ? (uint32_t)kAccZeroed : 0u));
}

extern "C" int msc_sweep_step_stats(const msc_state *st, uint64_t *eager_steps, uint64_t *graph_steps) {
  MSC_REQUIRE(st, "null argument");
  if (eager_steps) *eager_steps = st->step_graph.n_eager;
  if (graph_steps) *graph_steps = st->step_graph.n_replayed;
  return MSC_OK;
}

extern "C" int msc_state_reduce_buffers(msc_state *st, void **dev_i64, size_t *n_i64, void **dev_f64,
                                        size_t *n_f64) {
  MSC_REQUIRE(st, "null state");
  if (dev_i64) *dev_i64 = st->red_i64;
  if (n_i64) *n_i64 = st->n_i64;
  if (dev_f64) *dev_f64 = st->red_f64;
  if (n_f64) *n_f64 = st->n_f64;
  return MSC_OK;
}

// the same tables as ONE float64 buffer, for a collective that takes one dtype (counts ride along as doubles: exact
// below 2^53): pack -> all-reduce *pack_dev -> unpack -> msc_state_commit_reduce.  One launch each.
extern "C" int msc_state_reduce_pack(msc_state *st, void **pack_dev, size_t *n_f64) {
  MSC_REQUIRE(st && pack_dev, "null argument");
  MSC_HIP(hipSetDevice(st->ctx->device));
  if (!st->red_pack) MSC_TRY(dev_alloc(st->owned, &st->red_pack, st->n_i64 + st->n_f64));
  if (launch_pack64(st->ctx->stream, false, st->red_i64, st->n_i64, st->red_f64, st->n_f64, st->red_pack))
    return fail(MSC_EHIP, "k_pack64 launch failed");
  *pack_dev = st->red_pack;
  if (n_f64) *n_f64 = st->n_i64 + st->n_f64;
  return MSC_OK;
}
extern "C" int msc_state_reduce_unpack(msc_state *st) {
  MSC_REQUIRE(st, "null state");
  MSC_REQUIRE(st->red_pack, "msc_state_reduce_unpack before msc_state_reduce_pack");
  MSC_HIP(hipSetDevice(st->ctx->device));
  if (launch_pack64(st->ctx->stream, true, st->red_i64, st->n_i64, st->red_f64, st->n_f64, st->red_pack))
    return fail(MSC_EHIP, "k_unpack64 launch failed");
  return MSC_OK;
}

// A sweep chooses between the lane <-> row kernel and the tile kernels by ROW COUNT (sweep_rows_pays), and the two sum a
// row's features in different float associations: a shard must choose what the whole would, or a dart may cross a CDF
// step.  By default the count is the bound view's -- right for row ranges of one view.  A driver that gives every rank a
// view of its own shard (common_amd/dist.py ShardedSweep, msc_sweep_step_sharded callers) states the rows of the WHOLE
// here, once; 0 = back to the view's count.
extern "C" int msc_state_set_sweep_rows(msc_state *st, uint64_t global_rows) {
  MSC_REQUIRE(st, "null state");
  if (st->sweep_rows_hint != global_rows && st->step_graph.exec) {
    // (the hint chooses kernels -- lane <-> row or tile, PAIR mode, the narrow tiling -- for every view the state scores:
    // a captured step holds the old choice, and replaying it would draw other bits than the whole does.  ADVICE r04)
    (void)hipGraphExecDestroy(st->step_graph.exec);
    st->step_graph.exec = nullptr;
    st->step_graph.seen = 0;
  }
  st->sweep_rows_hint = global_rows;
  return MSC_OK;
}

extern "C" int msc_state_commit_reduce(msc_state *st) {
  MSC_REQUIRE(st, "null state");
  MSC_HIP(hipSetDevice(st->ctx->device));
  // (what follows a reduce is the next sweep's scoring: commit and prepare in one launch, as in msc_sweep_step)
  const bool bump = st->rng_bump_pending;
  MSC_TRY(commit_and_prepare(st, bump));
  if (bump) {
    st->rng_bump_pending = false;
    st->rng_valid = true;
    st->rng_sweep = st->rng_next_sweep;
  }
  return MSC_OK;
}

extern "C" int msc_relation_blocks(msc_context *ctx, uint32_t ndim, const uint64_t *shape,
                                   const int32_t *const *z_dev, const uint32_t *ngroups,
                                   const uint32_t *positions_dev, uint64_t ncells, int32_t *z_cell_dev) {
  MSC_REQUIRE(ctx && shape && z_dev && ngroups, "null argument");
  MSC_REQUIRE(ndim >= 1 && ndim <= 8, "a relation has 1..8 dimensions (got %u)", ndim);
  MSC_REQUIRE(z_cell_dev || ncells == 0, "null output");
  unsigned long long total = 1, blocks = 1;
  for (uint32_t d = 0; d < ndim; d++) {
    MSC_REQUIRE(shape[d] > 0 && z_dev[d] && ngroups[d] > 0, "dimension %u: empty shape, null assignment or no groups", d);
    total *= shape[d];
    blocks *= ngroups[d];
    MSC_REQUIRE(blocks <= (1ull << 31) - 1, "more than 2^31 - 1 blocks");
  }
  MSC_REQUIRE(positions_dev || ncells == total, "a dense relation of this shape has %llu cells, not %llu", total,
              (unsigned long long)ncells);
  MSC_HIP(hipSetDevice(ctx->device));
  if (launch_relation_blocks(ctx->stream, ndim, shape, z_dev, ngroups, positions_dev, ncells, z_cell_dev))
    return fail(MSC_EHIP, "k_relation_blocks launch failed");
  return MSC_OK;
}

extern "C" int msc_relation_slice_scores(msc_context *ctx, const float *scores_dev, uint64_t ld, uint32_t ndim,
                                         const uint64_t *shape, uint32_t dim, const uint32_t *seg_dev,
                                         const uint32_t *ids_dev, const int32_t *off_dev, uint32_t ncand,
                                         uint32_t cand_stride, uint64_t nent, float *out_dev, uint64_t ld_out) {
  MSC_REQUIRE(ctx && shape && off_dev, "null argument");
  MSC_REQUIRE(ndim >= 1 && ndim <= 8 && dim < ndim, "dimension %u of a %u-dimensional relation", dim, ndim);
  MSC_REQUIRE((seg_dev == nullptr) == (ids_dev == nullptr), "seg_dev and ids_dev come together (sparse) or not at all (dense)");
  MSC_REQUIRE(nent == 0 || ncand == 0 || (scores_dev && out_dev), "null scores / output");
  MSC_REQUIRE(ld_out >= ncand, "ld_out %llu < candidates %u", (unsigned long long)ld_out, ncand);
  MSC_REQUIRE(ncand == 0 || (uint64_t)(ncand - 1) * cand_stride < ld, "candidate %u x stride %u runs past the score row (%llu)",
              ncand, cand_stride, (unsigned long long)ld);
  for (uint32_t d = 0; d < ndim; d++) MSC_REQUIRE(shape[d] > 0, "dimension %u is empty", d);
  MSC_REQUIRE(seg_dev != nullptr || nent <= shape[dim], "%llu entities on a dimension of %llu", (unsigned long long)nent,
              (unsigned long long)shape[dim]);
  MSC_REQUIRE(nent < (1ull << 31), "too many entities for one launch");
  MSC_HIP(hipSetDevice(ctx->device));
  MSC_TRY(device_error_check(ctx));
  if (launch_relation_slice_scores(ctx->stream, scores_dev, ld, ndim, shape, dim, seg_dev, ids_dev, off_dev, ncand, cand_stride,
                                   nent, out_dev, ld_out))
    return fail(MSC_EHIP, "k_relation_slice_scores launch failed");
  return MSC_OK;
}

extern "C" int msc_value_op_single(msc_context *ctx, int family, uint32_t dim, int op,
                                   const float *host_hp, void *host_ss, const void *host_value,
                                   float *score) {
  MSC_REQUIRE(ctx, "null context");
  MSC_REQUIRE(family_ok(family), "unknown family %d", family);
  MSC_REQUIRE(op >= MSC_OP_ADD && op <= MSC_OP_SCORE_DATA, "unknown op %d", op);
  if (family == MSC_NOOP) {           // models/noop.hpp:13-27: nothing to compute, nothing to launch
    if (score) *score = 0.f;
    return MSC_OK;
  }
  MSC_REQUIRE(host_hp && host_ss, "null hp / suff-stats");
  MSC_REQUIRE(op == MSC_OP_SCORE_DATA || host_value, "null value");
  MSC_REQUIRE(op <= MSC_OP_REMOVE || score, "null score");
  if (family == MSC_DD) MSC_REQUIRE(dim >= 1 && dim <= kMaxDDDim, "dd dim %u outside 1..%u", dim, kMaxDDDim);
  if (family == MSC_NIW) MSC_REQUIRE(dim >= 1 && dim <= kMaxNiwDim, "niw dim %u outside 1..%u", dim, kMaxNiwDim);
  if (family == MSC_DM) MSC_REQUIRE(dim >= 1 && dim <= kMaxDDDim, "dm categories %u outside 1..%u", dim, kMaxDDDim);
  const size_t hp_bytes = msc_hp_floats(family, dim) * sizeof(float), ss_bytes = msc_ss_bytes(family, dim);
  const size_t v_bytes = (family == MSC_BB || family == MSC_BBNC) ? 1 : (family == MSC_NIW || family == MSC_DM) ? 4 * (size_t)dim : 4;
  auto up16 = [](size_t v) { return (v + 15) & ~size_t(15); };
  MailboxHeader hd;
  hd.family = family; hd.dim = (int32_t)dim; hd.op = op; hd.status = 0; hd.score = 0.f;
  hd.hp_off = 64;
  hd.ss_off = (uint32_t)up16(hd.hp_off + hp_bytes);
  hd.value_off = (uint32_t)up16(hd.ss_off + ss_bytes);
  MSC_REQUIRE(hd.value_off + v_bytes <= ctx->mailbox_bytes, "record too large for the mailbox");
  if (family == MSC_DD && host_value && op != MSC_OP_SCORE_DATA) {
    int32_t v;
    std::memcpy(&v, host_value, 4);
    MSC_REQUIRE(v >= 0 && (uint32_t)v < dim, "dd value %d outside [0,%u)", v, dim);
  }
  MSC_HIP(hipSetDevice(ctx->device));
  unsigned char *mb = static_cast<unsigned char *>(ctx->mailbox_host);
  std::memcpy(mb, &hd, sizeof hd);
  std::memcpy(mb + hd.hp_off, host_hp, hp_bytes);
  std::memcpy(mb + hd.ss_off, host_ss, ss_bytes);
  if (host_value) std::memcpy(mb + hd.value_off, host_value, v_bytes);
  if (launch_value_op(ctx->stream, ctx->mailbox_dev, dim, family))
    return fail(MSC_EHIP, "k_value_op launch failed: %s", hipGetErrorString(hipGetLastError()));
  // the kernel's last act is status = 1 in this (pinned, host-coherent) mailbox: watching for it costs a PCIe write's
  // latency, waking up from hipStreamSynchronize several microseconds more.  Bounded: the stream wait is the fallback.
  {
    volatile int32_t *flag = &reinterpret_cast<volatile MailboxHeader *>(mb)->status;
    bool seen = false;
    for (int spin = 0; spin < 200000 && !seen; spin++) seen = *flag == 1;
    if (seen) std::atomic_thread_fence(std::memory_order_acquire);
    else MSC_HIP(hipStreamSynchronize(ctx->stream));
  }
  MailboxHeader back;
  std::memcpy(&back, mb, sizeof back);
  if (back.status != 1) return fail(MSC_EHIP, "k_value_op did not complete");
  if (op <= MSC_OP_REMOVE) std::memcpy(host_ss, mb + hd.ss_off, ss_bytes);
  else *score = back.score;
  return MSC_OK;
}
