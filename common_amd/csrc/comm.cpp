// comm.cpp -- the one collective of the path, reachable from C / C++: a sum all-reduce of a state's additive suff-stat
// tables across row shards, over RCCL (xGMI inside a node).  The Python driver does the same exchange through
// torch.distributed (common_amd/dist.py); this is for hosts that are not Python -- a C++ mixture model linked against
// the plugin headers completes a sharded sweep with msc_sweep_step_sharded alone.
//
// librccl is resolved at the first msc_comm_* call (dlopen), so the library loads on machines without it and a Python
// process keeps the single copy torch already mapped.
#include <dlfcn.h>

#include <cstdlib>
#include <new>

#include "msc_internal.hpp"

namespace {

// the slice of rccl.h this file needs (layouts and enum values: /opt/rocm/include/rccl/rccl.h:40-43,448-470)
struct rccl_unique_id { char internal[128]; };
typedef void *rccl_comm_t;
enum { kRcclSuccess = 0, kRcclSum = 0, kRcclInt64 = 4, kRcclFloat64 = 8 };

struct RcclApi {
  void *handle = nullptr;
  int (*GetUniqueId)(rccl_unique_id *) = nullptr;
  int (*CommInitRank)(rccl_comm_t *, int, rccl_unique_id, int) = nullptr;
  int (*CommDestroy)(rccl_comm_t) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  const char *why = nullptr;
};

RcclApi &rccl() {
  static RcclApi api = [] {
    RcclApi a;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
      a.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (a.handle) break;
    }
    if (!a.handle) {
      a.why = "librccl.so.1 not found";
      return a;
    }
#define MSC_RCCL_SYM(field, sym)                                   \
  a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.handle, sym)); \
  if (!a.field) a.why = "librccl lacks " sym;
    MSC_RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
    MSC_RCCL_SYM(CommInitRank, "ncclCommInitRank")
    MSC_RCCL_SYM(CommDestroy, "ncclCommDestroy")
    MSC_RCCL_SYM(AllReduce, "ncclAllReduce")
    MSC_RCCL_SYM(GroupStart, "ncclGroupStart")
    MSC_RCCL_SYM(GroupEnd, "ncclGroupEnd")
    MSC_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef MSC_RCCL_SYM
    return a;
  }();
  return api;
}

}  // namespace

struct msc_comm {
  msc_context *ctx = nullptr;
  rccl_comm_t comm = nullptr;
  int nranks = 1, rank = 0;
  bool owned = true;        // created here (destroyed here) or adopted from the caller
};

using namespace msc;

#define MSC_RCCL(expr)                                                                                  \
  do {                                                                                                  \
    const int _r = (expr);                                                                              \
    if (_r != kRcclSuccess) return fail(MSC_EHIP, "%s failed: %s", #expr, rccl().GetErrorString(_r));   \
  } while (0)

extern "C" size_t msc_comm_unique_id_bytes(void) { return sizeof(rccl_unique_id); }

extern "C" int msc_comm_unique_id(void *id_out, size_t nbytes) {
  MSC_REQUIRE(id_out && nbytes == sizeof(rccl_unique_id), "the id is %zu bytes", sizeof(rccl_unique_id));
  if (rccl().why) return fail(MSC_EUNSUPPORTED, "RCCL unavailable: %s", rccl().why);
  rccl_unique_id id;
  MSC_RCCL(rccl().GetUniqueId(&id));
  std::memcpy(id_out, &id, sizeof id);
  return MSC_OK;
}

extern "C" int msc_comm_create(msc_context *ctx, const void *unique_id, size_t nbytes, int nranks, int rank, msc_comm **out) {
  MSC_REQUIRE(ctx && unique_id && out, "null argument");
  MSC_REQUIRE(nbytes == sizeof(rccl_unique_id), "the id is %zu bytes", sizeof(rccl_unique_id));
  MSC_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "rank %d of %d", rank, nranks);
  *out = nullptr;
  if (rccl().why) return fail(MSC_EUNSUPPORTED, "RCCL unavailable: %s", rccl().why);
  MSC_HIP(hipSetDevice(ctx->device));
  rccl_unique_id id;
  std::memcpy(&id, unique_id, sizeof id);
  rccl_comm_t c = nullptr;
  MSC_RCCL(rccl().CommInitRank(&c, nranks, id, rank));
  msc_comm *m = new (std::nothrow) msc_comm();
  if (!m) {
    (void)rccl().CommDestroy(c);
    return fail(MSC_ENOMEM, "out of host memory");
  }
  m->ctx = ctx; m->comm = c; m->nranks = nranks; m->rank = rank; m->owned = true;
  *out = m;
  return MSC_OK;
}

extern "C" int msc_comm_adopt(msc_context *ctx, void *nccl_comm, int nranks, int rank, msc_comm **out) {
  MSC_REQUIRE(ctx && nccl_comm && out, "null argument");
  MSC_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "rank %d of %d", rank, nranks);
  *out = nullptr;
  if (rccl().why) return fail(MSC_EUNSUPPORTED, "RCCL unavailable: %s", rccl().why);
  msc_comm *m = new (std::nothrow) msc_comm();
  if (!m) return fail(MSC_ENOMEM, "out of host memory");
  m->ctx = ctx; m->comm = nccl_comm; m->nranks = nranks; m->rank = rank; m->owned = false;
  *out = m;
  return MSC_OK;
}

extern "C" int msc_comm_destroy(msc_comm *comm) {
  if (!comm) return MSC_OK;
  if (comm->owned && comm->comm && !rccl().why) {
    (void)hipSetDevice(comm->ctx->device);
    (void)hipStreamSynchronize(comm->ctx->stream);
    (void)rccl().CommDestroy(comm->comm);
  }
  delete comm;
  return MSC_OK;
}

extern "C" int msc_comm_size(const msc_comm *comm, int *nranks, int *rank) {
  MSC_REQUIRE(comm, "null communicator");
  if (nranks) *nranks = comm->nranks;
  if (rank) *rank = comm->rank;
  return MSC_OK;
}

// the exchange: both additive tables summed in place across the ranks, as one RCCL group on the context's stream
extern "C" int msc_state_allreduce(msc_state *st, msc_comm *comm) {
  MSC_REQUIRE(st && comm, "null argument");
  MSC_REQUIRE(comm->ctx->device == st->ctx->device, "communicator and state live on different devices");
  MSC_HIP(hipSetDevice(st->ctx->device));                      // (a communicator of one rank still goes through RCCL: a copy in place)
  hipStream_t s = st->ctx->stream;
  MSC_RCCL(rccl().GroupStart());
  int r1 = kRcclSuccess, r2 = kRcclSuccess;
  if (st->n_i64) r1 = rccl().AllReduce(st->red_i64, st->red_i64, st->n_i64, kRcclInt64, kRcclSum, comm->comm, s);
  if (st->n_f64) r2 = rccl().AllReduce(st->red_f64, st->red_f64, st->n_f64, kRcclFloat64, kRcclSum, comm->comm, s);
  const int r3 = rccl().GroupEnd();
  if (r1 != kRcclSuccess) return fail(MSC_EHIP, "ncclAllReduce(int64) failed: %s", rccl().GetErrorString(r1));
  if (r2 != kRcclSuccess) return fail(MSC_EHIP, "ncclAllReduce(float64) failed: %s", rccl().GetErrorString(r2));
  if (r3 != kRcclSuccess) return fail(MSC_EHIP, "ncclGroupEnd failed: %s", rccl().GetErrorString(r3));
  return MSC_OK;
}

// one row-sharded sweep step, whole: msc_sweep_step_begin, the exchange, msc_state_commit_reduce
extern "C" int msc_sweep_step_sharded(msc_state *st, const msc_dataview *view, const uint32_t *cols, uint64_t row0,
                                      uint64_t nrows, uint64_t row_id0, int32_t *z_dev, uint64_t seed, uint64_t sweep,
                                      msc_comm *comm) {
  MSC_REQUIRE(st && comm, "null argument");
  if (comm->nranks == 1) return msc_sweep_step(st, view, cols, row0, nrows, row_id0, z_dev, seed, sweep);
  MSC_TRY(msc_sweep_step_begin(st, view, cols, row0, nrows, row_id0, z_dev, seed, sweep));
  MSC_TRY(msc_state_allreduce(st, comm));
  return msc_state_commit_reduce(st);
}

// suff-stats of the GLOBAL assignment: local accumulate, exchange, commit (what starts a sharded run)
extern "C" int msc_accumulate_sharded(msc_state *st, const msc_dataview *view, const uint32_t *cols, uint64_t row0,
                                      uint64_t nrows, const int32_t *z_dev, msc_comm *comm) {
  MSC_REQUIRE(st && comm, "null argument");
  MSC_TRY(msc_accumulate(st, view, cols, row0, nrows, z_dev, MSC_ACC_RESET | MSC_ACC_NO_COMMIT));
  MSC_TRY(msc_state_allreduce(st, comm));
  return msc_state_commit_reduce(st);
}
