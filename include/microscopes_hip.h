/*
 * microscopes_hip.h -- C ABI of the MI355X (gfx950) implementation of the
 * component-model scoring hot path of datamicroscopes/common.
 *
 * The reference has no FFI for this path: downstream C++ drives the virtual
 * microscopes::models::group API one value at a time
 * (include/microscopes/models/base.hpp:21-37) and Cython only wraps object
 * lifetimes (microscopes/_models.pyx:16-52).  This header is the boundary a
 * maintainer binds instead (cgo-free: plain C, pointers and sizes only, no torch
 * or C++ types); include/microscopes/ holds the C++ plugin surface that sits on
 * top of it and INTEGRATION.md shows the binding stubs.  Each entry point names
 * the reference interface it replaces.
 *
 * Conventions
 *   - every function returns MSC_OK (0) or a negative msc_status; the message of
 *     the last failure on the calling thread is msc_last_error().  Nothing throws
 *     across this boundary (reference convention: C++ exceptions,
 *     distributions.hpp:140,152,178,198 -- the C++ layer above re-throws).
 *   - "dev" pointers are device (HBM) addresses valid on the context's device;
 *     "host" pointers are ordinary host memory.  Work is enqueued on the
 *     context's HIP stream and is asynchronous unless stated otherwise.
 *   - there is no CPU fallback: without a usable gfx950 device context creation
 *     fails and nothing else can be called.
 */
#ifndef MICROSCOPES_HIP_H
#define MICROSCOPES_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSC_ABI_VERSION 1

typedef enum msc_status {
  MSC_OK = 0,
  MSC_EINVAL = -1,       /* bad argument (shape, family, null pointer, ...) */
  MSC_EHIP = -2,         /* HIP runtime error (message carries hipGetErrorString) */
  MSC_ENODEVICE = -3,    /* no gfx950 device / extension not usable */
  MSC_EUNSUPPORTED = -4, /* valid request outside what is built (e.g. dd dim > 128) */
  MSC_ENOMEM = -5,
  MSC_EDEVICE = -6       /* a kernel of an EARLIER call on this device reported that something it relies on did not hold
                            (an entity leaving a group it is not in, a relation offset past the score row, a wave barrier
                            that timed out); noticed at the synchronising and launching calls, reported once; the tables of
                            the state that call touched are to be rebuilt (msc_accumulate with MSC_ACC_RESET) */
} msc_status;

/* likelihood families; one kernel family each (distributions.hpp:58-64) */
typedef enum msc_family {
  MSC_BB = 0,   /* BetaBernoulli            value bool   (uint8)          */
  MSC_GP = 1,   /* GammaPoisson             value uint32                  */
  MSC_DD = 2,   /* DirichletDiscrete<128>   value int32 in [0, dim)       */
  MSC_NICH = 3, /* NormalInverseChiSq       value float                   */
  MSC_NIW = 4,  /* NormalInverseWishart<-1> value float[dim]              */
  MSC_NOOP = 5, /* noop model, models/noop.hpp:13-53 (API-overhead control) */
  MSC_BBNC = 6, /* non-conjugate Beta-Bernoulli with explicit p (src/models/bbnc.cpp:22-73), value bool */
  MSC_BNB = 7,  /* BetaNegativeBinomial (distributions.hpp:29-36,59-64), hp {alpha, beta, r}, value uint32 */
  MSC_DM = 8    /* Dirichlet-Multinomial (src/models/dm.cpp:10-97), hp alphas[dim], value int32[dim] */
} msc_family;

/* primitive types, include/microscopes/common/type_info.h:10-44 (same order) */
typedef enum msc_primitive_type {
  MSC_TYPE_B = 0, MSC_TYPE_I8, MSC_TYPE_U8, MSC_TYPE_I16, MSC_TYPE_U16, MSC_TYPE_I32,
  MSC_TYPE_U32, MSC_TYPE_I64, MSC_TYPE_U64, MSC_TYPE_F32, MSC_TYPE_F64, MSC_TYPE_NELEMS
} msc_primitive_type;

/* runtime_type{t, n, vec} (runtime_type.hpp:65-141); count == n() */
typedef struct msc_runtime_type {
  int32_t type;   /* msc_primitive_type */
  uint32_t count; /* elements per value: 1 for scalars, n for vector fields */
} msc_runtime_type;

typedef struct msc_feature_spec {
  int32_t family; /* msc_family */
  uint32_t dim;   /* dd: number of categories (<=128); niw: dimension (<=128); dm: categories (<=128); else 0 */
} msc_feature_spec;

typedef struct msc_context msc_context;
typedef struct msc_dataview msc_dataview;
typedef struct msc_state msc_state;

/* ---- library / context ------------------------------------------------- */
int msc_abi_version(void);
const char *msc_last_error(void);
const char *msc_build_info(void); /* "gfx950 hipcc <ver> ..." */
/*
 * Which kernel INSTANTIATION the library chose for this process's most recent scoring pass (which = 0: msc_score_value)
 * or fused assignment pass (which = 1: msc_sweep_assign / msc_sweep_step), spelled as rocprofv3 spells it, e.g.
 * "k_score_tile_roles<false, false, false>" ("" before the first such call).  Measurement tooling only: bench.py keys the
 * committed counter summaries (profiles/ *_pmc.json) by it, so that a roofline figure is always the figure of the kernel
 * that ran.  Nothing comparable upstream (the reference has no kernels).
 */
const char *msc_last_kernel(int which);

/* stream: a hipStream_t (may be NULL = the device's null stream). */
int msc_context_create(int device, void *stream, msc_context **out);
int msc_context_destroy(msc_context *ctx);
int msc_context_set_stream(msc_context *ctx, void *stream);
int msc_context_synchronize(msc_context *ctx);

/*
 * Device buffers (zero-filled), for hosts that keep no HIP headers of their own: the assignment vector z, a row of
 * scores, the [N, K] score matrix.  upload / download are ordered on the context's stream and complete before they
 * return.
 *
 * From 64 MiB on a buffer is PLACED for the write stream of a score matrix: the same 1 GB stream takes 5.5 TB/s into
 * most allocations and 7.0 TB/s into some, decided by where the driver put the pages (profiles/r02_placement_study.txt),
 * and no allocator argument selects that.  A candidate is mapped from 32 MiB physical chunks through the virtual-memory
 * API (such buffers land in the upper band more often than hipMalloc'ed ones), stream-filled a few times on the
 * context's stream, and kept when it takes the stream at 6.65 TB/s or better; otherwise the next candidate is tried, up
 * to 12, and the fastest is kept.  The search also ends when the candidates held side by side would exceed half of what
 * the device had free at the call, and after six candidates whose rates lie within 6 % of each other (a box without a
 * fast stretch: nothing to find).  SYNCHRONOUS, about 1 ms per candidate and GB; MSC_ALLOC_CANDIDATES /
 * MSC_ALLOC_ACCEPT_GBPS / MSC_ALLOC_FLAT_AFTER in the environment change the bounds, MSC_ALLOC_CANDIDATES=0 is plain
 * hipMalloc.
 * What the probe finds also decides how the single-nich scoring pass WRITES such a buffer: non-temporal stores into a
 * buffer that took the probe at 6.65 TB/s or better (the pass then runs at 0.85-0.88 of the HBM roof), plain stores into
 * every other buffer -- a placed one from the probe's lower bands, or one the caller brought -- where they run 0.78-0.83
 * whatever the placement and non-temporal ones 0.69-0.76 (profiles/r04_store_policy.txt; MSC_NICH1_STORES = nt | plain
 * overrides, msc_score_tune times both for a given buffer).
 * msc_device_alloc_probed is the same with the bounds given by the caller: all `candidates` are probed and the fastest
 * is returned; rates_gbps (nullable, `candidates` floats) receives every candidate's fill rate, *chosen (nullable) the
 * index kept.  msc_device_alloc_stats reports the same for the context's most recent placed allocation.
 * Free with msc_device_free.
 */
int msc_device_alloc(msc_context *ctx, size_t nbytes, void **out_dev);
int msc_device_alloc_probed(msc_context *ctx, size_t nbytes, uint32_t candidates, void **out_dev,
                            float *rates_gbps, uint32_t *chosen);
int msc_device_alloc_stats(msc_context *ctx, float *rates_gbps, uint32_t capacity, uint32_t *ntried, uint32_t *chosen);
int msc_device_free(msc_context *ctx, void *dev);
/*
 * Pinned host memory the device writes straight into (zero-copy): a row of scores the host reads after
 * msc_context_synchronize, with no copy in between -- what a per-entity caller wants for its one row of K floats.
 * *host is the CPU address, *dev the address to hand to the kernels (msc_score_value's out_dev).
 */
int msc_pinned_alloc(msc_context *ctx, size_t nbytes, void **host, void **dev);
int msc_pinned_free(msc_context *ctx, void *host);
int msc_device_upload(msc_context *ctx, void *dst_dev, const void *src_host, size_t nbytes);
int msc_device_download(msc_context *ctx, void *dst_host, const void *src_dev, size_t nbytes);

/* ---- columnar dataview (replaces recarray/dataview.hpp:194-217) -------- */
/*
 * Packed row-major records exactly as numpy_dataview hands them over
 * (microscopes/common/recarray/_dataview.pyx:61-92): n records of
 * sum(size(types[i])) bytes, no padding, optional mask with one bool per
 * element (runtime_type.hpp:123-134).  The records are copied to the device
 * once and transposed there into one contiguous column per feature, converted
 * with runtime_cast::cast semantics (runtime_type.hpp:145-166) to col_types[i]
 * (NULL: keep each feature's own primitive type).  Synchronous w.r.t. the host
 * buffers: they may be freed on return.
 */
int msc_dataview_from_records(msc_context *ctx, const void *host_records, const uint8_t *host_mask,
                              uint64_t nrows, const msc_runtime_type *types, uint32_t ntypes,
                              const int32_t *col_types, msc_dataview **out);
/*
 * Adopt columns that already live in HBM (generated on the device, or a torch
 * tensor): dev_columns[i] has nrows * types[i].count elements of types[i].type,
 * row-major for vector features, aligned to the element size (MSC_EINVAL otherwise).  dev_masks may be NULL or hold NULL entries;
 * a non-NULL entry has nrows * count bytes (nonzero = masked).  Borrowed, not
 * owned: the caller keeps them alive for the life of the view.
 */
int msc_dataview_from_device_columns(msc_context *ctx, uint64_t nrows,
                                     const msc_runtime_type *types, uint32_t ntypes,
                                     void *const *dev_columns, void *const *dev_masks,
                                     msc_dataview **out);
int msc_dataview_destroy(msc_dataview *view);
/*
 * What the library derives from a view's columns and keeps with the view (a column converted to a model's value type,
 * a masked column with the mask folded in, bool columns packed four to a byte, the maxima that size the exact count
 * tables) is a snapshot of the columns' CONTENTS.  After rewriting columns adopted by msc_dataview_from_device_columns
 * in place (minibatches through fixed buffers), call this: every derived copy is dropped and every state binds and
 * derives afresh at its next call.  Synchronises the context's stream.  (A view made from records owns its columns:
 * nothing can rewrite them.)
 */
int msc_dataview_invalidate(msc_dataview *view);
int msc_dataview_size(const msc_dataview *view, uint64_t *nrows, uint32_t *nfeatures);
int msc_dataview_column(const msc_dataview *view, uint32_t feature, void **dev_ptr,
                        msc_runtime_type *type);

/* ---- group tables: hypers + K groups of suff-stats per feature --------- */
/*
 * One state = the (hypers[f], groups[f][gid]) tables a mixture state object
 * keeps (entity_state.hpp:25-90): nfeatures component models, ngroups group
 * slots each, plus the CRP bookkeeping of group_manager (group_manager.hpp:
 * 218-283: per-group entity counts and alpha).  Group ids are the dense slot
 * numbers 0..ngroups-1.
 */
int msc_state_create(msc_context *ctx, const msc_feature_spec *features, uint32_t nfeatures,
                     uint32_t ngroups, msc_state **out);
int msc_state_destroy(msc_state *st);
int msc_state_shape(const msc_state *st, uint32_t *nfeatures, uint32_t *ngroups);

/*
 * hypers::set_hp / get_hp / get_hp_mutator (base.hpp:44-47) as flat float
 * blocks, field order as the reference names them (distributions.hpp:21-56):
 *   bb {alpha, beta}  bbnc {alpha, beta}  gp {alpha, inv_beta}  dd {alphas[dim]}
 *   nich {mu, kappa, sigmasq, nu}  niw {kappa, nu, mu[dim], psi[dim*dim]}
 *   bnb {alpha, beta, r (integral, carried as a float)}  dm {alphas[dim]} (dm.hpp:168)
 */
size_t msc_hp_floats(int family, uint32_t dim);
int msc_state_set_hp(msc_state *st, uint32_t feature, const float *host_hp, size_t nfloats);
int msc_state_get_hp(const msc_state *st, uint32_t feature, float *host_hp, size_t nfloats);

/*
 * group::set_ss / get_ss / get_ss_mutator (base.hpp:31-34) as packed host
 * records, one per group, float fields in float exactly as the reference keeps
 * them (distributions.hpp:21-56,79-91):
 *   bb   {u32 heads, u32 tails}
 *   bbnc {u32 heads, u32 tails, f32 p}       (p is state, not a count: accumulate leaves it alone)
 *   gp   {u32 count, u32 sum, f32 log_prod}
 *   dd   {u32 count_sum, u32 counts[dim]}
 *   nich {u32 count, f32 mean, f32 count_times_variance}
 *   niw  {u32 count, f32 sum_x[dim], f32 sum_xxT[dim*dim]}
 *   bnb  {u32 count, u32 sum}                (distributions.hpp:34-36)
 *   dm   {u32 counts[dim], f32 ratio}        (include/microscopes/models/dm.hpp:86-88)
 * Synchronous.
 */
size_t msc_ss_bytes(int family, uint32_t dim);
int msc_state_set_ss(msc_state *st, uint32_t feature, uint32_t first_group, uint32_t ngroups,
                     const void *host_records, size_t nbytes);
int msc_state_get_ss(msc_state *st, uint32_t feature, uint32_t first_group, uint32_t ngroups,
                     void *host_records, size_t nbytes);

/* group_manager: alpha (get_hp_mutator("alpha"), group_manager.hpp:124-130) and counts */
int msc_state_set_alpha(msc_state *st, float alpha);
int msc_state_set_group_counts(msc_state *st, const uint32_t *host_counts, uint32_t ngroups);
int msc_state_get_group_counts(msc_state *st, uint32_t *host_counts, uint32_t ngroups);

/* ---- the hot path ------------------------------------------------------ */
#define MSC_SCORE_CRP_PRIOR 0x1u /* add log(pseudocount(gid)), group_manager.hpp:274-283 */
#define MSC_SCORE_NIW_F32 0x2u   /* niw (dim <= 32) Mahalanobis on the f32 matrix pipe: 2x the rate, ~1e-5 instead of 1e-6;
                                    ignored for wider features */

/*
 * score_value for nrows rows x all groups x all features of the state:
 *   out[(r) * ld_out + k] = sum_f groups[f][k].score_value(hypers[f], row(row0+r)[cols[f]])
 * i.e. the K x D inner loop of entity_based_state_object::inplace_score_value
 * (entity_state.hpp:69-72) for a block of rows at once.
 * cols[f] = dataview column feeding state feature f (NULL: identity).
 * z_dev (nullable, int32[nrows] indexed from row0): leave-one-out -- row r is
 * scored against group z[r] with itself removed (remove_value before
 * score_value); z < 0 means unassigned, and so does an id >= ngroups
 * (no entry point indexes a table with an id it has not range-checked).
 * out_dev: float[nrows * ld_out], ld_out >= ngroups.  Any ld_out and alignment work; the kernels store 16 bytes a lane
 * when out_dev is 16-byte aligned and ld_out a multiple of 4 (8 bytes a lane when both are even), and a row that is whole
 * 64-byte lines -- ld_out a multiple of 16 -- is written at up to 1.7x the rate of one that is not (8 bb columns, 1M rows:
 * ld_out = 348: 0.46 ms, 352: 0.28; neighbouring rows share a line then, written by different waves at different times).
 */
int msc_score_value(msc_state *st, const msc_dataview *view, const uint32_t *cols, uint64_t row0,
                    uint64_t nrows, const int32_t *z_dev, uint32_t flags, float *out_dev,
                    uint64_t ld_out);

/*
 * Optional and SYNCHRONOUS (about 10 ms): settle the launch shape of the single-NICH scoring pass (the HBM-write-bound
 * kernel of config C2 / C5) for passes of nrows rows into out_dev -- eight shapes, seven launches each into the
 * caller's own buffer (every run writes the same scores).  The winner is remembered in the context for (out_dev,
 * nrows, ngroups) and, as the fallback, for (nrows, ngroups); msc_score_value itself never times anything and never
 * waits.  *shape_out (nullable) = index of the chosen shape or -1 when the state does not take that kernel,
 * *ms_out (nullable) = its time per pass.  Not on a capturing stream.
 */
int msc_score_tune(msc_state *st, const msc_dataview *view, const uint32_t *cols, uint64_t row0,
                   uint64_t nrows, float *out_dev, uint64_t ld_out, int *shape_out, float *ms_out);

#define MSC_ACC_RESET 0x1u    /* zero the tables first (then: suff-stats := f(z)) */
#define MSC_ACC_SUBTRACT 0x2u /* remove_value instead of add_value */
#define MSC_ACC_NO_COMMIT 0x4u /* leave the sums in the reduce buffer (all-reduce follows) */

/*
 * Bulk add_value / remove_value (base.hpp:25-26 + group_manager.hpp:218-248):
 * every row r in [row0, row0+nrows) with z[r] >= 0 is added to (removed from)
 * group z[r] of every feature, and the group counts follow.  Integer fields are
 * exact; float fields are accumulated in double in additive form and converted
 * to the reference's fields on commit.
 */
int msc_accumulate(msc_state *st, const msc_dataview *view, const uint32_t *cols, uint64_t row0,
                   uint64_t nrows, const int32_t *z_dev, uint32_t flags);

/*
 * ONE entity joins (sign > 0) or leaves (sign < 0) ONE group, the group passed by value: group_manager::add_value /
 * remove_value together with every component model's add_value / remove_value for that row (the per-entity calls of
 * entity_based_state_object, entity_state.hpp:57-68).  For states of scalar families this is a single launch that
 * leaves every table current (sums, the reference's fields, score constants and CRP terms of the one group that
 * changed), so a Gibbs move -- leave, msc_score_value of the row, join -- is three launches and one copy back;
 * niw / dm features take the general accumulate path.  z_dev (nullable): the caller's device assignment vector, of
 * which entry `row` is set to the group (join) or -1 (leave).  Asynchronous.
 * Preconditions the device checks (the reference asserts them, group_manager.hpp:218-248): a leave needs a non-empty
 * group -- and, with z_dev, z_dev[row] == group --, a join with z_dev needs z_dev[row] unassigned.  A violation skips the
 * update it concerns and surfaces as MSC_EDEVICE at the next synchronising or launching call.  MSC_EINVAL between
 * msc_sweep_step_begin and msc_state_commit_reduce (the additive tables hold one rank's uncommitted sums then).
 */
int msc_entity_op(msc_state *st, const msc_dataview *view, const uint32_t *cols, uint64_t row, uint32_t group,
                  int sign, int32_t *z_dev);

/* score_data (base.hpp:28) for every (feature, group): out_dev[f * ngroups + k] */
int msc_score_data(msc_state *st, float *out_dev);

/*
 * One synchronous Gibbs assignment sweep over rows [row0, row0+nrows)
 * (SURVEY 3.2 as a data-parallel schedule): every row is scored leave-one-out
 * against the tables as they stand, plus the CRP term, and re-drawn with
 * util::sample_discrete_log (util.hpp:125-156) using the counter-based uniform
 * Philox4x32-10(key = seed, counter = (global row id, sweep)).  row_id0 is the
 * global id of row0 (rank offset when rows are sharded).  z_dev is updated in
 * place; tables are NOT updated (call msc_accumulate with MSC_ACC_RESET next,
 * then all-reduce).
 */
int msc_sweep_assign(msc_state *st, const msc_dataview *view, const uint32_t *cols, uint64_t row0,
                     uint64_t nrows, uint64_t row_id0, int32_t *z_dev, uint64_t seed,
                     uint64_t sweep);

/*
 * One whole single-process sweep step: msc_sweep_assign, then
 * msc_accumulate(MSC_ACC_RESET) of the same rows with the new assignment
 * (commit included) -- the loop body of SURVEY 3.2 when nothing is sharded.
 * Same results as the two calls, in fewer launches: the fused sweep kernels
 * empty the additive tables on their way, and commit + prepare + the CRP terms
 * of the next sweep are one kernel (3 launches per step for a single nich
 * feature instead of 7), which is what bounds small problems.
 * MSC_SWEEP_GRAPH=1 in the environment additionally captures the step as a HIP
 * graph once consecutive calls repeat (same view, rows, z_dev, seed; sweep =
 * previous + 1) and replays it; off by default because it measured slower
 * than the launches it replaces on ROCm 7.2.
 */
int msc_sweep_step(msc_state *st, const msc_dataview *view, const uint32_t *cols, uint64_t row0,
                   uint64_t nrows, uint64_t row_id0, int32_t *z_dev, uint64_t seed, uint64_t sweep);
/*
 * The row-sharded form of the step, up to the exchange: msc_sweep_assign +
 * msc_accumulate(MSC_ACC_RESET | MSC_ACC_NO_COMMIT) with the step's fusions.
 * Then all-reduce msc_state_reduce_buffers and call msc_state_commit_reduce.
 */
int msc_sweep_step_begin(msc_state *st, const msc_dataview *view, const uint32_t *cols, uint64_t row0,
                         uint64_t nrows, uint64_t row_id0, int32_t *z_dev, uint64_t seed, uint64_t sweep);
/* how many msc_sweep_step calls on this state ran launch by launch / as a graph launch */
int msc_sweep_step_stats(const msc_state *st, uint64_t *eager_steps, uint64_t *graph_steps);

/* ---- multi-GPU hook ---------------------------------------------------- */
/*
 * The additive form of every table, ready for a sum all-reduce across row
 * shards: *dev_i64 = int64[n_i64] (counts), *dev_f64 = double[n_f64] (float
 * sums).  After reducing both in place call msc_state_commit_reduce.
 */
int msc_state_reduce_buffers(msc_state *st, void **dev_i64, size_t *n_i64, void **dev_f64,
                             size_t *n_f64);
/*
 * The same two tables as ONE float64 buffer, for a collective that takes one dtype (the payload is a few KB, so an
 * exchange costs the collective's latency: one all-reduce, not two): msc_state_reduce_pack copies counts (as doubles --
 * integers below 2^53 add exactly and in any order, so they come back bit-exact) and float sums into a buffer the state
 * owns and returns it; sum-all-reduce *pack_dev in place; msc_state_reduce_unpack copies both back; then
 * msc_state_commit_reduce.  One small launch each (common_amd/dist.py drives torch.distributed this way).
 */
int msc_state_reduce_pack(msc_state *st, void **pack_dev, size_t *n_f64);
int msc_state_reduce_unpack(msc_state *st);
/*
 * The rows of the WHOLE dataset, for a state whose sweeps run on a SHARD of it through a view of its own: a sweep picks
 * between two kernels by row count, and they associate a row's float sum differently, so a shard must pick what the
 * unsharded sweep would (it then draws identical assignments).  Row ranges of ONE view need nothing (the default is the
 * bound view's row count); 0 restores that default.
 */
int msc_state_set_sweep_rows(msc_state *st, uint64_t global_rows);
int msc_state_commit_reduce(msc_state *st);

/*
 * The same exchange for hosts that are not Python: a communicator over RCCL (xGMI inside a node), one rank per
 * process and GPU.  The 128-byte id is created on one rank (msc_comm_unique_id) and carried to the others by the
 * caller's own means (MPI, a file, a socket), exactly as ncclGetUniqueId / ncclCommInitRank want it;
 * msc_comm_adopt wraps an ncclComm_t the caller already has.  librccl is resolved at the first of these calls.
 *   msc_state_allreduce      sums both additive tables in place across the ranks (one RCCL group, context's stream)
 *   msc_sweep_step_sharded   msc_sweep_step_begin + that + msc_state_commit_reduce: a whole sharded sweep step
 *   msc_accumulate_sharded   suff-stats of the GLOBAL assignment: local accumulate, exchange, commit
 * With one rank these are msc_sweep_step / msc_accumulate.  Asynchronous (msc_comm_create / destroy are not).
 */
typedef struct msc_comm msc_comm;
size_t msc_comm_unique_id_bytes(void);
int msc_comm_unique_id(void *id_out, size_t nbytes);
int msc_comm_create(msc_context *ctx, const void *unique_id, size_t nbytes, int nranks, int rank, msc_comm **out);
int msc_comm_adopt(msc_context *ctx, void *nccl_comm, int nranks, int rank, msc_comm **out);
int msc_comm_destroy(msc_comm *comm);
int msc_comm_size(const msc_comm *comm, int *nranks, int *rank);
int msc_state_allreduce(msc_state *st, msc_comm *comm);
int msc_sweep_step_sharded(msc_state *st, const msc_dataview *view, const uint32_t *cols, uint64_t row0,
                           uint64_t nrows, uint64_t row_id0, int32_t *z_dev, uint64_t seed, uint64_t sweep,
                           msc_comm *comm);
int msc_accumulate_sharded(msc_state *st, const msc_dataview *view, const uint32_t *cols, uint64_t row0,
                           uint64_t nrows, const int32_t *z_dev, msc_comm *comm);

/* ---- per-value entry (the virtual group API, base.hpp:25-28) ----------- */
typedef enum msc_value_op {
  MSC_OP_ADD = 0, MSC_OP_REMOVE = 1, MSC_OP_SCORE_VALUE = 2, MSC_OP_SCORE_DATA = 3
} msc_value_op;
/*
 * One group, one value, evaluated on the device (a batch of one): host_hp and
 * host_ss as in msc_state_set_hp / set_ss, host_value one value of the family's
 * type.  ADD/REMOVE rewrite host_ss in place; SCORE_* write *score.  Synchronous;
 * latency-bound (one launch per call) -- the batched calls above are the fast path.
 */
int msc_value_op_single(msc_context *ctx, int family, uint32_t dim, int op, const float *host_hp,
                        void *host_ss, const void *host_value, float *score);

/* ---- relations (irm's per-cell data; relation/dataview.hpp:25-578) ------ */
/*
 * A relation reaches the kernels as a one-feature dataview whose rows are its
 * cells (dense: row-major order; compressed: the stored entries).  This turns the
 * per-dimension cluster assignments into the cell's block (= group) index:
 *   z_cell[c] = sum_d z_dev[d][index_d(c)] * prod_{e > d} ngroups[e]
 * (-1 when one of the cell's entities is unassigned).  positions_dev: null for a
 * dense relation, else uint32 [ncells][ndim] index tuples.  z_dev[d]: device int32
 * [shape[d]].  Asynchronous on the context's stream.
 */
int msc_relation_blocks(msc_context *ctx, uint32_t ndim, const uint64_t *shape,
                        const int32_t *const *z_dev, const uint32_t *ngroups,
                        const uint32_t *positions_dev, uint64_t ncells, int32_t *z_cell_dev);

/*
 * irm's slice reduction (what its assignment kernel does with relation::dataview::slice, dataview.hpp:265-578): entity e
 * of the domain on dimension `dim` is scored against every candidate cluster g of that domain by summing, over the
 * cells c of slice (dim, e), the cell's score against the block it would then lie in:
 *   out[e * ld_out + g] = sum_c scores[c * ld + g * cand_stride + off[c]]
 * scores_dev: the per-cell matrix msc_score_value wrote for the relation's cells ([ncells][ld], one column per block);
 * off_dev[c]: the block index of cell c with the candidate dimension's cluster set to 0 (msc_relation_blocks with an
 * all-zero assignment vector for `dim`; -1 = an entity of the cell is unassigned: the cell is skipped);
 * cand_stride: what one step of the candidate cluster adds to the block index (product of the cluster counts of the
 * later dimensions).  Dense relation: seg_dev = ids_dev = NULL and the slices are enumerated from shape; sparse: the
 * cells of entity e are ids_dev[seg_dev[e] .. seg_dev[e + 1]).  nent = entities scored (= shape[dim] for a dense one).
 * Sums are taken in double in a fixed order.  Asynchronous on the context's stream.
 */
int msc_relation_slice_scores(msc_context *ctx, const float *scores_dev, uint64_t ld, uint32_t ndim,
                              const uint64_t *shape, uint32_t dim, const uint32_t *seg_dev, const uint32_t *ids_dev,
                              const int32_t *off_dev, uint32_t ncand, uint32_t cand_stride, uint64_t nent,
                              float *out_dev, uint64_t ld_out);

#ifdef __cplusplus
}
#endif
#endif /* MICROSCOPES_HIP_H */
