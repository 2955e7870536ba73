// launchers.hpp -- host entry points of the kernel translation units.
#pragma once
#include <algorithm>

#include "msc_internal.hpp"

namespace msc {

// A function attribute (dynamic LDS beyond 64 KiB) is per device: `seen` is the launcher's own bit set of devices that
// have it.  True the first time the current device asks.
inline bool first_use_on_device(unsigned long long &seen) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return true;
  const unsigned long long bit = 1ull << dev;
  if (seen & bit) return false;
  seen |= bit;
  return true;
}

struct UnpackFeat {
  void *dst;
  uint8_t *dst_mask;     // may be null
  uint32_t offset;       // byte offset of the feature in the packed record
  uint32_t mask_offset;  // element offset of the feature in the mask record
  int32_t src_type, dst_type;
  uint32_t count;
  uint32_t pad;
};

// kernels_single.hip
struct MailboxHeader {
  int32_t family, dim, op, status;
  float score;
  uint32_t hp_off, ss_off, value_off;
};
int launch_value_op(hipStream_t stream, void *mailbox_dev, uint32_t dim, int family);

// every kernel translation unit that can report a device-side error (device_error.hpp): point its copy of the pointer at
// the device's pinned word; called once per device by msc_context_create
int bind_error_word_score(uint32_t *word_dev);
int bind_error_word_sweep(uint32_t *word_dev);
int bind_error_word_state(uint32_t *word_dev);

// kernels_score.hip
int launch_prepare(hipStream_t stream, const FeatDesc *feats_dev, uint32_t nfeat, uint32_t kpad, uint32_t value_slices);
int launch_dm_prepare(hipStream_t stream, const FeatDesc *feats_dev, int f, uint32_t dim, uint32_t kpad, uint32_t value_slices);
int launch_crp_prepare(hipStream_t stream, const uint32_t *cnt, uint32_t K, uint32_t kpad, float alpha,
                       float *crp);
int launch_entity_op(hipStream_t stream, const FeatDesc *feats_dev, int nfeat, uint32_t K, uint32_t kpad, uint64_t row,
                     uint32_t group, int sign, long long *cnt_acc, uint32_t *cnt_u32, float alpha, float *crp, int32_t *z_slot);
int launch_set_i32(hipStream_t stream, int32_t *dst, int32_t value);
int launch_loo_own(hipStream_t stream, int num_cus, bool heavy, bool staged, const FeatDesc *feats_dev, int nfeat, uint32_t K, uint32_t kpad, uint64_t row0,
                   uint64_t nrows, const int32_t *z, const float *crp, float *own);
int launch_gp_large_fix(hipStream_t stream, int num_cus, const FeatDesc *feats_dev, int f, uint32_t K,
                        uint32_t kpad, uint64_t row0, uint64_t nrows, const int32_t *z, float *out, uint64_t ld);
// which scoring kernel a state takes (abi.cpp decides from its feature list)
// (TILE_ROLES: a tile state whose first phase is lookup runs only and whose second phase is not empty -- with enough
// rows its workgroups split the phases between their waves, k_score_tile_roles)
enum { MSC_PATH_NICH1 = 0, MSC_PATH_TILE = 1, MSC_PATH_TILE_DM = 2, MSC_PATH_TILE_ROLES = 3, MSC_PATH_NICH_PACK = 4, MSC_PATH_LOOKUPS = 5 };
// k_score_nich1 launch shapes: a wave visits `visits` blocks of q consecutive rows (index 0 = the default)
struct Nich1Shape { int q, visits; };
constexpr int kNich1NumShapes = 8;
extern const Nich1Shape kNich1Shapes[kNich1NumShapes];
constexpr uint32_t kTailMaxGroups = 128;
constexpr uint64_t kNarrowMaxRows = 65536;   // views from this many rows on do not take the K <= 64 narrow tiling (abi.cpp narrow_lanes:
                                             // at 100k rows its sweeps already cost 2.5x the lane <-> row kernel's, at 20k they are level)
constexpr uint64_t kTailMinRows = 16384;     // fewer rows always stay with the tile kernels (lanes as groups)
// What a pass over `nrows` rows costs, in microseconds, on either kind of kernel -- only to choose between them.  The
// prices follow the PLAN (abi.cpp plan_cost; measured on one MI355X, 1M rows: tools/scans/c3_pieces.py --spec=...,
// tail_threshold.py, small_n.py, n_scan.py) -- with C3's prices for every plan (round 3) a state of lookup features only took the
// lane <-> row kernel up to 128 groups and ran 32 bool columns at 0.58 ms (K = 128) and 0.87 (K = 64) where the tile pass
// takes 0.34:
//   tile kernels: one workgroup of 128 rows per CU and round; a round costs ~6 us + 0.45 a staged lookup feature + 0.012 a
//   staged table row + 1.5 a nich feature (the role-split kernels overlap the two halves: half the nich share; the
//   nich-only kernels 1.3 a feature); C3: 50 us a round of a scoring pass, 55 of a fused sweep; at most 128 groups on the
//   role-split / nich-only kernels (PAIR mode, 256 rows a workgroup): kPairTileShare of that;
//   lane <-> row kernel: a launch of g groups takes ~2 us + g (0.5 + 0.03 a lookup feature + 0.057 a nich feature) per round
//   of 1024 rows a CU (two workgroups of 512); C3: 2 + 2.5 g (round 3 priced it 30 + 1.7 g: the same at 32-48 groups).
// (struct PlanCost: msc_internal.hpp -- the state keeps its plan's)
inline double tile_rounds_us(uint64_t workgroups, int num_cus, bool sweep, const PlanCost &pc = PlanCost()) {
  return (sweep ? pc.sweep_round_us : pc.tile_round_us) * (double)((workgroups + (uint64_t)num_cus - 1) / (uint64_t)num_cus);
}
inline double tail_rows_us(uint32_t groups, bool exact, uint64_t nrows, int num_cus, const PlanCost &pc = PlanCost()) {
  const uint32_t widest = exact ? 32u : 64u, nblk = (groups + widest - 1) / widest;      // (as launch_score_tail cuts them)
  const uint32_t per = ((groups + nblk - 1) / nblk + 15u) / 16u * 16u;
  // (a round of 1024 rows a CU that is not full costs less than a whole one -- the workgroups are 512 rows, most CUs get
  // none -- but not in proportion: sixteen dd columns, two exact launches: 0.066 ms at 262k rows, 0.050 at 100k, ~0.04 at 70k)
  const double frac = (double)nrows / (1024.0 * num_cus);
  const double rounds = frac >= 1.0 ? frac : std::max(0.6, 0.4 + 0.6 * frac);      // (100k rows, 0.38 of a round: 0.75 of its price)
  return (double)nblk * (pc.tail_fixed_us + pc.tail_group_us * per) * rounds;
}
// narrow_tail: score a partly filled last tile (<= kTailMaxGroups groups) with the narrow kernel, k_score_tail_rows (abi.cpp: the plan's
// first phase is lookup features only, the second plain nich features).  ok = false: no.
struct TailPlan {
  PlanCost cost;              // the plan's prices (set whether or not the narrow kernel may take the plan)
  bool ok = false;
  bool exact = true;          // the tile kernels' bits (two sums per group: score passes, where the row count picks the kernel);
                              // false: one sum per group, faster (sweeps: the choice of kernel follows the bound view's row count, not the call's)
  bool masked_nich = false;   // the first phase holds masked nich columns (the kernel's instantiation that evaluates them)
  bool dm = false;            // ... or dm features with their tables staged whole (the instantiation that looks them up)
  uint32_t max_rows = 0;      // the largest lookup table (rows a value may select)
  uint32_t pack_rows = 0;     // all lookup tables together
  float *pack = nullptr;      // scratch of pack_rows x 64 floats (the tail groups' tables, k_tail_pack), owned by the state
};
// the narrow kernel alone: groups [k0, K) of every row
int launch_score_tail(hipStream_t stream, int num_cus, const TailPlan &tp, const FeatDesc *feats_dev, int nfeat, int nsplit, uint32_t K,
                      uint32_t kpad, uint32_t k0, uint64_t row0, uint64_t nrows, const int32_t *z, const float *own, const float *crp,
                      float *out, uint64_t ld);
// two runs of 8-byte words that a fused sweep kernel zeroes on its way (the additive tables, which the accumulate
// pass that follows in a sweep step wants empty); all null / 0 when nothing follows
struct ZeroSpans {
  unsigned long long *a = nullptr;
  size_t na = 0;
  unsigned long long *b = nullptr;
  size_t nb = 0;
};
// (kernels_score.hip: the lane <-> row kernel with the draw in it)
int launch_sweep_rows(hipStream_t stream, int num_cus, const TailPlan &tp, const FeatDesc *feats_dev, int nfeat, int nsplit, uint32_t K,
                      uint32_t kpad, uint64_t row0, uint64_t nrows, uint64_t row_id0, int32_t *z, const float *own, const float *crp,
                      const uint64_t *rng, ZeroSpans zero);
int launch_score_pair_tail(hipStream_t stream, int num_cus, const FeatDesc *feats_dev, int nfeat, int nsplit, uint32_t K, uint32_t kpad,
                           uint64_t row0, uint64_t nrows, const int32_t *z, const float *own, const float *crp, float *tail, uint64_t ld);
int launch_score(hipStream_t stream, int num_cus, int path, const TailPlan &narrow_tail, int nich1_shape, const FeatDesc *feats_dev, int nfeat, int nsplit,
                 uint32_t K, uint32_t kpad, uint64_t row0, uint64_t nrows, const int32_t *z,
                 const float *own, const float *crp, float *out, uint64_t ld);

// which kernel INSTANTIATION the library chose for the most recent scoring pass (slot 0) / fused assignment pass (slot 1),
// spelled as rocprofv3 spells it ("k_score_tile_roles<false, false, false>"): bench.py and tools/ key the committed
// counter summaries by it (msc_last_kernel, include/microscopes_hip.h).  Process-wide, set by the launchers.
void note_kernel(int slot, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
const char *last_kernel(int slot);
inline const char *tf(bool b) { return b ? "true" : "false"; }

// kernels_sweep.hip  (return -2: shape not covered by this kernel)
int launch_sweep_nich1(hipStream_t stream, int num_cus, const FeatDesc *feats_dev, uint32_t K,
                       uint32_t kpad, uint64_t row0, uint64_t nrows, uint64_t row_id0, int32_t *z,
                       const float *own, const float *crp, const uint64_t *rng_dev, ZeroSpans zero);
bool tile_roles_enabled();
bool pair_mode_ok(int path, uint32_t K, bool few_rows);
#ifndef MSC_NICH_PACK_WAVES
#define MSC_NICH_PACK_WAVES 8
#endif
constexpr int kNichPackWaves = MSC_NICH_PACK_WAVES;   // waves a workgroup of the nich-only kernels (8: two a SIMD and 256 registers each -- no LDS ties them to sixteen)
#ifndef MSC_NICH_PACK_NC
#define MSC_NICH_PACK_NC 4
#endif
constexpr int kPackMaxLookups = 4;                      // lookup features (after fusing the bool columns) a plan may hold and still take the nich-only kernels
constexpr int kNichPackNC = MSC_NICH_PACK_NC;          // groups of the lane a block part of the nich-only kernels takes (256 registers a wave there)
constexpr double kPairTileShare = 0.62;     // what a pass of the role-split kernels costs in PAIR mode (<= 128 groups), of a full tile pass
int launch_sweep_mixed(hipStream_t stream, int num_cus, bool has_dm, bool roles_ok, bool pair, bool nich_only, bool lookups_only, const FeatDesc *feats_dev, int nfeat, int nsplit,
                       uint32_t K, uint32_t kpad, uint64_t row0, uint64_t nrows, uint64_t row_id0, int32_t *z,
                       const float *own, const float *crp, const uint64_t *rng, ZeroSpans zero);
int launch_sweep_roles_tail(hipStream_t stream, int num_cus, int kind, const FeatDesc *feats_dev, int nfeat, int nsplit, uint32_t K,
                            uint32_t kpad, uint64_t row0, uint64_t nrows, uint64_t row_id0, int32_t *z, const float *own,
                            const float *crp, const uint64_t *rng, ZeroSpans zero, const float *tail);
int sweep_niw1_max_groups(uint32_t dim);
// single nich feature beyond 1024 groups (lane <-> row, groups as scalar operands); `table`: device scratch of
// sweep_nich1_rows_table_floats(kpad) floats, rewritten by every call
size_t sweep_nich1_rows_table_floats(uint32_t kpad);
uint32_t sweep_nich1_rows_max_groups();
int launch_sweep_nich1_rows(hipStream_t stream, int num_cus, const FeatDesc *feats_dev, uint32_t K, uint32_t kpad,
                            uint64_t row0, uint64_t nrows, uint64_t row_id0, int32_t *z, const float *crp,
                            const uint64_t *rng, ZeroSpans zero, float *table);
int launch_sweep_niw1(hipStream_t stream, int num_cus, uint32_t dim, const FeatDesc *feats_dev, uint32_t K, uint32_t kpad,
                      uint64_t row0, uint64_t nrows, uint64_t row_id0, int32_t *z, const float *crp, const uint64_t *rng,
                      ZeroSpans zero);
int launch_narrow(hipStream_t stream, int num_cus, int lanes_per_row, uint32_t table_rows, bool sweep, const FeatDesc *feats_dev,
                  int nfeat, uint32_t K, uint32_t kpad, uint64_t row0, uint64_t nrows, const int32_t *z, const float *own,
                  const float *crp, float *out, uint64_t ld, uint64_t row_id0, int32_t *z_out, const uint64_t *rng,
                  ZeroSpans zero);
int launch_sample_rows(hipStream_t stream, int num_cus, const float *scores, uint64_t ld, uint32_t K,
                       uint64_t nrows, uint64_t row_id0, int32_t *z, const uint64_t *rng_dev);
int launch_rng_set(hipStream_t stream, uint64_t *rng_dev, uint64_t seed, uint64_t sweep);
int launch_rng_bump(hipStream_t stream, uint64_t *rng_dev);

// kernels_niw.hip
int launch_niw_prepare(hipStream_t stream, const FeatDesc *feats_dev, uint32_t f, uint32_t dim, uint32_t K,
                       uint32_t kpad);
int launch_niw_score_data(hipStream_t stream, const FeatDesc *feats_dev, uint32_t f, uint32_t K,
                          uint32_t kpad, float *out);
int launch_niw_score(hipStream_t stream, int num_cus, const FeatDesc *feats_dev, uint32_t f, uint32_t dim, uint32_t K,
                     uint32_t kpad, uint64_t row0, uint64_t nrows, const int32_t *z, bool accum, bool f32_fast,
                     double *qown /* [nrows] scratch, leave-one-out only */, float *out, uint64_t ld);
int launch_niw_accumulate(hipStream_t stream, int num_cus, const FeatDesc *feats_dev, uint32_t f, uint32_t K,
                          uint64_t row0, uint64_t nrows, const int32_t *z, int sign, uint32_t *scratch_dev, uint32_t dim);
int launch_niw_commit(hipStream_t stream, const FeatDesc *feats_dev, uint32_t f, uint32_t dim, uint32_t K,
                      uint32_t kpad, int to_raw);

// kernels_state.hip
int launch_accumulate(hipStream_t stream, int num_cus, const FeatDesc *feats_dev,
                      const FeatDesc *feats_host, int nfeat, uint32_t K, uint32_t kpad, uint64_t row0,
                      uint64_t nrows, const int32_t *z, int sign, long long *cnt_acc);
int launch_commit(hipStream_t stream, const FeatDesc *feats_dev, int nfeat, uint32_t kpad,
                  const long long *cnt_acc, uint32_t *cnt_u32);
int launch_commit_prepare(hipStream_t stream, const FeatDesc *feats_dev, int nfeat, uint32_t K, uint32_t kpad,
                          const long long *cnt_acc, uint32_t *cnt_u32, float alpha, float *crp, uint64_t *rng_bump,
                          uint32_t value_slices);
int launch_pack64(hipStream_t stream, bool unpack, long long *i64, size_t ni, double *f64, size_t nf, double *pack);
int launch_zero64(hipStream_t stream, void *a, size_t na, void *b, size_t nb);   // 8-byte words
int launch_stream_fill(hipStream_t stream, int num_cus, void *buf, size_t nbytes);   // the score kernels' store pattern, zeros
int launch_lift(hipStream_t stream, const FeatDesc *feats_dev, int nfeat, uint32_t kpad,
                long long *cnt_acc, const uint32_t *cnt_u32, int lift_cnt);
int launch_score_data(hipStream_t stream, const FeatDesc *feats_dev, int nfeat, uint32_t K,
                      uint32_t kpad, float *out);
int launch_relation_blocks(hipStream_t stream, uint32_t ndim, const uint64_t *shape, const int32_t *const *z_dev,
                           const uint32_t *ngroups, const uint32_t *positions_dev, uint64_t ncells, int32_t *out_dev);
int launch_relation_slice_scores(hipStream_t stream, const float *scores, uint64_t ld, uint32_t ndim, const uint64_t *shape,
                                 uint32_t dim, const uint32_t *seg_dev, const uint32_t *ids_dev, const int32_t *off_dev,
                                 uint32_t ncand, uint32_t cand_stride, uint64_t nent, float *out_dev, uint64_t ld_out);
int launch_dm_stats(hipStream_t stream, const uint32_t *col, uint64_t n, uint32_t dim, uint32_t *colmax_dev,
                    uint32_t *rowtot_dev);
int launch_col_max_u32(hipStream_t stream, const uint32_t *col, uint64_t n, uint32_t *out_dev);
int launch_mask_sentinel(hipStream_t stream, const void *col, const uint8_t *mask, uint64_t n, bool bytes, uint32_t sentinel, void *out);
int launch_pack_bits(hipStream_t stream, const void *const *cols, int m, uint32_t radix, uint64_t n, void *out);
int launch_pack_nich_x(hipStream_t stream, const float *const *cols_dev, uint32_t n2, uint32_t n2p, uint64_t n, float *out);
int launch_pack_look_idx(hipStream_t stream, const LookIdxSrc *src_dev, uint32_t nsrc, uint32_t l4, uint64_t n, uint32_t *out);
int launch_fuse_tables(hipStream_t stream, const FeatDesc *feats_dev, int nsplit, int nblocks, uint32_t kpad);
int launch_unpack(hipStream_t stream, const uint8_t *records, const uint8_t *mask, uint64_t nrows,
                  uint32_t rowsize, uint32_t maskrowsize, const void *feats_dev, uint32_t nfeat);

}  // namespace msc
