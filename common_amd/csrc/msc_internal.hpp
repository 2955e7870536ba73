// msc_internal.hpp -- host-side objects behind include/microscopes_hip.h and the
// plain-old-data descriptors shared with the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/microscopes_hip.h"

namespace msc {

constexpr int kWave = 64;          // gfx950 wavefront
constexpr int kGroupTile = 256;    // groups per k-tile: lane <-> 4 consecutive groups
constexpr unsigned kMaxDDDim = 128;
constexpr int kGrpRows = 128;   // table rows the LDS slot of the tile kernels holds (128 KiB): one feature group
constexpr unsigned kGpMaxTable = 1024;   // gp counts below this are exact table entries
constexpr float kNtFastGbps = 6650.f;    // probe fill rate from which a placed buffer takes non-temporal stores (msc_context::placed)

// ---- error plumbing --------------------------------------------------------
void set_error(const char *fmt, ...);
int fail(int code, const char *fmt, ...);

#define MSC_HIP(expr)                                                                    \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess)                                                                \
      return ::msc::fail(MSC_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                         __FILE__, __LINE__);                                            \
  } while (0)

#define MSC_REQUIRE(cond, ...)                                    \
  do {                                                            \
    if (!(cond)) return ::msc::fail(MSC_EINVAL, __VA_ARGS__);     \
  } while (0)

#define MSC_TRY(expr)                \
  do {                               \
    int _s = (expr);                 \
    if (_s != MSC_OK) return _s;     \
  } while (0)

inline uint32_t round_up(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }

// the CRP table (msc_state::logpc; kernels_score.hip crp_prepare_block): float offsets of the low halves
__host__ __device__ inline size_t crp_lo_cnt(uint32_t kpad) { return 2 * (size_t)kpad + 4; }     // lo of log(cnt)
__host__ __device__ inline size_t crp_lo_cntm1(uint32_t kpad) { return 3 * (size_t)kpad + 4; }   // lo of log(cnt - 1)
inline size_t crp_floats(uint32_t kpad) { return 4 * (size_t)kpad + 4; }

// ---- device-visible feature descriptor -------------------------------------
// One per state feature; lives in a small device array rebuilt when the (view,
// cols) binding changes.  All tables are struct-of-arrays with row stride kpad
// (ngroups rounded up to kGroupTile) so that lane l of a wave reads groups
// 4l..4l+3 of a k-tile with one 16-byte load.
//
// Derived score tables (float, filled by the prepare kernels from the raw tables):
//   bb / bbnc rows {s0, s1}                  log p(v=0), log p(v=1)
//   gp   rows {nse_hi, nse_lo, T[0..vcap)}   -stirlerr(a) as hi/lo, then the exact table log p(v) (family_math.hpp)
//   dd   rows {T[0..dim)}                    log p(v = i)
//   nich rows {mu_hi, mu_lo, c0, c1ln2, c1, c2}
//   niw  rows {c0, c1, A_loo, B_loo, C_loo, logdet_hi, logdet_lo} + the matrices below (kernels_niw.hip)
// Raw tables (the reference's own fields, u32 / f32):
//   bb   u32 {heads, tails}          bbnc: + f32 {p}
//   gp   u32 {count, sum}            f32 {log_prod}
//   dd   u32 {count_sum, counts[dim]}
//   nich u32 {count}                 f32 {mean, count_times_variance}
//   niw  u32 {count}                 f32 group-major {sum_x[d]} then {sum_xxT[d*d]} (see niw_* offsets)
//   bnb  u32 {count, sum}
//   dm   u32 {counts[dim]}           f32 {ratio}
// Additive tables (the all-reduce payload):
//   bb   i64 {heads, tails}
//   gp   i64 {count, sum}            f64 {log_prod}
//   dd   i64 {counts[dim]}
//   nich i64 {count}                 f64 {sum_x, sum_xx}
//   niw  i64 {count}                 f64 group-major {sum_x[d]}, {sum_xxT[d*d]}
//   bnb  i64 {count, sum}
//   dm   i64 {counts[dim]}           f64 {ratio}
// nich blocks (family_math.hpp): up to kNichBlock plain nich features of the plan's second phase that share c1
#ifndef MSC_NICH_BLOCK
#define MSC_NICH_BLOCK 4
#endif
constexpr int kNichBlock = MSC_NICH_BLOCK;
constexpr float kNichFarA = 32768.0f;                    // |a| beyond this: the row is "far" (see NichPlanInfo::xlim)
struct NichPlanInfo {                                    // per feature of the plan's second phase (FeatDesc::nich_info)
  float xlim;                                            // |x| <= xlim: no group's |a| = |s x - s mu| exceeds kNichFarA
  uint32_t blk_ok;                                       // at a block's FIRST feature: c1 ln2 is bit-equal over the block
};

// The role-split kernels' nich waves (score_block.hpp nich_phase_packed) read the second phase from three arrays made
// for them -- the plan's own descriptors are 300 bytes a feature and every number in them is a scalar load that waits for
// the one before (round 4: the nich waves issued 0.53 of their time, the rest were such chains):
//   * NichPos[n2p]: per POSITION i of the second phase (plan index - split; padded to a multiple of four) its block
//     (first position, length), the end of its LDS segment (the order of the sums follows the segments in every kernel),
//     and what the head kernel found (xlim, blk_ok: NichPlanInfo's two, copied here);
//   * the pack: (1 + kNichPackRows n2) rows of kpad floats -- row 0 the features' summed c0 (summed in plan order from 0,
//     as nich_c0_sum does), then per position mu_hi, mu_lo, c2, c1 ln2, c1 -- written by the head kernel at the head of
//     every scoring / sweep call from the tables as they stand: ONE buffer, scalar offsets;
//   * the x matrix: float [view rows][n2p], position-major inside a row -- a row's values of the second phase in one
//     64-byte stretch (abi.cpp nich_x_matrix: built when the view is bound, cached on the view).
// what a round of the tile kernels / a launch of the lane <-> row kernel costs for a plan, in microseconds (launchers.hpp
// tile_rounds_us / tail_rows_us; abi.cpp plan_groups computes it): only to choose between the kernels
struct PlanCost {
  double tile_round_us = 50.0, sweep_round_us = 55.0;    // (C3's, until a plan says otherwise)
  double tail_fixed_us = 30.0, tail_group_us = 1.7;
};
struct NichPos {                                         // (32-bit fields: a scalar load fetches no less)
  float xlim;
  uint32_t blk_ok;
  uint32_t blk;                                          // the position's block: first position | length << 16
  uint32_t seg_end;
};
constexpr uint32_t kNichPackRows = 5;                    // mu_hi, mu_lo, c2, c1 ln2, c1
struct FeatDesc {
  // -- what the tile kernels' lookup runs read per feature: one 32-byte block, one scalar load (score_block.hpp) --
  const void *col;         // bound dataview column (device), null until bound
  uint32_t grp_off;        // tile kernels, group plan (abi.cpp plan_groups): first row of this feature's table block
                           // inside the LDS slot
  uint32_t run_clamp;      // lookup kinds: the largest table row a value may select (dd: dim - 1, gp / bnb: staged rows - 1)
  uint32_t kind;           // MSC_KIND_*: which inner loop of the tile kernel scores this feature
  uint32_t run_end;        // lookup kinds: one past the last feature of the run of lookup features this one
                           // belongs to (within its group); generic: its own index
  uint32_t grp_rows;       // rows of the block staged in LDS (entries beyond are read from global)
  uint32_t grp_end;        // one past the last feature of this feature's group
  // ----------------------------------------------------------------------------------------------------------------
  int32_t family;
  uint32_t dim;
  int32_t col_type;        // msc_primitive_type of the bound column (value type of the family)
  uint32_t pad0;
  const uint8_t *mask;     // optional per-element mask column
  const float *hp;         // device copy of the hp block
  float *tab;              // derived score table rows
  uint32_t *raw_u32;
  float *raw_f32;
  long long *acc_i64;
  double *acc_f64;
  float *niw_w;            // niw, dim <= 32 only (the f32 MFMA kernel): [K][32][32] whitening matrix (row-major), scaled
  float *niw_b;            // niw, dim <= 32 only: [K][2][32] posterior mean, hi | lo floats
  double *niw_w64;         // niw: W_k = L^-1 sqrt(kn/(kn+1)) as the operand stream of the f64 MFMA kernel,
                           //      [K][niw_w_stream(dim)] (niw_w_index; kernels_niw.hip)
  double *niw_mu64;        // niw: W_k mu_k in the accumulator layout of that kernel, [K][niw_b_stream(dim)] (niw_b_index)
  double *niw_c64;         // niw only: [K][8] {c0, c1, A_loo, B_loo, C_loo}
  double aux;              // dd: sum of the alphas
  uint32_t vcap;           // gp, bnb: rows of the exact table (min(column max + 1, kGpMaxTable))
  uint32_t dm_rows;        // dm: rows of all its (hi, lo) count tables together (host: bind_dm_column), 0 otherwise
  const uint32_t *dm_meta;    // dm: [dim+1][2] per stage (device): {first table row, entries in the table};
                              //     count v of a stage occupies table rows first + 2v (hi) and first + 2v + 1 (lo)
  const uint32_t *dm_tot;     // dm: row totals of the bound column (device, owned by the view)
  // (the group plan's fields are the struct's head; 32-bit on purpose: the kernels read them with scalar loads, a
  // 16-bit field costs a vector load and a full wait)
  double *loo64;              // nich: per-group constants of the leave-one-out pass, [kpad][kNlooStride] (family_math.hpp)
  float *loo_tab;             // bb, gp, bnb, dd: score of value v against the group with one such value removed,
                              // [v][kpad] (k_prepare); the leave-one-out pass is a lookup for these families
  // the leave-one-out pass's own stage plan (abi.cpp plan_groups; k_loo_own_lds): consecutive features of the tile plan
  // whose leave-one-out blocks share the kernel's LDS slot
  uint32_t loo_off;           // first float of this feature's block inside the slot
  uint32_t loo_rows;          // rows of kpad floats staged (lookup kinds: run_clamp + 1 table rows; nich: 12 = 6 doubles
                              // per group, group-major); 0 = not staged: the feature reads global memory
  uint32_t loo_stage_end;     // one past the last feature of this feature's stage
  uint32_t pad1;
  // a masked lookup column (bb, bbnc, gp, bnb, dd) as the tile kernels want it: a copy in which a masked row holds the
  // index of the family's ZERO table row (bb 2, dd dim, gp / bnb vcap) -- abi.cpp bind_view makes it, plan_groups puts it
  // in the tile plan's copy of the descriptor (mask = null there), so a masked value is one more table row to the
  // lookup loops and nothing else; null when the column has no mask or the family no such row
  const void *col_sentinel;
  // FUSED bb / bbnc columns in the score / sweep kernels' plan (abi.cpp plan_groups): two to four bool columns read as
  // one byte column of their digits (digit j = member j's value -- a bit, or 0 / 1 / 2 = masked for columns with a mask;
  // the view keeps it, k_pack_bits) against one table of radix^n rows -- row i = the members' rows digit_j(i) summed in
  // member order -- that k_fuse_tables rebuilds into
  // `tab` at the head of every scoring / sweep call from the members' own tables (fuse_src): to every kernel it is a lookup
  // feature like any other, with a quarter of the reads and additions.  fuse_n = 0: an ordinary feature.
  const float *fuse_src[4];
  uint32_t fuse_n;
  uint32_t fuse_radix;     // 2: unmasked members; 3: masked ones (digit 2 = the member's zero row; three at a time, 27 rows)
  // nich BLOCKS of the plan's second phase (abi.cpp plan_groups; family_math.hpp): the plan's features
  // [blk_first, blk_end) are one block -- consecutive plain nich features with the same nu prior, at most kNichBlock, never
  // across an LDS feature group -- scored as ONE log1p of their product where the head kernel found their c1 ln2 rows
  // bit-equal (nich_info[0].blk_ok of the block's first feature), feature by feature otherwise.  nich_info: this
  // feature's own record (device; k_fuse_tables fills it at the head of every scoring / sweep call); null outside the
  // second phase.
  uint32_t blk_first, blk_end;
  NichPlanInfo *nich_info;
  // at the FIRST feature of the score / sweep plan's second phase, when the plan may take the role-split kernels: the
  // arrays of NichPos above (null otherwise); rn_n2 positions, rn_n2p = rn_n2 rounded up to four
  float *rn_pack;
  NichPos *rn_pos;
  const float *rn_x;
  uint32_t rn_n2, rn_n2p;
  // the accumulate pass's copy of a fused bb feature (abi.cpp plan_groups, desc_acc): the members' additive tables, in the
  // members' order -- one read of the byte column and of z feeds all of them (k_accumulate)
  long long *fuse_acc[4];
  // the first phase's LOOKUP INDEX MATRIX (round 5; abi.cpp look_idx_matrix, k_pack_look_idx): for the kernels whose first
  // phase is staged lookup runs only (role-split, lookups-only; PAIR mode too) every row's lookups as ready-made SLOT ROWS --
  // grp_off + the value clamped into the feature's block, one byte a feature, the features of an LDS feature group side by
  // side from a dword boundary on -- so a lookup wave fetches a group's indices for its rows with ONE load per group and
  // lane instead of a descriptor head, a value load, a clamp and a broadcast per feature.  Kept by the view (key: columns,
  // kinds, clamps and slot offsets).  lk_idx / lk_l4 at the plan's FIRST feature (null: the plan does not take those
  // kernels), lk_goff at every group's first feature.
  const uint32_t *lk_idx;  // uint32 [view rows][lk_l4] (+ 3 dwords of slack)
  uint32_t lk_l4;          // dwords per row
  uint32_t lk_goff;        // the group's first dword inside a row's record
};
// one feature of the index matrix as k_pack_look_idx reads it
struct LookIdxSrc {
  const void *col;
  uint32_t kind, clamp, grp_off, byte_at;   // byte_at: the feature's byte inside a row's record
};
constexpr uint32_t kLooSlotFloats = 32768;   // 128 KiB: one workgroup of k_loo_own_lds (1024 threads) per CU
constexpr int kLooStageFeats = 6;            // features a stage holds at most (their row values travel in registers)
enum { MSC_KIND_GENERIC = 0, MSC_KIND_LOOKUP_U8 = 1, MSC_KIND_LOOKUP_U32 = 2, MSC_KIND_LOOKUP_I32 = 3 };

// families whose score is a lookup of the row's count in an exact per-group table
__host__ __device__ inline bool is_count_family(int family) { return family == MSC_GP || family == MSC_BNB; }

inline uint32_t tab_rows(int family, uint32_t dim) {
  switch (family) {
    case MSC_BB: return 2;
    case MSC_BBNC: return 2;
    case MSC_GP: return 2 + kGpMaxTable;   // GP_T0 + table rows (family_math.hpp)
    case MSC_BNB: return 2 + kGpMaxTable;
    case MSC_DM: return 0;                  // sized when a column is bound (sum of the stages' tables)
    case MSC_DD: return dim;
    case MSC_NICH: return 6; // NICH_ROWS
    case MSC_NIW: return 8;  // NIW_ROWS + 1 (kernels_niw.hip)
    default: return 0;
  }
}
// (nich: kNlooStride doubles per group, group-major -- a row's leave-one-out pass reads ONE group's six constants: side
// by side they are 48 bytes, one cache line mostly; family_math.hpp NLOO_*)
constexpr uint32_t kNlooStride = 6;
inline uint32_t loo_rows(int family) { return family == MSC_NICH ? kNlooStride : 0u; }
// (+ 1: the zero row a masked value selects, FeatDesc::col_sentinel)
inline uint32_t loo_tab_rows(int family, uint32_t dim) {
  return family == MSC_BB ? 3u : (family == MSC_GP || family == MSC_BNB) ? kGpMaxTable + 1u : family == MSC_DD ? dim + 1u : 0u;
}
inline uint32_t raw_u32_rows(int family, uint32_t dim) {
  switch (family) {
    case MSC_BB: return 2;
    case MSC_BBNC: return 2;
    case MSC_GP: return 2;
    case MSC_BNB: return 2;
    case MSC_DM: return dim;
    case MSC_DD: return 1 + dim;
    case MSC_NICH: return 1;
    case MSC_NIW: return 1;
    default: return 0;
  }
}
// float raw rows with stride kpad (niw's vector fields are handled separately)
inline uint32_t raw_f32_rows(int family) {
  switch (family) {
    case MSC_GP: return 1;
    case MSC_BBNC: return 1;   // p
    case MSC_DM: return 1;     // ratio
    case MSC_NICH: return 2;
    default: return 0;
  }
}
inline uint32_t acc_i64_rows(int family, uint32_t dim) {
  switch (family) {
    case MSC_BB: return 2;
    case MSC_BBNC: return 2;
    case MSC_GP: return 2;
    case MSC_BNB: return 2;
    case MSC_DM: return dim;
    case MSC_DD: return dim;
    case MSC_NICH: return 1;
    case MSC_NIW: return 1;
    default: return 0;
  }
}
inline uint32_t acc_f64_rows(int family) {
  switch (family) {
    case MSC_GP: return 1;
    case MSC_DM: return 1;
    case MSC_NICH: return 2;
    default: return 0;
  }
}
inline int value_type_of(int family) {
  switch (family) {
    case MSC_BB: return MSC_TYPE_B;
    case MSC_BBNC: return MSC_TYPE_B;
    case MSC_GP: return MSC_TYPE_U32;
    case MSC_BNB: return MSC_TYPE_U32;
    case MSC_DM: return MSC_TYPE_I32;
    case MSC_DD: return MSC_TYPE_I32;
    case MSC_NICH: return MSC_TYPE_F32;
    case MSC_NIW: return MSC_TYPE_F32;
    default: return MSC_TYPE_B;
  }
}
size_t primitive_size(int t);

// rows of a niw feature's float table (k_niw_prepare fills them) and the padded width of its W matrices
enum { NIW_C0 = 0, NIW_C1 = 1, NIW_A_LOO = 2, NIW_B_LOO = 3, NIW_C_LOO = 4, NIW_LOGDET_HI = 5, NIW_LOGDET_LO = 6,
       NIW_ROWS = 7 };   // row NIW_ROWS holds the prior's ln det Psi (hi, lo) in its first two slots
constexpr int kNiwPad = 32;                 // the f32 kernel's padded matrix width (dim <= 32)
constexpr unsigned kMaxNiwDim = 128;        // what the prepare kernel's packed LDS triangles hold (2 x 66 KB)

// The f64 contraction works on 16-blocks: component block b (rows 16b .. 16b+15 of the lower-triangular W) meets
// feature blocks s4 = 0 .. b only.  A chunk = one (b, s4) pair = four MFMA steps; step e of the chunk contracts
// features 16 s4 + 4 e + kk (kk = lane / 16).  Per group the chunks are stored in the order the kernel walks them,
// 64 lanes x 4 doubles each, so a chunk is one coalesced 2 KiB read per wave.
__host__ __device__ inline uint32_t niw_blocks(uint32_t d) { return (d + 15u) / 16u; }
__host__ __device__ inline uint32_t niw_chunks(uint32_t d) { return niw_blocks(d) * (niw_blocks(d) + 1u) / 2u; }
__host__ __device__ inline size_t niw_w_stream(uint32_t d) { return (size_t)niw_chunks(d) * 256u; }   // doubles per group
__host__ __device__ inline size_t niw_b_stream(uint32_t d) { return (size_t)niw_blocks(d) * 256u; }
// where W[i][j] (j <= i) sits in a group's stream: lane (c = i % 16, kk = j % 4), slot e = (j % 16) / 4 of chunk (i / 16, j / 16)
__host__ __device__ inline size_t niw_w_index(uint32_t i, uint32_t j) {
  const uint32_t b = i >> 4, s4 = j >> 4;
  return (size_t)(b * (b + 1u) / 2u + s4) * 256u + (size_t)((i & 15u) + 16u * (j & 3u)) * 4u + ((j & 15u) >> 2);
}
// where (W mu)[i] sits: accumulator register r = (i % 16) / 4 of the lanes with kk = i % 4 (every c holds the same value;
// this is the c = 0 copy)
__host__ __device__ inline size_t niw_b_index(uint32_t i) {
  return (size_t)(i >> 4) * 256u + (size_t)(16u * (i & 3u)) * 4u + ((i & 15u) >> 2);
}

}  // namespace msc

// ---- opaque objects ---------------------------------------------------------
struct msc_context {
  int device = 0;
  hipStream_t stream = nullptr;
  int num_cus = 256;
  // k_score_nich1 launch shape (index into kNich1Shapes) per output buffer: which shape suits the write stream
  // depends on where the buffer lies (abi.cpp run_score); most recent first, at most 16 buffers remembered
  struct ShapeEntry { const void *out; uint64_t nrows; uint32_t K; int shape; int plain_stores; };   // plain_stores: -1 = not timed
  std::vector<ShapeEntry> nich1_shapes;
  // buffers this context placed (msc_device_alloc >= 64 MiB, msc_device_alloc_probed) and how their probe went: a buffer
  // that took the probe's NON-TEMPORAL fill at kNtFastGbps or better takes the single-nich pass's non-temporal stores at
  // 0.85-0.88 of the HBM roof; every other buffer (the probe's lower bands, a caller's own) is written with plain stores --
  // 0.78-0.83 whatever the placement, where non-temporal ones run 0.69-0.76 (profiles/r04_store_policy.txt)
  struct Placed { const void *base; size_t size; bool nt_fast; };
  std::vector<Placed> placed;
  // a stream of the library's own on which sweep steps are recorded (the caller's may be the null stream, which
  // cannot capture); nothing ever executes on it
  hipStream_t record_stream = nullptr;
  // buffers msc_device_alloc_probed mapped from separately created physical chunks (hipMemCreate + hipMemMap); freed
  // through msc_device_free
  struct VmmAlloc { void *va; size_t size; std::vector<hipMemGenericAllocationHandle_t> handles; };
  std::vector<VmmAlloc> vmm;
  // the candidates of the most recent placed allocation (msc_device_alloc >= 64 MiB, msc_device_alloc_probed): fill
  // rates in GB/s and the index kept (msc_device_alloc_stats)
  std::vector<float> last_alloc_rates;
  uint32_t last_alloc_chosen = 0;
  // pinned, device-mapped mailbox for msc_value_op_single
  void *mailbox_host = nullptr;
  void *mailbox_dev = nullptr;
  size_t mailbox_bytes = 0;
  // msc_context_synchronize: a word of pinned memory the stream writes a sequence number into (hipStreamWriteValue32)
  // and the host watches, instead of sleeping in hipStreamSynchronize
  uint32_t *sync_word_host = nullptr;
  void *sync_word_dev = nullptr;
  uint32_t sync_seq = 0;
  bool sync_word_ok = true;                // cleared when the stream operation is refused: plain stream waits from then on
  // the device's error word {code, detail} (device_error.hpp), pinned and shared by every context on the device
  volatile uint32_t *err_host = nullptr;
};

struct msc_dataview {
  msc_context *ctx = nullptr;
  uint64_t serial = 0;                   // unique per view: a state's binding is keyed on it, not on the address
  uint64_t nrows = 0;
  std::vector<msc_runtime_type> types;   // per feature, after conversion
  std::vector<void *> cols;              // device columns
  std::vector<void *> masks;             // device mask columns or null
  std::vector<void *> owned;             // allocations to free
  mutable std::vector<long long> col_max;  // lazily computed maximum of uint32 columns (-1 = unknown)
  mutable std::vector<std::vector<uint32_t>> dm_max;  // dm columns: maxima of each category and of the row totals (lazy)
  mutable std::vector<uint32_t *> dm_tot;             // dm columns: row totals (device, owned)
  mutable std::vector<void *> owned_lazy;
  // copies of a column converted to another primitive type with runtime_cast semantics, made the first time a state
  // binds the column to a model whose value type differs (abi.cpp column_as): per column, (type, device copy)
  mutable std::vector<std::vector<std::pair<int, const void *>>> converted;
  // masked lookup columns with the mask folded in as a sentinel value (abi.cpp sentinel_column): per column,
  // ((element type, sentinel), device copy)
  mutable std::vector<std::vector<std::pair<std::pair<int, uint32_t>, const void *>>> sentinels;
  // byte columns holding the bits of two to four bool columns (key: the member columns' device pointers), made when a
  // state's plan fuses them (abi.cpp plan_groups, k_pack_bits)
  mutable std::vector<std::pair<std::vector<const void *>, const void *>> packed_bits;
  // float [nrows][n2p] of the columns in the key, position-major (msc::NichPos; abi.cpp nich_x_matrix)
  mutable std::vector<std::pair<std::vector<const void *>, const float *>> nich_x;
  // uint32 [nrows][l4] lookup index matrices (msc::FeatDesc::lk_idx; abi.cpp look_idx_matrix); key: per feature
  // {column, kind, clamp, slot offset, byte} flattened
  mutable std::vector<std::pair<std::vector<uint64_t>, const uint32_t *>> look_idx;
};

struct msc_feature_host {
  int family = 0;
  uint32_t dim = 0;
  std::vector<float> hp;
  float *hp_dev = nullptr;
  float *tab = nullptr;
  uint32_t *raw_u32 = nullptr;
  float *raw_f32 = nullptr;
  float *niw_raw = nullptr;     // [K][d + d*d] float
  double *loo64 = nullptr;      // nich: leave-one-out constants
  float *loo_tab = nullptr;     // bb, gp, bnb, dd: leave-one-out tables
  float *niw_w = nullptr, *niw_b = nullptr;
  double *niw_w64 = nullptr, *niw_mu64 = nullptr, *niw_c64 = nullptr;
  size_t i64_off = 0, i64_len = 0;   // slices of the state's reduce buffers (elements)
  size_t f64_off = 0, f64_len = 0;
  size_t tab_rows_cap = 0;      // dm: rows (+4) the table buffer holds; it is (re)sized when a column is bound
  uint32_t *dm_meta_dev = nullptr;   // dm: [dim+1][2] stage tables (FeatDesc::dm_meta)
  std::vector<uint32_t> dm_meta;     // host copy
  bool raw_valid = true;        // raw tables hold the truth
  bool additive_valid = false;  // additive tables are in sync with raw
  bool derived_valid = false;   // score tables are in sync with raw
};

struct msc_state {
  msc_context *ctx = nullptr;
  uint32_t nfeat = 0, K = 0, kpad = 0;
  float alpha = 1.f;
  std::vector<msc_feature_host> feats;
  long long *red_i64 = nullptr;   // [cnt[kpad] | feature slices]
  double *red_f64 = nullptr;
  size_t n_i64 = 0, n_f64 = 0;
  double *red_pack = nullptr;     // both tables as one float64 buffer (msc_state_reduce_pack), made at the first call
  uint64_t sweep_rows_hint = 0;   // msc_state_set_sweep_rows: the rows of the WHOLE a sharded sweep's kernel choice goes by
  uint32_t *cnt_u32 = nullptr;    // group sizes (group_manager counts), [kpad]
  float *logpc = nullptr;         // log pseudocount per group, [kpad] (+ loo variants, see prepare)
  bool cnt_additive_valid = false;
  bool crp_valid = false;
  msc::FeatDesc *desc_dev = nullptr;
  std::vector<msc::FeatDesc> desc_host;
  // the same descriptors in the order the tile kernels walk them (abi.cpp plan_groups): every feature but the
  // unmasked nich ones, in the caller's order, then the unmasked nich features (from index tile_split on)
  msc::FeatDesc *desc_tile_dev = nullptr;
  std::vector<msc::FeatDesc> desc_tile_host;
  uint32_t tile_split = 0;
  // the same plan with runs of unmasked bb / bbnc columns fused (FeatDesc::fuse_*): what the score / sweep kernels walk;
  // the leave-one-out pass keeps the plan above (its double sum over features stays term by term)
  msc::FeatDesc *desc_fuse_dev = nullptr;
  std::vector<msc::FeatDesc> desc_fuse_host;
  uint32_t fuse_nfeat = 0, fuse_split = 0;
  bool fuse_any = false;
  float *fuse_tab = nullptr;          // the fused tables, 32 rows of kpad floats a fused feature (16 or 27 used)
  size_t fuse_tab_floats = 0;
  // what msc_accumulate walks: the fused bb features first (each feeds its members' tables from one byte column), then
  // every feature no fused one covers, as the caller gave it (masks and all)
  msc::FeatDesc *desc_acc_dev = nullptr;
  std::vector<msc::FeatDesc> desc_acc_host;
  msc::NichPlanInfo *nich_info = nullptr;   // [nfeat]: per feature of the plans' second phase (FeatDesc::nich_info)
  float *rn_pack = nullptr;                 // the role-split kernels' second phase (msc::NichPos): (1 + 5 nfeat) x kpad floats
  msc::NichPos *rn_pos = nullptr;           // [nfeat rounded up to four]
  bool nich_blocks_any = false;       // the plan has a nich block of two or more features
  const msc_dataview *bound_view = nullptr;
  uint64_t bound_serial = 0;
  std::vector<uint32_t> bound_cols;
  std::vector<void *> owned;
  float *scratch = nullptr;       // score chunk for the generic sweep path
  float *tail_scores = nullptr;   // 64 floats per row: the groups beyond the first tile (k_score_tail_rows -> k_sweep_tile_roles<true>)
  size_t tail_floats = 0;
  float *own = nullptr;           // per-row leave-one-out values (k_loo_own)
  bool tile_roles_ok = false;     // plan_groups: lookup runs only before tile_split, unmasked nich features after it
  msc::PlanCost plan_cost;            // plan_groups: what a round of the tile kernels / a launch of the lane <-> row kernel costs for THIS plan
  bool tile_lookups_only = false;     // plan_groups: staged lookup features and nothing else (k_score_lookups)
  bool tile_nich_only = false;    // plan_groups: no first phase at all, two or more plain nich features (k_score_nich_pack)
  bool tile_narrow_tail_ok = false;   // plan_groups: a partly filled last tile may take k_score_tail_rows
  uint32_t tail_max_rows = 0, tail_pack_rows = 0;   // the lookup tables of the tile plan's first phase: the largest, all together
  bool tail_masked_nich = false;      // ... and masked nich columns among them (evaluated in place, under the row's mask)
  bool tail_dm = false;               // ... or dm features with their tables staged whole
  float *tail_pack = nullptr;         // k_tail_pack's output: tail_pack_rows x 64 floats (grown on demand)
  size_t tail_pack_floats = 0;
  uint32_t loo_staged = 0;        // plan_groups: features whose leave-one-out block k_loo_own_lds stages in LDS
  float *rows_table = nullptr;    // k_sweep_nich1_rows: per-group constants as scalar operands (single nich, K > 1024)
  size_t own_cap = 0;
  uint32_t *colmax_dev = nullptr;
  size_t scratch_floats = 0;
  // (seed, sweep index) of the sampling kernels, device-resident (kernels_sweep.hip); the host tracks what it holds
  uint64_t *rng_dev = nullptr;
  uint64_t rng_seed = 0, rng_sweep = 0;
  bool rng_valid = false;
  bool rng_bump_pending = false;       // msc_sweep_step_begin left the increment of the sweep index to msc_state_commit_reduce
  uint64_t rng_next_sweep = 0;
  // a whole sweep step (assign + accumulate + commit) captured as a graph for its steady state (abi.cpp msc_sweep_step)
  struct StepGraph {
    hipGraphExec_t exec = nullptr;
    const void *view = nullptr, *z = nullptr;
    uint64_t view_serial = 0, row0 = 0, nrows = 0, row_id0 = 0;
    float alpha = 0.f;
    std::vector<uint32_t> cols;
    std::vector<uint8_t> flags;        // validity flags of the state at step entry
    int seen = 0;                      // eager steps with this key so far
    bool disabled = false;
    uint64_t n_eager = 0, n_replayed = 0;   // steps run either way (msc_sweep_step_stats)
  } step_graph;
  double *niw_qown = nullptr;        // q of every row's own group (niw leave-one-out), [niw_qown_cap]
  size_t niw_qown_cap = 0;
  int32_t *one_z = nullptr;          // a one-entry assignment vector (msc_entity_op's general path)
  uint32_t *niw_scratch = nullptr;   // row bucketing for niw accumulate: 2 K + 1 + rows uint32
  size_t niw_scratch_len = 0;
};
