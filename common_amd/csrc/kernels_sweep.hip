// kernels_sweep.hip -- the fused Gibbs assignment step (SURVEY 3.2 as one
// data-parallel pass): per row, leave-one-out scores against every group + the CRP
// term, then util::sample_discrete_log (util.hpp:125-156) -- without ever writing
// the [N, K] score matrix.  A wave owns the row; scores stay in registers; the
// max / sum / prefix over groups are wavefront reductions.
//
//   k_sweep_nich1<G>  one NICH feature, K <= 64*G: every per-group constant of the
//                     whole table lives in VGPRs (lane l owns groups G*l .. G*l+G-1),
//                     rows stream through; no LDS, no HBM traffic besides x, z.
//   k_sweep_tile<R>   any feature list, K <= 256: the workgroup tile of the batched kernel
//                     (tables staged through LDS), sampled from registers.
//   k_sample_rows     fallback for larger tables: samples rows of a score chunk that
//                     msc_score_value (leave-one-out + prior) wrote to scratch.
#include "family_math.hpp"
#include "launchers.hpp"
#include "score_block.hpp"

namespace msc {

// ---- wavefront primitives (64 lanes), on the DPP cross-lane path (no LDS traffic) ---------
// row_shr:n shifts within each row of 16 lanes; row_bcast:15 / :31 carry the running value of
// the previous row(s) into rows {1,3} / {2,3}.  Lanes without a source keep `identity`.
template <int CTRL, int ROW_MASK>
MSC_DEV float dpp_f32(float identity, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
// Both reductions are written as the DPP instruction itself: "v_op_dpp v, v, v <ctrl>" computes op(moved v, v) in the
// lanes that have a source and leaves the others alone, which is what a shift-reduction wants.  From the builtins
// hipcc makes a v_mov_b32_dpp with a fill value, the op, and (for fmaxf, IEEE mode) a canonicalising v_max on top:
// 3 instructions a step instead of 1.  s_nop 1 = the two wait states a DPP read needs after a VALU write of its source.
// (one asm block per reduction, so that exactly the needed wait states sit between the steps)
#define MSC_DPP_REDUCE(op, v)                                                     \
  asm volatile("s_nop 1\n\t" op " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"      \
               "s_nop 1\n\t" op " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"      \
               "s_nop 1\n\t" op " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"      \
               "s_nop 1\n\t" op " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"      \
               "s_nop 1\n\t" op " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"   \
               "s_nop 1\n\t" op " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"   \
               "s_nop 1"                                                          \
               : "+v"(v))
MSC_DEV float wave_max(float v) {
  MSC_DPP_REDUCE("v_max_f32_dpp", v);            // row_bcast:15 -> rows 1, 3; row_bcast:31 -> rows 2, 3
  return lane_bcast(v, 63);                      // lane 63 now holds the maximum over the wave
}
MSC_DEV float wave_incl_scan(float v, int) {
  MSC_DPP_REDUCE("v_add_f32_dpp", v);
  return v;
}

// Inverse-CDF draw over K groups laid out G per lane in k order (lane l: k = G*l + j).
// s[j] = -inf for k >= K.  Mirrors scores_to_probs + sample_discrete: subtract the max,
// exponentiate, and return the first k whose running sum reaches dart * total; K-1 if
// rounding lets the dart fall off the end (util.hpp:155).
// LOG2: the scores are already in units of log2 (the caller folded log2(e) into its constants).
// BOUND: `bound` is a wave-uniform upper bound of the scores the caller knows without looking at them; it stands in
// for the maximum (6 DPP steps + 4 compares saved per row) whenever the total it leads to is a comfortably normal float,
// and the exact maximum is taken otherwise (an outlier row far below the bound, or a bound that was not one).
// SHIFTED (with BOUND and LOG2): the caller already subtracted the bound from the scores (it folded it into constants)
template <int G, bool LOG2 = false, bool BOUND = false, bool SHIFTED = false>
MSC_DEV int sample_from_scores(const float (&s)[G], float u01, int lane, uint32_t K, float bound = 0.f) {
  float p[G], sum = 0.f, incl = 0.f, total = 0.f;
  bool done = false;
  if (BOUND) {
    sum = 0.f;
#pragma unroll
    for (int j = 0; j < G; j++) {
      p[j] = __builtin_amdgcn_exp2f(SHIFTED ? s[j] : LOG2 ? s[j] - bound : (s[j] - bound) * 1.44269504088896340736f);
      sum += p[j];
    }
    incl = wave_incl_scan(sum, lane);
    total = lane_bcast(incl, 63);
    done = total > 0x1p-60f && total < 0x1p100f;        // (false for NaN too)
  }
  if (!done) {
    float m = s[0];
#pragma unroll
    for (int j = 1; j < G; j++) m = fmaxf(m, s[j]);
    m = wave_max(m);
    sum = 0.f;
#pragma unroll
    for (int j = 0; j < G; j++) {
      p[j] = __builtin_amdgcn_exp2f(LOG2 ? s[j] - m : (s[j] - m) * 1.44269504088896340736f);   // exp(-inf) = 0
      sum += p[j];
    }
    incl = wave_incl_scan(sum, lane);
    total = lane_bcast(incl, 63);
  }
  const float dart = u01 * total;
  // the running sum is monotone, so the first entry that reaches the dart is found by counting the
  // entries that do not; entries with k >= K have p = 0 and cannot be the first
  float c = incl - sum;
  int nmiss = 0;
  if constexpr (G <= 4) {
#pragma unroll
    for (int j = 0; j < G; j++) {
      c += p[j];
      nmiss += c < dart ? 1 : 0;
    }
  } else {
    // 8 or 16 entries per lane: descend through the block sums instead of walking the entries (3 instructions per
    // entry otherwise): quads, then pairs, then the two entries -- every level asks "does everything before the
    // right half still miss?".  Sums are formed left to right inside a block, so a block's partial sums are the
    // walk's own values and the count is the walk's count.
    float q[G / 4];
#pragma unroll
    for (int b = 0; b < G / 4; b++) q[b] = ((p[4 * b] + p[4 * b + 1]) + p[4 * b + 2]) + p[4 * b + 3];
    int blk = 0;
#pragma unroll
    for (int b = 0; b < G / 4; b++) {                      // 2 or 4 quads: count those that miss entirely
      const float cq = c + q[b];
      const bool miss = (blk == b) && cq < dart;
      c = miss ? cq : c;
      blk += miss ? 1 : 0;
    }
    nmiss = 4 * blk;
    if (blk < G / 4) {
      float e0 = p[0], e1 = p[1], e2 = p[2];
#pragma unroll
      for (int b = 1; b < G / 4; b++) {
        e0 = blk == b ? p[4 * b] : e0;
        e1 = blk == b ? p[4 * b + 1] : e1;
        e2 = blk == b ? p[4 * b + 2] : e2;
      }
      c += e0;
      const bool m0 = c < dart;
      c += e1;
      const bool m1 = m0 && c < dart;
      c += e2;
      const bool m2 = m1 && c < dart;
      nmiss += (m0 ? 1 : 0) + (m1 ? 1 : 0) + (m2 ? 1 : 0);   // (the block reaches the dart, so its fourth entry does if these miss)
    }
  }
  const unsigned long long hit = __builtin_amdgcn_ballot_w64(nmiss < G);
  if (hit == 0ull) return (int)K - 1;
  const int l = (int)__builtin_ctzll(hit);
  const int k = G * l + lane_bcast(nmiss, l);
  return k < (int)K ? k : (int)K - 1;
}

// The same draw over a full 256-group tile (4 entries per lane) FOLLOWED by a tail of up to 64 TAILP groups (4 per lane on
// lanes 0 .. 16 TAILP - 1; -inf elsewhere and beyond K): one maximum, two running sums, the dart thrown at their total, the tile searched
// first -- the CDF order of sample_discrete (k ascending).
template <int TAILP>
MSC_DEV int sample_tile_and_tail(const float (&sm)[4], const float (&st)[4], float u01, int lane, uint32_t K) {
  float m = fmaxf(fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3])), fmaxf(fmaxf(st[0], st[1]), fmaxf(st[2], st[3])));
  m = wave_max(m);
  float pm[4], pt[4], summ = 0.f, sumt = 0.f;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    pm[j] = __builtin_amdgcn_exp2f((sm[j] - m) * 1.44269504088896340736f);   // exp(-inf) = 0
    pt[j] = __builtin_amdgcn_exp2f((st[j] - m) * 1.44269504088896340736f);
    summ += pm[j];
    sumt += pt[j];
  }
  const float inclm = wave_incl_scan(summ, lane), totm = lane_bcast(inclm, 63);
  const float inclt = wave_incl_scan(sumt, lane), tott = lane_bcast(inclt, 63);
  const float dart = u01 * (totm + tott);
  float c = inclm - summ;
  int nmiss = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    c += pm[j];
    nmiss += c < dart ? 1 : 0;
  }
  unsigned long long hit = __builtin_amdgcn_ballot_w64(nmiss < 4);
  if (hit != 0ull) {
    const int l = (int)__builtin_ctzll(hit);
    return 4 * l + lane_bcast(nmiss, l);                 // (< 256 <= K)
  }
  c = totm + (inclt - sumt);
  nmiss = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    c += pt[j];
    nmiss += c < dart ? 1 : 0;
  }
  hit = __builtin_amdgcn_ballot_w64(nmiss < 4 && lane < 16 * TAILP);
  if (hit == 0ull) return (int)K - 1;
  const int l = (int)__builtin_ctzll(hit);
  const int k = kGroupTile + 4 * l + lane_bcast(nmiss, l);
  return k < (int)K ? k : (int)K - 1;
}


// ---------------------------------------------------------------------------
// (G = 16 asks for three waves per SIMD: at two the dependent transcendental chains of one wave pair leave the vector
//  pipe idle; the register allocator then works within 168 VGPRs)
template <int G>
__global__ __launch_bounds__(256, G == 16 ? 3 : 1) void k_sweep_nich1(const FeatDesc *__restrict__ feats, uint32_t K,
                                                      uint32_t kpad, uint64_t row0, uint64_t nrows,
                                                      uint64_t row_id0, int32_t *__restrict__ z,
                                                      const float *__restrict__ own,
                                                      const float *__restrict__ crp,
                                                      const uint64_t *__restrict__ rng, int chunk_rows,
                                                      ZeroSpans zero) {
  const uint64_t seed = rng[0], sweep = rng[1];      // (device-resident so that a captured graph can replay the step)
  zero_spans(zero);
  const FeatDesc fd = feats[0];
  const int lane = threadIdx.x & 63;
  const uint32_t kb = (uint32_t)(G * lane);
  // per-group constants in registers for the whole kernel: s mu (hi, lo), s = sqrt(c2), c1, and c0' (below)
  float mh[G], ml[G], c0[G], c1[G], c2[G];
  bool empty[G];
#pragma unroll
  for (int j = 0; j < G; j++) {
    const uint32_t k = kb + j;
    const size_t kc = k < kpad ? k : 0;                  // (kpad >= 64 G only when K needs it)
    mh[j] = fd.tab[(size_t)NICH_MU_HI * kpad + kc];
    ml[j] = fd.tab[(size_t)NICH_MU_LO * kpad + kc];
    c0[j] = fd.tab[(size_t)NICH_C0 * kpad + kc];
    c1[j] = fd.tab[(size_t)NICH_C1 * kpad + kc];
    c2[j] = fd.tab[(size_t)NICH_C2 * kpad + kc];
    const float lcj = crp[kc];
    empty[j] = __builtin_isinf(lcj);
    c0[j] += empty[j] ? 0.f : lcj;                        // + log count (an empty group's prior is added below)
  }
  const float le0 = crp[2 * (size_t)kpad], le1 = crp[2 * (size_t)kpad + 1];
  // Sampling only needs the scores up to what exp2 sees, so everything is moved to log2 units once per lane:
  // score = c0' - c1 w with w = log2(1 + t) (the hardware logarithm plus the log1p remainder scaled by log2 e) and
  // c0' = (c0 + prior) log2e - bound.  The prior of an empty group, log(alpha / n_empty), is folded in (when that is
  // -inf no group is empty and the value is never used), and so is minus the wave-uniform bound that stands in for the
  // row maximum in the draw: a row's unnormalised probability is exp2(score) with nothing in between.  A row that is
  // its group's only member leaves one more empty group behind: every empty group's prior moves by
  // dle = log2(n_empty / (n_empty + 1)) for that row (wave-uniform side path, rare).
  constexpr float kLog2e = 1.44269504088896340736f;
  const bool any_empty = !__builtin_isinf(le0);
  const float dle = any_empty ? (le1 - le0) * kLog2e : 0.f;
  float c0s[G];
  unsigned emask = 0u;                                  // bit j: group kb + j is empty
  float bound = -INFINITY;
#pragma unroll
  for (int j = 0; j < G; j++) {
    if (empty[j] && kb + j < K) emask |= 1u << j;
    c0s[j] = (kb + j >= K) ? -INFINITY : (c0[j] + (empty[j] ? le0 : 0.f)) * kLog2e;
    if (kb + j < K) bound = fmaxf(bound, c0s[j]);
  }
  // no score exceeds its c0' (log1p >= 0); the own group's leave-one-out value, a masked row's prior-only scores and a
  // singleton row's empty groups may, by a little: 16 bits of headroom, and sample_from_scores checks
  bound = wave_max(bound) + 16.f;
#pragma unroll
  for (int j = 0; j < G; j++) c0s[j] -= bound;
  const float *xcol = reinterpret_cast<const float *>(fd.col) + row0;
  // a wave takes chunk_rows (<= 64) rows at a time, one per lane for the per-row setup, then scores them one after
  // the other: the rows of a chunk are a serial chain, so few rows want short chunks spread over many waves
  const uint64_t nchunks = (nrows + chunk_rows - 1) / chunk_rows;
  const uint64_t wave_id = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint64_t nwaves = (uint64_t)gridDim.x * 4;
  for (uint64_t chunk = wave_id; chunk < nchunks; chunk += nwaves) {
    const uint64_t rb = chunk * chunk_rows;
    const int nr = (int)((nrows - rb) < (uint64_t)chunk_rows ? (nrows - rb) : (uint64_t)chunk_rows);
    const bool has_row = lane < nr;
    const float xv = has_row ? xcol[rb + lane] : 0.0f;
    int gz = has_row ? z[rb + lane] : -1;
    if ((uint32_t)gz >= K) gz = -1;                           // (an id outside the table: not assigned, as msc_accumulate reads it)
    const float u01 = philox_uniform01(seed, sweep, row_id0 + rb + lane);
    const unsigned long long mbits =
        __builtin_amdgcn_ballot_w64(fd.mask != nullptr && has_row && fd.mask[row0 + rb + lane] != 0);
    float sloo = 0.f;
    bool single = false;
    if (gz >= 0) {
      // leave-one-out score + prior of the row's own group, in double (what k_loo_own does for the other
      // kernels; here every lane has a row of its own, so it costs ~2 % on top of the 64 x K evaluations)
      const float lm1 = crp[kpad + gz];
      single = __builtin_isinf(lm1);                          // the row is its group's only member
      double s = single ? (double)le1 + (double)crp[2 * (size_t)kpad + 3] : (double)lm1 + (double)crp[crp_lo_cntm1(kpad) + gz];
      if (!((mbits >> lane) & 1ull))
        s += nich_loo_tab(fd.hp, fd.loo64 + (size_t)gz * kNlooStride, 1, xv);
      sloo = (float)s * kLog2e - bound;
    }
    // rows that leave the straight path: masked ones, and singletons while other groups are empty
    const unsigned long long singles = any_empty ? __builtin_amdgcn_ballot_w64(single) : 0ull;
    const unsigned long long odd = mbits | singles;
    int znew = gz;
    for (int r = 0; r < nr; r++) {
      const float x = lane_bcast(xv, r), sl = lane_bcast(sloo, r);
      const int g = lane_bcast(gz, r);
      float s[G];
#pragma unroll
      for (int j = 0; j < G; j++) s[j] = nich_eval_log2(x, mh[j], ml[j], c0s[j], c1[j], c2[j]);
      if ((odd >> r) & 1ull) {                            // (wave-uniform, rare)
        if ((mbits >> r) & 1ull) {                        // masked value: only the prior speaks
          const float e_raw = ((singles >> r) & 1ull) ? le1 : le0;
#pragma unroll
          for (int j = 0; j < G; j++) {
            const float lcj = crp[kb + j < kpad ? kb + j : 0];
            s[j] = (kb + j >= K) ? -INFINITY : (__builtin_isinf(lcj) ? e_raw : lcj) * kLog2e - bound;
          }
        } else {                                          // one more empty group shares alpha for this row
#pragma unroll
          for (int j = 0; j < G; j++) s[j] += ((emask >> j) & 1u) ? dle : 0.f;
        }
      }
      if (g >= 0 && lane == g / G) {                       // the own group: its leave-one-out value
#pragma unroll
        for (int j = 0; j < G; j++)
          if (j == g % G) s[j] = sl;
      }
      const int pick = sample_from_scores<G, true, true, true>(s, lane_bcast(u01, r), lane, K);
      if (lane == r) znew = pick;
    }
    if (has_row) z[rb + lane] = znew;
  }
}

// ---------------------------------------------------------------------------
// k_sweep_nich1_t<G>: the same step with the draw TRANSPOSED.  In k_sweep_nich1 every row pays a cross-lane prefix scan
// (six dependent DPP steps with their wait states), a ballot / first-hit / broadcast chain and the own-group patch: ~50
// of its 94 vector instructions at K = 256.  Here a wave takes 32 rows and
//   1. streams them past its groups exactly as before, but keeps only each lane's SUM of the row's unnormalised
//      probabilities: one ds_write_b32 per row (row r's 64 lane sums lie side by side, rows padded to 65 words);
//   2. then lane r finishes row r on its own: it recomputes the entries of the lane that holds the row's own group with
//      the leave-one-out value in place (the streaming pass knows nothing of own groups), adds up the row's 64 lane sums
//      in lane order (eight block sums kept), throws the dart and finds the block, then the lane, that reaches it;
//   3. and recomputes that one lane's G entries to find the group.
// Steps 2 and 3 cost a few hundred instructions per lane -- per 32 rows.  Same CDF order (lane by lane, entry by entry),
// same uniforms, same side paths (masked rows, rows that are their group's only member); a row whose total leaves the
// comfortable float range is redone by the wave with the exact maximum, as before.
// ---------------------------------------------------------------------------
constexpr int kTRows = 32;            // rows of a wave's chunk (lanes 0..31 finish them)
constexpr int kTPad = 65;             // words per row of lane sums: lane r reading word j hits bank (r + j) % 64

template <int G>
struct NichConsts {
  float mh[G], ml[G], c0s[G], c1[G], c2[G];
  unsigned emask;                     // bit j: group G l + j exists and is empty
};
// one row against group k from memory (what the finishing lane needs of a lane that is not its own): unnormalised log2
// probability, bound subtracted.  masked -> the prior alone; single -> an empty group's prior is le1 instead of le0
struct NichRowCtx {
  const float *tab, *crp;
  uint32_t K, kpad;
  float le0, le1, dle, bound, x;
  bool masked, single;
};
MSC_DEV float nich_entry_log2(const NichRowCtx &c, uint32_t k) {
  constexpr float kLog2e = 1.44269504088896340736f;
  const size_t kc = k < c.kpad ? k : 0;
  const float lc = c.crp[kc];
  const bool empty = __builtin_isinf(lc);
  float s;
  if (c.masked) {
    s = (empty ? (c.single ? c.le1 : c.le0) : lc) * kLog2e - c.bound;
  } else {
    const float c0s = (c.tab[(size_t)NICH_C0 * c.kpad + kc] + (empty ? c.le0 : lc)) * kLog2e - c.bound;
    s = nich_eval_log2(c.x, c.tab[(size_t)NICH_MU_HI * c.kpad + kc], c.tab[(size_t)NICH_MU_LO * c.kpad + kc], c0s,
                       c.tab[(size_t)NICH_C1 * c.kpad + kc], c.tab[(size_t)NICH_C2 * c.kpad + kc]);
    s += (c.single && empty) ? c.dle : 0.f;
  }
  return k < c.K ? s : -INFINITY;
}
// the streaming lanes' version of the same for the rare rows (masked / only member of their group / redone)
template <int G>
MSC_DEV void nich_row_scores(float (&s)[G], const NichConsts<G> &c, float x, bool masked, bool single, float dle, int own,
                             float sl, const float *__restrict__ crp, uint32_t K, uint32_t kpad, uint32_t l, float le0,
                             float le1, float bound) {
  constexpr float kLog2e = 1.44269504088896340736f;
#pragma unroll
  for (int j = 0; j < G; j++) s[j] = nich_eval_log2(x, c.mh[j], c.ml[j], c.c0s[j], c.c1[j], c.c2[j]);
  if (masked) {
#pragma unroll
    for (int j = 0; j < G; j++) {
      const uint32_t k = (uint32_t)G * l + j;
      const float lc = crp[k < kpad ? k : 0];
      s[j] = (k >= K) ? -INFINITY : (__builtin_isinf(lc) ? (single ? le1 : le0) : lc) * kLog2e - bound;
    }
  } else if (single) {
#pragma unroll
    for (int j = 0; j < G; j++) s[j] += ((c.emask >> j) & 1u) ? dle : 0.f;
  }
#pragma unroll
  for (int j = 0; j < G; j++)
    if (j == own) s[j] = sl;
}

constexpr int kTCst = 5;              // words per group of the block's LDS copy of the constants (odd: random groups spread over the banks)
enum { TC_MH = 0, TC_ML, TC_C0, TC_C1, TC_C2 };

template <int G>
__global__ __launch_bounds__(256) void k_sweep_nich1_t(const FeatDesc *__restrict__ feats, uint32_t K, uint32_t kpad,
                                                        uint64_t row0, uint64_t nrows, uint64_t row_id0,
                                                        int32_t *z, const float *__restrict__ crp,
                                                        const uint64_t *__restrict__ rng, ZeroSpans zero) {
  constexpr int TR = kTRows;                  // (24 rows and five waves per SIMD at K <= 256: 133 us against 125)
  __shared__ float lsum[4][TR * kTPad];
  __shared__ float cst[64 * G * kTCst];       // every group's constants, for the finishing lanes (log2 units, prior folded)
  const uint64_t seed = rng[0], sweep = rng[1];
  zero_spans(zero);
  const FeatDesc fd = feats[0];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (uniform, and known to be)
  float *mysum = lsum[wave];
  constexpr float kLog2e = 1.44269504088896340736f;
  const float le0 = crp[2 * (size_t)kpad], le1 = crp[2 * (size_t)kpad + 1];
  const bool any_empty = !__builtin_isinf(le0);
  const float dle = any_empty ? (le1 - le0) * kLog2e : 0.f;
  for (uint32_t k = threadIdx.x; k < 64u * G; k += 256u) {
    const uint32_t kc = k < kpad ? k : 0u;
    const float lc = crp[kc];
    float *c = cst + k * kTCst;
    c[TC_MH] = fd.tab[NICH_MU_HI * kpad + kc];
    c[TC_ML] = fd.tab[NICH_MU_LO * kpad + kc];
    c[TC_C0] = k < K ? (fd.tab[NICH_C0 * kpad + kc] + (__builtin_isinf(lc) ? le0 : lc)) * kLog2e : -INFINITY;
    c[TC_C1] = fd.tab[NICH_C1 * kpad + kc];
    c[TC_C2] = fd.tab[NICH_C2 * kpad + kc];
  }
  __syncthreads();
  // the wave-uniform bound that stands in for the row maximum: no score exceeds its c0' (log1p >= 0); 16 bits of
  // headroom for what may (own-group values, masked rows, a singleton's empties), and the total is checked
  NichConsts<G> mine;
  mine.emask = 0u;
  float bound = -INFINITY;
#pragma unroll
  for (int j = 0; j < G; j++) {
    const uint32_t k = (uint32_t)(G * lane + j);
    const float *c = cst + k * kTCst;
    mine.mh[j] = c[TC_MH], mine.ml[j] = c[TC_ML], mine.c0s[j] = c[TC_C0], mine.c1[j] = c[TC_C1], mine.c2[j] = c[TC_C2];
    if (k < K && __builtin_isinf(crp[k])) mine.emask |= 1u << j;
    bound = fmaxf(bound, mine.c0s[j]);
  }
  bound = wave_max(bound) + 16.f;
#pragma unroll
  for (int j = 0; j < G; j++) mine.c0s[j] -= bound;
  const float *xcol = reinterpret_cast<const float *>(fd.col) + row0;
  // every wave takes one contiguous range of rows, the same number (+-1) for all: no wave runs a chunk longer than another
  const uint64_t wave_id = (uint64_t)blockIdx.x * 4 + wave, nwaves = (uint64_t)gridDim.x * 4;
  const uint64_t per = nrows / nwaves, extra = nrows % nwaves;
  const uint64_t rbeg = wave_id * per + (wave_id < extra ? wave_id : extra), rend = rbeg + per + (wave_id < extra ? 1 : 0);
  for (uint64_t rb = rbeg; rb < rend; rb += 64) {
    const int nr = __builtin_amdgcn_readfirstlane((int)((rend - rb) < 64u ? (rend - rb) : 64u));
    const bool has_row = lane < nr;
    // ---- per-row setup, lane r <-> row r: 64 rows at a time (the uniform and the leave-one-out value are per-lane work) ----
    const float xv = has_row ? xcol[rb + lane] : 0.0f;
    int gz = has_row ? z[rb + lane] : -1;
    if ((uint32_t)gz >= K) gz = -1;                           // (an id outside the table: not assigned, as msc_accumulate reads it)
    const float u01 = philox_uniform01(seed, sweep, row_id0 + rb + lane);
    const bool my_mask = fd.mask != nullptr && has_row && fd.mask[row0 + rb + lane] != 0;
    const unsigned long long mbits = __builtin_amdgcn_ballot_w64(my_mask);
    float sloo = 0.f;
    bool single = false;
    if (gz >= 0) {
      // leave-one-out score + prior of the row's own group
      const float lm1 = crp[kpad + gz];
      single = __builtin_isinf(lm1);                          // the row is its group's only member
      // (float: the prior's hi part and the float tail of nich_loo_tab_sweep -- what the other entries of the row get)
      float s = single ? le1 : lm1;
      if (!my_mask) s += nich_loo_tab_sweep(fd.hp, fd.loo64 + (size_t)gz * kNlooStride, 1, xv);
      sloo = s * kLog2e - bound;
    }
    const bool my_single = single && any_empty;               // (with no other empty group nothing moves)
    const unsigned long long singles = __builtin_amdgcn_ballot_w64(my_single);
    const unsigned long long odd = mbits | singles;
    // group k for this row: from the block's constants; the row's own group is its leave-one-out value
    const bool plain = !(my_mask || my_single);               // (the others read the prior from memory)
    const NichRowCtx rc = {fd.tab, crp, K, kpad, le0, le1, dle, bound, xv, my_mask, my_single};
    auto entry = [&](int k) -> float {
      const float *c = cst + k * kTCst;
      float s = nich_eval_log2_est(xv, c[TC_MH], c[TC_ML], c[TC_C0] - bound, c[TC_C1], c[TC_C2]);
      if (!plain) s = nich_entry_log2(rc, (uint32_t)k);
      return __builtin_amdgcn_exp2f(k == gz ? sloo : s);
    };
    // the lane that holds the row's own group, summed with the leave-one-out value in place (the streaming pass knows
    // nothing of own groups)
    const int own_lane = gz >= 0 ? gz / G : -1;
    float own_sum = 0.f;
    if (own_lane >= 0) {
      own_sum = entry(G * own_lane);
#pragma unroll
      for (int j = 1; j < G; j++) own_sum += entry(G * own_lane + j);
    }
    int pick = gz;
    bool redo = false;
    // the lane sums of 32 rows fit the wave's LDS: the chunk goes through steps 1-3 in two halves
    for (int h0 = 0; h0 < nr; h0 += TR) {
      const int h1 = h0 + TR < nr ? h0 + TR : nr;
      // ---- 1. stream the rows past the groups: lane sums only ----
      __builtin_amdgcn_s_waitcnt(0);                          // (nothing pending at the loop's head: no wait inside it)
      for (int r = h0; r < h1; r++) {
        const float x = lane_bcast(xv, r);
        float s[G];
        if ((odd >> r) & 1ull) {                             // (wave-uniform, rare)
          nich_row_scores<G>(s, mine, x, ((mbits >> r) & 1ull) != 0, ((singles >> r) & 1ull) != 0, dle, -1, 0.f, crp, K, kpad,
                             (uint32_t)lane, le0, le1, bound);
        } else {
#pragma unroll
          for (int j = 0; j < G; j++) s[j] = nich_eval_log2_est(x, mine.mh[j], mine.ml[j], mine.c0s[j], mine.c1[j], mine.c2[j]);
        }
        float sum = __builtin_amdgcn_exp2f(s[0]);            // exp2(-inf) = 0 beyond K
#pragma unroll
        for (int j = 1; j < G; j++) sum += __builtin_amdgcn_exp2f(s[j]);
        mysum[(r - h0) * kTPad + lane] = sum;
      }
      __builtin_amdgcn_wave_barrier();                        // (LDS traffic of one wave is in order; this stops the compiler)
      // ---- 2. lane r finishes row r ----
      if (lane >= h0 && lane < h1) {
        float *row = mysum + (lane - h0) * kTPad;
        if (own_lane >= 0) row[own_lane] = own_sum;
        float blk[8], total = 0.f;                            // eight blocks of eight lanes, summed in lane order
#pragma unroll
        for (int b = 0; b < 8; b++) {
          float t = row[8 * b];
#pragma unroll
          for (int i = 1; i < 8; i++) t += row[8 * b + i];
          blk[b] = t;
          total = b == 0 ? t : total + t;
        }
        if (total > 0x1p-60f && total < 0x1p100f) {           // (false for NaN too)
          const float dart = u01 * total;
          // the running sum is monotone: the first block, lane, entry that reaches the dart is found by stepping over
          // those that do not (util.hpp:145-156; rounding may let the dart fall off the end: the last one then)
          float c0 = 0.f;
          int b0 = 0;
#pragma unroll
          for (int b = 0; b < 7; b++) {
            const bool miss = b0 == b && c0 + blk[b] < dart;
            c0 = miss ? c0 + blk[b] : c0;
            b0 += miss ? 1 : 0;
          }
          int hl = 8 * b0;
#pragma unroll
          for (int i = 0; i < 7; i++) {
            const float v = row[8 * b0 + i];
            const bool miss = hl == 8 * b0 + i && c0 + v < dart;
            c0 = miss ? c0 + v : c0;
            hl += miss ? 1 : 0;
          }
          // ---- 3. the entries of lane hl ----
          int j0 = 0;
#pragma unroll
          for (int j = 0; j < G - 1; j++) {
            const float v = entry(G * hl + j);
            const bool miss = j0 == j && c0 + v < dart;
            c0 = miss ? c0 + v : c0;
            j0 += miss ? 1 : 0;
          }
          const int k = G * hl + j0;
          pick = k < (int)K ? k : (int)K - 1;
        } else {
          redo = true;
        }
      }
      __builtin_amdgcn_wave_barrier();                        // (the next half overwrites the lane sums)
    }
    // rows whose total left the float range (an outlier thousands of bits below the bound): the wave redoes them one
    // by one with the exact maximum
    unsigned long long todo = __builtin_amdgcn_ballot_w64(redo);
    while (todo != 0ull) {
      const int r = (int)__builtin_ctzll(todo);
      todo &= todo - 1ull;
      const float x = lane_bcast(xv, r);
      const int g = lane_bcast(gz, r);
      float s[G];
      nich_row_scores<G>(s, mine, x, ((mbits >> r) & 1ull) != 0, ((singles >> r) & 1ull) != 0, dle,
                         (g >= 0 && lane == g / G) ? g % G : -1, lane_bcast(sloo, r), crp, K, kpad, (uint32_t)lane, le0, le1, bound);
      const int p = sample_from_scores<G, true, false>(s, lane_bcast(u01, r), lane, K);
      if (lane == r) pick = p;
    }
    if (has_row) z[rb + lane] = pick;
  }
}

// ---------------------------------------------------------------------------
// k_sweep_nich1_rows: the single-nich step for tables beyond 1024 groups (what the registers of k_sweep_nich1_t hold).
// LANE <-> ROW here, two rows per lane, 128 rows per wave visit; the groups stream past as SCALAR operands: a helper
// kernel writes every group's constants as 32 bytes (log2 units, prior and -bound folded; the prior alone and the
// singleton shift beside them for masked rows and rows that are their group's only member), the wave reads them with
// s_load_dwordx4 pairs and every lane evaluates its two rows against the same group -- no cross-lane operation
// anywhere.  A lane keeps the running sum of its rows' probabilities per block of B groups (<= 32 blocks; LDS, a
// column per lane), then finishes alone: recomputes the block of the row's own group with the leave-one-out value
// in place, adds the block sums up, throws the dart, finds the block, recomputes that block's entries (per-lane
// gathers from the same table: an L2 hit) to find the group.  K + 2 B evaluations per row instead of K, 6-12 % more;
// the materialise-and-sample path this replaces ran 4.5 / 9.4 ms per million rows at K = 2048 / 4096.
// Rows whose total leaves the float range are redone by the wave with the exact maximum (three passes over the table).
// ---------------------------------------------------------------------------
// (-DMSC_DBG_ALWAYS_REDO=1 sends every row of k_sweep_nich1_rows through the exact-maximum pass, which outliers alone
// reach otherwise: how that pass was checked against the oracle tests)
#ifndef MSC_DBG_ALWAYS_REDO
#define MSC_DBG_ALWAYS_REDO 0
#endif
constexpr int kRowsNb = 16;                      // block sums per row
constexpr uint32_t kRowsMaxK = 32768;            // (linear in K up to 16384: 2.9 ms per 200 k rows there; 19 ms at 32768, where the
                                                 //  per-lane passes over blocks of 2048 groups weigh in -- the materialised path: 25 ms)
struct RowsGeom {
  uint32_t B, nb;
};
inline RowsGeom rows_geom(uint32_t kpad) {
  RowsGeom g;
  g.B = ((kpad + kRowsNb - 1) / kRowsNb + 63) / 64 * 64;
  g.nb = (kpad + g.B - 1) / g.B;
  return g;
}
enum { RT_MH = 0, RT_ML, RT_C0, RT_C1, RT_C2, RT_PRIOR, RT_DLE, RT_PAD, RT_WORDS };

__global__ __launch_bounds__(1024) void k_nich_rows_table(const FeatDesc *__restrict__ feats, uint32_t K, uint32_t kpad,
                                                           const float *__restrict__ crp, float *__restrict__ tab, uint32_t nent) {
  __shared__ float red[16];
  const FeatDesc fd = feats[0];
  constexpr float kLog2e = 1.44269504088896340736f;
  const float le0 = crp[2 * (size_t)kpad], le1 = crp[2 * (size_t)kpad + 1];
  const bool any_empty = !__builtin_isinf(le0);
  const float dle = any_empty ? (le1 - le0) * kLog2e : 0.f;
  float mx = -INFINITY;
  for (uint32_t k = threadIdx.x; k < K; k += 1024) {
    const float lc = crp[k];
    mx = fmaxf(mx, (fd.tab[(size_t)NICH_C0 * kpad + k] + (__builtin_isinf(lc) ? le0 : lc)) * kLog2e);
  }
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  float bound = red[0];
  for (int w = 1; w < 16; w++) bound = fmaxf(bound, red[w]);
  bound += 16.f;                                 // (headroom as in k_sweep_nich1_t; the total is checked)
  for (uint32_t k = threadIdx.x; k < nent; k += 1024) {
    float *e = tab + (size_t)k * RT_WORDS;
    const bool in = k < K;
    const uint32_t kc = in ? k : 0u;
    const float lc = crp[kc];
    const bool empty = __builtin_isinf(lc);
    const float pr = empty ? le0 : lc;
    e[RT_MH] = in ? fd.tab[(size_t)NICH_MU_HI * kpad + kc] : 0.f;
    e[RT_ML] = in ? fd.tab[(size_t)NICH_MU_LO * kpad + kc] : 0.f;
    e[RT_C0] = in ? (fd.tab[(size_t)NICH_C0 * kpad + kc] + pr) * kLog2e - bound : -INFINITY;
    e[RT_C1] = in ? fd.tab[(size_t)NICH_C1 * kpad + kc] : 0.f;
    e[RT_C2] = in ? fd.tab[(size_t)NICH_C2 * kpad + kc] : 0.f;
    e[RT_PRIOR] = in ? pr * kLog2e - bound : -INFINITY;
    e[RT_DLE] = in && empty ? dle : 0.f;
    e[RT_PAD] = 0.f;
  }
  if (threadIdx.x < 8 * RT_WORDS) tab[(size_t)nent * RT_WORDS + threadIdx.x] = 0.f;          // the spare entries
  if (threadIdx.x == 0) tab[((size_t)nent + 8) * RT_WORDS] = bound;
}

// two rows against one table entry, as packed arithmetic: the probabilities 2^s (nich_eval_log2_est's steps, bit for bit)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
MSC_DEV f32x2 rows_eval2(f32x2 x, f32x4 e0, float c2) {
#pragma clang fp contract(off)
  const f32x2 mh = {e0.x, e0.x}, ml = {e0.y, e0.y}, c0 = {e0.z, e0.z}, nc1 = {-e0.w, -e0.w}, sc = {c2, c2};
  const f32x2 one = {1.0f, 1.0f};
  const f32x2 a = __builtin_elementwise_fma(x, sc, -mh) - ml;
  const f32x2 u = __builtin_elementwise_fma(a, a, one);
  const u32x2 magic = {kLog2eOverU, kLog2eOverU};
  const f32x2 ru = __builtin_bit_cast(f32x2, magic - __builtin_bit_cast(u32x2, u));
  const f32x2 lg = {hw_log2(u.x), hw_log2(u.y)};
  const f32x2 s = __builtin_elementwise_fma(nc1, __builtin_elementwise_fma(__builtin_elementwise_fma(a, a, -(u - one)), ru, lg), c0);
  return f32x2{__builtin_amdgcn_exp2f(s.x), __builtin_amdgcn_exp2f(s.y)};
}
// one row against one table entry (unnormalised log2 probability)
MSC_DEV float rows_entry(float x, f32x4 e0, f32x4 e1, bool masked, bool single) {
  float s = nich_eval_log2_est(x, e0.x, e0.y, e0.z, e0.w, e1.x);
  s += single ? e1.z : 0.f;
  return masked ? e1.y + (single ? e1.z : 0.f) : s;
}

__global__ __launch_bounds__(128) void k_sweep_nich1_rows(const FeatDesc *__restrict__ feats, uint32_t K, uint32_t kpad,
                                                           uint64_t row0, uint64_t nrows, uint64_t row_id0,
                                                           int32_t *__restrict__ z, const float *__restrict__ crp,
                                                           const uint64_t *__restrict__ rng, ZeroSpans zero,
                                                           const float *__restrict__ tab, uint32_t B, uint32_t nb) {
  __shared__ float blk_all[2][kRowsNb * 2 * 64];
  typedef const __attribute__((address_space(4))) f32x4 *scalar_f4;
  const uint64_t seed = rng[0], sweep = rng[1];
  zero_spans(zero);
  const FeatDesc fd = feats[0];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float *blk = blk_all[wave];
  constexpr float kLog2e = 1.44269504088896340736f;
  const uint32_t nent = nb * B;
  const float bound = tab[((size_t)nent + 8) * RT_WORDS];
  const float le1 = crp[2 * (size_t)kpad + 1];
  const bool any_empty = !__builtin_isinf(crp[2 * (size_t)kpad]);
  const float *xcol = reinterpret_cast<const float *>(fd.col) + row0;
  const scalar_f4 stab = (scalar_f4)tab;
  const f32x4 *gtab = reinterpret_cast<const f32x4 *>(tab);
  const uint64_t wave_id = (uint64_t)blockIdx.x * 2 + wave, nwaves = (uint64_t)gridDim.x * 2;
  const uint64_t per = nrows / nwaves, extra = nrows % nwaves;
  const uint64_t rbeg = wave_id * per + (wave_id < extra ? wave_id : extra), rend = rbeg + per + (wave_id < extra ? 1 : 0);
  for (uint64_t rb = rbeg; rb < rend; rb += 128) {
    float x[2], u01[2], sloo[2];
    int gz[2], pick[2];
    bool has[2], masked[2], single[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const uint64_t n = rb + 64 * i + lane;
      has[i] = n < rend;
      x[i] = has[i] ? xcol[n] : 0.f;
      gz[i] = has[i] ? z[n] : -1;
      if ((uint32_t)gz[i] >= K) gz[i] = -1;                 // (an id outside the table: not assigned)
      u01[i] = philox_uniform01(seed, sweep, row_id0 + n);
      masked[i] = fd.mask != nullptr && has[i] && fd.mask[row0 + n] != 0;
      sloo[i] = 0.f;
      single[i] = false;
      if (gz[i] >= 0) {
        const float lm1 = crp[kpad + gz[i]];
        const bool only = __builtin_isinf(lm1);                // the row is its group's only member
        float s = only ? le1 : lm1;
        if (!masked[i]) s += nich_loo_tab_sweep(fd.hp, fd.loo64 + (size_t)gz[i] * kNlooStride, 1, x[i]);
        sloo[i] = s * kLog2e - bound;
        single[i] = only && any_empty;
      }
      pick[i] = gz[i];
    }
    const bool odd = __builtin_amdgcn_ballot_w64(masked[0] || masked[1] || single[0] || single[1]) != 0ull;
    // ---- 1. the groups stream past: block sums per row ----
    for (uint32_t b = 0; b < nb; b++) {
      float run0 = 0.f, run1 = 0.f;
      if (!odd) {
        // the lane's two rows as the halves of packed operations (one v_pk_* for both), four groups per trip with the
        // next four's constants already on their way (B is a multiple of 64; past the table's end: the first entries
        // again, never used)
        f32x2 run = {0.f, 0.f};
        const f32x2 xx = {x[0], x[1]};
        typedef const __attribute__((address_space(4))) float *scalar_f;
        const scalar_f sflt = (scalar_f)tab;
        // (eight groups a trip in two halves that take turns: nothing fetched ahead is copied from one scalar register
        // to another, and every address is the trip's base plus a constant -- the table carries eight spare entries)
        scalar_f4 t4 = stab + 2 * (size_t)(b * B);
        scalar_f tf = sflt + (size_t)RT_WORDS * (b * B);
        f32x4 ca[4], cb[4];
        float sa[4], sb[4];
#pragma unroll
        for (int q = 0; q < 4; q++) ca[q] = t4[2 * q], sa[q] = tf[RT_WORDS * q + RT_C2];
        for (uint32_t j = 0; j < B; j += 8, t4 += 16, tf += 8 * RT_WORDS) {
#pragma unroll
          for (int q = 0; q < 4; q++) cb[q] = t4[2 * (4 + q)], sb[q] = tf[RT_WORDS * (4 + q) + RT_C2];
          __builtin_amdgcn_sched_barrier(0);                 // (the loads stay up here: their wait belongs after the arithmetic)
#pragma unroll
          for (int q = 0; q < 4; q++) run += rows_eval2(xx, ca[q], sa[q]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < 4; q++) ca[q] = t4[2 * (8 + q)], sa[q] = tf[RT_WORDS * (8 + q) + RT_C2];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < 4; q++) run += rows_eval2(xx, cb[q], sb[q]);
          __builtin_amdgcn_sched_barrier(0);
        }
        run0 = run.x, run1 = run.y;
      } else {
        for (uint32_t j = 0; j < B; j++) {
          const uint32_t k = b * B + j;
          const f32x4 e0 = stab[2 * k], e1 = stab[2 * k + 1];
          run0 += __builtin_amdgcn_exp2f(rows_entry(x[0], e0, e1, masked[0], single[0]));
          run1 += __builtin_amdgcn_exp2f(rows_entry(x[1], e0, e1, masked[1], single[1]));
        }
      }
      blk[(b * 2 + 0) * 64 + lane] = run0;
      blk[(b * 2 + 1) * 64 + lane] = run1;
    }
    // ---- 2. / 3. every lane finishes its rows ----
    bool redo[2] = {false, false};
#pragma unroll
    for (int i = 0; i < 2; i++) {
      if (!has[i]) continue;
      // the B entries of block `base / B` for this row, in order, eight gathers in flight at a time (one at a time the
      // loop is two dependent L2 latencies per entry: 256 of them per visit cost more than the streaming pass);
      // the row's own group: its leave-one-out value
      auto block_pass = [&](uint32_t base, auto &&consume) {
        for (uint32_t j = 0; j < B; j += 4) {
          f32x4 ea[4], eb[4];
#pragma unroll
          for (int q = 0; q < 4; q++) {                       // (32-bit byte offsets from the uniform base: one address register)
            const uint32_t at = (base + j + q) * (uint32_t)(RT_WORDS * sizeof(float));
            ea[q] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(gtab) + at);
            eb[q] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(gtab) + at + 16u);
          }
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const float sc = rows_entry(x[i], ea[q], eb[q], masked[i], single[i]);
            consume(j + q, __builtin_amdgcn_exp2f((int)(base + j + q) == gz[i] ? sloo[i] : sc));
          }
        }
      };
      if (gz[i] >= 0) {
        const uint32_t bo = (uint32_t)gz[i] / B;
        float sum = 0.f;
        block_pass(bo * B, [&](uint32_t, float p) { sum += p; });
        blk[(bo * 2 + i) * 64 + lane] = sum;
      }
      float total = 0.f;
      for (uint32_t b = 0; b < nb; b++) total += blk[(b * 2 + i) * 64 + lane];
      if (MSC_DBG_ALWAYS_REDO || !(total > 0x1p-60f && total < 0x1p100f)) {
        redo[i] = true;
        continue;
      }
      const float dart = u01[i] * total;
      float c0 = 0.f;
      uint32_t b0 = 0;
      for (uint32_t b = 0; b + 1 < nb; b++) {                 // (monotone: step over the blocks that do not reach the dart)
        const float v = blk[(b * 2 + i) * 64 + lane];
        const bool miss = b0 == b && c0 + v < dart;
        c0 = miss ? c0 + v : c0;
        b0 += miss ? 1u : 0u;
      }
      uint32_t j0 = 0;
      block_pass(b0 * B, [&](uint32_t j, float v) {
        const bool miss = j0 == j && j + 1 < B && c0 + v < dart;
        c0 = miss ? c0 + v : c0;
        j0 += miss ? 1u : 0u;
      });
      const uint32_t k = b0 * B + j0;
      pick[i] = (int)(k < K ? k : K - 1);
    }
    // rows whose total left the float range: exact maximum, total, search -- three passes over the table for the wave
#pragma unroll
    for (int i = 0; i < 2; i++) {
      if (__builtin_amdgcn_ballot_w64(redo[i]) == 0ull) continue;
      float m = -INFINITY;
      for (uint32_t k = 0; k < K; k++) {
        const f32x4 e0 = stab[2 * k], e1 = stab[2 * k + 1];
        const float s = (int)k == gz[i] ? sloo[i] : rows_entry(x[i], e0, e1, masked[i], single[i]);
        m = fmaxf(m, s);
      }
      double total = 0.0;                                      // (thousands of terms in one running sum: in double)
      for (uint32_t k = 0; k < K; k++) {
        const f32x4 e0 = stab[2 * k], e1 = stab[2 * k + 1];
        const float s = (int)k == gz[i] ? sloo[i] : rows_entry(x[i], e0, e1, masked[i], single[i]);
        total += (double)__builtin_amdgcn_exp2f(s - m);
      }
      const double dart = (double)u01[i] * total;
      double c0 = 0.0;
      uint32_t k0 = 0;
      for (uint32_t k = 0; k + 1 < K; k++) {
        const f32x4 e0 = stab[2 * k], e1 = stab[2 * k + 1];
        const float s = (int)k == gz[i] ? sloo[i] : rows_entry(x[i], e0, e1, masked[i], single[i]);
        const double v = (double)__builtin_amdgcn_exp2f(s - m);
        const bool miss = k0 == k && c0 + v < dart;
        c0 = miss ? c0 + v : c0;
        k0 += miss ? 1u : 0u;
      }
      if (redo[i]) pick[i] = (int)k0;
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
      if (has[i]) z[rb + 64 * i + lane] = pick[i];
  }
}

// ---------------------------------------------------------------------------
// One niw feature of small dimension, K <= 64 -- a Gaussian mixture on low-dimensional vectors, the textbook use of
// the family: the whole Gibbs step fused like k_sweep_nich1.  A lane keeps one group (lower triangle of W_k, W_k mu_k
// and the constants in registers), the rows of a chunk stream past as wave-uniform values, and q = |W_k (x - mu_k)|^2
// gives both the plain score and, on the lane that owns the row's group, the leave-one-out one in closed form
// (kernels_niw.hip header).  Nothing is materialised: the step was k_loo_own + a prior pass + the score kernel in
// accumulate mode + k_niw_loo_patch + k_sample_rows before (210 us of kernels for 262k rows x 128 groups at dim 3).
// ---------------------------------------------------------------------------
template <int D, int G>
__global__ __launch_bounds__(256) void k_sweep_niw1(const FeatDesc *__restrict__ feats, uint32_t K, uint32_t kpad,
                                                     uint64_t row0, uint64_t nrows, uint64_t row_id0,
                                                     int32_t *__restrict__ z, const float *__restrict__ crp,
                                                     const uint64_t *__restrict__ rng, int chunk_rows, ZeroSpans zero) {
  const uint64_t seed = rng[0], sweep = rng[1];
  zero_spans(zero);
  const FeatDesc fd = feats[0];
  const int lane = threadIdx.x & 63;
  const uint32_t kb = (uint32_t)(G * lane);              // lane l owns groups G l .. G l + G - 1 (k order, as the draw wants)
  double w[G][D * (D + 1) / 2], nb[G][D];
  float c0[G], c1[G], a_loo[G], b_loo[G], c_loo[G], lc[G], lm1[G];
#pragma unroll
  for (int t = 0; t < G; t++) {
    const size_t kc = kb + t < K ? kb + t : 0;
#pragma unroll
    for (int i = 0; i < D; i++) {
#pragma unroll
      for (int j = 0; j <= i; j++) w[t][i * (i + 1) / 2 + j] = fd.niw_w64[kc * niw_w_stream(D) + niw_w_index(i, j)];
      nb[t][i] = -fd.niw_mu64[kc * niw_b_stream(D) + niw_b_index(i)];
    }
    c0[t] = fd.tab[(size_t)NIW_C0 * kpad + kc];
    c1[t] = fd.tab[(size_t)NIW_C1 * kpad + kc];
    a_loo[t] = fd.tab[(size_t)NIW_A_LOO * kpad + kc];
    b_loo[t] = fd.tab[(size_t)NIW_B_LOO * kpad + kc];
    c_loo[t] = fd.tab[(size_t)NIW_C_LOO * kpad + kc];
    lc[t] = crp[kc];                                       // log count, log(count - 1); -inf when that is zero
    lm1[t] = crp[kpad + kc];
  }
  const float le0 = crp[2 * (size_t)kpad], le1 = crp[2 * (size_t)kpad + 1];
  const float *X = reinterpret_cast<const float *>(fd.col);
  const uint64_t nchunks = (nrows + chunk_rows - 1) / chunk_rows;
  const uint64_t wave_id = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (uint64_t)gridDim.x * 4;
  for (uint64_t chunk = wave_id; chunk < nchunks; chunk += nwaves) {
    const uint64_t rb = chunk * chunk_rows;
    const int nr = (int)((nrows - rb) < (uint64_t)chunk_rows ? (nrows - rb) : (uint64_t)chunk_rows);
    const bool has_row = lane < nr;
    double xd[D];
    bool msk = false;
#pragma unroll
    for (int j = 0; j < D; j++) xd[j] = has_row ? (double)X[(row0 + rb + lane) * D + j] : 0.0;
    if (has_row && fd.mask != nullptr)
#pragma unroll
      for (int j = 0; j < D; j++) msk |= fd.mask[(row0 + rb + lane) * D + j] != 0;
    int gz = has_row ? z[rb + lane] : -1;
    if ((uint32_t)gz >= K) gz = -1;                           // (an id outside the table: not assigned, as msc_accumulate reads it)
    const float u01 = philox_uniform01(seed, sweep, row_id0 + rb + lane);
    const unsigned long long mbits = __builtin_amdgcn_ballot_w64(msk);
    int znew = gz;
    for (int r = 0; r < nr; r++) {
      const int g = __builtin_amdgcn_readlane(gz, r);
      bool own[G], own_single = false;
#pragma unroll
      for (int t = 0; t < G; t++) {
        own[t] = g >= 0 && (uint32_t)g == kb + t;
        own_single |= own[t] && __builtin_isinf(lm1[t]);
      }
      // removing the row empties its group when that group has one member: one more empty group shares alpha
      const float e_row = __builtin_amdgcn_ballot_w64(own_single) != 0ull ? le1 : le0;
      const bool masked = (mbits >> r) & 1ull;
      double x[D];
#pragma unroll
      for (int j = 0; j < D; j++)
        x[j] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(xd[j]), r), __builtin_amdgcn_readlane(__double2loint(xd[j]), r));
      float s[G];
#pragma unroll
      for (int t = 0; t < G; t++) {
        const float prior = own[t] ? (__builtin_isinf(lm1[t]) ? e_row : lm1[t]) : (__builtin_isinf(lc[t]) ? e_row : lc[t]);
        float like = 0.f;                                  // masked vector: only the prior speaks
        if (!masked) {
          double q = 0.0;
#pragma unroll
          for (int i = 0; i < D; i++) {
            double y = nb[t][i];
#pragma unroll
            for (int j = 0; j <= i; j++) y = fma(w[t][i * (i + 1) / 2 + j], x[j], y);
            q = fma(y, y, q);
          }
          const float qf = (float)q;
          like = own[t] ? fmaf(b_loo[t], log1p_acc(-fminf(c_loo[t] * qf, 0.99999994f)), a_loo[t]) : fmaf(-c1[t], log1p_acc(qf), c0[t]);
        }
        s[t] = kb + t < K ? prior + like : -INFINITY;
      }
      const int pick = sample_from_scores<G>(s, __shfl(u01, r, 64), lane, K);
      if (lane == r) znew = pick;
    }
    if (has_row) z[rb + lane] = znew;
  }
}
template <int D, int G>
static void launch_sweep_niw1_t(hipStream_t stream, int num_cus, const FeatDesc *feats_dev, uint32_t K, uint32_t kpad, uint64_t row0,
                                uint64_t nrows, uint64_t row_id0, int32_t *z, const float *crp, const uint64_t *rng, ZeroSpans zero) {
  int chunk_rows = 64;
  while (chunk_rows > 4 && (nrows + chunk_rows - 1) / chunk_rows < (uint64_t)num_cus * 32) chunk_rows >>= 1;
  uint64_t gx = ((nrows + chunk_rows - 1) / chunk_rows + 3) / 4;
  const uint64_t cap = (uint64_t)num_cus * 4;
  if (gx > cap) gx = cap;
  hipLaunchKernelGGL((k_sweep_niw1<D, G>), dim3((unsigned)(gx ? gx : 1)), dim3(256), 0, stream, feats_dev, K, kpad, row0, nrows, row_id0,
                     z, crp, rng, chunk_rows, zero);
}
// groups per lane the registers allow at a dimension: D (D + 3) / 2 doubles per group, up to 56 in all
int sweep_niw1_max_groups(uint32_t dim) { return dim <= 4 ? 256 : dim == 5 ? 128 : dim <= 8 ? 64 : 0; }
// one niw feature, K <= sweep_niw1_max_groups(dim)
MSC_DEFINE_BIND_ERROR_WORD(bind_error_word_sweep)

int launch_sweep_niw1(hipStream_t stream, int num_cus, uint32_t dim, const FeatDesc *feats_dev, uint32_t K, uint32_t kpad,
                      uint64_t row0, uint64_t nrows, uint64_t row_id0, int32_t *z, const float *crp, const uint64_t *rng,
                      ZeroSpans zero) {
  if ((int)K > sweep_niw1_max_groups(dim)) return -2;
#define MSC_NIW1(Dv, Gv) launch_sweep_niw1_t<Dv, Gv>(stream, num_cus, feats_dev, K, kpad, row0, nrows, row_id0, z, crp, rng, zero)
#define MSC_NIW1_G(Dv)                \
  if (K <= 64) MSC_NIW1(Dv, 1);       \
  else if (K <= 128) MSC_NIW1(Dv, 2); \
  else MSC_NIW1(Dv, 4)
  switch (dim) {
    case 1: MSC_NIW1_G(1); break;
    case 2: MSC_NIW1_G(2); break;
    case 3: MSC_NIW1_G(3); break;
    case 4: MSC_NIW1_G(4); break;
    case 5:
      if (K <= 64) MSC_NIW1(5, 1);
      else MSC_NIW1(5, 2);
      break;
    case 6: MSC_NIW1(6, 1); break;
    case 7: MSC_NIW1(7, 1); break;
    case 8: MSC_NIW1(8, 1); break;
    default: return -2;
  }
#undef MSC_NIW1_G
#undef MSC_NIW1
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---------------------------------------------------------------------------
// any feature list, K <= 256: the workgroup tile of score_block.hpp, sampled from registers
// ---------------------------------------------------------------------------
template <int R, int W, bool DM>
__global__ __launch_bounds__(W * 64, DM ? 2 : W / 4) void k_sweep_tile(const FeatDesc *__restrict__ feats, int nfeat, int nsplit,
                                                                 uint32_t K, uint32_t kpad, uint64_t row0,
                                                                 uint64_t nrows, uint64_t row_id0,
                                                                 int32_t *__restrict__ z,
                                                                 const float *__restrict__ own,
                                                                 const float *__restrict__ crp,
                                                                 const uint64_t *__restrict__ rng, ZeroSpans zero) {
  const uint64_t seed = rng[0], sweep = rng[1];
  zero_spans(zero);
  __shared__ float4 lds[kGrpRows * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t kb = lane * 4;            // single k-tile: K <= 256
  const float4 logcnt = ld4(crp + kb);
  const float le0 = crp[2 * (size_t)kpad], le1 = crp[2 * (size_t)kpad + 1];
  const uint64_t rows_per_wg = (uint64_t)W * R;
  const uint64_t nchunks = (nrows + rows_per_wg - 1) / rows_per_wg;
  for (uint64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const uint64_t rb = chunk * rows_per_wg + (uint64_t)wave * R;
    const int nr = rb >= nrows ? 0 : (int)((nrows - rb) < (uint64_t)R ? (nrows - rb) : (uint64_t)R);
    int gz = -1;
    float sloo = 0.f, erow = le0;
    if (lane < nr) {
      gz = z[rb + lane];
      if ((uint32_t)gz >= K) gz = -1;                         // (an id outside the table: not assigned)
      if (gz >= 0) {
        sloo = own[rb + lane];
        erow = __builtin_isinf(crp[kpad + gz]) ? le1 : le0;
      }
    }
    float4 acc[R];
    const bool tail_only = nsplit == 0;                    // (score_tile, SPLIT: the prior is added afterwards then)
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = tail_only ? make_float4(0, 0, 0, 0) : crp_prior4(logcnt, lane_bcast(erow, r));
    score_tile<R, W, DM, true, true>(feats, nfeat, nsplit, kpad, 0, lane, row0 + rb, nr, row0, lds, acc);   // (EST: the sweeps' nich form)
    if (tail_only) {
#pragma unroll
      for (int r = 0; r < R; r++) add4(acc[r], crp_prior4(logcnt, lane_bcast(erow, r)));
    }
    const float u01 = philox_uniform01(seed, sweep, row_id0 + rb + lane);
    int znew = gz;
#pragma unroll
    for (int r = 0; r < R; r++) {
      float4 s4 = acc[r];
      const int g = lane_bcast(gz, r);
      if (g >= 0) replace_own(s4, kb, g, lane_bcast(sloo, r));
      float s[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (kb + j >= K) s[j] = -INFINITY;
      const int pick = sample_from_scores<4>(s, lane_bcast(u01, r), lane, K);
      if (lane == r) znew = pick;
    }
    if (lane < nr) z[rb + lane] = znew;
  }
}

// ---------------------------------------------------------------------------
// k_sweep_tile_roles: k_sweep_tile with the workgroup's waves split between the lookup phase (waves 0-7, sixteen rows
// each, the table slot and a barrier of their own) and the nich phase (waves 8-15, the same rows, constants from L2) --
// kernels_score.hip k_score_tile_roles, where the why is written down.  The lookup waves take the sums over and draw.
// ---------------------------------------------------------------------------
// TAILP = 1, 2: 256 < K <= 320, 384 -- the groups beyond the tile were scored by k_score_tail_rows (leave-one-out value and
// prior included) into `tail`, 64 TAILP floats per row; a lookup wave fetches its sixteen rows of them into its pair's hand-over region once the
// nich sums are read (the region is the wave's own until every lookup wave has passed the next chunk's first barrier)
// and draws over tile + tail.  An instantiation of its own: the K <= 256 kernel keeps its registers.
// Round 4: the hand-over REVERSED as in k_score_tile_roles -- the lookup waves park prior + lookups in the slot and go on
// to the next chunk, the nich waves (the shorter half since the nich features go in blocks) add their sums, put the
// leave-one-out value in place and DRAW; with a tail (TAILP > 0) they fetch a row's tail scores from `tail` themselves,
// four rows at a time ahead of the draws (before: the lookup waves drew, the tail's rows through their pair's region of
// the slot).
// PAIR (with TAILP == 0; K <= 128): a lane carries two groups, a float4 of sums two rows, a wave 32 rows (score_block.hpp
// pair_dup; k_score_tile_roles<.., PAIR>); a row's draw runs over the wave's 64 x 2 entries.
template <int TAILP, bool PAIR = false>
__global__ __launch_bounds__(1024, 4) void k_sweep_tile_roles(const FeatDesc *__restrict__ feats, int nfeat, int nsplit,
                                                               uint32_t K, uint32_t kpad, uint64_t row0, uint64_t nrows,
                                                               uint64_t row_id0, int32_t *__restrict__ z,
                                                               const float *__restrict__ own, const float *__restrict__ crp,
                                                               const uint64_t *__restrict__ rng, ZeroSpans zero,
                                                               const float *__restrict__ tail) {
  constexpr int R = 16;
  constexpr bool TAIL = TAILP > 0;
  constexpr int TL = 16 * (TAIL ? TAILP : 1);            // lanes that hold a row's tail scores, four each
  const uint64_t seed = rng[0], sweep = rng[1];
  zero_spans(zero);
  __shared__ float4 lds[kGrpRows * 64];
  __shared__ uint32_t lookers_arrived;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool looker = wave < 8;
  const int pair = wave & 7;
  if (threadIdx.x == 0) lookers_arrived = 0u;
  __syncthreads();
  WaveSubsetBarrier<8> lbar{&lookers_arrived, 0u};
  static_assert(!PAIR || TAILP == 0, "PAIR mode: one tile of at most 128 groups");
  constexpr int RW = PAIR ? 2 * R : R;     // rows per wave
  const uint32_t kb = PAIR ? lane * 2 : lane * 4;            // single k-tile: K <= 256
  const float4 logcnt = PAIR ? pair_dup(ld2(crp + kb)) : ld4(crp + kb);
  const float le0 = crp[2 * (size_t)kpad], le1 = crp[2 * (size_t)kpad + 1];
  const uint64_t rows_per_wg = 8 * RW;
  const uint64_t nchunks = (nrows + rows_per_wg - 1) / rows_per_wg;
  {
    float4 *const handover = lds + (size_t)pair * R * 64 + lane;
    if (looker) {
      for (uint64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const uint64_t rb = chunk * rows_per_wg + (uint64_t)pair * RW;
        const int nr = rb >= nrows ? 0 : (int)((nrows - rb) < (uint64_t)RW ? (nrows - rb) : (uint64_t)RW);
        float erow = le0;                                 // (a row that is its group's only member leaves one more empty group)
        if (lane < nr) {
          const int g0 = z[rb + lane];
          if ((uint32_t)g0 < K && __builtin_isinf(crp[kpad + g0])) erow = le1;
        }
        float4 acc[R];
#pragma unroll
        for (int r = 0; r < R; r++)
          acc[r] = PAIR ? crp_prior_pair(make_float2(logcnt.x, logcnt.y), lane_bcast(erow, 2 * r), lane_bcast(erow, 2 * r + 1))
                        : crp_prior4(logcnt, lane_bcast(erow, r));
        score_tile_groups<R, 8, false, false, PAIR>(feats, nsplit, kpad, 0, lane, row0 + rb, nr, row0, lds, acc, lbar);
        lbar();                                           // every lookup wave is done reading the slot's tables
#pragma unroll
        for (int r = 0; r < R; r++) handover[r * 64] = acc[r];
        __syncthreads();                                  // (1) prior + lookups are in the slot
        __syncthreads();                                  // (2) the nich waves have read them
      }
      return;
    }
    for (uint64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
      const uint64_t rb = chunk * rows_per_wg + (uint64_t)pair * RW;
      const int nr = rb >= nrows ? 0 : (int)((nrows - rb) < (uint64_t)RW ? (nrows - rb) : (uint64_t)RW);
      // (the rows' groups and leave-one-out values stream in from HBM: asked for before the phase, two registers across
      // it; the uniforms are made after it -- kernels_score.hip k_score_tile_roles says why)
      int gz = -1;
      float sloo = 0.f;
      if (lane < nr) {
        gz = z[rb + lane];
        sloo = own[rb + lane];
      }
      float4 acc[R];
      const uint64_t myrow = row0 + (lane < nr ? rb + lane : (nr ? rb : 0));   // (a row of the call's range for idle lanes)
      nich_phase_packed<R, true, PAIR>(feats, nsplit, kpad, kb, row0 + rb, nr, myrow, acc);      // (EST: the sweeps' form)
      int32_t *zq = z;
      asm volatile("" : "+s"(zq));
      if ((uint32_t)gz >= K) gz = -1, sloo = 0.f;         // (an id outside the table: not assigned)
      const float u01 = philox_uniform01(seed, sweep, row_id0 + rb + lane);
      __syncthreads();                                    // (1)
#pragma unroll
      for (int r = 0; r < R; r++) {                       // (prior + lookups) + (nich features)
        float4 t = handover[r * 64];
        add4(t, acc[r]);
        acc[r] = t;
        if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // (four reads in flight: sixteen beside the sixteen sums spill)
      }
      __syncthreads();                                    // (2)
      int znew = gz;
      [[maybe_unused]] float4 t4[4];
#pragma unroll
      for (int r = 0; r < R; r++) {
        float4 s4 = acc[r];
        if constexpr (PAIR) {
          replace_own_pair(s4, lane, lane_bcast(gz, 2 * r), lane_bcast(sloo, 2 * r), lane_bcast(gz, 2 * r + 1), lane_bcast(sloo, 2 * r + 1));
          float sa[2] = {s4.x, s4.y}, sb[2] = {s4.z, s4.w};
#pragma unroll
          for (int j = 0; j < 2; j++)
            if (kb + j >= K) sa[j] = sb[j] = -INFINITY;
          const int pa = sample_from_scores<2>(sa, lane_bcast(u01, 2 * r), lane, K);
          const int pb = sample_from_scores<2>(sb, lane_bcast(u01, 2 * r + 1), lane, K);
          if (lane == 2 * r) znew = pa;
          if (lane == 2 * r + 1) znew = pb;
        } else {
          const int g = lane_bcast(gz, r);
          if (g >= 0) replace_own(s4, kb, g, lane_bcast(sloo, r));
          float sc[4] = {s4.x, s4.y, s4.z, s4.w};
          int pick;
          if constexpr (TAIL) {
            // the groups beyond the tile: this row's 64 TAILP scores (k_score_tail_rows wrote them, leave-one-out value and
            // prior included), four a lane on the first TL lanes, fetched four rows at a time ahead of their draws
            if ((r & 3) == 0) {
#pragma unroll
              for (int j = 0; j < 4; j++) {
                uint64_t tr = rb + (uint64_t)(r + j);
                tr = tr < nrows ? tr : nrows - 1;
                t4[j] = gld4(as_global(tail) + tr * (uint64_t)(4 * TL) + 4 * (lane % TL));
              }
            }
            const float4 tq = t4[r & 3];
            float st[4] = {tq.x, tq.y, tq.z, tq.w};
#pragma unroll
            for (int j = 0; j < 4; j++)
              if (lane >= TL || (uint32_t)kGroupTile + 4 * (uint32_t)lane + j >= K) st[j] = -INFINITY;
            pick = sample_tile_and_tail<TAIL ? TAILP : 1>(sc, st, lane_bcast(u01, r), lane, K);
          } else {
#pragma unroll
            for (int j = 0; j < 4; j++)
              if (kb + j >= K) sc[j] = -INFINITY;
            pick = sample_from_scores<4>(sc, lane_bcast(u01, r), lane, K);
          }
          if (lane == r) znew = pick;
        }
      }
      if (lane < nr) zq[rb + lane] = znew;
    }
  }
}

// ---------------------------------------------------------------------------
// k_sweep_lookups: the fused assignment step of a state of staged lookup features only (kernels_score.hip k_score_lookups):
// sixteen lookup waves of 16 sums -- 16 rows, or 32 in PAIR mode -- a workgroup; prior + lookups, the leave-one-out entry,
// the draw.  K <= 256; PAIR: K <= 128.
// ---------------------------------------------------------------------------
// TAILP = 1, 2 (not PAIR): 256 < K <= 320, 384 -- the groups beyond the tile scored by k_score_tail_rows into `tail`, 64 TAILP
// floats a row, and drawn over together with the tile (k_sweep_tile_roles has the same)
template <bool PAIR, int TAILP = 0>
__global__ __launch_bounds__(1024, 4) void k_sweep_lookups(const FeatDesc *__restrict__ feats, int nfeat, uint32_t K, uint32_t kpad, uint64_t row0,
                                                            uint64_t nrows, uint64_t row_id0, int32_t *__restrict__ z,
                                                            const float *__restrict__ own, const float *__restrict__ crp,
                                                            const uint64_t *__restrict__ rng, ZeroSpans zero, const float *__restrict__ tail = nullptr) {
  static_assert(!PAIR || TAILP == 0, "PAIR mode: one tile of at most 128 groups");
  constexpr bool TAIL = TAILP > 0;
  constexpr int TL = 16 * (TAIL ? TAILP : 1);
  constexpr int R = 16, RW = PAIR ? 32 : 16;
  const uint64_t seed = rng[0], sweep = rng[1];
  zero_spans(zero);
  __shared__ float4 lds[kGrpRows * 64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t kb = PAIR ? lane * 2 : lane * 4;
  const uint64_t rows_per_wg = 16 * RW;
  const uint64_t nchunks = (nrows + rows_per_wg - 1) / rows_per_wg;
  for (uint64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const uint64_t rb = chunk * rows_per_wg + (uint64_t)wave * RW;
    const int nr = rb >= nrows ? 0 : (int)((nrows - rb) < (uint64_t)RW ? (nrows - rb) : (uint64_t)RW);
    int gz = -1;
    float sloo = 0.f;
    float4 acc[R];
    {
      const float4 logcnt = PAIR ? pair_dup(ld2(crp + kb)) : ld4(crp + kb);
      const float le0 = crp[2 * (size_t)kpad], le1 = crp[2 * (size_t)kpad + 1];
      float erow = le0;                                   // (a row that is its group's only member leaves one more empty group)
      if (lane < nr) {
        gz = z[rb + lane];
        if ((uint32_t)gz >= K) gz = -1;                   // (an id outside the table: not assigned)
        if (gz >= 0) {
          sloo = own[rb + lane];
          if (__builtin_isinf(crp[kpad + gz])) erow = le1;
        }
      }
#pragma unroll
      for (int r = 0; r < R; r++)
        acc[r] = PAIR ? crp_prior_pair(make_float2(logcnt.x, logcnt.y), lane_bcast(erow, 2 * r), lane_bcast(erow, 2 * r + 1))
                      : crp_prior4(logcnt, lane_bcast(erow, r));
    }
    score_tile_groups<R, 16, false, false, PAIR>(feats, nfeat, kpad, 0, lane, row0 + rb, nr, row0, lds, acc);
    const float u01 = philox_uniform01(seed, sweep, row_id0 + rb + lane);
    int znew = gz;
    [[maybe_unused]] float4 t4[4];
#pragma unroll
    for (int r = 0; r < R; r++) {
      float4 s4 = acc[r];
      if constexpr (PAIR) {
        replace_own_pair(s4, lane, lane_bcast(gz, 2 * r), lane_bcast(sloo, 2 * r), lane_bcast(gz, 2 * r + 1), lane_bcast(sloo, 2 * r + 1));
        float sa[2] = {s4.x, s4.y}, sb[2] = {s4.z, s4.w};
#pragma unroll
        for (int j = 0; j < 2; j++)
          if (kb + j >= K) sa[j] = sb[j] = -INFINITY;
        const int pa = sample_from_scores<2>(sa, lane_bcast(u01, 2 * r), lane, K);
        const int pb = sample_from_scores<2>(sb, lane_bcast(u01, 2 * r + 1), lane, K);
        if (lane == 2 * r) znew = pa;
        if (lane == 2 * r + 1) znew = pb;
      } else {
        const int g = lane_bcast(gz, r);
        if (g >= 0) replace_own(s4, kb, g, lane_bcast(sloo, r));
        float sc[4] = {s4.x, s4.y, s4.z, s4.w};
        int pick;
        if constexpr (TAIL) {
          if ((r & 3) == 0) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
              uint64_t tr = rb + (uint64_t)(r + j);
              tr = tr < nrows ? tr : nrows - 1;
              t4[j] = gld4(as_global(tail) + tr * (uint64_t)(4 * TL) + 4 * (lane % TL));
            }
          }
          const float4 tq = t4[r & 3];
          float st[4] = {tq.x, tq.y, tq.z, tq.w};
#pragma unroll
          for (int j = 0; j < 4; j++)
            if (lane >= TL || (uint32_t)kGroupTile + 4 * (uint32_t)lane + j >= K) st[j] = -INFINITY;
          pick = sample_tile_and_tail<TAIL ? TAILP : 1>(sc, st, lane_bcast(u01, r), lane, K);
        } else {
#pragma unroll
          for (int j = 0; j < 4; j++)
            if (kb + j >= K) sc[j] = -INFINITY;
          pick = sample_from_scores<4>(sc, lane_bcast(u01, r), lane, K);
        }
        if (lane == r) znew = pick;
      }
    }
    if (lane < nr) z[rb + lane] = znew;
  }
}

// ---------------------------------------------------------------------------
// k_sweep_nich_pack: the fused assignment step of a state of plain nich features only (kernels_score.hip
// k_score_nich_pack, where the why is written down): sixteen nich waves a workgroup, no LDS, no barrier; nich sums + prior,
// the leave-one-out entry, the draw.  K <= 256; PAIR: K <= 128, two groups a lane, 32 rows a wave.
// ---------------------------------------------------------------------------
template <bool PAIR, bool LOOK, int TAILP = 0>
__global__ __launch_bounds__(kNichPackWaves * 64, kNichPackWaves / 4) void k_sweep_nich_pack(const FeatDesc *__restrict__ feats, int nsplit, uint32_t K, uint32_t kpad, uint64_t row0,
                                                              uint64_t nrows, uint64_t row_id0, int32_t *__restrict__ z,
                                                              const float *__restrict__ own, const float *__restrict__ crp,
                                                              const uint64_t *__restrict__ rng, ZeroSpans zero, const float *__restrict__ tail = nullptr) {
  static_assert(!PAIR || TAILP == 0, "PAIR mode: one tile of at most 128 groups");
  constexpr bool TAIL = TAILP > 0;                       // (256 < K <= 384: the groups beyond the tile from `tail`, as k_sweep_tile_roles)
  constexpr int TL = 16 * (TAIL ? TAILP : 1);
  constexpr int R = 16, RW = PAIR ? 32 : 16;
  const uint64_t seed = rng[0], sweep = rng[1];
  zero_spans(zero);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t kb = PAIR ? lane * 2 : lane * 4;
  const uint64_t rows_per_wg = kNichPackWaves * RW;
  const uint64_t nchunks = (nrows + rows_per_wg - 1) / rows_per_wg;
  for (uint64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const uint64_t rb = chunk * rows_per_wg + (uint64_t)wave * RW;
    const int nr = rb >= nrows ? 0 : (int)((nrows - rb) < (uint64_t)RW ? (nrows - rb) : (uint64_t)RW);
    if (nr == 0) continue;                                // (wave-uniform; nothing in this kernel waits for another wave)
    // (across the nich phase only the row's group and its leave-one-out value stay in registers -- the phase has the 4 R
    // sums, a block's constants and nothing to spare: with the prior's terms and the uniform held too the PAIR
    // instantiation spilled 403 registers and ran 1.04 ms where the scoring kernel takes 0.45)
    int gz = -1;
    float sloo = 0.f;
    if (lane < nr) {
      gz = z[rb + lane];
      if ((uint32_t)gz >= K) gz = -1;                     // (an id outside the table: not assigned)
      if (gz >= 0) sloo = own[rb + lane];
    }
    float4 acc[R];
    const uint64_t myrow = row0 + (lane < nr ? rb + lane : (nr ? rb : 0));   // (a row of the call's range for idle lanes and idle waves)
    [[maybe_unused]] float4 accl[LOOK ? R : 1];
    if constexpr (LOOK) {                                   // a first phase of a few lookup features: prior + lookups, from L2
      const float4 lc = PAIR ? pair_dup(ld2(crp + kb)) : ld4(crp + kb);
      const float l0 = crp[2 * (size_t)kpad], l1 = crp[2 * (size_t)kpad + 1];
      const float er = (gz >= 0 && __builtin_isinf(crp[kpad + (gz >= 0 ? gz : 0)])) ? l1 : l0;
#pragma unroll
      for (int r = 0; r < R; r++)
        accl[r] = PAIR ? crp_prior_pair(make_float2(lc.x, lc.y), lane_bcast(er, 2 * r), lane_bcast(er, 2 * r + 1)) : crp_prior4(lc, lane_bcast(er, r));
      pack_l2_lookups<R, PAIR>(feats, nsplit, kpad, kb, myrow, accl);
    }
    nich_phase_packed<R, true, PAIR, LOOK ? 2 : kNichPackNC>(feats, nsplit, kpad, kb, row0 + rb, nr, myrow, acc);      // (EST: the sweeps' form)
    __builtin_amdgcn_sched_barrier(0);                    // (the draw's temporaries stay behind the phase)
    const float *prior = crp;
    asm volatile("" : "+s"(prior));                       // (fetched per chunk, an L2 hit: not kept across the phase)
    const float4 logcnt = PAIR ? pair_dup(ld2(prior + kb)) : ld4(prior + kb);
    const float le0 = prior[2 * (size_t)kpad], le1 = prior[2 * (size_t)kpad + 1];
    const float erow = (gz >= 0 && __builtin_isinf(prior[kpad + (gz >= 0 ? gz : 0)])) ? le1 : le0;   // (its group's only member: one more empty group)
    const float u01 = philox_uniform01(seed, sweep, row_id0 + rb + lane);
    int znew = gz;
    [[maybe_unused]] float4 t4[4];
#pragma unroll
    for (int r = 0; r < R; r++) {
      float4 s4 = acc[r];
      if constexpr (LOOK) {                                 // (prior + lookups) + (nich features)
        s4 = accl[r];
        add4(s4, acc[r]);
      }
      if constexpr (PAIR) {
        if (!LOOK) add4(s4, crp_prior_pair(make_float2(logcnt.x, logcnt.y), lane_bcast(erow, 2 * r), lane_bcast(erow, 2 * r + 1)));
        replace_own_pair(s4, lane, lane_bcast(gz, 2 * r), lane_bcast(sloo, 2 * r), lane_bcast(gz, 2 * r + 1), lane_bcast(sloo, 2 * r + 1));
        float sa[2] = {s4.x, s4.y}, sb[2] = {s4.z, s4.w};
#pragma unroll
        for (int j = 0; j < 2; j++)
          if (kb + j >= K) sa[j] = sb[j] = -INFINITY;
        const int pa = sample_from_scores<2>(sa, lane_bcast(u01, 2 * r), lane, K);
        const int pb = sample_from_scores<2>(sb, lane_bcast(u01, 2 * r + 1), lane, K);
        if (lane == 2 * r) znew = pa;
        if (lane == 2 * r + 1) znew = pb;
      } else {
        if (!LOOK) add4(s4, crp_prior4(logcnt, lane_bcast(erow, r)));
        const int g = lane_bcast(gz, r);
        if (g >= 0) replace_own(s4, kb, g, lane_bcast(sloo, r));
        float sc[4] = {s4.x, s4.y, s4.z, s4.w};
        int pick;
        if constexpr (TAIL) {
          if ((r & 3) == 0) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
              uint64_t tr = rb + (uint64_t)(r + j);
              tr = tr < nrows ? tr : nrows - 1;
              t4[j] = gld4(as_global(tail) + tr * (uint64_t)(4 * TL) + 4 * (lane % TL));
            }
          }
          const float4 tq = t4[r & 3];
          float st[4] = {tq.x, tq.y, tq.z, tq.w};
#pragma unroll
          for (int j = 0; j < 4; j++)
            if (lane >= TL || (uint32_t)kGroupTile + 4 * (uint32_t)lane + j >= K) st[j] = -INFINITY;
          pick = sample_tile_and_tail<TAIL ? TAILP : 1>(sc, st, lane_bcast(u01, r), lane, K);
        } else {
#pragma unroll
          for (int j = 0; j < 4; j++)
            if (kb + j >= K) sc[j] = -INFINITY;
          pick = sample_from_scores<4>(sc, lane_bcast(u01, r), lane, K);
        }
        if (lane == r) znew = pick;
      }
    }
    if (lane < nr) z[rb + lane] = znew;
  }
}

// ---------------------------------------------------------------------------
// K <= 64: the narrow tiling.  In the 256-group tiling above a lane owns 4 groups of a 256-group tile, so with 16
// groups 4 lanes of 64 work (a million rows x 16 groups x 8 bb columns scored at 1/17 of what the bytes allow).
// Here L = ceil(K / 4) rounded to 4 / 8 / 16 lanes share a row -- lane q of them owns groups 4q .. 4q+3 -- and a
// wave takes 64 / L rows per step.  Every lane fetches its own row's value, so nothing is broadcast; the tables,
// compacted to 4 L groups per row, all sit in LDS (the host checks that they fit: narrow_lanes in abi.cpp); the
// own group's leave-one-out entry is a per-lane compare; the draw reduces over the L lanes of a row with
// width-L shuffles.  Features are added in the caller's order.  SWEEP = false writes scores, true draws.
// ---------------------------------------------------------------------------
// max / sum over the L (4, 8, 16) neighbouring lanes of a row, every lane getting the result, on the DPP path:
// quad_perm [1,0,3,2] and [2,3,0,1] exchange within a quad, row_half_mirror / row_mirror reverse 8 / 16 lanes (for a
// commutative reduction a reversal pairs each lane with the other half as well as an exchange would)
template <int L, bool MAX>
MSC_DEV float narrow_allreduce(float v) {
  float t = dpp_f32<0xB1, 0xf>(v, v);                    // quad_perm:[1,0,3,2]
  v = MAX ? fmaxf(v, t) : v + t;
  t = dpp_f32<0x4E, 0xf>(v, v);                          // quad_perm:[2,3,0,1]
  v = MAX ? fmaxf(v, t) : v + t;
  if (L > 4) {
    t = dpp_f32<0x141, 0xf>(v, v);                       // row_half_mirror
    v = MAX ? fmaxf(v, t) : v + t;
  }
  if (L > 8) {
    t = dpp_f32<0x140, 0xf>(v, v);                       // row_mirror
    v = MAX ? fmaxf(v, t) : v + t;
  }
  return v;
}
MSC_DEV uint32_t narrow_table_rows(const FeatDesc &fd) {
  switch (fd.family) {
    case MSC_BB: case MSC_BBNC: return 2;
    case MSC_NICH: return NICH_ROWS;
    case MSC_DD: return fd.dim;
    case MSC_GP: case MSC_BNB: return fd.vcap;
    default: return 0;
  }
}
template <int L, int S, bool SWEEP>
__global__ __launch_bounds__(256) void k_narrow(const FeatDesc *__restrict__ feats, int nfeat, uint32_t K, uint32_t kpad,
                                                 uint64_t row0, uint64_t nrows, const int32_t *__restrict__ z,
                                                 const float *__restrict__ own, const float *__restrict__ crp,
                                                 float *__restrict__ out, uint64_t ld, uint64_t row_id0,
                                                 int32_t *__restrict__ z_out, const uint64_t *__restrict__ rng, ZeroSpans zero) {
  extern __shared__ float4 nlds[];
  if (SWEEP) zero_spans(zero);
  // stage every feature's table, 4 L groups wide
  {
    uint32_t off = 0;
    for (int f = 0; f < nfeat; f++) {
      const FeatDesc &fd = feats[f];
      const uint32_t rows = narrow_table_rows(fd), first = is_count_family(fd.family) ? (uint32_t)GP_T0 : 0u;
      for (uint32_t i = threadIdx.x; i < rows * L; i += 256) {
        const uint32_t r = i / L, q = i % L;
        nlds[off + i] = ld4(fd.tab + (size_t)(first + r) * kpad + 4 * q);
      }
      off += rows * L;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, q = lane % L, sub = lane / L;
  constexpr int kRowsPerStep = 64 / L;
  const uint32_t kb = 4 * q;
  const bool loo = z != nullptr, pri = crp != nullptr;
  // the prior as (hi, lo) pairs (kernels_score.hip crp_prepare_block): an accumulator starts from lo and takes hi
  // after the last feature, so that no step of the sum rounds relative to log(count)
  float4 logcnt = make_float4(0, 0, 0, 0), logcnt_lo = make_float4(0, 0, 0, 0);
  float le0 = 0.f, le1 = 0.f, le0_lo = 0.f, le1_lo = 0.f;
  if (pri) {
    logcnt = ld4(crp + kb);
    logcnt_lo = ld4(crp + crp_lo_cnt(kpad) + kb);
    le0 = crp[2 * (size_t)kpad];
    le1 = crp[2 * (size_t)kpad + 1];
    le0_lo = crp[2 * (size_t)kpad + 2];
    le1_lo = crp[2 * (size_t)kpad + 3];
  }
  const uint64_t seed = SWEEP ? rng[0] : 0, sweep = SWEEP ? rng[1] : 0;
  const bool vec_ok = !SWEEP && ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  // A wave carries S steps at a time (S x 64 / L rows, one accumulator each) and walks the features once for all
  // of them: the descriptor of a feature is fetched once per S steps and the S value loads are in flight together
  // (one step at a time, a row's walk over its features was one load latency after the other).
  const uint64_t nsteps = (nrows + kRowsPerStep - 1) / kRowsPerStep;
  const uint64_t nbatches = (nsteps + S - 1) / S;
  const uint64_t wave_id = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (uint64_t)gridDim.x * 4;
  for (uint64_t batch = wave_id; batch < nbatches; batch += nwaves) {
    uint64_t nn[S];
    bool has[S];
    int gzs[S];
    bool single[S];                                        // removing the row empties its group
    float4 accs[S];
#pragma unroll
    for (int i = 0; i < S; i++) {
      nn[i] = (batch * S + i) * kRowsPerStep + sub;        // relative to row0
      has[i] = nn[i] < nrows;
      gzs[i] = -1;
      single[i] = false;
      if (loo && has[i]) {
        gzs[i] = z[nn[i]];
        if ((uint32_t)gzs[i] >= K) gzs[i] = -1;             // (an id outside the table: not assigned)
        single[i] = pri && gzs[i] >= 0 && __builtin_isinf(crp[kpad + gzs[i]]);
      }
      accs[i] = pri ? crp_prior4_lo(logcnt, logcnt_lo, single[i] ? le1_lo : le0_lo) : make_float4(0, 0, 0, 0);
    }
    uint32_t off = 0;
    for (int f = 0; f < nfeat; f++) {
      const FeatDesc &fd = feats[f];
      const float4 *tab = nlds + off + q;
      off += narrow_table_rows(fd) * L;
      if (fd.family == MSC_NOOP) continue;
      uint32_t raw[S];
      bool use[S];
#pragma unroll
      for (int i = 0; i < S; i++) {
        const uint64_t row = row0 + (has[i] ? nn[i] : 0);
        use[i] = has[i] && !(fd.mask != nullptr && fd.mask[row] != 0);
        raw[i] = fd.family == MSC_BB || fd.family == MSC_BBNC ? (uint32_t)(reinterpret_cast<const uint8_t *>(fd.col)[row] != 0)
                                                              : reinterpret_cast<const uint32_t *>(fd.col)[row];
      }
      switch (fd.family) {
        case MSC_BB: case MSC_BBNC:
#pragma unroll
          for (int i = 0; i < S; i++)
            if (use[i]) add4(accs[i], tab[raw[i] * L]);
          break;
        case MSC_DD:
#pragma unroll
          for (int i = 0; i < S; i++) {
            int v = (int)raw[i];
            v = v < 0 ? 0 : (v >= (int)fd.dim ? (int)fd.dim - 1 : v);
            if (use[i]) add4(accs[i], tab[v * L]);
          }
          break;
        case MSC_GP: case MSC_BNB:
#pragma unroll
          for (int i = 0; i < S; i++)
            if (use[i] && raw[i] < fd.vcap) add4(accs[i], tab[raw[i] * L]);   // (the host only takes this path when every count is in the table)
          break;
        case MSC_NICH: {
          const float4 mh = tab[NICH_MU_HI * L], ml = tab[NICH_MU_LO * L], c0 = tab[NICH_C0 * L], c1l = tab[NICH_C1LN2 * L],
                       c1 = tab[NICH_C1 * L], c2 = tab[NICH_C2 * L];
#pragma unroll
          for (int i = 0; i < S; i++) {
            if (!use[i]) continue;
            const float x = __uint_as_float(raw[i]);
            accs[i].x += nich_eval(x, mh.x, ml.x, c0.x, c1l.x, c1.x, c2.x);
            accs[i].y += nich_eval(x, mh.y, ml.y, c0.y, c1l.y, c1.y, c2.y);
            accs[i].z += nich_eval(x, mh.z, ml.z, c0.z, c1l.z, c1.z, c2.z);
            accs[i].w += nich_eval(x, mh.w, ml.w, c0.w, c1l.w, c1.w, c2.w);
          }
        } break;
        default: break;
      }
    }
#pragma unroll
    for (int si = 0; si < S; si++) {
    const uint64_t n = nn[si];
    const bool has_row = has[si];
    const int gz = gzs[si];
    float4 acc = accs[si];
    if (pri) add4(acc, crp_prior4(logcnt, single[si] ? le1 : le0));
    if (loo && gz >= 0 && (uint32_t)gz / 4 == (uint32_t)q) {     // this lane holds the row's own group
      const float v = own[n];
      const int c = gz & 3;
      acc.x = c == 0 ? v : acc.x; acc.y = c == 1 ? v : acc.y; acc.z = c == 2 ? v : acc.z; acc.w = c == 3 ? v : acc.w;
    }
    if (!SWEEP) {
      if (has_row) store_row(out, ld, n, kb, K, acc, vec_ok);
      continue;
    }
    // ---- draw: softmax + inverse CDF over the L lanes of the row (every lane of the wave takes part) ----
    float sc[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (kb + j >= K) sc[j] = -INFINITY;
    float m = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
    m = narrow_allreduce<L, true>(m);
    float p[4], sum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      p[j] = __builtin_amdgcn_exp2f((sc[j] - m) * 1.44269504088896340736f);
      sum += p[j];
    }
    float incl = sum;
    {                                                     // inclusive scan over the L lanes: row_shr by 1, 2, 4, 8 (zero fill)
      float t = dpp_f32<0x111, 0xf>(0.f, incl);
      if (q >= 1) incl += t;
      t = dpp_f32<0x112, 0xf>(0.f, incl);
      if (q >= 2) incl += t;
      if (L > 4) {
        t = dpp_f32<0x114, 0xf>(0.f, incl);
        if (q >= 4) incl += t;
      }
      if (L > 8) {
        t = dpp_f32<0x118, 0xf>(0.f, incl);
        if (q >= 8) incl += t;
      }
    }
    const float total = narrow_allreduce<L, false>(sum);
    const float dart = philox_uniform01(seed, sweep, row_id0 + n) * total;
    float c = incl - sum;
    int nmiss = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      c += p[j];
      nmiss += c < dart ? 1 : 0;
    }
    const unsigned long long hit = __builtin_amdgcn_ballot_w64(nmiss < 4);
    const unsigned long long mine = (hit >> (sub * L)) & ((L == 64) ? ~0ull : ((1ull << L) - 1ull));
    int pick = (int)K - 1;                               // rounding may let the dart fall off the end
    const int first = mine ? (int)__builtin_ctzll(mine) : 0;
    const int nm = __shfl(nmiss, sub * L + first, 64);
    if (mine) {
      const int k = 4 * first + nm;
      pick = k < (int)K ? k : (int)K - 1;
    }
    if (has_row && q == 0) z_out[n] = pick;
    }
  }
}

static size_t g_narrow_lds_limit = 64 * 1024;
template <int L, int S, bool SWEEP>
static int launch_narrow_s(hipStream_t stream, uint64_t gx, size_t lds_bytes, const FeatDesc *feats_dev, int nfeat, uint32_t K,
                           uint32_t kpad, uint64_t row0, uint64_t nrows, const int32_t *z, const float *own, const float *crp,
                           float *out, uint64_t ld, uint64_t row_id0, int32_t *z_out, const uint64_t *rng, ZeroSpans zero) {
  hipLaunchKernelGGL((k_narrow<L, S, SWEEP>), dim3((unsigned)(gx ? gx : 1)), dim3(256), lds_bytes, stream, feats_dev, nfeat, K, kpad,
                     row0, nrows, z, own, crp, out, ld, row_id0, z_out, rng, zero);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
template <int L, bool SWEEP>
static int launch_narrow_t(hipStream_t stream, int num_cus, size_t lds_bytes, const FeatDesc *feats_dev, int nfeat, uint32_t K,
                           uint32_t kpad, uint64_t row0, uint64_t nrows, const int32_t *z, const float *own, const float *crp,
                           float *out, uint64_t ld, uint64_t row_id0, int32_t *z_out, const uint64_t *rng, ZeroSpans zero) {
  if (lds_bytes > g_narrow_lds_limit) return -2;
  const uint64_t steps = (nrows + 64 / L - 1) / (64 / L);
  const uint64_t cap = (uint64_t)num_cus * 8;
  // 8 steps per wave visit once that still leaves ~4 waves per SIMD of work; few rows: one step each, spread out
  if (steps >= (uint64_t)num_cus * 16 * 8) {
    uint64_t gx = ((steps + 7) / 8 + 3) / 4;
    if (gx > cap) gx = cap;
    return launch_narrow_s<L, 8, SWEEP>(stream, gx, lds_bytes, feats_dev, nfeat, K, kpad, row0, nrows, z, own, crp, out, ld, row_id0, z_out, rng, zero);
  }
  uint64_t gx = (steps + 3) / 4;
  if (gx > cap) gx = cap;
  return launch_narrow_s<L, 1, SWEEP>(stream, gx, lds_bytes, feats_dev, nfeat, K, kpad, row0, nrows, z, own, crp, out, ld, row_id0, z_out, rng, zero);
}
// lanes_per_row: 4, 8 or 16 (abi.cpp narrow_lanes); table_rows: sum over features of narrow_table_rows
int launch_narrow(hipStream_t stream, int num_cus, int lanes_per_row, uint32_t table_rows, bool sweep, const FeatDesc *feats_dev,
                  int nfeat, uint32_t K, uint32_t kpad, uint64_t row0, uint64_t nrows, const int32_t *z, const float *own,
                  const float *crp, float *out, uint64_t ld, uint64_t row_id0, int32_t *z_out, const uint64_t *rng,
                  ZeroSpans zero) {
  const size_t lds = (size_t)table_rows * lanes_per_row * sizeof(float4);
#define MSC_NARROW(Lv)                                                                                                    \
  return sweep ? launch_narrow_t<Lv, true>(stream, num_cus, lds, feats_dev, nfeat, K, kpad, row0, nrows, z, own, crp, out, \
                                           ld, row_id0, z_out, rng, zero)                                                 \
               : launch_narrow_t<Lv, false>(stream, num_cus, lds, feats_dev, nfeat, K, kpad, row0, nrows, z, own, crp, out, \
                                            ld, row_id0, z_out, rng, zero)
  if (lanes_per_row == 4) { MSC_NARROW(4); }
  if (lanes_per_row == 8) { MSC_NARROW(8); }
  MSC_NARROW(16);
#undef MSC_NARROW
}

// ---------------------------------------------------------------------------
// fallback: one wave per row of a materialised [nrows, ld] score chunk.  The row is walked in tiles of 256 groups,
// lane l taking groups 4l .. 4l+3 of the tile (one coalesced 1 KiB load per tile and pass, the score kernels' own
// layout): maximum, then the total, then the walk to the first group whose running sum reaches the dart, which
// stops at the tile it lands in.  Totals and walk add up the same numbers in the same order, so the walk cannot
// miss by rounding.  (First version: lane l owned a contiguous K/64 slice -- 64 cache lines per load instruction,
// 42 ms per sweep of 500k rows x 2048 groups where the scores took 1 ms.)
// ---------------------------------------------------------------------------
MSC_DEV float4 load_score4(const float *__restrict__ s, uint32_t k, uint32_t K, bool vec_ok) {
  float4 v = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  if (vec_ok && k + 3 < K) return ld4(s + k);
  if (k < K) v.x = s[k];
  if (k + 1 < K) v.y = s[k + 1];
  if (k + 2 < K) v.z = s[k + 2];
  if (k + 3 < K) v.w = s[k + 3];
  return v;
}
// One pass of a row's NT tiles through registers (K <= 256 NT: up to 1024 groups), the next row's tiles fetched while
// this row is drawn: one memory round trip a row, overlapped, where the three passes below re-read the row for the
// maximum, the total and the walk (the first version of this kernel, which larger tables still take: 0.54 ms per
// million rows of 300 groups = 2.2 TB/s).  The same numbers added in the same order: the same draws.
template <int NT>
MSC_DEV void sample_rows_tiles(const float *__restrict__ scores, uint64_t ld, uint32_t K, uint64_t nrows, uint64_t row_id0,
                               int32_t *__restrict__ z, uint64_t seed, uint64_t sweep, bool vec_ok) {
  const int lane = threadIdx.x & 63;
  const uint64_t wave_id = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint64_t nwaves = (uint64_t)gridDim.x * 4;
  constexpr float kLog2e = 1.44269504088896340736f;
  float4 cur[NT], nxt[NT];
  auto fetch = [&](uint64_t row, float4 (&v)[NT]) {
    const float *s = scores + (row < nrows ? row : nrows - 1) * ld;
#pragma unroll
    for (int t = 0; t < NT; t++) v[t] = load_score4(s, (uint32_t)t * kGroupTile + 4 * lane, K, vec_ok);
  };
  if (wave_id < nrows) fetch(wave_id, cur);
  for (uint64_t row = wave_id; row < nrows; row += nwaves) {
    fetch(row + nwaves, nxt);                            // (past the end: the last row again, never used)
    if (NT == 1) {
      const float sc[4] = {cur[0].x, cur[0].y, cur[0].z, cur[0].w};
      const int pick1 = sample_from_scores<4>(sc, philox_uniform01(seed, sweep, row_id0 + row), lane, K);
      if (lane == 0) z[row] = pick1;
    } else {
      float m = -INFINITY;
#pragma unroll
      for (int t = 0; t < NT; t++) m = fmaxf(fmaxf(m, fmaxf(cur[t].x, cur[t].y)), fmaxf(cur[t].z, cur[t].w));
      m = wave_max(m);
      float p[NT][4], sum[NT], incl[NT];
      float total = 0.f;
#pragma unroll
      for (int t = 0; t < NT; t++) {
        p[t][0] = __builtin_amdgcn_exp2f((cur[t].x - m) * kLog2e);
        p[t][1] = __builtin_amdgcn_exp2f((cur[t].y - m) * kLog2e);
        p[t][2] = __builtin_amdgcn_exp2f((cur[t].z - m) * kLog2e);
        p[t][3] = __builtin_amdgcn_exp2f((cur[t].w - m) * kLog2e);
        sum[t] = ((p[t][0] + p[t][1]) + p[t][2]) + p[t][3];
        incl[t] = wave_incl_scan(sum[t], lane);
        total += lane_bcast(incl[t], 63);
      }
      const float dart = philox_uniform01(seed, sweep, row_id0 + row) * total;
      float before = 0.f;                                // running sum of the tiles already passed (wave-uniform)
      int pick = (int)K - 1;                             // rounding may let the dart fall off the end (util.hpp:155)
      bool found = false;
#pragma unroll
      for (int t = 0; t < NT; t++) {
        const float tile_total = lane_bcast(incl[t], 63);
        if (!found && before + tile_total >= dart) {     // (wave-uniform) the dart lands in this tile
          float c = before + (incl[t] - sum[t]);
          int nmiss = 0;
          c += p[t][0]; nmiss += c < dart ? 1 : 0;
          c += p[t][1]; nmiss += c < dart ? 1 : 0;
          c += p[t][2]; nmiss += c < dart ? 1 : 0;
          c += p[t][3]; nmiss += c < dart ? 1 : 0;
          const unsigned long long hit = __builtin_amdgcn_ballot_w64(nmiss < 4);
          if (hit != 0ull) {
            const int l = (int)__builtin_ctzll(hit);
            const int k = t * kGroupTile + 4 * l + lane_bcast(nmiss, l);
            pick = k < (int)K ? k : (int)K - 1;
            found = true;
          }
        }
        before += tile_total;
      }
      if (lane == 0) z[row] = pick;
    }
#pragma unroll
    for (int t = 0; t < NT; t++) cur[t] = nxt[t];
  }
}

__global__ __launch_bounds__(256) void k_sample_rows(const float *__restrict__ scores, uint64_t ld,
                                                      uint32_t K, uint64_t nrows, uint64_t row_id0,
                                                      int32_t *__restrict__ z,
                                                      const uint64_t *__restrict__ rng) {
  const uint64_t seed = rng[0], sweep = rng[1];
  const int lane = threadIdx.x & 63;
  const uint64_t wave_id = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint64_t nwaves = (uint64_t)gridDim.x * 4;
  const uint32_t ntiles = (K + kGroupTile - 1) / kGroupTile;
  const bool vec_ok = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(scores) & 15) == 0);
  constexpr float kLog2e = 1.44269504088896340736f;
  if (ntiles == 1) return sample_rows_tiles<1>(scores, ld, K, nrows, row_id0, z, seed, sweep, vec_ok);
  if (ntiles == 2) return sample_rows_tiles<2>(scores, ld, K, nrows, row_id0, z, seed, sweep, vec_ok);
  if (ntiles == 3) return sample_rows_tiles<3>(scores, ld, K, nrows, row_id0, z, seed, sweep, vec_ok);
  if (ntiles == 4) return sample_rows_tiles<4>(scores, ld, K, nrows, row_id0, z, seed, sweep, vec_ok);
  for (uint64_t row = wave_id; row < nrows; row += nwaves) {
    const float *s = scores + row * ld;
    float m = -INFINITY;
    for (uint32_t t = 0; t < ntiles; t++) {
      const float4 v = load_score4(s, t * kGroupTile + 4 * lane, K, vec_ok);
      m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
    m = wave_max(m);
    float total = 0.f;
    for (uint32_t t = 0; t < ntiles; t++) {
      const float4 v = load_score4(s, t * kGroupTile + 4 * lane, K, vec_ok);
      const float sum = ((__builtin_amdgcn_exp2f((v.x - m) * kLog2e) + __builtin_amdgcn_exp2f((v.y - m) * kLog2e)) +
                         __builtin_amdgcn_exp2f((v.z - m) * kLog2e)) + __builtin_amdgcn_exp2f((v.w - m) * kLog2e);
      total += lane_bcast(wave_incl_scan(sum, lane), 63);
    }
    const float dart = philox_uniform01(seed, sweep, row_id0 + row) * total;
    float before = 0.f;                                  // running sum of the tiles already passed (wave-uniform)
    int pick = (int)K - 1;                               // rounding may let the dart fall off the end (util.hpp:155)
    for (uint32_t t = 0; t < ntiles; t++) {
      const float4 v = load_score4(s, t * kGroupTile + 4 * lane, K, vec_ok);
      const float p0 = __builtin_amdgcn_exp2f((v.x - m) * kLog2e), p1 = __builtin_amdgcn_exp2f((v.y - m) * kLog2e),
                  p2 = __builtin_amdgcn_exp2f((v.z - m) * kLog2e), p3 = __builtin_amdgcn_exp2f((v.w - m) * kLog2e);
      const float sum = ((p0 + p1) + p2) + p3;
      const float incl = wave_incl_scan(sum, lane);
      const float tile_total = lane_bcast(incl, 63);
      if (before + tile_total >= dart) {                 // (wave-uniform) the dart lands in this tile
        float c = before + (incl - sum);
        int nmiss = 0;
        c += p0; nmiss += c < dart ? 1 : 0;
        c += p1; nmiss += c < dart ? 1 : 0;
        c += p2; nmiss += c < dart ? 1 : 0;
        c += p3; nmiss += c < dart ? 1 : 0;
        const unsigned long long hit = __builtin_amdgcn_ballot_w64(nmiss < 4);
        if (hit != 0ull) {
          const int l = (int)__builtin_ctzll(hit);
          const int k = (int)(t * kGroupTile) + 4 * l + lane_bcast(nmiss, l);
          pick = k < (int)K ? k : (int)K - 1;
          break;
        }
      }
      before += tile_total;
    }
    if (lane == 0) z[row] = pick;
  }
}

// ---------------------------------------------------------------------------
static uint64_t grid_for(uint64_t work_items_per_wave_chunk, int num_cus, int waves_per_cu_cap) {
  uint64_t gx = (work_items_per_wave_chunk + 3) / 4;
  const uint64_t cap = (uint64_t)num_cus * waves_per_cu_cap / 4;
  if (gx > cap) gx = cap;
  return gx ? gx : 1;
}

size_t sweep_nich1_rows_table_floats(uint32_t kpad) {
  const RowsGeom g = rows_geom(kpad);
  return ((size_t)g.nb * g.B + 8) * RT_WORDS + 16;       // (eight spare entries: the streaming pass fetches ahead)
}
uint32_t sweep_nich1_rows_max_groups() { return kRowsMaxK; }
int launch_sweep_nich1_rows(hipStream_t stream, int num_cus, const FeatDesc *feats_dev, uint32_t K, uint32_t kpad,
                            uint64_t row0, uint64_t nrows, uint64_t row_id0, int32_t *z, const float *crp,
                            const uint64_t *rng, ZeroSpans zero, float *table) {
  const RowsGeom g = rows_geom(kpad);
  hipLaunchKernelGGL(k_nich_rows_table, dim3(1), dim3(1024), 0, stream, feats_dev, K, kpad, crp, table, g.nb * g.B);
  uint64_t gx = (nrows + 255) / 256;                       // two waves a workgroup, 128 rows a wave visit
  const uint64_t cap = (uint64_t)num_cus * 10;             // (16 KiB of block sums per workgroup: ten per CU)
  if (gx > cap) gx = cap;
  if (gx == 0) gx = 1;
  hipLaunchKernelGGL(k_sweep_nich1_rows, dim3((unsigned)gx), dim3(128), 0, stream, feats_dev, K, kpad, row0, nrows, row_id0,
                     z, crp, rng, zero, table, g.B, g.nb);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_sweep_nich1(hipStream_t stream, int num_cus, const FeatDesc *feats_dev, uint32_t K,
                       uint32_t kpad, uint64_t row0, uint64_t nrows, uint64_t row_id0, int32_t *z,
                       const float *own, const float *crp, const uint64_t *rng, ZeroSpans zero) {
  // enough rows for a 32-row chunk per wave over most of the chip: the transposed draw (k_sweep_nich1_t)
  // MSC_SWEEP_NICH1 = 1 / 2 pins the row-at-a-time / the transposed kernel whatever the size (tests, A/B timing)
  const char *pin = std::getenv("MSC_SWEEP_NICH1");
  const int which = pin ? std::atoi(pin) : 0;
  // (measured: the transposed kernel is ahead from ~16 k rows on at K = 256 and at K = 1024 alike -- 4 k rows: 15 vs 11 us,
  // 16 k: 15 vs 18, 64 k: 19 vs 30, 512 k: 74 vs 120)
  const bool transposed = which == 2 || (which != 1 && nrows >= (uint64_t)num_cus * 64);
  if (transposed && K <= 1024) {
    // one wave per slot the registers leave (4 / 3 / 2 a SIMD for G <= 4 / 8 / 16): the waves split the rows evenly
    const uint64_t gxt = grid_for((nrows + kTRows - 1) / kTRows, num_cus, K <= 256 ? 16 : K <= 512 ? 12 : 8);
    const dim3 gridt((unsigned)gxt), blockt(256);
    if (K <= 64)
      hipLaunchKernelGGL(k_sweep_nich1_t<1>, (note_kernel(1, "k_sweep_nich1_t<1>"), gridt), blockt, 0, stream, feats_dev, K, kpad, row0, nrows, row_id0, z, crp, rng, zero);
    else if (K <= 128)
      hipLaunchKernelGGL(k_sweep_nich1_t<2>, (note_kernel(1, "k_sweep_nich1_t<2>"), gridt), blockt, 0, stream, feats_dev, K, kpad, row0, nrows, row_id0, z, crp, rng, zero);
    else if (K <= 256)
      hipLaunchKernelGGL(k_sweep_nich1_t<4>, (note_kernel(1, "k_sweep_nich1_t<4>"), gridt), blockt, 0, stream, feats_dev, K, kpad, row0, nrows, row_id0, z, crp, rng, zero);
    else if (K <= 512)
      hipLaunchKernelGGL(k_sweep_nich1_t<8>, (note_kernel(1, "k_sweep_nich1_t<8>"), gridt), blockt, 0, stream, feats_dev, K, kpad, row0, nrows, row_id0, z, crp, rng, zero);
    else
      hipLaunchKernelGGL(k_sweep_nich1_t<16>, (note_kernel(1, "k_sweep_nich1_t<16>"), gridt), blockt, 0, stream, feats_dev, K, kpad, row0, nrows, row_id0, z, crp, rng, zero);
    return hipGetLastError() == hipSuccess ? 0 : -1;
  }
  // 64 rows per wave visit once there are enough rows for ~8 waves per SIMD; fewer rows: halve the visit down to 4
  int chunk_rows = 64;
  while (chunk_rows > 4 && (nrows + chunk_rows - 1) / chunk_rows < (uint64_t)num_cus * 32) chunk_rows >>= 1;
  const uint64_t gx = grid_for((nrows + chunk_rows - 1) / chunk_rows, num_cus, 16);
  const dim3 grid((unsigned)gx), block(256);
  if (K <= 64)
    hipLaunchKernelGGL(k_sweep_nich1<1>, (note_kernel(1, "k_sweep_nich1<1>"), grid), block, 0, stream, feats_dev, K, kpad, row0, nrows, row_id0, z, own, crp, rng, chunk_rows, zero);
  else if (K <= 128)
    hipLaunchKernelGGL(k_sweep_nich1<2>, (note_kernel(1, "k_sweep_nich1<2>"), grid), block, 0, stream, feats_dev, K, kpad, row0, nrows, row_id0, z, own, crp, rng, chunk_rows, zero);
  else if (K <= 256)
    hipLaunchKernelGGL(k_sweep_nich1<4>, (note_kernel(1, "k_sweep_nich1<4>"), grid), block, 0, stream, feats_dev, K, kpad, row0, nrows, row_id0, z, own, crp, rng, chunk_rows, zero);
  else if (K <= 512)
    hipLaunchKernelGGL(k_sweep_nich1<8>, (note_kernel(1, "k_sweep_nich1<8>"), grid), block, 0, stream, feats_dev, K, kpad, row0, nrows, row_id0, z, own, crp, rng, chunk_rows, zero);
  else if (K <= 1024)
    hipLaunchKernelGGL(k_sweep_nich1<16>, (note_kernel(1, "k_sweep_nich1<16>"), grid), block, 0, stream, feats_dev, K, kpad, row0, nrows, row_id0, z, own, crp, rng, chunk_rows, zero);
  else
    return -2;
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_sweep_mixed(hipStream_t stream, int num_cus, bool has_dm, bool roles_ok, bool pair, bool nich_only, bool lookups_only, const FeatDesc *feats_dev, int nfeat, int nsplit,
                       uint32_t K, uint32_t kpad, uint64_t row0, uint64_t nrows, uint64_t row_id0, int32_t *z,
                       const float *own, const float *crp, const uint64_t *rng, ZeroSpans zero) {
  if (K > 256) return -2;
  // few rows: 2 rows per wave instead of 8, so that the chunks -- each a serial chain of feature lookups and R
  // draws -- spread over the chip instead of queueing in a quarter of it (N = 10k, 12 features: 26 -> ? us)
  // (and 4 rows per wave only while the 64-row workgroups fit one round themselves)
  const bool small = !has_dm && (nrows + 63) / 64 <= (uint64_t)num_cus;
  const bool small4 = small && (nrows + 31) / 32 > (uint64_t)num_cus;      // (2 rows per wave would need a second round)
  const uint64_t rows_per_wg = has_dm ? 64 : small4 ? 64 : small ? 32 : 128;
  uint64_t gx = (nrows + rows_per_wg - 1) / rows_per_wg;
  const uint64_t cap = (uint64_t)num_cus * 4;
  if (gx > cap) gx = cap;
  const dim3 grid((unsigned)(gx ? gx : 1));
  if (has_dm)
    hipLaunchKernelGGL((k_sweep_tile<8, 8, true>), (note_kernel(1, "k_sweep_tile<8, 8, true>"), grid), dim3(512), 0, stream, feats_dev, nfeat, nsplit, K, kpad, row0, nrows,
                       row_id0, z, own, crp, rng, zero);
  else if (lookups_only && pair && K <= 128)
    hipLaunchKernelGGL((k_sweep_lookups<true>), (note_kernel(1, "k_sweep_lookups<true, 0>"), dim3((unsigned)std::max<uint64_t>(1, std::min<uint64_t>((nrows + 511) / 512, cap)))), dim3(1024), 0, stream,
                       feats_dev, nfeat, K, kpad, row0, nrows, row_id0, z, own, crp, rng, zero);
  else if (lookups_only && !small && !pair)
    hipLaunchKernelGGL((k_sweep_lookups<false>), (note_kernel(1, "k_sweep_lookups<false, 0>"), dim3((unsigned)std::max<uint64_t>(1, std::min<uint64_t>((nrows + 255) / 256, cap)))), dim3(1024), 0, stream,
                       feats_dev, nfeat, K, kpad, row0, nrows, row_id0, z, own, crp, rng, zero);
  else if (nich_only && pair && K <= 128)                 // (PAIR follows the view's rows, whatever this call's are: see below)
  {
    const dim3 g((unsigned)std::max<uint64_t>(1, std::min<uint64_t>((nrows + 32 * kNichPackWaves - 1) / (32 * kNichPackWaves), cap * (16 / kNichPackWaves))));
    if (nsplit > 0)
      hipLaunchKernelGGL((k_sweep_nich_pack<true, true>), (note_kernel(1, "k_sweep_nich_pack<true, true, 0>"), g), dim3(kNichPackWaves * 64), 0, stream, feats_dev, nsplit, K, kpad, row0, nrows, row_id0, z, own, crp, rng, zero);
    else
      hipLaunchKernelGGL((k_sweep_nich_pack<true, false>), (note_kernel(1, "k_sweep_nich_pack<true, false, 0>"), g), dim3(kNichPackWaves * 64), 0, stream, feats_dev, nsplit, K, kpad, row0, nrows, row_id0, z, own, crp, rng, zero);
  }
  else if (nich_only && !small && !pair)
  {
    const dim3 g((unsigned)std::max<uint64_t>(1, std::min<uint64_t>((nrows + 16 * kNichPackWaves - 1) / (16 * kNichPackWaves), cap * (16 / kNichPackWaves))));
    if (nsplit > 0)
      hipLaunchKernelGGL((k_sweep_nich_pack<false, true>), (note_kernel(1, "k_sweep_nich_pack<false, true, 0>"), g), dim3(kNichPackWaves * 64), 0, stream, feats_dev, nsplit, K, kpad, row0, nrows, row_id0, z, own, crp, rng, zero);
    else
      hipLaunchKernelGGL((k_sweep_nich_pack<false, false>), (note_kernel(1, "k_sweep_nich_pack<false, false, 0>"), g), dim3(kNichPackWaves * 64), 0, stream, feats_dev, nsplit, K, kpad, row0, nrows, row_id0, z, own, crp, rng, zero);
  }
  else if (pair && roles_ok && K <= 128)
    // PAIR mode (abi.cpp decides on the bound view's rows, not this call's: its draw sums a row's entries two to a lane
    // where the other tile kernels sum four, so every row range of a view must take the same one)
    hipLaunchKernelGGL((k_sweep_tile_roles<0, true>), (note_kernel(1, "k_sweep_tile_roles<0, true>"), dim3((unsigned)std::max<uint64_t>(1, std::min<uint64_t>((nrows + 255) / 256, cap)))), dim3(1024), 0,
                       stream, feats_dev, nfeat, nsplit, K, kpad, row0, nrows, row_id0, z, own, crp, rng, zero, static_cast<const float *>(nullptr));
  else if (small4)
    hipLaunchKernelGGL((k_sweep_tile<4, 16, false>), (note_kernel(1, "k_sweep_tile<4, 16, false>"), grid), dim3(1024), 0, stream, feats_dev, nfeat, nsplit, K, kpad, row0, nrows,
                       row_id0, z, own, crp, rng, zero);
  else if (small)
    hipLaunchKernelGGL((k_sweep_tile<2, 16, false>), (note_kernel(1, "k_sweep_tile<2, 16, false>"), grid), dim3(1024), 0, stream, feats_dev, nfeat, nsplit, K, kpad, row0, nrows,
                       row_id0, z, own, crp, rng, zero);
  else if (roles_ok && tile_roles_enabled())
    hipLaunchKernelGGL(k_sweep_tile_roles<0>, (note_kernel(1, "k_sweep_tile_roles<0, false>"), grid), dim3(1024), 0, stream, feats_dev, nfeat, nsplit, K, kpad, row0, nrows,
                       row_id0, z, own, crp, rng, zero, static_cast<const float *>(nullptr));
  else
    hipLaunchKernelGGL((k_sweep_tile<8, 16, false>), (note_kernel(1, "k_sweep_tile<8, 16, false>"), grid), dim3(1024), 0, stream, feats_dev, nfeat, nsplit, K, kpad, row0, nrows,
                       row_id0, z, own, crp, rng, zero);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// 256 < K <= 384 (abi.cpp decides on the bound view's row count, so that a shard draws from the same bits as the whole): the role-split kernel
// over the full tile, the tail's scores from `tail` (k_score_tail_rows wrote them)
// (kind: 0 the role-split kernel, 1 the nich-only kernel -- with a few lookups when nsplit > 0 --, 2 the lookups-only kernel)
int launch_sweep_roles_tail(hipStream_t stream, int num_cus, int kind, const FeatDesc *feats_dev, int nfeat, int nsplit, uint32_t K,
                            uint32_t kpad, uint64_t row0, uint64_t nrows, uint64_t row_id0, int32_t *z, const float *own,
                            const float *crp, const uint64_t *rng, ZeroSpans zero, const float *tail) {
  uint64_t gx = (nrows + 127) / 128;
  const uint64_t cap = (uint64_t)num_cus * 4;
  if (gx > cap) gx = cap;
  const bool wide = K > (uint32_t)kGroupTile + 64;
  if (kind == 2) {
    const dim3 g((unsigned)std::max<uint64_t>(1, std::min<uint64_t>((nrows + 255) / 256, cap)));
    if (!wide) hipLaunchKernelGGL((k_sweep_lookups<false, 1>), (note_kernel(1, "k_sweep_lookups<false, 1>"), g), dim3(1024), 0, stream, feats_dev, nfeat, K, kpad, row0, nrows, row_id0, z, own, crp, rng, zero, tail);
    else hipLaunchKernelGGL((k_sweep_lookups<false, 2>), (note_kernel(1, "k_sweep_lookups<false, 2>"), g), dim3(1024), 0, stream, feats_dev, nfeat, K, kpad, row0, nrows, row_id0, z, own, crp, rng, zero, tail);
    return hipGetLastError() == hipSuccess ? 0 : -1;
  }
  if (kind == 1) {
    const dim3 g((unsigned)std::max<uint64_t>(1, std::min<uint64_t>((nrows + 16 * kNichPackWaves - 1) / (16 * kNichPackWaves), cap * (16 / kNichPackWaves))));
    const dim3 b(kNichPackWaves * 64);
    if (nsplit > 0) {
      if (!wide) hipLaunchKernelGGL((k_sweep_nich_pack<false, true, 1>), (note_kernel(1, "k_sweep_nich_pack<false, true, 1>"), g), b, 0, stream, feats_dev, nsplit, K, kpad, row0, nrows, row_id0, z, own, crp, rng, zero, tail);
      else hipLaunchKernelGGL((k_sweep_nich_pack<false, true, 2>), (note_kernel(1, "k_sweep_nich_pack<false, true, 2>"), g), b, 0, stream, feats_dev, nsplit, K, kpad, row0, nrows, row_id0, z, own, crp, rng, zero, tail);
    } else {
      if (!wide) hipLaunchKernelGGL((k_sweep_nich_pack<false, false, 1>), (note_kernel(1, "k_sweep_nich_pack<false, false, 1>"), g), b, 0, stream, feats_dev, nsplit, K, kpad, row0, nrows, row_id0, z, own, crp, rng, zero, tail);
      else hipLaunchKernelGGL((k_sweep_nich_pack<false, false, 2>), (note_kernel(1, "k_sweep_nich_pack<false, false, 2>"), g), b, 0, stream, feats_dev, nsplit, K, kpad, row0, nrows, row_id0, z, own, crp, rng, zero, tail);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
  }
  if (K <= (uint32_t)kGroupTile + 64)
    hipLaunchKernelGGL(k_sweep_tile_roles<1>, (note_kernel(1, "k_sweep_tile_roles<1, false>"), dim3((unsigned)(gx ? gx : 1))), dim3(1024), 0, stream, feats_dev, nfeat, nsplit, K, kpad,
                       row0, nrows, row_id0, z, own, crp, rng, zero, tail);
  else
    hipLaunchKernelGGL(k_sweep_tile_roles<2>, (note_kernel(1, "k_sweep_tile_roles<2, false>"), dim3((unsigned)(gx ? gx : 1))), dim3(1024), 0, stream, feats_dev, nfeat, nsplit, K, kpad,
                       row0, nrows, row_id0, z, own, crp, rng, zero, tail);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_sample_rows(hipStream_t stream, int num_cus, const float *scores, uint64_t ld, uint32_t K,
                       uint64_t nrows, uint64_t row_id0, int32_t *z, const uint64_t *rng) {
  const uint64_t gx = grid_for(nrows, num_cus, 32);
  hipLaunchKernelGGL(k_sample_rows, dim3((unsigned)gx), dim3(256), 0, stream, scores, ld, K, nrows, row_id0,
                     z, rng);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// the (seed, sweep index) pair of a state lives on the device: a sweep step captured as a graph replays with the
// index the previous step left behind
__global__ void k_rng_set(uint64_t *rng, uint64_t seed, uint64_t sweep) {
  rng[0] = seed;
  rng[1] = sweep;
}
__global__ void k_rng_bump(uint64_t *rng) { rng[1] += 1; }
int launch_rng_set(hipStream_t stream, uint64_t *rng, uint64_t seed, uint64_t sweep) {
  hipLaunchKernelGGL(k_rng_set, dim3(1), dim3(1), 0, stream, rng, seed, sweep);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int launch_rng_bump(hipStream_t stream, uint64_t *rng) {
  hipLaunchKernelGGL(k_rng_bump, dim3(1), dim3(1), 0, stream, rng);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace msc
