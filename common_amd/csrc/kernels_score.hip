// kernels_score.hip -- gfx950 kernels for the scoring side of the hot path:
//   k_prepare       suff-stats -> per-group float score constants / tables (double math)
//   k_crp_prepare   group sizes -> log pseudocounts (group_manager.hpp:274-283)
//   k_loo_own       per row: score of the row against its own group with the row removed
//                   (remove_value then score_value), summed over features, in double
//   k_score_nich1   one NICH feature, [nrows x K] scores, constants in VGPRs, streaming stores
//   k_score_tile    any feature list; the tables of a host-planned group of features are copied to LDS by
//                   global_load_lds, one barrier per group (score_block.hpp); scores summed over
//                   features in registers, one store per row
//   k_dm_prepare    the dim+1 exact count tables of a Dirichlet-Multinomial feature
//   k_gp_large_fix  gp / bnb / dm counts beyond the exact tables, in double (gp: Loader's saddle-point form)
//
// Mapping used by every score kernel: a wave owns a block of rows and one k-tile of 256
// groups; lane l owns groups 4l..4l+3 of the tile, so a row of the tile is one 16-byte value
// per lane = one 1 KiB contiguous store per wave (HBM-write-bound configs need nothing else on
// the critical path).  Row values are loaded coalesced (lane r <- row r of the block) and
// broadcast with v_readlane.
#include <cstdlib>
#include <type_traits>

#include "commit_ops.hpp"
#include "family_math.hpp"
#include "launchers.hpp"
#include "score_block.hpp"

namespace msc {

// ---------------------------------------------------------------------------
// prepare: one thread per (feature, group slot); pads (k >= K) are prepared from their
// zeroed raw stats so that vector loads of a full tile stay finite.
// ---------------------------------------------------------------------------
// (z, nz): the table rows of the count / categorical families are dealt out over nz threads per (feature, group) --
// a column with counts up to 1000 would otherwise be ~2000 lgamma chains in a row per thread; what a group needs
// once is done by z = 0
MSC_DEV void prepare_group(const FeatDesc &fd, uint32_t k, uint32_t kpad, uint32_t z, uint32_t nz) {
  if (z != 0 && fd.family != MSC_GP && fd.family != MSC_BNB && fd.family != MSC_DD) return;
  switch (fd.family) {
    // (every lookup family keeps one table row of zeros right after its last entry: what a masked value of a column with
    // the mask folded in selects, FeatDesc::col_sentinel)
    case MSC_BB: {
      float s0, s1;
      bb_prepare(fd.hp, fd.raw_u32[k], fd.raw_u32[kpad + k], s0, s1);
      fd.tab[k] = s0;
      fd.tab[kpad + k] = s1;
      fd.tab[2 * (size_t)kpad + k] = 0.f;
      if (fd.loo_tab != nullptr) fd.loo_tab[2 * (size_t)kpad + k] = 0.f;
      if (fd.loo_tab != nullptr) {          // (an entry is only read for a row that is in the group with that value)
        const uint32_t h = fd.raw_u32[k], t = fd.raw_u32[kpad + k];
        fd.loo_tab[k] = t ? (float)bb_loo(fd.hp, h, t, false) : 0.f;
        fd.loo_tab[kpad + k] = h ? (float)bb_loo(fd.hp, h, t, true) : 0.f;
      }
    } break;
    case MSC_BBNC: {
      float s0, s1;
      bbnc_prepare(fd.raw_f32[k], s0, s1);
      fd.tab[k] = s0;
      fd.tab[kpad + k] = s1;
      fd.tab[2 * (size_t)kpad + k] = 0.f;
    } break;
    case MSC_GP: {
      const uint32_t cnt = fd.raw_u32[k], sum = fd.raw_u32[kpad + k];
      if (z == 0) {
        gp_prepare_consts(fd.hp, cnt, sum, fd.tab[(size_t)GP_NSE_HI * kpad + k], fd.tab[(size_t)GP_NSE_LO * kpad + k]);
        fd.tab[(size_t)(GP_T0 + fd.vcap) * kpad + k] = 0.f;
        if (fd.loo_tab != nullptr) fd.loo_tab[(size_t)fd.vcap * kpad + k] = 0.f;
      }
      for (uint32_t v = z; v < fd.vcap; v += nz)
        fd.tab[(size_t)(GP_T0 + v) * kpad + k] = gp_prepare_table(fd.hp, cnt, sum, v);
      if (fd.loo_tab != nullptr)
        for (uint32_t v = z; v < fd.vcap; v += nz)
          fd.loo_tab[(size_t)v * kpad + k] = (cnt >= 1 && sum >= v) ? (float)gp_loo(fd.hp, cnt, sum, v) : 0.f;
    } break;
    case MSC_BNB: {
      const double cnt = fd.raw_u32[k], sum = fd.raw_u32[kpad + k];
      if (z == 0) {
        fd.tab[(size_t)(GP_T0 + fd.vcap) * kpad + k] = 0.f;
        if (fd.loo_tab != nullptr) fd.loo_tab[(size_t)fd.vcap * kpad + k] = 0.f;
      }
      for (uint32_t v = z; v < fd.vcap; v += nz)
        fd.tab[(size_t)(GP_T0 + v) * kpad + k] = (float)bnb_score(fd.hp, cnt, sum, (double)v);
      if (fd.loo_tab != nullptr)
        for (uint32_t v = z; v < fd.vcap; v += nz)
          fd.loo_tab[(size_t)v * kpad + k] = (cnt >= 1.0 && sum >= (double)v) ? (float)bnb_score(fd.hp, cnt - 1.0, sum - (double)v, (double)v) : 0.f;
    } break;
    case MSC_DD: {
      const uint32_t csum = fd.raw_u32[k];
      if (z == 0) {
        fd.tab[(size_t)fd.dim * kpad + k] = 0.f;
        if (fd.loo_tab != nullptr) fd.loo_tab[(size_t)fd.dim * kpad + k] = 0.f;
      }
      for (uint32_t i = z; i < fd.dim; i += nz)
        fd.tab[(size_t)i * kpad + k] =
            dd_prepare_entry(fd.hp[i], fd.raw_u32[(size_t)(1 + i) * kpad + k], fd.aux, csum);
      if (fd.loo_tab != nullptr)
        for (uint32_t i = z; i < fd.dim; i += nz) {
          const uint32_t c = fd.raw_u32[(size_t)(1 + i) * kpad + k];
          fd.loo_tab[(size_t)i * kpad + k] = c ? (float)dd_loo(fd.hp[i], c, fd.aux, csum) : 0.f;
        }
    } break;
    case MSC_NICH: {
      float o[NICH_ROWS];
      nich_prepare(fd.hp, fd.raw_u32[k], fd.raw_f32[k], fd.raw_f32[kpad + k], o);
#pragma unroll
      for (int i = 0; i < NICH_ROWS; i++) fd.tab[(size_t)i * kpad + k] = o[i];
      if (fd.loo64 != nullptr) nich_loo_prepare(fd.hp, fd.raw_u32[k], fd.raw_f32[k], fd.raw_f32[kpad + k], fd.loo64 + (size_t)k * kNlooStride, 1);
    } break;
    default: break;
  }
}

__global__ __launch_bounds__(256) void k_prepare(const FeatDesc *__restrict__ feats, uint32_t kpad) {
  const uint32_t k = blockIdx.x * 256 + threadIdx.x;
  if (k >= kpad) return;
  prepare_group(feats[blockIdx.y], k, kpad, blockIdx.z, gridDim.z);
}

// dm tables: blockIdx.y = stage (category i < dim, or dim = the row total); one thread per group slot
__global__ __launch_bounds__(256) void k_dm_prepare(const FeatDesc *__restrict__ feats, int f, uint32_t kpad) {
  const FeatDesc fd = feats[f];
  const uint32_t k = blockIdx.x * 256 + threadIdx.x, sub = blockIdx.y;
  if (k >= kpad || fd.dm_meta == nullptr) return;
  const uint32_t first = fd.dm_meta[2 * sub], rows = fd.dm_meta[2 * sub + 1];
  float *t = fd.tab + (size_t)first * kpad + k;         // count v: rows 2v (hi) and 2v + 1 (lo)
  double a = 0, n = 0;
  if (sub < fd.dim) {
    a = fd.hp[sub];
    n = fd.raw_u32[(size_t)sub * kpad + k];
  } else {
    for (uint32_t i = 0; i < fd.dim; i++) n += (double)fd.raw_u32[(size_t)i * kpad + k];
  }
  double *tl = fd.loo64 != nullptr ? fd.loo64 + (size_t)(first / 2) * kpad + k : nullptr;   // entry v of this stage
  for (uint32_t v = blockIdx.z; v < rows; v += gridDim.z) {   // (table rows dealt out over the z-slices, as in k_prepare)
    const double term = sub < fd.dim ? dm_cat_term(a, n, (double)v) : dm_sum_term(fd.aux, n, (double)v);
    dm_split(term, t[(size_t)(2 * v) * kpad], t[(size_t)(2 * v + 1) * kpad]);
    // the same count against the group without it (only read for a row that is in the group, so n >= v there)
    if (tl != nullptr)
      tl[(size_t)v * kpad] = n >= (double)v ? (sub < fd.dim ? dm_cat_term(a, n - (double)v, (double)v) : dm_sum_term(fd.aux, n - (double)v, (double)v))
                                            : 0.0;
  }
}

// crp layout (float): [0,kpad) log(cnt) or -inf when empty; [kpad,2kpad) log(cnt-1) or -inf;
// [2kpad] = log(alpha / n_empty), [2kpad+1] = log(alpha / (n_empty+1)).  Every term is a (hi, lo) pair: the float above
// is hi and lo = the rest of the double value sits at [2kpad+2], [2kpad+3] for the two "empty" terms, at
// [2kpad+4, 3kpad+4) for log(cnt) and at [3kpad+4, 4kpad+4) for log(cnt-1) (msc_internal.hpp crp_lo_*).  A group of
// 40k rows has log(cnt) = 10.6 while likelihood + prior is O(1): the float alone is off by up to 4.8e-7 of that
// result, the pair by 1e-14.  One block.
MSC_DEV void crp_split(double v, float &hi, float &lo) {
  hi = (float)v;
  lo = __builtin_isinf(hi) ? 0.f : (float)(v - (double)hi);
}
MSC_DEV void crp_prepare_block(const uint32_t *cnt, uint32_t K, uint32_t kpad, float alpha,
                               float *__restrict__ crp) {
  __shared__ uint32_t s_empty;
  if (threadIdx.x == 0) s_empty = 0;
  __syncthreads();
  uint32_t mine = 0;
  float *lo0 = crp + crp_lo_cnt(kpad), *lo1 = crp + crp_lo_cntm1(kpad);
  for (uint32_t k = threadIdx.x; k < kpad; k += 256) {
    const uint32_t c = k < K ? cnt[k] : 0;
    crp_split(c ? log((double)c) : -(double)INFINITY, crp[k], lo0[k]);
    crp_split(c > 1 ? log((double)c - 1.0) : -(double)INFINITY, crp[kpad + k], lo1[k]);
    if (k < K && c == 0) mine++;
  }
  atomicAdd(&s_empty, mine);
  __syncthreads();
  if (threadIdx.x == 0) {
    const double ne = s_empty;
    crp_split(ne > 0 ? log((double)alpha / ne) : -(double)INFINITY, crp[2 * (size_t)kpad], crp[2 * (size_t)kpad + 2]);
    crp_split(log((double)alpha / (ne + 1.0)), crp[2 * (size_t)kpad + 1], crp[2 * (size_t)kpad + 3]);
  }
}
__global__ __launch_bounds__(256) void k_crp_prepare(const uint32_t *__restrict__ cnt, uint32_t K,
                                                      uint32_t kpad, float alpha, float *crp) {
  crp_prepare_block(cnt, K, kpad, alpha, crp);
}

// The tail of a sweep step in one launch: commit (additive -> raw) and prepare (raw -> score tables) of every
// (feature, group), by the same thread so no grid-wide ordering is needed; the extra y-slice commits the group
// sizes and derives the CRP terms from them (one block walks all K), and moves the step's random stream on.
__global__ __launch_bounds__(256) void k_commit_prepare(const FeatDesc *__restrict__ feats, int nfeat, uint32_t K,
                                                         uint32_t kpad, const long long *__restrict__ cnt_acc,
                                                         uint32_t *__restrict__ cnt_u32, float alpha,
                                                         float *__restrict__ crp, uint64_t *__restrict__ rng_bump) {
  if ((int)blockIdx.y == nfeat) {
    if (blockIdx.x != 0 || blockIdx.z != 0) return;
    for (uint32_t k = threadIdx.x; k < kpad; k += 256) cnt_u32[k] = (uint32_t)cnt_acc[k];
    __syncthreads();                                   // (each thread reads back what it wrote; the barrier is for s_empty's init)
    crp_prepare_block(cnt_u32, K, kpad, alpha, crp);
    if (rng_bump != nullptr && threadIdx.x == 0) rng_bump[1] += 1;
    return;
  }
  const uint32_t k = blockIdx.x * 256 + threadIdx.x;
  if (k >= kpad) return;
  const FeatDesc fd = feats[blockIdx.y];
  if (fd.family == MSC_NIW) return;                    // (its own commit / prepare kernels)
  if (blockIdx.z != 0 && fd.family != MSC_GP && fd.family != MSC_BNB && fd.family != MSC_DD) return;
  commit_group(fd, k, kpad);                           // (every value slice commits: same numbers, and each reads back its own)
  prepare_group(fd, k, kpad, blockIdx.z, gridDim.z);
}

// ---------------------------------------------------------------------------
// One entity joins (sign > 0) or leaves (sign < 0) one group, the group given by value: group_manager::add_value /
// remove_value plus the component models' add_value / remove_value for that row (entity_state.hpp:57-68), with every
// table left current -- the additive sums, the reference's fields and the score constants of that one (feature, group),
// the group sizes and the CRP terms -- in ONE launch (the per-entity Gibbs move of hip::mixture_state was a 4-byte
// upload, accumulate, commit, and a prepare pass over all groups at the next score: three stream synchronisations).
// Block f < nfeat: feature f (its count / categorical table rows dealt out over the block); block nfeat: the counts.
// Scalar families only (niw and dm have prepare kernels of their own: the host takes the general path for them).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_entity_op(const FeatDesc *__restrict__ feats, int nfeat, uint32_t K, uint32_t kpad,
                                                    uint64_t row, uint32_t g, int sign, long long *__restrict__ cnt_acc,
                                                    uint32_t *__restrict__ cnt_u32, float alpha, float *__restrict__ crp,
                                                    int32_t *__restrict__ z_slot) {
  const long long sgn = sign;
  // What the host-side group manager of hip::mixture_state guarantees, a direct caller of msc_entity_op may not: a leave
  // needs a group with something in it (else the u32 count wraps to 4e9 and its logarithm lands in the CRP table) and,
  // when the assignment vector is at hand, a row that is in that group; a join needs an unassigned row.  The blocks run
  // side by side, so each refuses on what it can see without reading what another block writes -- this one the group
  // size and the row's slot, a feature block its own count of the group -- and reports (MSC_DEVERR_ENTITY_OP: the call
  // that notices returns MSC_EDEVICE and the state's tables are to be rebuilt).
  if ((int)blockIdx.x == nfeat) {
    if (threadIdx.x == 0) {
      const int32_t zs = z_slot != nullptr ? *z_slot : (sign > 0 ? -1 : (int32_t)g);
      const bool ok = sign > 0 ? (zs < 0 || (uint32_t)zs >= K) : (cnt_acc[g] > 0 && zs == (int32_t)g);
      if (ok) {
        cnt_acc[g] += sgn;
        cnt_u32[g] = (uint32_t)cnt_acc[g];
        if (z_slot != nullptr) *z_slot = sign > 0 ? (int32_t)g : -1;
      } else {
        report_device_error(MSC_DEVERR_ENTITY_OP, g);
      }
    }
    __syncthreads();
    crp_prepare_block(cnt_u32, K, kpad, alpha, crp);
    return;
  }
  const FeatDesc fd = feats[blockIdx.x];
  if (threadIdx.x == 0 && fd.col != nullptr && !load_masked(fd, row, true)) {
    // (a leave takes one unit out of a counter of this feature: refuse when that counter is empty)
    bool has = true;
    if (sign < 0) switch (fd.family) {
      case MSC_BBNC:
      case MSC_BB: has = fd.acc_i64[(reinterpret_cast<const uint8_t *>(fd.col)[row] != 0 ? 0 : kpad) + g] > 0; break;
      case MSC_GP:
      case MSC_BNB:
      case MSC_NICH: has = fd.acc_i64[g] > 0; break;
      case MSC_DD: {
        const int v = reinterpret_cast<const int32_t *>(fd.col)[row];
        has = !(v >= 0 && v < (int)fd.dim) || fd.acc_i64[(size_t)v * kpad + g] > 0;
      } break;
      default: break;
    }
    if (!has) report_device_error(MSC_DEVERR_ENTITY_OP, g);
    else switch (fd.family) {
      case MSC_BBNC:
      case MSC_BB:
        fd.acc_i64[(reinterpret_cast<const uint8_t *>(fd.col)[row] != 0 ? 0 : kpad) + g] += sgn;
        break;
      case MSC_GP: {
        const uint32_t v = reinterpret_cast<const uint32_t *>(fd.col)[row];
        fd.acc_i64[g] += sgn;
        fd.acc_i64[kpad + g] += sgn * (long long)v;
        fd.acc_f64[g] += (double)sign * log_factorial(v);
      } break;
      case MSC_BNB: {
        const uint32_t v = reinterpret_cast<const uint32_t *>(fd.col)[row];
        fd.acc_i64[g] += sgn;
        fd.acc_i64[kpad + g] += sgn * (long long)v;
      } break;
      case MSC_DD: {
        const int v = reinterpret_cast<const int32_t *>(fd.col)[row];
        if (v >= 0 && v < (int)fd.dim) fd.acc_i64[(size_t)v * kpad + g] += sgn;
      } break;
      case MSC_NICH: {
        const double x = reinterpret_cast<const float *>(fd.col)[row];
        fd.acc_i64[g] += sgn;
        fd.acc_f64[g] += (double)sign * x;
        fd.acc_f64[kpad + g] += (double)sign * x * x;
      } break;
      default: break;
    }
    commit_group(fd, g, kpad);
  }
  __syncthreads();                                     // (the group's fields are written; every thread prepares from them)
  prepare_group(fd, g, kpad, threadIdx.x, 256);
}

// ---------------------------------------------------------------------------
// leave-one-out pre-pass: own[n] = (prior of z[n] with the row removed, if crp) +
// sum over scalar features of score_value(group z[n] minus row n, row n).  One thread per row,
// everything in double; niw features add theirs inside the niw kernel.
// ---------------------------------------------------------------------------
// one feature of one row against the row's own group, read from global memory: masked columns, counts beyond the
// tables, nich, dm -- and every feature of the kernel that stages nothing.
// HEAVY = false leaves out the branches with lgamma chains in them (gp / bnb counts beyond the table, dm): with them
// inlined the kernel sits at 225 VGPRs, 2 waves per SIMD, for rows that never take them (launch_loo_own picks)
template <bool HEAVY>
MSC_DEV double loo_feature_global(const FeatDesc &fd, uint64_t row, int g, uint32_t kpad) {
  if (fd.family != MSC_NIW && load_masked(fd, row, true)) return 0.0;
  switch (fd.family) {
    case MSC_BB:       // lookup families: the table k_prepare made of "this value against the group minus one of it"
      return (double)fd.loo_tab[(reinterpret_cast<const uint8_t *>(fd.col)[row] != 0 ? kpad : 0u) + g];
    case MSC_BBNC:     // p does not move when a row leaves: the plain score's own table entry (a double log per row before)
      return (double)fd.tab[(reinterpret_cast<const uint8_t *>(fd.col)[row] != 0 ? kpad : 0u) + g];
    case MSC_GP: {
      const uint32_t v = reinterpret_cast<const uint32_t *>(fd.col)[row];
      if (v < fd.vcap) return (double)fd.loo_tab[(size_t)v * kpad + g];
      if (HEAVY) return gp_loo(fd.hp, fd.raw_u32[g], fd.raw_u32[kpad + g], v);
      return 0.0;
    }
    case MSC_BNB: {
      const uint32_t v = reinterpret_cast<const uint32_t *>(fd.col)[row];
      if (v < fd.vcap) return (double)fd.loo_tab[(size_t)v * kpad + g];
      if (HEAVY) return bnb_score(fd.hp, (double)fd.raw_u32[g] - 1.0, (double)fd.raw_u32[kpad + g] - (double)v, (double)v);
      return 0.0;
    }
    case MSC_DM: {
      // dim + 1 lookups in the leave-one-out tables k_dm_prepare fills (it was 2 (dim + 1) lgamma per row); rows
      // whose total is beyond the tables take the formula
      const int32_t *x = reinterpret_cast<const int32_t *>(fd.col) + row * fd.dim;
      const uint32_t tot = fd.dm_tot[row];
      double s = 0.0;
      if (tot < kGpMaxTable && fd.loo64 != nullptr && fd.dm_meta != nullptr) {
        for (uint32_t i = 0; i <= fd.dim; i++) {
          const uint32_t v = i < fd.dim ? (uint32_t)x[i] : tot;
          if (v) s += fd.loo64[(size_t)(fd.dm_meta[2 * i] / 2 + v) * kpad + g];      // (entry 0 is exactly zero)
        }
      } else if (HEAVY) {
        s = dm_score_direct(fd.hp, fd.dim, fd.raw_u32 + g, kpad, x, true);
      }
      return s;
    }
    case MSC_DD: {
      int v = reinterpret_cast<const int32_t *>(fd.col)[row];
      v = v < 0 ? 0 : (v >= (int)fd.dim ? (int)fd.dim - 1 : v);
      return (double)fd.loo_tab[(size_t)v * kpad + g];
    }
    case MSC_NICH:
      // (downdate and posterior in double, the two logarithms and the division in float like every other entry of
      // the row: family_math.hpp nich_loo_tab_sweep -- the all-double form was ~225 double instructions per feature,
      // 0.18 of this kernel's 0.32 ms on C3)
      return (double)nich_loo_tab_sweep(fd.hp, fd.loo64 + (size_t)g * kNlooStride, 1, reinterpret_cast<const float *>(fd.col)[row]);
    default: return 0.0;
  }
}

template <bool HEAVY>
__global__ __launch_bounds__(256) void k_loo_own(const FeatDesc *__restrict__ feats, int nfeat, uint32_t K,
                                                  uint32_t kpad, uint64_t row0, uint64_t nrows,
                                                  const int32_t *__restrict__ z,
                                                  const float *__restrict__ crp, float *__restrict__ own) {
  // (one thread per row.  Sharing a row among 2-16 lanes, a feature each, was tried for small inputs: the
  // descriptor loads stop being scalar, 10k rows x 12 features go from 13 to 9 us and everything larger gets
  // slower -- 100k rows 20 -> 29 us, C3 0.49 -> 0.66 ms even at one lane per row)
  const uint64_t n = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= nrows) return;
  const int g = z[n];
  if (g < 0 || (uint32_t)g >= K) {                          // (an id outside [0, ngroups) reads as not assigned)
    own[n] = 0.f;
    return;
  }
  const uint64_t row = row0 + n;
  double s = 0.0;
  if (crp) {
    const float lm1 = crp[kpad + g];
    s = __builtin_isinf(lm1) ? (double)crp[2 * (size_t)kpad + 1] + (double)crp[2 * (size_t)kpad + 3]
                             : (double)lm1 + (double)crp[crp_lo_cntm1(kpad) + g];
  }
  // The kernel walks the tile plan's copy of the descriptors (abi.cpp plan_groups): runs of unmasked lookup features
  // first.  For those a feature is value -> table entry, two dependent loads and nothing else, and with one row per
  // thread nobody hides them: four features at a time, without a branch between the loads (the value fetch and the
  // clamp of the tile kernels' lookup runs), so that four loads are in flight where one was.  Everything else --
  // masked columns, counts beyond the tables, nich, dm -- goes through loo_feature_global, feature by feature.
  auto run_word = [&](const FeatDesc &d) -> uint32_t {
    typedef const __attribute__((address_space(1))) unsigned char *g_u8;
    const bool u8 = d.kind == MSC_KIND_LOOKUP_U8;
    const uint64_t at = (reinterpret_cast<uint64_t>(d.col) + (u8 ? row : row * 4)) & ~(uint64_t)3;
    return *reinterpret_cast<const __attribute__((address_space(1))) uint32_t *>((g_u8)at);
  };
  auto run_entry = [&](const FeatDesc &d, uint32_t word) -> float {
    const bool u8 = d.kind == MSC_KIND_LOOKUP_U8;
    const uint32_t sh = u8 ? (uint32_t)((reinterpret_cast<uint64_t>(d.col) + row) & 3u) * 8u : 0u;
    const int v = (int)(u8 ? (word >> sh) & 0xffu : word);
    const uint32_t idx = (uint32_t)(v < 0 ? 0 : (v > (int)d.run_clamp ? (int)d.run_clamp : v));
    const float *src = d.family == MSC_BBNC ? d.tab : d.loo_tab;      // (bbnc: p does not move when a row leaves)
    return src[(size_t)idx * kpad + g];
  };
  int f = 0;
  while (f < nfeat) {
    if (feats[f].kind != MSC_KIND_GENERIC) {
      const int fe = (int)feats[f].run_end;
      for (; f + 4 <= fe; f += 4) {
        uint32_t w[4];
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; j++) w[j] = run_word(feats[f + j]);
#pragma unroll
        for (int j = 0; j < 4; j++) e[j] = run_entry(feats[f + j], w[j]);
#pragma unroll
        for (int j = 0; j < 4; j++) s += (double)e[j];
      }
      for (; f < fe; f++) s += (double)run_entry(feats[f], run_word(feats[f]));
      continue;
    }
    const FeatDesc fd = feats[f++];
    s += loo_feature_global<HEAVY>(fd, row, g, kpad);
  }
  own[n] = (float)s;
}

// ---------------------------------------------------------------------------
// The same pass for many rows: what a row gathers per feature comes from LDS.  In k_loo_own every lane of a wave reads
// its own group's entry -- 64 cache lines per wave instruction, two of them per nich feature and row (96 bytes of
// constants) -- and the kernel ran at the rate the address unit splits such gathers, 0.25 ms for C3's 64 columns
// (0.97 TB/s of 64-byte sectors for 4 + 4 useful bytes).  Here a workgroup of 1024 threads takes 4096 rows (four per
// thread, coalesced) through the stages of the leave-one-out plan (abi.cpp plan_groups: consecutive features whose
// blocks -- the lookup families' leave-one-out tables, nich's six doubles per group, for all kpad groups -- share the
// 128 KiB slot): the block copy is coalesced, the per-row reads are LDS reads at the lane's own address.  (Two workgroups
// of 512 threads with 64 KiB each, covering each other's copies: 149 us on C3 against 140 -- twice the stages.)  Same terms, same order of the double sum as k_loo_own: same bits.
// ---------------------------------------------------------------------------
constexpr int kLooRows = 4, kLooThreads = 1024;      // (kLooStageFeats: msc_internal.hpp)
template <bool HEAVY>
__global__ __launch_bounds__(kLooThreads) void k_loo_own_lds(const FeatDesc *__restrict__ feats,
                                                              int nfeat, uint32_t K, uint32_t kpad, uint64_t row0, uint64_t nrows,
                                                              const int32_t *__restrict__ z,
                                                              const float *__restrict__ crp, float *__restrict__ own) {
  extern __shared__ __attribute__((aligned(16))) float slot[];
  const uint64_t base = (uint64_t)blockIdx.x * (kLooRows * kLooThreads) + threadIdx.x;
  int g[kLooRows];
  double s[kLooRows];
  uint64_t row[kLooRows];
#pragma unroll
  for (int j = 0; j < kLooRows; j++) {
    const uint64_t n = base + (uint64_t)j * kLooThreads;
    const int gg = n < nrows ? z[n] : -1;
    g[j] = (uint32_t)gg < K ? gg : -1;
    row[j] = row0 + (n < nrows ? n : nrows - 1);          // (a row of the call's range for the loads of idle slots)
    s[j] = 0.0;
    if (crp && g[j] >= 0) {
      const float lm1 = crp[kpad + g[j]];
      s[j] = __builtin_isinf(lm1) ? (double)crp[2 * (size_t)kpad + 1] + (double)crp[2 * (size_t)kpad + 3]
                                  : (double)lm1 + (double)crp[crp_lo_cntm1(kpad) + g[j]];
    }
  }
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // The values of this thread's rows for every feature of a stage are fetched in one breath: one exposed memory latency per
  // stage instead of one per feature.  ONE dword load per value whatever the column's type (round 5) -- a byte column: the
  // aligned dword around the byte, the byte picked afterwards, as the tile kernels' lookup runs do; written as
  // `u8 ? byte load : dword load` the compiler made a branch per feature with a full wait behind each.  No branch around a
  // load: a slot past the stage's end reads the stage's last column again.
  // (Fetching them a stage AHEAD -- asked for right after this stage's block copies, waited for a stage later -- does not
  // pay: round 3 measured 148 -> 162 us with four rows a thread (126 registers); round 5, with the single-load form, 139 ->
  // 241 us at four rows (69 spilled registers) and 171 -> 174 us at two rows a thread, where two rows without it take 171:
  // the value latency is not what a stage waits for.  profiles/r05_loo_accumulate.txt)
  int f0 = 0;
  while (f0 < nfeat) {
    const int f1 = (int)feats[f0].loo_stage_end;           // (at most kLooStageFeats features: abi.cpp plan_groups)
    uint32_t w[kLooStageFeats][kLooRows];
    __syncthreads();                                      // the slot's previous readers are done
    // the blocks, 1 KiB per wave instruction straight into LDS (global_load_lds_dwordx4) ...
    for (int f = f0; f < f1; f++) {
      const FeatDesc &fd = feats[f];
      if (fd.loo_rows == 0) continue;
      const float *src = fd.family == MSC_NICH ? reinterpret_cast<const float *>(fd.loo64)
                                                : (fd.family == MSC_BBNC ? fd.tab : fd.loo_tab);
      const uint32_t nchunks = fd.loo_rows * (kpad / 256);  // (kpad is a multiple of 256)
      float4 *dst = reinterpret_cast<float4 *>(slot + fd.loo_off);
      for (uint32_t c = (uint32_t)wave; c < nchunks; c += kLooThreads / 64)
        glds16(src + (size_t)c * 256 + 4 * lane, dst + (size_t)c * 64);
    }
    // ... and, in the same breath, this stage's values
    uint32_t sh8[kLooStageFeats][kLooRows];                // byte columns: the byte's shift inside its dword, | 0x100
#pragma unroll
    for (int fi = 0; fi < kLooStageFeats; fi++) {
      const FeatDesc &fd = feats[f0 + fi < f1 ? f0 + fi : f1 - 1];
      const bool u8 = fd.col_type == MSC_TYPE_B || fd.col_type == MSC_TYPE_I8 || fd.col_type == MSC_TYPE_U8;
      const uint64_t cbase = reinterpret_cast<uint64_t>(fd.col);
#pragma unroll
      for (int j = 0; j < kLooRows; j++) {
        const uint64_t at = cbase + (u8 ? row[j] : row[j] * 4);
        w[fi][j] = *reinterpret_cast<const __attribute__((address_space(1))) uint32_t *>(
            (const __attribute__((address_space(1))) unsigned char *)(at & ~(uint64_t)3));
        sh8[fi][j] = u8 ? (uint32_t)(at & 3u) * 8u + 0x100u : 0u;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // my share of the blocks, and the values
#pragma unroll
    for (int fi = 0; fi < kLooStageFeats; fi++)
#pragma unroll
      for (int j = 0; j < kLooRows; j++)
        if (sh8[fi][j] & 0x100u) w[fi][j] = (w[fi][j] >> (sh8[fi][j] & 31u)) & 0xffu;
    __syncthreads();
#pragma unroll
    for (int fi = 0; fi < kLooStageFeats; fi++) {
      if (f0 + fi >= f1) break;
      const FeatDesc &fd = feats[f0 + fi];
      if (fd.loo_rows == 0) {                             // not staged: from global memory
#pragma unroll
        for (int j = 0; j < kLooRows; j++)
          if (g[j] >= 0) s[j] += loo_feature_global<HEAVY>(fd, row[j], g[j], kpad);
        continue;
      }
      const float *blk = slot + fd.loo_off;
      if (fd.family == MSC_NICH) {
#pragma unroll
        for (int j = 0; j < kLooRows; j++) {
          const double *t = reinterpret_cast<const double *>(blk) + (size_t)(g[j] >= 0 ? g[j] : 0) * kNlooStride;
          const float e = nich_loo_tab_sweep(fd.hp, t, 1, __uint_as_float(w[fi][j]));
          if (g[j] >= 0) s[j] += (double)e;
        }
      } else {
#pragma unroll
        for (int j = 0; j < kLooRows; j++) {
          const int v = (int)w[fi][j];
          const uint32_t idx = (uint32_t)(v < 0 ? 0 : (v > (int)fd.run_clamp ? (int)fd.run_clamp : v));
          const float e = blk[(size_t)idx * kpad + (g[j] >= 0 ? g[j] : 0)];
          if (g[j] >= 0) s[j] += (double)e;
        }
      }
    }
    f0 = f1;
  }
#pragma unroll
  for (int j = 0; j < kLooRows; j++) {
    const uint64_t n = base + (uint64_t)j * kLooThreads;
    if (n < nrows) own[n] = g[j] >= 0 ? (float)s[j] : 0.f;
  }
}

// ---------------------------------------------------------------------------
// single NICH feature (config C2 / C5 scoring pass)
//   A wave owns one 256-group tile (constants in VGPRs) and a "slot": it visits blocks of Q consecutive
//   rows that lie nslots*Q rows apart.  Waves are numbered tile-fastest, so the waves resident at one
//   moment write dense, contiguous windows of the score matrix (one per visit) that sweep forward
//   through it.  How many visits a wave makes is a launch parameter the host settles per context
//   (launch_score_t, abi.cpp run_score); everything the wave's rows need is fetched once, up front.
// ---------------------------------------------------------------------------
template <bool LOO, bool CRP, int Q, bool NT>
__global__ __launch_bounds__(256) void k_score_nich1(const FeatDesc *__restrict__ feats,
                                                      uint32_t K, uint32_t kpad, uint64_t row0,
                                                      uint64_t nrows, uint64_t nslots,
                                                      const int32_t *__restrict__ z,
                                                      const float *__restrict__ own,
                                                      const float *__restrict__ crp,
                                                      float *__restrict__ out, uint64_t ld) {
  const FeatDesc fd = feats[0];
  const int lane = threadIdx.x & 63;
  const uint32_t ktiles = kpad / kGroupTile;
  const uint64_t wave_id = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint64_t slot = wave_id / ktiles;
  if (slot >= nslots) return;
  const uint32_t kt = (uint32_t)(wave_id - slot * ktiles);
  const uint32_t kb = kt * kGroupTile + lane * 4;
  const float *tab = fd.tab + kb;
  const float4 mh = ld4(tab + (size_t)NICH_MU_HI * kpad), ml = ld4(tab + (size_t)NICH_MU_LO * kpad),
               c0 = ld4(tab + (size_t)NICH_C0 * kpad), c1l = ld4(tab + (size_t)NICH_C1LN2 * kpad),
               c1 = ld4(tab + (size_t)NICH_C1 * kpad), c2 = ld4(tab + (size_t)NICH_C2 * kpad);
  float4 logcnt = make_float4(0, 0, 0, 0), logcnt_lo = make_float4(0, 0, 0, 0);
  float le0 = 0, le1 = 0, le0_lo = 0, le1_lo = 0;
  if (CRP) {
    logcnt = ld4(crp + kb);
    logcnt_lo = ld4(crp + crp_lo_cnt(kpad) + kb);
    le0 = crp[2 * (size_t)kpad];
    le1 = crp[2 * (size_t)kpad + 1];
    le0_lo = crp[2 * (size_t)kpad + 2];
    le1_lo = crp[2 * (size_t)kpad + 3];
  }
  const bool vec_ok = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  const bool tile_full = (kt + 1) * kGroupTile <= K;
  const float *xcol = reinterpret_cast<const float *>(fd.col) + row0;
  // everything the wave's rows need, fetched at once: lane i holds row r = i % Q of visit k = i / Q
  // (a spread launch order -- the resident waves writing S windows far apart instead of one dense one -- and the stores'
  // cache policies were measured in round 4 and are kept as a patch: tools/microbench/r04_store_policy_experiment.patch,
  // profiles/r04_spread_store.txt, r04_store_policy.txt)
  const uint64_t nblocks = (nrows + Q - 1) / Q;
  const uint32_t nvis = (uint32_t)((nblocks + nslots - 1) / nslots);      // <= 64 / Q (launcher)
  const uint32_t vk = (uint32_t)lane / Q, vr = (uint32_t)lane % Q;
  const uint64_t myblock = slot + (uint64_t)vk * nslots;
  const uint64_t myrow = myblock * Q + vr;
  const bool mine = vk < nvis && myblock < nblocks && myrow < nrows;
  const float xv = mine ? xcol[myrow] : 0.0f;
  const unsigned long long mbits_all =
      __builtin_amdgcn_ballot_w64(fd.mask != nullptr && mine && fd.mask[row0 + myrow] != 0);
  int gz = -1;
  float sloo = 0, erow = le0, erow_lo = le0_lo;
  if (LOO && mine) {
    gz = z[myrow];
    if ((uint32_t)gz >= K) gz = -1;                         // (an id outside the table: not assigned)
    sloo = own[myrow];
    if (CRP && gz >= 0 && __builtin_isinf(crp[kpad + gz])) {
      erow = le1;
      erow_lo = le1_lo;
    }
  }
  for (uint32_t k = 0; k < nvis; k++) {
    const uint64_t L = slot + (uint64_t)k * nslots;
    if (L >= nblocks) break;
    const uint64_t rb = L * Q;
    if (rb >= nrows) continue;
    const int nr = (int)((nrows - rb) < (uint64_t)Q ? (nrows - rb) : (uint64_t)Q);
    const int l0 = (int)(k * Q);                              // first lane of this visit's values
    const unsigned long long mbits = (mbits_all >> l0) & ((1ull << Q) - 1ull);
    if (nr == Q && mbits == 0ull && vec_ok && tile_full) {     // straight-line: no per-row branches
#pragma unroll
      for (int r = 0; r < Q; r++) {
        const float x = lane_bcast(xv, l0 + r);
        float4 s;
        s.x = nich_eval(x, mh.x, ml.x, c0.x, c1l.x, c1.x, c2.x);
        s.y = nich_eval(x, mh.y, ml.y, c0.y, c1l.y, c1.y, c2.y);
        s.z = nich_eval(x, mh.z, ml.z, c0.z, c1l.z, c1.z, c2.z);
        s.w = nich_eval(x, mh.w, ml.w, c0.w, c1l.w, c1.w, c2.w);
        if (CRP) {                                             // lo first: the last add then rounds the result once
          add4(s, crp_prior4_lo(logcnt, logcnt_lo, LOO ? lane_bcast(erow_lo, l0 + r) : le0_lo));
          add4(s, crp_prior4(logcnt, LOO ? lane_bcast(erow, l0 + r) : le0));
        }
        if (LOO) {
          const int g = lane_bcast(gz, l0 + r);
          if (g >= 0) replace_own(s, kb, g, lane_bcast(sloo, l0 + r));
        }
        const f32x4 v = {s.x, s.y, s.z, s.w};
        f32x4 *p = reinterpret_cast<f32x4 *>(out + (rb + r) * ld + kb);
        if (NT) __builtin_nontemporal_store(v, p);
        else *p = v;
      }
    } else {
      for (int r = 0; r < nr; r++) {
        const float x = lane_bcast(xv, l0 + r);
        float4 s = make_float4(0, 0, 0, 0);
        if (!((mbits >> r) & 1ull)) {
          s.x = nich_eval(x, mh.x, ml.x, c0.x, c1l.x, c1.x, c2.x);
          s.y = nich_eval(x, mh.y, ml.y, c0.y, c1l.y, c1.y, c2.y);
          s.z = nich_eval(x, mh.z, ml.z, c0.z, c1l.z, c1.z, c2.z);
          s.w = nich_eval(x, mh.w, ml.w, c0.w, c1l.w, c1.w, c2.w);
        }
        if (CRP) {                                             // lo first: the last add then rounds the result once
          add4(s, crp_prior4_lo(logcnt, logcnt_lo, LOO ? lane_bcast(erow_lo, l0 + r) : le0_lo));
          add4(s, crp_prior4(logcnt, LOO ? lane_bcast(erow, l0 + r) : le0));
        }
        if (LOO) {
          const int g = lane_bcast(gz, l0 + r);
          if (g >= 0) replace_own(s, kb, g, lane_bcast(sloo, l0 + r));
        }
        store_row<NT>(out, ld, rb + r, kb, K, s, vec_ok);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// general path: any feature list, workgroup tiles with LDS-staged tables.
//   grid.x = row chunks of 8*R rows (grid-stride), grid.y = k-tiles, block = 8 waves
// ---------------------------------------------------------------------------
template <int R, int W, bool LOO, bool CRP, bool DM>
__global__ __launch_bounds__(W * 64, DM ? 2 : W / 4) void k_score_tile(const FeatDesc *__restrict__ feats,
                                                                 int nfeat, int nsplit, uint32_t K, uint32_t kpad,
                                                                 uint64_t row0, uint64_t nrows,
                                                                 const int32_t *__restrict__ z,
                                                                 const float *__restrict__ own,
                                                                 const float *__restrict__ crp,
                                                                 float *__restrict__ out, uint64_t ld) {
  __shared__ float4 lds[kGrpRows * 64 + (LOO ? W * 64 : 0)];       // the table slot (+ a KiB per wave for the LOO epilogue)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t kb = blockIdx.y * kGroupTile + lane * 4;
  const bool vec_ok = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  float4 logcnt = make_float4(0, 0, 0, 0);
  float le0 = 0, le1 = 0;
  if (CRP) {
    logcnt = ld4(crp + kb);
    le0 = crp[2 * (size_t)kpad];
    le1 = crp[2 * (size_t)kpad + 1];
  }
  const uint64_t rows_per_wg = (uint64_t)W * R;
  const uint64_t nchunks = (nrows + rows_per_wg - 1) / rows_per_wg;
  for (uint64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const uint64_t rb = chunk * rows_per_wg + (uint64_t)wave * R;       // relative to row0
    const int nr = rb >= nrows ? 0 : (int)((nrows - rb) < (uint64_t)R ? (nrows - rb) : (uint64_t)R);
    float4 acc[R];
    // The prior is a (hi, lo) pair per group: the accumulators start from lo (~1e-7) and hi (log count, ~10 for a
    // large group) is added after the last feature -- the likelihood terms largely cancel it, and a sum that starts at
    // 10 rounds every one of its steps relative to 10 instead of to the result (1.4-1.6e-6 of it measured).
    int single = 0;                                       // lane r: removing row r empties its group (kept in a VGPR:
    if (LOO && CRP && lane < nr) {                        //  the tile scorer leaves no SGPR pair free across its call)
      const int g0 = z[rb + lane];
      single = g0 >= 0 && (uint32_t)g0 < K && __builtin_isinf(crp[kpad + g0]) ? 1 : 0;
    }
    const bool tail_only = nsplit == 0;                  // (score_tile, SPLIT: the prior's lo part is added afterwards then)
    if (CRP && !tail_only) {
      const float4 lo = ld4(crp + crp_lo_cnt(kpad) + kb);
      const float e0 = crp[2 * (size_t)kpad + 2], e1 = crp[2 * (size_t)kpad + 3];
#pragma unroll
      for (int r = 0; r < R; r++) acc[r] = crp_prior4_lo(logcnt, lo, LOO && lane_bcast(single, r) ? e1 : e0);
    } else {
#pragma unroll
      for (int r = 0; r < R; r++) acc[r] = make_float4(0, 0, 0, 0);
    }
    score_tile<R, W, DM, true>(feats, nfeat, nsplit, kpad, blockIdx.y, lane, row0 + rb, nr, row0, lds, acc);
    if (CRP && tail_only) {
      const float4 lo = ld4(crp + crp_lo_cnt(kpad) + kb);
      const float e0 = crp[2 * (size_t)kpad + 2], e1 = crp[2 * (size_t)kpad + 3];
#pragma unroll
      for (int r = 0; r < R; r++) add4(acc[r], crp_prior4_lo(logcnt, lo, LOO && lane_bcast(single, r) ? e1 : e0));
    }
    // epilogue, row by row (one pass, so nothing of one row outlives its store): + hi of the prior; then the own
    // group's entry becomes the row's pre-computed leave-one-out value (k_loo_own) through a KiB of LDS that belongs
    // to the wave (beyond the table slot, so no barrier) -- park the row, one lane overwrites the entry, read the row
    // back.  Ways that cost more: merging the value into the owning lane's float4 in registers (replace_own per row;
    // the tiling has no SGPRs / VGPRs left: 60-100 bytes of scratch per lane, C3 +13 %); a 4-byte store after the
    // row stores (needs a fence and cached stores, +45 %); a pass of its own over the finished matrix (a million
    // random read-modify-writes in HBM, +12 %); all rows parked in the table slot (one more barrier per chunk, +8 %).
    float4 *mine = lds + (size_t)kGrpRows * 64 + (size_t)wave * 64;
    int gz = -1;
    float sloo = 0.f;
    float4 hi = make_float4(0, 0, 0, 0);
    if (CRP) {                                            // fetched again (an L2 hit per chunk) rather than held across
      const float *again = crp;                           // the tile scorer: 4 VGPRs and 4 SGPR pairs of isinf masks
      asm volatile("" : "+s"(again));
      hi = ld4(again + kb);
    }
    if (LOO && lane < nr) {
      gz = z[rb + lane];
      sloo = own[rb + lane];
    }
    if (LOO && gz >= 0 && ((uint32_t)gz >= K || (uint32_t)gz / kGroupTile != blockIdx.y)) gz = -1;   // not in this k-tile
#pragma unroll
    for (int r = 0; r < R; r++) {
      if (CRP) add4(acc[r], crp_prior4(hi, LOO && lane_bcast(single, r) ? le1 : le0));
      if (LOO) {
        const int g = lane_bcast(gz, r);
        if (g >= 0) {                                     // (wave-uniform)
          mine[lane] = acc[r];
          if (lane == 0) reinterpret_cast<float *>(mine)[(uint32_t)g % kGroupTile] = lane_bcast(sloo, r);
          __builtin_amdgcn_wave_barrier();
          acc[r] = mine[lane];
        }
      }
      if (r < nr) store_row(out, ld, rb + r, kb, K, acc[r], vec_ok);
    }
  }
}

// ---------------------------------------------------------------------------
// k_score_tile_roles: the general path when a state has BOTH staged lookup features and unmasked nich features and
// enough rows to fill the chip.  In k_score_tile every wave runs the lookup phase (bound by the CU's LDS port: a
// wave's ds_read_b128 is 1 KiB = 8 cycles) and then the nich phase (bound by the vector ALUs), and a chunk costs the
// SUM of the two (C3: ~60 k + ~95 k cycles of 193 k).  Here the workgroup's sixteen waves split the phases between
// them: waves 0-7 take the lookups of all 128 rows (16 rows each), waves 8-15 the nich features of the same rows --
// their constants straight from the tables in L2, no LDS, so nothing ties them to the staging of the lookup groups
// (the lookup waves synchronise among themselves: WaveSubsetBarrier).
// The hand-over (round 4: REVERSED).  With the nich features in blocks (family_math.hpp) the nich half is the shorter one
// (1.0 ms of C3's pass alone, the lookup half 1.4), so what follows the two sums -- adding them, the prior's high half,
// the leave-one-out entry, the 16 KiB of stores -- moved to the nich waves: at the end of a chunk lookup wave p parks its
// sums (prior lo + lookups) in the then idle table slot and goes on to stage the next chunk's first feature group, wave
// 8 + p adds its own to them -- (prior + lookups) + (nich features): the sum score_tile<SPLIT> forms, so a row gets the
// same bits from either kernel --, finishes the rows and stores them.  Two workgroup barriers a chunk, as before: the sums
// are in the slot / every nich wave has read them.  (Before: the nich waves parked THEIR sums and the lookup waves
// finished; with leave-one-out + prior that cost the lookup waves 0.5 ms of C3's 2.07 -- ~45 spilled reloads in their
// epilogue, harmless while the nich half was the longer one.)
// ---------------------------------------------------------------------------
constexpr int kRoleRows = 16;          // sums (float4) per wave: 16 rows, or -- PAIR -- 32
// PAIR: at most 128 groups (score_block.hpp pair_dup): a lane carries two groups, a float4 of sums two rows, a wave 32 rows,
// a workgroup 256
template <bool LOO, bool CRP, bool PAIR = false>
__global__ __launch_bounds__(1024, 4) void k_score_tile_roles(const FeatDesc *__restrict__ feats, int nfeat, int nsplit,
                                                               uint32_t K, uint32_t kpad, uint64_t row0, uint64_t nrows,
                                                               const int32_t *__restrict__ z, const float *__restrict__ own,
                                                               const float *__restrict__ crp, float *__restrict__ out,
                                                               uint64_t ld, uint32_t kt0) {
  constexpr int R = kRoleRows, RW = PAIR ? 2 * kRoleRows : kRoleRows;   // sums / rows per wave
  __shared__ float4 lds[kGrpRows * 64];                   // the table slot; between chunks the hand-over
  __shared__ uint32_t lookers_arrived;                    // the lookup waves' own barrier (WaveSubsetBarrier)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool looker = wave < 8;                           // waves 0-7: lookups; 8-15: nich, then the rows' finish
  if (threadIdx.x == 0) lookers_arrived = 0u;
  __syncthreads();
  WaveSubsetBarrier<8> lbar{&lookers_arrived, 0u};
  const int pair = wave & 7;
  // kt: the k-tile.  PAIR mode takes the FIRST 128 groups of tile kt0 (round 5: kt0 = 1 scores a last tile of 65-128 groups
  // beyond a full one -- 321 <= K <= 384 -- into a matrix of its own, `out` column j = group 256 kt0 + j: the fused sweeps'
  // tail); Kt: the groups of that tile
  const uint32_t kt = blockIdx.y + kt0;
  const uint32_t kb = kt * kGroupTile + (PAIR ? (uint32_t)lane * 2u : (uint32_t)lane * 4u);
  const uint32_t Kt = PAIR ? K - kt * kGroupTile : K;
  const bool vec_ok = PAIR ? ((ld & 1) == 0) && ((reinterpret_cast<uintptr_t>(out) & 7) == 0)
                           : ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  const uint64_t rows_per_wg = 8 * RW;
  const uint64_t nchunks = (nrows + rows_per_wg - 1) / rows_per_wg;
  float4 *const handover = lds + (size_t)pair * R * 64 + lane;           // the pair's 16 rows of the slot
  if (looker) {
    // ---- the lookup waves: prior lo + lookups, parked in the slot ----
    float4 logcnt = make_float4(0, 0, 0, 0);
    if (CRP) logcnt = PAIR ? pair_dup(ld2(crp + kb)) : ld4(crp + kb);
    for (uint64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
      const uint64_t rb = chunk * rows_per_wg + (uint64_t)pair * RW;      // relative to row0
      const int nr = rb >= nrows ? 0 : (int)((nrows - rb) < (uint64_t)RW ? (nrows - rb) : (uint64_t)RW);
      float4 acc[R];
      int single = 0;                                     // lane r: removing row r empties its group
      if (LOO && CRP && lane < nr) {
        const int g0 = z[rb + lane];
        single = g0 >= 0 && (uint32_t)g0 < K && __builtin_isinf(crp[kpad + g0]) ? 1 : 0;
      }
      if (CRP) {
        const float4 lo = PAIR ? pair_dup(ld2(crp + crp_lo_cnt(kpad) + kb)) : ld4(crp + crp_lo_cnt(kpad) + kb);
        const float e0 = crp[2 * (size_t)kpad + 2], e1 = crp[2 * (size_t)kpad + 3];
#pragma unroll
        for (int r = 0; r < R; r++) {
          if constexpr (PAIR)
            acc[r] = crp_prior_pair_lo(make_float2(logcnt.x, logcnt.y), make_float2(lo.x, lo.y), LOO && lane_bcast(single, 2 * r) ? e1 : e0,
                                       LOO && lane_bcast(single, 2 * r + 1) ? e1 : e0);
          else acc[r] = crp_prior4_lo(logcnt, lo, LOO && lane_bcast(single, r) ? e1 : e0);
        }
      } else {
#pragma unroll
        for (int r = 0; r < R; r++) acc[r] = make_float4(0, 0, 0, 0);
      }
      score_tile_groups<R, 8, false, false, PAIR>(feats, nsplit, kpad, kt, lane, row0 + rb, nr, row0, lds, acc, lbar);
      lbar();                                             // every lookup wave is done reading the slot's tables
#pragma unroll
      for (int r = 0; r < R; r++) handover[r * 64] = acc[r];
      __syncthreads();                                    // (1) the lookup sums are in the slot
      __syncthreads();                                    // (2) the nich waves have read them: the slot may be staged again
    }
    return;
  }
  // ---- the nich waves: the second phase from the pack, no LDS and no barrier until the hand-over; block by block
  // (score_block.hpp nich_phase_packed: the steps every tile kernel takes); then the rows' finish ----
  // Little of the finish is alive during the phase (round 5): the prior's high half and what depends on it are fetched
  // AFTER it, through a pointer the compiler cannot see through.  Fetched up front they were held across the block loops
  // in registers the loops did not have.
  for (uint64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const uint64_t rb = chunk * rows_per_wg + (uint64_t)pair * RW;        // relative to row0
    const int nr = rb >= nrows ? 0 : (int)((nrows - rb) < (uint64_t)RW ? (nrows - rb) : (uint64_t)RW);
    // (the rows' groups and leave-one-out values stream in from HBM: asked for before the phase, two registers across it;
    // what comes out of the small, cache-resident prior table is fetched after it)
    int gz = -1, single = 0;
    float sloo = 0.f;
    if (LOO && lane < nr) {
      gz = z[rb + lane];
      sloo = own[rb + lane];
    }
    float4 acc[R];
    const uint64_t myrow = row0 + (lane < nr ? rb + lane : (nr ? rb : 0));   // (a row of the call's range for idle lanes)
    nich_phase_packed<R, false, PAIR>(feats, nsplit, kpad, kb, row0 + rb, nr, myrow, acc);
    const float *crpq = crp;
    asm volatile("" : "+s"(crpq));
    float4 hi = make_float4(0, 0, 0, 0);
    float le0 = 0, le1 = 0;
    if (CRP) {
      hi = PAIR ? pair_dup(ld2(crpq + kb)) : ld4(crpq + kb);
      le0 = crpq[2 * (size_t)kpad];
      le1 = crpq[2 * (size_t)kpad + 1];
    }
    if (LOO && CRP && lane < nr) single = gz >= 0 && (uint32_t)gz < K && __builtin_isinf(crpq[kpad + gz]) ? 1 : 0;
    if (LOO && gz >= 0 && ((uint32_t)gz >= K || (uint32_t)gz / kGroupTile != kt)) gz = -1;   // not in this k-tile
    if (PAIR && gz >= 0) gz -= (int)(kt * kGroupTile);     // (PAIR: the lane that holds a group goes by its index in the tile)
    __syncthreads();                                      // (1) the lookup sums are in the slot
#pragma unroll
    for (int r = 0; r < R; r++) {                         // (prior lo + lookups) + (nich features)
      float4 t = handover[r * 64];
      add4(t, acc[r]);
      acc[r] = t;
      if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // (four reads in flight: sixteen beside the sixteen sums spill)
    }
    __syncthreads();                                      // (2) read: the lookup waves go on
    // the rows' finish: + hi of the prior, then the own group's entry becomes the row's leave-one-out value (the lane and
    // component that hold group g take it: score_block.hpp replace_own)
#pragma unroll
    for (int r = 0; r < R; r++) {
      if constexpr (PAIR) {
        if (CRP)
          add4(acc[r], crp_prior_pair(make_float2(hi.x, hi.y), LOO && lane_bcast(single, 2 * r) ? le1 : le0,
                                      LOO && lane_bcast(single, 2 * r + 1) ? le1 : le0));
        if (LOO)
          replace_own_pair(acc[r], lane, lane_bcast(gz, 2 * r), lane_bcast(sloo, 2 * r), lane_bcast(gz, 2 * r + 1), lane_bcast(sloo, 2 * r + 1));
        if (2 * r < nr) store_half_row(out, ld, rb + 2 * r, lane, Kt, acc[r].x, acc[r].y, vec_ok);
        if (2 * r + 1 < nr) store_half_row(out, ld, rb + 2 * r + 1, lane, Kt, acc[r].z, acc[r].w, vec_ok);
      } else {
        if (CRP) add4(acc[r], crp_prior4(hi, LOO && lane_bcast(single, r) ? le1 : le0));
        if (LOO) {
          const int g = lane_bcast(gz, r);
          if (g >= 0) replace_own(acc[r], kb, g, lane_bcast(sloo, r));   // (wave-uniform g; in registers: no LDS left to these waves)
        }
        if (r < nr) store_row(out, ld, rb + r, kb, K, acc[r], vec_ok);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// k_score_lookups: a state of staged lookup features ONLY (bb / gp / dd / bnb columns, none masked beyond the folded kind:
// a mixture of categoricals and counts).  k_score_tile gives such a plan sixteen waves of 8 rows -- its registers are
// sized for the nich phase it does not have --; here every wave is a lookup wave of 16 sums (16 rows, or -- PAIR, at most
// 128 groups -- 32), so a staged feature group serves 256 (512) rows: half (a quarter of) the table copies and barriers a
// row.  prior lo + lookups in plan order + prior hi: k_score_tile's sums for such a plan, same bits.
// ---------------------------------------------------------------------------
template <bool LOO, bool CRP, bool PAIR>
__global__ __launch_bounds__(1024, 4) void k_score_lookups(const FeatDesc *__restrict__ feats, int nfeat, uint32_t K, uint32_t kpad, uint64_t row0,
                                                            uint64_t nrows, const int32_t *__restrict__ z, const float *__restrict__ own,
                                                            const float *__restrict__ crp, float *__restrict__ out, uint64_t ld) {
  constexpr int R = kRoleRows, RW = PAIR ? 2 * kRoleRows : kRoleRows;
  __shared__ float4 lds[kGrpRows * 64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t kb = PAIR ? (uint32_t)lane * 2u : blockIdx.y * kGroupTile + lane * 4;
  const bool vec_ok = PAIR ? ((ld & 1) == 0) && ((reinterpret_cast<uintptr_t>(out) & 7) == 0)
                           : ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  const uint64_t rows_per_wg = 16 * RW;
  const uint64_t nchunks = (nrows + rows_per_wg - 1) / rows_per_wg;
  for (uint64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const uint64_t rb = chunk * rows_per_wg + (uint64_t)wave * RW;        // relative to row0
    const int nr = rb >= nrows ? 0 : (int)((nrows - rb) < (uint64_t)RW ? (nrows - rb) : (uint64_t)RW);
    float4 acc[R];
    int single = 0;
    if (LOO && CRP && lane < nr) {
      const int g0 = z[rb + lane];
      single = g0 >= 0 && (uint32_t)g0 < K && __builtin_isinf(crp[kpad + g0]) ? 1 : 0;
    }
    if (CRP) {
      const float4 hi0 = PAIR ? pair_dup(ld2(crp + kb)) : ld4(crp + kb);
      const float4 lo = PAIR ? pair_dup(ld2(crp + crp_lo_cnt(kpad) + kb)) : ld4(crp + crp_lo_cnt(kpad) + kb);
      const float e0 = crp[2 * (size_t)kpad + 2], e1 = crp[2 * (size_t)kpad + 3];
#pragma unroll
      for (int r = 0; r < R; r++) {
        if constexpr (PAIR)
          acc[r] = crp_prior_pair_lo(make_float2(hi0.x, hi0.y), make_float2(lo.x, lo.y), LOO && lane_bcast(single, 2 * r) ? e1 : e0,
                                     LOO && lane_bcast(single, 2 * r + 1) ? e1 : e0);
        else acc[r] = crp_prior4_lo(hi0, lo, LOO && lane_bcast(single, r) ? e1 : e0);
      }
    } else {
#pragma unroll
      for (int r = 0; r < R; r++) acc[r] = make_float4(0, 0, 0, 0);
    }
    score_tile_groups<R, 16, false, false, PAIR>(feats, nfeat, kpad, blockIdx.y, lane, row0 + rb, nr, row0, lds, acc);
    // the rows' finish: + hi of the prior (fetched again: an L2 hit per chunk, not held across the lookups), the
    // leave-one-out entry in registers, the stores
    float4 hi = make_float4(0, 0, 0, 0);
    float le0 = 0, le1 = 0;
    if (CRP) {
      const float *again = crp;
      asm volatile("" : "+s"(again));
      hi = PAIR ? pair_dup(ld2(again + kb)) : ld4(again + kb);
      le0 = again[2 * (size_t)kpad], le1 = again[2 * (size_t)kpad + 1];
    }
    int gz = -1;
    float sloo = 0.f;
    if (LOO && lane < nr) {
      gz = z[rb + lane];
      sloo = own[rb + lane];
    }
    if (LOO && gz >= 0 && ((uint32_t)gz >= K || (uint32_t)gz / kGroupTile != blockIdx.y)) gz = -1;   // not in this k-tile
#pragma unroll
    for (int r = 0; r < R; r++) {
      if constexpr (PAIR) {
        if (CRP)
          add4(acc[r], crp_prior_pair(make_float2(hi.x, hi.y), LOO && lane_bcast(single, 2 * r) ? le1 : le0,
                                      LOO && lane_bcast(single, 2 * r + 1) ? le1 : le0));
        if (LOO)
          replace_own_pair(acc[r], lane, lane_bcast(gz, 2 * r), lane_bcast(sloo, 2 * r), lane_bcast(gz, 2 * r + 1), lane_bcast(sloo, 2 * r + 1));
        if (2 * r < nr) store_half_row(out, ld, rb + 2 * r, lane, K, acc[r].x, acc[r].y, vec_ok);
        if (2 * r + 1 < nr) store_half_row(out, ld, rb + 2 * r + 1, lane, K, acc[r].z, acc[r].w, vec_ok);
      } else {
        if (CRP) add4(acc[r], crp_prior4(hi, LOO && lane_bcast(single, r) ? le1 : le0));
        if (LOO) {
          const int g = lane_bcast(gz, r);
          if (g >= 0) replace_own(acc[r], kb, g, lane_bcast(sloo, r));
        }
        if (r < nr) store_row(out, ld, rb + r, kb, K, acc[r], vec_ok);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// k_score_nich_pack: a state of plain nich features only (two or more: a mixture of independent Gaussians per dimension).
// The tile plan has no first phase then, and what the role-split kernels' nich waves do needs neither the table slot nor
// a barrier: here ALL sixteen waves of a workgroup are such waves -- 16 rows each (32 in PAIR mode, at most 128 groups),
// the constants from the pack in L2 (score_block.hpp nich_phase_packed), the rows' values from the x matrix -- and a
// chunk is 256 (512) rows.  (prior lo + nothing) + (nich features) + prior hi, the leave-one-out entry in registers:
// the sums score_tile<SPLIT> forms for such a plan, same bits.  (The kernels that stage the nich constants through LDS
// feature group by feature group ran sixteen nich columns on a million rows at 1.16 ms; this one: see DESIGN section 5.)
// ---------------------------------------------------------------------------
// LOOK: the plan has a first phase of a FEW lookup features (abi.cpp plan_groups: at most kPackMaxLookups): their table rows
// are gathered from L2 into a second set of sums (pack_l2_lookups) -- (prior lo + lookups) + (nich features), as everywhere.
template <bool LOO, bool CRP, bool PAIR, bool LOOK>
__global__ __launch_bounds__(kNichPackWaves * 64, kNichPackWaves / 4) void k_score_nich_pack(const FeatDesc *__restrict__ feats, int nsplit, uint32_t K, uint32_t kpad, uint64_t row0,
                                                              uint64_t nrows, const int32_t *__restrict__ z, const float *__restrict__ own,
                                                              const float *__restrict__ crp, float *__restrict__ out, uint64_t ld) {
  constexpr int R = kRoleRows, RW = PAIR ? 2 * kRoleRows : kRoleRows;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t kb = PAIR ? (uint32_t)lane * 2u : blockIdx.y * kGroupTile + lane * 4;
  const bool vec_ok = PAIR ? ((ld & 1) == 0) && ((reinterpret_cast<uintptr_t>(out) & 7) == 0)
                           : ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  const uint64_t rows_per_wg = kNichPackWaves * RW;
  const uint64_t nchunks = (nrows + rows_per_wg - 1) / rows_per_wg;
  float4 hi = make_float4(0, 0, 0, 0), lo = make_float4(0, 0, 0, 0);
  float le0 = 0, le1 = 0, e0 = 0, e1 = 0;
  if (CRP) {
    hi = PAIR ? pair_dup(ld2(crp + kb)) : ld4(crp + kb);
    lo = PAIR ? pair_dup(ld2(crp + crp_lo_cnt(kpad) + kb)) : ld4(crp + crp_lo_cnt(kpad) + kb);
    le0 = crp[2 * (size_t)kpad], le1 = crp[2 * (size_t)kpad + 1];
    e0 = crp[2 * (size_t)kpad + 2], e1 = crp[2 * (size_t)kpad + 3];
  }
  for (uint64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const uint64_t rb = chunk * rows_per_wg + (uint64_t)wave * RW;        // relative to row0
    const int nr = rb >= nrows ? 0 : (int)((nrows - rb) < (uint64_t)RW ? (nrows - rb) : (uint64_t)RW);
    if (nr == 0) continue;                                // (wave-uniform; nothing in this kernel waits for another wave)
    int gz = -1, single = 0;
    float sloo = 0.f;
    if (LOO && lane < nr) {
      gz = z[rb + lane];
      sloo = own[rb + lane];
      if (CRP) single = gz >= 0 && (uint32_t)gz < K && __builtin_isinf(crp[kpad + gz]) ? 1 : 0;
    }
    if (LOO && gz >= 0 && ((uint32_t)gz >= K || (uint32_t)gz / kGroupTile != blockIdx.y)) gz = -1;   // not in this k-tile
    float4 acc[R];
    const uint64_t myrow = row0 + (lane < nr ? rb + lane : (nr ? rb : 0));   // (a row of the call's range for idle lanes and idle waves)
    [[maybe_unused]] float4 accl[LOOK ? R : 1];
    if constexpr (LOOK) {
#pragma unroll
      for (int r = 0; r < R; r++) {
        if (!CRP) accl[r] = make_float4(0, 0, 0, 0);
        else if constexpr (PAIR)
          accl[r] = crp_prior_pair_lo(make_float2(hi.x, hi.y), make_float2(lo.x, lo.y), LOO && lane_bcast(single, 2 * r) ? e1 : e0,
                                      LOO && lane_bcast(single, 2 * r + 1) ? e1 : e0);
        else accl[r] = crp_prior4_lo(hi, lo, LOO && lane_bcast(single, r) ? e1 : e0);
      }
      pack_l2_lookups<R, PAIR>(feats, nsplit, kpad, kb, myrow, accl);
    }
    nich_phase_packed<R, false, PAIR, LOOK ? 2 : kNichPackNC>(feats, nsplit, kpad, kb, row0 + rb, nr, myrow, acc);
    if constexpr (LOOK) {                                   // (prior lo + lookups) + (nich features), then the prior's high half
#pragma unroll
      for (int r = 0; r < R; r++) {
        float4 t = accl[r];
        add4(t, acc[r]);
        if (CRP) {
          if constexpr (PAIR)
            add4(t, crp_prior_pair(make_float2(hi.x, hi.y), LOO && lane_bcast(single, 2 * r) ? le1 : le0, LOO && lane_bcast(single, 2 * r + 1) ? le1 : le0));
          else add4(t, crp_prior4(hi, LOO && lane_bcast(single, r) ? le1 : le0));
        }
        acc[r] = t;
      }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
      if constexpr (PAIR) {
        if (CRP && !LOOK) {
          const bool sa = LOO && lane_bcast(single, 2 * r), sb = LOO && lane_bcast(single, 2 * r + 1);
          float4 t = crp_prior_pair_lo(make_float2(hi.x, hi.y), make_float2(lo.x, lo.y), sa ? e1 : e0, sb ? e1 : e0);
          add4(t, acc[r]);                                // (prior lo) + (nich features)
          add4(t, crp_prior_pair(make_float2(hi.x, hi.y), sa ? le1 : le0, sb ? le1 : le0));
          acc[r] = t;
        }
        if (LOO)
          replace_own_pair(acc[r], lane, lane_bcast(gz, 2 * r), lane_bcast(sloo, 2 * r), lane_bcast(gz, 2 * r + 1), lane_bcast(sloo, 2 * r + 1));
        if (2 * r < nr) store_half_row(out, ld, rb + 2 * r, lane, K, acc[r].x, acc[r].y, vec_ok);
        if (2 * r + 1 < nr) store_half_row(out, ld, rb + 2 * r + 1, lane, K, acc[r].z, acc[r].w, vec_ok);
      } else {
        if (CRP && !LOOK) {
          const bool s1 = LOO && lane_bcast(single, r);
          float4 t = crp_prior4_lo(hi, lo, s1 ? e1 : e0);
          add4(t, acc[r]);
          add4(t, crp_prior4(hi, s1 ? le1 : le0));
          acc[r] = t;
        }
        if (LOO) {
          const int g = lane_bcast(gz, r);
          if (g >= 0) replace_own(acc[r], kb, g, lane_bcast(sloo, r));
        }
        if (r < nr) store_row(out, ld, rb + r, kb, K, acc[r], vec_ok);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// counts beyond the exact tables of gp / bnb / dm (only launched when a column's maximum exceeds
// the table cap): one wave per row, lanes over groups; adds the exact value in double (the own
// group's value is already the leave-one-out one).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gp_large_fix(const FeatDesc *__restrict__ feats, int f, uint32_t K,
                                                       uint32_t kpad, uint64_t row0, uint64_t nrows,
                                                       const int32_t *__restrict__ z, float *__restrict__ out,
                                                       uint64_t ld) {
  const FeatDesc fd = feats[f];
  const int lane = threadIdx.x & 63;
  const uint64_t wave_id = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint64_t nwaves = (uint64_t)gridDim.x * 4;
  if (fd.family == MSC_DM) {       // rows whose total is beyond the tables were skipped whole by the tile kernel
    for (uint64_t n = wave_id; n < nrows; n += nwaves) {
      if (fd.dm_tot[row0 + n] < kGpMaxTable || load_masked(fd, row0 + n, true)) continue;
      const int32_t *x = reinterpret_cast<const int32_t *>(fd.col) + (row0 + n) * fd.dim;
      const int g = z ? z[n] : -1;
      for (uint32_t k = lane; k < K; k += 64)
        if ((int)k != g) out[n * ld + k] += (float)dm_score_direct(fd.hp, fd.dim, fd.raw_u32 + k, kpad, x, false);
    }
    return;
  }
  const double al = fd.hp[0], ib = fd.hp[1];
  for (uint64_t n = wave_id; n < nrows; n += nwaves) {
    const uint32_t v = reinterpret_cast<const uint32_t *>(fd.col)[row0 + n];
    if (v < fd.vcap || (fd.mask != nullptr && fd.mask[row0 + n] != 0)) continue;
    const int g = z ? z[n] : -1;
    const double rowc = fd.family == MSC_GP ? gp_row_const(v) : 0.0;
    for (uint32_t k = lane; k < K; k += 64) {
      if ((int)k == g) continue;
      if (fd.family == MSC_GP) {
        const double a = al + (double)fd.raw_u32[(size_t)kpad + k], b = ib + (double)fd.raw_u32[k];
        const double nse = (double)fd.tab[(size_t)GP_NSE_HI * kpad + k] + (double)fd.tab[(size_t)GP_NSE_LO * kpad + k];
        out[n * ld + k] += gp_eval_large((double)v, rowc, a, b, nse);
      } else {
        out[n * ld + k] += (float)bnb_score(fd.hp, (double)fd.raw_u32[k], (double)fd.raw_u32[(size_t)kpad + k], (double)v);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// host-side launchers (called from abi.cpp)
// ---------------------------------------------------------------------------
MSC_DEFINE_BIND_ERROR_WORD(bind_error_word_score)

int launch_prepare(hipStream_t stream, const FeatDesc *feats_dev, uint32_t nfeat, uint32_t kpad, uint32_t value_slices) {
  dim3 grid((kpad + 255) / 256, nfeat, value_slices ? value_slices : 1);
  hipLaunchKernelGGL(k_prepare, grid, dim3(256), 0, stream, feats_dev, kpad);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_commit_prepare(hipStream_t stream, const FeatDesc *feats_dev, int nfeat, uint32_t K, uint32_t kpad,
                          const long long *cnt_acc, uint32_t *cnt_u32, float alpha, float *crp, uint64_t *rng_bump,
                          uint32_t value_slices) {
  hipLaunchKernelGGL(k_commit_prepare, dim3((kpad + 255) / 256, nfeat + 1, value_slices ? value_slices : 1), dim3(256), 0, stream, feats_dev, nfeat, K, kpad,
                     cnt_acc, cnt_u32, alpha, crp, rng_bump);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_dm_prepare(hipStream_t stream, const FeatDesc *feats_dev, int f, uint32_t dim, uint32_t kpad, uint32_t value_slices) {
  hipLaunchKernelGGL(k_dm_prepare, dim3((kpad + 255) / 256, dim + 1, value_slices ? value_slices : 1), dim3(256), 0, stream, feats_dev, f, kpad);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_entity_op(hipStream_t stream, const FeatDesc *feats_dev, int nfeat, uint32_t K, uint32_t kpad, uint64_t row,
                     uint32_t group, int sign, long long *cnt_acc, uint32_t *cnt_u32, float alpha, float *crp, int32_t *z_slot) {
  hipLaunchKernelGGL(k_entity_op, dim3((unsigned)nfeat + 1), dim3(256), 0, stream, feats_dev, nfeat, K, kpad, row, group, sign,
                     cnt_acc, cnt_u32, alpha, crp, z_slot);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_crp_prepare(hipStream_t stream, const uint32_t *cnt, uint32_t K, uint32_t kpad, float alpha,
                       float *crp) {
  hipLaunchKernelGGL(k_crp_prepare, dim3(1), dim3(256), 0, stream, cnt, K, kpad, alpha, crp);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_loo_own(hipStream_t stream, int num_cus, bool heavy, bool staged, const FeatDesc *feats_dev, int nfeat, uint32_t K,
                   uint32_t kpad, uint64_t row0, uint64_t nrows, const int32_t *z, const float *crp, float *own) {
  // staged: the plan has features whose leave-one-out blocks fit the LDS slot (abi.cpp plan_groups).  A workgroup of the
  // staged kernel takes as long for its 4096 rows as the plan has stages (~10 us each) however many workgroups run, so it
  // pays once the rows (nearly) fill the chip -- C3, 1M rows: 140 us staged, 196 us gathered; below about three quarters
  // of a workgroup per CU the gather kernel's time, which falls with the rows, is the shorter one (a 131k-row chunk:
  // 0.33 ms staged against 0.03 ms gathered)
  const char *knob = std::getenv("MSC_LOO_LDS");            // 0 / 1: A/B knob, read per call (the tests pin either kernel)
  const int lds_mode = knob ? std::atoi(knob) : -1;
  const bool use_lds = staged && (lds_mode < 0 ? nrows >= (uint64_t)kLooRows * kLooThreads * (uint64_t)num_cus * 3 / 4 : lds_mode != 0);
  if (use_lds) {
    static unsigned long long attr_devices = 0;
    if (first_use_on_device(attr_devices)) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_loo_own_lds<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLooSlotFloats * 4);
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_loo_own_lds<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kLooSlotFloats * 4);
    }
    const unsigned blocks = (unsigned)((nrows + kLooRows * kLooThreads - 1) / (kLooRows * kLooThreads));
    if (heavy)
      hipLaunchKernelGGL(k_loo_own_lds<true>, dim3(blocks), dim3(kLooThreads), kLooSlotFloats * 4, stream, feats_dev, nfeat, K,
                         kpad, row0, nrows, z, crp, own);
    else
      hipLaunchKernelGGL(k_loo_own_lds<false>, dim3(blocks), dim3(kLooThreads), kLooSlotFloats * 4, stream, feats_dev, nfeat, K,
                         kpad, row0, nrows, z, crp, own);
    return hipGetLastError() == hipSuccess ? 0 : -1;
  }
  if (heavy)
    hipLaunchKernelGGL(k_loo_own<true>, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, stream, feats_dev, nfeat,
                       K, kpad, row0, nrows, z, crp, own);
  else
    hipLaunchKernelGGL(k_loo_own<false>, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, stream, feats_dev, nfeat,
                       K, kpad, row0, nrows, z, crp, own);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_gp_large_fix(hipStream_t stream, int num_cus, const FeatDesc *feats_dev, int f, uint32_t K,
                        uint32_t kpad, uint64_t row0, uint64_t nrows, const int32_t *z, float *out, uint64_t ld) {
  uint64_t gx = (nrows + 3) / 4;
  const uint64_t cap = (uint64_t)num_cus * 8;
  if (gx > cap) gx = cap;
  hipLaunchKernelGGL(k_gp_large_fix, dim3((unsigned)(gx ? gx : 1)), dim3(256), 0, stream, feats_dev, f, K, kpad,
                     row0, nrows, z, out, ld);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// PAIR mode of the role-split kernels: one k-tile of at most 128 groups, rows enough for the role-split kernels at all
bool pair_mode_ok(int path, uint32_t K, bool few_rows) {
  static const bool off = std::getenv("MSC_NO_PAIR") != nullptr;     // (A/B knob)
  return !off && (path == MSC_PATH_TILE_ROLES || path == MSC_PATH_NICH_PACK || path == MSC_PATH_LOOKUPS) && tile_roles_enabled() && K <= 128 && !few_rows;
}
// (the A/B switch between these kernels and the ones that run the phases one after the other is the plan's: MSC_NO_ROLES,
// abi.cpp plan_groups)
bool tile_roles_enabled() { return true; }

// (blocks of one row, like a memset's, lose in this kernel although they win as a bare fill: a wave that visits only a few
// rows does not pay for loading the group constants, and with many visits the fronts are too many)
const Nich1Shape kNich1Shapes[kNich1NumShapes] = {{4, 2}, {4, 1}, {4, 4}, {4, 6}, {4, 3}, {4, 8}, {4, 5}, {4, 12}};

// ---------------------------------------------------------------------------
// k_score_tail_rows: the LAST, partly filled 256-group tile of a tile state, when it holds at most 64 groups.  A pass of the
// tile kernels costs what a full tile costs however few groups it holds -- the nich half is vector issue, 56 cycles per
// wave evaluation whatever the lanes carry -- so a CRP state that has just grown past 256 groups paid twice (K = 300 on
// C3's columns: 3.6 ms a scoring pass against 1.9 at K = 256).  Here LANE <-> ROW: a lane owns a row, its values arrive
// with coalesced loads and are used as they stand (no broadcast), and the tail's groups stream past: a lookup feature is
// one LDS read per group at a constant offset from the row's table row (tables of the tail's groups alone, packed
// [table row][64 groups] by k_tail_pack once per pass, staged with a 65-float row stride -- rows that differ in their
// value hit different banks), a nich feature takes its constants per group as SCALAR operands (s_load from the feature's
// table: no register, no LDS).  Same plan as the tile kernels (abi.cpp plan_groups: lookup features in the first phase,
// plain nich features in the second), sums in plan order from the prior's low half, the prior's high half last, the
// leave-one-out value in place of the row's own group.  TGP = the tail's groups rounded up to 16 (a lane's sums stay in
// registers across the whole plan).  Every row count takes this kernel for the tail of an eligible state (launch_score),
// so a shard reproduces the whole.  (Round 3's first version had the lanes as (row slot, group quad) with the row values
// broadcast by shuffles: 1.08 ms for C3's tail of 44 groups; this one 0.46 ms -- tools/microbench/README.md.)
// ---------------------------------------------------------------------------
constexpr int kTailRowsWaves = 8;                     // 512 rows per workgroup visit
constexpr uint32_t kTailStride = 65;                   // floats per staged table row (64 groups + 1)

// table rows a first-phase feature brings to the slot: a lookup feature's selectable rows; none for a masked nich column
// (the one generic kind the plan admits there: evaluated like the second phase's, under the row's mask)
// ... and all dim + 1 (hi, lo) tables of a dm feature (the plan admits it only when they are staged whole: small counts)
__host__ __device__ inline uint32_t tail_table_rows(uint32_t family, uint32_t kind, uint32_t run_clamp, uint32_t dm_rows) {
  return family == MSC_DM ? dm_rows : kind == MSC_KIND_GENERIC ? 0u : run_clamp + 1;
}

__global__ __launch_bounds__(256) void k_tail_pack(const FeatDesc *__restrict__ feats, uint32_t kpad, uint32_t k0,
                                                    float *__restrict__ pack) {
  const int f = blockIdx.x;
  uint32_t off = 0;
  for (int i = 0; i < f; i++) off += tail_table_rows(feats[i].family, feats[i].kind, feats[i].run_clamp, feats[i].dm_rows);
  const FeatDesc &fd = feats[f];
  const uint32_t first_row = is_count_family(fd.family) ? (uint32_t)GP_T0 : 0u, rows = tail_table_rows(fd.family, fd.kind, fd.run_clamp, fd.dm_rows);
  for (uint32_t e = threadIdx.x; e < rows * 64u; e += 256u) {
    const uint32_t r = e >> 6, g = e & 63u;
    pack[(size_t)(off + r) * 64 + g] = fd.tab[(size_t)(first_row + r) * kpad + k0 + g];      // (k0 + 63 < kpad)
  }
}

// DRAW (a state of at most 64 groups, one launch, k0 = 0): nothing is stored -- a lane holds its row's whole score vector,
// so it draws the row's new group by itself (maximum, exponentials, running sum against the dart: sample_discrete's CDF
// order, no cross-lane step) and writes it to z; (seed, sweep) from `rng`, the uniform of global row row_id0 + r.
// MNICH: the first phase may hold masked nich columns (an instantiation of its own: the branch costs the others registers)
// EST: the sweeps' nich form (family_math.hpp nich_accum<true>)
// DMF: the first phase may hold dm features (an instantiation of its own, like MNICH's): dim + 1 count lookups a row, each a
// (hi, lo) pair of staged table rows, summed apart sixteen groups at a time and added to the score once (score_block.hpp
// score_dm_feature_staged: the same sums in the same order); a masked row, or one whose total is beyond the tables, reads
// entry 0 of every table, which is exactly zero
template <int TGP, bool SPLIT, bool DRAW = false, bool MNICH = false, bool EST = false, bool DMF = false>
__global__ __launch_bounds__(kTailRowsWaves * 64, (MNICH || DMF) ? 2 : 4) void k_score_tail_rows(
    const FeatDesc *__restrict__ feats_g, int nfeat, int nsplit, uint32_t K, uint32_t kpad, uint32_t k0, uint64_t row0,
    uint64_t nrows, int32_t *z, const float *__restrict__ own, const float *__restrict__ crp,
    float *__restrict__ out, uint64_t ld, const float *__restrict__ pack, uint32_t cap_rows, uint32_t kend,
    const uint64_t *__restrict__ rng, uint64_t row_id0, ZeroSpans zero) {
  // (groups [k0, kend) are this launch's; K is the table's: an id outside [0, K) is an unassigned row)
  uint64_t seed = 0, sweep = 0;
  if (DRAW) {
    seed = rng[0];
    sweep = rng[1];
    zero_spans(zero);                                    // (the additive tables, for the accumulate pass that follows)
  }
  extern __shared__ __attribute__((aligned(16))) float tl[];              // cap_rows x kTailStride
  // the prior of the tail's groups as a row starts from it / ends with it: [0] low halves, [1] the same for a row that is
  // its group's only member, [2] / [3] the high halves likewise -- a lane picks its pair of rows by address
  __shared__ __attribute__((aligned(16))) float pr[4][64];
  __shared__ __attribute__((aligned(16))) float c0s[64];                  // the nich features' c0 summed per group (plan order)
  float *const mhl = tl + (size_t)cap_rows * kTailStride;                 // [nich feature][64]: s*mu (hi) -- see the second phase
  typedef const __attribute__((address_space(4))) FeatDesc *scalar_fd;    // (constant address space: scalar loads)
  typedef const __attribute__((address_space(4))) float *scalar_f;
  const scalar_fd feats = (scalar_fd)feats_g;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool loo = z != nullptr, pri = crp != nullptr;
  const bool vec_ok = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  const uint64_t rows_per_wg = (uint64_t)kTailRowsWaves * 64;
  const uint64_t nchunks = (nrows + rows_per_wg - 1) / rows_per_wg;
  if (threadIdx.x < 64) {
    float hi = 0.f, lo = 0.f, le0 = 0.f, le1 = 0.f, e0 = 0.f, e1 = 0.f;
    if (pri) {
      hi = crp[k0 + threadIdx.x];                          // (k0 + 63 < kpad)
      lo = crp[crp_lo_cnt(kpad) + k0 + threadIdx.x];
      le0 = crp[2 * (size_t)kpad];
      le1 = crp[2 * (size_t)kpad + 1];
      e0 = crp[2 * (size_t)kpad + 2];
      e1 = crp[2 * (size_t)kpad + 3];
    }
    // (without a prior: sums start from +0 and end with "+ -0", which leaves every float as it is -- no branch per group)
    const bool empty = __builtin_isinf(hi);
    pr[0][threadIdx.x] = !pri ? 0.f : empty ? e0 : lo;
    pr[1][threadIdx.x] = !pri ? 0.f : empty ? e1 : lo;
    pr[2][threadIdx.x] = !pri ? -0.f : empty ? le0 : hi;
    pr[3][threadIdx.x] = !pri ? -0.f : empty ? le1 : hi;
    float c0 = 0.f;
    for (int f = nsplit; f < nfeat; f++) {
      c0 += feats_g[f].tab[(size_t)NICH_C0 * kpad + k0 + threadIdx.x];
      mhl[(size_t)(f - nsplit) * 64 + threadIdx.x] = feats_g[f].tab[(size_t)NICH_MU_HI * kpad + k0 + threadIdx.x];
    }
    c0s[threadIdx.x] = c0;
  }
  __syncthreads();
  auto load_value = [&](int f, uint64_t rr) -> uint32_t {                 // the row's value of feature f, raw
    const int fi = f < nfeat ? f : nfeat - 1;
    const int ct = feats[fi].col_type;
    const void *col = feats[fi].col;
    const bool u8 = ct == MSC_TYPE_B || ct == MSC_TYPE_I8 || ct == MSC_TYPE_U8;
    return u8 ? (uint32_t)reinterpret_cast<const uint8_t *>(col)[rr] : reinterpret_cast<const uint32_t *>(col)[rr];
  };
  for (uint64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const uint64_t r = chunk * rows_per_wg + (uint64_t)wave * 64 + lane;     // relative to row0
    const bool has_row = r < nrows;
    const uint64_t rr = row0 + (has_row ? r : 0);
    int gown = (loo && has_row) ? z[r] : -1;
    if ((uint32_t)gown >= K) gown = -1;
    const bool single = pri && gown >= 0 && __builtin_isinf(crp[kpad + gown]);   // the row is its group's only member
    const float ownv = gown >= 0 ? own[r] : 0.f;
    const float *prl = pr[single ? 1 : 0];
    // SPLIT: (prior + lookups) + (nich features), then the prior's high half -- the association of score_tile, two sums per
    // group, the tile kernels' bits; else one sum per group: prior + lookups + the nich features' c0, then the evaluations
    float acc[TGP], accn[SPLIT ? TGP : 1];
    // the prior is a (hi, lo) pair per group: the sums start from lo, hi is added after the last feature (k_score_tile)
#pragma unroll
    for (int g = 0; g < TGP; g++) acc[g] = prl[g];
    // ---- first phase: lookup features, as many consecutive ones per stage as the slot holds ----
    uint32_t srow = 0, off = 0;
    int stage_end = 0;
    for (int fb = 0; fb < nsplit; fb += 8) {
      // eight features' values of the row go out together: one memory round trip per batch (the other waves of the SIMD work meanwhile)
      uint32_t w0 = load_value(fb, rr), w1 = load_value(fb + 1, rr), w2 = load_value(fb + 2, rr), w3 = load_value(fb + 3, rr),
               w4 = load_value(fb + 4, rr), w5 = load_value(fb + 5, rr), w6 = load_value(fb + 6, rr), w7 = load_value(fb + 7, rr);
#pragma unroll 1
      for (int i = 0; i < 8; i++) {
        const int f = fb + i;
        if (f >= nsplit) break;
        if (f == stage_end) {
          srow += off;
          off = 0;
          uint32_t rows = 0;
          while (stage_end < nsplit) {
            const uint32_t rw = tail_table_rows(feats[stage_end].family, feats[stage_end].kind, feats[stage_end].run_clamp, feats[stage_end].dm_rows);
            if (rows + rw > cap_rows && rows > 0) break;       // (the launcher sizes the slot for the largest table)
            rows += rw;
            stage_end++;
          }
          __syncthreads();                               // the slot's previous readers are done
          for (uint32_t e = threadIdx.x * 4u; e < rows * 64u; e += kTailRowsWaves * 64u * 4u) {
            const float4 v = *reinterpret_cast<const float4 *>(pack + (size_t)srow * 64 + e);
            float *d = tl + (size_t)(e >> 6) * kTailStride + (e & 63u);
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
          }
          __syncthreads();
        }
        if (DMF && feats[f].family == MSC_DM) {
          const uint32_t dim = feats[f].dim;
          const uint32_t *xr = reinterpret_cast<const uint32_t *>(feats[f].col) + rr * dim;
          const uint32_t *meta = feats[f].dm_meta;
          const uint32_t tot = has_row ? feats[f].dm_tot[rr] : 0u;
          bool dead = !has_row || tot >= kGpMaxTable;
          if (feats[f].mask != nullptr)
            for (uint32_t e = 0; e < dim; e++) dead |= feats[f].mask[rr * dim + e] != 0;
#pragma unroll
          for (int gb = 0; gb < TGP; gb += 16) {
            float hi[16], lo[16];
#pragma unroll
            for (int g = 0; g < 16; g++) hi[g] = lo[g] = 0.f;
#pragma unroll 1
            for (uint32_t st = 0; st <= dim; st++) {
              const uint32_t first_row = meta[2 * st], vcap = meta[2 * st + 1];
              uint32_t v = st < dim ? xr[st] : tot;
              v = (dead || v >= vcap) ? 0u : v;
              const float *b = tl + (size_t)(off + first_row + 2u * v) * kTailStride + gb;
#pragma unroll
              for (int g = 0; g < 16; g++) {
                hi[g] += b[g];
                lo[g] += b[kTailStride + g];
              }
            }
#pragma unroll
            for (int g = 0; g < 16; g++) acc[gb + g] += hi[g] + lo[g];
          }
          off += feats[f].dm_rows;
        } else if (MNICH && feats[f].kind == MSC_KIND_GENERIC) {
          // a masked nich column, in the caller's place among the lookups: acc += nich_eval(...) unless the row's value is
          // masked (add_feature's generic branch, the same bits); constants as scalar operands, four groups at a time
          const bool masked = feats[f].mask[rr] != 0;
          const float x = __uint_as_float(w0);
          const scalar_f tab = (scalar_f)(feats[f].tab) + k0;
          typedef float f32x4s __attribute__((ext_vector_type(4)));
          typedef const volatile __attribute__((address_space(4))) f32x4s *scalar_f4;
#pragma unroll
          for (int gb = 0; gb < TGP; gb += 4) {
            const f32x4s mh = *(scalar_f4)(tab + (size_t)NICH_MU_HI * kpad + gb), ml = *(scalar_f4)(tab + (size_t)NICH_MU_LO * kpad + gb),
                         c0 = *(scalar_f4)(tab + (size_t)NICH_C0 * kpad + gb), c1l = *(scalar_f4)(tab + (size_t)NICH_C1LN2 * kpad + gb),
                         c1 = *(scalar_f4)(tab + (size_t)NICH_C1 * kpad + gb), c2 = *(scalar_f4)(tab + (size_t)NICH_C2 * kpad + gb);
#pragma unroll
            for (int j = 0; j < 4; j++) {
              const float e = acc[gb + j] + nich_eval(x, mh[j], ml[j], c0[j], c1l[j], c1[j], c2[j]);
              acc[gb + j] = masked ? acc[gb + j] : e;
            }
            __builtin_amdgcn_sched_barrier(0);               // (four evaluations' temporaries at a time)
          }
        } else {
          const uint32_t rc = feats[f].run_clamp;
          const int v = (int)w0;
          const uint32_t idx = v < 0 ? 0u : ((uint32_t)v > rc ? rc : (uint32_t)v);
          const float *b = tl + (size_t)(off + idx) * kTailStride;
#pragma unroll
          for (int g = 0; g < TGP; g++) acc[g] += b[g];
          off += rc + 1;
        }
        w0 = w1; w1 = w2; w2 = w3; w3 = w4; w4 = w5; w5 = w6; w6 = w7;    // (the batch's values pass through w0: no indexed register)
      }
    }
    // ---- second phase: plain nich features, in the plan's BLOCKS (family_math.hpp "nich BLOCKS"; the steps of
    // score_block.hpp nich_segment, so that SPLIT gives the tile kernels' bits).  The accumulator starts from the
    // features' summed c0; a block whose c1 the head kernel found equal is ONE compensated log1p of the product of its
    // members' 1 + t per group, every other feature nich_accum's two fused multiply-adds.  The groups' constants are
    // SCALAR operands but for s*mu (hi), which an instruction needs beside s (one scalar operand an instruction on this
    // chip): that one comes from LDS, a broadcast read.  A FAR row (a value beyond its feature's xlim) takes nich_accum
    // for every feature: the lane decides for its row (a divergent region, never entered with data that sits where its
    // groups are).
    if (SPLIT) {
#pragma unroll
      for (int g = 0; g < TGP; g++) accn[g] = c0s[g];
    } else if (nsplit < nfeat) {
#pragma unroll
      for (int g = 0; g < TGP; g++) acc[g] += c0s[g];
    }
    bool far = false;
    const bool blocks = nsplit < nfeat && feats[nsplit].nich_info != nullptr;     // (records for all of the phase's features, or none)
    for (int fb = nsplit; blocks && fb < nfeat; fb += 8) {
      const uint32_t w0 = load_value(fb, rr), w1 = load_value(fb + 1, rr), w2 = load_value(fb + 2, rr), w3 = load_value(fb + 3, rr),
                     w4 = load_value(fb + 4, rr), w5 = load_value(fb + 5, rr), w6 = load_value(fb + 6, rr), w7 = load_value(fb + 7, rr);
      auto lim = [&](int f) { return feats[f < nfeat ? f : nfeat - 1].nich_info->xlim; };     // (load_value clamps the same way)
      far |= !(__builtin_fabsf(__uint_as_float(w0)) <= lim(fb)) | !(__builtin_fabsf(__uint_as_float(w1)) <= lim(fb + 1)) |
             !(__builtin_fabsf(__uint_as_float(w2)) <= lim(fb + 2)) | !(__builtin_fabsf(__uint_as_float(w3)) <= lim(fb + 3)) |
             !(__builtin_fabsf(__uint_as_float(w4)) <= lim(fb + 4)) | !(__builtin_fabsf(__uint_as_float(w5)) <= lim(fb + 5)) |
             !(__builtin_fabsf(__uint_as_float(w6)) <= lim(fb + 6)) | !(__builtin_fabsf(__uint_as_float(w7)) <= lim(fb + 7));
    }
    // one feature, nich_accum: NB groups' constants at a time, 4 NB scalar registers (volatile: left to itself the compiler
    // merges the loads of sixteen groups, holds hundreds of registers' worth at once and spills them through vector
    // lanes); four at a time where two sums per group leave few registers for the evaluations' temporaries
    auto plain_feature = [&](int f, float x) {
      const scalar_f tab = (scalar_f)(feats[f].tab) + k0;
      const float *mhf = mhl + (size_t)(f - nsplit) * 64;
      constexpr int NB = (!SPLIT || TGP <= 32) ? 8 : 4;
      typedef float f32xn __attribute__((ext_vector_type(NB)));
      typedef const volatile __attribute__((address_space(4))) f32xn *scalar_fn;
#pragma unroll
      for (int gb = 0; gb < TGP; gb += NB) {
        const f32xn ml = *(scalar_fn)(tab + (size_t)NICH_MU_LO * kpad + gb), c1l = *(scalar_fn)(tab + (size_t)NICH_C1LN2 * kpad + gb),
                    c1 = *(scalar_fn)(tab + (size_t)NICH_C1 * kpad + gb), c2 = *(scalar_fn)(tab + (size_t)NICH_C2 * kpad + gb);
#pragma unroll
        for (int j = 0; j < NB; j++) {
          float &a = SPLIT ? accn[gb + j] : acc[gb + j];
          a = nich_accum<EST>(a, x, mhf[gb + j], ml[j], c1l[j], c1[j], c2[j]);
        }
        __builtin_amdgcn_sched_barrier(0);               // (a block's temporaries at a time: left to interleave the blocks, some instantiations spill)
      }
    };
    // a block of M features: four groups' constants at a time, (2 M + 1) x 4 scalar registers
    auto product_block = [&](int f, auto mtag, const float (&x)[4]) {
      constexpr int M = decltype(mtag)::value;
      typedef float f32x4s __attribute__((ext_vector_type(4)));
      typedef const volatile __attribute__((address_space(4))) f32x4s *scalar_f4;
      scalar_f tab[M];
      const float *mhf[M];
#pragma unroll
      for (int j = 0; j < M; j++) {
        tab[j] = (scalar_f)(feats[f + j].tab) + k0;
        mhf[j] = mhl + (size_t)(f + j - nsplit) * 64;
      }
#pragma unroll
      for (int gb = 0; gb < TGP; gb += 4) {
        f32x4s ml[M], sc[M];
#pragma unroll
        for (int j = 0; j < M; j++) {
          ml[j] = *(scalar_f4)(tab[j] + (size_t)NICH_MU_LO * kpad + gb);
          sc[j] = *(scalar_f4)(tab[j] + (size_t)NICH_C2 * kpad + gb);
        }
        const f32x4s c1l = *(scalar_f4)(tab[0] + (size_t)NICH_C1LN2 * kpad + gb);
#pragma unroll
        for (int i = 0; i < 4; i++) {
          float t[M];
#pragma unroll
          for (int j = 0; j < M; j++) t[j] = nich_t(x[j], mhf[j][gb + i], ml[j][i], sc[j][i]);
          float &a = SPLIT ? accn[gb + i] : acc[gb + i];
          a = nich_block_finish<EST>(a, nich_block_product<M>(t), c1l[i]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // the order of score_block.hpp nich_segment: per LDS feature group of the plan, its blocks of four that go as one, then
    // three, then two, then every feature left
    for (int s0 = nsplit; s0 < nfeat;) {
      const int s1 = (int)feats[s0].grp_end;
      if (blocks) {
#pragma unroll 1
        for (int M = kNichBlock; M >= 2; M--) {
          for (int f = s0; f < s1;) {
            const int len = (int)feats[f].blk_end - f;
            if (len == M && feats[f].nich_info->blk_ok != 0u) {
              const float x[4] = {__uint_as_float(load_value(f, rr)), __uint_as_float(load_value(f + 1, rr)),
                                  __uint_as_float(load_value(f + 2, rr)), __uint_as_float(load_value(f + 3, rr))};
              if (!far) {
                if (M == 2) product_block(f, std::integral_constant<int, 2>(), x);
                else if (M == 3) product_block(f, std::integral_constant<int, 3>(), x);
                else product_block(f, std::integral_constant<int, 4>(), x);
              }
            }
            f += len;
          }
        }
      }
#pragma unroll 1
      for (int f = s0; f < s1; f++) {
        const bool covered = blocks && feats[feats[f].blk_first].nich_info->blk_ok != 0u;    // (went as one with its block)
        const float x = __uint_as_float(load_value(f, rr));
        if (far || !covered) plain_feature(f, x);
      }
      s0 = s1;
    }
    const float *prh = pr[single ? 3 : 2];
    if (DRAW) {
      float m = -INFINITY;
#pragma unroll
      for (int g = 0; g < TGP; g++) {
        const float sg = gown == g ? ownv : (SPLIT && nsplit < nfeat ? acc[g] + accn[g] : acc[g]) + prh[g];
        acc[g] = (uint32_t)g < K ? sg : -INFINITY;
        m = fmaxf(m, acc[g]);
      }
      float total = 0.f;
#pragma unroll
      for (int g = 0; g < TGP; g++) {
        acc[g] = __builtin_amdgcn_exp2f((acc[g] - m) * 1.44269504088896340736f);   // exp(-inf) = 0
        total += acc[g];
      }
      const float dart = philox_uniform01(seed, sweep, row_id0 + r) * total;
      float c = 0.f;
      int pick = 0;
#pragma unroll
      for (int g = 0; g < TGP; g++) {
        c += acc[g];
        pick += c < dart ? 1 : 0;
      }
      if (has_row) z[r] = pick < (int)K ? pick : (int)K - 1;
      continue;
    }
#pragma unroll
    for (int q = 0; q < TGP / 4; q++) {
      float sv[4];
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const int g = 4 * q + c;
        sv[c] = gown == (int)(k0 + g) ? ownv : (SPLIT && nsplit < nfeat ? acc[g] + accn[g] : acc[g]) + prh[g];
      }
      if (has_row) store_row<false>(out, ld, r, k0 + 4 * q, kend, make_float4(sv[0], sv[1], sv[2], sv[3]), vec_ok);
    }
  }
}

template <int TGP, bool SPLIT, bool DRAW, bool MNICH, bool EST, bool DMF = false>
static void launch_tail_rows_m(hipStream_t stream, unsigned grid, size_t lds, const FeatDesc *feats_dev, int nfeat, int nsplit, uint32_t K,
                               uint32_t kpad, uint32_t k0, uint64_t row0, uint64_t nrows, const int32_t *z, const float *own,
                               const float *crp, float *out, uint64_t ld, const float *pack, uint32_t cap_rows, uint32_t kend,
                               const uint64_t *rng, uint64_t row_id0, ZeroSpans zero) {
  static unsigned long long attr_devices = 0;
  if (first_use_on_device(attr_devices))
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_score_tail_rows<TGP, SPLIT, DRAW, MNICH, EST, DMF>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  hipLaunchKernelGGL((k_score_tail_rows<TGP, SPLIT, DRAW, MNICH, EST, DMF>), (note_kernel(DRAW ? 1 : 0, "k_score_tail_rows<%d, %s, %s, %s, %s, %s>", TGP, tf(SPLIT), tf(DRAW), tf(MNICH), tf(EST), tf(DMF)), dim3(grid)), dim3(kTailRowsWaves * 64), lds, stream, feats_dev, nfeat, nsplit, K, kpad, k0,
                     row0, nrows, const_cast<int32_t *>(z), own, crp, out, ld, pack, cap_rows, kend, rng, row_id0, zero);
}
template <int TGP, bool SPLIT, bool DRAW = false, bool EST = false>
static void launch_tail_rows_t(hipStream_t stream, unsigned grid, size_t lds, int mnich, const FeatDesc *feats_dev, int nfeat, int nsplit, uint32_t K,
                               uint32_t kpad, uint32_t k0, uint64_t row0, uint64_t nrows, const int32_t *z, const float *own,
                               const float *crp, float *out, uint64_t ld, const float *pack, uint32_t cap_rows, uint32_t kend,
                               const uint64_t *rng = nullptr, uint64_t row_id0 = 0, ZeroSpans zero = ZeroSpans()) {
  // (mnich: 0 plain, 1 masked nich columns in the first phase, 2 dm features there -- TailPlan::masked_nich / dm)
  if (mnich == 2) launch_tail_rows_m<TGP, SPLIT, DRAW, false, EST, true>(stream, grid, lds, feats_dev, nfeat, nsplit, K, kpad, k0, row0, nrows, z, own, crp, out, ld, pack, cap_rows, kend, rng, row_id0, zero);
  else if (mnich) launch_tail_rows_m<TGP, SPLIT, DRAW, true, EST>(stream, grid, lds, feats_dev, nfeat, nsplit, K, kpad, k0, row0, nrows, z, own, crp, out, ld, pack, cap_rows, kend, rng, row_id0, zero);
  else launch_tail_rows_m<TGP, SPLIT, DRAW, false, EST>(stream, grid, lds, feats_dev, nfeat, nsplit, K, kpad, k0, row0, nrows, z, own, crp, out, ld, pack, cap_rows, kend, rng, row_id0, zero);
}

// the geometry the launches of one pass share; false: not for this kernel
static bool tail_rows_geometry(const TailPlan &tp, int num_cus, int nfeat, int nsplit, uint64_t nrows, uint32_t &cap_rows, size_t &lds,
                               unsigned &grid) {
  if (!tp.ok || (tp.pack_rows > 0 && tp.pack == nullptr)) return false;
  // the slot: up to 200 table rows (52 KiB) in what 64 KiB leave beside the second phase's block (64 floats a nich
  // feature) -- two workgroups a CU; it must hold the largest table
  const size_t nich_bytes = (size_t)(nfeat - nsplit) * 64 * sizeof(float);
  if (nich_bytes + (size_t)std::max<uint32_t>(1u, tp.max_rows) * kTailStride * sizeof(float) > 64u * 1024u) return false;
  const uint32_t fit = (uint32_t)((64u * 1024u - nich_bytes) / (kTailStride * sizeof(float)));
  cap_rows = std::max<uint32_t>(1u, std::min<uint32_t>(tp.pack_rows, std::min<uint32_t>(200u, fit)));
  lds = (size_t)cap_rows * kTailStride * sizeof(float) + nich_bytes;
  const uint64_t rows_wg = (uint64_t)kTailRowsWaves * 64;
  const uint64_t tchunks = (nrows + rows_wg - 1) / rows_wg;
  grid = (unsigned)std::min<uint64_t>(tchunks ? tchunks : 1, (uint64_t)num_cus * 2);
  return true;
}

// a state of at most 64 groups: leave-one-out + prior scores and the draw in one launch of the lane <-> row kernel (own from
// launch_loo_own; zero: the additive tables a sweep step wants emptied).  -> 0: launched; 1: not for this kernel
int launch_sweep_rows(hipStream_t stream, int num_cus, const TailPlan &tp, const FeatDesc *feats_dev, int nfeat, int nsplit, uint32_t K,
                      uint32_t kpad, uint64_t row0, uint64_t nrows, uint64_t row_id0, int32_t *z, const float *own, const float *crp,
                      const uint64_t *rng, ZeroSpans zero) {
  uint32_t cap_rows = 0;
  size_t lds = 0;
  unsigned grid = 0;
  if (K > 64 || crp == nullptr || !tail_rows_geometry(tp, num_cus, nfeat, nsplit, nrows, cap_rows, lds, grid)) return 1;
  if (nsplit > 0) hipLaunchKernelGGL(k_tail_pack, dim3((unsigned)nsplit), dim3(256), 0, stream, feats_dev, kpad, 0u, tp.pack);
  const uint32_t tgp = (K + 15u) / 16u * 16u;
#define MSC_SWEEP_ROWS(T) launch_tail_rows_t<T, false, true, true>(stream, grid, lds, tp.dm ? 2 : tp.masked_nich ? 1 : 0, feats_dev, nfeat, nsplit, K, kpad, 0u, row0, nrows, z, own, crp, nullptr, 0, tp.pack, cap_rows, K, rng, row_id0, zero)
  if (tgp == 16) MSC_SWEEP_ROWS(16);
  else if (tgp == 32)        // (two sums per group here: the one-sum instantiation of 32 spills -- the register allocator's quirk)
    launch_tail_rows_t<32, true, true, true>(stream, grid, lds, tp.dm ? 2 : tp.masked_nich ? 1 : 0, feats_dev, nfeat, nsplit, K, kpad, 0u, row0, nrows, z, own, crp, nullptr, 0, tp.pack, cap_rows, K, rng, row_id0, zero);
  else if (tgp == 48) MSC_SWEEP_ROWS(48);
  else MSC_SWEEP_ROWS(64);
#undef MSC_SWEEP_ROWS
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// -> 0: launched; 1: the groups are not for this kernel (the caller's tile kernels take them)
int launch_score_tail(hipStream_t stream, int num_cus, const TailPlan &tp, const FeatDesc *feats_dev, int nfeat, int nsplit, uint32_t K,
                      uint32_t kpad, uint32_t k0, uint64_t row0, uint64_t nrows, const int32_t *z, const float *own, const float *crp,
                      float *out, uint64_t ld) {
  // up to kTailMaxGroups groups, at most 48 (two sums per group: 32 or 48 at four waves a SIMD) or 64 a launch -- a lane's
  // sums are its registers; beyond, the tile kernels keep the tile: three launches cost what a tile pass costs
  uint32_t cap_rows = 0;
  size_t lds = 0;
  unsigned grid = 0;
  if (K <= k0 || K - k0 > kTailMaxGroups || !tail_rows_geometry(tp, num_cus, nfeat, nsplit, nrows, cap_rows, lds, grid)) return 1;
  if (nrows == 0) return 0;
  // launches of equal width; with two sums a group (exact) at most 32 groups a launch -- the 48-group instantiation holds 96
  // sums a lane and costs a light plan 2.8x a 32-group launch (8 bb columns, 1M rows: K = 48 0.22 ms, K = 64 = 32 + 32 0.16;
  // C3's 64 columns: 0.49 against 0.29 + 0.19) --, with one sum up to 64
  const uint32_t groups = K - k0, widest = tp.exact ? 32u : 64u, nblk = (groups + widest - 1) / widest;
  const uint32_t blk = ((groups + nblk - 1) / nblk + 15u) / 16u * 16u;
  for (uint32_t kb = k0; kb < K; kb += blk) {
    const uint32_t kend = std::min<uint32_t>(K, kb + blk), tgp = (kend - kb + 15u) / 16u * 16u;
    // (the packed tables are this launch's: the stream orders the next block's k_tail_pack behind it)
    if (nsplit > 0) hipLaunchKernelGGL(k_tail_pack, dim3((unsigned)nsplit), dim3(256), 0, stream, feats_dev, kpad, kb, tp.pack);
#define MSC_TAIL_ROWS(T, S, E) launch_tail_rows_t<T, S, false, E>(stream, grid, lds, tp.dm ? 2 : tp.masked_nich ? 1 : 0, feats_dev, nfeat, nsplit, K, kpad, kb, row0, nrows, z, own, crp, out, ld, tp.pack, cap_rows, kend)
    if (tp.exact) {
      if (tgp == 16) MSC_TAIL_ROWS(16, true, false);
      else if (tgp == 32) MSC_TAIL_ROWS(32, true, false);
      else MSC_TAIL_ROWS(48, true, false);
    } else {
      if (tgp == 16) MSC_TAIL_ROWS(16, false, true);
      else if (tgp == 32) MSC_TAIL_ROWS(32, true, true);    // (as above: the one-sum instantiation of 32 spills)
      else if (tgp == 48) MSC_TAIL_ROWS(48, false, true);
      else MSC_TAIL_ROWS(64, false, true);
    }
#undef MSC_TAIL_ROWS
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <bool LOO, bool CRP>
static void launch_score_t(hipStream_t stream, int num_cus, int path, const TailPlan &narrow_tail, int nich1_shape, const FeatDesc *feats_dev,
                           int nfeat, int nsplit, uint32_t K, uint32_t kpad, uint64_t row0, uint64_t nrows,
                           const int32_t *z, const float *own, const float *crp, float *out, uint64_t ld) {
  const uint32_t ktiles = kpad / kGroupTile;
  if (path == MSC_PATH_NICH1) {
    // A wave visits `visits` blocks of Q consecutive rows, nslots * Q rows apart; waves are numbered
    // tile-fastest, so the waves resident at one moment write `visits` dense fronts that sweep the matrix.
    // Which (Q, visits) suits the HBM write stream depends on the box and on where the score buffer landed
    // (profiles/r01_nich1_variants.txt); abi.cpp run_score times the shapes of kNich1Shapes at the first large
    // pass of a context and passes the winner's index here (0 = the default).
    // (nich1_shape: the shape's index; bit 8: plain stores instead of non-temporal ones -- abi.cpp nich1_shape_for)
    const bool plain_stores = (nich1_shape & 0x100) != 0;
    nich1_shape &= 0xff;
    const Nich1Shape sh = kNich1Shapes[nich1_shape >= 0 && nich1_shape < kNich1NumShapes ? nich1_shape : 0];
    const uint64_t nvisits_all = (nrows + sh.q - 1) / sh.q;
    const uint64_t max_slots = ((uint64_t)1 << 32) / ktiles;          // keeps grid.x below 2^30 workgroups
    uint64_t visits = sh.visits;
    while ((nvisits_all + visits - 1) / visits > max_slots && visits * sh.q < 64) visits++;
    uint64_t nslots = (nvisits_all + visits - 1) / visits;
    if (nslots == 0) nslots = 1;
    const uint64_t gx = (nslots * ktiles + 3) / 4;
    if (plain_stores)
      hipLaunchKernelGGL((k_score_nich1<LOO, CRP, 4, false>), (note_kernel(0, "k_score_nich1<%s, %s, 4, false>", tf(LOO), tf(CRP)), dim3((unsigned)gx)), dim3(256), 0, stream,
                         feats_dev, K, kpad, row0, nrows, nslots, z, own, crp, out, ld);
    else
      hipLaunchKernelGGL((k_score_nich1<LOO, CRP, 4, true>), (note_kernel(0, "k_score_nich1<%s, %s, 4, true>", tf(LOO), tf(CRP)), dim3((unsigned)gx)), dim3(256), 0, stream,
                         feats_dev, K, kpad, row0, nrows, nslots, z, own, crp, out, ld);
  } else {
    // one workgroup per CU (2 x 64 KiB of LDS), 128 rows per workgroup, two tilings:
    // 16 waves x 8 rows (4 waves/SIMD, default) or 8 waves x 16 rows (2 waves/SIMD)
    // states with a dm feature: 8 waves x 8 rows (64 rows per workgroup, 256-register budget for the hi/lo sums)
    // few rows: a chunk is a serial chain (feature after feature, the code fetched once), so what counts is that
    // the chunks spread over the chip in ONE round: 4 or 2 rows per wave while that still fits
    const bool dm = path == MSC_PATH_TILE_DM;
    const uint64_t round = (uint64_t)num_cus / ktiles;
    constexpr uint64_t rounds4 = 1;
    // (4 rows per wave only while the 64-row workgroups themselves fit one round: 20000 rows made 313 of them -- two
    // rounds, 0.113 ms -- where 157 workgroups of 128 rows take one)
    const bool small4 = !dm && (nrows + 63) / 64 <= round * rounds4, small2 = small4 && (nrows + 31) / 32 <= round;
    const uint64_t rows_per_wg = dm ? 64 : small2 ? 32 : small4 ? 64 : 128;
    const uint64_t nchunks = (nrows + rows_per_wg - 1) / rows_per_wg;
    uint64_t gx = nchunks;
    const uint64_t cap = (uint64_t)num_cus * 4;
    if (gx > cap) gx = cap;
    if (gx == 0) gx = 1;
    // the last tile alone on the lane <-> row kernel when it is partly filled and the plan allows (flag from abi.cpp) and
    // the rows are many: its bits are the tile kernels', so the choice is free; few rows (a per-entity call's one) are
    // better off with lanes as groups.  A state of at most kTailMaxGroups groups is all "last tile".
    // (it fills the chip from ~260k rows on, 512 a workgroup, where the tile kernels run in rounds of 128 rows a CU:
    // whichever the cost model of launchers.hpp prices lower for these rows -- the last tile's share of the tile pass
    // against the launches of the lane <-> row kernel)
    const char *forced = std::getenv("MSC_TAIL_MIN_ROWS");          // (tests: the kernel on a few thousand rows)
    const uint64_t c128 = (nrows + 127) / 128;
    // (a state of at most 128 groups on the role-split kernels: PAIR mode, 256 rows a workgroup at about the price of 128)
    const bool pair = pair_mode_ok(path, K, small4);
    const double tile_us = (pair ? kPairTileShare : 1.0) *
                           (tile_rounds_us(c128 * ktiles, num_cus, false, narrow_tail.cost) - (ktiles > 1 ? tile_rounds_us(c128 * (ktiles - 1), num_cus, false, narrow_tail.cost) : 0.0));
    const bool many_rows = forced ? nrows >= (uint64_t)std::atoll(forced)
                                  : nrows >= kTailMinRows && tail_rows_us(K - (ktiles - 1) * kGroupTile, true, nrows, num_cus, narrow_tail.cost) < tile_us;
    // a LAST tile of 65 .. 128 groups beyond full ones on the role-split kernels: PAIR mode at that tile (round 5) -- about
    // 0.62 of a full tile's price, the tile kernels' own bits, where the lane <-> row kernel takes two or three launches
    const uint32_t last_groups = K - (ktiles - 1) * kGroupTile;
    bool tail = false;
    if (ktiles > 1 && path == MSC_PATH_TILE_ROLES && !small4 && last_groups > 64 && last_groups <= 128 && pair_mode_ok(path, last_groups, false)) {
      hipLaunchKernelGGL((k_score_tile_roles<LOO, CRP, true>), (note_kernel(0, "k_score_tile_roles<%s, %s, true>", tf(LOO), tf(CRP)), dim3((unsigned)std::min<uint64_t>((nrows + 255) / 256, cap), 1)), dim3(1024), 0, stream,
                         feats_dev, nfeat, nsplit, K, kpad, row0, nrows, z, own, crp, out + (size_t)(ktiles - 1) * kGroupTile, ld, ktiles - 1);
      tail = true;
    }
    if (!tail)
      tail = many_rows && launch_score_tail(stream, num_cus, narrow_tail, feats_dev, nfeat, nsplit, K, kpad,
                                            (ktiles - 1) * kGroupTile, row0, nrows, z, own, crp, out, ld) == 0;
    if (tail && ktiles == 1) return;
    const dim3 grid((unsigned)gx, tail ? ktiles - 1 : ktiles);
    if (path == MSC_PATH_TILE_DM)
      hipLaunchKernelGGL((k_score_tile<8, 8, LOO, CRP, true>), (note_kernel(0, "k_score_tile<8, 8, %s, %s, true>", tf(LOO), tf(CRP)), grid), dim3(512), 0, stream, feats_dev, nfeat, nsplit, K, kpad, row0,
                         nrows, z, own, crp, out, ld);
    else if (path == MSC_PATH_LOOKUPS && pair)
      hipLaunchKernelGGL((k_score_lookups<LOO, CRP, true>), (note_kernel(0, "k_score_lookups<%s, %s, true>", tf(LOO), tf(CRP)), dim3((unsigned)std::min<uint64_t>((nrows + 511) / 512, cap), 1)), dim3(1024), 0, stream,
                         feats_dev, nfeat, K, kpad, row0, nrows, z, own, crp, out, ld);
    else if (path == MSC_PATH_LOOKUPS && !small4)
      hipLaunchKernelGGL((k_score_lookups<LOO, CRP, false>), (note_kernel(0, "k_score_lookups<%s, %s, false>", tf(LOO), tf(CRP)), dim3((unsigned)std::min<uint64_t>((nrows + 255) / 256, cap), grid.y)), dim3(1024), 0, stream,
                         feats_dev, nfeat, K, kpad, row0, nrows, z, own, crp, out, ld);
    else if (pair && path == MSC_PATH_NICH_PACK)
    {
      const dim3 g((unsigned)std::min<uint64_t>((nrows + 32 * kNichPackWaves - 1) / (32 * kNichPackWaves), cap * (16 / kNichPackWaves)), 1);
      if (nsplit > 0)
        hipLaunchKernelGGL((k_score_nich_pack<LOO, CRP, true, true>), (note_kernel(0, "k_score_nich_pack<%s, %s, true, true>", tf(LOO), tf(CRP)), g), dim3(kNichPackWaves * 64), 0, stream, feats_dev, nsplit, K, kpad, row0, nrows, z, own, crp, out, ld);
      else
        hipLaunchKernelGGL((k_score_nich_pack<LOO, CRP, true, false>), (note_kernel(0, "k_score_nich_pack<%s, %s, true, false>", tf(LOO), tf(CRP)), g), dim3(kNichPackWaves * 64), 0, stream, feats_dev, nsplit, K, kpad, row0, nrows, z, own, crp, out, ld);
    }
    else if (path == MSC_PATH_NICH_PACK && !small4)
    {
      const dim3 g((unsigned)std::min<uint64_t>((nrows + 16 * kNichPackWaves - 1) / (16 * kNichPackWaves), cap * (16 / kNichPackWaves)), grid.y);
      if (nsplit > 0)
        hipLaunchKernelGGL((k_score_nich_pack<LOO, CRP, false, true>), (note_kernel(0, "k_score_nich_pack<%s, %s, false, true>", tf(LOO), tf(CRP)), g), dim3(kNichPackWaves * 64), 0, stream, feats_dev, nsplit, K, kpad, row0, nrows, z, own, crp, out, ld);
      else
        hipLaunchKernelGGL((k_score_nich_pack<LOO, CRP, false, false>), (note_kernel(0, "k_score_nich_pack<%s, %s, false, false>", tf(LOO), tf(CRP)), g), dim3(kNichPackWaves * 64), 0, stream, feats_dev, nsplit, K, kpad, row0, nrows, z, own, crp, out, ld);
    }
    else if (pair)
      hipLaunchKernelGGL((k_score_tile_roles<LOO, CRP, true>), (note_kernel(0, "k_score_tile_roles<%s, %s, true>", tf(LOO), tf(CRP)), dim3((unsigned)std::min<uint64_t>((nrows + 255) / 256, cap), 1)), dim3(1024), 0, stream,
                         feats_dev, nfeat, nsplit, K, kpad, row0, nrows, z, own, crp, out, ld, 0u);
    else if (!small4 && path == MSC_PATH_TILE_ROLES && tile_roles_enabled())
      hipLaunchKernelGGL((k_score_tile_roles<LOO, CRP>), (note_kernel(0, "k_score_tile_roles<%s, %s, false>", tf(LOO), tf(CRP)), grid), dim3(1024), 0, stream, feats_dev, nfeat, nsplit, K, kpad, row0,
                         nrows, z, own, crp, out, ld, 0u);
    else if (small2)
      hipLaunchKernelGGL((k_score_tile<2, 16, LOO, CRP, false>), (note_kernel(0, "k_score_tile<2, 16, %s, %s, false>", tf(LOO), tf(CRP)), grid), dim3(1024), 0, stream, feats_dev, nfeat, nsplit, K, kpad, row0,
                         nrows, z, own, crp, out, ld);
    else if (small4)
      hipLaunchKernelGGL((k_score_tile<4, 16, LOO, CRP, false>), (note_kernel(0, "k_score_tile<4, 16, %s, %s, false>", tf(LOO), tf(CRP)), grid), dim3(1024), 0, stream, feats_dev, nfeat, nsplit, K, kpad, row0,
                         nrows, z, own, crp, out, ld);
    else
      hipLaunchKernelGGL((k_score_tile<8, 16, LOO, CRP, false>), (note_kernel(0, "k_score_tile<8, 16, %s, %s, false>", tf(LOO), tf(CRP)), grid), dim3(1024), 0, stream, feats_dev, nfeat, nsplit, K, kpad, row0,
                         nrows, z, own, crp, out, ld);
  }
}

// The groups 256 .. K - 1 (65 .. 128 of them) of a role-split plan, leave-one-out value and prior included, into
// tail[row * ld + (group - 256)]: the role-split kernel in PAIR mode at k-tile 1 (round 5) -- one pass at about 0.62 of a
// full tile's price where the lane <-> row kernel took three launches of up to 48 groups.  The tile kernels' sums.
int launch_score_pair_tail(hipStream_t stream, int num_cus, const FeatDesc *feats_dev, int nfeat, int nsplit, uint32_t K, uint32_t kpad,
                           uint64_t row0, uint64_t nrows, const int32_t *z, const float *own, const float *crp, float *tail, uint64_t ld) {
  if (K <= (uint32_t)kGroupTile + 64u || K > (uint32_t)kGroupTile + 128u || z == nullptr || crp == nullptr) return -2;
  const uint64_t cap = (uint64_t)num_cus * 4;
  hipLaunchKernelGGL((k_score_tile_roles<true, true, true>), (note_kernel(0, "k_score_tile_roles<true, true, true>"), dim3((unsigned)std::max<uint64_t>(1, std::min<uint64_t>((nrows + 255) / 256, cap)), 1)),
                     dim3(1024), 0, stream, feats_dev, nfeat, nsplit, K, kpad, row0, nrows, z, own, crp, tail, ld, 1u);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// own: per-row leave-one-out values from launch_loo_own (required when z != null)
int launch_score(hipStream_t stream, int num_cus, int path, const TailPlan &narrow_tail, int nich1_shape, const FeatDesc *feats_dev, int nfeat, int nsplit,
                 uint32_t K, uint32_t kpad, uint64_t row0, uint64_t nrows, const int32_t *z,
                 const float *own, const float *crp, float *out, uint64_t ld) {
  const bool loo = z != nullptr, pri = crp != nullptr;
  if (loo && pri) launch_score_t<true, true>(stream, num_cus, path, narrow_tail, nich1_shape, feats_dev, nfeat, nsplit, K, kpad, row0, nrows, z, own, crp, out, ld);
  else if (loo) launch_score_t<true, false>(stream, num_cus, path, narrow_tail, nich1_shape, feats_dev, nfeat, nsplit, K, kpad, row0, nrows, z, own, crp, out, ld);
  else if (pri) launch_score_t<false, true>(stream, num_cus, path, narrow_tail, nich1_shape, feats_dev, nfeat, nsplit, K, kpad, row0, nrows, z, own, crp, out, ld);
  else launch_score_t<false, false>(stream, num_cus, path, narrow_tail, nich1_shape, feats_dev, nfeat, nsplit, K, kpad, row0, nrows, z, own, crp, out, ld);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace msc
