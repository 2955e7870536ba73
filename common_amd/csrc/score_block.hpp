// score_block.hpp -- the workgroup-tile scorer shared by the batched score kernel and the
// fused sweep kernel.
//
// A workgroup owns a block of consecutive rows (128) and one tile of 256 groups.  The host packs
// consecutive features into groups whose per-group tables for the tile (bb: 2 rows, nich: 6
// constants, dd: one row per category, gp / bnb: one row per count) fit one 128 KiB LDS slot
// together; the workgroup copies a group with global_load_lds, synchronises once, and every wave
// then adds the group's features for its rows into registers on its own: conflict-free
// ds_read_b128 (lane <-> 4 groups, 1 KiB table rows), runs of plain lookup features in a loop of
// their own (a software pipeline: descriptor head two features ahead, value one ahead).  What bounds it: the lookup
// phase the CU's LDS port (a wave's ds_read_b128 is 1 KiB = 8 cycles), the nich phase the vector ALUs, one after the
// other; a lane <-> 1 group tiling (512 rows per workgroup, 4x fewer table bytes) was measured
// 2x slower (more instructions per evaluation).
// Nothing in here evaluates a transcendental in double: leave-one-out terms arrive precomputed
// per row (k_loo_own), counts beyond the tables are patched afterwards (k_gp_large_fix).
// Table rows beyond the 64 a feature may stage are gathered from L2; Dirichlet-Multinomial tables
// always are.
#pragma once

#include "device_error.hpp"
#include "family_math.hpp"
#include "launchers.hpp"

namespace msc {

MSC_DEV float lane_bcast(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
MSC_DEV int lane_bcast(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

MSC_DEV float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Loads through pointers that come out of a FeatDesc are FLAT to the compiler (it cannot know the address space of a
// pointer it loaded): a 64-bit vector address per load, and a wait on the LDS counter as well.  The tables and columns
// are global memory: said so, a load is the descriptor's scalar base + one 32-bit lane offset shared by all of them.
typedef const __attribute__((address_space(1))) float *gfloat_p;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) f32x2 *gfloat2_p;
typedef const __attribute__((address_space(1))) f32x4 *gfloat4_p;
MSC_DEV gfloat_p as_global(const void *p) { return (gfloat_p)(const float *)p; }
// ... and the descriptors themselves are read-only for the kernel's lifetime and indexed by wave-uniform numbers: through
// the constant address space their fields are SCALAR loads whatever the compiler can prove about the index (a feature
// index that came out of a descriptor -- the end of a nich block -- otherwise turns every later descriptor access into a
// per-lane load with a wait behind it)
typedef const __attribute__((address_space(4))) FeatDesc *scalar_feats;
MSC_DEV scalar_feats as_scalar(const FeatDesc *f) { return (scalar_feats)f; }
MSC_DEV int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// a load at (uniform base + uniform element offset) + this lane's BYTE offset: written so that the lane's part is one
// 32-bit register zero-extended (the instruction's own addressing: scalar base, 32-bit vector offset) -- an element
// offset would be a shift of the extension, i.e. 64-bit vector arithmetic and a register pair per address
typedef const __attribute__((address_space(1))) char *gchar_p;
MSC_DEV gfloat_p at_lane(gfloat_p base, size_t uniform_elems, uint32_t lane_bytes) {
  return (gfloat_p)((gchar_p)(base + uniform_elems) + (size_t)lane_bytes);
}
MSC_DEV float gld1(gfloat_p p) { return *p; }
MSC_DEV float2 gld2(gfloat_p p) { const f32x2 v = *(gfloat2_p)p; return make_float2(v.x, v.y); }
MSC_DEV float4 gld4(gfloat_p p) { const f32x4 v = *(gfloat4_p)p; return make_float4(v.x, v.y, v.z, v.w); }


// 16-byte global -> LDS copy that bypasses the VGPRs (global_load_lds_dwordx4): every lane
// supplies its own source address, the destination is lds_wave_base + lane * 16.
MSC_DEV void glds16(const float *gsrc, float4 *lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                   (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// CRP prior for one row of a k-tile.  e_row = log(alpha / n_empty') for this row (n_empty'
// counts the row's own group if removing it empties it).
MSC_DEV float4 crp_prior4(float4 logcnt, float e_row) {
  float4 p;
  p.x = __builtin_isinf(logcnt.x) ? e_row : logcnt.x;
  p.y = __builtin_isinf(logcnt.y) ? e_row : logcnt.y;
  p.z = __builtin_isinf(logcnt.z) ? e_row : logcnt.z;
  p.w = __builtin_isinf(logcnt.w) ? e_row : logcnt.w;
  return p;
}
// the low halves of the same terms (kernels_score.hip crp_prepare_block): where hi is -inf the group is empty
MSC_DEV float4 crp_prior4_lo(float4 logcnt_hi, float4 logcnt_lo, float e_row_lo) {
  float4 p;
  p.x = __builtin_isinf(logcnt_hi.x) ? e_row_lo : logcnt_lo.x;
  p.y = __builtin_isinf(logcnt_hi.y) ? e_row_lo : logcnt_lo.y;
  p.z = __builtin_isinf(logcnt_hi.z) ? e_row_lo : logcnt_lo.z;
  p.w = __builtin_isinf(logcnt_hi.w) ? e_row_lo : logcnt_lo.w;
  return p;
}
// (g is wave-uniform: one vector compare finds the lane, the component is picked by scalar conditions -- keeps
// kb + 1 .. kb + 3 out of the registers of kernels that have none to spare)
MSC_DEV void replace_own(float4 &s, uint32_t kb, int g, float v) {
  const bool mine = (int)kb == (g & ~3);
  const int c = g & 3;
  s.x = (mine && c == 0) ? v : s.x;
  s.y = (mine && c == 1) ? v : s.y;
  s.z = (mine && c == 2) ? v : s.z;
  s.w = (mine && c == 3) ? v : s.w;
}
MSC_DEV void add4(float4 &a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }

// ---------------------------------------------------------------------------
// PAIR mode of the role-split kernels (round 4): a state of at most 128 groups.  A lane carries TWO groups (2 lane,
// 2 lane + 1) and a float4 of sums holds TWO rows of the wave -- (x, y): row 2 r, (z, w): row 2 r + 1 -- so the same 64
// registers of sums cover 32 rows a wave and 256 a workgroup: per row half the LDS bytes (a table row is 512 B, a lookup
// one ds_read_b64), half the additions, one block part instead of two, the table copies spread over twice the rows.
// Every (row, group) sum is formed by the same operations in the same order as in the other tile kernels: same bits.
// ---------------------------------------------------------------------------
MSC_DEV float4 pair_dup(float2 v) { return make_float4(v.x, v.y, v.x, v.y); }
MSC_DEV float4 crp_prior_pair(float2 logcnt, float e_a, float e_b) {
  float4 p;
  p.x = __builtin_isinf(logcnt.x) ? e_a : logcnt.x;
  p.y = __builtin_isinf(logcnt.y) ? e_a : logcnt.y;
  p.z = __builtin_isinf(logcnt.x) ? e_b : logcnt.x;
  p.w = __builtin_isinf(logcnt.y) ? e_b : logcnt.y;
  return p;
}
MSC_DEV float4 crp_prior_pair_lo(float2 hi, float2 lo, float e_a, float e_b) {
  float4 p;
  p.x = __builtin_isinf(hi.x) ? e_a : lo.x;
  p.y = __builtin_isinf(hi.y) ? e_a : lo.y;
  p.z = __builtin_isinf(hi.x) ? e_b : lo.x;
  p.w = __builtin_isinf(hi.y) ? e_b : lo.y;
  return p;
}
// (ga / gb wave-uniform, -1: none; the lane that holds group g is g >> 1, the component g & 1 of the row's half)
MSC_DEV void replace_own_pair(float4 &s, int lane, int ga, float va, int gb, float vb) {
  const bool la = ga >= 0 && lane == (ga >> 1), lb = gb >= 0 && lane == (gb >> 1);
  s.x = (la && (ga & 1) == 0) ? va : s.x;
  s.y = (la && (ga & 1) == 1) ? va : s.y;
  s.z = (lb && (gb & 1) == 0) ? vb : s.z;
  s.w = (lb && (gb & 1) == 1) ? vb : s.w;
}

template <bool NT = true>
MSC_DEV void store_row(float *__restrict__ out, uint64_t ld, uint64_t row, uint32_t kb, uint32_t K,
                       float4 s, bool vec_ok) {
  float *p = out + row * ld + kb;
  if (vec_ok && kb + 3 < K) {
    const f32x4 v = {s.x, s.y, s.z, s.w};
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(p));
    else *reinterpret_cast<f32x4 *>(p) = v;
  } else if (kb < K) {
    const uint32_t rem = K - kb;
    // (rows that start on 8 but not on 16 bytes -- a caller's [N, K] matrix with K = 350 --: two 8-byte stores, not four
    // 4-byte ones: the scoring pass of such a matrix ran at twice the time of its neighbours K = 320 / 384)
    typedef float f32x2s __attribute__((ext_vector_type(2)));
    if (rem >= 4 && (ld & 1) == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0) {
      const f32x2s a = {s.x, s.y}, b = {s.z, s.w};
      *reinterpret_cast<f32x2s *>(p) = a;
      *reinterpret_cast<f32x2s *>(p + 2) = b;
      return;
    }
    p[0] = s.x;
    if (rem > 1) p[1] = s.y;
    if (rem > 2) p[2] = s.z;
    if (rem > 3) p[3] = s.w;
  }
}

// one row's half of a PAIR of sums: two floats a lane, 512 B a wave instruction
MSC_DEV void store_half_row(float *__restrict__ out, uint64_t ld, uint64_t row, int lane, uint32_t K, float a, float b, bool vec_ok) {
  const uint32_t k0 = (uint32_t)lane * 2u;
  float *p = out + row * ld + k0;
  typedef float f32x2s __attribute__((ext_vector_type(2)));
  if (vec_ok && k0 + 1 < K) {
    const f32x2s v = {a, b};
    *reinterpret_cast<f32x2s *>(p) = v;
  } else if (k0 < K) {
    p[0] = a;
    if (k0 + 1 < K) p[1] = b;
  }
}

// The feature's value of this lane's row, as raw 32 bits (reinterpreted per family).  A dm feature is
// scored in dim + 1 *stages* (one count lookup per category, one for the row total; family_math.hpp),
// `sub` selects the stage.
template <bool DM>
MSC_DEV uint32_t load_raw_value(const FeatDesc &fd, uint32_t sub, uint64_t row, bool has_row) {
  if (!has_row || fd.col == nullptr) return 0u;
  if (fd.family == MSC_BB || fd.family == MSC_BBNC) return (uint32_t)(reinterpret_cast<const uint8_t *>(fd.col)[row] != 0);
  if (fd.family == MSC_NIW || fd.family == MSC_NOOP) return 0u;
  if (fd.family == MSC_DM) {
    if constexpr (!DM) return 0u;
    const uint32_t tot = fd.dm_tot[row];
    if (tot >= kGpMaxTable) return 0xffffffffu;        // the whole row goes to the large-count kernel
    return sub < fd.dim ? reinterpret_cast<const uint32_t *>(fd.col)[row * fd.dim + sub] : tot;
  }
  return reinterpret_cast<const uint32_t *>(fd.col)[row];
}

// is this lane's row masked for the feature?  (one mask byte per element, runtime_type.hpp:131;
// a masked value takes no part in scoring or in the suff-stats, as the reference's callers skip it)
template <bool DM = true>
MSC_DEV bool load_masked(const FeatDesc &fd, uint64_t row, bool has_row) {
  if (!has_row || fd.mask == nullptr) return false;
  if (fd.family != MSC_NIW && !(DM && fd.family == MSC_DM)) return fd.mask[row] != 0;
  bool m = false;
  for (uint32_t e = 0; e < fd.dim; e++) m |= fd.mask[row * fd.dim + e] != 0;
  return m;
}

// one stage's contribution to the R rows of this wave (tables already in `buf`)
template <int R, bool MASKED>
MSC_DEV void add_feature(const FeatDesc &fd, const float4 *__restrict__ buf, const uint32_t nrows_lds, uint32_t kpad,
                         uint32_t kb, int lane, uint32_t raw, unsigned long long mbits, float4 (&acc)[R]) {
  // Lookup families, common case: no masked row and every row's table entry is in the LDS block
  // (wave-uniform test).  Branch-free, so the R reads go out back to back: one ds_read_b128 and
  // four adds per row.
  if (!MASKED && fd.family != MSC_NICH) {
    uint32_t idx = raw;
    if (fd.family == MSC_DD) {
      const int v = (int)raw;
      idx = (uint32_t)(v < 0 ? 0 : (v >= (int)fd.dim ? (int)fd.dim - 1 : v));   // keep the gather in bounds
    }
    const bool lookup = fd.family == MSC_BB || fd.family == MSC_BBNC || fd.family == MSC_DD || is_count_family(fd.family);
    if (lookup && __builtin_amdgcn_ballot_w64(idx >= nrows_lds) == 0ull) {
#pragma unroll
      for (int r0 = 0; r0 < R; r0 += 4) {
        float4 t[4];
#pragma unroll
        for (int j = 0; j < 4; j++) t[j] = buf[(uint32_t)lane_bcast((int)idx, r0 + j) * 64 + lane];
#pragma unroll
        for (int j = 0; j < 4; j++) add4(acc[r0 + j], t[j]);
      }
      return;
    }
  }
  switch (fd.family) {
    case MSC_BBNC:
    case MSC_BB: {
      const float4 s0 = buf[lane], s1 = buf[64 + lane];
#pragma unroll
      for (int r = 0; r < R; r++) {
        if (MASKED && ((mbits >> r) & 1ull)) continue;
        const bool vr = lane_bcast((int)raw, r) != 0;
        acc[r].x += vr ? s1.x : s0.x;
        acc[r].y += vr ? s1.y : s0.y;
        acc[r].z += vr ? s1.z : s0.z;
        acc[r].w += vr ? s1.w : s0.w;
      }
    } break;
    case MSC_DD: {
      int v = (int)raw;
      v = v < 0 ? 0 : (v >= (int)fd.dim ? (int)fd.dim - 1 : v);       // keep the gather in bounds
#pragma unroll
      for (int r = 0; r < R; r++) {
        if (MASKED && ((mbits >> r) & 1ull)) continue;
        const uint32_t vr = (uint32_t)lane_bcast(v, r);
        add4(acc[r], vr < nrows_lds ? buf[vr * 64 + lane] : ld4(fd.tab + (size_t)vr * kpad + kb));
      }
    } break;
    case MSC_BNB:
    case MSC_GP: {
#pragma unroll
      for (int r = 0; r < R; r++) {
        if (MASKED && ((mbits >> r) & 1ull)) continue;
        const uint32_t vr = (uint32_t)lane_bcast((int)raw, r);
        if (vr < nrows_lds) add4(acc[r], buf[vr * 64 + lane]);
        else if (vr < fd.vcap) add4(acc[r], ld4(fd.tab + (size_t)(GP_T0 + vr) * kpad + kb));
        // counts beyond the table contribute through k_gp_large_fix
      }
    } break;
    case MSC_NICH: {
      const float4 mh = buf[NICH_MU_HI * 64 + lane], ml = buf[NICH_MU_LO * 64 + lane],
                   c0 = buf[NICH_C0 * 64 + lane], c1l = buf[NICH_C1LN2 * 64 + lane],
                   c1 = buf[NICH_C1 * 64 + lane], c2 = buf[NICH_C2 * 64 + lane];
      const float xv = __uint_as_float(raw);
#pragma unroll
      for (int r = 0; r < R; r++) {
        if (MASKED && ((mbits >> r) & 1ull)) continue;
        const float x = lane_bcast(xv, r);
        acc[r].x += nich_eval(x, mh.x, ml.x, c0.x, c1l.x, c1.x, c2.x);
        acc[r].y += nich_eval(x, mh.y, ml.y, c0.y, c1l.y, c1.y, c2.y);
        acc[r].z += nich_eval(x, mh.z, ml.z, c0.z, c1l.z, c1.z, c2.z);
        acc[r].w += nich_eval(x, mh.w, ml.w, c0.w, c1l.w, c1.w, c2.w);
      }
    } break;
    default: break;   // noop contributes 0 (models/noop.hpp:17); niw has its own MFMA pass
  }
}

// ---------------------------------------------------------------------------
// One Dirichlet-Multinomial feature: dim + 1 count lookups per row (one per category, one for the row
// total; family_math.hpp), each a (hi, lo) pair of table rows, the exact hi sums and the lo sums kept
// apart.  The tables are read straight from L2: a stage's block has 2 x (largest count + 1) rows and a
// workgroup's rows use each of them about once, so staging it in LDS moves more bytes than the
// lookups themselves (measured: 3.46 ms staged, 2.72 ms direct for 4 x dm(4) on 1M rows).  Entry 0 of
// every table is exactly zero (an absent category adds nothing) and is skipped.
// dm_meta per stage: {first table row, entries in the table}
// ---------------------------------------------------------------------------
constexpr int kDmBatch = 8;       // stages whose row values are fetched together
template <int R>
MSC_DEV void score_dm_feature(const FeatDesc &fd, uint32_t kpad, uint32_t kb, int lane, uint64_t myrow, bool has_row,
                              float4 (&acc)[R]) {
  const uint32_t nst = fd.dim + 1;
  const unsigned long long mbits = fd.mask == nullptr ? 0ull : __builtin_amdgcn_ballot_w64(load_masked<true>(fd, myrow, has_row));
  float4 hi[R], lo[R];
#pragma unroll
  for (int r = 0; r < R; r++) hi[r] = lo[r] = make_float4(0, 0, 0, 0);
  // this lane's row: its total decides whether the tables apply at all (else k_gp_large_fix scores the row)
  const uint32_t *xrow = reinterpret_cast<const uint32_t *>(fd.col) + myrow * fd.dim;
  const uint32_t tot = (has_row && fd.col != nullptr) ? fd.dm_tot[myrow] : 0u;
  const bool tabled = has_row && fd.col != nullptr && tot < kGpMaxTable;
  for (uint32_t s0 = 0; s0 < nst; s0 += kDmBatch) {
    uint32_t vals[kDmBatch];                            // one memory latency per batch of stages, not per stage
#pragma unroll
    for (int j = 0; j < kDmBatch; j++) {
      const uint32_t st = s0 + j;
      vals[j] = 0u;                                     // entry 0: contributes nothing
      if (st < nst && tabled) vals[j] = st < fd.dim ? xrow[st] : tot;
    }
#pragma unroll
    for (int j = 0; j < kDmBatch; j++) {
      const uint32_t st = s0 + j;
      if (st >= nst) break;
      const uint32_t first_row = fd.dm_meta[2 * st], vcap = fd.dm_meta[2 * st + 1];
      const float *tab = fd.tab + (size_t)first_row * kpad + kb;
#pragma unroll
      for (int r = 0; r < R; r++) {
        const uint32_t vr = (uint32_t)lane_bcast((int)vals[j], r);
        if (vr == 0u || vr >= vcap || ((mbits >> r) & 1ull)) continue;   // (beyond the table: k_gp_large_fix took the whole row)
        const float *p = tab + (size_t)(2 * vr) * kpad;
        add4(hi[r], ld4(p));
        add4(lo[r], ld4(p + kpad));
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; r++) {
    add4(hi[r], lo[r]);
    add4(acc[r], hi[r]);
  }
}

// The same feature with its tables staged in LDS by the group (abi.cpp plan_groups does that when all dim + 1 of
// them fit the slot: small counts).  Branch-free: what must not count -- a masked row, a row whose total is beyond
// the tables (k_gp_large_fix scores it whole) -- reads entry 0, which is exactly zero; four rows' reads go out before
// their adds.  (From L2 the lookups of 4 x dm(4) on 1M rows ran at 1.9 ms; staged, with the branches, 1.75; this: 1.38.)
template <int R>
MSC_DEV void score_dm_feature_staged(const FeatDesc &fd, int lane, uint64_t myrow, bool has_row,
                                     const float4 *__restrict__ lds, float4 (&acc)[R]) {
  const uint32_t nst = fd.dim + 1;
  const unsigned long long mbits = fd.mask == nullptr ? 0ull : __builtin_amdgcn_ballot_w64(load_masked<true>(fd, myrow, has_row));
  float4 hi[R], lo[R];
#pragma unroll
  for (int r = 0; r < R; r++) hi[r] = lo[r] = make_float4(0, 0, 0, 0);
  const uint32_t *xrow = reinterpret_cast<const uint32_t *>(fd.col) + myrow * fd.dim;
  const uint32_t tot = (has_row && fd.col != nullptr) ? fd.dm_tot[myrow] : 0u;
  const bool tabled = has_row && fd.col != nullptr && tot < kGpMaxTable;
  for (uint32_t s0 = 0; s0 < nst; s0 += kDmBatch) {
    uint32_t vals[kDmBatch];
#pragma unroll
    for (int j = 0; j < kDmBatch; j++) {
      const uint32_t st = s0 + j;
      vals[j] = 0u;
      if (st < nst && tabled) vals[j] = st < fd.dim ? xrow[st] : tot;
    }
#pragma unroll
    for (int j = 0; j < kDmBatch; j++) {
      const uint32_t st = s0 + j;
      if (st >= nst) break;
      const uint32_t first_row = fd.dm_meta[2 * st], vcap = fd.dm_meta[2 * st + 1];
      const float4 *stab = lds + (size_t)(fd.grp_off + first_row) * 64 + lane;
#pragma unroll
      for (int r0 = 0; r0 < R; r0 += 4) {
        float4 th[4], tl[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          uint32_t vr = (uint32_t)lane_bcast((int)vals[j], r0 + q);
          vr = (vr >= vcap || ((mbits >> (r0 + q)) & 1ull)) ? 0u : vr;
          th[q] = stab[(2 * vr) * 64];
          tl[q] = stab[(2 * vr + 1) * 64];
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
          add4(hi[r0 + q], th[q]);
          add4(lo[r0 + q], tl[q]);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; r++) {
    add4(hi[r], lo[r]);
    add4(acc[r], hi[r]);
  }
}

// ---------------------------------------------------------------------------
// States without a dm feature.  The host packs consecutive features into groups whose table blocks
// fit the 128 KiB LDS slot together (FeatDesc::grp_*, abi.cpp plan_groups); the workgroup copies a
// whole group, synchronises once, and then every wave runs through the group's features on its
// own: value of its rows, lookups / evaluations, next feature.  Inside a group the waves drift
// apart, so one wave's load latency sits under another's arithmetic; per (feature, 128-row chunk)
// this costs ~1000 cycles where a barrier per feature cost ~2500 (profiles/r01_c3_stage_costs.txt).
// ---------------------------------------------------------------------------
// copy the table blocks of the feature group [f0, f1) into the slot and wait for them (two barriers)
// (the barrier is a policy: the whole workgroup, or -- k_score_tile_roles -- only the waves that share the slot)
struct WorkgroupBarrier {
  MSC_DEV void operator()() const { __syncthreads(); }
};
// A barrier among the first NW waves of the workgroup, through a counter in LDS: every wave adds one and waits until
// the count reaches NW times the number of barriers it has passed.  The waves that take part must execute the same
// sequence of barriers (they do: the same loop over the same feature groups); the wait is bounded all the same, so
// that a mistake costs a reported error (MSC_DEVERR_BARRIER_TIMEOUT) and not a hung GPU.
template <int NW>
struct WaveSubsetBarrier {
  uint32_t *counter;                                    // in LDS, zeroed by the workgroup before first use
  uint32_t passed;
  MSC_DEV void operator()() {
    passed++;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const uint32_t want = passed * (uint32_t)NW;
    bool met = false;
    for (int spin = 0; spin < (1 << 22); spin++) {
      if ((int32_t)(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - want) >= 0) {
        met = true;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    // (never seen; if it happens the rows of this workgroup are wrong and the host must hear of it: the next
    // synchronising or launching call on the context returns MSC_EDEVICE, device_error.hpp)
    if (!met && (threadIdx.x & 63) == 0) report_device_error(MSC_DEVERR_BARRIER_TIMEOUT, blockIdx.x);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
};
template <int W, bool PAIR = false, typename Barrier = WorkgroupBarrier>
MSC_DEV void stage_group(const FeatDesc *__restrict__ feats, int f0, int f1, uint32_t kpad, uint32_t ktile, int lane,
                         int wave, float4 *__restrict__ lds, Barrier &&bar = Barrier()) {
  bar();                                                // the slot's previous readers are done
  for (int f = f0; f < f1; f++) {
    const FeatDesc &fd = feats[f];
    const uint32_t first_row = is_count_family(fd.family) ? (uint32_t)GP_T0 : 0u;
    const float *tile = fd.tab + (size_t)first_row * kpad + (size_t)ktile * kGroupTile;
    if constexpr (PAIR) {
      // the first 128 groups of two table rows per wave instruction: lanes 0-31 row 2 q, lanes 32-63 row 2 q + 1; in the
      // slot a table row is 32 float4 (an odd block's last instruction writes with its upper half masked off)
      float4 *dst = lds + (size_t)fd.grp_off * 32;
      for (uint32_t q = (uint32_t)wave; 2u * q < fd.grp_rows; q += W) {
        const uint32_t row = 2u * q + ((uint32_t)lane >> 5);
        if (row < fd.grp_rows) glds16(tile + (size_t)row * kpad + 4 * (lane & 31), dst + q * 64);
      }
      continue;
    }
    float4 *dst = lds + (size_t)fd.grp_off * 64;
    for (uint32_t row = (uint32_t)wave; row < fd.grp_rows; row += W)     // one 1 KiB table row per wave instruction
      glds16(tile + (size_t)row * kpad + 4 * lane, dst + row * 64);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // my share of the group's tables has landed
  bar();                                                // ... everyone's
}

// ---------------------------------------------------------------------------
// second phase: the plain (unmasked) nich features, which the host puts last and in BLOCKS (abi.cpp plan_groups;
// family_math.hpp "nich BLOCKS").  Every tile kernel forms a row's second-phase sum by the same steps in the same order --
// the features' summed c0, then block by block either ONE compensated log1p of the block's product (nich_block_finish) or,
// where the head kernel found the block's c1 differing, nich_accum feature by feature -- so a row gets the same bits from
// the kernel whose waves split the phases (k_score_tile_roles), from the ones that run them one after the other, and from
// the lane <-> row kernel (k_score_tail_rows<SPLIT>).  A FAR row (a value beyond its feature's xlim: some group's |a| may
// exceed 2^15 and a product of four could leave the float range) takes nich_accum for ALL its plain nich features: the
// hot loops do not look, the rows are found up front (nich_far_rows) and their sums REPLACED afterwards by a call
// (nich_row_plain, out of line: never taken with data that sits where its groups are).
// ---------------------------------------------------------------------------
// the sum of c0 over the second phase's features, for this lane's four groups: what the phase's accumulators start from
// (family_math.hpp nich_accum); read from the tables in L2, once per chunk
MSC_DEV float4 nich_c0_sum(const FeatDesc *__restrict__ feats, int f0, int nfeat, uint32_t kpad, uint32_t kb) {
  float4 s = make_float4(0, 0, 0, 0);
  const scalar_feats sf = as_scalar(feats);
  for (int f = f0; f < nfeat; f++) add4(s, gld4(at_lane(as_global(sf[f].tab), (size_t)NICH_C0 * kpad, kb * 4u)));
  return s;
}
MSC_DEV float2 ld2(const float *p) { return *reinterpret_cast<const float2 *>(p); }

// bit r: the wave's row r (lane r holds it: `myrow`) has a plain nich value beyond its feature's xlim (NaN included)
// (a plan without a block of two or more has no records -- nich_info is null throughout -- and no far rows)
typedef const __attribute__((address_space(4))) NichPlanInfo *scalar_info;
MSC_DEV unsigned long long nich_far_rows(const FeatDesc *__restrict__ feats, int f0, int nfeat, uint64_t myrow, bool has_row) {
  const scalar_feats sf = as_scalar(feats);
  if (f0 >= nfeat || sf[f0].nich_info == nullptr) return 0ull;
  bool far = false;
  for (int f = f0; f < nfeat; f++) {
    const float x = gld1(as_global(sf[f].col) + myrow);
    far |= !(__builtin_fabsf(x) <= ((scalar_info)sf[f].nich_info)->xlim);
  }
  return __builtin_amdgcn_ballot_w64(far && has_row);
}
// a far row's second-phase sum for this lane's four groups, feature by feature (the steps of a block-less plan)
template <bool EST>
static __device__ __attribute__((noinline)) float4 nich_row_plain(const FeatDesc *__restrict__ feats, int f0, int nfeat,
                                                                   uint32_t kpad, uint32_t kb, uint64_t row) {
  float4 a = nich_c0_sum(feats, f0, nfeat, kpad, kb);
  for (int f = f0; f < nfeat; f++) {
    const FeatDesc &fd = feats[f];
    const float *t = fd.tab + kb;
    const float4 mh = ld4(t + (size_t)NICH_MU_HI * kpad), ml = ld4(t + (size_t)NICH_MU_LO * kpad),
                 c1l = ld4(t + (size_t)NICH_C1LN2 * kpad), c1 = ld4(t + (size_t)NICH_C1 * kpad), c2 = ld4(t + (size_t)NICH_C2 * kpad);
    const float x = reinterpret_cast<const float *>(fd.col)[row];
    a.x = nich_accum<EST>(a.x, x, mh.x, ml.x, c1l.x, c1.x, c2.x);
    a.y = nich_accum<EST>(a.y, x, mh.y, ml.y, c1l.y, c1.y, c2.y);
    a.z = nich_accum<EST>(a.z, x, mh.z, ml.z, c1l.z, c1.z, c2.z);
    a.w = nich_accum<EST>(a.w, x, mh.w, ml.w, c1l.w, c1.w, c2.w);
  }
  return a;
}
// One block of M features against NC (1, 2 or 4) of the lane's four groups -- components C0 .. C0 + NC - 1 of the
// accumulators -- for the wave's R rows: 3 M constants of NC groups each in registers (a block of four against all four
// groups would be 52 registers beside 64 of sums and the kernels run four waves a SIMD: MSC_NICH_NC), the rows' values
// broadcast per use.
template <int C> MSC_DEV float &comp(float4 &v) {
  if constexpr (C == 0) return v.x;
  else if constexpr (C == 1) return v.y;
  else if constexpr (C == 2) return v.z;
  else return v.w;
}
// (PAIR: sums r holds the wave's rows 2 r -- components 0, 1 -- and 2 r + 1 -- components 2, 3 --, against the SAME two
// groups: the part for C0 = 2 takes the same constants and the other row's values)
// (XS: the source hands out a wave row's values as SCALAR operands -- NichPacked::xs, s_load from the x matrix --; without
// it they are broadcast from the lanes that hold them, one v_readlane per use)
template <int M, int R, int C0, int NC, bool EST, bool PAIR = false, typename Src>
MSC_DEV void nich_block_rows(const Src &src, int f, const float (&xv)[M], const float (&mh)[M][NC], const float (&ml)[M][NC],
                             const float (&sc)[M][NC], const float (&c1l)[NC], float4 (&acc)[R]) {
  auto squares = [&](int r, float (&t)[NC][M]) {
    float x[M];
    if constexpr (Src::kScalarX) {
      src.template xs<M>(f, PAIR ? 2 * r + C0 / 2 : r, x);
    } else {
#pragma unroll
      for (int j = 0; j < M; j++) x[j] = lane_bcast(xv[j], PAIR ? 2 * r + C0 / 2 : r);
    }
#pragma unroll
    for (int c = 0; c < NC; c++)
#pragma unroll
      for (int j = 0; j < M; j++) t[c][j] = nich_t(x[j], mh[j][c], ml[j][c], sc[j][c]);
  };
  auto finish = [&](int r, const float (&p)[NC]) {
    if constexpr (NC >= 1) comp<C0>(acc[r]) = nich_block_finish<EST>(comp<C0>(acc[r]), p[0], c1l[0]);
    if constexpr (NC >= 2) comp<C0 + 1>(acc[r]) = nich_block_finish<EST>(comp<C0 + 1>(acc[r]), p[1], c1l[1]);
    if constexpr (NC >= 4) {
      comp<C0 + 2>(acc[r]) = nich_block_finish<EST>(comp<C0 + 2>(acc[r]), p[2], c1l[2]);
      comp<C0 + 3>(acc[r]) = nich_block_finish<EST>(comp<C0 + 3>(acc[r]), p[3], c1l[3]);
    }
  };
  // (asking for four rows' values TOGETHER -- one wait a batch instead of the load-wait-use per row the compiler makes of the
  // rows after the first four -- left the scoring kernels where they were and cost the sweep 6 %: its registers.  r05_c3_notes.txt)
#pragma unroll
  for (int r = 0; r < R; r++) {
    float t[NC][M], p[NC];
    squares(r, t);
#pragma unroll
    for (int c = 0; c < NC; c++) p[c] = nich_block_product<M>(t[c]);
    finish(r, p);
    if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // (four rows of temporaries at a time)
  }
}
// where a block's constants come from: the tables in L2 (the role-split kernels' nich waves) ...
struct NichFromGlobal {
  static constexpr bool kScalarX = false;
  const FeatDesc *__restrict__ feats;
  uint32_t kpad, kb;
  // BUFFER loads: the feature's table as a raw buffer (scalar descriptor: base and size), the row and pair as the scalar
  // offset, kb * 4 as the one 32-bit lane offset of every load of the phase.  (Through global loads the compiler formed
  // a 64-bit vector address per load -- thirteen register pairs a block beside 64 registers of sums: the sums spilled.)
  typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));
  typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
  MSC_DEV __amdgpu_buffer_rsrc_t table(int f) const {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(as_scalar(feats)[f].tab), 0, 0x7fffffff, 0x00020000);
  }
  template <int NC> MSC_DEV void comps(int f, int row, int c0, float (&o)[NC]) const {
    const uint32_t so = (uint32_t)(row * kpad + c0) * 4u;
    if constexpr (NC == 1) {
      o[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(table(f), kb * 4u, so, 0));
    } else if constexpr (NC == 2) {
      const u32x2v v = __builtin_amdgcn_raw_buffer_load_b64(table(f), kb * 4u, so, 0);
      o[0] = __uint_as_float(v.x), o[1] = __uint_as_float(v.y);
    } else {
      const u32x4v v = __builtin_amdgcn_raw_buffer_load_b128(table(f), kb * 4u, so, 0);
      o[0] = __uint_as_float(v.x), o[1] = __uint_as_float(v.y), o[2] = __uint_as_float(v.z), o[3] = __uint_as_float(v.w);
    }
  }
  MSC_DEV float4 quad(int f, int row) const {
    const u32x4v v = __builtin_amdgcn_raw_buffer_load_b128(table(f), kb * 4u, (uint32_t)(row * kpad) * 4u, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
  }
  MSC_DEV float x(int f, uint64_t myrow) const { return gld1(as_global(as_scalar(feats)[f].col) + myrow); }
};
// ... the same rows from the PACK (msc_internal.hpp NichPos; k_fuse_tables writes it at the head of every call): one buffer
// for the whole phase -- its descriptor four scalar registers for the kernel's life, a row's offset scalar arithmetic --
// and the rows' values from the x matrix; `f` is a POSITION of the second phase here.  (Through the plan's descriptors
// every block began with a dozen scalar loads, each waiting for the one before: the nich waves issued half the time.)
typedef const __attribute__((address_space(4))) float *scalar_floats;
struct NichPacked {
  static constexpr bool kScalarX = true;
  __amdgpu_buffer_rsrc_t pack;
  const float *__restrict__ xrow;                          // this lane's row of the x matrix
  uint32_t kpad, kb;
  // the wave's rows of the x matrix as SCALAR operands (round 5): row r of the wave's, positions f .. f + M - 1, by s_load
  // from (wave-uniform) xwave + min(r, last) * n2p + f.  (Broadcast from the lanes that hold them they cost a v_readlane per
  // use on the vector pipe -- and the compiler kept the 64 broadcasts of a block across its two parts in scalar registers
  // it did not have: 110-134 of them spilled through v_writelane, profiles/r05_regs.txt.)
  scalar_floats xwave;
  int n2p, last;                                           // last: the wave's last row with data (rows beyond repeat it)
  template <int M> MSC_DEV void xs(int f, int r, float (&o)[M]) const {
    const scalar_floats q = xwave + (size_t)((r < last ? r : last) * n2p + f);
#pragma unroll
    for (int j = 0; j < M; j++) o[j] = q[j];
  }
  typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));
  typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
  static MSC_DEV int pack_row(int nich_row) {
    return nich_row == NICH_MU_HI ? 0 : nich_row == NICH_MU_LO ? 1 : nich_row == NICH_C2 ? 2 : nich_row == NICH_C1LN2 ? 3 : 4;
  }
  MSC_DEV uint32_t at(int f, int row) const { return (uint32_t)((1 + (int)kNichPackRows * f + pack_row(row)) * (int)kpad) * 4u; }
  template <int NC> MSC_DEV void comps(int f, int row, int c0, float (&o)[NC]) const {
    const uint32_t so = at(f, row) + (uint32_t)c0 * 4u;
    if constexpr (NC == 1) {
      o[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(pack, kb * 4u, so, 0));
    } else if constexpr (NC == 2) {
      const u32x2v v = __builtin_amdgcn_raw_buffer_load_b64(pack, kb * 4u, so, 0);
      o[0] = __uint_as_float(v.x), o[1] = __uint_as_float(v.y);
    } else {
      const u32x4v v = __builtin_amdgcn_raw_buffer_load_b128(pack, kb * 4u, so, 0);
      o[0] = __uint_as_float(v.x), o[1] = __uint_as_float(v.y), o[2] = __uint_as_float(v.z), o[3] = __uint_as_float(v.w);
    }
  }
  MSC_DEV float4 quad_at(uint32_t so) const {
    const u32x4v v = __builtin_amdgcn_raw_buffer_load_b128(pack, kb * 4u, so, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
  }
  MSC_DEV float4 quad(int f, int row) const { return quad_at(at(f, row)); }
  // PAIR mode (kb = 2 lane): the lane's two groups, as (a, b, a, b)
  MSC_DEV float4 dup_at(uint32_t so) const {
    const u32x2v v = __builtin_amdgcn_raw_buffer_load_b64(pack, kb * 4u, so, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.x), __uint_as_float(v.y));
  }
  MSC_DEV float4 dup(int f, int row) const { return dup_at(at(f, row)); }
  MSC_DEV float x(int f, uint64_t) const { return gld1(as_global(xrow) + f); }
};
// ... or the staged feature group in LDS (the kernels that run the phases one after the other)
struct NichFromLds {
  static constexpr bool kScalarX = false;
  const FeatDesc *__restrict__ feats;
  const float4 *__restrict__ lds;
  int lane;
  template <int NC> MSC_DEV void comps(int f, int row, int c0, float (&o)[NC]) const {
    const float *p = reinterpret_cast<const float *>(lds + ((size_t)as_scalar(feats)[f].grp_off + row) * 64 + lane) + c0;
    if constexpr (NC == 1) {
      o[0] = p[0];
    } else if constexpr (NC == 2) {
      const float2 v = *reinterpret_cast<const float2 *>(p);
      o[0] = v.x, o[1] = v.y;
    } else {
      const float4 v = *reinterpret_cast<const float4 *>(p);
      o[0] = v.x, o[1] = v.y, o[2] = v.z, o[3] = v.w;
    }
  }
  MSC_DEV float4 quad(int f, int row) const { return lds[((size_t)as_scalar(feats)[f].grp_off + row) * 64 + lane]; }
  MSC_DEV float x(int f, uint64_t myrow) const { return gld1(as_global(as_scalar(feats)[f].col) + myrow); }
};
// groups of the lane's four a block part takes: 2 (26 registers of constants at M = 4 beside the 4 R sums; with 1 the rows'
// values are broadcast four times over -- C3 1.78 ms against 1.57 --, with 4 the constants spill: 2.76)
#ifndef MSC_NICH_NC
#define MSC_NICH_NC 2
#endif
template <int M, int R, int C0, int NC, bool EST, typename Src, bool PAIR = false>
MSC_DEV void nich_block_part(const FeatDesc *__restrict__ feats, int f, const Src &src, const float (&xv)[M], float4 (&acc)[R]) {
  float mh[M][NC], ml[M][NC], sc[M][NC], c1l[NC];
  constexpr int CG = PAIR ? 0 : C0;                     // (PAIR: both halves of the sums against the lane's two groups)
#pragma unroll
  for (int j = 0; j < M; j++) {
    src.template comps<NC>(f + j, NICH_MU_HI, CG, mh[j]);
    src.template comps<NC>(f + j, NICH_MU_LO, CG, ml[j]);
    src.template comps<NC>(f + j, NICH_C2, CG, sc[j]);
  }
  src.template comps<NC>(f, NICH_C1LN2, CG, c1l);
  nich_block_rows<M, R, C0, NC, EST, PAIR>(src, f, xv, mh, ml, sc, c1l, acc);
  __builtin_amdgcn_sched_barrier(0);                          // (the next part's constants after this part's rows)
}
template <int M, int R, bool EST, typename Src, bool PAIR = false, int NCSEL = MSC_NICH_NC>
MSC_DEV void nich_block(const FeatDesc *__restrict__ feats, int f, const Src &src, uint64_t myrow, float4 (&acc)[R]) {
  float xv[M];
#pragma unroll
  for (int j = 0; j < M; j++) xv[j] = Src::kScalarX ? 0.f : src.x(f + j, myrow);
  if constexpr (PAIR) {
    // the lane's two groups against one row of every pair, then against the other.  The constants are fetched again for
    // the second half (an L1 hit), as the four-group form fetches its second part's: kept across both halves they cost the
    // allocator the sums (141-202 spilled registers in three of the four instantiations).
    nich_block_part<M, R, 0, 2, EST, Src, true>(feats, f, src, xv, acc);
    nich_block_part<M, R, 2, 2, EST, Src, true>(feats, f, src, xv, acc);
    return;
  }
  constexpr int NC = NCSEL;                             // (groups of the lane a part takes: MSC_NICH_NC; 4 where the registers allow)
  nich_block_part<M, R, 0, NC, EST>(feats, f, src, xv, acc);
  if constexpr (NC <= 2) {
    if constexpr (Src::kScalarX) {
      // (the second part fetches the rows' values AGAIN -- scalar loads, the scalar cache has them --: read once for both
      // parts they are 16 M scalar registers alive across the block, which the allocator spilled through v_writelane)
      Src again = src;
      asm volatile("" : "+s"(again.xwave));
      nich_block_part<M, R, NC, NC, EST>(feats, f, again, xv, acc);
    } else {
      nich_block_part<M, R, NC, NC, EST>(feats, f, src, xv, acc);
    }
  }
  if constexpr (NC == 1) {
    nich_block_part<M, R, 2, NC, EST>(feats, f, src, xv, acc);
    nich_block_part<M, R, 3, NC, EST>(feats, f, src, xv, acc);
  }
}
// One SEGMENT of the second phase -- the features [f0, f1) of one LDS feature group (abi.cpp plan_layout; whole blocks) --
// in the order every kernel keeps: the segment's blocks of four that go as one, then those of three, of two, then every
// feature left (on its own in the plan, or in a block whose c1 differ), nich_accum.  Four loops with ONE body each: a
// single loop that branches to the four bodies costs ~300 bytes of scratch per lane (the allocator gives up on the 4 R
// sums across the arms) and runs C3 slower than no blocks at all; so the order of the sums follows the loops.
// (Every number a loop branches on goes through readfirstlane: a branch the compiler cannot prove wave-uniform becomes
// predication, with a merge of all the sums behind it.)
template <int M, int R, bool EST, typename Src>
MSC_DEV void nich_pass_blocks(const FeatDesc *__restrict__ feats, int f0, int f1, const Src &src, uint64_t myrow, float4 (&acc)[R]) {
  const scalar_feats sf = as_scalar(feats);
  for (int f = f0; f < f1;) {
    const int len = uniform((int)sf[f].blk_end) - f;
    if (len == M && uniform((int)((scalar_info)sf[f].nich_info)->blk_ok) != 0) nich_block<M, R, EST>(feats, f, src, myrow, acc);
    f = uniform(f + len);
  }
}
template <int R, bool EST, typename Src>
MSC_DEV void nich_pass_plain(const FeatDesc *__restrict__ feats, int f0, int f1, bool blocks, const Src &src, uint64_t myrow, float4 (&acc)[R]) {
  const scalar_feats sf = as_scalar(feats);
  for (int f = f0; f < f1; f = uniform(f + 1)) {
    // (a member of a block that went as one: the block's flag sits at its first feature)
    if (blocks && uniform((int)((scalar_info)sf[uniform((int)sf[f].blk_first)].nich_info)->blk_ok) != 0) continue;
    const float4 mh = src.quad(f, NICH_MU_HI), ml = src.quad(f, NICH_MU_LO), c1l = src.quad(f, NICH_C1LN2),
                 c1 = src.quad(f, NICH_C1), c2 = src.quad(f, NICH_C2);
    const float xv = src.x(f, myrow);
#pragma unroll
    for (int r = 0; r < R; r++) {
      const float x = lane_bcast(xv, r);
      acc[r].x = nich_accum<EST>(acc[r].x, x, mh.x, ml.x, c1l.x, c1.x, c2.x);
      acc[r].y = nich_accum<EST>(acc[r].y, x, mh.y, ml.y, c1l.y, c1.y, c2.y);
      acc[r].z = nich_accum<EST>(acc[r].z, x, mh.z, ml.z, c1l.z, c1.z, c2.z);
      acc[r].w = nich_accum<EST>(acc[r].w, x, mh.w, ml.w, c1l.w, c1.w, c2.w);
      if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // four rows of temporaries at a time: the sums are the registers
    }
  }
}
template <int R, bool EST, typename Src>
MSC_DEV void nich_segment(const FeatDesc *__restrict__ feats, int f0, int f1, bool blocks, const Src &src, uint64_t myrow, float4 (&acc)[R]) {
  if (blocks) {
    if (kNichBlock >= 4) nich_pass_blocks<4, R, EST>(feats, f0, f1, src, myrow, acc);
    if (kNichBlock >= 3) nich_pass_blocks<3, R, EST>(feats, f0, f1, src, myrow, acc);
    nich_pass_blocks<2, R, EST>(feats, f0, f1, src, myrow, acc);
  }
  nich_pass_plain<R, EST>(feats, f0, f1, blocks, src, myrow, acc);
}
// far rows: their sums replaced (wave-uniform; rows in range only)
template <int R, bool EST>
MSC_DEV void nich_redo_far_rows(const FeatDesc *__restrict__ feats, int f0, int nfeat, uint32_t kpad, uint32_t kb,
                                uint64_t row_abs0, unsigned long long far, float4 (&acc)[R]) {
  if (far == 0ull) return;
#pragma unroll
  for (int r = 0; r < R; r++)
    if ((far >> r) & 1ull) acc[r] = nich_row_plain<EST>(feats, f0, nfeat, kpad, kb, row_abs0 + r);
}
// the whole phase with the constants from L2 (k_score_tile_roles / k_sweep_tile_roles: the nich waves); acc is SET
template <int R, bool EST>
MSC_DEV void nich_phase_global(const FeatDesc *__restrict__ feats, int f0, int nfeat, uint32_t kpad, uint32_t kb,
                               uint64_t row_abs0, int nr, uint64_t myrow, float4 (&acc)[R]) {
  const unsigned long long far = nich_far_rows(feats, f0, nfeat, myrow, (threadIdx.x & 63) < nr);
  {
    const float4 c0s = nich_c0_sum(feats, f0, nfeat, kpad, kb);
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = c0s;
  }
  const bool blocks = uniform(as_scalar(feats)[f0].nich_info != nullptr ? 1 : 0) != 0;      // (records for all of the phase's features, or none)
  const NichFromGlobal src{feats, kpad, kb};
  for (int s0 = uniform(f0); s0 < nfeat;) {                    // (segment by segment, as the kernels that stage them in LDS must)
    const int s1 = uniform((int)as_scalar(feats)[s0].grp_end);
    nich_segment<R, EST>(feats, s0, s1, blocks, src, myrow, acc);
    s0 = s1;
  }
  nich_redo_far_rows<R, EST>(feats, f0, nfeat, kpad, kb, row_abs0, far, acc);
}
// The same phase from the pack (NichPacked): the same sums in the same order -- segment by segment, blocks of four, three,
// two, the rest --, the plan read from NichPos records (16 bytes a position, one scalar load) instead of the descriptors.
// `head`: the plan's first second-phase feature (FeatDesc::rn_*).  acc is SET.
typedef const __attribute__((address_space(4))) NichPos *scalar_pos;
// (PAIR: kb = 2 lane, the lane's two groups in x, y; z, w repeat them)
template <bool EST, bool PAIR = false>
static __device__ __attribute__((noinline)) float4 nich_row_plain_packed(const float *pack, const float *xrow, int n2, uint32_t kpad, uint32_t kb) {
  auto ldq = [](const float *p) { return PAIR ? pair_dup(ld2(p)) : ld4(p); };
  float4 a = ldq(pack + kb);
  for (int i = 0; i < n2; i++) {
    const float *t = pack + (size_t)(1 + (int)kNichPackRows * i) * kpad + kb;
    const float4 mh = ldq(t), ml = ldq(t + kpad), c2 = ldq(t + 2 * (size_t)kpad), c1l = ldq(t + 3 * (size_t)kpad), c1 = ldq(t + 4 * (size_t)kpad);
    const float x = xrow[i];
    a.x = nich_accum<EST>(a.x, x, mh.x, ml.x, c1l.x, c1.x, c2.x);
    a.y = nich_accum<EST>(a.y, x, mh.y, ml.y, c1l.y, c1.y, c2.y);
    a.z = nich_accum<EST>(a.z, x, mh.z, ml.z, c1l.z, c1.z, c2.z);
    a.w = nich_accum<EST>(a.w, x, mh.w, ml.w, c1l.w, c1.w, c2.w);
  }
  return a;
}
template <int M, int R, bool EST, bool PAIR, int NCSEL>
MSC_DEV void nich_pass_blocks_packed(scalar_pos pos, int p0, int p1, const NichPacked &src, float4 (&acc)[R]) {
  for (int p = p0; p < p1;) {
    const int len = uniform((int)(pos[p].blk >> 16));
    if (len == M && uniform((int)pos[p].blk_ok) != 0) nich_block<M, R, EST, NichPacked, PAIR, NCSEL>(nullptr, p, src, 0, acc);
    p = uniform(p + len);
  }
}
// (PAIR: kb = 2 lane; nr counts ROWS -- up to 2 R --, lane i < nr holds row i)
template <int R, bool EST, bool PAIR = false, int NCSEL = MSC_NICH_NC>
MSC_DEV void nich_phase_packed(const FeatDesc *__restrict__ feats, int f0, uint32_t kpad, uint32_t kb, uint64_t row_abs0, int nr,
                               uint64_t myrow, float4 (&acc)[R]) {
  const scalar_feats sf = as_scalar(feats);
  const float *const packp = sf[f0].rn_pack;
  const scalar_pos pos = (scalar_pos)sf[f0].rn_pos;
  const int n2 = uniform((int)sf[f0].rn_n2), n2p = uniform((int)sf[f0].rn_n2p);
  const float *const xrow = sf[f0].rn_x + myrow * (uint64_t)n2p;
  // (lane 0's row is the wave's first: its own first row, or -- a wave without rows -- the call's first)
  const uint64_t wave_row = ((uint64_t)(uint32_t)uniform((int)(myrow >> 32)) << 32) | (uint32_t)uniform((int)(uint32_t)myrow);
  const NichPacked src{__builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(packp), 0, 0x7fffffff, 0x00020000), xrow, kpad, kb,
                       (scalar_floats)(sf[f0].rn_x + wave_row * (uint64_t)n2p), n2p, (nr > 0 ? nr : 1) - 1};
  // far rows (bit r: the wave's row r; NaN included): the row's values a quad at a time against the positions' limits
  // (the padding: value 0 against +inf)
  bool far = false;
  for (int q = 0; q < n2p; q += 4) {
    const float4 x4 = gld4(as_global(xrow) + q);
    far |= !(__builtin_fabsf(x4.x) <= pos[q].xlim) | !(__builtin_fabsf(x4.y) <= pos[q + 1].xlim) |
           !(__builtin_fabsf(x4.z) <= pos[q + 2].xlim) | !(__builtin_fabsf(x4.w) <= pos[q + 3].xlim);
  }
  const unsigned long long farbits = __builtin_amdgcn_ballot_w64(far && (int)(threadIdx.x & 63) < nr);
  {
    const float4 c0s = PAIR ? src.dup_at(0u) : src.quad_at(0u);
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = c0s;
  }
  for (int s0 = 0; s0 < n2;) {
    const int s1 = uniform((int)pos[s0].seg_end);
    if (kNichBlock >= 4) nich_pass_blocks_packed<4, R, EST, PAIR, NCSEL>(pos, s0, s1, src, acc);
    if (kNichBlock >= 3) nich_pass_blocks_packed<3, R, EST, PAIR, NCSEL>(pos, s0, s1, src, acc);
    nich_pass_blocks_packed<2, R, EST, PAIR, NCSEL>(pos, s0, s1, src, acc);
    for (int p = s0; p < s1; p = uniform(p + 1)) {
      if (uniform((int)pos[uniform((int)(pos[p].blk & 0xffffu))].blk_ok) != 0) continue;     // (went as one with its block)
      auto row_of = [&](int nich_row) { return PAIR ? src.dup(p, nich_row) : src.quad(p, nich_row); };
      const float4 mh = row_of(NICH_MU_HI), ml = row_of(NICH_MU_LO), c1l = row_of(NICH_C1LN2), c1 = row_of(NICH_C1), c2 = row_of(NICH_C2);
      const float xv = src.x(p, 0);
#pragma unroll
      for (int r = 0; r < R; r++) {
        const float x = lane_bcast(xv, PAIR ? 2 * r : r), xb = PAIR ? lane_bcast(xv, 2 * r + 1) : x;
        acc[r].x = nich_accum<EST>(acc[r].x, x, mh.x, ml.x, c1l.x, c1.x, c2.x);
        acc[r].y = nich_accum<EST>(acc[r].y, x, mh.y, ml.y, c1l.y, c1.y, c2.y);
        acc[r].z = nich_accum<EST>(acc[r].z, xb, mh.z, ml.z, c1l.z, c1.z, c2.z);
        acc[r].w = nich_accum<EST>(acc[r].w, xb, mh.w, ml.w, c1l.w, c1.w, c2.w);
        if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
    }
    s0 = s1;
  }
  if (farbits != 0ull) {
    if constexpr (PAIR) {
#pragma unroll
      for (int r = 0; r < R; r++) {
        if ((farbits >> (2 * r)) & 1ull) {
          const float4 v = nich_row_plain_packed<EST, true>(packp, sf[f0].rn_x + (row_abs0 + 2 * r) * (uint64_t)n2p, n2, kpad, kb);
          acc[r].x = v.x, acc[r].y = v.y;
        }
        if ((farbits >> (2 * r + 1)) & 1ull) {
          const float4 v = nich_row_plain_packed<EST, true>(packp, sf[f0].rn_x + (row_abs0 + 2 * r + 1) * (uint64_t)n2p, n2, kpad, kb);
          acc[r].z = v.x, acc[r].w = v.y;
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < R; r++)
        if ((farbits >> r) & 1ull)
          acc[r] = nich_row_plain_packed<EST>(packp, sf[f0].rn_x + (row_abs0 + r) * (uint64_t)n2p, n2, kpad, kb);
    }
  }
}
// The first phase of a plan with FEW lookup features, without the table slot: every row's table row gathered straight from
// the tables in L2 (1 KiB a row and feature, 512 B in PAIR mode) and added to the sums in plan order -- what the lookup waves
// of the role-split kernels do through LDS, for the kernels whose waves are all nich waves (k_score_nich_pack with a
// first phase).  Costs the texture path 16 cycles a row and feature, which is why it is for few features only
// (profiles/r04_roles_phases.txt: sixteen dd features moved this way cost more than the LDS saved).
template <int R, bool PAIR>
MSC_DEV void pack_l2_lookups(const FeatDesc *__restrict__ feats, int nsplit, uint32_t kpad, uint32_t kb, uint64_t myrow, float4 (&acc)[R]) {
  const scalar_feats sf = as_scalar(feats);
  typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));
  typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
  for (int f = 0; f < nsplit; f = uniform(f + 1)) {
    // the value of this lane's row, as the lookup runs take it (score_tile_groups): a byte or a dword, clamped into the table
    const bool u8 = uniform((int)sf[f].kind) == MSC_KIND_LOOKUP_U8;
    const uint64_t base = reinterpret_cast<uint64_t>(sf[f].col);
    const uint64_t at = (base + (u8 ? myrow : myrow * 4)) & ~(uint64_t)3;
    const uint32_t word = *reinterpret_cast<const __attribute__((address_space(1))) uint32_t *>((const __attribute__((address_space(1))) unsigned char *)at);
    const uint32_t sh = u8 ? (uint32_t)((base + myrow) & 3u) * 8u : 0u;
    const int v = (int)(u8 ? (word >> sh) & 0xffu : word);
    const int clampv = uniform((int)sf[f].run_clamp);
    const int idx = v < 0 ? 0 : (v > clampv ? clampv : v);
    const uint32_t first_row = is_count_family(uniform((int)sf[f].family)) ? (uint32_t)GP_T0 : 0u;
    const __amdgpu_buffer_rsrc_t tab = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(sf[f].tab), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int r0 = 0; r0 < R; r0 += 4) {
      if constexpr (PAIR) {
        u32x2v ta[4], tb[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          ta[j] = __builtin_amdgcn_raw_buffer_load_b64(tab, kb * 4u, (first_row + (uint32_t)lane_bcast(idx, 2 * (r0 + j))) * kpad * 4u, 0);
          tb[j] = __builtin_amdgcn_raw_buffer_load_b64(tab, kb * 4u, (first_row + (uint32_t)lane_bcast(idx, 2 * (r0 + j) + 1)) * kpad * 4u, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
          add4(acc[r0 + j], make_float4(__uint_as_float(ta[j].x), __uint_as_float(ta[j].y), __uint_as_float(tb[j].x), __uint_as_float(tb[j].y)));
      } else {
        u32x4v t[4];
#pragma unroll
        for (int j = 0; j < 4; j++)
          t[j] = __builtin_amdgcn_raw_buffer_load_b128(tab, kb * 4u, (first_row + (uint32_t)lane_bcast(idx, r0 + j)) * kpad * 4u, 0);
#pragma unroll
        for (int j = 0; j < 4; j++)
          add4(acc[r0 + j], make_float4(__uint_as_float(t[j].x), __uint_as_float(t[j].y), __uint_as_float(t[j].z), __uint_as_float(t[j].w)));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}
// (acc is SET here: the phase's sums start from nich_c0_sum, not from what acc held)
template <int R, int W, bool EST = false>
MSC_DEV void score_tile_nich_tail(const FeatDesc *__restrict__ feats, int f0, int nfeat, uint32_t kpad, uint32_t ktile,
                                  int lane, int wave, uint64_t row_abs0, int nr, uint64_t row_safe, float4 *__restrict__ lds,
                                  float4 (&acc)[R]) {
  const uint32_t kb = ktile * kGroupTile + lane * 4;
  const uint64_t myrow = lane < nr ? row_abs0 + lane : row_safe;    // (a row of the call's range for idle lanes)
  const int first = f0;
  const unsigned long long far = nich_far_rows(feats, f0, nfeat, myrow, lane < nr);
  {
    const float4 c0s = nich_c0_sum(feats, f0, nfeat, kpad, kb);
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = c0s;
  }
  const bool blocks = uniform(as_scalar(feats)[first].nich_info != nullptr ? 1 : 0) != 0;
  while (f0 < nfeat) {
    const int f1 = uniform((int)as_scalar(feats)[f0].grp_end);  // (a block never lies across two groups: abi.cpp plan_layout)
    stage_group<W>(feats, f0, f1, kpad, ktile, lane, wave, lds);
    nich_segment<R, EST>(feats, f0, f1, blocks, NichFromLds{feats, lds, lane}, myrow, acc);
    f0 = f1;
  }
  nich_redo_far_rows<R, EST>(feats, first, nfeat, kpad, kb, row_abs0, far, acc);
}

// GENERIC = false: the caller knows every feature of the phase to be of a lookup kind (k_score_tile_roles: the host
// checks the plan), and the branch for everything else -- with its temporaries -- is not compiled in
// PAIR (the role-split kernels at <= 128 groups, see pair_dup above): nr counts rows -- up to 2 R, lane i holds row i's value
// --, a table row in the slot is 64 float2 and a lookup of a pair of rows two ds_read_b64
template <int R, int W, bool DM, bool GENERIC = true, bool PAIR = false, typename Barrier = WorkgroupBarrier>
MSC_DEV void score_tile_groups(const FeatDesc *__restrict__ feats, int nfeat, uint32_t kpad, uint32_t ktile,
                               int lane, uint64_t row_abs0, int nr, uint64_t row_safe, float4 *__restrict__ lds,
                               float4 (&acc)[R], Barrier &&bar = Barrier()) {
  static_assert(!PAIR || (!GENERIC && !DM), "PAIR mode: lookup runs only");
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t kb = ktile * kGroupTile + lane * 4;
  const bool has_row = lane < nr;
  const uint64_t myrow = row_abs0 + lane;
  if constexpr (!GENERIC) {
    // ---- lookup runs from the plan's INDEX MATRIX (round 5; FeatDesc::lk_idx, abi.cpp look_idx_matrix) ----
    // A feature used to be a chain descriptor head -> value load -> clamp -> broadcast -> LDS reads (the pipeline below,
    // which the kernels with generic features keep); here a row's lookups arrive ready-made: per feature group ONE 16-byte
    // load per lane (lane r: row r's slot rows for up to sixteen features of the group, a byte each), issued before the
    // group's table copies so that its latency lies under theirs.  A dword of it serves four features: its sixteen
    // (PAIR: thirty-two) rows' words are broadcast once (v_readlane) and a lookup is then s_bfe (scalar unit) +
    // v_lshl_add + ds_read + the adds.  Same table rows added in the same order: same bits.
    constexpr int RWI = PAIR ? 2 * R : R;
    typedef uint32_t u32x4i __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) u32x4i *gu32x4_p;
    typedef const __attribute__((address_space(1))) uint32_t *gu32_p;
    const scalar_feats sf = as_scalar(feats);
    const uint32_t l4 = (uint32_t)uniform((int)sf[0].lk_l4);
    const gu32_p rec = (gu32_p)sf[0].lk_idx + (has_row ? myrow : row_safe) * (uint64_t)l4;
    for (int f0 = 0; f0 < nfeat;) {
      const int f1 = uniform((int)sf[f0].grp_end), ng = f1 - f0;
      const uint32_t goff = (uint32_t)uniform((int)sf[f0].lk_goff);
      u32x4i w = *(gu32x4_p)(rec + goff);                // (dword-aligned; the matrix ends in three dwords of slack)
      stage_group<W, PAIR>(feats, f0, f1, kpad, ktile, lane, wave, lds, bar);
      for (int q = 0; 4 * q < ng; q = uniform(q + 1)) {
        if (q != 0 && (q & 3) == 0) w = *(gu32x4_p)(rec + goff + q);       // (a group of more than sixteen features)
        const int c = q & 3;
        const uint32_t word = c == 0 ? w.x : c == 1 ? w.y : c == 2 ? w.z : w.w;
        uint32_t sw[RWI];
#pragma unroll
        for (int r = 0; r < RWI; r++) sw[r] = (uint32_t)lane_bcast((int)word, r);
        const int nj = ng - 4 * q < 4 ? ng - 4 * q : 4;
        for (int j = 0; j < nj; j = uniform(j + 1)) {
          const uint32_t sh = 8u * (uint32_t)j;
          if constexpr (PAIR) {
            const float2 *buf2 = reinterpret_cast<const float2 *>(lds) + lane;
#pragma unroll
            for (int r0 = 0; r0 < R; r0 += 4) {
              float2 ta[4], tb[4];
#pragma unroll
              for (int jj = 0; jj < 4; jj++) {
                ta[jj] = buf2[((sw[2 * (r0 + jj)] >> sh) & 0xffu) * 64u];
                tb[jj] = buf2[((sw[2 * (r0 + jj) + 1] >> sh) & 0xffu) * 64u];
              }
#pragma unroll
              for (int jj = 0; jj < 4; jj++) add4(acc[r0 + jj], make_float4(ta[jj].x, ta[jj].y, tb[jj].x, tb[jj].y));
              __builtin_amdgcn_sched_barrier(0);
            }
          } else {
            const float4 *buf = lds + lane;
#pragma unroll
            for (int r0 = 0; r0 < R; r0 += 4) {
              float4 t[4];
#pragma unroll
              for (int jj = 0; jj < 4; jj++) t[jj] = buf[((sw[r0 + jj] >> sh) & 0xffu) * 64u];
#pragma unroll
              for (int jj = 0; jj < 4; jj++) add4(acc[r0 + jj], t[jj]);
              __builtin_amdgcn_sched_barrier(0);          // four reads in flight, not eight: 16 fewer live registers
            }
          }
        }
      }
      f0 = f1;
    }
    return;
  }
  int f0 = 0;
  while (f0 < nfeat) {
    const int f1 = (int)feats[f0].grp_end;
    stage_group<W, PAIR>(feats, f0, f1, kpad, ktile, lane, wave, lds, bar);
    int f = f0;
    while (f < f1) {
      // A run of unmasked lookup features (bb, gp, bnb, dd with their whole table staged; the host marks them
      // and the run's end): nothing but the value load, eight ds_read_b128 and the adds.  One entry, one exit,
      // so the accumulators stay where they are.
      const int fe = (int)feats[f].run_end;
      // A feature is a chain descriptor -> value -> LDS reads, and the waves spent half their time parked on it
      // (SQ_WAIT_ANY 0.52 of the wave cycles).  So the run is a software pipeline, written so that
      // the compiler's wait-count bookkeeping can follow it: no branch around a load (past the run's end the last
      // feature is fetched again; lanes without a row fetch the wave's first row), global -- not flat -- loads (a flat
      // load counts on lgkmcnt too, and every wait for an LDS read would wait for it), and two features per trip so
      // that nothing fetched ahead has to be copied from one register to another.  Values: one dword load for either
      // column type (a bool column: the aligned dword around the byte, then the byte); bool columns as 0 / 1, 32-bit
      // ones clamped into the staged block (negative -> 0, never taken with valid data: the rows cover the column's
      // maximum) -- min(max(v, 0), clamp) does both, the clamp of a bool column being 1.
      struct Head { const void *col; uint32_t off, clamp, kind; };
      auto head_of = [&](int g) -> Head {
        const FeatDesc &d = feats[g < fe ? g : fe - 1];
        return Head{d.col, d.grp_off, d.run_clamp, d.kind};
      };
      const uint64_t ldrow = has_row ? myrow : row_safe;      // (a row of the call's range, whatever this wave's share is)
      auto fetch = [&](const Head &h) -> uint32_t {
        typedef const __attribute__((address_space(1))) unsigned char *g_u8;
        const bool u8 = h.kind == MSC_KIND_LOOKUP_U8;
        const uint64_t byte = u8 ? ldrow : ldrow * 4;
        const uint64_t base = reinterpret_cast<uint64_t>(h.col);
        const uint64_t at = (base + byte) & ~(uint64_t)3;     // (the columns are dword-aligned for 32-bit types anyway)
        return *reinterpret_cast<const __attribute__((address_space(1))) uint32_t *>((g_u8)at);
      };
      auto index_of = [&](const Head &h, uint32_t word) -> uint32_t {
        const bool u8 = h.kind == MSC_KIND_LOOKUP_U8;
        const uint32_t sh = u8 ? (uint32_t)((reinterpret_cast<uint64_t>(h.col) + ldrow) & 3u) * 8u : 0u;
        const int v = (int)(u8 ? (word >> sh) & 0xffu : word);
        return (uint32_t)(v < 0 ? 0 : (v > (int)h.clamp ? (int)h.clamp : v));
      };
      auto lookups = [&](const Head &h, uint32_t idx) {
        const float4 *buf = lds + (size_t)h.off * 64 + lane;
        if constexpr (PAIR) {
          const float2 *buf2 = reinterpret_cast<const float2 *>(lds) + (size_t)h.off * 64 + lane;
#pragma unroll
          for (int r0 = 0; r0 < R; r0 += 4) {
            float2 ta[4], tb[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
              ta[j] = buf2[(uint32_t)lane_bcast((int)idx, 2 * (r0 + j)) * 64];
              tb[j] = buf2[(uint32_t)lane_bcast((int)idx, 2 * (r0 + j) + 1) * 64];
            }
#pragma unroll
            for (int j = 0; j < 4; j++) add4(acc[r0 + j], make_float4(ta[j].x, ta[j].y, tb[j].x, tb[j].y));
            __builtin_amdgcn_sched_barrier(0);
          }
          return;
        }
#pragma unroll
        for (int r0 = 0; r0 < R; r0 += 4) {
          float4 t[4];
#pragma unroll
          for (int j = 0; j < 4; j++) t[j] = buf[(uint32_t)lane_bcast((int)idx, r0 + j) * 64];
#pragma unroll
          for (int j = 0; j < 4; j++) add4(acc[r0 + j], t[j]);
          __builtin_amdgcn_sched_barrier(0);            // four reads in flight, not eight: 16 fewer live registers
        }
      };
      if (f < fe) {
        Head ha = head_of(f), hb = head_of(f + 1);
        uint32_t wa = fetch(ha);
        for (; f < fe; f += 2) {
          // (the value fetched a feature ago is TAKEN before the next fetch is issued -- the scheduling barrier keeps the
          // load below -- or the wait for it counts the fresh load too: the ISA had vmcnt(0) behind every other prefetch)
          const uint32_t ia = index_of(ha, wa);
          __builtin_amdgcn_sched_barrier(0);
          const uint32_t wb = fetch(hb);                      // value of f + 1
          const Head hc = head_of(f + 2);                     // descriptor of f + 2
          lookups(ha, ia);
          if (f + 1 >= fe) { f++; break; }
          const uint32_t ib = index_of(hb, wb);
          __builtin_amdgcn_sched_barrier(0);
          wa = fetch(hc);                                     // value of f + 2
          ha = hc;
          const Head hd = head_of(f + 3);                     // descriptor of f + 3
          lookups(hb, ib);
          hb = hd;
        }
        if (f > fe) f = fe;
      }
      if (f >= f1) break;
      if (!GENERIC) continue;                           // (the next run starts here)
      const FeatDesc &fd = feats[f];
      if (fd.kind != MSC_KIND_GENERIC) continue;        // the next run starts here
      if (DM && fd.family == MSC_DM) {
        if (fd.dm_meta != nullptr) {
          if (fd.grp_rows != 0) score_dm_feature_staged<R>(fd, lane, myrow, has_row, lds, acc);
          else score_dm_feature<R>(fd, kpad, kb, lane, myrow, has_row, acc);
        }
        f++;
        continue;
      }
      const uint32_t raw = load_raw_value<false>(fd, 0, myrow, has_row);
      const unsigned long long mbits =
          fd.mask == nullptr ? 0ull : __builtin_amdgcn_ballot_w64(load_masked<false>(fd, myrow, has_row));
      const float4 *buf = lds + (size_t)fd.grp_off * 64;
      if (mbits == 0ull) add_feature<R, false>(fd, buf, fd.grp_rows, kpad, kb, lane, raw, mbits, acc);
      else add_feature<R, true>(fd, buf, fd.grp_rows, kpad, kb, lane, raw, mbits, acc);
      f++;
    }
    f0 = f1;
  }
}

// ---------------------------------------------------------------------------
// acc[r] (+)= sum over features of score_value(row rb+r, groups kb..kb+3).
// All W waves of the workgroup must call this together (it contains barriers); a wave whose
// rows are out of range passes nr = 0; row_safe = any row of the call's range (lanes without a row read it and
// drop what they read).  lds: the kGrpRows * 64 float4 slot.
// ---------------------------------------------------------------------------
// SPLIT (the scoring kernels): the unmasked nich features are summed on their own, from zero, and the two sums added
// at the end -- (what acc held + the first phase's features) + (the second phase's) -- so that the kernel whose waves
// split the two phases between them (k_score_tile_roles) and the ones that run them one after the other give a row
// the same bits.  A state of unmasked nich features only (nsplit == 0) has nothing in the first sum: its caller
// starts acc at zero and adds what it would have started from afterwards (the same sum, one accumulator alive).
template <int R, int W, bool DM, bool SPLIT = false, bool EST = false>
MSC_DEV void score_tile(const FeatDesc *__restrict__ feats, int nfeat, int nsplit, uint32_t kpad, uint32_t ktile,
                        int lane, uint64_t row_abs0, int nr, uint64_t row_safe, float4 *__restrict__ lds, float4 (&acc)[R]) {
  score_tile_groups<R, W, DM>(feats, nsplit, kpad, ktile, lane, row_abs0, nr, row_safe, lds, acc);
  if (nsplit < nfeat) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    static_assert(SPLIT, "the second phase is always summed on its own (score_tile_nich_tail sets its accumulators)");
    if (nsplit == 0) {                                    // (the caller passed zeros and adds the prior afterwards)
      score_tile_nich_tail<R, W, EST>(feats, nsplit, nfeat, kpad, ktile, lane, wave, row_abs0, nr, row_safe, lds, acc);
    } else {
      float4 accn[R];
      score_tile_nich_tail<R, W, EST>(feats, nsplit, nfeat, kpad, ktile, lane, wave, row_abs0, nr, row_safe, lds, accn);
#pragma unroll
      for (int r = 0; r < R; r++) add4(acc[r], accn[r]);
    }
  }
}

// ---- what the kernels that draw share (kernels_sweep.hip, and the lane <-> row kernel of kernels_score.hip) ----
// ---- Philox-4x32-10 (Salmon et al. SC'11): uniform of (seed, sweep, global row) ----
MSC_DEV float philox_uniform01(uint64_t seed, uint64_t sweep, uint64_t row) {
  uint32_t c0 = (uint32_t)row, c1 = (uint32_t)(row >> 32), c2 = (uint32_t)sweep, c3 = (uint32_t)(sweep >> 32);
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; r++) {
    const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
    const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
    c0 = n0; c1 = l1; c2 = n2; c3 = l0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return (float)(c0 >> 8) * (1.0f / 16777216.0f);
}

// the whole grid shares the zeroing of the additive tables (nothing in a sweep kernel reads them)
MSC_DEV void zero_spans(const ZeroSpans &zs) {
  const size_t n = zs.na + zs.nb;
  if (n == 0) return;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    if (i < zs.na) zs.a[i] = 0ull;
    else zs.b[i - zs.na] = 0ull;
  }
}

}  // namespace msc
