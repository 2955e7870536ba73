// score_block.hpp -- the workgroup-tile scorer shared by the batched score kernel and the
// fused sweep kernel.
//
// A workgroup owns a block of consecutive rows and one tile of groups.  Features are processed
// one after the other: the feature's per-group table for the tile (bb: 2 rows, nich: 6
// constants, dd: one row per category, gp: one row per count) is copied into LDS once per
// workgroup with global_load_lds (double-buffered: the copy of feature f+1 runs under the
// arithmetic of feature f), then every wave adds that feature's scores for its rows into
// registers with conflict-free ds_read_b128 (lane <-> 4 groups of a 256-group tile, 1 KiB
// table rows, 128 rows per workgroup).  Table bytes moved per evaluation are
// table_rows * 4 B / rows_per_workgroup and the global->LDS path moves ~10 B/clk/CU, which is
// what bounds a table-heavy state (config C3); a lane <-> 1 group tiling (512 rows per
// workgroup, 4x fewer table bytes) was measured 2x slower: it triples the per-evaluation
// instruction count of the lookups.
// Nothing in here evaluates a transcendental in double: leave-one-out terms arrive precomputed
// per row (k_loo_own), gp counts beyond the table are patched afterwards (k_gp_large_fix).
// Table rows beyond the 64 that fit an LDS buffer are gathered from L2.
#pragma once

#include "family_math.hpp"

namespace msc {

MSC_DEV float lane_bcast(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
MSC_DEV int lane_bcast(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

MSC_DEV float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kLdsRows = 64;                  // table rows staged per feature and buffer (64 KiB)

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
MSC_DEV void replace_own(float4 &s, uint32_t kb, int g, float v) {
  if ((int)kb == g) s.x = v;
  if ((int)kb + 1 == g) s.y = v;
  if ((int)kb + 2 == g) s.z = v;
  if ((int)kb + 3 == g) s.w = v;
}
MSC_DEV void add4(float4 &a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }

template <bool NT = true>
MSC_DEV void store_row(float *__restrict__ out, uint64_t ld, uint64_t row, uint32_t kb, uint32_t K,
                       float4 s, bool vec_ok) {
  float *p = out + row * ld + kb;
  if (vec_ok && kb + 3 < K) {
    const f32x4 v = {s.x, s.y, s.z, s.w};
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(p));
    else *reinterpret_cast<f32x4 *>(p) = v;
  } else {
    if (kb < K) p[0] = s.x;
    if (kb + 1 < K) p[1] = s.y;
    if (kb + 2 < K) p[2] = s.z;
    if (kb + 3 < K) p[3] = s.w;
  }
}

// A feature is scored in one or more *stages*, each one table block staged through LDS and one
// value per row: every family has a single stage except dm, which has dim + 1 (one count lookup
// per category, one for the row total; family_math.hpp).
MSC_DEV uint32_t nstages_of(const FeatDesc &fd) { return fd.family == MSC_DM ? fd.dim + 1 : 1; }

// rows of a single-stage feature's table block that go to LDS, and where the block starts in fd.tab
// wg_row0 / wg_rows: the absolute row range the workgroup scores in this chunk; for count tables only
// the rows up to the largest count in that range are worth copying (the per-128-row maxima
// were computed once when the column was bound).
MSC_DEV uint32_t chunk_need(const uint16_t *cmax, uint32_t need, uint64_t wg_row0, uint32_t wg_rows) {
  if (cmax != nullptr) {
    uint32_t m = 0;
    for (uint64_t c = wg_row0 >> 7; c <= (wg_row0 + wg_rows - 1) >> 7; c++) {
      const uint32_t v = cmax[c];
      m = v > m ? v : m;
    }
    need = m + 1 < need ? m + 1 : need;
  }
  return need;
}
MSC_DEV uint32_t lds_rows_of(const FeatDesc &fd, uint32_t &first_row, uint64_t wg_row0, uint32_t wg_rows) {
  first_row = 0;
  switch (fd.family) {
    case MSC_BB: return 2;
    case MSC_BBNC: return 2;
    case MSC_NICH: return NICH_ROWS;
    case MSC_DD: return fd.dim < (uint32_t)kLdsRows ? fd.dim : (uint32_t)kLdsRows;
    case MSC_BNB:
    case MSC_GP: {
      first_row = GP_T0;
      const uint32_t need = chunk_need(fd.chunk_max, fd.vcap, wg_row0, wg_rows);
      return need < (uint32_t)kLdsRows ? need : (uint32_t)kLdsRows;
    }
    default: return 0;
  }
}
// the same for stage `sub` of a dm feature: entries are (hi, lo) row pairs
MSC_DEV uint32_t dm_lds_rows_of(const FeatDesc &fd, uint32_t sub, uint32_t &first_row, uint64_t wg_row0, uint32_t wg_rows) {
  first_row = fd.dm_meta[2 * sub];
  const uint32_t need = 2 * chunk_need(fd.chunk_max ? fd.chunk_max + (size_t)sub * fd.cm_stride : nullptr,
                                       fd.dm_meta[2 * sub + 1], wg_row0, wg_rows);
  return need < (uint32_t)kLdsRows ? need : (uint32_t)kLdsRows;
}

// the stage's value of this lane's row, as raw 32 bits (reinterpreted per family)
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

// issue the async copy of the stage's table block for this k-tile into buf (no wait); returns the rows copied
template <int W, bool DM>
MSC_DEV uint32_t stage_table(const FeatDesc &fd, uint32_t sub, uint32_t kpad, uint32_t ktile, float4 *buf,
                             uint64_t wg_row0, uint32_t wg_rows) {
  uint32_t first_row;
  const uint32_t nrows_lds = (DM && fd.family == MSC_DM) ? dm_lds_rows_of(fd, sub, first_row, wg_row0, wg_rows)
                                                         : lds_rows_of(fd, first_row, wg_row0, wg_rows);
  const float *tile = fd.tab + (size_t)first_row * kpad + (size_t)ktile * kGroupTile;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t row = wave; row < nrows_lds; row += W)          // one 1 KiB table row per wave instruction
    glds16(tile + (size_t)row * kpad + 4 * lane, buf + row * 64);
  return nrows_lds;
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

// one dm stage: the exact (hi) and remainder (lo) parts are summed separately (family_math.hpp)
template <int R, bool MASKED>
MSC_DEV void add_dm_stage(const FeatDesc &fd, uint32_t sub, const float4 *__restrict__ buf, const uint32_t nrows_lds,
                          uint32_t kpad, uint32_t kb, int lane, uint32_t raw, unsigned long long mbits,
                          float4 (&hi)[R], float4 (&lo)[R]) {
  const uint32_t first_row = fd.dm_meta[2 * sub], vcap = fd.dm_meta[2 * sub + 1];
#pragma unroll
  for (int r = 0; r < R; r++) {
    if (MASKED && ((mbits >> r) & 1ull)) continue;
    const uint32_t vr = (uint32_t)lane_bcast((int)raw, r);
    if (vr >= vcap) continue;                         // rows beyond the tables: k_gp_large_fix
    if (2 * vr + 1 < nrows_lds) {
      add4(hi[r], buf[(2 * vr) * 64 + lane]);
      add4(lo[r], buf[(2 * vr + 1) * 64 + lane]);
    } else {
      const float *p = fd.tab + (size_t)(first_row + 2 * vr) * kpad + kb;
      add4(hi[r], ld4(p));
      add4(lo[r], ld4(p + kpad));
    }
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
template <int R, int W>
MSC_DEV void score_tile_groups(const FeatDesc *__restrict__ feats, int nfeat, uint32_t kpad, uint32_t ktile,
                               int lane, uint64_t row_abs0, int nr, float4 *__restrict__ lds, float4 (&acc)[R]) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t kb = ktile * kGroupTile + lane * 4;
  const bool has_row = lane < nr;
  const uint64_t myrow = row_abs0 + lane;
  int f0 = 0;
  while (f0 < nfeat) {
    const int f1 = (int)feats[f0].grp_end;
    __syncthreads();                                    // the slot's previous readers are done
    for (int f = f0; f < f1; f++) {
      const FeatDesc &fd = feats[f];
      const uint32_t first_row = is_count_family(fd.family) ? (uint32_t)GP_T0 : 0u;
      const float *tile = fd.tab + (size_t)first_row * kpad + (size_t)ktile * kGroupTile;
      float4 *dst = lds + (size_t)fd.grp_off * 64;
      for (uint32_t row = (uint32_t)wave; row < fd.grp_rows; row += W)   // one 1 KiB table row per wave instruction
        glds16(tile + (size_t)row * kpad + 4 * lane, dst + row * 64);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // my share of the group's tables has landed
    __syncthreads();                                    // ... everyone's
    int f = f0;
    while (f < f1) {
      // A run of unmasked lookup features (bb, gp, bnb, dd with their whole table staged; the host marks them
      // and the run's end): nothing but the value load, eight ds_read_b128 and the adds.  One entry, one exit,
      // so the accumulators stay where they are.
      const int fe = (int)feats[f].run_end;
      for (; f < fe; f++) {
        const FeatDesc &fd = feats[f];
        const uint32_t kind = fd.kind;
        uint32_t idx = 0;
        if (has_row) {
          if (kind == MSC_KIND_LOOKUP_U8) idx = (uint32_t)(reinterpret_cast<const uint8_t *>(fd.col)[myrow] != 0);
          else idx = reinterpret_cast<const uint32_t *>(fd.col)[myrow];
        }
        if (kind == MSC_KIND_LOOKUP_I32) {
          const int v = (int)idx;
          idx = (uint32_t)(v < 0 ? 0 : (v >= (int)fd.dim ? (int)fd.dim - 1 : v));     // keep the gather in bounds
        } else if (kind == MSC_KIND_LOOKUP_U32) {
          idx = idx < fd.grp_rows ? idx : fd.grp_rows - 1;                              // (never taken: rows cover the column's maximum)
        }
        const float4 *buf = lds + (size_t)fd.grp_off * 64 + lane;
#pragma unroll
        for (int r0 = 0; r0 < R; r0 += 4) {
          float4 t[4];
#pragma unroll
          for (int j = 0; j < 4; j++) t[j] = buf[(uint32_t)lane_bcast((int)idx, r0 + j) * 64];
#pragma unroll
          for (int j = 0; j < 4; j++) add4(acc[r0 + j], t[j]);
          __builtin_amdgcn_sched_barrier(0);            // four reads in flight, not eight: 16 fewer live registers
        }
      }
      if (f >= f1) break;
      const FeatDesc &fd = feats[f];
      if (fd.kind != MSC_KIND_GENERIC) continue;        // the next run starts here
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
// rows are out of range passes nr = 0.  lds: two buffers of kLdsRows * 64 float4.
// Pipeline per stage s: issue the value load and the table copy of s+1 (buffer (s+1)&1),
// compute s from buffer s&1, wait for the copies, one barrier.  The value load is issued
// *before* the table copy because vmcnt retires in order.
// ---------------------------------------------------------------------------
// DM: the state holds a Dirichlet-Multinomial feature (their hi/lo accumulators cost 2 R float4 of
// registers, so states without one run the kernel compiled without that branch).
template <int R, int W, bool DM>
MSC_DEV void score_tile(const FeatDesc *__restrict__ feats, int nfeat, uint32_t kpad, uint32_t ktile,
                        int lane, uint64_t row_abs0, int nr, uint64_t wg_row0, uint32_t wg_rows,
                        float4 *__restrict__ lds, float4 (&acc)[R]) {
  if constexpr (!DM) {
    score_tile_groups<R, W>(feats, nfeat, kpad, ktile, lane, row_abs0, nr, lds, acc);
    return;
  }
  const uint32_t kb = ktile * kGroupTile + lane * 4;
  const bool has_row = lane < nr;
  const uint64_t myrow = row_abs0 + lane;
  __syncthreads();                                      // the previous chunk's readers are done
  uint32_t raw = load_raw_value<DM>(feats[0], 0, myrow, has_row);
  unsigned long long mbits = __builtin_amdgcn_ballot_w64(load_masked<DM>(feats[0], myrow, has_row));
  uint32_t nrows_lds = stage_table<W, DM>(feats[0], 0, kpad, ktile, lds, wg_row0, wg_rows);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if constexpr (!DM) return;                          // (states without a dm feature take score_tile_groups)
  // advance the pipeline by one stage: prefetch (value, mask, table) of the stage after (f, sub)
  // into the other buffer, run `body` on the current one, wait, barrier, rotate
  int f = 0;
  uint32_t parity = 0;
  auto step = [&](const FeatDesc &fd, uint32_t sub, auto &&body) {
    const float4 *buf = lds + (size_t)parity * kLdsRows * 64;
    int nf = f;
    uint32_t nsub = sub + 1;
    if (nsub >= nstages_of(fd)) {
      nf = f + 1;
      nsub = 0;
    }
    uint32_t raw_next = 0, nrows_next = 0;
    unsigned long long mbits_next = mbits;              // the stages of one feature share its mask
    if (nf < nfeat) {
      raw_next = load_raw_value<DM>(feats[nf], nsub, myrow, has_row);
      if (nf != f) mbits_next = __builtin_amdgcn_ballot_w64(load_masked<DM>(feats[nf], myrow, has_row));
      nrows_next = stage_table<W, DM>(feats[nf], nsub, kpad, ktile, lds + (size_t)(parity ^ 1u) * kLdsRows * 64, wg_row0, wg_rows);
    }
    body(buf);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the next table has landed (this wave's share)
    __syncthreads();                                    // ... everyone's share; this buffer is free again
    raw = raw_next;
    mbits = mbits_next;
    nrows_lds = nrows_next;
    parity ^= 1u;
  };
  while (f < nfeat) {
    const FeatDesc fd = feats[f];
    if (DM && fd.family == MSC_DM) {
      float4 hi[R], lo[R];                              // live only across this feature's stages
#pragma unroll
      for (int r = 0; r < R; r++) hi[r] = lo[r] = make_float4(0, 0, 0, 0);
      for (uint32_t sub = 0; sub <= fd.dim; sub++)
        step(fd, sub, [&](const float4 *buf) {
          if (mbits == 0ull) add_dm_stage<R, false>(fd, sub, buf, nrows_lds, kpad, kb, lane, raw, mbits, hi, lo);
          else add_dm_stage<R, true>(fd, sub, buf, nrows_lds, kpad, kb, lane, raw, mbits, hi, lo);
        });
#pragma unroll
      for (int r = 0; r < R; r++) {
        add4(hi[r], lo[r]);
        add4(acc[r], hi[r]);
      }
    } else if (fd.family != MSC_DM) {
      step(fd, 0, [&](const float4 *buf) {
        if (mbits == 0ull) add_feature<R, false>(fd, buf, nrows_lds, kpad, kb, lane, raw, mbits, acc);
        else add_feature<R, true>(fd, buf, nrows_lds, kpad, kb, lane, raw, mbits, acc);
      });
    }
    f++;
  }
}

}  // namespace msc
