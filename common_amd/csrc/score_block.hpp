// score_block.hpp -- the workgroup-tile scorer shared by the batched score kernel and the
// fused sweep kernel.
//
// A workgroup of 8 waves owns 8*R consecutive rows and one k-tile of 256 groups; lane l owns
// groups 4l..4l+3 of the tile, so every per-group quantity is one 16-byte element per lane.
// Features are processed one after the other: the workgroup stages the feature's per-group
// table for the tile into LDS once (bb: 2 rows, nich: 6 constants, dd: one row per category,
// gp: one row per count; a row = 256 floats = 1 KiB), then every wave adds that feature's
// scores for its R rows into registers with conflict-free ds_read_b128 (address = row*1 KiB +
// lane*16 B).  Nothing in here evaluates a transcendental in double: leave-one-out terms
// arrive precomputed per row (k_loo_own), gp counts beyond the table are patched afterwards
// (k_gp_large_fix).  Table rows beyond the 64 that fit the LDS block are gathered from L2.
#pragma once

#include "family_math.hpp"

namespace msc {

MSC_DEV float lane_bcast(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
MSC_DEV int lane_bcast(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

MSC_DEV float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kTileWaves = 8;                 // waves per workgroup of the tile kernels
constexpr int kTileThreads = kTileWaves * 64;
constexpr int kLdsRows = 64;                  // table rows staged per feature (64 KiB)

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

MSC_DEV void store_row(float *__restrict__ out, uint64_t ld, uint64_t row, uint32_t kb, uint32_t K,
                       float4 s, bool vec_ok) {
  float *p = out + row * ld + kb;
  if (vec_ok && kb + 3 < K) {
    const f32x4 v = {s.x, s.y, s.z, s.w};
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(p));
  } else {
    if (kb < K) p[0] = s.x;
    if (kb + 1 < K) p[1] = s.y;
    if (kb + 2 < K) p[2] = s.z;
    if (kb + 3 < K) p[3] = s.w;
  }
}

// rows of the feature's table block that go to LDS, and where the block starts in fd.tab
MSC_DEV uint32_t lds_rows_of(const FeatDesc &fd, uint32_t &first_row) {
  first_row = 0;
  switch (fd.family) {
    case MSC_BB: return 2;
    case MSC_NICH: return NICH_ROWS;
    case MSC_DD: return fd.dim < (uint32_t)kLdsRows ? fd.dim : (uint32_t)kLdsRows;
    case MSC_GP: first_row = GP_T0; return fd.vcap < (uint32_t)kLdsRows ? fd.vcap : (uint32_t)kLdsRows;
    default: return 0;
  }
}

// ---------------------------------------------------------------------------
// acc[r] (+)= sum over features of score_value(row rb+r, groups kb..kb+3).
// All 8 waves of the workgroup must call this together (it contains barriers); a wave whose
// rows are out of range passes nr = 0.  lds: kLdsRows * 64 float4.
// ---------------------------------------------------------------------------
template <int R>
MSC_DEV void score_tile(const FeatDesc *__restrict__ feats, int nfeat, uint32_t kpad, uint32_t ktile,
                        int lane, uint64_t row_abs0, int nr, float4 *__restrict__ lds, float4 (&acc)[R]) {
  const uint32_t kb = ktile * kGroupTile + lane * 4;
  const bool has_row = lane < nr;
  const uint64_t myrow = row_abs0 + lane;
  for (int f = 0; f < nfeat; f++) {
    const FeatDesc fd = feats[f];
    uint32_t first_row;
    const uint32_t nrows_lds = lds_rows_of(fd, first_row);
    const float *tile = fd.tab + (size_t)first_row * kpad + (size_t)ktile * kGroupTile;
    __syncthreads();                                    // readers of the previous feature are done
    for (uint32_t idx = threadIdx.x; idx < nrows_lds * 64; idx += kTileThreads)
      lds[idx] = ld4(tile + (size_t)(idx >> 6) * kpad + 4 * (idx & 63));
    __syncthreads();
    switch (fd.family) {
      case MSC_BB: {
        const float4 s0 = lds[lane], s1 = lds[64 + lane];
        const int v = has_row ? (int)(reinterpret_cast<const uint8_t *>(fd.col)[myrow] != 0) : 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
          const bool vr = lane_bcast(v, r) != 0;
          acc[r].x += vr ? s1.x : s0.x;
          acc[r].y += vr ? s1.y : s0.y;
          acc[r].z += vr ? s1.z : s0.z;
          acc[r].w += vr ? s1.w : s0.w;
        }
      } break;
      case MSC_DD: {
        int v = has_row ? reinterpret_cast<const int32_t *>(fd.col)[myrow] : 0;
        v = v < 0 ? 0 : (v >= (int)fd.dim ? (int)fd.dim - 1 : v);       // keep the gather in bounds
#pragma unroll
        for (int r = 0; r < R; r++) {
          const uint32_t vr = (uint32_t)lane_bcast(v, r);
          add4(acc[r], vr < nrows_lds ? lds[vr * 64 + lane] : ld4(fd.tab + (size_t)vr * kpad + kb));
        }
      } break;
      case MSC_GP: {
        const uint32_t v = has_row ? reinterpret_cast<const uint32_t *>(fd.col)[myrow] : 0u;
#pragma unroll
        for (int r = 0; r < R; r++) {
          const uint32_t vr = (uint32_t)lane_bcast((int)v, r);
          if (vr < nrows_lds) add4(acc[r], lds[vr * 64 + lane]);
          else if (vr < fd.vcap) add4(acc[r], ld4(fd.tab + (size_t)(GP_T0 + vr) * kpad + kb));
          // counts beyond the table contribute through k_gp_large_fix
        }
      } break;
      case MSC_NICH: {
        const float4 mh = lds[NICH_MU_HI * 64 + lane], ml = lds[NICH_MU_LO * 64 + lane],
                     c0 = lds[NICH_C0 * 64 + lane], c1l = lds[NICH_C1LN2 * 64 + lane],
                     c1 = lds[NICH_C1 * 64 + lane], c2 = lds[NICH_C2 * 64 + lane];
        const float xv = has_row ? reinterpret_cast<const float *>(fd.col)[myrow] : 0.0f;
#pragma unroll
        for (int r = 0; r < R; r++) {
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
}

}  // namespace msc
