// score_block.hpp -- the per-(row block, k-tile) body shared by the batched score
// kernel and the fused sweep kernel: scores of R rows against the 4 groups this
// lane owns, summed over every feature of the state, kept in registers.
#pragma once

#include "family_math.hpp"

namespace msc {

MSC_DEV float lane_bcast(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
MSC_DEV int lane_bcast(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

MSC_DEV float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------
// CRP prior for one row of a k-tile (4 groups per lane).  e_row = log(alpha/n_empty')
// for this row (n_empty' counts the row's own group if removing it empties it).
// ---------------------------------------------------------------------------
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
MSC_DEV void add_own(float4 &s, uint32_t kb, int g, float v) {
  if ((int)kb == g) s.x += v;
  if ((int)kb + 1 == g) s.y += v;
  if ((int)kb + 2 == g) s.z += v;
  if ((int)kb + 3 == g) s.w += v;
}

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

// ---------------------------------------------------------------------------
// acc[r] = sum over features of score_value(row rb+r, groups kb..kb+3) (+ CRP prior).
// With LOO, on return lane r holds in (gz, own) the own group of row rb+r and the
// leave-one-out score (incl. prior) that must replace acc[r]'s entry for that group.
// ---------------------------------------------------------------------------
template <int R, bool LOO, bool CRP>
MSC_DEV void score_block(const FeatDesc *__restrict__ feats, int nfeat, uint32_t kpad, uint32_t kb,
                         int lane, uint64_t row0, uint64_t rb, int nr, const int32_t *__restrict__ z,
                         const float *__restrict__ crp, float4 logcnt, float le0, float le1,
                         float4 (&acc)[R], int &gz, float &own) {
      const bool has_row = lane < nr;
      const uint64_t myrow = row0 + rb + lane;           // absolute row of this lane (if has_row)
      gz = -1;
      own = 0;     // lane r: sum over features of the loo score of row r's own group (+ prior)
      float erow = le0;
      if (LOO && has_row) gz = z[rb + lane];
      if (LOO && CRP && gz >= 0) {
        const float lm1 = crp[kpad + gz];
        const bool single = __builtin_isinf(lm1);
        own = single ? le1 : lm1;
        erow = single ? le1 : le0;
      }
      #pragma unroll
      for (int r = 0; r < R; r++) {
        if (CRP) acc[r] = crp_prior4(logcnt, LOO ? lane_bcast(erow, r) : le0);
        else acc[r] = make_float4(0, 0, 0, 0);
      }
      for (int f = 0; f < nfeat; f++) {
        const FeatDesc fd = feats[f];
        const float *tab = fd.tab + kb;
        switch (fd.family) {
          case MSC_BB: {
            const float4 s0 = ld4(tab), s1 = ld4(tab + kpad);
            const int v = has_row ? (int)(reinterpret_cast<const uint8_t *>(fd.col)[myrow] != 0) : 0;
            if (LOO && gz >= 0) own += (float)bb_loo(fd.hp, fd.raw_u32[gz], fd.raw_u32[kpad + gz], v != 0);
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
            v = v < 0 ? 0 : (v >= (int)fd.dim ? (int)fd.dim - 1 : v);   // keep the gather in bounds
            if (LOO && gz >= 0) {
              double asum = 0;
              for (uint32_t i = 0; i < fd.dim; i++) asum += (double)fd.hp[i];
              own += (float)dd_loo(fd.hp[v], fd.raw_u32[(size_t)(1 + v) * kpad + gz], asum, fd.raw_u32[gz]);
            }
  #pragma unroll
            for (int r = 0; r < R; r++) {
              const float4 t = ld4(tab + (size_t)lane_bcast(v, r) * kpad);
              acc[r].x += t.x; acc[r].y += t.y; acc[r].z += t.z; acc[r].w += t.w;
            }
          } break;
          case MSC_GP: {
            const uint32_t v = has_row ? reinterpret_cast<const uint32_t *>(fd.col)[myrow] : 0u;
            if (LOO && gz >= 0) own += (float)gp_loo(fd.hp, fd.raw_u32[gz], fd.raw_u32[kpad + gz], v);
            // wave-uniform: does any row of the block need the large-count path?
            const bool any_large = __builtin_amdgcn_ballot_w64(v >= (uint32_t)GP_TABLE) != 0ull;
            double ga[4], gb[4], gn[4], rowc = 0.0;
            if (any_large) {
              const double al = fd.hp[0], ib = fd.hp[1];
  #pragma unroll
              for (int j = 0; j < 4; j++) {
                gb[j] = ib + (double)fd.raw_u32[kb + j];                      // row 0: count
                ga[j] = al + (double)fd.raw_u32[(size_t)kpad + kb + j];      // row 1: sum
                gn[j] = (double)tab[(size_t)GP_NSE_HI * kpad + j] + (double)tab[(size_t)GP_NSE_LO * kpad + j];
              }
              if (v >= (uint32_t)GP_TABLE) rowc = gp_row_const(v);
            }
  #pragma unroll
            for (int r = 0; r < R; r++) {
              const uint32_t vr = (uint32_t)lane_bcast((int)v, r);
              float4 s;
              if (vr < (uint32_t)GP_TABLE) {
                s = ld4(tab + (size_t)(GP_T0 + vr) * kpad);
              } else {
                const double vd = (double)vr;
                const double rc = __hiloint2double(lane_bcast(__double2hiint(rowc), r), lane_bcast(__double2loint(rowc), r));
                s.x = gp_eval_large(vd, rc, ga[0], gb[0], gn[0]);
                s.y = gp_eval_large(vd, rc, ga[1], gb[1], gn[1]);
                s.z = gp_eval_large(vd, rc, ga[2], gb[2], gn[2]);
                s.w = gp_eval_large(vd, rc, ga[3], gb[3], gn[3]);
              }
              acc[r].x += s.x; acc[r].y += s.y; acc[r].z += s.z; acc[r].w += s.w;
            }
          } break;
          case MSC_NICH: {
            const float4 mh = ld4(tab + (size_t)NICH_MU_HI * kpad), ml = ld4(tab + (size_t)NICH_MU_LO * kpad),
                         c0 = ld4(tab + (size_t)NICH_C0 * kpad), c1l = ld4(tab + (size_t)NICH_C1LN2 * kpad),
                         c1 = ld4(tab + (size_t)NICH_C1 * kpad), c2 = ld4(tab + (size_t)NICH_C2 * kpad);
            const float xv = has_row ? reinterpret_cast<const float *>(fd.col)[myrow] : 0.0f;
            if (LOO && gz >= 0)
              own += (float)nich_loo(fd.hp, fd.raw_u32[gz], fd.raw_f32[gz], fd.raw_f32[kpad + gz], xv);
  #pragma unroll
            for (int r = 0; r < R; r++) {
              const float x = lane_bcast(xv, r);
              acc[r].x += nich_eval(x, mh.x, ml.x, c0.x, c1l.x, c1.x, c2.x);
              acc[r].y += nich_eval(x, mh.y, ml.y, c0.y, c1l.y, c1.y, c2.y);
              acc[r].z += nich_eval(x, mh.z, ml.z, c0.z, c1l.z, c1.z, c2.z);
              acc[r].w += nich_eval(x, mh.w, ml.w, c0.w, c1l.w, c1.w, c2.w);
            }
          } break;
          default: break;   // noop model contributes 0 (models/noop.hpp:17)
        }
      }
}

}  // namespace msc
