// kernels_score.hip -- gfx950 kernels for the scoring side of the hot path:
//   k_prepare      suff-stats -> per-group float score constants (double math)
//   k_crp_prepare  group sizes -> log pseudocounts (group_manager.hpp:274-283)
//   k_score_nich1  one NICH feature, [nrows x K] scores, streaming 1 KiB stores
//   k_score_mixed  any feature list; scores summed over features in registers
//
// Mapping used by every score kernel: a wave owns a block of rows and one k-tile
// of 256 groups; lane l owns groups 4l..4l+3 of the tile, so a row of the tile
// is one 16-byte value per lane = one 1 KiB contiguous store per wave
// (HBM-write-bound configs need nothing else on the critical path).  Row values
// are loaded coalesced (lane r <- row r of the block) and broadcast with
// v_readlane; per-group constants live in VGPRs for the whole block.
#include "family_math.hpp"
#include "launchers.hpp"

namespace msc {

MSC_DEV float lane_bcast(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
MSC_DEV int lane_bcast(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

MSC_DEV float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------
// prepare: one thread per (feature, group slot); pads (k >= K) are prepared from
// their zeroed raw stats so that vector loads of a full tile stay finite.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_prepare(const FeatDesc *__restrict__ feats, uint32_t kpad) {
  const FeatDesc fd = feats[blockIdx.y];
  const uint32_t k = blockIdx.x * 256 + threadIdx.x;
  if (k >= kpad) return;
  switch (fd.family) {
    case MSC_BB: {
      float s0, s1;
      bb_prepare(fd.hp, fd.raw_u32[k], fd.raw_u32[kpad + k], s0, s1);
      fd.tab[k] = s0;
      fd.tab[kpad + k] = s1;
    } break;
    case MSC_GP: {
      const uint32_t cnt = fd.raw_u32[k], sum = fd.raw_u32[kpad + k];
      gp_prepare_consts(fd.hp, cnt, sum, fd.tab[(size_t)GP_NSE_HI * kpad + k], fd.tab[(size_t)GP_NSE_LO * kpad + k]);
      for (uint32_t v = 0; v < GP_TABLE; v++)
        fd.tab[(size_t)(GP_T0 + v) * kpad + k] = gp_prepare_table(fd.hp, cnt, sum, v);
    } break;
    case MSC_DD: {
      double asum = 0;
      for (uint32_t i = 0; i < fd.dim; i++) asum += (double)fd.hp[i];
      const uint32_t csum = fd.raw_u32[k];
      for (uint32_t i = 0; i < fd.dim; i++)
        fd.tab[(size_t)i * kpad + k] =
            dd_prepare_entry(fd.hp[i], fd.raw_u32[(size_t)(1 + i) * kpad + k], asum, csum);
    } break;
    case MSC_NICH: {
      float o[NICH_ROWS];
      nich_prepare(fd.hp, fd.raw_u32[k], fd.raw_f32[k], fd.raw_f32[kpad + k], o);
#pragma unroll
      for (int i = 0; i < NICH_ROWS; i++) fd.tab[(size_t)i * kpad + k] = o[i];
    } break;
    default: break;
  }
}

// crp layout: [0,kpad) log(cnt) or -inf when empty; [kpad,2kpad) log(cnt-1) or -inf;
// [2kpad] = log(alpha / n_empty), [2kpad+1] = log(alpha / (n_empty+1)).  One block.
__global__ __launch_bounds__(256) void k_crp_prepare(const uint32_t *__restrict__ cnt, uint32_t K,
                                                      uint32_t kpad, float alpha, float *crp) {
  __shared__ uint32_t s_empty;
  if (threadIdx.x == 0) s_empty = 0;
  __syncthreads();
  uint32_t mine = 0;
  for (uint32_t k = threadIdx.x; k < kpad; k += 256) {
    const uint32_t c = k < K ? cnt[k] : 0;
    crp[k] = c ? (float)log((double)c) : -INFINITY;
    crp[kpad + k] = c > 1 ? (float)log((double)c - 1.0) : -INFINITY;
    if (k < K && c == 0) mine++;
  }
  atomicAdd(&s_empty, mine);
  __syncthreads();
  if (threadIdx.x == 0) {
    const double ne = s_empty;
    crp[2 * (size_t)kpad] = ne > 0 ? (float)log((double)alpha / ne) : -INFINITY;
    crp[2 * (size_t)kpad + 1] = (float)log((double)alpha / (ne + 1.0));
  }
}

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
// single NICH feature (config C2 / C5 scoring pass)
//   grid.x = row chunks (grid-stride), grid.y = k-tiles, block = 4 waves
// ---------------------------------------------------------------------------
template <bool LOO, bool CRP>
__global__ __launch_bounds__(256) void k_score_nich1(const FeatDesc *__restrict__ feats,
                                                      uint32_t K, uint32_t kpad, uint64_t row0,
                                                      uint64_t nrows, const int32_t *__restrict__ z,
                                                      const float *__restrict__ crp,
                                                      float *__restrict__ out, uint64_t ld) {
  const FeatDesc fd = feats[0];
  const int lane = threadIdx.x & 63;
  const uint32_t kb = blockIdx.y * kGroupTile + lane * 4;
  const float *tab = fd.tab + kb;
  const float4 mh = ld4(tab + (size_t)NICH_MU_HI * kpad), ml = ld4(tab + (size_t)NICH_MU_LO * kpad),
               c0 = ld4(tab + (size_t)NICH_C0 * kpad), c1l = ld4(tab + (size_t)NICH_C1LN2 * kpad),
               c1 = ld4(tab + (size_t)NICH_C1 * kpad), c2 = ld4(tab + (size_t)NICH_C2 * kpad);
  float4 logcnt = make_float4(0, 0, 0, 0);
  float le0 = 0, le1 = 0;
  if (CRP) {
    logcnt = ld4(crp + kb);
    le0 = crp[2 * (size_t)kpad];
    le1 = crp[2 * (size_t)kpad + 1];
  }
  const bool vec_ok = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  const float *xcol = reinterpret_cast<const float *>(fd.col) + row0;
  const uint64_t nchunks = (nrows + 63) / 64;
  const uint64_t wave_id = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint64_t nwaves = (uint64_t)gridDim.x * 4;
  for (uint64_t chunk = wave_id; chunk < nchunks; chunk += nwaves) {
    const uint64_t rb = chunk * 64;
    const int nr = (int)((nrows - rb) < 64 ? (nrows - rb) : 64);
    const float xv = lane < nr ? xcol[rb + lane] : 0.0f;
    int gz = -1;
    float sloo = 0, pg = 0, erow = le0;
    if (LOO) {
      if (lane < nr) gz = z[rb + lane];
      if (gz >= 0) {
        const uint32_t cg = fd.raw_u32[gz];
        sloo = (float)nich_loo(fd.hp, cg, fd.raw_f32[gz], fd.raw_f32[kpad + gz], xv);
        if (CRP) {
          const float lm1 = crp[kpad + gz];
          const bool single = __builtin_isinf(lm1);
          pg = single ? le1 : lm1;
          erow = single ? le1 : le0;
          sloo += pg;
        }
      }
    }
#pragma unroll 4
    for (int r = 0; r < nr; r++) {
      const float x = lane_bcast(xv, r);
      float4 s;
      s.x = nich_eval(x, mh.x, ml.x, c0.x, c1l.x, c1.x, c2.x);
      s.y = nich_eval(x, mh.y, ml.y, c0.y, c1l.y, c1.y, c2.y);
      s.z = nich_eval(x, mh.z, ml.z, c0.z, c1l.z, c1.z, c2.z);
      s.w = nich_eval(x, mh.w, ml.w, c0.w, c1l.w, c1.w, c2.w);
      if (CRP) {
        const float4 p = crp_prior4(logcnt, LOO ? lane_bcast(erow, r) : le0);
        s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
      }
      if (LOO) {
        const int g = lane_bcast(gz, r);
        if (g >= 0) replace_own(s, kb, g, lane_bcast(sloo, r));
      }
      store_row(out, ld, rb + r, kb, K, s, vec_ok);
    }
  }
}

// ---------------------------------------------------------------------------
// general path: any feature list, R rows per wave block kept in registers.
// ---------------------------------------------------------------------------
template <int R, bool LOO, bool CRP>
__global__ __launch_bounds__(256) void k_score_mixed(const FeatDesc *__restrict__ feats, int nfeat,
                                                      uint32_t K, uint32_t kpad, uint64_t row0,
                                                      uint64_t nrows, const int32_t *__restrict__ z,
                                                      const float *__restrict__ crp,
                                                      float *__restrict__ out, uint64_t ld) {
  const int lane = threadIdx.x & 63;
  const uint32_t kb = blockIdx.y * kGroupTile + lane * 4;
  const bool vec_ok = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  const uint64_t nblocks = (nrows + R - 1) / R;
  const uint64_t wave_id = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint64_t nwaves = (uint64_t)gridDim.x * 4;
  float4 logcnt = make_float4(0, 0, 0, 0);
  float le0 = 0, le1 = 0;
  if (CRP) {
    logcnt = ld4(crp + kb);
    le0 = crp[2 * (size_t)kpad];
    le1 = crp[2 * (size_t)kpad + 1];
  }
  for (uint64_t blk = wave_id; blk < nblocks; blk += nwaves) {
    const uint64_t rb = blk * R;                       // relative to row0
    const int nr = (int)((nrows - rb) < (uint64_t)R ? (nrows - rb) : (uint64_t)R);
    const bool has_row = lane < nr;
    const uint64_t myrow = row0 + rb + lane;           // absolute row of this lane (if has_row)
    int gz = -1;
    float own = 0;     // lane r: sum over features of the loo score of row r's own group (+ prior)
    float erow = le0;
    if (LOO && has_row) gz = z[rb + lane];
    if (LOO && CRP && gz >= 0) {
      const float lm1 = crp[kpad + gz];
      const bool single = __builtin_isinf(lm1);
      own = single ? le1 : lm1;
      erow = single ? le1 : le0;
    }
    float4 acc[R];
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
#pragma unroll
    for (int r = 0; r < R; r++) {
      if (r < nr) {
        float4 s = acc[r];
        if (LOO) {
          const int g = lane_bcast(gz, r);
          if (g >= 0) replace_own(s, kb, g, lane_bcast(own, r));
        }
        store_row(out, ld, rb + r, kb, K, s, vec_ok);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// host-side launchers (called from abi.cpp)
// ---------------------------------------------------------------------------
int launch_prepare(hipStream_t stream, const FeatDesc *feats_dev, uint32_t nfeat, uint32_t kpad) {
  dim3 grid((kpad + 255) / 256, nfeat);
  hipLaunchKernelGGL(k_prepare, grid, dim3(256), 0, stream, feats_dev, kpad);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_crp_prepare(hipStream_t stream, const uint32_t *cnt, uint32_t K, uint32_t kpad, float alpha,
                       float *crp) {
  hipLaunchKernelGGL(k_crp_prepare, dim3(1), dim3(256), 0, stream, cnt, K, kpad, alpha, crp);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <bool LOO, bool CRP>
static void launch_score_t(hipStream_t stream, int num_cus, bool nich1, const FeatDesc *feats_dev,
                           int nfeat, uint32_t K, uint32_t kpad, uint64_t row0, uint64_t nrows,
                           const int32_t *z, const float *crp, float *out, uint64_t ld) {
  const uint32_t ktiles = kpad / kGroupTile;
  if (nich1) {
    const uint64_t nchunks = (nrows + 63) / 64;
    uint64_t gx = (nchunks + 3) / 4;
    const uint64_t cap = (uint64_t)num_cus * 16;
    if (gx > cap) gx = cap;
    if (gx == 0) gx = 1;
    hipLaunchKernelGGL((k_score_nich1<LOO, CRP>), dim3((unsigned)gx, ktiles), dim3(256), 0, stream,
                       feats_dev, K, kpad, row0, nrows, z, crp, out, ld);
  } else {
    constexpr int R = 16;
    const uint64_t nblocks = (nrows + R - 1) / R;
    uint64_t gx = (nblocks + 3) / 4;
    const uint64_t cap = (uint64_t)num_cus * 16;
    if (gx > cap) gx = cap;
    if (gx == 0) gx = 1;
    hipLaunchKernelGGL((k_score_mixed<R, LOO, CRP>), dim3((unsigned)gx, ktiles), dim3(256), 0, stream,
                       feats_dev, nfeat, K, kpad, row0, nrows, z, crp, out, ld);
  }
}

int launch_score(hipStream_t stream, int num_cus, bool nich1, const FeatDesc *feats_dev, int nfeat,
                 uint32_t K, uint32_t kpad, uint64_t row0, uint64_t nrows, const int32_t *z,
                 const float *crp, float *out, uint64_t ld) {
  const bool loo = z != nullptr, pri = crp != nullptr;
  if (loo && pri) launch_score_t<true, true>(stream, num_cus, nich1, feats_dev, nfeat, K, kpad, row0, nrows, z, crp, out, ld);
  else if (loo) launch_score_t<true, false>(stream, num_cus, nich1, feats_dev, nfeat, K, kpad, row0, nrows, z, crp, out, ld);
  else if (pri) launch_score_t<false, true>(stream, num_cus, nich1, feats_dev, nfeat, K, kpad, row0, nrows, z, crp, out, ld);
  else launch_score_t<false, false>(stream, num_cus, nich1, feats_dev, nfeat, K, kpad, row0, nrows, z, crp, out, ld);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace msc
