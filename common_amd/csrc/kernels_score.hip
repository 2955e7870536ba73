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
#include "score_block.hpp"

namespace msc {

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
    float4 acc[R];
    int gz;
    float own;
    score_block<R, LOO, CRP>(feats, nfeat, kpad, kb, lane, row0, rb, nr, z, crp, logcnt, le0, le1, acc, gz, own);
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
