// kernels_niw.hip -- Normal-Inverse-Wishart (distributions.hpp:87-91,481-509: NormalInverseWishart<-1>, the
// dimension is a run-time value), dim <= 128.
//
//   k_niw_prepare   one wave per group: posterior (kappa_n, nu_n, mu_n, Psi_n) in double,
//                   Cholesky Psi_n = L L^T and L^-1 as packed triangles in LDS, then
//                     W_k = L^-1 * sqrt(kappa_n / (kappa_n + 1))
//                   so that the multivariate-t predictive is
//                     score(x) = c0_k - c1_k * log1p(|W_k (x - mu_k)|^2)
//                   (the reference refactors Sigma for every (row, group) pair).  W_k is lower
//                   triangular and is written as the operand stream of the f64 kernel: 16 x 16 blocks, only those
//                   on or below the diagonal (msc_internal.hpp niw_w_index).
//   k_score_niw     the only dense contraction on the path -> fp32 MFMA
//                   (v_mfma_f32_32x32x2_f32, exact f32 fma chains): per group, A = W_k
//                   (32 x 32), B = (X_tile - mu_k)^T (32 features x 32 rows); |.|^2 over the
//                   output rows is an in-register sum + one cross-half shuffle; lane <-> data row.
//   k_score_niw64   the default: the same contraction on the f64 matrix pipe
//                   (v_mfma_f64_16x16x4_f64), block by block of the triangle: component block b meets feature
//                   blocks 0 .. b only, so dim 32 costs 12 matrix instructions per (16 rows, group) instead of
//                   the 16 of the full square, and any dimension up to 128 is NB (NB + 1) / 2 chunks of four.
//                   In float the score error is c1 * eps(q) with c1 >= dim/2, which breaks the 1e-6 gate for
//                   small groups at dim 32; the f32 kernel (dim <= 32) stays available behind MSC_SCORE_NIW_F32
//                   at twice the matrix rate.
//   k_score_niw64_lag   dim <= 32 since round 5: the same contraction with as few vector instructions as it takes --
//                   on gfx950 a vector instruction beside the f64 matrix instruction costs its full issue time
//                   (tools/microbench/mfma_f64_shadow.hip); k_score_niw64_wide: dims 33 .. 128 likewise (k_score_niw64
//                   stays for operand streams beyond 2 GiB, where the buffer loads' 32-bit offsets end)
//   k_niw_bucket_*, k_niw_group_sums   sum_x, sum_xxT by group: rows bucketed by group, summed in registers
//
// Leave-one-out needs no second factorisation: with u = x - mu_n, t = u^T Psi_n^-1 u,
// c = kappa_n/(kappa_n-1):  Psi' = Psi_n - c u u^T,  det Psi' = det Psi_n (1 - c t),
// u^T Psi'^-1 u = t / (1 - c t)  (Sherman-Morrison), x - mu' = c u, hence
//   score_loo = A_k + B_k * log1p(-C_k q),   q = |W_k u|^2 = t kappa_n/(kappa_n+1),
//   C_k = (kappa_n+1)/(kappa_n-1),  B_k = (dof-1+d-1)/2,
//   A_k = lgamma((dof-1+d)/2) - lgamma((dof-1)/2) - (d/2) ln((dof-1) pi)
//         - logdet(Psi_n)/2 - (d/2) ln(kappa_n/((kappa_n-1)(dof-1)))
#include <algorithm>
#include <cstdlib>

#include "family_math.hpp"
#include "launchers.hpp"
#include "score_block.hpp"

namespace msc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// (NIW_* table rows and kNiwPad: msc_internal.hpp)

// hp layout: {kappa, nu, mu[d], psi[d*d]}; raw_f32 per group: {sum_x[d], sum_xxT[d*d]}
// LDS: A (Psi_n -> L) and Li (L^-1) as packed lower triangles, tri(i, j) = i (i + 1) / 2 + j, then mu_n[d], (W mu)[d]
MSC_DEV size_t tri(uint32_t i, uint32_t j) { return (size_t)i * (i + 1u) / 2u + j; }
__global__ __launch_bounds__(64) void k_niw_prepare(const FeatDesc *__restrict__ feats, uint32_t f,
                                                     uint32_t K, uint32_t kpad) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const FeatDesc fd = feats[f];
  const uint32_t d = fd.dim, k = blockIdx.x;     // k == K: the prior (n = 0), for score_data
  const int t = threadIdx.x;
  const size_t ntri = (size_t)d * (d + 1u) / 2u;
  double *A = reinterpret_cast<double *>(smem);   // packed: Psi_n -> L
  double *Li = A + ntri;                          // packed: L^-1
  double *mun = Li + ntri;                        // [d]
  double *wmu = mun + d;                          // [d] (L^-1 mu_n)
  const double kappa = fd.hp[0], nu = fd.hp[1];
  const float *mu = fd.hp + 2, *psi = fd.hp + 2 + d;
  const bool prior = k >= K;
  const double n = prior ? 0.0 : (double)fd.raw_u32[k];
  const float *sx = fd.raw_f32 + (size_t)(prior ? 0 : k) * (d + (size_t)d * d), *sxx = sx + d;
  const double kn = kappa + n, nun = nu + n;
  for (uint32_t i = t; i < d; i += 64) mun[i] = (kappa * (double)mu[i] + (prior ? 0.0 : (double)sx[i])) / kn;
  __syncthreads();
  for (uint32_t idx = t; idx < d * d; idx += 64) {
    const uint32_t i = idx / d, j = idx - i * d;
    if (j > i) continue;
    A[tri(i, j)] = (double)psi[idx] + (prior ? 0.0 : (double)sxx[idx]) + kappa * (double)mu[i] * (double)mu[j] -
                   kn * mun[i] * mun[j];
  }
  __syncthreads();
  // right-looking Cholesky, lower triangle
  double logdet = 0;
  for (uint32_t j = 0; j < d; j++) {
    const double ljj = sqrt(A[tri(j, j)]);
    logdet += 2.0 * log(ljj);
    __syncthreads();
    for (uint32_t i = j + 1 + t; i < d; i += 64) A[tri(i, j)] /= ljj;
    if (t == 0) A[tri(j, j)] = ljj;
    __syncthreads();
    const uint32_t m = d - j - 1;
    for (uint32_t idx = t; idx < m * m; idx += 64) {
      const uint32_t i = j + 1 + idx / m, c = j + 1 + idx % m;
      if (c <= i) A[tri(i, c)] -= A[tri(i, j)] * A[tri(c, j)];
    }
    __syncthreads();
  }
  // L^-1 by columns: lane c solves L y = e_c
  for (uint32_t c = t; c < d; c += 64) {
    Li[tri(c, c)] = 1.0 / A[tri(c, c)];
    for (uint32_t i = c + 1; i < d; i++) {
      double s = 0;
      for (uint32_t m = c; m < i; m++) s += A[tri(i, m)] * Li[tri(m, c)];
      Li[tri(i, c)] = -s / A[tri(i, i)];
    }
  }
  __syncthreads();
  if (prior) {
    if (t == 0) fd.tab[(size_t)NIW_ROWS * kpad] = (float)logdet;       // ln det Psi (hi)
    if (t == 1) fd.tab[(size_t)NIW_ROWS * kpad + 1] = (float)(logdet - (double)(float)logdet);
    return;
  }
  const double scale = sqrt(kn / (kn + 1.0));
  // the operand stream of the f64 kernel: every slot of every chunk (zeros above the diagonal and beyond dim)
  {
    double *W = fd.niw_w64 + (size_t)k * niw_w_stream(d);
    const uint32_t nb = niw_blocks(d);
    for (uint32_t b = 0; b < nb; b++)
      for (uint32_t s4 = 0; s4 <= b; s4++) {
        double *chunk = W + (size_t)(b * (b + 1u) / 2u + s4) * 256u;
        for (uint32_t slot = t; slot < 256u; slot += 64) {
          const uint32_t lane = slot >> 2, e = slot & 3u, i = 16u * b + (lane & 15u), j = 16u * s4 + 4u * e + (lane >> 4);
          chunk[slot] = (i < d && j <= i) ? Li[tri(i, j)] * scale : 0.0;
        }
      }
    // b = W mu_n in the accumulator layout (register r of the lanes with kk = i % 4 holds row 4 r + kk of a block), so
    // that W x - b starts from acc = -b
    for (uint32_t i = t; i < d; i += 64) {
      double bsum = 0.0;
      for (uint32_t j = 0; j <= i; j++) bsum += Li[tri(i, j)] * mun[j];
      wmu[i] = bsum * scale;
    }
    __syncthreads();
    double *B = fd.niw_mu64 + (size_t)k * niw_b_stream(d);
    for (uint32_t slot = t; slot < nb * 256u; slot += 64) {
      const uint32_t b = slot >> 8, lane = (slot >> 2) & 63u, r = slot & 3u, i = 16u * b + 4u * r + (lane >> 4);
      B[slot] = i < d ? wmu[i] : 0.0;
    }
  }
  if (d <= (uint32_t)kNiwPad && fd.niw_w != nullptr) {            // operands of the f32 kernel
    float *W = fd.niw_w + (size_t)k * kNiwPad * kNiwPad;
    for (uint32_t idx = t; idx < kNiwPad * kNiwPad; idx += 64) {
      const uint32_t i = idx / kNiwPad, j = idx % kNiwPad;
      W[idx] = (i < d && j <= i) ? (float)(Li[tri(i, j)] * scale) : 0.0f;
    }
    float *B = fd.niw_b + (size_t)k * 2 * kNiwPad;      // mu hi[32] | lo[32]
    for (uint32_t i = t; i < kNiwPad; i += 64) {
      float hi = 0.f, lo = 0.f;
      if (i < d) split_hi_lo(mun[i], hi, lo);
      B[i] = hi;
      B[kNiwPad + i] = lo;
    }
  }
  if (t == 0) {
    const double dd = d, dof = nun - dd + 1.0;
    const double s = (kn + 1.0) / (kn * dof);
    const double logdet_sigma = logdet + dd * log(s);
    fd.tab[(size_t)NIW_C0 * kpad + k] =
        (float)(lgamma(0.5 * (dof + dd)) - lgamma(0.5 * dof) - 0.5 * dd * log(dof * kPi) - 0.5 * logdet_sigma);
    fd.tab[(size_t)NIW_C1 * kpad + k] = (float)(0.5 * (dof + dd));
    split_hi_lo(logdet, fd.tab[(size_t)NIW_LOGDET_HI * kpad + k], fd.tab[(size_t)NIW_LOGDET_LO * kpad + k]);
    double a_loo = 0, b_loo = 0, c_loo = 0;
    if (n >= 1.0) {
      const double k1 = kn - 1.0, dof1 = dof - 1.0;
      a_loo = lgamma(0.5 * (dof1 + dd)) - lgamma(0.5 * dof1) - 0.5 * dd * log(dof1 * kPi) - 0.5 * logdet -
              0.5 * dd * log(kn / (k1 * dof1));
      b_loo = 0.5 * (dof1 + dd - 1.0);
      c_loo = (kn + 1.0) / k1;
    }
    fd.tab[(size_t)NIW_A_LOO * kpad + k] = (float)a_loo;
    fd.tab[(size_t)NIW_B_LOO * kpad + k] = (float)b_loo;
    fd.tab[(size_t)NIW_C_LOO * kpad + k] = (float)c_loo;
    double *c64 = fd.niw_c64 + (size_t)k * 8;     // the same constants, unrounded, for k_score_niw64
    c64[0] = lgamma(0.5 * (dof + dd)) - lgamma(0.5 * dof) - 0.5 * dd * log(dof * kPi) - 0.5 * logdet_sigma;
    c64[1] = 0.5 * (dof + dd);
    c64[2] = a_loo; c64[3] = b_loo; c64[4] = c_loo;
  }
}

// score_data (distributions niw; special.hpp:13-22 for the multivariate gamma)
MSC_DEV double lmultigamma(uint32_t d, double a) {
  double t = 0.25 * (double)(d * (d - 1)) * kLogPi;
  for (uint32_t j = 1; j <= d; j++) t += lgamma(a + 0.5 * (1.0 - (double)j));
  return t;
}
__global__ __launch_bounds__(256) void k_niw_score_data(const FeatDesc *__restrict__ feats, uint32_t f,
                                                         uint32_t K, uint32_t kpad, float *__restrict__ out) {
  const FeatDesc fd = feats[f];
  const uint32_t k = blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  const double d = fd.dim, kappa = fd.hp[0], nu = fd.hp[1], n = fd.raw_u32[k];
  const double ld0 = (double)fd.tab[(size_t)NIW_ROWS * kpad] + (double)fd.tab[(size_t)NIW_ROWS * kpad + 1];
  const double ldn = (double)fd.tab[(size_t)NIW_LOGDET_HI * kpad + k] + (double)fd.tab[(size_t)NIW_LOGDET_LO * kpad + k];
  const double kn = kappa + n, nun = nu + n;
  out[(size_t)f * K + k] = n == 0.0 ? 0.0f
      : (float)(lmultigamma(fd.dim, 0.5 * nun) - lmultigamma(fd.dim, 0.5 * nu) + 0.5 * nu * ld0 -
                0.5 * nun * ldn + 0.5 * d * log(kappa / kn) - 0.5 * n * d * kLogPi);
}

// ---------------------------------------------------------------------------
// score: a wave owns T tiles of 32 rows and walks all groups.
// MFMA 32x32x2 operand maps (cdna_hip_programming.md section 3): lane l = (r = l & 31, h = l >> 5)
//   A[i = r][kk = h], B[kk = h][j = r], D[i = (reg&3) + 8 (reg>>2) + 4 h][j = r].
// The contraction order over the 32 features is free, so step s pairs features
// (s, 16 + s): lane (r, h) then needs the 16 *contiguous* floats W[r][16h .. 16h+15] and
// x[row r][16h .. 16h+15] -- four 16-byte loads each, whole 128-B rows per lane pair.
// ---------------------------------------------------------------------------
template <int T, bool LOO, bool ACCUM>
__global__ __launch_bounds__(256) void k_score_niw(const FeatDesc *__restrict__ feats, uint32_t f,
                                                    uint32_t K, uint32_t kpad, uint64_t row0,
                                                    uint64_t nrows, const int32_t *__restrict__ z,
                                                    float *__restrict__ out, uint64_t ld) {
  const FeatDesc fd = feats[f];
  const uint32_t d = fd.dim;
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const uint64_t nblocks = (nrows + 32 * T - 1) / (32 * T);
  const uint64_t wave_id = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint64_t nwaves = (uint64_t)gridDim.x * 4;
  const float *X = reinterpret_cast<const float *>(fd.col);
  const bool vec_ok = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  for (uint64_t blk = wave_id; blk < nblocks; blk += nwaves) {
    const uint64_t rb = blk * 32 * T;
    float xr[T][16];
    int gz[T];
    bool live[T], msk[T];
#pragma unroll
    for (int t = 0; t < T; t++) {
      const uint64_t row = rb + 32 * t + r;          // relative to row0
      live[t] = row < nrows;
      const float *xp = X + (row0 + (live[t] ? row : 0)) * d + 16 * h;
      if (d == 32) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const float4 v = live[t] ? ld4(xp + 4 * q) : make_float4(0, 0, 0, 0);
          xr[t][4 * q] = v.x; xr[t][4 * q + 1] = v.y; xr[t][4 * q + 2] = v.z; xr[t][4 * q + 3] = v.w;
        }
      } else {
#pragma unroll
        for (int s = 0; s < 16; s++) xr[t][s] = (live[t] && (uint32_t)(16 * h + s) < d) ? xp[s] : 0.0f;
      }
      gz[t] = (LOO && live[t]) ? z[row] : -1;
      msk[t] = false;
      if (live[t] && fd.mask != nullptr)
        for (uint32_t e = 0; e < d; e++) msk[t] |= fd.mask[(row0 + row) * d + e] != 0;
    }
    float qkeep[T][4];           // |W(x - mu)|^2 of the 4 groups this lane finalises per batch of 8
    for (uint32_t k = 0; k < K; k++) {
      const float *Wk = fd.niw_w + ((size_t)k * kNiwPad + r) * kNiwPad + 16 * h;
      const float *Bk = fd.niw_b + (size_t)k * 2 * kNiwPad + 16 * h;
      float w[16], mh[16], ml[16];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const float4 a = ld4(Wk + 4 * q), b = ld4(Bk + 4 * q), c = ld4(Bk + kNiwPad + 4 * q);
        w[4 * q] = a.x; w[4 * q + 1] = a.y; w[4 * q + 2] = a.z; w[4 * q + 3] = a.w;
        mh[4 * q] = b.x; mh[4 * q + 1] = b.y; mh[4 * q + 2] = b.z; mh[4 * q + 3] = b.w;
        ml[4 * q] = c.x; ml[4 * q + 1] = c.y; ml[4 * q + 2] = c.z; ml[4 * q + 3] = c.w;
      }
      const uint32_t slot = k & 3;
      const bool mine = (int)((k >> 2) & 1) == h;
#pragma unroll
      for (int t = 0; t < T; t++) {
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 16; s++)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[s], (xr[t][s] - mh[s]) - ml[s], acc, 0, 0, 0);
        float qp = 0.f;
#pragma unroll
        for (int i = 0; i < 16; i++) qp = fmaf(acc[i], acc[i], qp);
        const float q = qp + __shfl_xor(qp, 32, 64);
        if (mine) {
          if (slot == 0) qkeep[t][0] = q;
          else if (slot == 1) qkeep[t][1] = q;
          else if (slot == 2) qkeep[t][2] = q;
          else qkeep[t][3] = q;
        }
      }
      if ((k & 7) == 7 || k == K - 1) {
        const uint32_t k0 = (k & ~7u) + 4 * h;       // first group of this lane's float4
        float4 pend[T];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const uint32_t kg = k0 + i;
          const bool valid = kg <= k;
          const size_t kc = valid ? kg : 0;
          const float c0 = fd.tab[(size_t)NIW_C0 * kpad + kc], c1 = fd.tab[(size_t)NIW_C1 * kpad + kc];
#pragma unroll
          for (int t = 0; t < T; t++) {
            const float q = valid ? qkeep[t][i] : 0.f;
            float sc = fmaf(-c1, log1p_acc(q), c0);
            if (LOO && valid && gz[t] == (int)kg) {
              const float y = fminf(fd.tab[(size_t)NIW_C_LOO * kpad + kc] * q, 0.99999994f);
              sc = fmaf(fd.tab[(size_t)NIW_B_LOO * kpad + kc], log1p_acc(-y), fd.tab[(size_t)NIW_A_LOO * kpad + kc]);
            }
            if (msk[t]) sc = 0.f;
            if (i == 0) pend[t].x = sc;
            else if (i == 1) pend[t].y = sc;
            else if (i == 2) pend[t].z = sc;
            else pend[t].w = sc;
          }
        }
#pragma unroll
        for (int t = 0; t < T; t++) {
          if (!live[t] || k0 > k) continue;
          float *p = out + (rb + 32 * t + r) * ld + k0;
          if (vec_ok && k0 + 3 <= k) {
            float4 v = pend[t];
            if (ACCUM) { const float4 o = *reinterpret_cast<const float4 *>(p); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            *reinterpret_cast<float4 *>(p) = v;
          } else {
            const float vals[4] = {pend[t].x, pend[t].y, pend[t].z, pend[t].w};
#pragma unroll
            for (int i = 0; i < 4; i++)
              if (k0 + i <= k) p[i] = ACCUM ? p[i] + vals[i] : vals[i];
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// f64 matrix pipe.  v_mfma_f64_16x16x4_f64 maps (cdna_hip_programming.md section 3): lane
// l = (c = l & 15, kk = l >> 4): A[i = c][k = kk], B[k = kk][j = c],
// D[i = kk + 4 reg][j = c], reg in [0,4).  i = whitened component within a 16-block, j = data row.
// W_k is lower triangular, so component block b (rows 16b .. 16b+15) only meets feature blocks 0 .. b: a chunk (b, s4)
// is four steps, step e contracting features 16 s4 + 4 e + kk -- lane (c, kk) holds x[row c][4 s + kk] for every step
// s, as floats, widened on the way into the matrix instruction (one v_cvt per 64-cycle MFMA).  The A operands of a
// chunk arrive as one 32-byte read per lane from the stream k_niw_prepare wrote in exactly this order, the
// accumulators of a block start from -(W mu) in their own layout.  dim 32: 12 matrix instructions per (16 rows,
// group) where the full square took 16.
// ---------------------------------------------------------------------------
typedef double f64x4 __attribute__((ext_vector_type(4)));

MSC_DEV double shfl_xor_f64(double v, int mask) {
  const int hi = __shfl_xor(__double2hiint(v), mask, 64), lo = __shfl_xor(__double2loint(v), mask, 64);
  return __hiloint2double(hi, lo);
}

// v + v[lane ^ 16], then that + its [lane ^ 32]: gfx950's v_permlane16_swap / v_permlane32_swap exchange the odd rows of
// one register with the even rows of another, so (x, x) comes back as (the even partner, the odd partner) of every lane
// -- two VALU instructions per 32-bit half where ds_bpermute needs an address and an LDS round trip.  Same additions
// in the same order as the shuffles it replaces.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
MSC_DEV double xsum_rows_f64(double v) {
  u32x2 a = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
  u32x2 b = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
  const double s = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
  a = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(s), (unsigned)__double2loint(s), false, false);
  b = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(s), (unsigned)__double2hiint(s), false, false);
  return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}

// One step of a reduce-scatter over lane rows: the rows of a pair (ROWS = 16: rows 0|1 and 2|3; ROWS = 32: rows 0,1 | 2,3)
// add their x's into the first of them and their y's into the second -- one swap per 32-bit half and one add.
template <int ROWS>
MSC_DEV double pair_rows_f64(double x, double y) {
  u32x2 lo, hi;
  if (ROWS == 16) {
    lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
    hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  } else {
    lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
    hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  }
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}

// log(u) for u >= 1 in double without the library routine's branches and tables: u = m 2^e with m in [1/sqrt2, sqrt2),
// log m = 2 atanh(s), s = (m - 1) / (m + 1), |s| <= 0.1716, as an odd series to s^15 (next term 2 s^17 / 17 < 2e-14);
// the quotient through v_rcp_f64 and two Newton steps.  ~30 double instructions.
MSC_DEV double log_ge1_f64(double u) {
  int hi = __double2hiint(u);
  int e = (hi >> 20) - 1023;
  double m = __hiloint2double((hi & 0x000fffff) | 0x3ff00000, __double2loint(u));      // [1, 2)
  if (m > 1.4142135623730951) {
    m *= 0.5;
    e += 1;
  }
  const double den = m + 1.0;
  double y = __builtin_amdgcn_rcp(den);
  y = y * fma(-den, y, 2.0);
  y = y * fma(-den, y, 2.0);
  const double sq = (m - 1.0) * y, s2 = sq * sq;
  double p = 1.0 / 15.0;
  p = fma(p, s2, 1.0 / 13.0);
  p = fma(p, s2, 1.0 / 11.0);
  p = fma(p, s2, 1.0 / 9.0);
  p = fma(p, s2, 1.0 / 7.0);
  p = fma(p, s2, 1.0 / 5.0);
  p = fma(p, s2, 1.0 / 3.0);
  p = fma(p, s2, 1.0);
  return fma((double)e, 0.69314718055994530942, 2.0 * sq * p);
}
// score = c0 - c1 log1p(q) of the MFMA kernel, all in double.  A float logarithm (1e-7 of the term) is not enough here:
// with tiny groups in high dimension c0 and the term cancel (c0 = +130, c1 log1p(q) = 129.2 at dim 48 with five rows in
// the group), and 1e-7 of the term is 1e-5 of the score.  The series above costs ~2 % of the matrix work it follows; a
// branch to the library log1p for the cancelling lanes only cost 25 % (C4 1.01 -> 1.27 ms: its registers).
MSC_DEV double niw_score_from_q(double c0, double c1, double q) { return c0 - c1 * log_ge1_f64(1.0 + q); }

// The end of a batch of 16 groups (or of the table): lane (c, kk) turns its 4 kept q's into the scores of groups
// k0 .. k0 + 3 of its row of every block and stores them as one float4.
template <int JB, bool LOO, bool ACCUM, bool FULL = false>
MSC_DEV void niw64_finish_batch(const FeatDesc &fd, uint32_t k, int kk, int c, uint64_t rb, const double (&qkeep)[JB][4],
                                const int (&gz)[JB], const bool (&msk)[JB], const bool (&live)[JB],
                                double *__restrict__ qown, float *__restrict__ out, uint64_t ld, bool vec_ok) {
  const uint32_t k0 = (k & ~15u) + 4 * kk;
  float4 pend[JB];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const uint32_t kg = k0 + i;
    const bool valid = FULL || kg <= k;       // FULL: a whole batch of 16, no lane has a slot to skip
    const double *c64 = fd.niw_c64 + (size_t)(valid ? kg : 0) * 8;
    const double c0 = c64[0], c1 = c64[1];
#pragma unroll
    for (int jb = 0; jb < JB; jb++) {
      const double q = valid ? qkeep[jb][i] : 0.0;
      double sc = niw_score_from_q(c0, c1, q);
      if (LOO && valid && gz[jb] == (int)kg) {       // the own group: its value comes from k_niw_loo_patch
        qown[rb + 16 * jb + c] = q;
        sc = 0.0;
      }
      if (msk[jb]) sc = 0.0;
      const float scf = (float)sc;
      if (i == 0) pend[jb].x = scf;
      else if (i == 1) pend[jb].y = scf;
      else if (i == 2) pend[jb].z = scf;
      else pend[jb].w = scf;
    }
  }
#pragma unroll
  for (int jb = 0; jb < JB; jb++) {
    if (!live[jb] || (!FULL && k0 > k)) continue;
    float *p = out + (rb + 16 * jb + c) * ld + k0;
    if (vec_ok && (FULL || k0 + 3 <= k)) {
      float4 v = pend[jb];
      if (ACCUM) { const float4 o = *reinterpret_cast<const float4 *>(p); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
      *reinterpret_cast<float4 *>(p) = v;
    } else {
      const float vals[4] = {pend[jb].x, pend[jb].y, pend[jb].z, pend[jb].w};
#pragma unroll
      for (int i = 0; i < 4; i++)
        if (k0 + i <= k) p[i] = ACCUM ? p[i] + vals[i] : vals[i];
    }
  }
}

// NB = 16-blocks of the dimension (1 .. 8), JB = 16-row blocks a wave carries
template <int NB, int JB, bool LOO, bool ACCUM>
__global__ __launch_bounds__(256) void k_score_niw64(const FeatDesc *__restrict__ feats, uint32_t f,
                                                      uint32_t K, uint32_t kpad, uint64_t row0,
                                                      uint64_t nrows, const int32_t *__restrict__ z,
                                                      double *__restrict__ qown, float *__restrict__ out, uint64_t ld) {
  constexpr int NS = 4 * NB;                                   // steps of the padded dimension
  constexpr int NCH = NB * (NB + 1) / 2;
  const FeatDesc fd = feats[f];
  const uint32_t d = fd.dim;
  const int lane = threadIdx.x & 63, c = lane & 15, kk = lane >> 4;
  const uint64_t nblocks = (nrows + 16 * JB - 1) / (16 * JB);
  const uint64_t wave_id = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint64_t nwaves = (uint64_t)gridDim.x * 4;
  const float *X = reinterpret_cast<const float *>(fd.col);
  const bool vec_ok = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  // (round 5: giving the two waves of a SIMD different priorities (s_setprio by the parity of their slot) or starting one
  // of them 2000 cycles late changed nothing -- 0.98-1.00 ms either way, profiles/r05_c4_notes.txt: they are not in lockstep)
  for (uint64_t blk = wave_id; blk < nblocks; blk += nwaves) {
    const uint64_t rb = blk * 16 * JB;
    float xf[JB][NS];            // this lane's feature of every step, for its row of each block
    int gz[JB];
    bool live[JB], msk[JB];
#pragma unroll
    for (int jb = 0; jb < JB; jb++) {
      const uint64_t row = rb + 16 * jb + c;
      live[jb] = row < nrows;
      const float *xp = X + (row0 + (live[jb] ? row : 0)) * d + kk;
#pragma unroll
      for (int s = 0; s < NS; s++) xf[jb][s] = (live[jb] && (uint32_t)(4 * s + kk) < d) ? xp[4 * s] : 0.0f;
      gz[jb] = (LOO && live[jb]) ? z[row] : -1;
      msk[jb] = false;
      if (live[jb] && fd.mask != nullptr)
        for (uint32_t e = 0; e < d; e++) msk[jb] |= fd.mask[(row0 + row) * d + e] != 0;
    }
    double qkeep[JB][4];         // |W(x - mu)|^2 of the 4 groups this lane finalises per batch of 16
    for (uint32_t k = 0; k < K; k++) {
      const double *Wk = fd.niw_w64 + (size_t)k * (NCH * 256) + lane * 4;
      const double *Bk = fd.niw_mu64 + (size_t)k * (NB * 256) + lane * 4;
      double qp[JB];
#pragma unroll
      for (int jb = 0; jb < JB; jb++) qp[jb] = 0.0;
      // Beyond dim 32 the group's operand stream is NB (NB + 1) / 2 chunks of 2 KiB; left to itself the compiler hoists
      // every chunk's load to the top of the group (80 registers of operands at dim 64, 288 at dim 128) and the kernel
      // runs one wave per SIMD.  PIPE: chunk c + 1 is fetched when chunk c starts to multiply and a scheduling
      // barrier after every chunk keeps it at that -- two chunks of operands live, whatever the dimension.
      constexpr bool PIPE = NB > 2;
      double2 n01 = {0.0, 0.0}, n23 = {0.0, 0.0};
      if (PIPE) {
        n01 = *reinterpret_cast<const double2 *>(Wk);
        n23 = *reinterpret_cast<const double2 *>(Wk + 2);
      }
#pragma unroll
      for (int b = 0; b < NB; b++) {
        const double2 b01 = *reinterpret_cast<const double2 *>(Bk + b * 256), b23 = *reinterpret_cast<const double2 *>(Bk + b * 256 + 2);
        f64x4 acc[JB];
#pragma unroll
        for (int jb = 0; jb < JB; jb++) acc[jb] = f64x4{-b01.x, -b01.y, -b23.x, -b23.y};   // W x - W mu
#pragma unroll
        for (int s4 = 0; s4 <= b; s4++) {
          constexpr int kLast = NCH - 1;
          const int ch = b * (b + 1) / 2 + s4;
          const double *wc = Wk + ch * 256;
          double2 a01, a23;
          if (PIPE) {
            a01 = n01;
            a23 = n23;
            if (ch < kLast) {
              n01 = *reinterpret_cast<const double2 *>(wc + 256);
              n23 = *reinterpret_cast<const double2 *>(wc + 256 + 2);
            }
          } else {
            a01 = *reinterpret_cast<const double2 *>(wc);
            a23 = *reinterpret_cast<const double2 *>(wc + 2);
          }
          const double a[4] = {a01.x, a01.y, a23.x, a23.y};
#pragma unroll
          for (int e = 0; e < 4; e++) {
            // (steps of a partial last block whose features lie beyond dim multiply zeros: a run-time skip cuts every
            //  chunk into basic blocks and costs far more than the <= 3 spare steps -- C4 with the skip 1.46 ms, without 1.01)
#pragma unroll
            for (int jb = 0; jb < JB; jb++) {
              float xv = xf[jb][4 * s4 + e];
              // beyond dim 32 the widened copies of a wave's features (4 NB doubles per row block) would not fit
              // the register file if they were made once per row block: the value is opaque here, so the
              // conversion stays next to its matrix instruction (one v_cvt_f64_f32 per 64-cycle MFMA)
              if (NB > 2) asm volatile("" : "+v"(xv));
              acc[jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[e], (double)xv, acc[jb], 0, 0, 0);
            }
          }
          if (PIPE) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int jb = 0; jb < JB; jb++)
#pragma unroll
          for (int i = 0; i < 4; i++) qp[jb] = fma(acc[jb][i], acc[jb][i], qp[jb]);
      }
      const uint32_t slot = k & 3;
      const bool mine = (int)((k >> 2) & 3) == kk;
#pragma unroll
      for (int jb = 0; jb < JB; jb++) {
        const double q = xsum_rows_f64(qp[jb]);       // over the four lanes c, c + 16, c + 32, c + 48
        if (mine) {
          if (slot == 0) qkeep[jb][0] = q;
          else if (slot == 1) qkeep[jb][1] = q;
          else if (slot == 2) qkeep[jb][2] = q;
          else qkeep[jb][3] = q;
        }
      }
      if ((k & 15) == 15 || k == K - 1) {
        // finish the batch: lane (c, kk) turns its 4 kept q's into scores of groups k0 .. k0+3
        const uint32_t k0 = (k & ~15u) + 4 * kk;
        float4 pend[JB];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const uint32_t kg = k0 + i;
          const bool valid = kg <= k;
          const double *c64 = fd.niw_c64 + (size_t)(valid ? kg : 0) * 8;
          const double c0 = c64[0], c1 = c64[1];
#pragma unroll
          for (int jb = 0; jb < JB; jb++) {
            const double q = valid ? qkeep[jb][i] : 0.0;
            double sc = niw_score_from_q(c0, c1, q);
            if (LOO && valid && gz[jb] == (int)kg) {       // the own group: its value comes from k_niw_loo_patch
              qown[rb + 16 * jb + c] = q;
              sc = 0.0;
            }
            if (msk[jb]) sc = 0.0;
            const float scf = (float)sc;
            if (i == 0) pend[jb].x = scf;
            else if (i == 1) pend[jb].y = scf;
            else if (i == 2) pend[jb].z = scf;
            else pend[jb].w = scf;
          }
        }
#pragma unroll
        for (int jb = 0; jb < JB; jb++) {
          if (!live[jb] || k0 > k) continue;
          float *p = out + (rb + 16 * jb + c) * ld + k0;
          if (vec_ok && k0 + 3 <= k) {
            float4 v = pend[jb];
            if (ACCUM) { const float4 o = *reinterpret_cast<const float4 *>(p); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            *reinterpret_cast<float4 *>(p) = v;
          } else {
            const float vals[4] = {pend[jb].x, pend[jb].y, pend[jb].z, pend[jb].w};
#pragma unroll
            for (int i = 0; i < 4; i++)
              if (k0 + i <= k) p[i] = ACCUM ? p[i] + vals[i] : vals[i];
          }
        }
      }
    }
  }
}

// dim <= 32, the same arithmetic with as few vector instructions as it takes.  On gfx950 the f64 matrix instruction
// has no shadow: a vector instruction issued next to it (same wave or the SIMD's other wave) adds its own ~4 cycles
// (f64), ~2.5 (f32), ~1.4 (v_mov), ~9 (v_permlane*_swap) to the 64 of the matrix instruction
// (tools/microbench/mfma_f64_shadow.hip) -- the kernel's time is its matrix instructions PLUS its vector instructions, and
// k_score_niw64 above spends ~216 of them a group.  Here:
//   * the accumulators of a block start as the matrix instruction's C operand: acc = W (-x) + W mu, the features negated
//     once per row block -- no copies of -(W mu) into four accumulator sets (64 v_mov a group), no subtraction;
//   * the sum over the four lanes of a row is a reduce-scatter over FOUR groups: the groups of a batch of 16 are taken in
//     the order s, 4 + s, 8 + s, 12 + s (s = 0 .. 3), and of such a set lane row kk ends with the total of group 4 kk + s --
//     the one it finishes and stores -- after 3 swaps + 1.5 adds a group and row block where the all-to-all sum took
//     4 swaps + 2 adds + the selects that filed it.  (The last K mod 16 groups go in their natural order with the
//     all-to-all sum: a set of theirs may belong to one lane row.)
//   * a unit's operands (dim > 16: block 0 = one chunk + W mu, block 1 = two chunks + W mu; dim <= 16: the group) are
//     fetched into the registers its matrix instructions just read, a group (dim <= 16: two) ahead of their use; units
//     alternate between two accumulator sets and a unit's squares follow the next unit's matrix instructions.
template <int NB, bool LOO, bool ACCUM>
__global__ __launch_bounds__(256, 2) void k_score_niw64_lag(const FeatDesc *__restrict__ feats, uint32_t f,
                                                             uint32_t K, uint32_t kpad, uint64_t row0,
                                                             uint64_t nrows, const int32_t *z,
                                                             double *__restrict__ qown, float *__restrict__ out, uint64_t ld) {
  static_assert(NB == 1 || NB == 2, "dim <= 32");
  constexpr int JB = 4, NS = 4 * NB, NCH = NB * (NB + 1) / 2;
  const FeatDesc fd = feats[f];
  const uint32_t d = fd.dim;
  const int lane = threadIdx.x & 63, c = lane & 15, kk = lane >> 4;
  const uint64_t nblocks = (nrows + 16 * JB - 1) / (16 * JB);
  const uint64_t wave_id = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint64_t nwaves = (uint64_t)gridDim.x * 4;
  const float *X = reinterpret_cast<const float *>(fd.col);
  const bool vec_ok = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  // the operand streams as buffers: a group's offset is a scalar, the lane's 32 bytes of a chunk a constant register
  typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t Wb = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(fd.niw_w64), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t Bb = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(fd.niw_mu64), 0, 0x7fffffff, 0x00020000);
  const uint32_t lane_off = lane * 32;
  const uint32_t Kfull = K & ~15u;
  // the group taken at time t (clamped past the table: the last group again, into slots the batch finish skips)
  auto group_at = [&](uint32_t t) -> uint32_t {
    const uint32_t g = t < Kfull ? (t & ~15u) + 4 * (t & 3) + ((t >> 2) & 3) : t;
    return g < K ? g : K - 1;
  };
  for (uint64_t blk = wave_id; blk < nblocks; blk += nwaves) {
    const uint64_t rb = blk * 16 * JB;
    float xn[JB][NS];            // MINUS this lane's feature of every step, for its row of each block
    int gz[JB];
    bool live[JB], msk[JB];
#pragma unroll
    for (int jb = 0; jb < JB; jb++) {
      const uint64_t row = rb + 16 * jb + c;
      live[jb] = row < nrows;
      const float *xp = X + (row0 + (live[jb] ? row : 0)) * d;
#pragma unroll
      for (int s = 0; s < NS; s++) {
        const bool has = live[jb] && (uint32_t)(4 * s + kk) < d;
        const float v = xp[has ? 4 * s + kk : 0];
        xn[jb][s] = has ? -v : 0.0f;
      }
      gz[jb] = (LOO && live[jb]) ? z[row] : -1;
      msk[jb] = false;
      if (live[jb] && fd.mask != nullptr)
        for (uint32_t e = 0; e < d; e++) msk[jb] |= fd.mask[(row0 + row) * d + e] != 0;
    }
    double qkeep[JB][4];
    // operands.  dim > 16: w0 = chunk (0, 0); w1, w2 = chunks (1, 0), (1, 1); m0, m1 = W mu of the two blocks, in
    // accumulator layout.  dim <= 16: a group is one unit -- w0 / m0 serve the groups at even times, w1 / m1 at odd times.
    double2 w0[2], w1[2], w2[2], m0[2], m1[2];
    auto load2 = [&](double2 (&o)[2], __amdgpu_buffer_rsrc_t buf, uint32_t byte) {
      const u32x4v lo = __builtin_amdgcn_raw_buffer_load_b128(buf, lane_off, byte, 0);
      const u32x4v hi = __builtin_amdgcn_raw_buffer_load_b128(buf, lane_off + 16, byte, 0);
      o[0] = double2{__hiloint2double((int)lo[1], (int)lo[0]), __hiloint2double((int)lo[3], (int)lo[2])};
      o[1] = double2{__hiloint2double((int)hi[1], (int)hi[0]), __hiloint2double((int)hi[3], (int)hi[2])};
    };
    // one chunk of a unit; the first starts every accumulator from W mu as the instruction's C operand
    auto chunk = [&](f64x4 (&acc)[JB], const double2 (&w)[2], int s4, const double2 (*m)[2]) {
      const double a[4] = {w[0].x, w[0].y, w[1].x, w[1].y};
#pragma unroll
      for (int e = 0; e < 4; e++)
#pragma unroll
        for (int jb = 0; jb < JB; jb++) {
          const f64x4 from = (m != nullptr && e == 0) ? f64x4{(*m)[0].x, (*m)[0].y, (*m)[1].x, (*m)[1].y} : acc[jb];
          acc[jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[e], (double)xn[jb][4 * s4 + e], from, 0, 0, 0);
        }
    };
    auto squares = [](double (&qp)[JB], const f64x4 (&acc)[JB], bool fresh) {
#pragma unroll
      for (int jb = 0; jb < JB; jb++)
#pragma unroll
        for (int i = 0; i < 4; i++) qp[jb] = (fresh && i == 0) ? acc[jb][i] * acc[jb][i] : fma(acc[jb][i], acc[jb][i], qp[jb]);
    };
    f64x4 accA[JB], accB[JB];
    double qp[JB], held[JB], half[JB];
    if constexpr (NB == 2) {
      const uint32_t g0 = group_at(0), g1 = group_at(1);
      load2(w0, Wb, g0 * (NCH * 2048u));
      load2(m0, Bb, g0 * (NB * 2048u));
      load2(w1, Wb, g0 * (NCH * 2048u) + 2048);
      load2(w2, Wb, g0 * (NCH * 2048u) + 4096);
      load2(m1, Bb, g0 * (NB * 2048u) + 2048);
      chunk(accA, w0, 0, &m0);                                           // unit (group_at(0), 0)
      load2(w0, Wb, g1 * (NCH * 2048u));
      load2(m0, Bb, g1 * (NB * 2048u));
    } else {
      const uint32_t g0 = group_at(0), g1 = group_at(1), g2 = group_at(2);
      load2(w0, Wb, g0 * 2048u);
      load2(m0, Bb, g0 * 2048u);
      load2(w1, Wb, g1 * 2048u);
      load2(m1, Bb, g1 * 2048u);
      chunk(accA, w0, 0, &m0);                                           // the group at time 0
      load2(w0, Wb, g2 * 2048u);
      load2(m0, Bb, g2 * 2048u);
    }
    for (uint32_t t4 = 0; t4 < K; t4 += 4) {        // four groups a trip
      const bool tail = t4 >= Kfull;
      const bool mine = (int)((t4 >> 2) & 3) == kk;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        if constexpr (NB == 2) {
          const uint32_t gn = group_at(t4 + i + 1), gn2 = group_at(t4 + i + 2);
          // block 1 of this group issues, then block 0's squares; block 1's operands of the next group are fetched
          chunk(accB, w1, 0, &m1);
          chunk(accB, w2, 1, nullptr);
          squares(qp, accA, true);
          load2(w1, Wb, gn * (NCH * 2048u) + 2048);
          load2(w2, Wb, gn * (NCH * 2048u) + 4096);
          load2(m1, Bb, gn * (NB * 2048u) + 2048);
          __builtin_amdgcn_sched_barrier(0);
          // block 0 of the next group issues, then block 1's squares
          chunk(accA, w0, 0, &m0);
          squares(qp, accB, false);
          load2(w0, Wb, gn2 * (NCH * 2048u));
          load2(m0, Bb, gn2 * (NB * 2048u));
          __builtin_amdgcn_sched_barrier(0);
        } else {
          // the next group issues into the other accumulator set, then this group's squares; the operands just read are
          // replaced by those of the group two times on
          const uint32_t gn3 = group_at(t4 + i + 3);
          if ((i & 1) == 0) {
            chunk(accB, w1, 0, &m1);
            squares(qp, accA, true);
            load2(w1, Wb, gn3 * 2048u);
            load2(m1, Bb, gn3 * 2048u);
          } else {
            chunk(accA, w0, 0, &m0);
            squares(qp, accB, true);
            load2(w0, Wb, gn3 * 2048u);
            load2(m0, Bb, gn3 * 2048u);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (!tail) {
          // the reduce-scatter: times 0, 1 of a set pair up across lane rows (0,1) and (2,3), times 2, 3 likewise, the
          // two halves across (0,2) and (1,3): row kk ends with the total of time kk's group = 4 kk + s of the batch
          if (i == 0 || i == 2) {
#pragma unroll
            for (int jb = 0; jb < JB; jb++) held[jb] = qp[jb];
          } else {
#pragma unroll
            for (int jb = 0; jb < JB; jb++) {
              const double pr = pair_rows_f64<16>(held[jb], qp[jb]);
              if (i == 1) half[jb] = pr;
              else {
                const double q = pair_rows_f64<32>(half[jb], pr);
                qkeep[jb][0] = qkeep[jb][1];
                qkeep[jb][1] = qkeep[jb][2];
                qkeep[jb][2] = qkeep[jb][3];
                qkeep[jb][3] = q;                     // set s lands in slot s after the four sets of a batch
              }
            }
          }
        } else {
#pragma unroll
          for (int jb = 0; jb < JB; jb++) {
            const double q = xsum_rows_f64(qp[jb]);
            qkeep[jb][i] = mine ? q : qkeep[jb][i];
          }
        }
      }
      const uint32_t kl = t4 + 3 < K ? t4 + 3 : K - 1;
      if ((kl & 15) == 15) niw64_finish_batch<JB, LOO, ACCUM, true>(fd, kl, kk, c, rb, qkeep, gz, msk, live, qown, out, ld, vec_ok);
      else if (kl == K - 1) niw64_finish_batch<JB, LOO, ACCUM>(fd, kl, kk, c, rb, qkeep, gz, msk, live, qown, out, ld, vec_ok);
    }
  }
}

// 32 < dim <= 128 with the same cuts (round 5): two row blocks a wave and the operand stream pipelined by chunk as in
// k_score_niw64<NB, 2> (a chunk's successor -- the next group's first included -- is fetched when the chunk starts to
// multiply: two chunks of operands live whatever the dimension), buffer loads, accumulators started from W mu as the
// C operand against negated features (kept as floats: the conversion stays next to its matrix instruction, the widened
// copies of NB = 8 would be 256 registers), the reduce-scatter lane sums over the groups in k_score_niw64_lag's order and
// its batch finish.  The occupancy is asked for -- three waves a SIMD up to dim 80, two beyond: left to itself the compiler
// took 208 .. 292 registers where 155 .. 203 do without a spill -- and with it N = 256k, K = 128 goes from 1.94 / 3.08 / 4.94 /
// 6.86 / 9.03 / 11.5 ms at dim 48 / 64 / 80 / 96 / 112 / 128 (k_score_niw64<NB, 2>) to 1.75 / 2.80 / 4.11 / 5.94 / 7.83 / 9.95,
// leave-one-out at dim 128 from 17.5 to 9.96.
template <int NB, bool LOO, bool ACCUM>
__global__ __launch_bounds__(256, NB <= 5 ? 3 : 2) void k_score_niw64_wide(const FeatDesc *__restrict__ feats, uint32_t f,
                                                           uint32_t K, uint32_t kpad, uint64_t row0,
                                                           uint64_t nrows, const int32_t *z,
                                                           double *__restrict__ qown, float *__restrict__ out, uint64_t ld) {
  static_assert(NB > 2 && NB <= 8, "32 < dim <= 128");
  constexpr int JB = 2, NS = 4 * NB, NCH = NB * (NB + 1) / 2;
  const FeatDesc fd = feats[f];
  const uint32_t d = fd.dim;
  const int lane = threadIdx.x & 63, c = lane & 15, kk = lane >> 4;
  const uint64_t nblocks = (nrows + 16 * JB - 1) / (16 * JB);
  const uint64_t wave_id = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint64_t nwaves = (uint64_t)gridDim.x * 4;
  const float *X = reinterpret_cast<const float *>(fd.col);
  const bool vec_ok = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t Wb = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(fd.niw_w64), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t Bb = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(fd.niw_mu64), 0, 0x7fffffff, 0x00020000);
  const uint32_t lane_off = lane * 32;
  const uint32_t Kfull = K & ~15u;
  auto group_at = [&](uint32_t t) -> uint32_t {
    const uint32_t g = t < Kfull ? (t & ~15u) + 4 * (t & 3) + ((t >> 2) & 3) : t;
    return g < K ? g : K - 1;
  };
  auto load2 = [&](double2 (&o)[2], __amdgpu_buffer_rsrc_t buf, uint32_t byte) {
    const u32x4v lo = __builtin_amdgcn_raw_buffer_load_b128(buf, lane_off, byte, 0);
    const u32x4v hi = __builtin_amdgcn_raw_buffer_load_b128(buf, lane_off + 16, byte, 0);
    o[0] = double2{__hiloint2double((int)lo[1], (int)lo[0]), __hiloint2double((int)lo[3], (int)lo[2])};
    o[1] = double2{__hiloint2double((int)hi[1], (int)hi[0]), __hiloint2double((int)hi[3], (int)hi[2])};
  };
  for (uint64_t blk = wave_id; blk < nblocks; blk += nwaves) {
    const uint64_t rb = blk * 16 * JB;
    float xn[JB][NS];            // MINUS this lane's feature of every step, for its row of each block
    int gz[JB];
    bool live[JB], msk[JB];
#pragma unroll
    for (int jb = 0; jb < JB; jb++) {
      const uint64_t row = rb + 16 * jb + c;
      live[jb] = row < nrows;
      const float *xp = X + (row0 + (live[jb] ? row : 0)) * d;
#pragma unroll
      for (int s = 0; s < NS; s++) {
        const bool has = live[jb] && (uint32_t)(4 * s + kk) < d;
        const float v = xp[has ? 4 * s + kk : 0];
        xn[jb][s] = has ? -v : 0.0f;
      }
      gz[jb] = (LOO && live[jb]) ? z[row] : -1;
      msk[jb] = false;
      if (live[jb] && fd.mask != nullptr)
        for (uint32_t e = 0; e < d; e++) msk[jb] |= fd.mask[(row0 + row) * d + e] != 0;
    }
    double qkeep[JB][4], held[JB], half[JB];
    double2 nxt[2];                                         // the chunk after the one that multiplies
    load2(nxt, Wb, group_at(0) * (NCH * 2048u));
    const uint32_t Kt = (K + 3u) & ~3u;                     // times: whole sets of four (past the table: the last group again)
    for (uint32_t t = 0; t < Kt; t++) {
      const uint32_t g = group_at(t), gn = group_at(t + 1);
      const int i = (int)(t & 3u);
      const bool tail = t >= Kfull;
      double qp[JB];
#pragma unroll
      for (int jb = 0; jb < JB; jb++) qp[jb] = 0.0;
#pragma unroll
      for (int b = 0; b < NB; b++) {
        double2 m[2];
        load2(m, Bb, g * (NB * 2048u) + b * 2048u);
        const f64x4 mv = {m[0].x, m[0].y, m[1].x, m[1].y};
        f64x4 acc[JB];
#pragma unroll
        for (int s4 = 0; s4 <= b; s4++) {
          constexpr int kLast = NCH - 1;
          const int ch = b * (b + 1) / 2 + s4;
          const double a[4] = {nxt[0].x, nxt[0].y, nxt[1].x, nxt[1].y};
          if (ch < kLast) load2(nxt, Wb, g * (NCH * 2048u) + (ch + 1) * 2048u);
          else load2(nxt, Wb, gn * (NCH * 2048u));
#pragma unroll
          for (int e = 0; e < 4; e++)
#pragma unroll
            for (int jb = 0; jb < JB; jb++) {
              float xv = xn[jb][4 * s4 + e];
              asm volatile("" : "+v"(xv));                  // (the conversion stays next to its matrix instruction)
              acc[jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[e], (double)xv, (s4 == 0 && e == 0) ? mv : acc[jb], 0, 0, 0);
            }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int jb = 0; jb < JB; jb++)
#pragma unroll
          for (int q = 0; q < 4; q++) qp[jb] = fma(acc[jb][q], acc[jb][q], qp[jb]);
      }
      // lane sums: k_score_niw64_lag's reduce-scatter over the four groups of a set (wave-uniform branches on the time
      // within the set); the last K mod 16 groups all-to-all into the slot their time names
      if (!tail) {
        if (i == 0 || i == 2) {
#pragma unroll
          for (int jb = 0; jb < JB; jb++) held[jb] = qp[jb];
        } else {
#pragma unroll
          for (int jb = 0; jb < JB; jb++) {
            const double pr = pair_rows_f64<16>(held[jb], qp[jb]);
            if (i == 1) half[jb] = pr;
            else {
              const double q = pair_rows_f64<32>(half[jb], pr);
              qkeep[jb][0] = qkeep[jb][1];
              qkeep[jb][1] = qkeep[jb][2];
              qkeep[jb][2] = qkeep[jb][3];
              qkeep[jb][3] = q;
            }
          }
        }
      } else {
        const bool mine = (int)((t >> 2) & 3) == kk;
#pragma unroll
        for (int jb = 0; jb < JB; jb++) {
          const double q = xsum_rows_f64(qp[jb]);
          if (mine) {
            if (i == 0) qkeep[jb][0] = q;
            else if (i == 1) qkeep[jb][1] = q;
            else if (i == 2) qkeep[jb][2] = q;
            else qkeep[jb][3] = q;
          }
        }
      }
      if (i == 3) {
        const uint32_t kl = t < K ? t : K - 1;
        if ((kl & 15) == 15) niw64_finish_batch<JB, LOO, ACCUM, true>(fd, kl, kk, c, rb, qkeep, gz, msk, live, qown, out, ld, vec_ok);
        else if (kl == K - 1) niw64_finish_batch<JB, LOO, ACCUM>(fd, kl, kk, c, rb, qkeep, gz, msk, live, qown, out, ld, vec_ok);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// dim <= 8: the padded 32 x 32 contraction above would spend 16 .. 256 times the arithmetic the problem has (a
// 3-d niw feature scored at the rate of a 32-d one: 2.7e10 evals/s whatever the dimension).  Here a lane keeps one
// group -- the lower triangle of W_k, W_k mu_k, c0, c1, all in double registers -- and the rows of a 64-row chunk
// stream past it as wave-uniform values (lane r loads row r, v_readlane hands it round): D (D + 3) / 2 double fma
// per evaluation on the vector pipe, the same float log1p epilogue, same own-group / mask protocol as
// k_score_niw64.  grid.y = tiles of 64 groups; a row of 64 scores is one 256-byte store.
// ---------------------------------------------------------------------------
MSC_DEV double bcast_f64(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
template <int D>
__global__ __launch_bounds__(256) void k_score_niw_small(const FeatDesc *__restrict__ feats, uint32_t f, uint32_t K,
                                                          uint32_t kpad, uint64_t row0, uint64_t nrows,
                                                          const int32_t *__restrict__ z, double *__restrict__ qown,
                                                          float *__restrict__ out, uint64_t ld, int accum) {
  const FeatDesc fd = feats[f];
  const int lane = threadIdx.x & 63;
  const uint32_t k = blockIdx.y * 64 + lane;
  const bool has_k = k < K;
  const size_t kc = has_k ? k : 0;
  double w[D * (D + 1) / 2], nb[D];
#pragma unroll
  for (int i = 0; i < D; i++) {
#pragma unroll
    for (int j = 0; j <= i; j++) w[i * (i + 1) / 2 + j] = fd.niw_w64[kc * niw_w_stream(D) + niw_w_index(i, j)];
    nb[i] = -fd.niw_mu64[kc * niw_b_stream(D) + niw_b_index(i)];       // (the f64 MFMA kernel's operand streams, k_niw_prepare)
  }
  const double c0 = fd.niw_c64[kc * 8], c1 = fd.niw_c64[kc * 8 + 1];
  const float *X = reinterpret_cast<const float *>(fd.col);
  const uint64_t nchunks = (nrows + 63) / 64;
  const uint64_t wave_id = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint64_t nwaves = (uint64_t)gridDim.x * 4;
  for (uint64_t chunk = wave_id; chunk < nchunks; chunk += nwaves) {
    const uint64_t rb = chunk * 64;
    const int nr = (int)((nrows - rb) < 64 ? (nrows - rb) : 64);
    double xd[D];
    int gz = -1;
    bool msk = false;
    if (lane < nr) {
      const uint64_t row = row0 + rb + lane;
#pragma unroll
      for (int j = 0; j < D; j++) xd[j] = (double)X[row * D + j];
      if (z != nullptr) gz = z[rb + lane];
      if (fd.mask != nullptr)
#pragma unroll
        for (int j = 0; j < D; j++) msk |= fd.mask[row * D + j] != 0;
    } else {
#pragma unroll
      for (int j = 0; j < D; j++) xd[j] = 0.0;
    }
    const unsigned long long mbits = __builtin_amdgcn_ballot_w64(msk);
    for (int r = 0; r < nr; r++) {
      float *p = out + (rb + r) * ld + k;
      if ((mbits >> r) & 1ull) {                          // a masked value adds nothing (wave-uniform branch)
        if (!accum && has_k) *p = 0.f;
        continue;
      }
      double x[D];
#pragma unroll
      for (int j = 0; j < D; j++) x[j] = bcast_f64(xd[j], r);
      double q = 0.0;
#pragma unroll
      for (int i = 0; i < D; i++) {
        double y = nb[i];                                 // W x - W mu
#pragma unroll
        for (int j = 0; j <= i; j++) y = fma(w[i * (i + 1) / 2 + j], x[j], y);
        q = fma(y, y, q);
      }
      // (dim <= 8: c0 and the term stay small, the compensated float log1p -- 1e-7 of the term -- is enough, and this
      //  kernel is bound by its vector arithmetic)
      double sc = c0 - c1 * (double)log1p_acc((float)q);
      const int g = __builtin_amdgcn_readlane(gz, r);
      if (g >= 0 && (uint32_t)g == k) {                   // the own group: its value comes from k_niw_loo_patch
        qown[rb + r] = q;
        sc = 0.0;
      }
      if (has_k) *p = accum ? *p + (float)sc : (float)sc;
    }
  }
}

// leave-one-out value of every row's own group (closed form in the file header) from the q the score kernel
// left in qown; the score kernel wrote 0 there, so this adds.  One thread per row.
__global__ __launch_bounds__(256) void k_niw_loo_patch(const FeatDesc *__restrict__ feats, uint32_t f, uint32_t K,
                                                        uint64_t row0, uint64_t nrows, const int32_t *__restrict__ z,
                                                        const double *__restrict__ qown, float *__restrict__ out,
                                                        uint64_t ld) {
  const FeatDesc fd = feats[f];
  const uint64_t n = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= nrows) return;
  const int g = z[n];
  if (g < 0 || (uint32_t)g >= K) return;
  if (fd.mask != nullptr)
    for (uint32_t e = 0; e < fd.dim; e++)
      if (fd.mask[(row0 + n) * fd.dim + e] != 0) return;
  const double *c64 = fd.niw_c64 + (size_t)g * 8;
  out[n * ld + g] += (float)(c64[2] + c64[3] * log1p(-fmin(c64[4] * qown[n], 1.0 - 1e-15)));
}

// ---------------------------------------------------------------------------
// accumulate: sum_x and sum_xxT by group.  Rows are first bucketed by group (histogram, scan,
// scatter of row indices: three small integer kernels), then every group's rows are summed by a few
// workgroups in registers -- lane (i, half) keeps sum_xxT[i][16 half .. 16 half + 15] in sixteen
// doubles and lane (i, 0) sum_x[i]; products of two floats are exact in double -- and each
// workgroup adds its 1056 partial sums once.  (First version: one double atomic per row and
// element, 2.7e8 of them on C4: 1.9 ms; this one: see DESIGN.md section 5.)
// scratch (uint32): [0, K] offsets | [K+1, 2K+1) cursors | [2K+1, 2K+1+nrows) row indices
// ---------------------------------------------------------------------------
MSC_DEV bool niw_row_counts(const FeatDesc &fd, uint32_t K, uint64_t row, int g) {
  if (g < 0 || (uint32_t)g >= K) return false;
  if (fd.mask != nullptr)
    for (uint32_t e = 0; e < fd.dim; e++)
      if (fd.mask[row * fd.dim + e] != 0) return false;       // a masked vector is not part of the group
  return true;
}
// LDS: per-workgroup histogram first (the groups are few and every row would otherwise hit one of K global
// words); K beyond the LDS budget goes straight to global atomics.
constexpr uint32_t kBucketLdsGroups = 8192;
__global__ __launch_bounds__(256) void k_niw_bucket_count(const FeatDesc *__restrict__ feats, uint32_t f, uint32_t K,
                                                           uint64_t row0, uint64_t nrows,
                                                           const int32_t *__restrict__ z, uint32_t *__restrict__ scratch) {
  extern __shared__ uint32_t hist[];
  const FeatDesc fd = feats[f];
  const bool lds = K <= kBucketLdsGroups;
  if (lds) {
    for (uint32_t k = threadIdx.x; k < K; k += 256) hist[k] = 0u;
    __syncthreads();
  }
  const uint64_t per = (nrows + gridDim.x - 1) / gridDim.x, lo = (uint64_t)blockIdx.x * per;
  const uint64_t hi = lo + per < nrows ? lo + per : nrows;
  for (uint64_t n = lo + threadIdx.x; n < hi; n += 256)
    if (niw_row_counts(fd, K, row0 + n, z[n])) atomicAdd(lds ? &hist[z[n]] : &scratch[K + 1 + z[n]], 1u);   // counts, in the cursor slots
  if (lds) {
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < K; k += 256)
      if (hist[k]) atomicAdd(&scratch[K + 1 + k], hist[k]);
  }
}
// one block: offsets[k] = sum of counts before k, offsets[K] = total; cursors := 0
__global__ __launch_bounds__(1024) void k_niw_bucket_scan(uint32_t K, uint32_t *__restrict__ scratch) {
  __shared__ uint32_t part[1024];
  const uint32_t per = (K + 1023) / 1024, lo = threadIdx.x * per, hi = lo + per < K ? lo + per : K;
  uint32_t s = 0;
  for (uint32_t k = lo; k < hi; k++) s += scratch[K + 1 + k];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const uint32_t v = threadIdx.x >= (uint32_t)off ? part[threadIdx.x - off] : 0u;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t run = threadIdx.x ? part[threadIdx.x - 1] : 0u;
  for (uint32_t k = lo; k < hi; k++) {
    const uint32_t c = scratch[K + 1 + k];
    scratch[k] = run;
    scratch[K + 1 + k] = 0u;
    run += c;
  }
  if (threadIdx.x == 1023) scratch[K] = part[1023];
}
__global__ __launch_bounds__(256) void k_niw_bucket_scatter(const FeatDesc *__restrict__ feats, uint32_t f, uint32_t K,
                                                             uint64_t row0, uint64_t nrows,
                                                             const int32_t *__restrict__ z, uint32_t *__restrict__ scratch) {
  extern __shared__ uint32_t sh[];                   // [K] counts of this workgroup's slab, [K] where its share starts
  const FeatDesc fd = feats[f];
  uint32_t *idx = scratch + 2 * (size_t)K + 1;
  const uint64_t per = (nrows + gridDim.x - 1) / gridDim.x, lo = (uint64_t)blockIdx.x * per;
  const uint64_t hi = lo + per < nrows ? lo + per : nrows;
  if (K > kBucketLdsGroups) {
    for (uint64_t n = lo + threadIdx.x; n < hi; n += 256) {
      const int g = z[n];
      if (niw_row_counts(fd, K, row0 + n, g)) idx[scratch[g] + atomicAdd(&scratch[K + 1 + g], 1u)] = (uint32_t)n;
    }
    return;
  }
  uint32_t *cnt = sh, *base = sh + K;
  for (uint32_t k = threadIdx.x; k < K; k += 256) cnt[k] = 0u;
  __syncthreads();
  for (uint64_t n = lo + threadIdx.x; n < hi; n += 256)
    if (niw_row_counts(fd, K, row0 + n, z[n])) atomicAdd(&cnt[z[n]], 1u);
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < K; k += 256) {   // reserve this workgroup's share of every group's range
    base[k] = cnt[k] ? scratch[k] + atomicAdd(&scratch[K + 1 + k], cnt[k]) : 0u;
    cnt[k] = 0u;
  }
  __syncthreads();
  for (uint64_t n = lo + threadIdx.x; n < hi; n += 256) {
    const int g = z[n];
    if (niw_row_counts(fd, K, row0 + n, g)) idx[base[g] + atomicAdd(&cnt[g], 1u)] = (uint32_t)n;
  }
}
// grid (K, splits, tiles): the 4 waves of a block take the group's rows round-robin; blockIdx.z = a 32 x 32 tile
// (ti, tj) of sum_xxT (one tile up to dim 32, 16 at dim 128).  Lane (i, half) keeps rows 32 ti + i, columns
// 32 tj + 16 half .. + 15 in sixteen doubles; the tiles of the first column also keep sum_x
__global__ __launch_bounds__(256) void k_niw_group_sums(const FeatDesc *__restrict__ feats, uint32_t f, uint32_t K,
                                                         uint64_t row0, const uint32_t *__restrict__ scratch, int sign) {
  __shared__ float xs[4][2][32];                       // per wave: the row's values of the tile's row range | column range
  __shared__ double red[4][17][64];
  const FeatDesc fd = feats[f];
  const uint32_t d = fd.dim, k = blockIdx.x;
  const uint32_t nt = (d + 31u) / 32u, ti = blockIdx.z / nt, tj = blockIdx.z % nt;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t i = (uint32_t)lane >> 1, half = (uint32_t)lane & 1u;
  const uint32_t beg = scratch[k], end = scratch[k + 1];
  const uint32_t *idx = scratch + 2 * (size_t)K + 1;
  const float *X = reinterpret_cast<const float *>(fd.col);
  double sxx[16], sx = 0.0;
#pragma unroll
  for (int j = 0; j < 16; j++) sxx[j] = 0.0;
  for (uint32_t p = beg + blockIdx.y * 4 + wave; p < end; p += gridDim.y * 4) {
    const float *x = X + (row0 + idx[p]) * d;
    const uint32_t e = (uint32_t)(lane & 31), src = 32u * (lane < 32 ? ti : tj) + e;
    xs[wave][lane >> 5][e] = src < d ? x[src] : 0.f;
    const double xi = (double)xs[wave][0][i];
    if (half == 0) sx += xi;
#pragma unroll
    for (int j = 0; j < 16; j++) sxx[j] = fma(xi, (double)xs[wave][1][half * 16 + j], sxx[j]);
  }
  // the four waves' partial sums, then one atomic per element and block
#pragma unroll
  for (int j = 0; j < 16; j++) red[wave][j][lane] = sxx[j];
  red[wave][16][lane] = sx;
  __syncthreads();
  if (end == beg) return;
  const size_t stride = d + (size_t)d * d;
  double *dst = fd.acc_f64 + (size_t)k * stride;
  for (uint32_t e = threadIdx.x; e < 17 * 64; e += 256) {
    const uint32_t j = e >> 6, l = e & 63, ii = 32u * ti + (l >> 1), hh = l & 1u;
    const double v = red[0][j][l] + red[1][j][l] + red[2][j][l] + red[3][j][l];
    if (ii >= d) continue;
    if (j == 16) {
      if (tj == 0 && hh == 0 && v != 0.0) atomicAdd(&dst[ii], (double)sign * v);
    } else {
      const uint32_t col = 32u * tj + hh * 16 + j;
      if (col < d && v != 0.0) atomicAdd(&dst[d + (size_t)ii * d + col], (double)sign * v);
    }
  }
  if (blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)
    atomicAdd(reinterpret_cast<unsigned long long *>(&fd.acc_i64[k]), (unsigned long long)((long long)sign * (long long)(end - beg)));
}

// additive <-> raw for one niw feature (thread per element of the group-major block)
__global__ __launch_bounds__(256) void k_niw_commit(const FeatDesc *__restrict__ feats, uint32_t f,
                                                     uint32_t K, uint32_t kpad, int to_raw) {
  const FeatDesc fd = feats[f];
  const size_t stride = fd.dim + (size_t)fd.dim * fd.dim, total = (size_t)K * stride;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < total) {
    if (to_raw) fd.raw_f32[i] = (float)fd.acc_f64[i];
    else fd.acc_f64[i] = (double)fd.raw_f32[i];
  }
  if (i < kpad) {
    if (to_raw) fd.raw_u32[i] = (uint32_t)fd.acc_i64[i];
    else fd.acc_i64[i] = fd.raw_u32[i];
  }
}

// ---------------------------------------------------------------------------
int launch_niw_prepare(hipStream_t stream, const FeatDesc *feats_dev, uint32_t f, uint32_t dim, uint32_t K,
                       uint32_t kpad) {
  const size_t lds = sizeof(double) * ((size_t)dim * (dim + 1) + 2 * (size_t)dim);     // two packed triangles, mu_n, W mu_n
  static unsigned long long attr_devices = 0;
  if (lds > 64u * 1024u && first_use_on_device(attr_devices))
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_niw_prepare), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(k_niw_prepare, dim3(K + 1), dim3(64), lds, stream, feats_dev, f, K, kpad);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_niw_score_data(hipStream_t stream, const FeatDesc *feats_dev, uint32_t f, uint32_t K,
                          uint32_t kpad, float *out) {
  hipLaunchKernelGGL(k_niw_score_data, dim3((K + 255) / 256), dim3(256), 0, stream, feats_dev, f, K, kpad, out);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <int NB, bool LOO, bool ACCUM>
static void launch_niw64_nb(hipStream_t stream, const dim3 grid, const FeatDesc *feats_dev, uint32_t f, uint32_t K, uint32_t kpad,
                            uint64_t row0, uint64_t nrows, const int32_t *z, double *qown, float *out, uint64_t ld) {
  // Row blocks of 16 per wave: four up to dim 32 (C4: 179 + 64 registers, two waves per SIMD, 0.88 of the f64 matrix
  // peak), two beyond -- with four the wave's features, sums and epilogue take 290-512 registers and the kernel runs one
  // wave per SIMD: dim 64 4.94 ms; with two (and the operand stream pipelined, PIPE in the kernel) 3.06 ms, three
  // waves per SIMD; dim 128 17.2 -> 11.4 ms (profiles/r03_niw_dims.txt).
  // (k_score_niw64_lag reads the operand streams as buffers, 32-bit offsets: 2 or 6 KiB a group)
  if constexpr (NB <= 2) {
    if ((uint64_t)K * 6144u < 0x7fffffffull) {
      hipLaunchKernelGGL((k_score_niw64_lag<NB, LOO, ACCUM>), (note_kernel(0, "k_score_niw64_lag<%d, %s, %s>", NB, tf(LOO), tf(ACCUM)), grid), dim3(256), 0, stream, feats_dev, f, K, kpad, row0, nrows, z, qown, out, ld);
      return;
    }
  }
  if constexpr (NB > 2) {
    if ((uint64_t)K * (NB * (NB + 1) / 2) * 2048u < 0x7fffffffull)        // (buffer loads: 32-bit offsets into the operand stream)
      hipLaunchKernelGGL((k_score_niw64_wide<NB, LOO, ACCUM>), (note_kernel(0, "k_score_niw64_wide<%d, %s, %s>", NB, tf(LOO), tf(ACCUM)), dim3(grid.x * 2)), dim3(256), 0, stream, feats_dev, f, K, kpad, row0, nrows, z, qown, out, ld);
    else
      hipLaunchKernelGGL((k_score_niw64<NB, 2, LOO, ACCUM>), (note_kernel(0, "k_score_niw64<%d, 2, %s, %s>", NB, tf(LOO), tf(ACCUM)), dim3(grid.x * 2)), dim3(256), 0, stream, feats_dev, f, K, kpad, row0, nrows, z, qown, out, ld);
  } else
    hipLaunchKernelGGL((k_score_niw64<NB, 4, LOO, ACCUM>), (note_kernel(0, "k_score_niw64<%d, 4, %s, %s>", NB, tf(LOO), tf(ACCUM)), grid), dim3(256), 0, stream, feats_dev, f, K, kpad, row0, nrows, z, qown, out, ld);
}
template <bool LOO, bool ACCUM>
static void launch_niw_score_t(hipStream_t stream, int num_cus, bool f32_fast, uint32_t dim, const FeatDesc *feats_dev,
                               uint32_t f, uint32_t K, uint32_t kpad, uint64_t row0, uint64_t nrows,
                               const int32_t *z, double *qown, float *out, uint64_t ld) {
  constexpr int kRowsPerWave = 64;             // 2 tiles of 32 (f32) or 4 blocks of 16 (f64)
  const uint64_t nblocks = (nrows + kRowsPerWave - 1) / kRowsPerWave;
  uint64_t gx = (nblocks + 3) / 4;
  const uint64_t cap = (uint64_t)num_cus * 8;
  if (gx > cap) gx = cap;
  const dim3 grid((unsigned)(gx ? gx : 1)), block(256);
  if (f32_fast && dim <= (uint32_t)kNiwPad)
    hipLaunchKernelGGL((k_score_niw<2, LOO, ACCUM>), (note_kernel(0, "k_score_niw<2, %s, %s>", tf(LOO), tf(ACCUM)), grid), block, 0, stream, feats_dev, f, K, kpad, row0, nrows, z, out, ld);
  else {
    switch (niw_blocks(dim)) {
      case 1: launch_niw64_nb<1, LOO, ACCUM>(stream, grid, feats_dev, f, K, kpad, row0, nrows, z, qown, out, ld); break;
      case 2: launch_niw64_nb<2, LOO, ACCUM>(stream, grid, feats_dev, f, K, kpad, row0, nrows, z, qown, out, ld); break;
      case 3: launch_niw64_nb<3, LOO, ACCUM>(stream, grid, feats_dev, f, K, kpad, row0, nrows, z, qown, out, ld); break;
      case 4: launch_niw64_nb<4, LOO, ACCUM>(stream, grid, feats_dev, f, K, kpad, row0, nrows, z, qown, out, ld); break;
      case 5: launch_niw64_nb<5, LOO, ACCUM>(stream, grid, feats_dev, f, K, kpad, row0, nrows, z, qown, out, ld); break;
      case 6: launch_niw64_nb<6, LOO, ACCUM>(stream, grid, feats_dev, f, K, kpad, row0, nrows, z, qown, out, ld); break;
      case 7: launch_niw64_nb<7, LOO, ACCUM>(stream, grid, feats_dev, f, K, kpad, row0, nrows, z, qown, out, ld); break;
      default: launch_niw64_nb<8, LOO, ACCUM>(stream, grid, feats_dev, f, K, kpad, row0, nrows, z, qown, out, ld); break;
    }
    if (LOO)
      hipLaunchKernelGGL(k_niw_loo_patch, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, stream, feats_dev, f, K, row0,
                         nrows, z, qown, out, ld);
  }
}

template <int D>
static void launch_niw_small(hipStream_t stream, int num_cus, const FeatDesc *feats_dev, uint32_t f, uint32_t K, uint32_t kpad,
                             uint64_t row0, uint64_t nrows, const int32_t *z, double *qown, float *out, uint64_t ld, bool accum) {
  const uint32_t ktiles = (K + 63) / 64;
  uint64_t gx = ((nrows + 63) / 64 + 3) / 4;
  const uint64_t cap = std::max<uint64_t>(1, (uint64_t)num_cus * 8 / ktiles);
  if (gx > cap) gx = cap;
  hipLaunchKernelGGL(k_score_niw_small<D>, dim3((unsigned)(gx ? gx : 1), ktiles), dim3(256), 0, stream, feats_dev, f, K, kpad,
                     row0, nrows, z, qown, out, ld, accum ? 1 : 0);
  if (z != nullptr)
    hipLaunchKernelGGL(k_niw_loo_patch, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, stream, feats_dev, f, K, row0,
                       nrows, z, qown, out, ld);
}

int launch_niw_score(hipStream_t stream, int num_cus, const FeatDesc *feats_dev, uint32_t f, uint32_t dim, uint32_t K,
                     uint32_t kpad, uint64_t row0, uint64_t nrows, const int32_t *z, bool accum, bool f32_fast,
                     double *qown, float *out, uint64_t ld) {
  if (!f32_fast && dim >= 1 && dim <= 8 && nrows > 0) {
    switch (dim) {
      case 1: launch_niw_small<1>(stream, num_cus, feats_dev, f, K, kpad, row0, nrows, z, qown, out, ld, accum); break;
      case 2: launch_niw_small<2>(stream, num_cus, feats_dev, f, K, kpad, row0, nrows, z, qown, out, ld, accum); break;
      case 3: launch_niw_small<3>(stream, num_cus, feats_dev, f, K, kpad, row0, nrows, z, qown, out, ld, accum); break;
      case 4: launch_niw_small<4>(stream, num_cus, feats_dev, f, K, kpad, row0, nrows, z, qown, out, ld, accum); break;
      case 5: launch_niw_small<5>(stream, num_cus, feats_dev, f, K, kpad, row0, nrows, z, qown, out, ld, accum); break;
      case 6: launch_niw_small<6>(stream, num_cus, feats_dev, f, K, kpad, row0, nrows, z, qown, out, ld, accum); break;
      case 7: launch_niw_small<7>(stream, num_cus, feats_dev, f, K, kpad, row0, nrows, z, qown, out, ld, accum); break;
      default: launch_niw_small<8>(stream, num_cus, feats_dev, f, K, kpad, row0, nrows, z, qown, out, ld, accum); break;
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
  }
  if (z && accum) launch_niw_score_t<true, true>(stream, num_cus, f32_fast, dim, feats_dev, f, K, kpad, row0, nrows, z, qown, out, ld);
  else if (z) launch_niw_score_t<true, false>(stream, num_cus, f32_fast, dim, feats_dev, f, K, kpad, row0, nrows, z, qown, out, ld);
  else if (accum) launch_niw_score_t<false, true>(stream, num_cus, f32_fast, dim, feats_dev, f, K, kpad, row0, nrows, z, qown, out, ld);
  else launch_niw_score_t<false, false>(stream, num_cus, f32_fast, dim, feats_dev, f, K, kpad, row0, nrows, z, qown, out, ld);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---------------------------------------------------------------------------
// accumulate for small dimensions: d + d^2 sums per group are few enough for the per-workgroup LDS histogram the
// scalar families use (ds_add_f64; float x float is exact in double), flushed with one global atomic per non-zero
// bin.  The bucketing pass above earns its three extra kernels at dim 32 (1056 sums per row); at dim 3 it was
// 110 us of a 420 us sweep step (262k rows x 128 groups), this is ~10.
// LDS: double sums[K][d + d*d] | uint32 counts[K]
// ---------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(1024) void k_niw_accumulate_small(const FeatDesc *__restrict__ feats, uint32_t f, uint32_t K,
                                                                uint64_t row0, uint64_t nrows, const int32_t *__restrict__ z,
                                                                int sign) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr uint32_t kStride = D + D * D;
  const FeatDesc fd = feats[f];
  double *sums = reinterpret_cast<double *>(smem);
  uint32_t *cnt = reinterpret_cast<uint32_t *>(sums + (size_t)K * kStride);
  for (uint32_t i = threadIdx.x; i < K * kStride; i += blockDim.x) sums[i] = 0.0;
  for (uint32_t i = threadIdx.x; i < K; i += blockDim.x) cnt[i] = 0u;
  __syncthreads();
  const uint64_t per = (nrows + gridDim.x - 1) / gridDim.x;
  const uint64_t lo = (uint64_t)blockIdx.x * per, hi = lo + per < nrows ? lo + per : nrows;
  const float *X = reinterpret_cast<const float *>(fd.col);
  for (uint64_t n = lo + threadIdx.x; n < hi; n += blockDim.x) {
    const int g = z[n];
    const uint64_t row = row0 + n;
    if (!niw_row_counts(fd, K, row, g)) continue;
    double x[D];
#pragma unroll
    for (int j = 0; j < D; j++) x[j] = (double)X[row * D + j];
    double *dst = sums + (size_t)g * kStride;
    atomicAdd(&cnt[g], 1u);
#pragma unroll
    for (int i = 0; i < D; i++) {
      atomicAdd(&dst[i], x[i]);
#pragma unroll
      for (int j = 0; j < D; j++) atomicAdd(&dst[D + i * D + j], x[i] * x[j]);
    }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < K * kStride; i += blockDim.x)
    if (sums[i] != 0.0) atomicAdd(&fd.acc_f64[i], (double)sign * sums[i]);
  for (uint32_t i = threadIdx.x; i < K; i += blockDim.x)
    if (cnt[i]) atomicAdd(reinterpret_cast<unsigned long long *>(&fd.acc_i64[i]), (unsigned long long)((long long)sign * (long long)cnt[i]));
}
template <int D>
static void launch_niw_accumulate_small(hipStream_t stream, int num_cus, const FeatDesc *feats_dev, uint32_t f, uint32_t K,
                                        uint64_t row0, uint64_t nrows, const int32_t *z, int sign, size_t lds) {
  static unsigned long long attr_devices = 0;
  if (first_use_on_device(attr_devices))
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_niw_accumulate_small<D>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  uint64_t blocks = (nrows + 4095) / 4096;
  if (blocks < (uint64_t)num_cus) blocks = (nrows + 1023) / 1024;
  if (blocks > (uint64_t)num_cus * 2) blocks = (uint64_t)num_cus * 2;
  hipLaunchKernelGGL(k_niw_accumulate_small<D>, dim3((unsigned)(blocks ? blocks : 1)), dim3(1024), lds, stream, feats_dev, f, K,
                     row0, nrows, z, sign);
}

// scratch_dev: 2 K + 1 + nrows uint32 (see k_niw_bucket_*)
int launch_niw_accumulate(hipStream_t stream, int num_cus, const FeatDesc *feats_dev, uint32_t f, uint32_t K,
                          uint64_t row0, uint64_t nrows, const int32_t *z, int sign, uint32_t *scratch_dev, uint32_t dim) {
  if (nrows == 0) return 0;
  const size_t small_lds = (size_t)K * ((dim + (size_t)dim * dim) * sizeof(double) + sizeof(uint32_t));
  if (dim >= 1 && dim <= 8 && small_lds <= 96u * 1024u) {
    switch (dim) {
      case 1: launch_niw_accumulate_small<1>(stream, num_cus, feats_dev, f, K, row0, nrows, z, sign, small_lds); break;
      case 2: launch_niw_accumulate_small<2>(stream, num_cus, feats_dev, f, K, row0, nrows, z, sign, small_lds); break;
      case 3: launch_niw_accumulate_small<3>(stream, num_cus, feats_dev, f, K, row0, nrows, z, sign, small_lds); break;
      case 4: launch_niw_accumulate_small<4>(stream, num_cus, feats_dev, f, K, row0, nrows, z, sign, small_lds); break;
      case 5: launch_niw_accumulate_small<5>(stream, num_cus, feats_dev, f, K, row0, nrows, z, sign, small_lds); break;
      case 6: launch_niw_accumulate_small<6>(stream, num_cus, feats_dev, f, K, row0, nrows, z, sign, small_lds); break;
      case 7: launch_niw_accumulate_small<7>(stream, num_cus, feats_dev, f, K, row0, nrows, z, sign, small_lds); break;
      default: launch_niw_accumulate_small<8>(stream, num_cus, feats_dev, f, K, row0, nrows, z, sign, small_lds); break;
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
  }
  if (hipMemsetAsync(scratch_dev, 0, sizeof(uint32_t) * (2 * (size_t)K + 1), stream) != hipSuccess) return -1;
  uint64_t gx = (nrows + 255) / 256;
  const uint64_t cap = (uint64_t)num_cus * 8;
  if (gx > cap) gx = cap;
  const size_t lds1 = K <= kBucketLdsGroups ? sizeof(uint32_t) * K : 0, lds2 = 2 * lds1;
  hipLaunchKernelGGL(k_niw_bucket_count, dim3((unsigned)gx), dim3(256), lds1, stream, feats_dev, f, K, row0, nrows, z, scratch_dev);
  hipLaunchKernelGGL(k_niw_bucket_scan, dim3(1), dim3(1024), 0, stream, K, scratch_dev);
  hipLaunchKernelGGL(k_niw_bucket_scatter, dim3((unsigned)gx), dim3(256), lds2, stream, feats_dev, f, K, row0, nrows, z, scratch_dev);
  uint32_t splits = (uint32_t)(((uint64_t)num_cus * 4 + K - 1) / K);      // about four workgroups per CU over all groups
  const uint64_t by_rows = (nrows / K + 63) / 64;                           // ... but at least ~64 rows per workgroup
  if (splits > by_rows) splits = (uint32_t)(by_rows ? by_rows : 1);
  if (splits == 0) splits = 1;
  if (splits > 65535u) splits = 65535u;
  const uint32_t nt = (dim + 31u) / 32u;
  hipLaunchKernelGGL(k_niw_group_sums, dim3(K, splits, nt * nt), dim3(256), 0, stream, feats_dev, f, K, row0, scratch_dev, sign);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_niw_commit(hipStream_t stream, const FeatDesc *feats_dev, uint32_t f, uint32_t dim, uint32_t K,
                      uint32_t kpad, int to_raw) {
  size_t total = (size_t)K * (dim + (size_t)dim * dim);
  if (total < kpad) total = kpad;
  hipLaunchKernelGGL(k_niw_commit, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, feats_dev, f, K, kpad, to_raw);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace msc
