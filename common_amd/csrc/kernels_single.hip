// kernels_single.hip -- the per-value entry of the plugin API (group::add_value /
// remove_value / score_value / score_data, base.hpp:25-28) as a batch of one.
//
// The host writes {op, hp, suff-stats, value} into a pinned, device-mapped mailbox,
// launches one wave and waits; the wave computes in double (these calls are
// launch-latency-bound, ~10 us each, so arithmetic cost is irrelevant) and writes the
// updated suff-stats / the score back into the mailbox.  The batched kernels are the
// throughput path; this exists so that code written against the virtual API runs on
// the same device arithmetic, with no host-side evaluation anywhere.
#include "family_math.hpp"
#include "launchers.hpp"

namespace msc {

// mailbox layout (byte offsets)
//   0   int32 family, dim, op, status
//   16  float score
//   64  float hp[...]
//   then (16-aligned) the suff-stat record exactly as msc_state_set_ss takes it
//   then (16-aligned) the value

// in-place lower Cholesky of a[d][d] (LDS), one wave; returns ln det
__device__ double chol_lds(double *a, uint32_t d, int t) {
  double logdet = 0;
  for (uint32_t j = 0; j < d; j++) {
    const double ljj = sqrt(a[(size_t)j * d + j]);
    logdet += 2.0 * log(ljj);
    __syncthreads();
    for (uint32_t i = j + 1 + t; i < d; i += 64) a[(size_t)i * d + j] /= ljj;
    if (t == 0) a[(size_t)j * d + j] = ljj;
    __syncthreads();
    const uint32_t m = d - j - 1;
    for (uint32_t idx = t; idx < m * m; idx += 64) {
      const uint32_t i = j + 1 + idx / m, c = j + 1 + idx % m;
      if (c <= i) a[(size_t)i * d + c] -= a[(size_t)i * d + j] * a[(size_t)c * d + j];
    }
    __syncthreads();
  }
  return logdet;
}

__device__ double lmultigamma_d(uint32_t d, double a) {
  double t = 0.25 * (double)(d * (d - 1)) * kLogPi;
  for (uint32_t j = 1; j <= d; j++) t += lgamma(a + 0.5 * (1.0 - (double)j));
  return t;
}

__global__ __launch_bounds__(64) void k_value_op(unsigned char *__restrict__ mb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  MailboxHeader *h = reinterpret_cast<MailboxHeader *>(mb);
  const int t = threadIdx.x;
  const int family = h->family, op = h->op;
  const uint32_t d = (uint32_t)h->dim;
  const float *hp = reinterpret_cast<const float *>(mb + h->hp_off);
  unsigned char *ss = mb + h->ss_off;
  const unsigned char *val = mb + h->value_off;
  uint32_t *su = reinterpret_cast<uint32_t *>(ss);
  float *sf = reinterpret_cast<float *>(ss);
  double score = 0.0;

  if (family == MSC_NIW) {
    // record {u32 count, f32 sum_x[d], f32 sum_xxT[d*d]}, hp {kappa, nu, mu[d], psi[d*d]}
    float *sx = sf + 1, *sxx = sf + 1 + d;
    const float *x = reinterpret_cast<const float *>(val);
    if (op == MSC_OP_ADD || op == MSC_OP_REMOVE) {
      const float sg = op == MSC_OP_ADD ? 1.f : -1.f;
      for (uint32_t i = t; i < d; i += 64) sx[i] += sg * x[i];
      for (uint32_t idx = t; idx < d * d; idx += 64) sxx[idx] += sg * x[idx / d] * x[idx % d];
      if (t == 0) su[0] += op == MSC_OP_ADD ? 1u : 0xffffffffu;
    } else {
      // LDS: one d x d matrix (Psi_n, then -- for score_data -- the prior's Psi in the same place), mu_n[d], y[d]:
      // 130 KB at dim 128
      double *A = reinterpret_cast<double *>(smem), *mun = A + (size_t)d * d, *y = mun + d;
      const double kappa = hp[0], nu = hp[1], n = su[0];
      const float *mu = hp + 2, *psi = hp + 2 + d;
      const double kn = kappa + n, nun = nu + n;
      for (uint32_t i = t; i < d; i += 64) mun[i] = (kappa * (double)mu[i] + (double)sx[i]) / kn;
      __syncthreads();
      for (uint32_t idx = t; idx < d * d; idx += 64) {
        const uint32_t i = idx / d, j = idx % d;
        A[idx] = (double)psi[idx] + (double)sxx[idx] + kappa * (double)mu[i] * (double)mu[j] - kn * mun[i] * mun[j];
      }
      __syncthreads();
      const double ldn = chol_lds(A, d, t);
      if (op == MSC_OP_SCORE_VALUE) {
        const double dof = nun - (double)d + 1.0, s = (kn + 1.0) / (kn * dof);
        if (t == 0) {
          double q = 0;
          for (uint32_t i = 0; i < d; i++) {         // forward solve L y = x - mu_n
            double r = (double)x[i] - mun[i];
            for (uint32_t m = 0; m < i; m++) r -= A[(size_t)i * d + m] * y[m];
            y[i] = r / A[(size_t)i * d + i];
            q += y[i] * y[i];
          }
          q /= s;                                     // Sigma = s * Psi_n
          const double dd = d;
          score = lgamma(0.5 * (dof + dd)) - lgamma(0.5 * dof) - 0.5 * dd * log(dof * kPi) -
                  0.5 * (ldn + dd * log(s)) - 0.5 * (dof + dd) * log1p(q / dof);
        }
      } else {
        __syncthreads();
        for (uint32_t idx = t; idx < d * d; idx += 64) A[idx] = psi[idx];
        __syncthreads();
        const double ld0 = chol_lds(A, d, t);
        if (t == 0)
          score = n == 0.0 ? 0.0
                           : lmultigamma_d(d, 0.5 * nun) - lmultigamma_d(d, 0.5 * nu) + 0.5 * nu * ld0 -
                                 0.5 * nun * ldn + 0.5 * (double)d * log(kappa / kn) - 0.5 * n * (double)d * kLogPi;
      }
    }
  } else if (t == 0) {
    switch (family) {
      case MSC_BB: {
        const bool v = val[0] != 0;
        if (op == MSC_OP_ADD) su[v ? 0 : 1]++;
        else if (op == MSC_OP_REMOVE) su[v ? 0 : 1]--;
        else if (op == MSC_OP_SCORE_VALUE) {
          const double a = hp[0], b = hp[1], hh = su[0], tt = su[1];
          score = log((v ? a + hh : b + tt) / (a + b + hh + tt));
        } else score = bb_score_data(hp, su[0], su[1]);
      } break;
      case MSC_BBNC: {       // record {u32 heads, u32 tails, f32 p}
        const bool v = val[0] != 0;
        if (op == MSC_OP_ADD) su[v ? 0 : 1]++;
        else if (op == MSC_OP_REMOVE) su[v ? 0 : 1]--;
        else if (op == MSC_OP_SCORE_VALUE) score = log(v ? (double)sf[2] : 1.0 - (double)sf[2]);
        else score = bbnc_score_data(hp, su[0], su[1], sf[2]);
      } break;
      case MSC_GP: {
        const uint32_t v = *reinterpret_cast<const uint32_t *>(val);
        if (op == MSC_OP_ADD) { su[0]++; su[1] += v; sf[2] += (float)lgamma((double)v + 1.0); }
        else if (op == MSC_OP_REMOVE) { su[0]--; su[1] -= v; sf[2] -= (float)lgamma((double)v + 1.0); }
        else if (op == MSC_OP_SCORE_VALUE)
          score = gp_score_exact((double)hp[0] + (double)su[1], (double)hp[1] + (double)su[0], (double)v);
        else score = gp_score_data(hp, su[0], su[1], (double)sf[2]);
      } break;
      case MSC_BNB: {       // record {u32 count, u32 sum}
        const uint32_t v = *reinterpret_cast<const uint32_t *>(val);
        if (op == MSC_OP_ADD) { su[0]++; su[1] += v; }
        else if (op == MSC_OP_REMOVE) { su[0]--; su[1] -= v; }
        else if (op == MSC_OP_SCORE_VALUE) score = bnb_score(hp, (double)su[0], (double)su[1], (double)v);
        else score = bnb_score_data(hp, su[0], su[1]);
      } break;
      case MSC_DM: {        // record {u32 counts[d], f32 ratio} (dm.hpp:86-88)
        const int32_t *x = reinterpret_cast<const int32_t *>(val);
        if (op == MSC_OP_ADD || op == MSC_OP_REMOVE) {
          for (uint32_t i = 0; i < d; i++) su[i] += op == MSC_OP_ADD ? (uint32_t)x[i] : 0u - (uint32_t)x[i];
          const double rr = dm_row_ratio(d, x);
          sf[d] = (float)((double)sf[d] + (op == MSC_OP_ADD ? rr : -rr));
        } else if (op == MSC_OP_SCORE_VALUE) score = dm_score_direct(hp, d, su, 1, x, false);
        else score = dm_score_data(hp, d, su, 1, (double)sf[d]);
      } break;
      case MSC_DD: {
        const int v = *reinterpret_cast<const int32_t *>(val);
        if (op == MSC_OP_ADD) { su[0]++; su[1 + v]++; }
        else if (op == MSC_OP_REMOVE) { su[0]--; su[1 + v]--; }
        else {
          double asum = 0;
          for (uint32_t i = 0; i < d; i++) asum += (double)hp[i];
          if (op == MSC_OP_SCORE_VALUE) score = log(((double)hp[v] + (double)su[1 + v]) / (asum + (double)su[0]));
          else {
            for (uint32_t i = 0; i < d; i++) score += lgamma((double)hp[i] + (double)su[1 + i]) - lgamma((double)hp[i]);
            score += lgamma(asum) - lgamma(asum + (double)su[0]);
          }
        }
      } break;
      case MSC_NICH: {
        // Welford update exactly as the reference's float fields evolve, taken in
        // double and rounded once per field
        const double x = *reinterpret_cast<const float *>(val);
        if (op == MSC_OP_ADD) {
          const double n1 = (double)su[0] + 1.0, mean = sf[1], delta = x - mean, m2 = mean + delta / n1;
          su[0]++;
          sf[1] = (float)m2;
          sf[2] = (float)((double)sf[2] + delta * (x - m2));
        } else if (op == MSC_OP_REMOVE) {
          const double mean = sf[1], total = mean * (double)su[0], delta = x - mean;
          su[0]--;
          const double m2 = su[0] == 0 ? 0.0 : (total - x) / (double)su[0];
          sf[1] = (float)m2;
          sf[2] = su[0] <= 1 ? 0.f : (float)((double)sf[2] - delta * (x - m2));
        } else if (op == MSC_OP_SCORE_VALUE) {
          const NichPost p = nich_posterior(hp, (double)su[0], (double)sf[1], (double)sf[2]);
          double c0, c1, c2;
          nich_coeffs(p, c0, c1, c2);
          const double dd = x - p.mu;
          score = c0 - c1 * log1p(c2 * dd * dd);
        } else score = nich_score_data(hp, su[0], sf[1], sf[2]);
      } break;
      default: break;   // noop model
    }
  }
  // the host spins on `status` in the pinned mailbox instead of waiting for the stream: everything any thread wrote
  // must be visible to the host before the flag is
  __threadfence_system();
  __syncthreads();
  if (t == 0) {
    h->score = (float)score;
    __threadfence_system();
    __hip_atomic_store(&h->status, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

int launch_value_op(hipStream_t stream, void *mailbox_dev, uint32_t dim, int family) {
  const size_t lds = family == MSC_NIW ? sizeof(double) * ((size_t)dim * dim + 2 * dim) : 0;
  static unsigned long long attr_devices = 0;
  if (lds > 64u * 1024u && first_use_on_device(attr_devices))
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_value_op), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(k_value_op, dim3(1), dim3(64), lds, stream, static_cast<unsigned char *>(mailbox_dev));
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace msc
