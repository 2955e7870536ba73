// commit_ops.hpp -- additive sums -> the reference's fields for one (feature, group): the body of k_commit,
// shared with the fused commit + prepare kernel of the sweep step (kernels_score.hip k_commit_prepare).
#pragma once
#include "family_math.hpp"

namespace msc {

MSC_DEV void commit_group(const FeatDesc &fd, uint32_t k, uint32_t kpad) {
  switch (fd.family) {
    case MSC_BBNC:
    case MSC_BB:
      fd.raw_u32[k] = (uint32_t)fd.acc_i64[k];
      fd.raw_u32[kpad + k] = (uint32_t)fd.acc_i64[kpad + k];
      break;
    case MSC_GP:
      fd.raw_u32[k] = (uint32_t)fd.acc_i64[k];
      fd.raw_u32[kpad + k] = (uint32_t)fd.acc_i64[kpad + k];
      fd.raw_f32[k] = (float)fd.acc_f64[k];
      break;
    case MSC_BNB:
      fd.raw_u32[k] = (uint32_t)fd.acc_i64[k];
      fd.raw_u32[kpad + k] = (uint32_t)fd.acc_i64[kpad + k];
      break;
    case MSC_DM:
      for (uint32_t i = 0; i < fd.dim; i++) fd.raw_u32[(size_t)i * kpad + k] = (uint32_t)fd.acc_i64[(size_t)i * kpad + k];
      fd.raw_f32[k] = (float)fd.acc_f64[k];
      break;
    case MSC_DD: {
      long long tot = 0;
      for (uint32_t i = 0; i < fd.dim; i++) {
        const long long c = fd.acc_i64[(size_t)i * kpad + k];
        fd.raw_u32[(size_t)(1 + i) * kpad + k] = (uint32_t)c;
        tot += c;
      }
      fd.raw_u32[k] = (uint32_t)tot;
    } break;
    case MSC_NICH: {
      const long long n = fd.acc_i64[k];
      const double sx = fd.acc_f64[k], sxx = fd.acc_f64[kpad + k];
      double mean = 0, ctv = 0;
      if (n > 0) mean = sx / (double)n;
      if (n > 1) {
        ctv = sxx - (double)n * mean * mean;
        if (ctv < 0) ctv = 0;
      }
      fd.raw_u32[k] = (uint32_t)n;
      fd.raw_f32[k] = (float)mean;
      fd.raw_f32[kpad + k] = (float)ctv;
    } break;
    default: break;
  }
}

}  // namespace msc
