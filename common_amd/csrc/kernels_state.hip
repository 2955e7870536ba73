// kernels_state.hip -- gfx950 kernels for the state side of the hot path:
//   k_accumulate   bulk add_value / remove_value keyed by the assignment vector
//   k_commit       additive sums (all-reduce payload) -> the reference's fields
//   k_lift         the reference's fields -> additive sums
//   k_score_data   marginal likelihood of every (feature, group)
//   k_unpack       packed row-major records -> one typed column per feature
#include "commit_ops.hpp"
#include "device_error.hpp"
#include "family_math.hpp"
#include <cstdarg>
#include <cstdio>
#include "launchers.hpp"

namespace msc {

// ---------------------------------------------------------------------------
// accumulate.  Each workgroup owns a contiguous slice of rows and, feature by
// feature, histograms it into LDS (ds_add_u32 / ds_add_u64 / ds_add_f64), then
// flushes the non-zero bins with one global atomic each.  Integer fields are
// exact whatever the order; float fields are summed in double.
// LDS layout per pass: f64[K * nf64] | u64[K * nu64] | u32[K * nu32]
// ---------------------------------------------------------------------------
struct AccShape { uint32_t nu32, nu64, nf64; };
__device__ __host__ inline AccShape acc_shape(int family, uint32_t dim_slice) {
  switch (family) {
    case MSC_BB: return {2, 0, 0};
    case MSC_BBNC: return {2, 0, 0};
    case MSC_GP: return {1, 1, 1};
    case MSC_BNB: return {1, 1, 0};
    case MSC_DM: return {0, dim_slice, 1};      // u64 per category of the slice; f64 = ratio (first slice only)
    case MSC_DD: return {dim_slice, 0, 0};
    case MSC_NICH: return {1, 0, 2};
    default: return {0, 0, 0};
  }
}

// One workgroup = one feature (blockIdx.x; nfeat = the group sizes of group_manager) x one slice of `per` rows
// (blockIdx.y).  The grid is sized for ~8 workgroups per CU (launch_accumulate), so a slice is tens of thousands of
// rows when there are many features: the histogram of a dd(32) column has 8192 (category, group) bins, and a workgroup
// that sees 4096 rows flushes ~3300 of them with a global atomic apiece -- that flush, and 64 dependent
// zero / barrier / load / barrier / flush rounds per workgroup with four rows per thread in each, was the old kernel's
// time on C3 (0.34 ms for 212 MB).  Here a workgroup makes one round, keeps eight rows per thread in flight, and
// flushes once per ~32k rows; workgroups of different features run side by side and hide each other's latencies.
__global__ __launch_bounds__(1024) void k_accumulate(const FeatDesc *__restrict__ feats, int nfeat,
                                                      uint32_t K, uint32_t kpad, uint64_t row0,
                                                      uint64_t nrows, const int32_t *__restrict__ z,
                                                      int sign, long long *__restrict__ cnt_acc,
                                                      uint32_t dd_slice, uint64_t per) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint64_t lo = (uint64_t)blockIdx.y * per;
  const uint64_t hi = lo + per < nrows ? lo + per : nrows;
  if (lo >= hi) return;
  const long long sgn = sign;
  constexpr int U = 8;                                   // rows a thread keeps in flight
  const uint32_t nt = blockDim.x;

  if ((int)blockIdx.x == nfeat) {
    uint32_t *c32 = reinterpret_cast<uint32_t *>(smem);
    for (uint32_t i = threadIdx.x; i < K; i += nt) c32[i] = 0;
    __syncthreads();
    for (uint64_t base = lo; base < hi; base += (uint64_t)U * nt) {
      int g[U];
#pragma unroll
      for (int j = 0; j < U; j++) {
        const uint64_t n = base + threadIdx.x + (uint64_t)j * nt;
        g[j] = n < hi ? z[n] : -1;
      }
#pragma unroll
      for (int j = 0; j < U; j++)
        if ((uint32_t)g[j] < K) atomicAdd(&c32[g[j]], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < K; i += nt)
      if (c32[i]) atomicAdd(reinterpret_cast<unsigned long long *>(&cnt_acc[i]),
                            (unsigned long long)(sgn * (long long)c32[i]));
    return;
  }

  const FeatDesc fd = feats[blockIdx.x];
  // dd and dm walk their categories in slices that fit the LDS budget (dm bins are 8 bytes wide)
  const bool sliced = fd.family == MSC_DD || fd.family == MSC_DM;
  const uint32_t slice = fd.family == MSC_DM ? (dd_slice > 1 ? dd_slice / 2 : 1) : dd_slice;
  const uint32_t nslices = sliced ? (fd.dim + slice - 1) / slice : 1;
  for (uint32_t sl = 0; sl < nslices; sl++) {
    const uint32_t c_lo = sl * slice;
    const uint32_t c_n = sliced ? (fd.dim - c_lo < slice ? fd.dim - c_lo : slice) : 0;
    const bool fused = fd.fuse_n >= 2;                             // (bb members: heads | tails of member j at rows 2 j, 2 j + 1)
    const AccShape sh = fused ? AccShape{2u * fd.fuse_n, 0u, 0u} : acc_shape(fd.family, c_n);
    double *f64 = reinterpret_cast<double *>(smem);
    unsigned long long *u64 = reinterpret_cast<unsigned long long *>(f64 + (size_t)K * sh.nf64);
    uint32_t *u32 = reinterpret_cast<uint32_t *>(u64 + (size_t)K * sh.nu64);
    for (uint32_t i = threadIdx.x; i < K * sh.nf64; i += nt) f64[i] = 0.0;
    for (uint32_t i = threadIdx.x; i < K * sh.nu64; i += nt) u64[i] = 0ull;
    for (uint32_t i = threadIdx.x; i < K * sh.nu32; i += nt) u32[i] = 0u;
    __syncthreads();
    const bool scalar_col = fd.family == MSC_BB || fd.family == MSC_BBNC || fd.family == MSC_GP || fd.family == MSC_BNB ||
                            fd.family == MSC_DD || fd.family == MSC_NICH;
    if (scalar_col && fd.mask == nullptr) {
      // an unmasked scalar column: U (group, value) pairs per thread go out together before the first LDS atomic
      const bool u8 = fd.family == MSC_BB || fd.family == MSC_BBNC;
      // four consecutive rows a load where the addresses allow (16 bytes of z, 16 or 4 of the column): a quarter of the
      // memory instructions for the same bytes (sixteen bb columns alone 50 -> 35 us; fetching the next batch ahead of
      // this batch's atomics, tried on top, nearly doubled every family's time)
      const bool vec4 = hi - lo >= 4 && ((lo | row0) & 3) == 0 && (reinterpret_cast<uintptr_t>(z) & 15) == 0 &&
                        (reinterpret_cast<uintptr_t>(fd.col) & 15) == 0;
      for (uint64_t base = lo; base < hi; base += (uint64_t)U * nt) {
        int g[U];
        uint32_t w[U];
        if (vec4) {
#pragma unroll
          for (int j = 0; j < U / 4; j++) {
            const uint64_t n = base + 4 * ((uint64_t)threadIdx.x + (uint64_t)j * nt);
            const bool whole = n + 3 < hi;
            const uint64_t ns = whole ? n : lo;                   // (a whole quad of the slice; a partial last quad is redone below)
            const int4 zq = *reinterpret_cast<const int4 *>(z + ns);
            uint32_t q0, q1, q2, q3;
            if (u8) {
              const uint32_t b = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(fd.col) + row0 + ns);
              q0 = b & 0xffu; q1 = (b >> 8) & 0xffu; q2 = (b >> 16) & 0xffu; q3 = b >> 24;
            } else {
              const uint4 b = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint32_t *>(fd.col) + row0 + ns);
              q0 = b.x; q1 = b.y; q2 = b.z; q3 = b.w;
            }
            g[4 * j] = whole ? zq.x : -1; g[4 * j + 1] = whole ? zq.y : -1; g[4 * j + 2] = whole ? zq.z : -1; g[4 * j + 3] = whole ? zq.w : -1;
            w[4 * j] = q0; w[4 * j + 1] = q1; w[4 * j + 2] = q2; w[4 * j + 3] = q3;
            if (!whole && n < hi) {                                // the slice's last, partial quad: row by row
#pragma unroll
              for (int c = 0; c < 4; c++) {
                const uint64_t m = n + c;
                if (m < hi) {
                  g[4 * j + c] = z[m];
                  w[4 * j + c] = u8 ? (uint32_t)reinterpret_cast<const uint8_t *>(fd.col)[row0 + m] : reinterpret_cast<const uint32_t *>(fd.col)[row0 + m];
                }
              }
            }
          }
        } else {
#pragma unroll
          for (int j = 0; j < U; j++) {
            const uint64_t n = base + threadIdx.x + (uint64_t)j * nt;
            g[j] = n < hi ? z[n] : -1;
          }
#pragma unroll
          for (int j = 0; j < U; j++) {
            const uint64_t n = base + threadIdx.x + (uint64_t)j * nt;
            const uint64_t row = row0 + (n < hi ? n : lo);           // (a row of the slice whatever this thread's share is)
            w[j] = u8 ? (uint32_t)reinterpret_cast<const uint8_t *>(fd.col)[row] : reinterpret_cast<const uint32_t *>(fd.col)[row];
          }
        }
#pragma unroll
        for (int j = 0; j < U; j++) {
          if ((uint32_t)g[j] >= K) continue;
          const uint32_t gg = (uint32_t)g[j], raw = w[j];
          if (fused) {                                           // the byte's digits: member j's value (2 = masked: not counted)
            uint32_t rest = raw;
            for (uint32_t m = 0; m < fd.fuse_n; m++) {
              const uint32_t d = fd.fuse_radix == 2 ? (rest & 1u) : rest % 3u;
              rest = fd.fuse_radix == 2 ? rest >> 1 : rest / 3u;
              if (d < 2u) atomicAdd(&u32[(size_t)(2u * m + (d != 0 ? 0u : 1u)) * K + gg], 1u);
            }
            continue;
          }
          switch (fd.family) {
            case MSC_BBNC:
            case MSC_BB: atomicAdd(&u32[(raw != 0 ? 0 : K) + gg], 1u); break;
            case MSC_GP:
              atomicAdd(&u32[gg], 1u);
              atomicAdd(&u64[gg], (unsigned long long)raw);
              atomicAdd(&f64[gg], log_factorial(raw));        // ln Gamma(v + 1)
              break;
            case MSC_BNB:
              atomicAdd(&u32[gg], 1u);
              atomicAdd(&u64[gg], (unsigned long long)raw);
              break;
            case MSC_DD: {
              const int v = (int)raw;
              if (v >= (int)c_lo && v < (int)(c_lo + c_n)) atomicAdd(&u32[(size_t)(v - c_lo) * K + gg], 1u);
            } break;
            default: {                                             // nich
              const double x = __uint_as_float(raw);
              atomicAdd(&u32[gg], 1u);
              atomicAdd(&f64[gg], x);
              atomicAdd(&f64[K + gg], x * x);
            } break;
          }
        }
      }
    } else
    for (uint64_t n = lo + threadIdx.x; n < hi; n += nt) {
      const int g = z[n];
      if (g < 0 || (uint32_t)g >= K) continue;
      const uint64_t row = row0 + n;
      if (fd.mask != nullptr) {                                    // masked value: not part of the group
        bool m = false;
        if (fd.family == MSC_DM) for (uint32_t e = 0; e < fd.dim; e++) m |= fd.mask[row * fd.dim + e] != 0;
        else m = fd.mask[row] != 0;
        if (m) continue;
      }
      switch (fd.family) {
        case MSC_BBNC:
        case MSC_BB: {
          const bool v = reinterpret_cast<const uint8_t *>(fd.col)[row] != 0;
          atomicAdd(&u32[(v ? 0 : K) + g], 1u);
        } break;
        case MSC_GP: {
          const uint32_t v = reinterpret_cast<const uint32_t *>(fd.col)[row];
          atomicAdd(&u32[g], 1u);
          atomicAdd(&u64[g], (unsigned long long)v);
          atomicAdd(&f64[g], log_factorial(v));        // ln Gamma(v + 1)
        } break;
        case MSC_BNB: {
          const uint32_t v = reinterpret_cast<const uint32_t *>(fd.col)[row];
          atomicAdd(&u32[g], 1u);
          atomicAdd(&u64[g], (unsigned long long)v);
        } break;
        case MSC_DM: {
          const int32_t *x = reinterpret_cast<const int32_t *>(fd.col) + row * fd.dim;
          for (uint32_t i = 0; i < c_n; i++) {
            const uint32_t v = (uint32_t)x[c_lo + i];
            if (v) atomicAdd(&u64[(size_t)i * K + g], (unsigned long long)v);
          }
          if (sl == 0) atomicAdd(&f64[g], dm_row_ratio(fd.dim, x));
        } break;
        case MSC_DD: {
          const int v = reinterpret_cast<const int32_t *>(fd.col)[row];
          if (v >= (int)c_lo && v < (int)(c_lo + c_n)) atomicAdd(&u32[(size_t)(v - c_lo) * K + g], 1u);
        } break;
        case MSC_NICH: {
          const double x = reinterpret_cast<const float *>(fd.col)[row];
          atomicAdd(&u32[g], 1u);
          atomicAdd(&f64[g], x);
          atomicAdd(&f64[K + g], x * x);
        } break;
        default: break;
      }
    }
    __syncthreads();
    // flush: row r of the pass maps to a row of the feature's additive tables
    for (uint32_t i = threadIdx.x; i < K * sh.nu32; i += nt) {
      const uint32_t r = i / K, k = i - r * K;
      if (!u32[i]) continue;
      uint32_t dst_row = r;
      if (fd.family == MSC_DD) dst_row = c_lo + r;
      // (no run-time index into the descriptor's array: that would move the whole descriptor copy to scratch memory)
      const uint32_t mem = r >> 1;
      long long *const mbase = mem == 0 ? fd.fuse_acc[0] : mem == 1 ? fd.fuse_acc[1] : mem == 2 ? fd.fuse_acc[2] : fd.fuse_acc[3];
      long long *dst = fused ? mbase + (size_t)(r & 1u) * kpad + k : &fd.acc_i64[(size_t)dst_row * kpad + k];
      atomicAdd(reinterpret_cast<unsigned long long *>(dst), (unsigned long long)(sgn * (long long)u32[i]));
    }
    for (uint32_t i = threadIdx.x; i < K * sh.nu64; i += nt) {
      const uint32_t r = i / K, k = i - r * K;
      if (!u64[i]) continue;
      // gp, bnb: u64 row 0 is `sum`, additive row 1; dm: category c_lo + r
      const uint32_t dst_row = fd.family == MSC_DM ? c_lo + r : 1 + r;
      atomicAdd(reinterpret_cast<unsigned long long *>(&fd.acc_i64[(size_t)dst_row * kpad + k]),
                (unsigned long long)(sgn * (long long)u64[i]));
    }
    for (uint32_t i = threadIdx.x; i < K * sh.nf64; i += nt) {
      const uint32_t r = i / K, k = i - r * K;
      if (f64[i] != 0.0) atomicAdd(&fd.acc_f64[(size_t)r * kpad + k], (double)sign * f64[i]);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// accumulate when the per-workgroup histograms no longer fit LDS (K beyond ~8000 groups): one thread per
// (row, feature) -- blockIdx.y = 0 the group sizes, 1 + f feature f -- adding straight into the additive tables.
// With that many groups the rows of a wave rarely meet in one, which is what the LDS stage was there to absorb.
// ---------------------------------------------------------------------------
MSC_DEV void add_i64(long long *p, long long v) { atomicAdd(reinterpret_cast<unsigned long long *>(p), (unsigned long long)v); }
__global__ __launch_bounds__(256) void k_accumulate_global(const FeatDesc *__restrict__ feats, uint32_t K, uint32_t kpad,
                                                            uint64_t row0, uint64_t nrows, const int32_t *__restrict__ z,
                                                            int sign, long long *__restrict__ cnt_acc) {
  const uint64_t n = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= nrows) return;
  const int g = z[n];
  if (g < 0 || (uint32_t)g >= K) return;
  const long long sgn = sign;
  if (blockIdx.y == 0) {
    add_i64(&cnt_acc[g], sgn);
    return;
  }
  const FeatDesc fd = feats[blockIdx.y - 1];
  const uint64_t row = row0 + n;
  if (fd.mask != nullptr) {
    bool m = false;
    if (fd.family == MSC_DM) for (uint32_t e = 0; e < fd.dim; e++) m |= fd.mask[row * fd.dim + e] != 0;
    else m = fd.mask[row] != 0;
    if (m) return;
  }
  switch (fd.family) {
    case MSC_BBNC:
    case MSC_BB:
      add_i64(&fd.acc_i64[(reinterpret_cast<const uint8_t *>(fd.col)[row] != 0 ? 0 : kpad) + g], sgn);
      break;
    case MSC_GP: {
      const uint32_t v = reinterpret_cast<const uint32_t *>(fd.col)[row];
      add_i64(&fd.acc_i64[g], sgn);
      add_i64(&fd.acc_i64[kpad + g], sgn * (long long)v);
      atomicAdd(&fd.acc_f64[g], (double)sign * log_factorial(v));
    } break;
    case MSC_BNB: {
      const uint32_t v = reinterpret_cast<const uint32_t *>(fd.col)[row];
      add_i64(&fd.acc_i64[g], sgn);
      add_i64(&fd.acc_i64[kpad + g], sgn * (long long)v);
    } break;
    case MSC_DM: {
      const int32_t *x = reinterpret_cast<const int32_t *>(fd.col) + row * fd.dim;
      for (uint32_t i = 0; i < fd.dim; i++)
        if (x[i]) add_i64(&fd.acc_i64[(size_t)i * kpad + g], sgn * (long long)(uint32_t)x[i]);
      atomicAdd(&fd.acc_f64[g], (double)sign * dm_row_ratio(fd.dim, x));
    } break;
    case MSC_DD: {
      const int v = reinterpret_cast<const int32_t *>(fd.col)[row];
      if (v >= 0 && v < (int)fd.dim) add_i64(&fd.acc_i64[(size_t)v * kpad + g], sgn);
    } break;
    case MSC_NICH: {
      const double x = reinterpret_cast<const float *>(fd.col)[row];
      add_i64(&fd.acc_i64[g], sgn);
      atomicAdd(&fd.acc_f64[g], (double)sign * x);
      atomicAdd(&fd.acc_f64[kpad + g], (double)sign * x * x);
    } break;
    default: break;
  }
}

// ---------------------------------------------------------------------------
// commit: additive -> raw.  blockIdx.y = feature (nfeat = the group-size table)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_commit(const FeatDesc *__restrict__ feats, int nfeat,
                                                 uint32_t kpad, const long long *__restrict__ cnt_acc,
                                                 uint32_t *__restrict__ cnt_u32) {
  const uint32_t k = blockIdx.x * 256 + threadIdx.x;
  if (k >= kpad) return;
  if ((int)blockIdx.y == nfeat) {
    cnt_u32[k] = (uint32_t)cnt_acc[k];
    return;
  }
  commit_group(feats[blockIdx.y], k, kpad);
}

__global__ __launch_bounds__(256) void k_lift(const FeatDesc *__restrict__ feats, int nfeat,
                                               uint32_t kpad, long long *__restrict__ cnt_acc,
                                               const uint32_t *__restrict__ cnt_u32, int lift_cnt) {
  const uint32_t k = blockIdx.x * 256 + threadIdx.x;
  if (k >= kpad) return;
  if ((int)blockIdx.y == nfeat) {
    if (lift_cnt) cnt_acc[k] = cnt_u32[k];
    return;
  }
  const FeatDesc fd = feats[blockIdx.y];
  if (fd.acc_i64 == nullptr) return;   // feature already in sync (host passes null to skip)
  switch (fd.family) {
    case MSC_BBNC:
    case MSC_BB:
      fd.acc_i64[k] = fd.raw_u32[k];
      fd.acc_i64[kpad + k] = fd.raw_u32[kpad + k];
      break;
    case MSC_GP:
      fd.acc_i64[k] = fd.raw_u32[k];
      fd.acc_i64[kpad + k] = fd.raw_u32[kpad + k];
      fd.acc_f64[k] = fd.raw_f32[k];
      break;
    case MSC_BNB:
      fd.acc_i64[k] = fd.raw_u32[k];
      fd.acc_i64[kpad + k] = fd.raw_u32[kpad + k];
      break;
    case MSC_DM:
      for (uint32_t i = 0; i < fd.dim; i++) fd.acc_i64[(size_t)i * kpad + k] = fd.raw_u32[(size_t)i * kpad + k];
      fd.acc_f64[k] = fd.raw_f32[k];
      break;
    case MSC_DD:
      for (uint32_t i = 0; i < fd.dim; i++)
        fd.acc_i64[(size_t)i * kpad + k] = fd.raw_u32[(size_t)(1 + i) * kpad + k];
      break;
    case MSC_NICH: {
      const double n = fd.raw_u32[k], mean = fd.raw_f32[k], ctv = fd.raw_f32[kpad + k];
      fd.acc_i64[k] = fd.raw_u32[k];
      fd.acc_f64[k] = n * mean;
      fd.acc_f64[kpad + k] = ctv + n * mean * mean;
    } break;
    default: break;
  }
}

// ---------------------------------------------------------------------------
// score_data for every (feature, group): out[f * K + k]
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_score_data(const FeatDesc *__restrict__ feats, uint32_t K,
                                                     uint32_t kpad, float *__restrict__ out) {
  const FeatDesc fd = feats[blockIdx.y];
  const uint32_t k = blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  double s = 0;
  switch (fd.family) {
    case MSC_BB: s = bb_score_data(fd.hp, fd.raw_u32[k], fd.raw_u32[kpad + k]); break;
    case MSC_BBNC: s = bbnc_score_data(fd.hp, fd.raw_u32[k], fd.raw_u32[kpad + k], fd.raw_f32[k]); break;
    case MSC_GP: s = gp_score_data(fd.hp, fd.raw_u32[k], fd.raw_u32[kpad + k], (double)fd.raw_f32[k]); break;
    case MSC_DD: {
      double asum = 0;
      for (uint32_t i = 0; i < fd.dim; i++) {
        const double a = fd.hp[i];
        asum += a;
        s += lgamma(a + (double)fd.raw_u32[(size_t)(1 + i) * kpad + k]) - lgamma(a);
      }
      s += lgamma(asum) - lgamma(asum + (double)fd.raw_u32[k]);
    } break;
    case MSC_NICH: s = nich_score_data(fd.hp, fd.raw_u32[k], fd.raw_f32[k], fd.raw_f32[kpad + k]); break;
    case MSC_BNB: s = bnb_score_data(fd.hp, fd.raw_u32[k], fd.raw_u32[kpad + k]); break;
    case MSC_DM: s = dm_score_data(fd.hp, fd.dim, fd.raw_u32 + k, kpad, (double)fd.raw_f32[k]); break;
    default: break;
  }
  out[(size_t)blockIdx.y * K + k] = (float)s;
}

// ---------------------------------------------------------------------------
// unpack: one thread per (row, feature element).  src are packed records;
// element e of feature f lands at dst[f][row * count + e] converted src->dst type.
// ---------------------------------------------------------------------------

template <typename D>
MSC_DEV void store_as(void *dst, uint64_t idx, int src_type, const uint8_t *px) {
  // px may be unaligned inside a packed record: assemble the value bytewise
  unsigned long long raw = 0;
  const int n = src_type == MSC_TYPE_B || src_type == MSC_TYPE_I8 || src_type == MSC_TYPE_U8 ? 1
              : src_type == MSC_TYPE_I16 || src_type == MSC_TYPE_U16 ? 2
              : src_type == MSC_TYPE_I32 || src_type == MSC_TYPE_U32 || src_type == MSC_TYPE_F32 ? 4 : 8;
  for (int i = 0; i < n; i++) raw |= (unsigned long long)px[i] << (8 * i);
  D out;
  switch (src_type) {
    case MSC_TYPE_B: out = (D)(raw != 0); break;
    case MSC_TYPE_I8: out = (D)(int8_t)raw; break;
    case MSC_TYPE_U8: out = (D)(uint8_t)raw; break;
    case MSC_TYPE_I16: out = (D)(int16_t)raw; break;
    case MSC_TYPE_U16: out = (D)(uint16_t)raw; break;
    case MSC_TYPE_I32: out = (D)(int32_t)raw; break;
    case MSC_TYPE_U32: out = (D)(uint32_t)raw; break;
    case MSC_TYPE_I64: out = (D)(long long)raw; break;
    case MSC_TYPE_U64: out = (D)raw; break;
    case MSC_TYPE_F32: out = (D)__uint_as_float((uint32_t)raw); break;
    default: out = (D)__longlong_as_double((long long)raw); break;
  }
  reinterpret_cast<D *>(dst)[idx] = out;
}

__global__ __launch_bounds__(256) void k_unpack(const uint8_t *__restrict__ records,
                                                 const uint8_t *__restrict__ mask, uint64_t nrows,
                                                 uint32_t rowsize, uint32_t maskrowsize,
                                                 const UnpackFeat *__restrict__ feats, uint32_t nfeat) {
  const uint64_t row = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= nrows) return;
  const uint8_t *rec = records + row * rowsize;
  for (uint32_t f = 0; f < nfeat; f++) {
    const UnpackFeat uf = feats[f];
    const uint32_t psz = uf.src_type <= MSC_TYPE_U8 ? 1 : uf.src_type <= MSC_TYPE_U16 ? 2
                       : (uf.src_type <= MSC_TYPE_U32 || uf.src_type == MSC_TYPE_F32) ? 4 : 8;
    for (uint32_t e = 0; e < uf.count; e++) {
      const uint8_t *px = rec + uf.offset + e * psz;
      const uint64_t idx = row * uf.count + e;
      switch (uf.dst_type) {
        case MSC_TYPE_B: {
          double tmp;
          store_as<double>(&tmp, 0, uf.src_type, px);
          reinterpret_cast<uint8_t *>(uf.dst)[idx] = tmp != 0.0;
        } break;
        case MSC_TYPE_I8: store_as<int8_t>(uf.dst, idx, uf.src_type, px); break;
        case MSC_TYPE_U8: store_as<uint8_t>(uf.dst, idx, uf.src_type, px); break;
        case MSC_TYPE_I16: store_as<int16_t>(uf.dst, idx, uf.src_type, px); break;
        case MSC_TYPE_U16: store_as<uint16_t>(uf.dst, idx, uf.src_type, px); break;
        case MSC_TYPE_I32: store_as<int32_t>(uf.dst, idx, uf.src_type, px); break;
        case MSC_TYPE_U32: store_as<uint32_t>(uf.dst, idx, uf.src_type, px); break;
        case MSC_TYPE_I64: store_as<long long>(uf.dst, idx, uf.src_type, px); break;
        case MSC_TYPE_U64: store_as<unsigned long long>(uf.dst, idx, uf.src_type, px); break;
        case MSC_TYPE_F32: store_as<float>(uf.dst, idx, uf.src_type, px); break;
        default: store_as<double>(uf.dst, idx, uf.src_type, px); break;
      }
      if (uf.dst_mask) uf.dst_mask[idx] = mask ? mask[row * maskrowsize + uf.mask_offset + e] : 0;
    }
  }
}

// two to four bool columns as one byte column of their digits (digit j = column j's value): what a fused bb feature of
// the score / sweep plan reads (FeatDesc::fuse_*)
struct PackCols { const uint8_t *c[4]; };
// (radix 2: digit = value != 0; radix 3: the columns are the mask-folded copies -- 0, 1, 2 = masked)
__global__ __launch_bounds__(256) void k_pack_bits(PackCols cols, int m, uint32_t radix, uint64_t n, uint8_t *__restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t v = 0, place = 1;
  for (int j = 0; j < m; j++) {
    const uint32_t c = cols.c[j][i];
    v += (radix == 2 ? (c != 0 ? 1u : 0u) : (c > 2u ? 2u : c)) * place;
    place *= radix;
  }
  out[i] = (uint8_t)v;
}
// float columns as one row-major matrix (msc::NichPos: the x matrix); a 64-row tile through LDS so that both sides move
// whole lines
__global__ __launch_bounds__(256) void k_pack_nich_x(const float *const *__restrict__ cols, uint32_t n2, uint32_t n2p, uint64_t n,
                                                     float *__restrict__ out) {
  __shared__ float tile[64][65];
  const uint64_t r0 = (uint64_t)blockIdx.x * 64;
  for (uint32_t c0 = 0; c0 < n2p; c0 += 64) {
    for (uint32_t e = threadIdx.x; e < 64 * 64; e += 256) {
      const uint32_t c = e >> 6, r = e & 63u;
      tile[r][c] = (c0 + c < n2 && r0 + r < n) ? cols[c0 + c][r0 + r] : 0.f;
    }
    __syncthreads();
    const uint32_t w = n2p - c0 < 64u ? n2p - c0 : 64u;
    for (uint32_t e = threadIdx.x; e < 64 * w; e += 256) {
      const uint32_t r = e / w, c = e % w;
      if (r0 + r < n) out[(r0 + r) * n2p + c0 + c] = tile[r][c];
    }
    __syncthreads();
  }
}
int launch_pack_nich_x(hipStream_t stream, const float *const *cols_dev, uint32_t n2, uint32_t n2p, uint64_t n, float *out) {
  if (n == 0 || n2 == 0) return 0;
  hipLaunchKernelGGL(k_pack_nich_x, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, stream, cols_dev, n2, n2p, n, out);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// the lookup index matrix of a plan's first phase (FeatDesc::lk_idx): byte `byte_at` of row r's record = the feature's slot
// row for that row -- grp_off + min(max(value, 0), clamp), the value a byte (bool / fused digits) or a 32-bit integer as the
// lookup runs read it (score_block.hpp).  One thread a row; the record's unused bytes are zero.
__global__ __launch_bounds__(256) void k_pack_look_idx(const LookIdxSrc *__restrict__ src, uint32_t nsrc, uint32_t l4, uint64_t n,
                                                       uint32_t *__restrict__ out) {
  const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  uint32_t *rec = out + r * l4;
  uint32_t word = 0u, at = 0u;                           // the dword being filled and its index
  for (uint32_t i = 0; i < nsrc; i++) {
    const LookIdxSrc f = src[i];
    const int v = f.kind == MSC_KIND_LOOKUP_U8 ? (int)static_cast<const uint8_t *>(f.col)[r] : static_cast<const int *>(f.col)[r];
    const uint32_t row = f.grp_off + (uint32_t)(v < 0 ? 0 : (v > (int)f.clamp ? (int)f.clamp : v));
    const uint32_t dw = f.byte_at >> 2;                  // (byte_at ascends with i)
    while (at < dw) rec[at++] = word, word = 0u;
    word |= (row & 0xffu) << (8u * (f.byte_at & 3u));
  }
  while (at < l4) rec[at++] = word, word = 0u;
}
static char g_last_kernel[2][160] = {"", ""};
void note_kernel(int slot, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_kernel[slot & 1], sizeof g_last_kernel[0], fmt, ap);
  va_end(ap);
}
const char *last_kernel(int slot) { return g_last_kernel[slot & 1]; }

int launch_pack_look_idx(hipStream_t stream, const LookIdxSrc *src_dev, uint32_t nsrc, uint32_t l4, uint64_t n, uint32_t *out) {
  if (n == 0 || nsrc == 0) return 0;
  hipLaunchKernelGGL(k_pack_look_idx, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src_dev, nsrc, l4, n, out);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_pack_bits(hipStream_t stream, const void *const *cols, int m, uint32_t radix, uint64_t n, void *out) {
  if (n == 0) return 0;
  PackCols pc;
  for (int j = 0; j < 4; j++) pc.c[j] = static_cast<const uint8_t *>(cols[j < m ? j : 0]);
  hipLaunchKernelGGL(k_pack_bits, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, pc, m, radix, n, static_cast<uint8_t *>(out));
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// the tables of the plan's fused bb runs: row i of a fused feature = its members' rows (i >> j) & 1, summed in member
// order (floats, one association everywhere: every kernel that scores the run reads this table).  One block per feature
// of the plan's first phase; the others leave at once.
// Blocks beyond the first phase (launched only when the plan has a nich block of two or more features): what the block
// arithmetic of the plain nich features goes by (family_math.hpp "nich BLOCKS"), from the tables as they stand now --
//   xlim    = (2^15 - max_g |s mu|) / max_g s over ALL kpad groups: a value within it cannot make any group's |a| exceed
//             2^15 (-1 when the tables hold something that is not a finite number: every row is "far" then);
//   blk_ok  = at a block's first feature of two or more: every member's c1 ln2 row is bit-equal to the first's (suff-stats
//             set feature by feature need not agree on the counts); 0 everywhere else.
__global__ __launch_bounds__(256) void k_fuse_tables(const FeatDesc *__restrict__ feats, int nsplit, uint32_t kpad) {
  const FeatDesc &fd = feats[blockIdx.x];
  if ((int)blockIdx.x >= nsplit) {
    // the role-split kernels' copy of the second phase (msc::NichPos): this position's five rows, and -- the first
    // position's block -- the summed c0 of all of them, summed as nich_c0_sum does (plan order, from zero)
    const FeatDesc &head = feats[nsplit];
    float *const pack = head.rn_pack;
    const uint32_t pos = blockIdx.x - (uint32_t)nsplit;
    if (pack != nullptr) {
      float *mine = pack + (size_t)(1 + kNichPackRows * pos) * kpad;
      for (uint32_t k = threadIdx.x; k < kpad; k += 256) {
        mine[k] = fd.tab[(size_t)NICH_MU_HI * kpad + k];
        mine[(size_t)kpad + k] = fd.tab[(size_t)NICH_MU_LO * kpad + k];
        mine[(size_t)2 * kpad + k] = fd.tab[(size_t)NICH_C2 * kpad + k];
        mine[(size_t)3 * kpad + k] = fd.tab[(size_t)NICH_C1LN2 * kpad + k];
        mine[(size_t)4 * kpad + k] = fd.tab[(size_t)NICH_C1 * kpad + k];
        if (pos == 0) {
          float c0s = 0.f;
          for (uint32_t j = 0; j < head.rn_n2; j++) c0s += feats[nsplit + j].tab[(size_t)NICH_C0 * kpad + k];
          pack[k] = c0s;
        }
      }
    }
    if (fd.nich_info == nullptr) return;                   // (no block of two or more in the plan: no far rows, NichPos as the host set it)
    float smax = 0.f, bmax = 0.f;
    bool fine = true, same = true;
    const bool leads = fd.blk_first == blockIdx.x && fd.blk_end > blockIdx.x + 1u;
    for (uint32_t k = threadIdx.x; k < kpad; k += 256) {
      const float s = fd.tab[(size_t)NICH_C2 * kpad + k];
      const float b = __builtin_fabsf(fd.tab[(size_t)NICH_MU_HI * kpad + k]) + __builtin_fabsf(fd.tab[(size_t)NICH_MU_LO * kpad + k]);
      fine &= s > 0.f && s < INFINITY && b < INFINITY;         // (false for NaN)
      smax = fmaxf(smax, s);
      bmax = fmaxf(bmax, b);
      if (leads) {
        const uint32_t mine = __float_as_uint(fd.tab[(size_t)NICH_C1LN2 * kpad + k]);
        for (uint32_t j = blockIdx.x + 1; j < fd.blk_end; j++) same &= __float_as_uint(feats[j].tab[(size_t)NICH_C1LN2 * kpad + k]) == mine;
      }
    }
    // (maxima over the block: across a wave by shuffles, the four waves through LDS -- thread 0 walking 256 LDS entries one
    // after the other was 10 of this kernel's 18 us, at the head of every scoring / sweep call)
    __shared__ float s_s[4], s_b[4];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      smax = fmaxf(smax, __shfl_xor(smax, off));
      bmax = fmaxf(bmax, __shfl_xor(bmax, off));
    }
    if ((threadIdx.x & 63) == 0) s_s[threadIdx.x >> 6] = smax, s_b[threadIdx.x >> 6] = bmax;
    const int all_fine = __syncthreads_and(fine ? 1 : 0), all_same = __syncthreads_and(same ? 1 : 0);
    if (threadIdx.x == 0) {
      for (int i = 1; i < 4; i++) smax = fmaxf(smax, s_s[i]), bmax = fmaxf(bmax, s_b[i]);
      const float xlim = (all_fine && bmax < kNichFarA) ? (kNichFarA - bmax) / smax : -1.f;
      fd.nich_info->xlim = xlim;
      fd.nich_info->blk_ok = (leads && all_same) ? 1u : 0u;
      if (head.rn_pos != nullptr) head.rn_pos[pos].xlim = xlim, head.rn_pos[pos].blk_ok = (leads && all_same) ? 1u : 0u;
    }
    return;
  }
  const uint32_t m = fd.fuse_n, radix = fd.fuse_radix;
  if (m < 2) return;
  uint32_t rows = 1;
  for (uint32_t j = 0; j < m; j++) rows *= radix;
  for (uint32_t k = threadIdx.x; k < kpad; k += 256) {
    float e[4][3];                                       // the members' entries for this group (digit 2 of a masked member: its zero row)
    for (uint32_t j = 0; j < m; j++)
      for (uint32_t d = 0; d < radix; d++) e[j][d] = fd.fuse_src[j][(size_t)d * kpad + k];
    for (uint32_t i = 0; i < rows; i++) {                // (i and its digits are uniform: scalar arithmetic)
      uint32_t rest = i / radix;
      float s = e[0][i % radix];
      for (uint32_t j = 1; j < m; j++) {
        s += e[j][rest % radix];
        rest /= radix;
      }
      fd.tab[(size_t)i * kpad + k] = s;
    }
  }
}
// nblocks: the plan's first phase alone (nsplit) or the whole plan (its nich blocks' records too)
int launch_fuse_tables(hipStream_t stream, const FeatDesc *feats_dev, int nsplit, int nblocks, uint32_t kpad) {
  if (nblocks <= 0) return 0;
  hipLaunchKernelGGL(k_fuse_tables, dim3((unsigned)nblocks), dim3(256), 0, stream, feats_dev, nsplit, kpad);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// a masked lookup column with the mask folded in: masked rows hold `sentinel` (the index of the family's zero table row)
template <typename T>
__global__ __launch_bounds__(256) void k_mask_sentinel(const T *__restrict__ col, const uint8_t *__restrict__ mask, uint64_t n,
                                                        T sentinel, T *__restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = mask[i] != 0 ? sentinel : col[i];
}
int launch_mask_sentinel(hipStream_t stream, const void *col, const uint8_t *mask, uint64_t n, bool bytes, uint32_t sentinel, void *out) {
  if (n == 0) return 0;
  const dim3 grid((unsigned)((n + 255) / 256));
  if (bytes)
    hipLaunchKernelGGL(k_mask_sentinel<uint8_t>, grid, dim3(256), 0, stream, static_cast<const uint8_t *>(col), mask, n, (uint8_t)sentinel,
                       static_cast<uint8_t *>(out));
  else
    hipLaunchKernelGGL(k_mask_sentinel<uint32_t>, grid, dim3(256), 0, stream, static_cast<const uint32_t *>(col), mask, n, sentinel,
                       static_cast<uint32_t *>(out));
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// maximum of a uint32 column (sizes the gp tables)
__global__ __launch_bounds__(256) void k_col_max_u32(const uint32_t *__restrict__ col, uint64_t n,
                                                      uint32_t *__restrict__ out) {
  uint32_t m = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
    m = col[i] > m ? col[i] : m;
  for (int off = 32; off >= 1; off >>= 1) {
    const uint32_t o = (uint32_t)__shfl_xor((int)m, off, 64);
    m = o > m ? o : m;
  }
  if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

// dm column (uint32 [n][dim]): the column-wide maxima of each category and of the row totals
// (colmax[dim + 1], atomicMax; zeroed by the caller), and every row's total (rowtot[n]).  One thread per row.
__global__ __launch_bounds__(256) void k_dm_stats(const uint32_t *__restrict__ col, uint64_t n, uint32_t dim,
                                                   uint32_t *__restrict__ colmax, uint32_t *__restrict__ rowtot) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint32_t *x = col + i * dim;
  uint32_t tot = 0;
  for (uint32_t s = 0; s <= dim; s++) {
    uint32_t m = 0;
    if (i < n) {
      m = s < dim ? x[s] : tot;
      if (s < dim) tot += m;
    }
    for (int off = 32; off >= 1; off >>= 1) {
      const uint32_t o = (uint32_t)__shfl_xor((int)m, off, 64);
      m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(&colmax[s], m);
  }
  if (i < n) rowtot[i] = tot;
}

// cell -> block index of a relation (msc_relation_blocks): one thread per cell
struct RelArgs {
  uint32_t ndim;
  uint64_t shape[8];
  const int32_t *z[8];
  uint32_t ngroups[8];
};
__global__ __launch_bounds__(256) void k_relation_blocks(RelArgs a, const uint32_t *__restrict__ positions,
                                                          uint64_t ncells, int32_t *__restrict__ out) {
  const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= ncells) return;
  uint64_t rem = c;
  long long block = 0, mult = 1;
  bool ok = true;
  for (int d = (int)a.ndim - 1; d >= 0; d--) {        // last dimension fastest, in the cell order and in the block index
    uint64_t idx;
    if (positions != nullptr) idx = positions[c * a.ndim + d];
    else {
      idx = rem % a.shape[d];
      rem /= a.shape[d];
    }
    const int g = idx < a.shape[d] ? a.z[d][idx] : -1;
    ok &= g >= 0 && (uint32_t)g < a.ngroups[d];
    block += (long long)g * mult;
    mult *= a.ngroups[d];
  }
  out[c] = ok ? (int32_t)block : -1;
}

// irm's slice reduction: out[e][g] = sum over the cells c of slice (dim, e) of scores[c][g * cand_stride + off[c]] --
// "entity e joins cluster g of its domain": every cell of its slice then lies in block (g, clusters of the cell's
// other entities), whose index is g * cand_stride + off[c]; off[c] < 0 (an entity of the cell is unassigned) skips the
// cell, a masked cell scored 0 already.  One workgroup per entity; the four waves split the slice's cells, lane <->
// candidate (64 at a time), sums in double in a fixed order (bit-reproducible), one LDS pass to combine the waves.
// Dense relations enumerate the slice analytically (relation/dataview.hpp:265-407: the other dimensions in row-major
// order); a sparse one passes its rows as CSR (seg, ids).
struct SliceArgs {
  uint64_t slice_cells, inner, extent;     // dense: cells per slice, product of the dimensions after `dim`, shape[dim]
  const uint32_t *seg, *ids;               // sparse: cells of entity e are ids[seg[e] .. seg[e + 1])
};
__global__ __launch_bounds__(256) void k_relation_slice_scores(const float *__restrict__ scores, uint64_t ld, SliceArgs a,
                                                                const int32_t *__restrict__ off, uint32_t ncand,
                                                                uint32_t cand_stride, float *__restrict__ out, uint64_t ld_out) {
  __shared__ double part[4][64];
  const uint64_t e = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t beg = a.seg != nullptr ? a.seg[e] : 0, n = a.seg != nullptr ? a.seg[e + 1] - beg : a.slice_cells;
  for (uint32_t g0 = 0; g0 < ncand; g0 += 64) {
    const uint32_t g = g0 + lane;
    double acc = 0.0;
    for (uint64_t m = wave; m < n; m += 4) {
      uint64_t c;
      if (a.ids != nullptr) c = a.ids[beg + m];
      else c = ((m / a.inner) * a.extent + e) * a.inner + m % a.inner;
      const int o = off[c];                                  // (wave-uniform)
      if (o < 0) continue;
      // off is the caller's data (msc_relation_blocks with possibly other cluster counts): a block index past the
      // score row is skipped and reported, never read
      const uint64_t col = (uint64_t)g * cand_stride + (uint32_t)o;
      if (g < ncand && col < ld) acc += (double)scores[c * ld + col];
      else if (g < ncand) report_device_error(MSC_DEVERR_RELATION_RANGE, (uint32_t)c);
    }
    part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && g < ncand) out[e * ld_out + g] = (float)(((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane]);
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
MSC_DEFINE_BIND_ERROR_WORD(bind_error_word_state)

int launch_relation_slice_scores(hipStream_t stream, const float *scores, uint64_t ld, uint32_t ndim, const uint64_t *shape,
                                 uint32_t dim, const uint32_t *seg_dev, const uint32_t *ids_dev, const int32_t *off_dev,
                                 uint32_t ncand, uint32_t cand_stride, uint64_t nent, float *out_dev, uint64_t ld_out) {
  if (nent == 0 || ncand == 0) return 0;
  SliceArgs a;
  a.seg = seg_dev;
  a.ids = ids_dev;
  a.inner = 1;
  a.extent = shape[dim];
  uint64_t total = 1;
  for (uint32_t d = 0; d < ndim; d++) {
    total *= shape[d];
    if (d > dim) a.inner *= shape[d];
  }
  a.slice_cells = total / shape[dim];
  hipLaunchKernelGGL(k_relation_slice_scores, dim3((unsigned)nent), dim3(256), 0, stream, scores, ld, a, off_dev, ncand, cand_stride,
                     out_dev, ld_out);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_relation_blocks(hipStream_t stream, uint32_t ndim, const uint64_t *shape, const int32_t *const *z_dev,
                           const uint32_t *ngroups, const uint32_t *positions_dev, uint64_t ncells, int32_t *out_dev) {
  if (ncells == 0) return 0;
  RelArgs a;
  a.ndim = ndim;
  for (uint32_t d = 0; d < 8; d++) {
    a.shape[d] = d < ndim ? shape[d] : 1;
    a.z[d] = d < ndim ? z_dev[d] : nullptr;
    a.ngroups[d] = d < ndim ? ngroups[d] : 1;
  }
  hipLaunchKernelGGL(k_relation_blocks, dim3((unsigned)((ncells + 255) / 256)), dim3(256), 0, stream, a, positions_dev,
                     ncells, out_dev);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_dm_stats(hipStream_t stream, const uint32_t *col, uint64_t n, uint32_t dim, uint32_t *colmax_dev,
                    uint32_t *rowtot_dev) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_dm_stats, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, col, n, dim, colmax_dev, rowtot_dev);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_col_max_u32(hipStream_t stream, const uint32_t *col, uint64_t n, uint32_t *out_dev) {
  uint64_t blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(k_col_max_u32, dim3((unsigned)blocks), dim3(256), 0, stream, col, n, out_dev);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

size_t accumulate_lds_bytes(const FeatDesc *feats_host, int nfeat, uint32_t K, uint32_t dd_slice) {
  size_t need = (size_t)K * 4;
  for (int f = 0; f < nfeat; f++) {
    const uint32_t cap = feats_host[f].family == MSC_DM ? (dd_slice > 1 ? dd_slice / 2 : 1) : dd_slice;
    const uint32_t sl = feats_host[f].family == MSC_DD || feats_host[f].family == MSC_DM
                            ? (feats_host[f].dim < cap ? feats_host[f].dim : cap) : 0;
    AccShape sh = acc_shape(feats_host[f].family, sl);
    if (feats_host[f].fuse_n >= 2) sh = AccShape{2u * feats_host[f].fuse_n, 0u, 0u};      // a fused bb feature: its members' counters
    const size_t b = (size_t)K * (8u * sh.nf64 + 8u * sh.nu64 + 4u * sh.nu32);
    if (b > need) need = b;
  }
  return need;
}

int launch_accumulate(hipStream_t stream, int num_cus, const FeatDesc *feats_dev, const FeatDesc *feats_host,
                      int nfeat, uint32_t K, uint32_t kpad, uint64_t row0, uint64_t nrows,
                      const int32_t *z, int sign, long long *cnt_acc) {
  // pick the dd category slice so one pass fits 64 KiB of LDS
  uint32_t dd_slice = kMaxDDDim;
  while (dd_slice > 1 && (size_t)K * 4u * dd_slice > 64u * 1024u) dd_slice /= 2;
  const size_t lds = accumulate_lds_bytes(feats_host, nfeat, K, dd_slice);
  if (lds > 160u * 1024u) {                                // too many groups for LDS histograms
    hipLaunchKernelGGL(k_accumulate_global, dim3((unsigned)((nrows + 255) / 256), (unsigned)nfeat + 1), dim3(256), 0, stream,
                       feats_dev, K, kpad, row0, nrows, z, sign, cnt_acc);
    return hipGetLastError() == hipSuccess ? 0 : -1;
  }
  static unsigned long long attr_devices = 0;
  if (first_use_on_device(attr_devices))
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_accumulate),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  // a workgroup per (feature, slice of rows): ~8 workgroups per CU in all, a slice never shorter than 4096 rows (fewer
  // when that would leave most of the chip idle) -- C3: 65 x 31 workgroups of 32k rows, C2: 2 x 245 of 4096.
  constexpr int wgs_per_cu = 8;                             // (tools/scans/acc_scan.sh: 4 ... 16 within noise of each other)
  const uint64_t nf1 = (uint64_t)nfeat + 1;
  uint64_t slices = std::max<uint64_t>(1, ((uint64_t)num_cus * wgs_per_cu + nf1 - 1) / nf1);
  uint64_t per = std::max<uint64_t>((nrows + slices - 1) / slices, 4096);
  if ((nrows + per - 1) / per * nf1 < (uint64_t)num_cus)                  // few rows: spread them, down to 1024 per workgroup
    per = std::max<uint64_t>(1024, (nrows * nf1 + num_cus - 1) / num_cus);
  slices = std::max<uint64_t>(1, (nrows + per - 1) / per);
  if (slices > 65535) {
    slices = 65535;
    per = (nrows + slices - 1) / slices;
  }
  per = (per + 3) / 4 * 4;                                 // (slices start on a multiple of four rows: 16-byte loads)
  slices = std::max<uint64_t>(1, (nrows + per - 1) / per);
  // small histograms: 256-thread workgroups, several per CU; large ones (many groups) keep 1024 threads to zero and
  // flush their bins
  const unsigned threads = lds <= 40u * 1024u ? 256u : 1024u;
  hipLaunchKernelGGL(k_accumulate, dim3((unsigned)nf1, (unsigned)slices), dim3(threads), lds, stream, feats_dev, nfeat, K,
                     kpad, row0, nrows, z, sign, cnt_acc, dd_slice, per);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_commit(hipStream_t stream, const FeatDesc *feats_dev, int nfeat, uint32_t kpad,
                  const long long *cnt_acc, uint32_t *cnt_u32) {
  hipLaunchKernelGGL(k_commit, dim3((kpad + 255) / 256, nfeat + 1), dim3(256), 0, stream, feats_dev,
                     nfeat, kpad, cnt_acc, cnt_u32);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// zero two runs of 8-byte words (the additive tables).  A kernel, not hipMemsetAsync: memset nodes captured into a
// graph replayed with garbage on ROCm 7.2 (msc_sweep_step), and one launch is cheaper than two anyway.
__global__ void k_zero64(unsigned long long *__restrict__ a, size_t na, unsigned long long *__restrict__ b, size_t nb) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < na + nb; i += stride) {
    if (i < na) a[i] = 0ull;
    else b[i - na] = 0ull;
  }
}
// The additive tables as ONE float64 buffer for an all-reduce that wants one dtype (common_amd/dist.py: the payload is a
// few KB, so the exchange costs the collective's latency -- one collective, not two): counts as doubles (integers below
// 2^53 add exactly and in any order, so they come back bit-exact) followed by the float64 sums; and back.
__global__ void k_pack64(const long long *__restrict__ i64, size_t ni, const double *__restrict__ f64, size_t nf, double *__restrict__ pack) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ni + nf; i += stride)
    pack[i] = i < ni ? (double)i64[i] : f64[i - ni];
}
__global__ void k_unpack64(long long *__restrict__ i64, size_t ni, double *__restrict__ f64, size_t nf, const double *__restrict__ pack) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ni + nf; i += stride) {
    if (i < ni) i64[i] = (long long)pack[i];
    else f64[i - ni] = pack[i];
  }
}
int launch_pack64(hipStream_t stream, bool unpack, long long *i64, size_t ni, double *f64, size_t nf, double *pack) {
  if (ni + nf == 0) return 0;
  const unsigned blocks = (unsigned)std::min<size_t>((ni + nf + 255) / 256, 2048);
  if (unpack) hipLaunchKernelGGL(k_unpack64, dim3(blocks), dim3(256), 0, stream, i64, ni, f64, nf, pack);
  else hipLaunchKernelGGL(k_pack64, dim3(blocks), dim3(256), 0, stream, i64, ni, f64, nf, pack);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_zero64(hipStream_t stream, void *a, size_t na, void *b, size_t nb) {
  if (na + nb == 0) return 0;
  const unsigned blocks = (unsigned)std::min<size_t>((na + nb + 255) / 256, 2048);
  hipLaunchKernelGGL(k_zero64, dim3(blocks), dim3(256), 0, stream, static_cast<unsigned long long *>(a), na,
                     static_cast<unsigned long long *>(b), nb);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// msc_device_alloc_probed's probe: zeros written the way the score kernels write -- a wave owns a slot and visits two
// blocks of four 1 KiB rows, waves numbered so that the resident ones form two dense fronts; non-temporal
__global__ __launch_bounds__(256) void k_stream_fill(float4 *__restrict__ out, size_t nrows, size_t nslots) {
  const int lane = threadIdx.x & 63;
  const size_t slot = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (slot >= nslots) return;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  for (size_t rb = slot * 4; rb < nrows; rb += nslots * 4)
#pragma unroll
    for (int r = 0; r < 4; r++)
      if (rb + r < nrows) __builtin_nontemporal_store(zero, reinterpret_cast<f32x4 *>(out + (rb + r) * 64 + lane));
}
int launch_stream_fill(hipStream_t stream, int num_cus, void *buf, size_t nbytes) {
  (void)num_cus;
  const size_t nrows = nbytes / 1024;                    // whole KiB rows (the tail, < 1 KiB, is left to the memset)
  if (nrows == 0) return 0;
  const size_t nslots = (nrows / 4 + 1) / 2 ? (nrows / 4 + 1) / 2 : 1;
  hipLaunchKernelGGL(k_stream_fill, dim3((unsigned)((nslots + 3) / 4)), dim3(256), 0, stream, static_cast<float4 *>(buf), nrows, nslots);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

__global__ void k_set_i32(int32_t *dst, int32_t value) { *dst = value; }
int launch_set_i32(hipStream_t stream, int32_t *dst, int32_t value) {
  hipLaunchKernelGGL(k_set_i32, dim3(1), dim3(1), 0, stream, dst, value);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_lift(hipStream_t stream, const FeatDesc *feats_dev, int nfeat, uint32_t kpad,
                long long *cnt_acc, const uint32_t *cnt_u32, int lift_cnt) {
  hipLaunchKernelGGL(k_lift, dim3((kpad + 255) / 256, nfeat + 1), dim3(256), 0, stream, feats_dev, nfeat,
                     kpad, cnt_acc, cnt_u32, lift_cnt);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_score_data(hipStream_t stream, const FeatDesc *feats_dev, int nfeat, uint32_t K,
                      uint32_t kpad, float *out) {
  hipLaunchKernelGGL(k_score_data, dim3((K + 255) / 256, nfeat), dim3(256), 0, stream, feats_dev, K,
                     kpad, out);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_unpack(hipStream_t stream, const uint8_t *records, const uint8_t *mask, uint64_t nrows,
                  uint32_t rowsize, uint32_t maskrowsize, const void *feats_dev, uint32_t nfeat) {
  hipLaunchKernelGGL(k_unpack, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, stream, records, mask,
                     nrows, rowsize, maskrowsize, reinterpret_cast<const UnpackFeat *>(feats_dev), nfeat);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace msc
