// family_math.hpp -- device-side arithmetic of the component models, shared by
// every kernel (batched score, fused sweep, per-value mailbox).  gfx950 only.
//
// Split of labour
//   * "prepare" functions run once per (feature, group) in double and turn the
//     reference's suff-stats (distributions.hpp:21-56 field names) into a few
//     float constants per group, so that the per-evaluation work is a handful of
//     float instructions (the reference instead rebuilds the posterior on every
//     score_value call).
//   * "eval" functions are the per-(row, group) float work.  They are written so
//     that the result stays within 1e-6 of the double evaluation: differences of
//     large numbers are taken in the prepare step, log1p is compensated, and the
//     posterior mean is carried as a hi/lo float pair.
//   * "loo" functions evaluate one row against its own group with the row
//     removed (remove_value then score_value) in double; they run
//     lane-parallel over rows, once per row, so their cost is amortised over K.
#pragma once

#include <hip/hip_runtime.h>

#include "msc_internal.hpp"

namespace msc {

#define MSC_DEV __device__ __forceinline__

constexpr double kPi = 3.14159265358979323846;
constexpr double kLogPi = 1.1447298858494001741;
constexpr double kHalfLog2Pi = 0.91893853320467274178;
constexpr float kLn2f = 0.69314718055994530942f;

// v_log_f32 is log2 with ~1 ulp; arguments here are >= 1 so denormals never occur.
MSC_DEV float hw_log2(float x) { return __builtin_amdgcn_logf(x); }
MSC_DEV float hw_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// log2(1+t) and the first-order compensation term r such that
//   log(1+t) = ln2 * log2(u) + r,  u = fl(1+t),  r = (t - (u-1)) / u     (t >= 0)
// (u-1 is exact for u >= 1, t-(u-1) is exact, so only the division rounds.)
MSC_DEV void log1p_parts(float t, float &l2, float &r) {
#pragma clang fp contract(off)   // same bits from every kernel that inlines this, whatever surrounds it
  const float u = 1.0f + t;
  l2 = hw_log2(u);
  r = (t - (u - 1.0f)) * hw_rcp(u);
}
// The same parts of log(1 + a^2), from a: u = fma(a, a, 1) is 1 + a^2 rounded once, and fma(a, a, -(u - 1)) is what that
// rounding dropped (u - 1 is exact) -- four plain instructions + two transcendental where log1p_parts(a * a) takes five,
// and the remainder now carries the rounding of a^2 itself as well.  The nich evaluations spend their issue slots on
// exactly this: one in ten fewer.
// log2(1 + a^2) in one go: v_log_f32 of u plus the remainder times log2e / u, the latter from the exponent-flip estimate
// bits(log2e / u) ~ kLog2eOverU - bits(u) (within 1.6 % for every u >= 1, tests/test_host_numerics.py; the remainder is at
// most 2^-24, so what the estimate leaves, <= 1.4e-9, is a sixtieth of v_log_f32's own error): five plain instructions
// and ONE transcendental.
constexpr uint32_t kLog2eOverU = 0x7F35D5C7u;
MSC_DEV float log2_1p_sq(float a) {
#pragma clang fp contract(off)   // same bits from every kernel that inlines this, whatever surrounds it
  const float u = fmaf(a, a, 1.0f);
  const float ru = __uint_as_float(kLog2eOverU - __float_as_uint(u));
  return fmaf(fmaf(a, a, -(u - 1.0f)), ru, hw_log2(u));
}
MSC_DEV void log1p_sq_parts(float a, float &l2, float &r) {
#pragma clang fp contract(off)   // same bits from every kernel that inlines this, whatever surrounds it
  const float u = fmaf(a, a, 1.0f);
  l2 = hw_log2(u);
  r = fmaf(a, a, -(u - 1.0f)) * hw_rcp(u);
}
MSC_DEV float log1p_acc(float t) {
  float l2, r;
  log1p_parts(t, l2, r);
  return fmaf(l2, kLn2f, r);
}

// lgamma for x > 0 without the general routine's branches: shift up to x >= 16 with the recurrence
// (one log of the product of the skipped factors), then Stirling with five correction terms
// (next term 691 / (360360 x^11) < 1e-16 there).  Used by the per-row leave-one-out pass, which is
// lgamma-bound otherwise.
MSC_DEV double lgamma_pos(double x) {
  double shift = 0.0;
  if (x < 16.0) {
    double prod = 1.0;
    while (x < 16.0) {
      prod *= x;
      x += 1.0;
    }
    shift = log(prod);
  }
  const double r = 1.0 / x, r2 = r * r;
  const double tail = r * (1.0 / 12.0 - r2 * (1.0 / 360.0 - r2 * (1.0 / 1260.0 - r2 * (1.0 / 1680.0 - r2 * (1.0 / 1188.0)))));
  return (x - 0.5) * log(x) - x + kHalfLog2Pi + tail - shift;
}
// lgamma(a) - lgamma(a - v) for an integer 0 <= v <= a - (something positive): the log of a falling product
// for the few-factor case, two Stirling evaluations otherwise
MSC_DEV double lgamma_drop(double a, uint32_t v) {
  if (v <= 12u) {
    double prod = 1.0;
    for (uint32_t i = 1; i <= v; i++) prod *= a - (double)i;
    return log(prod);
  }
  return lgamma_pos(a) - lgamma_pos(a - (double)v);
}
// log(v!): a table for the counts that occur, Stirling beyond
__device__ const double kLogFactorial[32] = {0, 0, 0.69314718055994495, 1.7917594692280554, 3.1780538303479449, 4.7874917427820467, 6.5792512120101021, 8.5251613610654147, 10.604602902745249, 12.801827480081467, 15.104412573075514, 17.502307845873887, 19.987214495661885, 22.552163853123421, 25.191221182738683, 27.89927138384089, 30.671860106080672, 33.505073450136891, 36.395445208033053, 39.339884187199495, 42.335616460753485, 45.380138898476908, 48.47118135183522, 51.606675567764377, 54.784729398112319, 58.003605222980518, 61.261701761002008, 64.557538627006338, 67.889743137181526, 71.257038967168, 74.658236348830172, 78.092223553315307};
// (the rare branch is a real call: inlined, hipcc hoists its ~30 double constants out of the caller's row loop and
// then spills them -- k_accumulate carried 244 bytes of scratch per lane for it)
static __device__ __attribute__((noinline)) double log_factorial_big(uint32_t v) { return lgamma_pos((double)v + 1.0); }
MSC_DEV double log_factorial(uint32_t v) { return v < 32u ? kLogFactorial[v] : log_factorial_big(v); }

MSC_DEV void split_hi_lo(double v, float &hi, float &lo) {
  hi = (float)v;
  lo = (float)(v - (double)hi);
}

// ============================ Beta-Bernoulli ================================
// tab rows: 0 = log p(v = 0), 1 = log p(v = 1)
MSC_DEV void bb_prepare(const float *hp, uint32_t heads, uint32_t tails, float &s0, float &s1) {
  const double a = hp[0], b = hp[1], h = heads, t = tails;
  const double den = a + b + h + t;
  s0 = (float)log((b + t) / den);
  s1 = (float)log((a + h) / den);
}
MSC_DEV double bb_loo(const float *hp, uint32_t heads, uint32_t tails, bool v) {
  const double a = hp[0], b = hp[1];
  const double h = (double)heads - (v ? 1.0 : 0.0), t = (double)tails - (v ? 0.0 : 1.0);
  return log((v ? a + h : b + t) / (a + b + h + t));
}
MSC_DEV double bb_score_data(const float *hp, uint32_t heads, uint32_t tails) {
  const double a = hp[0], b = hp[1], h = heads, t = tails;
  return lgamma(a + b) - lgamma(a + b + h + t) + lgamma(a + h) - lgamma(a) + lgamma(b + t) -
         lgamma(b);
}

// ============== non-conjugate Beta-Bernoulli (src/models/bbnc.cpp:22-73) =================
// the group carries an explicit p; counts only enter score_data
MSC_DEV void bbnc_prepare(float p, float &s0, float &s1) {
  s0 = (float)log(1.0 - (double)p);
  s1 = (float)log((double)p);
}
MSC_DEV double bbnc_score_data(const float *hp, uint32_t heads, uint32_t tails, float pf) {
  const double p = pf, a = hp[0], b = hp[1];
  if (p < 0.0 || p > 1.0) return -INFINITY;
  const double lbeta = lgamma(a) + lgamma(b) - lgamma(a + b);
  return (a - 1.0) * log(p) + (b - 1.0) * log(1.0 - p) - lbeta + (double)heads * log(p) + (double)tails * log(1.0 - p);
}

// ============================ Dirichlet-Discrete ============================
// tab rows: i in [0, dim) -> log p(v = i)
MSC_DEV float dd_prepare_entry(float alpha_i, uint32_t count_i, double alpha_sum, uint32_t count_sum) {
  return (float)log(((double)alpha_i + (double)count_i) / (alpha_sum + (double)count_sum));
}
MSC_DEV double dd_loo(float alpha_v, uint32_t count_v, double alpha_sum, uint32_t count_sum) {
  return log(((double)alpha_v + (double)count_v - 1.0) / (alpha_sum + (double)count_sum - 1.0));
}

// ============================ Gamma-Poisson =================================
// Posterior a = alpha + sum, b = inv_beta + count; negative-binomial predictive
//   score(v) = lgamma(a+v) - lgamma(a) - lgamma(v+1) + a ln b - (a+v) ln(1+b).
// The terms are ~v ln a each while the result is O(1..v): float cannot subtract
// them to 1e-6.  So:
//  * v < vcap = min(max count of the bound column + 1, 1024): an exact per-group table built
//    in double by the prepare step (rows GP_T0 + v) -- one 16-byte load per (row, 4 groups).
//  * v >= 1024 (k_gp_large_fix, a separate pass that only exists for such columns): Loader's saddle-point form (C. Loader
//    2000, the form R's dnbinom uses), in which every term is of the size of the
//    result, evaluated in double:
//      score = -log1p(v/a)/2 - ln(2 pi v)/2 - e(v) + S(a+v) - e(a) - a g(-d/a) - v g(d/v)
//      d = (a - v b)/(1+b),  g(y) = y - log1p(y),  e(y) = stirlerr(y) = lgamma(y+1) - Stirling(y)
//    e(a) comes from the prepare step (hi/lo floats), e(v) and ln(2 pi v) once per row.
enum { GP_NSE_HI = 0, GP_NSE_LO = 1, GP_T0 = 2 };   // rows GP_T0 + v, v < FeatDesc::vcap: the exact table

MSC_DEV double gp_score_exact(double a, double b, double v) {
  return lgamma(a + v) - lgamma(a) - lgamma(v + 1.0) + a * log(b) - (a + v) * log1p(b);
}
MSC_DEV void gp_prepare_consts(const float *hp, uint32_t count, uint32_t sum, float &nse_hi, float &nse_lo) {
  const double a = (double)hp[0] + (double)sum;
  (void)count;
  const double stirlerr_a = lgamma(a + 1.0) - ((a + 0.5) * log(a) - a + kHalfLog2Pi);
  split_hi_lo(-stirlerr_a, nse_hi, nse_lo);
}
MSC_DEV float gp_prepare_table(const float *hp, uint32_t count, uint32_t sum, uint32_t v) {
  return (float)gp_score_exact((double)hp[0] + (double)sum, (double)hp[1] + (double)count, (double)v);
}
MSC_DEV double stirling_tail(double z) {  // S(z) = 1/(12z) - 1/(360z^3) + 1/(1260z^5) - 1/(1680z^7), z >= 1024
  const double r = 1.0 / z, r2 = r * r;
  return r * (1.0 / 12.0 - r2 * (1.0 / 360.0 - r2 * (1.0 / 1260.0 - r2 * (1.0 / 1680.0))));
}
// per-row constant of the large-count path: -ln(2 pi v)/2 - stirlerr(v)
MSC_DEV double gp_row_const(uint32_t v) {
  const double x = (double)v;
  const double stirlerr_v = lgamma(x + 1.0) - ((x + 0.5) * log(x) - x + kHalfLog2Pi);
  return -0.5 * log(2.0 * kPi * x) - stirlerr_v;
}
MSC_DEV float gp_eval_large(double v, double rowc, double a, double b, double nse_a) {
  const double q = 1.0 / (1.0 + b);
  const double d = (a - v * b) * q;
  const double u = -d / a, w = d / v;
  return (float)(rowc + nse_a - 0.5 * log1p(v / a) + stirling_tail(a + v) - a * (u - log1p(u)) -
                 v * (w - log1p(w)));
}
MSC_DEV double gp_score_data(const float *hp, uint32_t count, uint32_t sum, double log_prod) {
  const double al = hp[0], ib = hp[1];
  const double a = al + (double)sum, b = ib + (double)count;
  return lgamma(a) - lgamma(al) + al * log(ib) - a * log(b) - log_prod;
}

// ============================ Beta-Negative-Binomial ========================
// hp {alpha, beta, r}; posterior a = alpha + r*count, b = beta + sum; predictive
//   score(v) = lgamma(r+v) - lgamma(v+1) - lgamma(r) + lgamma(a+b) - lgamma(a) - lgamma(b)
//            + lgamma(a+r) + lgamma(b+v) - lgamma(a+r+b+v)
// Same evaluation scheme as gp: an exact per-group table (rows GP_T0 + v) for the counts the
// bound column holds, and the formula in double for counts beyond the table.
MSC_DEV double bnb_score_exact(double a, double b, double r, double v) {
  return lgamma(r + v) - lgamma(v + 1.0) - lgamma(r) + lgamma(a + b) - lgamma(a) - lgamma(b) +
         lgamma(a + r) + lgamma(b + v) - lgamma(a + r + b + v);
}
MSC_DEV double bnb_score(const float *hp, double count, double sum, double v) {
  const double r = hp[2];
  return bnb_score_exact((double)hp[0] + r * count, (double)hp[1] + sum, r, v);
}
// marginal up to the data-only term sum_i log C(r+v_i-1, v_i), which {count, sum} cannot carry
MSC_DEV double bnb_score_data(const float *hp, uint32_t count, uint32_t sum) {
  const double al = hp[0], be = hp[1], r = hp[2];
  const double a = al + r * (double)count, b = be + (double)sum;
  return lgamma(al + be) - lgamma(al) - lgamma(be) + lgamma(a) + lgamma(b) - lgamma(a + b);
}

// ============================ Dirichlet-Multinomial =========================
// src/models/dm.cpp:39-76.  With n_i the group's count of category i, N = sum n_i, A = sum alpha_i,
// X = sum x_i:
//   score(x) = sum_i [ lgamma(a_i+n_i+x_i) - lgamma(a_i+n_i) - lgamma(x_i+1) ]
//            + lgamma(A+N) - lgamma(A+N+X) + lgamma(X+1)
// i.e. dim+1 independent count lookups: one exact table per category (indexed by x_i) and one for
// the row total (indexed by X).  The tile kernel treats them as dim+1 gp-like stages.
// The terms are each ~x ln n while their sum is O(10): a float per entry cannot carry that
// cancellation.  Every entry is therefore stored as a pair: hi = the value rounded to a multiple
// of 2^-6 (|hi| < 2^17, so float sums of hi parts are EXACT while |sum| < 2^18, which holds for
// row totals below the table cap) and lo = the remainder (|lo| <= 2^-7).  The kernel sums the hi
// and lo parts of a feature's stages separately and adds them to the score once.
// Rows whose total is beyond the table cap are scored by the large-count kernel in double.
constexpr double kDmGrid = 64.0;
MSC_DEV void dm_split(double t, float &hi, float &lo) {
  const double h = rint(t * kDmGrid) / kDmGrid;
  hi = (float)h;
  lo = (float)(t - h);
}
MSC_DEV double dm_cat_term(double alpha_i, double n_i, double x) {
  return lgamma(alpha_i + n_i + x) - lgamma(alpha_i + n_i) - lgamma(x + 1.0);
}
MSC_DEV double dm_sum_term(double A, double N, double X) {
  return lgamma(A + N) - lgamma(A + N + X) + lgamma(X + 1.0);
}
// score of the vector at `x` (dim int32) against counts read with stride `cstride` (in u32),
// each lowered by the vector itself when loo
MSC_DEV double dm_score_direct(const float *hp, uint32_t dim, const uint32_t *counts, size_t cstride,
                               const int32_t *x, bool loo) {
  double s = 0, A = 0, N = 0, X = 0;
  for (uint32_t i = 0; i < dim; i++) {
    const double xi = (double)(uint32_t)x[i];
    const double ni = (double)counts[(size_t)i * cstride] - (loo ? xi : 0.0);
    s += dm_cat_term((double)hp[i], ni, xi);
    A += (double)hp[i];
    N += ni;
    X += xi;
  }
  return s + dm_sum_term(A, N, X);
}
// the row's contribution to `ratio` (dm.cpp:10-22)
MSC_DEV double dm_row_ratio(uint32_t dim, const int32_t *x) {
  double r = 0;
  uint32_t tot = 0;
  for (uint32_t i = 0; i < dim; i++) {
    r -= log_factorial((uint32_t)x[i]);
    tot += (uint32_t)x[i];
  }
  return r + log_factorial(tot);
}
MSC_DEV double dm_score_data(const float *hp, uint32_t dim, const uint32_t *counts, size_t cstride, double ratio) {
  double s = ratio, A = 0, N = 0;
  for (uint32_t i = 0; i < dim; i++) {
    const double a = hp[i], c = counts[(size_t)i * cstride];
    A += a;
    N += c;
    s += lgamma(c + a) - lgamma(a);
  }
  return s + lgamma(A) - lgamma(A + N);
}

// ============================ Normal-Inverse-Chi^2 ==========================
// Posterior (kappa', mu', nu', sigmasq') from (count, mean, count_times_variance);
// Student-t predictive  score(x) = c0 - c1 * log1p(c2 (x - mu')^2)
//   c0 = lgamma((nu'+1)/2) - lgamma(nu'/2) + ln(lambda/(pi nu'))/2,  c1 = (nu'+1)/2,
//   c2 = lambda/nu',  lambda = kappa'/((kappa'+1) sigmasq')
// NICH_MU_* hold s*mu and NICH_C2 holds s = sqrt(c2) (see nich_prepare)
enum { NICH_MU_HI = 0, NICH_MU_LO = 1, NICH_C0 = 2, NICH_C1LN2 = 3, NICH_C1 = 4, NICH_C2 = 5, NICH_ROWS = 6 };

struct NichPost { double mu, kappa, sigmasq, nu; };
MSC_DEV NichPost nich_posterior(const float *hp, double n, double mean, double ctv) {
  const double mu = hp[0], kappa = hp[1], sigmasq = hp[2], nu = hp[3];
  NichPost p;
  const double d = mu - mean;
  p.kappa = kappa + n;
  p.mu = (kappa * mu + mean * n) / p.kappa;
  p.nu = nu + n;
  p.sigmasq = (nu * sigmasq + ctv + (n * kappa * d * d) / p.kappa) / p.nu;
  return p;
}
MSC_DEV void nich_coeffs(const NichPost &p, double &c0, double &c1, double &c2) {
  const double lambda = p.kappa / ((p.kappa + 1.0) * p.sigmasq);
  c0 = lgamma(0.5 * p.nu + 0.5) - lgamma(0.5 * p.nu) + 0.5 * log(lambda / (kPi * p.nu));
  c1 = 0.5 * p.nu + 0.5;
  c2 = lambda / p.nu;
}
MSC_DEV void nich_prepare(const float *hp, uint32_t count, float mean, float ctv, float *out /*NICH_ROWS*/) {
  const NichPost p = nich_posterior(hp, (double)count, (double)mean, (double)ctv);
  double c0, c1, c2;
  nich_coeffs(p, c0, c1, c2);
  // the eval works on a = s (x - mu) with s = fl(sqrt(c2)): t = c2 (x - mu)^2 = a^2 up to a relative
  // error of a few ulp (harmless: the score depends on t through c1 log1p(t)), and
  // s mu is carried as a hi/lo pair computed from the ROUNDED s so that a is exact in x - mu.
  const float s = (float)sqrt(c2);
  split_hi_lo(p.mu * (double)s, out[NICH_MU_HI], out[NICH_MU_LO]);
  out[NICH_C0] = (float)c0;
  out[NICH_C1LN2] = (float)(c1 * 0.69314718055994530942);
  out[NICH_C1] = (float)c1;
  out[NICH_C2] = s;
}
// rows of the table: s*mu (hi, lo), c0, c1*ln2, c1, s = sqrt(c2)
MSC_DEV float nich_eval(float x, float smu_hi, float smu_lo, float c0, float c1ln2, float c1, float s) {
#pragma clang fp contract(off)   // explicit fmaf only: a row's score must not depend on the code path that scored it
  const float a = fmaf(x, s, -smu_hi) - smu_lo;
  float l2, r;
  log1p_sq_parts(a, l2, r);
  return fmaf(-c1, r, fmaf(-c1ln2, l2, c0));
}
// The tile kernels' second phase sums the nich features of a row WITHOUT a separate addition per evaluation: the
// accumulator starts at the sum of the features' c0 (a constant of the group: nich_c0_sum) and every evaluation is the two
// fused multiply-adds below on it -- 9 plain + 2 transcendental instructions where `acc += nich_eval(...)` is 10 + 2, and
// the nich phase runs at the issue rate of exactly that mix.  Same mathematics, another
// association of the float sum: every tile kernel uses this form for its plain (unmasked) nich features.
// EST (the SWEEP kernels' instantiations): c1 ln2 times log2_1p_sq -- the remainder through the exponent-flip estimate
// of log2e / u, one transcendental and one constant fewer an evaluation (C3 sweep step 2.03 -> 1.93 ms).  What the
// estimate leaves, <= 1.4e-9 c1 ln2, is far inside what a draw notices (the oracle tests place every disagreeing draw on
// a CDF step) but not inside the 1e-6 gate of score_value for groups of thousands of rows (c1 ln2 ~ 1400 at C2: 1.9e-6
// measured): the score kernels keep v_rcp_f32.
template <bool EST = false>
MSC_DEV float nich_accum(float acc, float x, float smu_hi, float smu_lo, float c1ln2, float c1, float s) {
#pragma clang fp contract(off)
  const float a = fmaf(x, s, -smu_hi) - smu_lo;
  if (EST) return fmaf(-c1ln2, log2_1p_sq(a), acc);
  float l2, r;
  log1p_sq_parts(a, l2, r);
  return fmaf(-c1, r, fmaf(-c1ln2, l2, acc));
}
// ---- nich BLOCKS: the plain nich features of a state that share c1 --------------------------------------------------
// A mixture state's unmasked nich columns with the same nu prior have, per group, the same c1 = (nu + n_g + 1) / 2 (the
// group's size is every such column's count), so over a block of M <= kNichBlock of them
//     sum_f c1 log(1 + t_f) = c1 log(1 + P),   1 + P = prod_f (1 + t_f),   t_f = a_f^2,
// and P is carried RELATIVE to P -- (1 + p)(1 + q) - 1 = p q + (p + q), one fused multiply-add and one addition, nothing
// rounds at 1 -- pairwise, so four factors are two levels of rounding.  Per evaluation: a (2 instructions), t = a a (1),
// 3/4 of a join (1.5), and a QUARTER of the compensated log1p that `acc -= c1 log1p(P)` costs (6 plain + 2 transcendental
// per block): 6 plain + 0.5 transcendental where nich_accum is 9 + 2 (the mixed kernels run at the
// issue rate of exactly this mix).  What it needs:
//   * c1 bit-equal over the block for every group -- the host groups features by their nu prior (abi.cpp plan_groups),
//     the head kernel of every scoring / sweep call (kernels_state.hip k_fuse_tables) compares the c1 ln2 rows the
//     suff-stats actually produced and clears the block's flag when they differ (suff-stats set feature by feature need
//     not agree on the counts): such a block is evaluated feature by feature, nich_accum;
//   * the product inside the float range: four factors below 2^30 each.  |a| <= s |x| + |s mu|, so a row whose values all
//     satisfy |x_f| <= xlim_f = (2^15 - max_g |s mu|) / max_g s (head kernel, per feature) cannot overflow whatever the
//     group; a row with a value beyond -- 32768 posterior scales from some group -- is a "far" row and takes nich_accum
//     for all its plain nich features.  The decision is the row's, so a row's bits do not depend on what shares its wave.
// Error: t_f carries 2.5 eps relative (a: two roundings; the square: one half), the joins add at most one eps a level
// and the compensated log1p its own, so c1 log1p(P) is within 6 eps P / ((1 + P) log1p(P)) <= 6 eps of ITSELF in the
// worst case (5.5 seen in 400k draws; the median is below one eps); nich_accum's per-feature term is within ~2.  Against
// the gate on a sum of D features, 1e-6 sum_f max(1, |score_f|) = 16.8 eps per feature, both are small;
// tests/test_host_numerics.py replays the float arithmetic against double.
// (kNichBlock, kNichFarA, NichPlanInfo: msc_internal.hpp)
MSC_DEV float nich_t(float x, float smu_hi, float smu_lo, float s) {
#pragma clang fp contract(off)
  const float a = fmaf(x, s, -smu_hi) - smu_lo;
  return a * a;
}
MSC_DEV float nich_join(float p, float q) {              // (1 + p)(1 + q) - 1
#pragma clang fp contract(off)
  return fmaf(p, q, p + q);
}
// acc - c1 log(1 + P), as c1 ln2 times log2(1 + P) = log2(u) + log2e (P - (u - 1)) / u, u = fl(1 + P): the remainder
// through v_rcp_f32 (score kernels: 6 plain + 2 transcendental) or through the exponent-flip estimate of log2e / u (EST,
// the sweeps: 6 + 1), as in log1p_parts / log2_1p_sq
template <bool EST>
MSC_DEV float nich_block_finish(float acc, float P, float c1ln2) {
#pragma clang fp contract(off)
  const float u = 1.0f + P;
  const float e = P - (u - 1.0f);
  float w;
  if (EST) w = fmaf(e, __uint_as_float(kLog2eOverU - __float_as_uint(u)), hw_log2(u));
  else w = fmaf(e * hw_rcp(u), 1.44269504088896340736f, hw_log2(u));
  return fmaf(-c1ln2, w, acc);
}
template <int M>
MSC_DEV float nich_block_product(const float (&t)[M]) {
  static_assert(M >= 2 && M <= 4, "blocks of two to four features");
  if constexpr (M == 2) return nich_join(t[0], t[1]);
  else if constexpr (M == 3) return nich_join(nich_join(t[0], t[1]), t[2]);
  else return nich_join(nich_join(t[0], t[1]), nich_join(t[2], t[3]));
}

// The sweep kernels' form, in log2 units: c0' - c1 log2(1 + t), with log2(1 + t) = log2(u) + log2e (t - (u - 1)) / u
// assembled first -- the same accuracy (the product with c1 rounds at the term's own size either way) with one
// per-group constant fewer to keep in registers than nich_eval's two factors.
MSC_DEV float nich_eval_log2(float x, float smu_hi, float smu_lo, float c0, float c1, float s) {
#pragma clang fp contract(off)
  const float a = fmaf(x, s, -smu_hi) - smu_lo;
  return fmaf(-c1, log2_1p_sq(a), c0);
}
// The transposed sweep kernel's form of the same: the compensation term log2e (1 + a^2 - u) / u is at most 2^-24 log2e, so
// log2e / u is needed to a few bits only -- the exponent-flip estimate bits(log2e / u) ~ kLog2eOverU - bits(u) (within
// 1.6 % for every u >= 1: with the factor log2e folded into the constant the piecewise-linear error straddles zero better
// than the plain reciprocal's 0x7EF311C7, 5.1 %; tests/test_host_numerics.py pins both) instead of v_rcp_f32 and a
// multiplication: the term enters with ONE fused multiply-add, eight plain + two transcendental instructions an entry.  What
// is left of the term's error, <= 1.4e-9 in log2 units, is a sixtieth of v_log_f32's own.
MSC_DEV float nich_eval_log2_est(float x, float smu_hi, float smu_lo, float c0, float c1, float s) {
  return nich_eval_log2(x, smu_hi, smu_lo, c0, c1, s);    // (one form everywhere since the estimate took log2e in)
}
// Leave-one-out (remove_value, i.e. the Welford downdate, then score_value), all in double:
//   n = count - 1, m2 = (mean count - x) / n, v2 = ctv - (x - mean)(x - m2), then the posterior and the
//   Student-t of the section header with (n, m2, v2).
// It runs through per-group constants (k_prepare fills them; FeatDesc::loo64, kNlooStride doubles per group side by
// side): what depends on the group alone and costs a division or an lgamma -- 1/n, 1/kappa_n, 1/nu_n, the constant
// term -- is computed once per group; what is a multiplication away from those (n/kappa_n = 1 - kappa/kappa_n, ...) the
// row derives itself.  Six doubles = 48 bytes = three 16-byte reads per (row, feature): the block was twelve doubles
// until the rows' gathers of it turned out to be what the leave-one-out pass spends its time on (the LDS port at a
// 96-byte stride: 16-way bank conflicts; profiles/r03_loo_own.txt).
//   STATS    the group's own float fields (mean in the low word, count_times_variance in the high one)
//   TOTAL    mean * count
//   INV_N    1 / n (0 when the row is the group's only member); n > 1 <=> 0 < INV_N < 1
//   IKN      1 / kappa_n,  kappa_n = kappa + n
//   INV_NUN  1 / nu_n,     nu_n = nu + n
//   C01      two floats: c0 = lgamma((nu_n + 1)/2) - lgamma(nu_n/2) + log(kfac / (pi nu_n)) / 2 (low word) and
//            c1 = (nu_n + 1) / 2 (high word); kfac = kappa_n / (kappa_n + 1) = 1 / (1 + IKN)
enum { NLOO_STATS = 0, NLOO_TOTAL, NLOO_INV_N, NLOO_IKN, NLOO_INV_NUN, NLOO_C01, NLOO_ROWS };
MSC_DEV void nich_loo_prepare(const float *hp, uint32_t count, float mean_f, float ctv_f, double *out, size_t stride) {
  out[NLOO_STATS * stride] = __hiloint2double(__float_as_int(ctv_f), __float_as_int(mean_f));
  const double kappa = hp[1], nu = hp[3];
  const double n = (double)count - 1.0, kn = kappa + n, nun = nu + n, kfac = kn / (kn + 1.0);
  out[NLOO_TOTAL * stride] = (double)mean_f * (double)count;
  out[NLOO_INV_N * stride] = count <= 1 ? 0.0 : 1.0 / n;
  out[NLOO_IKN * stride] = 1.0 / kn;
  out[NLOO_INV_NUN * stride] = 1.0 / nun;
  const float c0 = (float)(lgamma_pos(0.5 * nun + 0.5) - lgamma_pos(0.5 * nun) + 0.5 * log(kfac / (kPi * nun)));
  const float c1 = (float)(0.5 * nun + 0.5);
  out[NLOO_C01 * stride] = __hiloint2double(__float_as_int(c1), __float_as_int(c0));
}
// the double part both forms share: the downdated group's posterior scale `sig` and the squared distance `dd2`
MSC_DEV void nich_loo_core(const float *hp, const double *t, size_t stride, float xf, double &sig, double &dd2, double &ikn) {
  const double stats = t[NLOO_STATS * stride];
  const double x = xf, mean = __int_as_float(__double2loint(stats)), ctv = __int_as_float(__double2hiint(stats));
  const double mu = hp[0], kappa = hp[1], sigmasq = hp[2], nu = hp[3];
  const double inv_n = t[NLOO_INV_N * stride];
  ikn = t[NLOO_IKN * stride];
  const double m2 = (t[NLOO_TOTAL * stride] - x) * inv_n;
  // (a sum of squares: below zero it is the rounding of the float fields it is rebuilt from -- an outlier 1e6 away
  // leaves count_times_variance ~1e12 with an ulp of 1e5 -- and a negative variance would turn the score into NaN)
  const double v2 = (inv_n > 0.0 && inv_n < 1.0) ? fmax(ctv - (x - mean) * (x - m2), 0.0) : 0.0;     // n > 1
  const double n_ikn = 1.0 - kappa * ikn;                  // n / kappa_n
  const double mun = (kappa * mu) * ikn + m2 * n_ikn;
  const double d = mu - m2;
  sig = (nu * sigmasq + v2 + (kappa * n_ikn) * d * d) * t[NLOO_INV_NUN * stride];
  const double dd = x - mun;
  dd2 = dd * dd * t[NLOO_INV_NUN * stride];               // (times kfac: the Student-t's (x - mu_n)^2 lambda / nu_n)
}
MSC_DEV double nich_loo_tab(const float *hp, const double *t, size_t stride, float xf) {
  double sig, dd2, ikn;
  nich_loo_core(hp, t, stride, xf, sig, dd2, ikn);
  const double c01 = t[NLOO_C01 * stride];
  const double c0 = __int_as_float(__double2loint(c01)), c1 = __int_as_float(__double2hiint(c01));
  return c0 - 0.5 * log(sig) - c1 * log1p(dd2 / ((1.0 + ikn) * sig));
}
// The sweep kernels' and the leave-one-out pass's form: the downdate and the posterior -- where the cancellations are --
// in double as above, the two logarithms and the divisions in float (log1p_acc: hardware log2 + the compensation term),
// which is how every other entry of the row is evaluated.
MSC_DEV float nich_loo_tab_sweep(const float *hp, const double *t, size_t stride, float xf) {
  double sig, dd2, ikn;
  nich_loo_core(hp, t, stride, xf, sig, dd2, ikn);
  const double c01 = t[NLOO_C01 * stride];
  const float c0 = __int_as_float(__double2loint(c01)), c1 = __int_as_float(__double2hiint(c01));
  const float sigf = (float)sig, q = (float)dd2 * hw_rcp(1.0f + (float)ikn);      // kfac = 1 / (1 + 1/kappa_n)
  const float tt = q * hw_rcp(sigf);                     // (v_rcp_f32: 1 ulp, far inside what log1p of it keeps)
  return c0 - 0.5f * (hw_log2(sigf) * kLn2f) - c1 * log1p_acc(tt);
}
// gp leave-one-out: posterior (a - v, b - 1) of the group's own (a, b); a' + v = a, so
//   score = [lgamma(a) - lgamma(a - v)] - log v! + (a - v) ln b' - a ln(1 + b')
MSC_DEV double gp_loo(const float *hp, uint32_t count, uint32_t sum, uint32_t v) {
  const double a = (double)hp[0] + (double)sum, b1 = (double)hp[1] + (double)count - 1.0;
  return lgamma_drop(a, v) - log_factorial(v) + (a - (double)v) * log(b1) - a * log1p(b1);
}
MSC_DEV double nich_score_data(const float *hp, uint32_t count, float mean, float ctv) {
  const NichPost p = nich_posterior(hp, (double)count, (double)mean, (double)ctv);
  const double kappa = hp[1], sigmasq = hp[2], nu = hp[3];
  return lgamma(0.5 * p.nu) - lgamma(0.5 * nu) + 0.5 * log(kappa / p.kappa) +
         0.5 * nu * log(nu * sigmasq) - 0.5 * p.nu * log(p.nu * p.sigmasq) -
         0.5 * (double)count * kLogPi;
}

// ============================ typed column loads ============================
// runtime_cast::cast<T>(px, t) (runtime_type.hpp:145-166): load as the stored C
// type, convert with the implicit C++ conversion.
template <typename T>
MSC_DEV T load_as(const void *base, uint64_t idx, int type) {
  switch (type) {
    case MSC_TYPE_B: return (T)(((const uint8_t *)base)[idx] != 0);
    case MSC_TYPE_I8: return (T)((const int8_t *)base)[idx];
    case MSC_TYPE_U8: return (T)((const uint8_t *)base)[idx];
    case MSC_TYPE_I16: return (T)((const int16_t *)base)[idx];
    case MSC_TYPE_U16: return (T)((const uint16_t *)base)[idx];
    case MSC_TYPE_I32: return (T)((const int32_t *)base)[idx];
    case MSC_TYPE_U32: return (T)((const uint32_t *)base)[idx];
    case MSC_TYPE_I64: return (T)((const long long *)base)[idx];
    case MSC_TYPE_U64: return (T)((const unsigned long long *)base)[idx];
    case MSC_TYPE_F32: return (T)((const float *)base)[idx];
    case MSC_TYPE_F64: return (T)((const double *)base)[idx];
    default: return T();
  }
}

}  // namespace msc
