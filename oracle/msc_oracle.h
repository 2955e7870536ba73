/*
 * msc_oracle.h -- CPU restatement of the component-model scoring hot path of
 * datamicroscopes/common.  TEST INFRASTRUCTURE ONLY: nothing under oracle/ is
 * linked, imported or executed by the product (common_amd/, include/); only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * PARITY UNPINNED against the real reference: the arithmetic of the five
 * families lives in the third-party `distributions` C++ library (>=2.0.23,
 * conda/microscopes-common/meta.yaml:14,21), which is absent from
 * /root/reference and from this image, and the reference's own tests pin no
 * score.  The restatement follows the published
 * conjugate formulae at the reference's call sites
 * (include/microscopes/models/distributions.hpp:266-291 forwards to
 * T::Group::{add_value,remove_value,score_value,score_data}; field names at
 * :21-56,:79-91,:163-200) and is pinned instead against scipy closed forms
 * (tests/golden/make_golden.py) and against the data the reference's own
 * layout / bookkeeping tests hold (test/test_dataview.py:31-75,
 * test/cxx/test_group_manager.cpp:22-66).
 *
 * Every entry point exists twice: orc_f32_* computes in float exactly as the
 * reference lays its state out (float suff-stats, float hypers, libm logf /
 * lgammaf in place of the unavailable fast_log / fast_lgamma), and orc_f64_*
 * is the double "twin" that is the yardstick for the 1e-6 tolerance.
 */
#ifndef MSC_ORACLE_H
#define MSC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* family tags (same numbering as include/microscopes_hip.h) */
enum { ORC_BB = 0, ORC_GP = 1, ORC_DD = 2, ORC_NICH = 3, ORC_NIW = 4, ORC_NOOP = 5, ORC_BBNC = 6,
       ORC_BNB = 7, ORC_DM = 8, ORC_NFAMILIES = 9 };

/* primitive types, include/microscopes/common/type_info.h:10-44 */
enum { ORC_TYPE_B = 0, ORC_TYPE_I8, ORC_TYPE_U8, ORC_TYPE_I16, ORC_TYPE_U16,
       ORC_TYPE_I32, ORC_TYPE_U32, ORC_TYPE_I64, ORC_TYPE_U64, ORC_TYPE_F32,
       ORC_TYPE_F64, ORC_TYPE_NELEMS };

/*
 * State layouts (AoS, one record per group).  `R` is float for orc_f32_*,
 * double for orc_f64_*.  Hypers are always float (the reference's Shared
 * structs are float); the twin widens them.
 *
 *   bb   hp {alpha, beta}                ss {u32 heads, u32 tails}
 *   gp   hp {alpha, inv_beta}            ss {u32 count, u32 sum, R log_prod}   (f64: 4 B pad before log_prod -> 16 B)
 *   dd   hp {alphas[dim]}                ss {u32 count_sum, u32 counts[dim]}
 *   nich hp {mu, kappa, sigmasq, nu}     ss {u32 count, (f64: u32 pad), R mean, R count_times_variance}
 *   niw  hp {kappa, nu, mu[d], psi[d*d]} ss {u32 count, (f64: u32 pad), R sum_x[d], R sum_xxT[d*d]}
 *   noop hp {}                           ss {u32 unused}
 *   bbnc hp {alpha, beta}                ss {u32 heads, u32 tails, R p}   (src/models/bbnc.cpp:22-73; f64: 16 B)
 *   bnb  hp {alpha, beta, r}             ss {u32 count, u32 sum}          (distributions.hpp:29-36; r is integral)
 *   dm   hp {alphas[dim]}                ss {u32 counts[dim], (pad), R ratio}   (include/microscopes/models/dm.hpp:86-88)
 *
 * Values: bb uint8 (bool), gp/bnb uint32, dd int32, nich float, niw float[d], dm int32[dim].
 */
size_t orc_f32_ss_size(int family, unsigned dim);
size_t orc_f64_ss_size(int family, unsigned dim);
size_t orc_hp_size(int family, unsigned dim);      /* bytes of the float hp block */
size_t orc_value_size(int family, unsigned dim);

#define ORC_DECLARE(P, R)                                                                          \
  void P##_init(int family, unsigned dim, const float *hp, void *ss);                              \
  void P##_add_value(int family, unsigned dim, const float *hp, void *ss, const void *value);      \
  void P##_remove_value(int family, unsigned dim, const float *hp, void *ss, const void *value);   \
  R P##_score_value(int family, unsigned dim, const float *hp, const void *ss, const void *value); \
  R P##_score_data(int family, unsigned dim, const float *hp, const void *ss);                     \
  /* out[n*K + k] = score_value(group k, row n) */                                                 \
  void P##_score_matrix(int family, unsigned dim, const float *hp, const void *ss, size_t K,       \
                        const void *values, size_t N, R *out);                                     \
  /* same, but row n is first removed from its own group z[n] (z[n] < 0: not assigned) */          \
  void P##_score_matrix_loo(int family, unsigned dim, const float *hp, const void *ss, size_t K,   \
                            const void *values, const int32_t *z, size_t N, R *out);               \
  /* ss := init; then add_value(row n -> group z[n]) for n = 0..N-1 in order */                    \
  void P##_accumulate(int family, unsigned dim, const float *hp, void *ss, size_t K,               \
                      const void *values, const int32_t *z, size_t N);                             \
  void P##_score_data_all(int family, unsigned dim, const float *hp, const void *ss, size_t K,     \
                          R *out);                                                                 \
  /* util.hpp:125-136 */                                                                           \
  void P##_scores_to_probs(R *scores, size_t K);                                                   \
  /* group_manager.hpp:274-283 */                                                                  \
  R P##_pseudocount(uint64_t count, float alpha, size_t nempty);                                   \
  /* group_manager.hpp:250-272 */                                                                  \
  R P##_score_assignment(const int64_t *assignments, size_t n, float alpha);

ORC_DECLARE(orc_f32, float)
ORC_DECLARE(orc_f64, double)
#undef ORC_DECLARE

/* util.hpp:145-156: inverse-CDF draw given the uniform dart */
size_t orc_sample_discrete(const float *probs, size_t K, float dart);

/* runtime_type.hpp:123-134 -- offsets[i], *rowsize, *maskrowsize for a packed record */
void orc_offsets_and_size(const int32_t *prim_types, const uint32_t *counts, size_t ntypes,
                          size_t *offsets, size_t *rowsize, size_t *maskrowsize);
size_t orc_primitive_size(int prim_type);

/*
 * runtime_cast::cast / copy (runtime_type.hpp:145-211): read element `elem` of
 * feature at `offset` of every packed record and convert it with the implicit
 * C++ conversion src_type -> dst_type.  out has N elements of dst_type.
 */
void orc_unpack_column(const uint8_t *records, size_t rowsize, size_t offset, size_t elem,
                       int src_type, int dst_type, size_t N, void *out);

/*
 * Counter-based uniform in [0,1) shared by the oracle sweep and the HIP sweep
 * kernel: Philox-4x32-10 keyed on (seed), counter (row, sweep, 0, 0), first
 * output word, mapped with (w >> 8) * 2^-24.
 */
float orc_uniform01(uint64_t seed, uint64_t sweep, uint64_t row);
void orc_philox4x32_10(const uint32_t key[2], const uint32_t ctr[4], uint32_t out[4]);

/*
 * One synchronous ("stale statistics") Gibbs assignment sweep over fixed K
 * groups (SURVEY.md section 3.2 restated as a data-parallel schedule): every row
 * is scored against the suff-stats as they stood at the start of the sweep,
 * minus the row itself, plus log(pseudocount) (group_manager.hpp:274-283), and
 * re-drawn with util.hpp:138-156 using orc_uniform01(seed, sweep, row).
 * nfeat features; feature f has family fam[f], dim[f], hp[f], ss[f] (K records),
 * column values[f].  cnt[k] = rows currently assigned to k.  Writes z_out[N]
 * and, if scores_out != NULL, the [N*K] matrix of summed scores.
 */
void orc_f64_sweep(size_t nfeat, const int *fam, const unsigned *dim, const float *const *hp,
                   const void *const *ss, const void *const *values, size_t K, float alpha,
                   const int32_t *z_in, size_t N, uint64_t seed, uint64_t sweep,
                   int32_t *z_out, double *scores_out);
void orc_f32_sweep(size_t nfeat, const int *fam, const unsigned *dim, const float *const *hp,
                   const void *const *ss, const void *const *values, size_t K, float alpha,
                   const int32_t *z_in, size_t N, uint64_t seed, uint64_t sweep,
                   int32_t *z_out, float *scores_out);

#ifdef __cplusplus
}
#endif
#endif
