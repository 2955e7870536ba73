/*
 * msc_oracle.c -- CPU restatement (plain C) of the scoring hot path of
 * datamicroscopes/common.  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED against
 * the real reference (see msc_oracle.h).  Build: oracle/Makefile.
 */
#define _GNU_SOURCE
#include "msc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

size_t orc_hp_size(int family, unsigned dim) {
  switch (family) {
    case ORC_BB: return 2 * sizeof(float);
    case ORC_BBNC: return 2 * sizeof(float);
    case ORC_GP: return 2 * sizeof(float);
    case ORC_DD: return dim * sizeof(float);
    case ORC_NICH: return 4 * sizeof(float);
    case ORC_NIW: return (2u + (size_t)dim + (size_t)dim * dim) * sizeof(float);
    case ORC_BNB: return 3 * sizeof(float);
    case ORC_DM: return dim * sizeof(float);
    default: return 0;
  }
}

size_t orc_value_size(int family, unsigned dim) {
  switch (family) {
    case ORC_BB: return 1;
    case ORC_BBNC: return 1;
    case ORC_GP: return 4;
    case ORC_DD: return 4;
    case ORC_NICH: return 4;
    case ORC_NIW: return 4u * (size_t)dim;
    case ORC_BNB: return 4;
    case ORC_DM: return 4u * (size_t)dim;
    default: return 1;
  }
}

/* util.hpp:145-156 */
size_t orc_sample_discrete(const float *probs, size_t K, float dart) {
  for (size_t i = 0; i < K; i++) {
    dart -= probs[i];
    if (dart <= 0.f) return i;
  }
  return K - 1;
}

/* ---- Philox-4x32-10 (Salmon et al., SC'11), the counter-based RNG ------ */
void orc_philox4x32_10(const uint32_t key[2], const uint32_t ctr[4], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; r++) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

float orc_uniform01(uint64_t seed, uint64_t sweep, uint64_t row) {
  const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  const uint32_t ctr[4] = {(uint32_t)row, (uint32_t)(row >> 32), (uint32_t)sweep,
                           (uint32_t)(sweep >> 32)};
  uint32_t out[4];
  orc_philox4x32_10(key, ctr, out);
  return (float)(out[0] >> 8) * (1.0f / 16777216.0f);
}

/* ---- packed-record layout (runtime_type.hpp:123-134) ------------------- */
static const size_t k_prim_size[ORC_TYPE_NELEMS] = {1, 1, 1, 2, 2, 4, 4, 8, 8, 4, 8};

size_t orc_primitive_size(int t) { return (t >= 0 && t < ORC_TYPE_NELEMS) ? k_prim_size[t] : 0; }

void orc_offsets_and_size(const int32_t *prim_types, const uint32_t *counts, size_t ntypes,
                          size_t *offsets, size_t *rowsize, size_t *maskrowsize) {
  size_t row = 0, mrow = 0;
  for (size_t i = 0; i < ntypes; i++) {
    offsets[i] = row;
    row += orc_primitive_size(prim_types[i]) * counts[i];
    mrow += counts[i];
  }
  *rowsize = row;
  *maskrowsize = mrow;
}

/* runtime_cast::cast<T>(px, t): load as the stored C type, convert implicitly */
#define LOAD_AS(T_out, px, t, dst)                                                  \
  switch (t) {                                                                      \
    case ORC_TYPE_B:   { _Bool v;    memcpy(&v, px, 1); dst = (T_out)v; } break;     \
    case ORC_TYPE_I8:  { int8_t v;   memcpy(&v, px, 1); dst = (T_out)v; } break;     \
    case ORC_TYPE_U8:  { uint8_t v;  memcpy(&v, px, 1); dst = (T_out)v; } break;     \
    case ORC_TYPE_I16: { int16_t v;  memcpy(&v, px, 2); dst = (T_out)v; } break;     \
    case ORC_TYPE_U16: { uint16_t v; memcpy(&v, px, 2); dst = (T_out)v; } break;     \
    case ORC_TYPE_I32: { int32_t v;  memcpy(&v, px, 4); dst = (T_out)v; } break;     \
    case ORC_TYPE_U32: { uint32_t v; memcpy(&v, px, 4); dst = (T_out)v; } break;     \
    case ORC_TYPE_I64: { int64_t v;  memcpy(&v, px, 8); dst = (T_out)v; } break;     \
    case ORC_TYPE_U64: { uint64_t v; memcpy(&v, px, 8); dst = (T_out)v; } break;     \
    case ORC_TYPE_F32: { float v;    memcpy(&v, px, 4); dst = (T_out)v; } break;     \
    case ORC_TYPE_F64: { double v;   memcpy(&v, px, 8); dst = (T_out)v; } break;     \
    default: dst = (T_out)0; break;                                                  \
  }

void orc_unpack_column(const uint8_t *records, size_t rowsize, size_t offset, size_t elem,
                       int src_type, int dst_type, size_t N, void *out) {
  const size_t ssz = orc_primitive_size(src_type);
  for (size_t n = 0; n < N; n++) {
    const uint8_t *px = records + n * rowsize + offset + elem * ssz;
    switch (dst_type) {
      case ORC_TYPE_B:   { _Bool d;    LOAD_AS(_Bool, px, src_type, d);    ((uint8_t *)out)[n] = d; } break;
      case ORC_TYPE_I8:  { int8_t d;   LOAD_AS(int8_t, px, src_type, d);   ((int8_t *)out)[n] = d; } break;
      case ORC_TYPE_U8:  { uint8_t d;  LOAD_AS(uint8_t, px, src_type, d);  ((uint8_t *)out)[n] = d; } break;
      case ORC_TYPE_I16: { int16_t d;  LOAD_AS(int16_t, px, src_type, d);  ((int16_t *)out)[n] = d; } break;
      case ORC_TYPE_U16: { uint16_t d; LOAD_AS(uint16_t, px, src_type, d); ((uint16_t *)out)[n] = d; } break;
      case ORC_TYPE_I32: { int32_t d;  LOAD_AS(int32_t, px, src_type, d);  ((int32_t *)out)[n] = d; } break;
      case ORC_TYPE_U32: { uint32_t d; LOAD_AS(uint32_t, px, src_type, d); ((uint32_t *)out)[n] = d; } break;
      case ORC_TYPE_I64: { int64_t d;  LOAD_AS(int64_t, px, src_type, d);  ((int64_t *)out)[n] = d; } break;
      case ORC_TYPE_U64: { uint64_t d; LOAD_AS(uint64_t, px, src_type, d); ((uint64_t *)out)[n] = d; } break;
      case ORC_TYPE_F32: { float d;    LOAD_AS(float, px, src_type, d);    ((float *)out)[n] = d; } break;
      case ORC_TYPE_F64: { double d;   LOAD_AS(double, px, src_type, d);   ((double *)out)[n] = d; } break;
      default: break;
    }
  }
}

/* ---- the two precisions ------------------------------------------------ */
#define REAL float
#define REAL_IS_DOUBLE 0
#define PFX orc_f32
#include "msc_oracle_impl.inc"
#undef REAL
#undef REAL_IS_DOUBLE
#undef PFX

#define REAL double
#define REAL_IS_DOUBLE 1
#define PFX orc_f64
#include "msc_oracle_impl.inc"
#undef REAL
#undef REAL_IS_DOUBLE
#undef PFX
