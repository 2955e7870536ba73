// hip_models.hpp -- the concrete component models of the plugin surface, backed by the
// MI355X library.  Class names, template shape and member names follow what downstream
// code uses from include/microscopes/models/distributions.hpp (distributions_model<T>,
// distributions_hypers<T>, distributions_group<T>, the public `repr_` members, the
// get_hp_mutator / get_ss_mutator keys of :21-56,:163-200), but nothing here evaluates a
// likelihood on the host: every add / remove / score call is msc_value_op_single, i.e. a
// batch of one on the device (latency-bound; see microscopes_hip.h for the batched calls).
//
// The `distributions::` namespace below only supplies the *tag types* that select a
// family (the real library is absent): their nested Shared / Group structs
// are plain field holders laid out as the C ABI's hp blocks and suff-stat records.
#pragma once

#include <cstdlib>
#include <mutex>

#include "../microscopes_hip.h"
#include "plugin.hpp"
#include "wire.hpp"

namespace distributions {

typedef std::default_random_engine rng_t;

struct BetaBernoulli {
  typedef bool Value;
  struct Shared { float alpha = 1.f, beta = 1.f; };
  struct Group {
    uint32_t heads = 0, tails = 0;
    void init(const Shared &, rng_t &) { heads = tails = 0; }
  };
};

struct GammaPoisson {
  typedef uint32_t Value;
  struct Shared { float alpha = 1.f, inv_beta = 1.f; };
  struct Group {
    uint32_t count = 0, sum = 0;
    float log_prod = 0.f;
    void init(const Shared &, rng_t &) { count = sum = 0; log_prod = 0.f; }
  };
};

struct NormalInverseChiSq {
  typedef float Value;
  struct Shared { float mu = 0.f, kappa = 1.f, sigmasq = 1.f, nu = 1.f; };
  struct Group {
    uint32_t count = 0;
    float mean = 0.f, count_times_variance = 0.f;
    void init(const Shared &, rng_t &) { count = 0; mean = count_times_variance = 0.f; }
  };
};

struct DirichletDiscrete128 {
  typedef int Value;
  enum { max_dim = 128 };
  struct Shared {
    int dim = 0;
    float alphas[max_dim];
    Shared() { for (float &a : alphas) a = 1.f; }
  };
  struct Group {
    int dim = 0;
    uint32_t count_sum = 0;          // count_sum and counts are contiguous: the ABI record
    uint32_t counts[max_dim];
    Group() { for (uint32_t &c : counts) c = 0; }
    void init(const Shared &s, rng_t &) {
      dim = s.dim;
      count_sum = 0;
      for (uint32_t &c : counts) c = 0;
    }
  };
};

struct NormalInverseWishartV {
  struct Shared {
    float kappa = 1.f, nu = 0.f;
    std::vector<float> mu, psi;      // dim, dim*dim (row-major)
    unsigned dim() const { return unsigned(mu.size()); }
    void set_default(unsigned d) {   // microscopes/models.pyx:264-269
      kappa = 1.f;
      nu = float(d);
      mu.assign(d, 0.f);
      psi.assign(std::size_t(d) * d, 0.f);
      for (unsigned i = 0; i < d; i++) psi[std::size_t(i) * d + i] = 1.f;
    }
  };
  struct Group {
    uint32_t count = 0;
    std::vector<float> sum_x, sum_xxT;
    void init(const Shared &s, rng_t &) {
      count = 0;
      sum_x.assign(s.dim(), 0.f);
      sum_xxT.assign(std::size_t(s.dim()) * s.dim(), 0.f);
    }
  };
};

// tag for the in-tree non-conjugate Beta-Bernoulli (src/models/bbnc.cpp): p lives in the group
struct BetaBernoulliNonConj {
  typedef bool Value;
  struct Shared { float alpha = 0.f, beta = 0.f; };
  struct Group {
    uint32_t heads = 0, tails = 0;   // {heads, tails, p}: the ABI record
    float p = 0.5f;
    void init(const Shared &s, rng_t &rng) {
      heads = tails = 0;
      // p ~ Beta(alpha, beta) as the ratio of two gamma draws (bbnc.cpp:129-133 uses sample_beta)
      const float x = std::gamma_distribution<float>(s.alpha > 0 ? s.alpha : 1.f, 1.f)(rng);
      const float y = std::gamma_distribution<float>(s.beta > 0 ? s.beta : 1.f, 1.f)(rng);
      p = x / (x + y);
    }
  };
};

// distributions.hpp:29-36: Shared {alpha, beta, r}, Group {count, sum}
struct BetaNegativeBinomial {
  typedef uint32_t Value;
  struct Shared {
    float alpha = 1.f, beta = 1.f;
    uint32_t r = 1;                  // models.pyx:200
  };
  struct Group {
    uint32_t count = 0, sum = 0;     // {count, sum}: the ABI record
    void init(const Shared &, rng_t &) { count = sum = 0; }
  };
};

// tag for the in-tree Dirichlet-Multinomial (include/microscopes/models/dm.hpp:19-170)
struct DirichletMultinomial {
  struct Shared {
    std::vector<float> alphas;       // dm.hpp:168
    unsigned categories() const { return unsigned(alphas.size()); }
  };
  struct Group {
    std::vector<uint32_t> counts;    // dm.hpp:86-88
    float ratio = 0.f;
    void init(const Shared &s, rng_t &) {
      counts.assign(s.categories(), 0u);
      ratio = 0.f;
    }
  };
};

}  // namespace distributions

namespace microscopes {
namespace hip {

// process-wide context on device $MICROSCOPES_HIP_DEVICE (default 0), null stream
inline msc_context *default_context() {
  static msc_context *ctx = nullptr;
  static std::once_flag once;
  static std::string err;
  std::call_once(once, [] {
    const char *dev = std::getenv("MICROSCOPES_HIP_DEVICE");
    if (msc_context_create(dev ? std::atoi(dev) : 0, nullptr, &ctx) != MSC_OK) {
      err = msc_last_error();
      ctx = nullptr;
    }
  });
  if (!ctx) throw std::runtime_error("microscopes HIP backend unavailable: " + err);
  return ctx;
}

inline void check(int status) {
  if (status != MSC_OK) throw std::runtime_error(msc_last_error());
}

}  // namespace hip

namespace models {
namespace detail {

using distributions::BetaBernoulli;
using distributions::DirichletDiscrete128;
using distributions::GammaPoisson;
using distributions::NormalInverseChiSq;
using distributions::NormalInverseWishartV;
typedef DirichletDiscrete128 DD128;

// family_traits<T>: how T's Shared / Group map onto the C ABI
template <typename T> struct family_traits;

template <> struct family_traits<BetaBernoulli> {
  enum { family = MSC_BB };
  static unsigned dim(const BetaBernoulli::Shared &) { return 0; }
  static common::runtime_type value_type(unsigned) { return common::runtime_type(TYPE_B); }
  static void pack_hp(const BetaBernoulli::Shared &s, std::vector<float> &o) { o = {s.alpha, s.beta}; }
  static void *record(BetaBernoulli::Group &g, std::vector<uint8_t> &) { return &g.heads; }
  static void unpack(BetaBernoulli::Group &, const std::vector<uint8_t> &) {}
  static void pack_value(const common::value_accessor &v, std::vector<uint8_t> &o) { o.assign(1, uint8_t(v.get<bool>(0))); }
};
template <> struct family_traits<distributions::BetaBernoulliNonConj> {
  typedef distributions::BetaBernoulliNonConj T;
  enum { family = MSC_BBNC };
  static unsigned dim(const T::Shared &) { return 0; }
  static common::runtime_type value_type(unsigned) { return common::runtime_type(TYPE_B); }
  static void pack_hp(const T::Shared &s, std::vector<float> &o) { o = {s.alpha, s.beta}; }
  static void *record(T::Group &g, std::vector<uint8_t> &) { return &g.heads; }
  static void unpack(T::Group &, const std::vector<uint8_t> &) {}
  static void pack_value(const common::value_accessor &v, std::vector<uint8_t> &o) { o.assign(1, uint8_t(v.get<bool>(0))); }
};
template <> struct family_traits<GammaPoisson> {
  enum { family = MSC_GP };
  static unsigned dim(const GammaPoisson::Shared &) { return 0; }
  static common::runtime_type value_type(unsigned) { return common::runtime_type(TYPE_U32); }
  static void pack_hp(const GammaPoisson::Shared &s, std::vector<float> &o) { o = {s.alpha, s.inv_beta}; }
  static void *record(GammaPoisson::Group &g, std::vector<uint8_t> &) { return &g.count; }
  static void unpack(GammaPoisson::Group &, const std::vector<uint8_t> &) {}
  static void pack_value(const common::value_accessor &v, std::vector<uint8_t> &o) {
    const uint32_t x = v.get<uint32_t>(0);
    o.resize(4);
    std::memcpy(o.data(), &x, 4);
  }
};
template <> struct family_traits<distributions::BetaNegativeBinomial> {
  typedef distributions::BetaNegativeBinomial T;
  enum { family = MSC_BNB };
  static unsigned dim(const T::Shared &) { return 0; }
  static common::runtime_type value_type(unsigned) { return common::runtime_type(TYPE_U32); }
  static void pack_hp(const T::Shared &s, std::vector<float> &o) { o = {s.alpha, s.beta, float(s.r)}; }
  static void *record(T::Group &g, std::vector<uint8_t> &) { return &g.count; }
  static void unpack(T::Group &, const std::vector<uint8_t> &) {}
  static void pack_value(const common::value_accessor &v, std::vector<uint8_t> &o) {
    const uint32_t x = v.get<uint32_t>(0);
    o.resize(4);
    std::memcpy(o.data(), &x, 4);
  }
};
template <> struct family_traits<distributions::DirichletMultinomial> {
  typedef distributions::DirichletMultinomial T;
  enum { family = MSC_DM };
  static unsigned dim(const T::Shared &s) { return s.categories(); }
  static common::runtime_type value_type(unsigned d) { return common::runtime_type(TYPE_I32, d); }
  static void pack_hp(const T::Shared &s, std::vector<float> &o) { o = s.alphas; }
  static void *record(T::Group &g, std::vector<uint8_t> &buf) {          // {u32 counts[d], f32 ratio}
    buf.resize(4 * (g.counts.size() + 1));
    std::memcpy(buf.data(), g.counts.data(), 4 * g.counts.size());
    std::memcpy(buf.data() + 4 * g.counts.size(), &g.ratio, 4);
    return buf.data();
  }
  static void unpack(T::Group &g, const std::vector<uint8_t> &buf) {
    std::memcpy(g.counts.data(), buf.data(), 4 * g.counts.size());
    std::memcpy(&g.ratio, buf.data() + 4 * g.counts.size(), 4);
  }
  static void pack_value(const common::value_accessor &v, std::vector<uint8_t> &o) {
    o.resize(4 * v.shape());
    for (unsigned i = 0; i < v.shape(); i++) {
      const uint32_t x = v.get<unsigned>(i);                               // dm.cpp:16 reads the counts as unsigned
      std::memcpy(o.data() + 4 * i, &x, 4);
    }
  }
};
template <> struct family_traits<NormalInverseChiSq> {
  enum { family = MSC_NICH };
  static unsigned dim(const NormalInverseChiSq::Shared &) { return 0; }
  static common::runtime_type value_type(unsigned) { return common::runtime_type(TYPE_F32); }
  static void pack_hp(const NormalInverseChiSq::Shared &s, std::vector<float> &o) { o = {s.mu, s.kappa, s.sigmasq, s.nu}; }
  static void *record(NormalInverseChiSq::Group &g, std::vector<uint8_t> &) { return &g.count; }
  static void unpack(NormalInverseChiSq::Group &, const std::vector<uint8_t> &) {}
  static void pack_value(const common::value_accessor &v, std::vector<uint8_t> &o) {
    const float x = v.get<float>(0);
    o.resize(4);
    std::memcpy(o.data(), &x, 4);
  }
};
template <> struct family_traits<DD128> {
  enum { family = MSC_DD };
  static unsigned dim(const DD128::Shared &s) { return unsigned(s.dim); }
  static common::runtime_type value_type(unsigned) { return common::runtime_type(TYPE_I32); }
  static void pack_hp(const DD128::Shared &s, std::vector<float> &o) { o.assign(s.alphas, s.alphas + s.dim); }
  static void *record(DD128::Group &g, std::vector<uint8_t> &) { return &g.count_sum; }
  static void unpack(DD128::Group &, const std::vector<uint8_t> &) {}
  static void pack_value(const common::value_accessor &v, std::vector<uint8_t> &o) {
    const int32_t x = v.get<int32_t>(0);
    o.resize(4);
    std::memcpy(o.data(), &x, 4);
  }
};
template <> struct family_traits<NormalInverseWishartV> {
  enum { family = MSC_NIW };
  static unsigned dim(const NormalInverseWishartV::Shared &s) { return s.dim(); }
  static common::runtime_type value_type(unsigned d) { return common::runtime_type(TYPE_F32, d); }
  static void pack_hp(const NormalInverseWishartV::Shared &s, std::vector<float> &o) {
    o = {s.kappa, s.nu};
    o.insert(o.end(), s.mu.begin(), s.mu.end());
    o.insert(o.end(), s.psi.begin(), s.psi.end());
  }
  static void *record(NormalInverseWishartV::Group &g, std::vector<uint8_t> &buf) {
    buf.resize(4 * (1 + g.sum_x.size() + g.sum_xxT.size()));
    std::memcpy(buf.data(), &g.count, 4);
    std::memcpy(buf.data() + 4, g.sum_x.data(), 4 * g.sum_x.size());
    std::memcpy(buf.data() + 4 + 4 * g.sum_x.size(), g.sum_xxT.data(), 4 * g.sum_xxT.size());
    return buf.data();
  }
  static void unpack(NormalInverseWishartV::Group &g, const std::vector<uint8_t> &buf) {
    std::memcpy(&g.count, buf.data(), 4);
    std::memcpy(g.sum_x.data(), buf.data() + 4, 4 * g.sum_x.size());
    std::memcpy(g.sum_xxT.data(), buf.data() + 4 + 4 * g.sum_x.size(), 4 * g.sum_xxT.size());
  }
  static void pack_value(const common::value_accessor &v, std::vector<uint8_t> &o) {
    o.resize(4 * v.shape());
    for (unsigned i = 0; i < v.shape(); i++) {
      const float x = v.get<float>(i);
      std::memcpy(o.data() + 4 * i, &x, 4);
    }
  }
};

// ---- key -> raw pointer (get_hp_mutator / get_ss_mutator) --------------------------------
template <typename T> struct field_access {
  static common::value_mutator hp(typename T::Shared &, const std::string &) { throw std::runtime_error("not supported"); }
  static common::value_mutator ss(typename T::Group &, const std::string &) { throw std::runtime_error("not supported"); }
};
#define MSC_FIELD(obj, name) if (key == #name) return common::value_mutator(&obj.name)
template <> struct field_access<BetaBernoulli> {
  static common::value_mutator hp(BetaBernoulli::Shared &s, const std::string &key) {
    MSC_FIELD(s, alpha); MSC_FIELD(s, beta);
    throw std::runtime_error("Unknown shared HP param key: " + key);
  }
  static common::value_mutator ss(BetaBernoulli::Group &g, const std::string &key) {
    MSC_FIELD(g, heads); MSC_FIELD(g, tails);
    throw std::runtime_error("Unknown group SS param key: " + key);
  }
};
template <> struct field_access<distributions::BetaBernoulliNonConj> {
  static common::value_mutator hp(distributions::BetaBernoulliNonConj::Shared &s, const std::string &key) {
    MSC_FIELD(s, alpha); MSC_FIELD(s, beta);
    throw std::runtime_error("unknown key: " + key);
  }
  static common::value_mutator ss(distributions::BetaBernoulliNonConj::Group &g, const std::string &key) {
    MSC_FIELD(g, p);                       // the only key the reference exposes (bbnc.cpp:112-117)
    throw std::runtime_error("unknown key: " + key);
  }
};
template <> struct field_access<GammaPoisson> {
  static common::value_mutator hp(GammaPoisson::Shared &s, const std::string &key) {
    MSC_FIELD(s, alpha); MSC_FIELD(s, inv_beta);
    throw std::runtime_error("Unknown shared HP param key: " + key);
  }
  static common::value_mutator ss(GammaPoisson::Group &g, const std::string &key) {
    MSC_FIELD(g, count); MSC_FIELD(g, sum); MSC_FIELD(g, log_prod);
    throw std::runtime_error("Unknown group SS param key: " + key);
  }
};
template <> struct field_access<NormalInverseChiSq> {
  static common::value_mutator hp(NormalInverseChiSq::Shared &s, const std::string &key) {
    MSC_FIELD(s, mu); MSC_FIELD(s, kappa); MSC_FIELD(s, sigmasq); MSC_FIELD(s, nu);
    throw std::runtime_error("Unknown shared HP param key: " + key);
  }
  static common::value_mutator ss(NormalInverseChiSq::Group &g, const std::string &key) {
    MSC_FIELD(g, count); MSC_FIELD(g, mean); MSC_FIELD(g, count_times_variance);
    throw std::runtime_error("Unknown group SS param key: " + key);
  }
};
template <> struct field_access<distributions::BetaNegativeBinomial> {
  typedef distributions::BetaNegativeBinomial T;
  static common::value_mutator hp(T::Shared &s, const std::string &key) {
    MSC_FIELD(s, alpha); MSC_FIELD(s, beta); MSC_FIELD(s, r);
    throw std::runtime_error("Unknown shared HP param key: " + key);
  }
  static common::value_mutator ss(T::Group &g, const std::string &key) {
    MSC_FIELD(g, count); MSC_FIELD(g, sum);
    throw std::runtime_error("Unknown group SS param key: " + key);
  }
};
#undef MSC_FIELD
template <> struct field_access<distributions::DirichletMultinomial> {
  typedef distributions::DirichletMultinomial T;
  static common::value_mutator hp(T::Shared &s, const std::string &key) {    // dm.hpp:137-147
    if (key == "alphas")
      return common::value_mutator(reinterpret_cast<uint8_t *>(s.alphas.data()), common::runtime_type(TYPE_F32, s.categories()));
    throw std::runtime_error("unknown key: " + key);
  }
  static common::value_mutator ss(T::Group &, const std::string &) { throw std::runtime_error("no mutation allowed"); }  // dm.hpp:68-72
};
template <> struct field_access<DD128> {
  static common::value_mutator hp(DD128::Shared &s, const std::string &key) {
    if (key == "alphas")
      return common::value_mutator(reinterpret_cast<uint8_t *>(s.alphas), common::runtime_type(TYPE_F32, unsigned(s.dim)));
    throw std::runtime_error("Unknown shared HP param key: " + key);
  }
  static common::value_mutator ss(DD128::Group &g, const std::string &key) {
    if (key == "count_sum") return common::value_mutator(&g.count_sum);
    if (key == "counts")
      return common::value_mutator(reinterpret_cast<uint8_t *>(g.counts), common::runtime_type(TYPE_U32, unsigned(g.dim)));
    throw std::runtime_error("Unknown group SS param key: " + key);
  }
};

// ---- bags: proto2 messages of the `distributions` schema (see wire.hpp) --------------------
template <typename T> struct bag;
template <> struct bag<BetaBernoulli> {
  static std::string dump(const BetaBernoulli::Shared &s) { wire::writer w; w.put_float_field(1, s.alpha); w.put_float_field(2, s.beta); return w.str(); }
  static void load(BetaBernoulli::Shared &s, const std::string &b) {
    for (const auto &f : wire::parse(b)) { if (f.number == 1) s.alpha = f.f32; if (f.number == 2) s.beta = f.f32; }
  }
  static std::string dump(const BetaBernoulli::Group &g) { wire::writer w; w.put_varint_field(1, g.heads); w.put_varint_field(2, g.tails); return w.str(); }
  static void load(BetaBernoulli::Group &g, const std::string &b) {
    for (const auto &f : wire::parse(b)) { if (f.number == 1) g.heads = uint32_t(f.varint); if (f.number == 2) g.tails = uint32_t(f.varint); }
  }
};
template <> struct bag<distributions::BetaBernoulliNonConj> {      // microscopes/io/schema.proto:7-19
  typedef distributions::BetaBernoulliNonConj T;
  static std::string dump(const T::Shared &s) { wire::writer w; w.put_float_field(1, s.alpha); w.put_float_field(2, s.beta); return w.str(); }
  static void load(T::Shared &s, const std::string &b) {
    for (const auto &f : wire::parse(b)) { if (f.number == 1) s.alpha = f.f32; if (f.number == 2) s.beta = f.f32; }
  }
  static std::string dump(const T::Group &g) { wire::writer w; w.put_float_field(1, g.p); w.put_varint_field(2, g.heads); w.put_varint_field(3, g.tails); return w.str(); }
  static void load(T::Group &g, const std::string &b) {
    for (const auto &f : wire::parse(b)) { if (f.number == 1) g.p = f.f32; if (f.number == 2) g.heads = uint32_t(f.varint); if (f.number == 3) g.tails = uint32_t(f.varint); }
  }
};
template <> struct bag<distributions::BetaNegativeBinomial> {      // distributions' schema: Shared {alpha, beta, r}, Group {count, sum}
  typedef distributions::BetaNegativeBinomial T;
  static std::string dump(const T::Shared &s) { wire::writer w; w.put_float_field(1, s.alpha); w.put_float_field(2, s.beta); w.put_varint_field(3, s.r); return w.str(); }
  static void load(T::Shared &s, const std::string &b) {
    for (const auto &f : wire::parse(b)) { if (f.number == 1) s.alpha = f.f32; if (f.number == 2) s.beta = f.f32; if (f.number == 3) s.r = uint32_t(f.varint); }
  }
  static std::string dump(const T::Group &g) { wire::writer w; w.put_varint_field(1, g.count); w.put_varint_field(2, g.sum); return w.str(); }
  static void load(T::Group &g, const std::string &b) {
    for (const auto &f : wire::parse(b)) { if (f.number == 1) g.count = uint32_t(f.varint); if (f.number == 2) g.sum = uint32_t(f.varint); }
  }
};
template <> struct bag<distributions::DirichletMultinomial> {      // microscopes/io/schema.proto:21-30 (in tree: pinned)
  typedef distributions::DirichletMultinomial T;
  static std::string dump(const T::Shared &s) { wire::writer w; for (float a : s.alphas) w.put_float_field(1, a); return w.str(); }
  static void load(T::Shared &s, const std::string &b) {                  // dm.hpp:117-130
    std::vector<float> a;
    for (const auto &f : wire::parse(b)) if (f.number == 1) wire::collect_floats(f, a);
    if (a.size() != s.alphas.size()) throw std::runtime_error("# categories mismatch");
    for (float x : a) if (!(x > 0.f)) throw std::runtime_error("alphas need to be positive reals");
    s.alphas = a;
  }
  static std::string dump(const T::Group &g) { wire::writer w; for (uint32_t c : g.counts) w.put_varint_field(1, c); w.put_float_field(2, g.ratio); return w.str(); }
  static void load(T::Group &g, const std::string &b) {                   // dm.hpp:43-55
    std::vector<uint64_t> c;
    float ratio = 0.f;
    for (const auto &f : wire::parse(b)) { if (f.number == 1) wire::collect_varints(f, c); if (f.number == 2) ratio = f.f32; }
    if (c.size() != g.counts.size()) throw std::runtime_error("# categories mismatch");
    if (ratio < 0.f) throw std::runtime_error("negative partition");
    for (std::size_t i = 0; i < c.size(); i++) g.counts[i] = uint32_t(c[i]);
    g.ratio = ratio;
  }
};
template <> struct bag<GammaPoisson> {
  static std::string dump(const GammaPoisson::Shared &s) { wire::writer w; w.put_float_field(1, s.alpha); w.put_float_field(2, s.inv_beta); return w.str(); }
  static void load(GammaPoisson::Shared &s, const std::string &b) {
    for (const auto &f : wire::parse(b)) { if (f.number == 1) s.alpha = f.f32; if (f.number == 2) s.inv_beta = f.f32; }
  }
  static std::string dump(const GammaPoisson::Group &g) { wire::writer w; w.put_varint_field(1, g.count); w.put_varint_field(2, g.sum); w.put_float_field(3, g.log_prod); return w.str(); }
  static void load(GammaPoisson::Group &g, const std::string &b) {
    for (const auto &f : wire::parse(b)) { if (f.number == 1) g.count = uint32_t(f.varint); if (f.number == 2) g.sum = uint32_t(f.varint); if (f.number == 3) g.log_prod = f.f32; }
  }
};
template <> struct bag<NormalInverseChiSq> {
  static std::string dump(const NormalInverseChiSq::Shared &s) { wire::writer w; w.put_float_field(1, s.mu); w.put_float_field(2, s.kappa); w.put_float_field(3, s.sigmasq); w.put_float_field(4, s.nu); return w.str(); }
  static void load(NormalInverseChiSq::Shared &s, const std::string &b) {
    for (const auto &f : wire::parse(b)) { if (f.number == 1) s.mu = f.f32; if (f.number == 2) s.kappa = f.f32; if (f.number == 3) s.sigmasq = f.f32; if (f.number == 4) s.nu = f.f32; }
  }
  static std::string dump(const NormalInverseChiSq::Group &g) { wire::writer w; w.put_varint_field(1, g.count); w.put_float_field(2, g.mean); w.put_float_field(3, g.count_times_variance); return w.str(); }
  static void load(NormalInverseChiSq::Group &g, const std::string &b) {
    for (const auto &f : wire::parse(b)) { if (f.number == 1) g.count = uint32_t(f.varint); if (f.number == 2) g.mean = f.f32; if (f.number == 3) g.count_times_variance = f.f32; }
  }
};
template <> struct bag<DD128> {
  static std::string dump(const DD128::Shared &s) { wire::writer w; for (int i = 0; i < s.dim; i++) w.put_float_field(1, s.alphas[i]); return w.str(); }
  static void load(DD128::Shared &s, const std::string &b) {
    std::vector<float> a;
    for (const auto &f : wire::parse(b)) if (f.number == 1) wire::collect_floats(f, a);
    if (int(a.size()) != s.dim) throw std::runtime_error("wrong dimension");
    for (int i = 0; i < s.dim; i++) s.alphas[i] = a[std::size_t(i)];
  }
  static std::string dump(const DD128::Group &g) { wire::writer w; for (int i = 0; i < g.dim; i++) w.put_varint_field(1, g.counts[i]); return w.str(); }
  static void load(DD128::Group &g, const std::string &b) {
    std::vector<uint64_t> c;
    for (const auto &f : wire::parse(b)) if (f.number == 1) wire::collect_varints(f, c);
    if (int(c.size()) != g.dim) throw std::runtime_error("wrong dimension");
    g.count_sum = 0;
    for (int i = 0; i < g.dim; i++) { g.counts[i] = uint32_t(c[std::size_t(i)]); g.count_sum += g.counts[i]; }
  }
};
template <> struct bag<NormalInverseWishartV> {
  static std::string dump(const NormalInverseWishartV::Shared &s) {
    wire::writer w;
    for (float m : s.mu) w.put_float_field(1, m);
    w.put_float_field(2, s.kappa);
    for (float p : s.psi) w.put_float_field(3, p);
    w.put_float_field(4, s.nu);
    return w.str();
  }
  static void load(NormalInverseWishartV::Shared &s, const std::string &b) {
    s.mu.clear(); s.psi.clear();
    for (const auto &f : wire::parse(b)) {
      if (f.number == 1) wire::collect_floats(f, s.mu);
      if (f.number == 2) s.kappa = f.f32;
      if (f.number == 3) wire::collect_floats(f, s.psi);
      if (f.number == 4) s.nu = f.f32;
    }
    if (s.psi.size() != s.mu.size() * s.mu.size()) throw std::runtime_error("wrong dimension");
  }
  static std::string dump(const NormalInverseWishartV::Group &g) {
    wire::writer w;
    w.put_varint_field(1, g.count);
    for (float v : g.sum_x) w.put_float_field(2, v);
    for (float v : g.sum_xxT) w.put_float_field(3, v);
    return w.str();
  }
  static void load(NormalInverseWishartV::Group &g, const std::string &b) {
    g.sum_x.clear(); g.sum_xxT.clear();
    for (const auto &f : wire::parse(b)) {
      if (f.number == 1) g.count = uint32_t(f.varint);
      if (f.number == 2) wire::collect_floats(f, g.sum_x);
      if (f.number == 3) wire::collect_floats(f, g.sum_xxT);
    }
  }
};

// ---- sample_value (base.hpp:29): one draw from the group's posterior predictive ---------------------------------
// Host-side, from the Shared / Group structs alone: it is not on the scoring path and there is no stream of the
// reference's to reproduce (its draws come from the absent library with the caller's std engine), so these are the
// textbook samplers of the predictives that score_value scores (SURVEY 8a formulas), driven by the caller's rng_t.
template <typename T> struct sampler {
  static void draw(const typename T::Shared &, const typename T::Group &, common::value_mutator &, common::rng_t &) {
    throw std::runtime_error("sample_value is not implemented for this model");      // dm: as upstream (dm.cpp:100-111)
  }
};
inline double draw_gamma(double shape, double scale, common::rng_t &rng) { return std::gamma_distribution<double>(shape, scale)(rng); }
inline double draw_beta(double a, double b, common::rng_t &rng) {
  const double x = draw_gamma(a, 1.0, rng), y = draw_gamma(b, 1.0, rng);
  return x / (x + y);
}
template <> struct sampler<BetaBernoulli> {
  static void draw(const BetaBernoulli::Shared &s, const BetaBernoulli::Group &g, common::value_mutator &v, common::rng_t &rng) {
    const double p = (double(s.alpha) + g.heads) / (double(s.alpha) + s.beta + g.heads + g.tails);
    v.set<bool>(std::bernoulli_distribution(p)(rng));
  }
};
template <> struct sampler<distributions::BetaBernoulliNonConj> {
  typedef distributions::BetaBernoulliNonConj T;
  static void draw(const T::Shared &, const T::Group &g, common::value_mutator &v, common::rng_t &rng) {
    v.set<bool>(std::bernoulli_distribution(g.p)(rng));                              // bbnc.cpp:76-82
  }
};
template <> struct sampler<GammaPoisson> {
  static void draw(const GammaPoisson::Shared &s, const GammaPoisson::Group &g, common::value_mutator &v, common::rng_t &rng) {
    const double rate = draw_gamma(double(s.alpha) + g.sum, 1.0 / (double(s.inv_beta) + g.count), rng);
    v.set<uint32_t>(uint32_t(std::poisson_distribution<uint64_t>(rate)(rng)));
  }
};
template <> struct sampler<distributions::BetaNegativeBinomial> {
  typedef distributions::BetaNegativeBinomial T;
  static void draw(const T::Shared &s, const T::Group &g, common::value_mutator &v, common::rng_t &rng) {
    const double p = draw_beta(double(s.alpha) + double(s.r) * g.count, double(s.beta) + g.sum, rng);
    const double rate = draw_gamma(double(s.r), (1.0 - p) / p, rng);                 // failures before the r-th success
    v.set<uint32_t>(uint32_t(std::poisson_distribution<uint64_t>(rate)(rng)));
  }
};
template <> struct sampler<DD128> {
  static void draw(const DD128::Shared &s, const DD128::Group &g, common::value_mutator &v, common::rng_t &rng) {
    double total = 0;
    for (int i = 0; i < s.dim; i++) total += double(s.alphas[i]) + g.counts[i];
    double dart = std::uniform_real_distribution<double>(0.0, total)(rng);
    int pick = s.dim - 1;
    for (int i = 0; i < s.dim; i++)
      if ((dart -= double(s.alphas[i]) + g.counts[i]) <= 0.0) {
        pick = i;
        break;
      }
    v.set<int>(pick);
  }
};
template <> struct sampler<NormalInverseChiSq> {
  static void draw(const NormalInverseChiSq::Shared &s, const NormalInverseChiSq::Group &g, common::value_mutator &v,
                   common::rng_t &rng) {
    const double n = g.count, kn = double(s.kappa) + n, nun = double(s.nu) + n;
    const double mun = (double(s.kappa) * s.mu + n * g.mean) / kn, d = double(s.mu) - g.mean;
    const double sigsq = (double(s.nu) * s.sigmasq + g.count_times_variance + n * s.kappa * d * d / kn) / nun;
    const double t = std::student_t_distribution<double>(nun)(rng);
    v.set<float>(float(mun + std::sqrt(sigsq * (kn + 1.0) / kn) * t));
  }
};
template <> struct sampler<NormalInverseWishartV> {
  static void draw(const NormalInverseWishartV::Shared &s, const NormalInverseWishartV::Group &g, common::value_mutator &v,
                   common::rng_t &rng) {
    const unsigned d = s.dim();
    const double n = g.count, kn = double(s.kappa) + n, nun = double(s.nu) + n, dof = nun - double(d) + 1.0;
    std::vector<double> mun(d), L(std::size_t(d) * d, 0.0), z(d);
    for (unsigned i = 0; i < d; i++) mun[i] = (double(s.kappa) * s.mu[i] + g.sum_x[i]) / kn;
    // Sigma = Psi_n (kn + 1) / (kn dof), Psi_n = Psi + sum xxT + kappa mu muT - kn mun munT; lower Cholesky in place
    const double scale = (kn + 1.0) / (kn * dof);
    for (unsigned i = 0; i < d; i++)
      for (unsigned j = 0; j <= i; j++) {
        double a = (double(s.psi[std::size_t(i) * d + j]) + g.sum_xxT[std::size_t(i) * d + j] +
                    double(s.kappa) * s.mu[i] * s.mu[j] - kn * mun[i] * mun[j]) * scale;
        for (unsigned k = 0; k < j; k++) a -= L[std::size_t(i) * d + k] * L[std::size_t(j) * d + k];
        if (i == j) {
          if (!(a > 0.0)) throw std::runtime_error("niw posterior scale matrix is not positive definite");
          L[std::size_t(i) * d + i] = std::sqrt(a);
        } else {
          L[std::size_t(i) * d + j] = a / L[std::size_t(j) * d + j];
        }
      }
    for (double &zi : z) zi = std::normal_distribution<double>()(rng);
    const double w = std::sqrt(dof / std::chi_squared_distribution<double>(dof)(rng));
    for (unsigned i = 0; i < d; i++) {
      double x = 0;
      for (unsigned k = 0; k <= i; k++) x += L[std::size_t(i) * d + k] * z[k];
      v.set<float>(float(mun[i] + w * x), i);
    }
  }
};

}  // namespace detail

template <typename T> class distributions_hypers;

// ---- group ---------------------------------------------------------------------------------
template <typename T>
class distributions_group : public group {
  typedef detail::family_traits<T> traits;

public:
  void add_value(const hypers &m, const common::value_accessor &value, common::rng_t &) override {
    run(MSC_OP_ADD, m, &value);
  }
  void remove_value(const hypers &m, const common::value_accessor &value, common::rng_t &) override {
    run(MSC_OP_REMOVE, m, &value);
  }
  float score_value(const hypers &m, const common::value_accessor &value, common::rng_t &) const override {
    return const_cast<distributions_group *>(this)->run(MSC_OP_SCORE_VALUE, m, &value);
  }
  float score_data(const hypers &m, common::rng_t &) const override {
    return const_cast<distributions_group *>(this)->run(MSC_OP_SCORE_DATA, m, nullptr);
  }
  void sample_value(const hypers &m, common::value_mutator &value, common::rng_t &rng) const override {
    detail::sampler<T>::draw(static_cast<const distributions_hypers<T> &>(m).repr_, repr_, value, rng);
  }
  common::suffstats_bag_t get_ss() const override { return detail::bag<T>::dump(repr_); }
  void set_ss(const common::suffstats_bag_t &ss) override { detail::bag<T>::load(repr_, ss); }
  void set_ss(const group &g) override { repr_ = static_cast<const distributions_group<T> &>(g).repr_; }
  common::value_mutator get_ss_mutator(const std::string &key) override { return detail::field_access<T>::ss(repr_, key); }
  std::string debug_str() const override { return "<group family " + std::to_string(int(traits::family)) + ">"; }
  bool device_record_get(const hypers &m, std::vector<uint8_t> &rec) const override {
    const typename T::Shared &shared = static_cast<const distributions_hypers<T> &>(m).repr_;
    std::vector<uint8_t> buf;
    const void *p = traits::record(const_cast<typename T::Group &>(repr_), buf);
    const uint8_t *b = static_cast<const uint8_t *>(p);
    rec.assign(b, b + msc_ss_bytes(traits::family, traits::dim(shared)));
    return true;
  }
  bool device_record_set(const hypers &m, const std::vector<uint8_t> &rec) override {
    const typename T::Shared &shared = static_cast<const distributions_hypers<T> &>(m).repr_;
    if (rec.size() != msc_ss_bytes(traits::family, traits::dim(shared))) throw std::runtime_error("record size mismatch");
    std::vector<uint8_t> buf;
    void *p = traits::record(repr_, buf);
    if (p == buf.data()) traits::unpack(repr_, rec);          // packed families: straight from the record
    else std::memcpy(p, rec.data(), rec.size());              // plain-struct families: the record is the struct's head
    return true;
  }

  typename T::Group repr_;

private:
  float run(int op, const hypers &m, const common::value_accessor *value) {
    // unchecked downcast, as the reference does (distributions.hpp:523-528)
    const typename T::Shared &shared = static_cast<const distributions_hypers<T> &>(m).repr_;
    std::vector<float> hp;
    std::vector<uint8_t> rec, val;
    traits::pack_hp(shared, hp);
    void *record = traits::record(repr_, rec);
    if (value) traits::pack_value(*value, val);
    float score = 0.f;
    hip::check(msc_value_op_single(hip::default_context(), traits::family, traits::dim(shared), op, hp.data(),
                                   record, value ? val.data() : nullptr, &score));
    if (op <= MSC_OP_REMOVE) traits::unpack(repr_, rec);
    return score;
  }
};

// ---- hypers --------------------------------------------------------------------------------
namespace detail {
template <typename T>
class hypers_base : public hypers {
public:
  std::shared_ptr<group> create_group(common::rng_t &rng) const override {
    auto p = std::make_shared<distributions_group<T>>();
    p->repr_.init(repr_, rng);
    return p;
  }
  common::hyperparam_bag_t get_hp() const override { return bag<T>::dump(repr_); }
  void set_hp(const common::hyperparam_bag_t &hp) override { bag<T>::load(repr_, hp); }
  void set_hp(const hypers &m) override { repr_ = static_cast<const hypers_base<T> &>(m).repr_; }
  common::value_mutator get_hp_mutator(const std::string &key) override { return field_access<T>::hp(repr_, key); }
  std::string debug_str() const override { return "<hypers family " + std::to_string(int(family_traits<T>::family)) + ">"; }
  bool device_spec(msc_feature_spec &spec, std::vector<float> &hp) const override {
    spec.family = family_traits<T>::family;
    spec.dim = family_traits<T>::dim(repr_);
    family_traits<T>::pack_hp(repr_, hp);
    return true;
  }

  typename T::Shared repr_;
};
}  // namespace detail

template <typename T>
class distributions_hypers : public detail::hypers_base<T> {};

template <>
class distributions_hypers<detail::DD128> : public detail::hypers_base<detail::DD128> {
public:
  explicit distributions_hypers(unsigned size) {
    if (size == 0 || size > detail::DD128::max_dim) throw std::runtime_error("dd size outside 1..128");
    this->repr_.dim = int(size);
  }
  void set_hp(const hypers &m) override {
    const auto &that = static_cast<const distributions_hypers<detail::DD128> &>(m);
    if (that.repr_.dim != this->repr_.dim) throw std::runtime_error("wrong dimension");
    this->repr_ = that.repr_;
  }
  void set_hp(const common::hyperparam_bag_t &hp) override { detail::bag<detail::DD128>::load(this->repr_, hp); }
};

template <>
class distributions_hypers<detail::NormalInverseWishartV> : public detail::hypers_base<detail::NormalInverseWishartV> {
public:
  distributions_hypers() {}
  explicit distributions_hypers(unsigned dim) { this->repr_.set_default(dim); }
};

// ---- model ---------------------------------------------------------------------------------
template <typename T>
class distributions_model : public model {
public:
  std::shared_ptr<hypers> create_hypers() const override { return std::make_shared<distributions_hypers<T>>(); }
  common::runtime_type get_runtime_type() const override { return detail::family_traits<T>::value_type(0); }
};

template <>
class distributions_model<detail::DD128> : public model {
public:
  explicit distributions_model(unsigned dim) : dim_(dim) {}
  std::shared_ptr<hypers> create_hypers() const override { return std::make_shared<distributions_hypers<detail::DD128>>(dim_); }
  common::runtime_type get_runtime_type() const override { return common::runtime_type(TYPE_I32); }

private:
  unsigned dim_;
};

template <>
class distributions_model<detail::NormalInverseWishartV> : public model {
public:
  explicit distributions_model(unsigned dim) : dim_(dim) {}
  std::shared_ptr<hypers> create_hypers() const override {
    return std::make_shared<distributions_hypers<detail::NormalInverseWishartV>>(dim_);
  }
  common::runtime_type get_runtime_type() const override { return common::runtime_type(TYPE_F32, dim_); }

private:
  unsigned dim_;
};

// ---- Dirichlet-Multinomial under its reference names (include/microscopes/models/dm.hpp) ------------
template <>
class distributions_hypers<distributions::DirichletMultinomial> : public detail::hypers_base<distributions::DirichletMultinomial> {
public:
  explicit distributions_hypers(unsigned categories) { this->repr_.alphas.assign(categories, 0.f); }   // dm.hpp:92-93
  void set_hp(const hypers &m) override {                                                               // dm.hpp:132-139
    const auto &that = static_cast<const distributions_hypers<distributions::DirichletMultinomial> &>(m);
    if (that.categories() != categories()) throw std::runtime_error("# categories mismatch");
    this->repr_ = that.repr_;
  }
  void set_hp(const common::hyperparam_bag_t &hp) override { detail::bag<distributions::DirichletMultinomial>::load(this->repr_, hp); }
  std::size_t categories() const { return this->repr_.alphas.size(); }
  const std::vector<float> &alphas() const { return this->repr_.alphas; }
};

template <>
class distributions_model<distributions::DirichletMultinomial> : public model {
public:
  explicit distributions_model(unsigned categories) : categories_(categories) {
    if (categories < 2) throw std::runtime_error("need at least two outcomes");                       // dm.hpp:174-178
    if (categories > 128) throw std::runtime_error("dm: more than 128 categories are not built for the HIP backend");
  }
  std::shared_ptr<hypers> create_hypers() const override {
    return std::make_shared<distributions_hypers<distributions::DirichletMultinomial>>(categories_);
  }
  common::runtime_type get_runtime_type() const override { return common::runtime_type(TYPE_I32, categories_); }   // dm.hpp:186-190
  std::size_t categories() const { return categories_; }

private:
  unsigned categories_;
};
typedef distributions_group<distributions::DirichletMultinomial> dm_group;
typedef distributions_hypers<distributions::DirichletMultinomial> dm_hypers;
typedef distributions_model<distributions::DirichletMultinomial> dm_model;

// the in-tree non-conjugate model under its reference names (include/microscopes/models/bbnc.hpp:9-73)
typedef distributions_group<distributions::BetaBernoulliNonConj> bbnc_group;
typedef distributions_hypers<distributions::BetaBernoulliNonConj> bbnc_hypers;
typedef distributions_model<distributions::BetaBernoulliNonConj> bbnc_model;

typedef distributions_model<detail::DD128> distributions_model_dd128;
typedef distributions_model<detail::NormalInverseWishartV> distributions_model_niwv;

}  // namespace models
}  // namespace microscopes
