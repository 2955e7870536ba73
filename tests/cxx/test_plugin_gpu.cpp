// GPU: the virtual model / hypers / group API (include/microscopes/models) driven exactly
// as downstream code drives the reference's, checked against the oracle's double twin.
// Every add / remove / score here is msc_value_op_single -> one launch on the device.
#include <microscopes/common/recarray/dataview.hpp>
#include <microscopes/models/distributions.hpp>
#include <microscopes/models/dm.hpp>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#include "../../oracle/msc_oracle.h"
#include "audit.hpp"

using namespace distributions;
using namespace microscopes;
using namespace microscopes::common;

#define CHECK(cond)                                                          \
  do {                                                                       \
    if (!(cond)) {                                                           \
      std::fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__); \
      std::exit(1);                                                          \
    }                                                                        \
  } while (0)

static bool close(double got, double want, double tol = 1e-6) {
  const bool ok = std::fabs(got - want) <= tol * std::fmax(1.0, std::fabs(want));
  if (!ok) std::fprintf(stderr, "  got %.9g want %.9g\n", got, want);
  return ok;
}

// The double twin's record built from the FLOAT state the group itself holds (repr_, the reference's own fields,
// distributions.hpp:21-56): what a score of that group is answerable to (audit.hpp).  Layouts: oracle/msc_oracle.h.
template <typename T> static const typename T::Group &repr_of(const models::group &g) {
  return static_cast<const models::distributions_group<T> &>(g).repr_;
}
static std::vector<uint8_t> fed(const BetaBernoulli::Group &r) {
  std::vector<uint8_t> o(8);
  std::memcpy(o.data(), &r.heads, 4); std::memcpy(o.data() + 4, &r.tails, 4);
  return o;
}
static std::vector<uint8_t> fed(const GammaPoisson::Group &r) {
  struct { uint32_t count, sum; double log_prod; } t = {r.count, r.sum, double(r.log_prod)};
  std::vector<uint8_t> o(sizeof t);
  std::memcpy(o.data(), &t, sizeof t);
  return o;
}
static std::vector<uint8_t> fed(const NormalInverseChiSq::Group &r) {
  struct { uint32_t count, pad; double mean, ctv; } t = {r.count, 0u, double(r.mean), double(r.count_times_variance)};
  std::vector<uint8_t> o(sizeof t);
  std::memcpy(o.data(), &t, sizeof t);
  return o;
}
static std::vector<uint8_t> fed(const NormalInverseWishartV::Group &r) {
  std::vector<double> t(1 + r.sum_x.size() + r.sum_xxT.size(), 0.0);
  std::memcpy(t.data(), &r.count, 4);
  for (size_t i = 0; i < r.sum_x.size(); i++) t[1 + i] = double(r.sum_x[i]);
  for (size_t i = 0; i < r.sum_xxT.size(); i++) t[1 + r.sum_x.size() + i] = double(r.sum_xxT[i]);
  std::vector<uint8_t> o(8 * t.size());
  std::memcpy(o.data(), t.data(), o.size());
  return o;
}
static std::vector<uint8_t> fed(const distributions::DirichletMultinomial::Group &r) {
  const size_t off = (4 * r.counts.size() + 7) / 8 * 8;
  std::vector<uint8_t> o(off + 8, 0);
  std::memcpy(o.data(), r.counts.data(), 4 * r.counts.size());
  const double ratio = double(r.ratio);
  std::memcpy(o.data() + off, &ratio, 8);
  return o;
}
// the float fields of the group against the all-double chain `oss` after n per-value updates (n half-ulps each), the
// integer ones bit for bit
static bool state_follows_chain(const char *fam, const GammaPoisson::Group &r, const void *oss, int n) {
  struct T { uint32_t count, sum; double log_prod; } t;
  std::memcpy(&t, oss, sizeof t);
  return r.count == t.count && r.sum == t.sum && audit::field(std::string("plugin.float_field.") + fam + ".log_prod", r.log_prod, t.log_prod, n);
}
static bool state_follows_chain(const char *fam, const NormalInverseChiSq::Group &r, const void *oss, int n) {
  struct T { uint32_t count, pad; double mean, ctv; } t;
  std::memcpy(&t, oss, sizeof t);
  return r.count == t.count && audit::field(std::string("plugin.float_field.") + fam + ".mean", r.mean, t.mean, n) &&
         audit::field(std::string("plugin.float_field.") + fam + ".count_times_variance", r.count_times_variance, t.ctv, n);
}
static bool state_follows_chain(const char *, const BetaBernoulli::Group &r, const void *oss, int) {
  uint32_t t[2];
  std::memcpy(t, oss, 8);
  return r.heads == t[0] && r.tails == t[1];
}

// drive one scalar family through the API and the oracle side by side
template <typename Tag, typename V, typename Gen>
static void run_scalar(int fam, unsigned dim, models::model &m, const std::vector<float> &hp,
                       const std::vector<std::pair<const char *, float>> &hp_keys, Gen gen) {
  rng_t r(73);
  auto h = m.create_hypers();
  for (const auto &kv : hp_keys) h->get_hp_mutator(kv.first).template set<float>(kv.second);
  auto g = h->create_group(r);
  std::vector<uint8_t> oss(orc_f64_ss_size(fam, dim));
  orc_f64_init(fam, dim, hp.data(), oss.data());
  const char *fname = fam == ORC_BB ? "bb" : fam == ORC_GP ? "gp" : "nich";
  // a score against the twin fed the group's own float state (1e-6); the state itself against the all-double chain
  // `oss` (integers exact, float fields at one half-ulp per update so far)
  auto scores_follow = [&](const V &probe, const char *when) {
    const std::vector<uint8_t> own = fed(repr_of<Tag>(*g));
    return audit::score(std::string("plugin.score_value.") + fname + "." + when, g->score_value(*h, value_accessor(&probe), r),
                        orc_f64_score_value(fam, dim, hp.data(), own.data(), &probe)) &&
           audit::score(std::string("plugin.score_data.") + fname + "." + when, g->score_data(*h, r),
                        orc_f64_score_data(fam, dim, hp.data(), own.data()));
  };
  V vals[25];   // (not std::vector: vector<bool> has no addressable elements)
  for (int i = 0; i < 25; i++) {
    vals[i] = gen(r);
    g->add_value(*h, value_accessor(&vals[i]), r);
    orc_f64_add_value(fam, dim, hp.data(), oss.data(), &vals[i]);
    CHECK(state_follows_chain(fname, repr_of<Tag>(*g), oss.data(), i + 1));
    if (i % 6 == 0) CHECK(scores_follow(gen(r), "after_add"));
  }
  // remove half again
  for (int i = 0; i < 12; i++) {
    g->remove_value(*h, value_accessor(&vals[i]), r);
    orc_f64_remove_value(fam, dim, hp.data(), oss.data(), &vals[i]);
    CHECK(state_follows_chain(fname, repr_of<Tag>(*g), oss.data(), 26 + i));
  }
  const V probe = gen(r);
  CHECK(scores_follow(probe, "after_remove"));                // (round 1-3: 2e-6 against the chain twin)
  // bags round-trip, copies keep scoring identically
  auto g2 = h->create_group(r);
  g2->set_ss(g->get_ss());
  CHECK(g2->score_value(*h, value_accessor(&probe), r) == g->score_value(*h, value_accessor(&probe), r));
  auto g3 = h->create_group(r);
  g3->set_ss(*g);
  CHECK(g3->score_data(*h, r) == g->score_data(*h, r));
  auto h2 = m.create_hypers();
  h2->set_hp(h->get_hp());
  CHECK(g->score_value(*h2, value_accessor(&probe), r) == g->score_value(*h, value_accessor(&probe), r));
  auto h3 = m.create_hypers();
  h3->set_hp(*h);
  CHECK(g->score_data(*h3, r) == g->score_data(*h, r));
  // a fresh group scores like the prior
  auto g0 = h->create_group(r);
  std::vector<uint8_t> o0(orc_f64_ss_size(fam, dim));
  orc_f64_init(fam, dim, hp.data(), o0.data());
  CHECK(close(g0->score_value(*h, value_accessor(&probe), r), orc_f64_score_value(fam, dim, hp.data(), o0.data(), &probe)));
  CHECK(std::fabs(g0->score_data(*h, r)) < 1e-6f);   // empty group: log marginal of no data (0 up to fma rounding)
}

static void test_bb_mutators_write_through() {
  rng_t r(1);
  models::distributions_model<BetaBernoulli> m;
  CHECK(m.get_runtime_type() == runtime_type(TYPE_B));
  auto h = m.create_hypers();
  h->get_hp_mutator("alpha").set<float>(2.0f);
  h->get_hp_mutator("beta").set<float>(2.0f);
  auto g = h->create_group(r);
  g->get_ss_mutator("heads").set<uint32_t>(7);      // raw pointer into the live state
  g->get_ss_mutator("tails").set<int>(1);           // any source type converts
  const bool t = true;
  CHECK(close(g->score_value(*h, value_accessor(&t), r), std::log(9.0 / 12.0)));
  CHECK(g->get_ss_mutator("heads").accessor().get<uint32_t>(0) == 7);
  bool threw = false;
  try { g->get_ss_mutator("nope"); } catch (const std::runtime_error &) { threw = true; }
  CHECK(threw);
  threw = false;
  try { h->get_hp_mutator("nope"); } catch (const std::runtime_error &) { threw = true; }
  CHECK(threw);
  // values of another primitive type are cast on the way in (runtime_cast)
  const double as_double = 1.0;
  g->add_value(*h, value_accessor(&as_double), r);
  CHECK(g->get_ss_mutator("heads").accessor().get<uint32_t>(0) == 8);
}

static void test_dd_and_niw() {
  rng_t r(3);
  {
    const unsigned dim = 5;
    models::distributions_model_dd128 m(dim);
    auto h = m.create_hypers();
    auto am = h->get_hp_mutator("alphas");
    CHECK(am.shape() == dim);
    std::vector<float> hp(dim);
    for (unsigned i = 0; i < dim; i++) { hp[i] = 0.5f + 0.25f * float(i); am.set<float>(hp[i], i); }
    auto g = h->create_group(r);
    std::vector<uint8_t> oss(orc_f64_ss_size(ORC_DD, dim));
    orc_f64_init(ORC_DD, dim, hp.data(), oss.data());
    for (int i = 0; i < 40; i++) {
      const int v = std::uniform_int_distribution<int>(0, int(dim) - 1)(r);
      g->add_value(*h, value_accessor(&v), r);
      orc_f64_add_value(ORC_DD, dim, hp.data(), oss.data(), &v);
    }
    for (int v = 0; v < int(dim); v++)
      CHECK(close(g->score_value(*h, value_accessor(&v), r), orc_f64_score_value(ORC_DD, dim, hp.data(), oss.data(), &v)));
    CHECK(close(g->score_data(*h, r), orc_f64_score_data(ORC_DD, dim, hp.data(), oss.data())));
    CHECK(g->get_ss_mutator("count_sum").accessor().get<uint32_t>(0) == 40);
    auto g2 = h->create_group(r);
    g2->set_ss(g->get_ss());
    const int v0 = 2;
    CHECK(g2->score_value(*h, value_accessor(&v0), r) == g->score_value(*h, value_accessor(&v0), r));
  }
  for (const unsigned d : {4u, 40u, 128u}) {     // NormalInverseWishart<-1>: the dimension is a run-time value (distributions.hpp:87-91)
    models::distributions_model_niwv m(d);
    CHECK(m.get_runtime_type() == runtime_type(TYPE_F32, d));
    auto h = m.create_hypers();     // default hp: mu = 0, kappa = 1, psi = I, nu = d
    std::vector<float> hp(2 + d + d * d, 0.f);
    hp[0] = 1.f; hp[1] = float(d);
    for (unsigned i = 0; i < d; i++) hp[2 + d + i * d + i] = 1.f;
    auto g = h->create_group(r);
    std::vector<uint8_t> oss(orc_f64_ss_size(ORC_NIW, d));
    orc_f64_init(ORC_NIW, d, hp.data(), oss.data());
    std::normal_distribution<float> nd(1.f, 2.f);
    const runtime_type vt(TYPE_F32, d);
    std::vector<std::vector<float>> xs;
    for (int i = 0; i < (d > 32 ? 6 : 15); i++) {
      xs.emplace_back(d);
      for (auto &x : xs.back()) x = nd(r);
      g->add_value(*h, value_accessor(reinterpret_cast<const uint8_t *>(xs.back().data()), nullptr, vt), r);
      orc_f64_add_value(ORC_NIW, d, hp.data(), oss.data(), xs.back().data());
    }
    std::vector<float> probe(d);
    for (auto &x : probe) x = nd(r);
    const value_accessor pa(reinterpret_cast<const uint8_t *>(probe.data()), nullptr, vt);
    // (rounds 1-3 held these against the chain twin `oss` at 5e-6 / 2e-5: what they measured was the float STATE --
    // sum_xxT after 15 float updates, then a downdate -- not the evaluation.  The evaluation, on the state the group
    // holds: the plain gate.  The state, against the chain: one half-ulp per update of the field's largest entry.)
    auto niw_checks = [&](const char *when, int nupd) {
      const NormalInverseWishartV::Group &own = repr_of<NormalInverseWishartV>(*g);
      const std::vector<uint8_t> rec = fed(own);
      const std::string tag = std::string("plugin.niw_d") + std::to_string(d) + "." + when;
      bool ok = audit::score(tag + ".score_value", g->score_value(*h, pa, r), orc_f64_score_value(ORC_NIW, d, hp.data(), rec.data(), probe.data())) &&
                audit::score(tag + ".score_data", g->score_data(*h, r), orc_f64_score_data(ORC_NIW, d, hp.data(), rec.data()));
      const double *chain = reinterpret_cast<const double *>(oss.data());
      double big_x = 1.0, big_xx = 1.0;
      for (unsigned i = 0; i < d; i++) big_x = std::fmax(big_x, std::fabs(chain[1 + i]));
      for (unsigned i = 0; i < d * d; i++) big_xx = std::fmax(big_xx, std::fabs(chain[1 + d + i]));
      double ex = 0, exx = 0;
      for (unsigned i = 0; i < d; i++) ex = std::fmax(ex, std::fabs(double(own.sum_x[i]) - chain[1 + i]));
      for (unsigned i = 0; i < d * d; i++) exx = std::fmax(exx, std::fabs(double(own.sum_xxT[i]) - chain[1 + d + i]));
      ok = ok && audit::check(tag + ".sum_x_vs_chain", ex / (std::ldexp(big_x, -24) * nupd), 1.0) &&
           audit::check(tag + ".sum_xxT_vs_chain", exx / (std::ldexp(big_xx, -24) * nupd), 1.0);
      uint32_t cnt;
      std::memcpy(&cnt, oss.data(), 4);
      return ok && own.count == cnt;
    };
    const int nadd = int(xs.size());
    CHECK(niw_checks("after_add", nadd));
    g->remove_value(*h, value_accessor(reinterpret_cast<const uint8_t *>(xs[0].data()), nullptr, vt), r);
    orc_f64_remove_value(ORC_NIW, d, hp.data(), oss.data(), xs[0].data());
    CHECK(niw_checks("after_remove", nadd + 1));
    auto g2 = h->create_group(r);
    g2->set_ss(g->get_ss());
    CHECK(g2->score_value(*h, pa, r) == g->score_value(*h, pa, r));
  }
}

static void test_bbnc() {
  rng_t r(9);
  models::bbnc_model m;
  CHECK(m.get_runtime_type() == runtime_type(TYPE_B));
  auto h = m.create_hypers();
  h->get_hp_mutator("alpha").set<float>(2.0f);
  h->get_hp_mutator("beta").set<float>(3.0f);
  auto g = h->create_group(r);                       // draws p ~ Beta(2, 3)
  const float p = g->get_ss_mutator("p").accessor().get<float>(0);
  CHECK(p > 0.f && p < 1.f);
  const bool t = true, f = false;
  for (int i = 0; i < 5; i++) g->add_value(*h, value_accessor(&t), r);
  for (int i = 0; i < 2; i++) g->add_value(*h, value_accessor(&f), r);
  g->remove_value(*h, value_accessor(&t), r);
  CHECK(close(g->score_value(*h, value_accessor(&t), r), std::log((double)p)));
  CHECK(close(g->score_value(*h, value_accessor(&f), r), std::log1p(-(double)p)));
  const float hp[2] = {2.f, 3.f};
  struct { uint32_t heads, tails; double p; } oss = {4, 2, (double)p};
  CHECK(close(g->score_data(*h, r), orc_f64_score_data(ORC_BBNC, 0, hp, &oss)));
  auto g2 = h->create_group(r);
  g2->set_ss(g->get_ss());
  CHECK(g2->score_data(*h, r) == g->score_data(*h, r));
}

// BetaNegativeBinomial through distributions_model<T> (distributions.hpp:59-64): r is an integer field
static void test_bnb() {
  rng_t r(5);
  models::distributions_model<BetaNegativeBinomial> m;
  CHECK(m.get_runtime_type() == runtime_type(TYPE_U32));
  auto h = m.create_hypers();
  h->get_hp_mutator("alpha").set<float>(2.5f);
  h->get_hp_mutator("beta").set<float>(1.5f);
  h->get_hp_mutator("r").set<uint32_t>(3);
  const float hp[3] = {2.5f, 1.5f, 3.f};
  auto g = h->create_group(r);
  struct { uint32_t count, sum; } oss = {0, 0};
  uint32_t vals[20];
  for (int i = 0; i < 20; i++) {
    vals[i] = (uint32_t)std::negative_binomial_distribution<int>(3, 0.4)(r);
    g->add_value(*h, value_accessor(&vals[i]), r);
    orc_f64_add_value(ORC_BNB, 0, hp, &oss, &vals[i]);
  }
  for (int i = 0; i < 7; i++) {
    g->remove_value(*h, value_accessor(&vals[i]), r);
    orc_f64_remove_value(ORC_BNB, 0, hp, &oss, &vals[i]);
  }
  CHECK(g->get_ss_mutator("count").accessor().get<uint32_t>(0) == oss.count);
  CHECK(g->get_ss_mutator("sum").accessor().get<uint32_t>(0) == oss.sum);
  for (uint32_t probe : {0u, 1u, 9u, 40u, 5000u})
    CHECK(close(g->score_value(*h, value_accessor(&probe), r), orc_f64_score_value(ORC_BNB, 0, hp, &oss, &probe)));
  CHECK(close(g->score_data(*h, r), orc_f64_score_data(ORC_BNB, 0, hp, &oss)));
  auto h2 = m.create_hypers();
  h2->set_hp(h->get_hp());                            // bag round trip keeps r
  CHECK(h2->get_hp_mutator("r").accessor().get<uint32_t>(0) == 3u);
  auto g2 = h2->create_group(r);
  g2->set_ss(g->get_ss());
  const uint32_t probe = 4;
  CHECK(g2->score_value(*h2, value_accessor(&probe), r) == g->score_value(*h, value_accessor(&probe), r));
}

// the in-tree Dirichlet-Multinomial under its reference names (dm.hpp:19-200, dm.cpp:10-111)
static void test_dm() {
  rng_t r(6);
  const unsigned C = 5;
  models::dm_model m(C);
  CHECK(m.get_runtime_type() == runtime_type(TYPE_I32, C));
  CHECK(m.categories() == C);
  bool threw = false;
  try { models::dm_model bad(1); } catch (const std::runtime_error &) { threw = true; }   // dm.hpp:177
  CHECK(threw);
  auto h = m.create_hypers();
  const float alphas[C] = {0.5f, 1.f, 2.f, 0.25f, 3.f};
  auto mut = h->get_hp_mutator("alphas");
  CHECK(mut.shape() == C);
  for (unsigned i = 0; i < C; i++) mut.set<float>(alphas[i], i);
  auto g = h->create_group(r);
  std::vector<uint8_t> oss(orc_f64_ss_size(ORC_DM, C));
  orc_f64_init(ORC_DM, C, alphas, oss.data());
  const runtime_type vt(TYPE_I32, C);
  int32_t rows[12][C];
  for (int n = 0; n < 12; n++) {
    for (unsigned i = 0; i < C; i++) rows[n][i] = std::poisson_distribution<int>(2.0 + i)(r);
    g->add_value(*h, value_accessor(reinterpret_cast<const uint8_t *>(rows[n]), nullptr, vt), r);
    orc_f64_add_value(ORC_DM, C, alphas, oss.data(), rows[n]);
  }
  for (int n = 0; n < 4; n++) {
    g->remove_value(*h, value_accessor(reinterpret_cast<const uint8_t *>(rows[n]), nullptr, vt), r);
    orc_f64_remove_value(ORC_DM, C, alphas, oss.data(), rows[n]);
  }
  const int32_t probes[3][C] = {{0, 0, 0, 0, 0}, {1, 0, 3, 0, 2}, {40, 2, 0, 7, 1500}};
  for (const auto &pr : probes)
    CHECK(audit::score("plugin.score_value.dm", g->score_value(*h, value_accessor(reinterpret_cast<const uint8_t *>(pr), nullptr, vt), r),
                       orc_f64_score_value(ORC_DM, C, alphas, oss.data(), pr)));
  {
    // score_data = the float `ratio` field (dm.hpp:86-88) + lgamma terms of the exact counts: against the twin on the
    // group's own ratio the plain gate; the ratio itself after 16 float updates against the double chain (round 1-3: 2e-5
    // on the sum of both)
    const auto &own = static_cast<const models::dm_group &>(*g).repr_;
    const std::vector<uint8_t> rec = fed(own);
    CHECK(audit::score("plugin.score_data.dm", g->score_data(*h, r), orc_f64_score_data(ORC_DM, C, alphas, rec.data())));
    double chain_ratio;
    std::memcpy(&chain_ratio, oss.data() + (4 * C + 7) / 8 * 8, 8);
    CHECK(audit::field("plugin.float_field.dm.ratio", own.ratio, chain_ratio, 16));
    CHECK(std::memcmp(rec.data(), oss.data(), 4 * C) == 0);   // counts: bit-exact
  }
  // bags are the in-tree schema.proto:21-30 messages
  auto g2 = h->create_group(r);
  g2->set_ss(g->get_ss());
  CHECK(g2->score_data(*h, r) == g->score_data(*h, r));
  auto h2 = m.create_hypers();
  h2->set_hp(h->get_hp());
  CHECK(h2->get_hp_mutator("alphas").accessor().get<float>(4) == 3.f);
  threw = false;
  try { g->get_ss_mutator("counts"); } catch (const std::runtime_error &) { threw = true; }   // dm.hpp:68-72
  CHECK(threw);
  threw = false;
  try { models::dm_model(3).create_hypers()->set_hp(h->get_hp()); } catch (const std::runtime_error &) { threw = true; }
  CHECK(threw);                                       // "# categories mismatch", dm.hpp:121-123
}

int main() {
  test_bbnc();
  test_bnb();
  test_dm();
  {
    models::distributions_model<BetaBernoulli> m;
    run_scalar<BetaBernoulli, bool>(ORC_BB, 0, m, {2.f, 0.5f}, {{"alpha", 2.f}, {"beta", 0.5f}},
                                    [](rng_t &r) { return std::bernoulli_distribution(0.7)(r); });
  }
  {
    models::distributions_model<GammaPoisson> m;
    CHECK(m.get_runtime_type() == runtime_type(TYPE_U32));
    run_scalar<GammaPoisson, uint32_t>(ORC_GP, 0, m, {1.5f, 0.75f}, {{"alpha", 1.5f}, {"inv_beta", 0.75f}},
                                       [](rng_t &r) { return uint32_t(std::poisson_distribution<int>(6.0)(r)); });
  }
  {
    models::distributions_model<NormalInverseChiSq> m;
    CHECK(m.get_runtime_type() == runtime_type(TYPE_F32));
    run_scalar<NormalInverseChiSq, float>(ORC_NICH, 0, m, {0.5f, 2.f, 1.5f, 3.f},
                                          {{"mu", 0.5f}, {"kappa", 2.f}, {"sigmasq", 1.5f}, {"nu", 3.f}},
                                          [](rng_t &r) { return std::normal_distribution<float>(4.f, 2.f)(r); });
  }
  test_bb_mutators_write_through();
  test_dd_and_niw();
  audit::dump();
  std::puts("test_plugin_gpu ok");
  return 0;
}
