// group::sample_value (base.hpp:29) of the device-backed models: draws come from the posterior predictive that
// score_value scores.  Host-side code, no device call: checked by moments against the closed forms.
#include <cmath>
#include <cstdio>
#include <vector>

#include <microscopes/models/distributions.hpp>
#include <microscopes/models/bbnc.hpp>
#include <microscopes/models/dm.hpp>

using namespace microscopes;
using namespace microscopes::common;

#define CHECK(c)                                                        \
  do {                                                                  \
    if (!(c)) {                                                         \
      std::fprintf(stderr, "%s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #c); \
      return 1;                                                         \
    }                                                                   \
  } while (0)

template <typename V>
static std::vector<double> draw_many(models::group &g, const models::hypers &h, runtime_type t, int n, rng_t &rng) {
  std::vector<double> sum(t.n(), 0.0);
  V buf[64];                                               // (not std::vector: vector<bool> has no data())
  for (int i = 0; i < n; i++) {
    value_mutator m(reinterpret_cast<uint8_t *>(buf), t);
    g.sample_value(h, m, rng);
    for (unsigned e = 0; e < t.n(); e++) sum[e] += double(buf[e]);
  }
  for (double &s : sum) s /= n;
  return sum;
}

int main() {
  rng_t rng(3);
  const int n = 40000;
  {  // bb: P(1) = (alpha + heads) / (alpha + beta + heads + tails)
    models::distributions_model<distributions::BetaBernoulli> m;
    auto h = m.create_hypers();
    h->get_hp_mutator("alpha").set<float>(2.f);
    h->get_hp_mutator("beta").set<float>(3.f);
    auto g = h->create_group(rng);
    g->get_ss_mutator("heads").set<uint32_t>(30);
    g->get_ss_mutator("tails").set<uint32_t>(10);
    CHECK(std::fabs(draw_many<bool>(*g, *h, m.get_runtime_type(), n, rng)[0] - 32.0 / 45.0) < 0.01);
  }
  {  // gp: negative binomial with mean (alpha + sum) / (inv_beta + count)
    models::distributions_model<distributions::GammaPoisson> m;
    auto h = m.create_hypers();
    h->get_hp_mutator("alpha").set<float>(1.5f);
    h->get_hp_mutator("inv_beta").set<float>(0.5f);
    auto g = h->create_group(rng);
    g->get_ss_mutator("count").set<uint32_t>(20);
    g->get_ss_mutator("sum").set<uint32_t>(90);
    CHECK(std::fabs(draw_many<uint32_t>(*g, *h, m.get_runtime_type(), n, rng)[0] - 91.5 / 20.5) < 0.06);
  }
  {  // dd: frequencies (alpha_i + c_i) / total
    models::distributions_model_dd128 m(4);
    auto h = m.create_hypers();
    auto g = h->create_group(rng);
    const uint32_t counts[4] = {5, 0, 20, 1};
    for (int i = 0; i < 4; i++) g->get_ss_mutator("counts").set<uint32_t>(counts[i], i);
    g->get_ss_mutator("count_sum").set<uint32_t>(26);
    std::vector<int> hist(4, 0);
    int v = 0;
    for (int i = 0; i < n; i++) {
      value_mutator mu(&v);
      g->sample_value(*h, mu, rng);
      CHECK(v >= 0 && v < 4);
      hist[v]++;
    }
    for (int i = 0; i < 4; i++) CHECK(std::fabs(hist[i] / double(n) - (1.0 + counts[i]) / 30.0) < 0.012);
  }
  {  // nich: Student-t centred on the posterior mean
    models::distributions_model<distributions::NormalInverseChiSq> m;
    auto h = m.create_hypers();
    h->get_hp_mutator("nu").set<float>(5.f);
    auto g = h->create_group(rng);
    g->get_ss_mutator("count").set<uint32_t>(50);
    g->get_ss_mutator("mean").set<float>(3.f);
    g->get_ss_mutator("count_times_variance").set<float>(60.f);
    CHECK(std::fabs(draw_many<float>(*g, *h, m.get_runtime_type(), n, rng)[0] - 150.0 / 51.0) < 0.03);
  }
  {  // bnb: finite mean r b / (a - 1) with a = alpha + r count, b = beta + sum
    models::distributions_model<distributions::BetaNegativeBinomial> m;
    auto h = m.create_hypers();
    h->get_hp_mutator("alpha").set<float>(3.f);
    h->get_hp_mutator("beta").set<float>(2.f);
    h->get_hp_mutator("r").set<uint32_t>(4);
    auto g = h->create_group(rng);
    g->get_ss_mutator("count").set<uint32_t>(10);
    g->get_ss_mutator("sum").set<uint32_t>(25);
    CHECK(std::fabs(draw_many<uint32_t>(*g, *h, m.get_runtime_type(), n, rng)[0] - 4.0 * 27.0 / 42.0) < 0.08);
  }
  {  // niw: multivariate t centred on mu_n
    models::distributions_model_niwv m(3);
    auto h = m.create_hypers();
    auto g = h->create_group(rng);
    auto *gg = static_cast<models::distributions_group<distributions::NormalInverseWishartV> *>(g.get());
    gg->repr_.count = 40;
    const float mean[3] = {1.f, -2.f, 0.5f};
    for (int i = 0; i < 3; i++) gg->repr_.sum_x[i] = 40 * mean[i];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) gg->repr_.sum_xxT[i * 3 + j] = 40 * (mean[i] * mean[j] + (i == j ? 1.f : 0.2f));
    const auto mu = draw_many<float>(*g, *h, m.get_runtime_type(), n, rng);
    for (int i = 0; i < 3; i++) CHECK(std::fabs(mu[i] - 40.0 * mean[i] / 41.0) < 0.03);
  }
  {  // bbnc: Bernoulli(p); dm: throws, as upstream
    models::bbnc_model m;
    auto h = m.create_hypers();
    auto g = h->create_group(rng);
    g->get_ss_mutator("p").set<float>(0.2f);
    CHECK(std::fabs(draw_many<bool>(*g, *h, m.get_runtime_type(), n, rng)[0] - 0.2) < 0.01);
    models::dm_model dm(3);
    auto dh = dm.create_hypers();
    auto dg = dh->create_group(rng);
    int32_t out[3];
    value_mutator mu(reinterpret_cast<uint8_t *>(out), dm.get_runtime_type());
    bool threw = false;
    try { dg->sample_value(*dh, mu, rng); } catch (const std::runtime_error &) { threw = true; }
    CHECK(threw);
  }
  std::puts("test_sample_value ok");
  return 0;
}
