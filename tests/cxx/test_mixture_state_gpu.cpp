// mixture_state (an entity_based_state_object on the device tables) against a host twin built from the same
// plugin API: the per-entity Gibbs loop of entity_state.hpp, group creation / deletion, likelihoods,
// suff-stat bags, hyper-parameter changes, and the batched sweep.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <random>

#include <microscopes/common/entity_state.hpp>
#include <microscopes/models/distributions.hpp>
#include <microscopes/models/dm.hpp>
#include <microscopes_amd/mixture_state.hpp>

#include "audit.hpp"

using namespace microscopes;
using namespace microscopes::common;

#define CHECK(c)                                                        \
  do {                                                                  \
    if (!(c)) {                                                         \
      std::fprintf(stderr, "%s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #c); \
      return 1;                                                         \
    }                                                                   \
  } while (0)

// What the state's answers are held against (audit.hpp): plugin groups FED THE STATE'S OWN SUFF-STATS (get_suffstats ->
// set_ss: the float fields the device tables hold), evaluated through the per-value API -- one score at a time, in
// double.  The plain 1e-6 gate per score; a sum of component scores + log pseudocount at 1e-6 times the terms'
// magnitudes.  (Rounds 1-3 compared with a twin that lived through the same add / remove calls value by value: its
// float fields are a chain of float roundings, the state's are one rounding of double sums -- the 2e-5 / 1e-4 gates there
// measured that difference of STATE, which is now checked on its own: one half-ulp per update of the chain.)

#pragma pack(push, 1)
struct Row {
  bool b;
  uint32_t c;
  int32_t d;
  float x;
};
#pragma pack(pop)

int main() {
  rng_t rng(5);
  const size_t N = 300, KMAX = 24;
  std::vector<Row> rows(N);
  std::mt19937 gen(11);
  for (size_t i = 0; i < N; i++) {
    const int comp = int(i % 3);
    rows[i].b = std::bernoulli_distribution(comp == 0 ? 0.9 : 0.2)(gen);
    rows[i].c = uint32_t(std::poisson_distribution<int>(2 + 5 * comp)(gen));
    rows[i].d = int32_t((comp * 2 + std::uniform_int_distribution<int>(0, 1)(gen)) % 6);
    rows[i].x = float(std::normal_distribution<double>(4.0 * comp, 1.0)(gen));
  }
  const std::vector<runtime_type> types = {runtime_type(TYPE_B), runtime_type(TYPE_U32), runtime_type(TYPE_I32),
                                           runtime_type(TYPE_F32)};
  recarray::row_major_dataview data(reinterpret_cast<const uint8_t *>(rows.data()), nullptr, N, types);
  std::vector<models::model_shared_ptr> mdl = {
      std::make_shared<models::distributions_model<distributions::BetaBernoulli>>(),
      std::make_shared<models::distributions_model<distributions::GammaPoisson>>(),
      std::make_shared<models::distributions_model_dd128>(6),
      std::make_shared<models::distributions_model<distributions::NormalInverseChiSq>>()};

  hip::mixture_state st(mdl, data, KMAX);
  entity_based_state_object &iface = st;                     // everything below goes through the reference's interface
  CHECK(iface.nentities() == N && iface.ncomponents() == 4 && iface.ngroups() == 0);
  iface.get_cluster_hp_mutator("alpha").set<float>(1.5f);
  iface.get_component_hp_mutator(0, "alpha").set<float>(2.f);
  iface.get_component_hp_mutator(0, "beta").set<float>(2.f);

  // the host twin: one plugin group per (component, gid), driven by the same calls
  std::vector<models::hypers_shared_ptr> hy;
  for (auto &m : mdl) hy.push_back(m->create_hypers());
  hy[0]->get_hp_mutator("alpha").set<float>(2.f);
  hy[0]->get_hp_mutator("beta").set<float>(2.f);
  std::map<size_t, std::vector<models::group_shared_ptr>> twin;
  std::map<size_t, int> twin_updates;                        // per gid: add / remove calls the twin's groups have lived through
  auto twin_group = [&](size_t gid) -> std::vector<models::group_shared_ptr> & {
    auto it = twin.find(gid);
    if (it == twin.end()) {
      std::vector<models::group_shared_ptr> gs;
      for (auto &h : hy) gs.push_back(h->create_group(rng));
      it = twin.emplace(gid, gs).first;
    }
    return it->second;
  };
  auto twin_apply = [&](size_t gid, size_t eid, bool add) {
    auto acc = data.get(eid);
    auto &gs = twin_group(gid);
    twin_updates[gid]++;
    for (size_t f = 0; f < 4; f++, acc.bump()) {
      if (add) gs[f]->add_value(*hy[f], acc.get(), rng);
      else gs[f]->remove_value(*hy[f], acc.get(), rng);
    }
  };

  // seed three groups entity by entity
  std::vector<size_t> g0 = {iface.create_group(rng), iface.create_group(rng), iface.create_group(rng)};
  CHECK(iface.ngroups() == 3 && iface.empty_groups().size() == 3);
  for (size_t e = 0; e < N; e++) {
    const size_t g = g0[(e + (e % 5 == 0)) % 3];              // every fifth entity starts in the wrong group
    iface.add_value(g, e, rng);
    twin_apply(g, e, true);
  }
  CHECK(iface.empty_groups().empty() && iface.groupsize(g0[0]) + iface.groupsize(g0[1]) + iface.groupsize(g0[2]) == N);
  bool threw = false;
  try { iface.add_value(g0[0], 0, rng); } catch (const std::runtime_error &) { threw = true; }
  CHECK(threw);                                              // already assigned

  // the Gibbs assignment kernel, entity by entity, for a stretch of entities
  const auto t_loop = std::chrono::steady_clock::now();
  size_t moved = 0;
  for (size_t e = 0; e < 60; e++) {
    if (iface.empty_groups().empty()) iface.create_group(rng);
    const size_t old = iface.remove_value(e, rng);
    twin_apply(old, e, false);
    auto sc = iface.score_value(e, rng);
    CHECK(sc.first == iface.groups() && sc.second.size() == sc.first.size());
    // log pseudocount + sum of the components' score_value, each component evaluated by a plugin group fed the state's
    // own suff-stats of that (component, group)
    for (size_t i = 0; i < sc.first.size(); i++) {
      const size_t gid = sc.first[i];
      const size_t cnt = iface.groupsize(gid);
      double want = std::log(cnt ? double(cnt) : 1.5 / double(iface.empty_groups().size()));
      double mag = std::max(1.0, std::fabs(want));
      auto acc = data.get(e);
      for (size_t f = 0; f < 4; f++, acc.bump()) {
        auto own = hy[f]->create_group(rng);
        own->set_ss(iface.get_suffstats(f, gid));
        const double s = own->score_value(*hy[f], acc.get(), rng);
        want += s;
        mag += std::max(1.0, std::fabs(s));
      }
      CHECK(audit::sum("mixture_state.score_value.prior_plus_4_components", sc.second[i], want, mag));
      // ... and END TO END against the independent twin, whose groups lived through the same add / remove calls value by
      // value and never saw the device's tables (a wrong but self-consistent table passes the check above).
      // Its float fields are a chain of float roundings where the state's are one rounding of double sums: 2e-5 of the
      // terms' magnitudes, the gate rounds 1-3 held this comparison to (the difference of STATE is gated on its own below)
      double indep = std::log(cnt ? double(cnt) : 1.5 / double(iface.empty_groups().size()));
      auto acc2 = data.get(e);
      for (size_t f = 0; f < 4; f++, acc2.bump()) indep += twin_group(gid)[f]->score_value(*hy[f], acc2.get(), rng);
      CHECK(audit::check("mixture_state.score_value.vs_independent_twin_2e-5", std::fabs(sc.second[i] - indep) / std::fmax(mag, std::fabs(indep)), 2e-5));
    }
    // the state's float fields against the value-by-value twin (nich: mean, count_times_variance; gp: log_prod): one
    // half-ulp per update the twin has seen; the integer fields byte for byte
    for (size_t gid : iface.groups()) {
      if (!iface.groupsize(gid)) continue;
      for (size_t f = 0; f < 4; f++) {
        if (f == 0 || f == 2) { CHECK(iface.get_suffstats(f, gid) == twin_group(gid)[f]->get_ss()); continue; }
        auto own = hy[f]->create_group(rng);
        own->set_ss(iface.get_suffstats(f, gid));
        const int nupd = twin_updates[gid];                      // (the updates THIS group's twin has lived through)
        if (f == 1) {
          const auto &a = static_cast<models::distributions_group<distributions::GammaPoisson> &>(*own).repr_;
          const auto &b = static_cast<models::distributions_group<distributions::GammaPoisson> &>(*twin_group(gid)[f]).repr_;
          CHECK(a.count == b.count && a.sum == b.sum);
          CHECK(audit::field("mixture_state.float_field.gp.log_prod", a.log_prod, b.log_prod, nupd));
        } else {
          const auto &a = static_cast<models::distributions_group<distributions::NormalInverseChiSq> &>(*own).repr_;
          const auto &b = static_cast<models::distributions_group<distributions::NormalInverseChiSq> &>(*twin_group(gid)[f]).repr_;
          CHECK(a.count == b.count);
          CHECK(audit::field("mixture_state.float_field.nich.mean", a.mean, b.mean, nupd));
          CHECK(audit::field("mixture_state.float_field.nich.count_times_variance", a.count_times_variance, b.count_times_variance, nupd));
        }
      }
    }
    const size_t pick = sc.first[util::sample_discrete_log(sc.second, rng)];
    iface.add_value(pick, e, rng);
    twin_apply(pick, e, true);
    moved += pick != old;
    for (size_t gid : iface.empty_groups())                  // keep exactly one empty group around
      if (iface.empty_groups().size() > 1) iface.delete_group(gid);
  }
  CHECK(moved > 0);
  std::printf("per-entity gibbs move (remove + score + twin checks + add, 4 components): %.0f us\n",
              std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_loop).count() / 60.0);
  size_t total = 0;
  for (size_t gid : iface.groups()) total += iface.groupsize(gid);
  CHECK(total == N);

  {  // the same move without the twin's per-value device calls: what a downstream kernel pays per entity
    const auto t0 = std::chrono::steady_clock::now();
    for (size_t e = 100; e < 160; e++) {
      const size_t old = iface.remove_value(e, rng);
      twin_apply(old, e, false);
      auto sc = iface.score_value(e, rng);
      const size_t pick = sc.first[util::sample_discrete_log(sc.second, rng)];
      iface.add_value(pick, e, rng);
      twin_apply(pick, e, true);
    }
    std::printf("per-entity gibbs move incl. 8 per-value twin calls: %.0f us\n",
                std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 60.0);
  }
  {  // the move alone (no twin): leave, score against every group, join -- msc_entity_op, msc_score_value + one copy
     // back, msc_entity_op.  Entities put back where they were, so the twin stays in step.
    const auto t0 = std::chrono::steady_clock::now();
    size_t n_moves = 0;
    for (int rep = 0; rep < 4; rep++)
      for (size_t e = 200; e < 260; e++) {
        const size_t old = iface.remove_value(e, rng);
        auto sc = iface.score_value(e, rng);
        (void)sc;
        iface.add_value(old, e, rng);
        n_moves++;
      }
    std::printf("per-entity gibbs move alone (remove + score + add, 4 components): %.1f us\n",
                std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / double(n_moves));
  }
  // likelihoods against plugin groups fed the state's own suff-stats; integer bags against the value-by-value twin
  auto fed_group = [&](std::vector<models::hypers_shared_ptr> &hs, entity_based_state_object &s, size_t f, size_t gid) {
    auto own = hs[f]->create_group(rng);
    own->set_ss(s.get_suffstats(f, gid));
    return own;
  };
  for (size_t gid : iface.groups())
    for (size_t f = 0; f < 4; f++) {
      CHECK(audit::score("mixture_state.score_likelihood.component_group", iface.score_likelihood(f, gid, rng),
                         fed_group(hy, iface, f, gid)->score_data(*hy[f], rng)));
      if (f == 0 || f == 2) CHECK(iface.get_suffstats(f, gid) == twin_group(gid)[f]->get_ss());   // integer suff-stats: same bytes
    }
  double lsum = 0, lmag = 0;
  for (size_t gid : iface.groups()) {
    const double s = fed_group(hy, iface, 3, gid)->score_data(*hy[3], rng);
    lsum += s;
    lmag += std::max(1.0, std::fabs(s));
  }
  CHECK(audit::sum("mixture_state.score_likelihood.component_all_groups", iface.score_likelihood(3, rng), lsum, lmag));
  CHECK(std::isfinite(iface.score_assignment()));

  // set_suffstats round trip: overwrite a group's gp stats with another group's
  {
    const auto gs = iface.groups();
    const auto bag = iface.get_suffstats(1, gs[0]);
    const auto keep = iface.get_suffstats(1, gs[1]);
    iface.set_suffstats(1, gs[1], bag);
    CHECK(iface.get_suffstats(1, gs[1]) == bag);
    CHECK(audit::score("mixture_state.set_suffstats_round_trip", iface.score_likelihood(1, gs[1], rng), iface.score_likelihood(1, gs[0], rng)));
    iface.set_suffstats(1, gs[1], keep);
  }

  // a hyper-parameter written through a mutator reaches the device before the next score
  {
    const size_t gid = iface.groups()[0];
    const float before = iface.score_likelihood(3, gid, rng);
    iface.get_component_hp_mutator(3, "kappa").set<float>(7.f);
    hy[3]->get_hp_mutator("kappa").set<float>(7.f);
    const float after = iface.score_likelihood(3, gid, rng);
    CHECK(before != after && audit::score("mixture_state.score_likelihood.after_hp_change", after, fed_group(hy, iface, 3, gid)->score_data(*hy[3], rng)));
  }

  // the batched sweep: the partition stays a partition, group sizes follow, tables equal a rebuild from scratch
  {
    for (int sweep = 0; sweep < 3; sweep++) st.gibbs_sweep(99, uint64_t(sweep), rng);
    const auto as = iface.assignments();
    std::map<size_t, size_t> cnt;
    for (ssize_t a : as) {
      CHECK(a >= 0);
      cnt[size_t(a)]++;
    }
    for (size_t gid : iface.groups()) CHECK(iface.groupsize(gid) == (cnt.count(gid) ? cnt[gid] : 0));
    for (const auto &c : cnt) CHECK(iface.groupsize(c.first) == c.second);
    // a fresh state given the same partition in one call holds the same integer suff-stats
    hip::mixture_state st2(mdl, data, KMAX);
    st2.get_cluster_hp_mutator("alpha").set<float>(1.5f);
    std::vector<size_t> labels(as.begin(), as.end());
    st2.assign_all(labels, rng);
    std::map<size_t, std::string> by_size_a, by_size_b;      // compare through a key that does not depend on ids
    for (size_t gid : iface.groups())
      if (iface.groupsize(gid)) by_size_a[iface.groupsize(gid) * 1000003 + size_t(as.end() - std::find(as.begin(), as.end(), ssize_t(gid)))] = iface.get_suffstats(2, gid);
    const auto as2 = st2.assignments();
    for (size_t gid : st2.groups())
      if (st2.groupsize(gid)) by_size_b[st2.groupsize(gid) * 1000003 + size_t(as2.end() - std::find(as2.begin(), as2.end(), ssize_t(gid)))] = st2.get_suffstats(2, gid);
    CHECK(by_size_a == by_size_b);
  }
  // a run of batched sweeps on a larger state: the host partition is only rebuilt when somebody looks
  {
    const size_t M = 200000;
    std::vector<Row> big(M);
    for (size_t i = 0; i < M; i++) big[i] = rows[i % N];
    recarray::row_major_dataview bdata(reinterpret_cast<const uint8_t *>(big.data()), nullptr, M, types);
    hip::mixture_state bs(mdl, bdata, 32);
    bs.get_cluster_hp_mutator("alpha").set<float>(1.f);
    std::vector<size_t> labels(M);
    for (size_t i = 0; i < M; i++) labels[i] = i % 5;
    bs.assign_all(labels, rng);
    bs.gibbs_sweep(7, 0, rng);                                // (first step: setup)
    const auto t0 = std::chrono::steady_clock::now();
    for (int sw = 1; sw <= 20; sw++) bs.gibbs_sweep(7, uint64_t(sw), rng);
    const auto as = bs.assignments();                         // <- download + rebuild happen here, once
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    std::printf("20 batched sweeps of %zu entities + one partition sync: %.0f us per sweep\n", M, us / 20.0);
    std::map<size_t, size_t> cnt;
    for (ssize_t a : as) {
      CHECK(a >= 0);
      cnt[size_t(a)]++;
    }
    size_t tot = 0;
    for (size_t gid : bs.groups()) {
      CHECK(bs.groupsize(gid) == (cnt.count(gid) ? cnt[gid] : 0));
      tot += bs.groupsize(gid);
    }
    CHECK(tot == M);
    // and the per-entity calls continue from the synced partition
    const size_t old = bs.remove_value(17, rng);
    auto sc = bs.score_value(17, rng);
    CHECK(sc.first == bs.groups());
    bs.add_value(old, 17, rng);
    CHECK(bs.groupsize(old) == cnt[old]);
  }
  // vector-valued components (packed records on the ABI): niw(2) and dm(3) through the same interface
  {
#pragma pack(push, 1)
    struct VRow { float v[2]; int32_t c[3]; };
#pragma pack(pop)
    const size_t M = 40;
    std::vector<VRow> vr(M);
    for (size_t i = 0; i < M; i++) {
      vr[i].v[0] = float(std::normal_distribution<double>(i % 2 ? 3.0 : -3.0, 1.0)(gen));
      vr[i].v[1] = float(std::normal_distribution<double>(0.0, 1.0)(gen));
      for (int j = 0; j < 3; j++) vr[i].c[j] = std::uniform_int_distribution<int>(0, 4)(gen);
    }
    const std::vector<runtime_type> vt = {runtime_type(TYPE_F32, 2), runtime_type(TYPE_I32, 3)};
    recarray::row_major_dataview vdata(reinterpret_cast<const uint8_t *>(vr.data()), nullptr, M, vt);
    std::vector<models::model_shared_ptr> vm = {std::make_shared<models::distributions_model_niwv>(2),
                                                std::make_shared<models::dm_model>(3)};
    hip::mixture_state vs(vm, vdata, 8);
    vs.get_cluster_hp_mutator("alpha").set<float>(1.f);
    std::vector<models::hypers_shared_ptr> vh = {vm[0]->create_hypers(), vm[1]->create_hypers()};
    for (int i = 0; i < 3; i++) {                            // (a fresh dm_hypers holds zero alphas, as upstream: dm.hpp:168)
      vs.get_component_hp_mutator(1, "alphas").set<float>(0.5f + i, i);
      vh[1]->get_hp_mutator("alphas").set<float>(0.5f + i, i);
    }
    const size_t ga = vs.create_group(rng), gb = vs.create_group(rng);
    std::vector<models::group_shared_ptr> ta = {vh[0]->create_group(rng), vh[1]->create_group(rng)},
                                          tb = {vh[0]->create_group(rng), vh[1]->create_group(rng)};
    for (size_t e = 0; e + 1 < M; e++) {
      vs.add_value(e % 2 ? ga : gb, e, rng);
      auto acc = vdata.get(e);
      auto &t = e % 2 ? ta : tb;
      for (size_t f = 0; f < 2; f++, acc.bump()) t[f]->add_value(*vh[f], acc.get(), rng);
    }
    {  // dm counts are exact (the float `ratio` next to them is a float sum on one side, a double one on the other)
      auto g = vh[1]->create_group(rng);
      g->set_ss(vs.get_suffstats(1, ga));
      auto *got = static_cast<models::dm_group *>(g.get());
      auto *want = static_cast<models::dm_group *>(ta[1].get());
      CHECK(got->repr_.counts == want->repr_.counts);
    }
    (void)ta; (void)tb;
    for (size_t f = 0; f < 2; f++)
      for (size_t gid : {ga, gb})
        CHECK(audit::score(f == 0 ? "mixture_state.score_likelihood.niw2" : "mixture_state.score_likelihood.dm3",
                           vs.score_likelihood(f, gid, rng), fed_group(vh, vs, f, gid)->score_data(*vh[f], rng)));
    {  // the state's niw sums (one rounding of double sums) against the value-by-value twin's (a float chain): a half-ulp
       // per update of the largest entry
      for (size_t gid : {ga, gb}) {
        const auto own = fed_group(vh, vs, 0, gid);
        const auto &a = static_cast<models::distributions_group<distributions::NormalInverseWishartV> &>(*own).repr_;
        const auto &b = static_cast<models::distributions_group<distributions::NormalInverseWishartV> &>(*(gid == ga ? ta : tb)[0]).repr_;
        CHECK(a.count == b.count);
        double big = 1.0, err = 0.0;
        for (size_t i = 0; i < 4; i++) big = std::max(big, std::fabs(double(b.sum_xxT[i]))), err = std::max(err, std::fabs(double(a.sum_xxT[i]) - double(b.sum_xxT[i])));
        CHECK(audit::check("mixture_state.float_field.niw2.sum_xxT", err / (std::ldexp(big, -24) * double(M)), 1.0));
      }
    }
    auto sc = vs.score_value(M - 1, rng);                    // the last entity was never added
    CHECK(sc.first.size() == 2);
    for (size_t i = 0; i < 2; i++) {
      double want = std::log(double(vs.groupsize(sc.first[i])));
      double mag = std::max(1.0, std::fabs(want));
      auto acc = vdata.get(M - 1);
      for (size_t f = 0; f < 2; f++, acc.bump()) {
        const double s = fed_group(vh, vs, f, sc.first[i])->score_value(*vh[f], acc.get(), rng);
        want += s;
        mag += std::max(1.0, std::fabs(s));
      }
      CHECK(audit::sum("mixture_state.score_value.prior_plus_niw2_dm3", sc.second[i], want, mag));
    }
    // niw suff-stats round trip through the packed record
    const auto bag = vs.get_suffstats(0, ga);
    vs.set_suffstats(0, gb, bag);
    CHECK(audit::score("mixture_state.niw_suffstats_round_trip", vs.score_likelihood(0, gb, rng), vs.score_likelihood(0, ga, rng)));
  }
  // a non-conjugate component (bbnc: every group carries its own p ~ Beta(alpha, beta), bbnc.cpp:129-133):
  // create_group must put the model's initial group into the slot, a recycled slot must not keep its previous
  // occupant's p, and the free slots a sweep offers as empty groups need a p of their own
  {
    const size_t M = 64;
    std::vector<uint8_t> bits(M);
    for (size_t i = 0; i < M; i++) bits[i] = uint8_t(std::bernoulli_distribution(i % 2 ? 0.85 : 0.15)(gen));
    const std::vector<runtime_type> bt = {runtime_type(TYPE_B)};
    recarray::row_major_dataview bdata(bits.data(), nullptr, M, bt);
    std::vector<models::model_shared_ptr> bm = {std::make_shared<models::bbnc_model>()};
    hip::mixture_state bs(bm, bdata, 6);
    auto p_of = [&](size_t gid) {
      auto g = bm[0]->create_hypers()->create_group(rng);
      g->set_ss(bs.get_suffstats(0, gid));
      return static_cast<models::bbnc_group *>(g.get())->repr_.p;
    };
    bool threw_alpha = false;                                 // alpha unset (0): scoring must refuse, not draw under alpha = 1
    const size_t ga = bs.create_group(rng), gb = bs.create_group(rng);
    try { (void)bs.score_value(0, rng); } catch (const std::runtime_error &) { threw_alpha = true; }
    CHECK(threw_alpha);
    bs.get_cluster_hp_mutator("alpha").set<float>(1.f);
    const float pa = p_of(ga), pb = p_of(gb);
    CHECK(pa > 0.f && pa < 1.f && pb > 0.f && pb < 1.f && pa != pb);
    auto sc = bs.score_value(0, rng);
    CHECK(sc.second.size() == 2);
    for (size_t i = 0; i < 2; i++) {                          // log(alpha / 2) + log p(v | p of that group)
      const float pg = sc.first[i] == ga ? pa : pb;
      CHECK(std::isfinite(sc.second[i]));
      CHECK(audit::score("mixture_state.score_value.bbnc", sc.second[i], std::log(0.5) + std::log(bits[0] ? double(pg) : 1.0 - double(pg))));
    }
    bs.delete_group(gb);                                      // the slot goes back ...
    const size_t gc = bs.create_group(rng);                   // ... and comes out again with a p of its own
    const float pc = p_of(gc);
    CHECK(pc > 0.f && pc < 1.f && pc != pb);
    std::vector<size_t> labels(M);
    bs.delete_group(gc);
    bs.delete_group(ga);
    for (size_t i = 0; i < M; i++) labels[i] = i % 2;
    bs.assign_all(labels, rng);
    for (int sw = 0; sw < 4; sw++) bs.gibbs_sweep(3, uint64_t(sw), rng);
    size_t tot = 0;
    for (size_t gid : bs.groups()) {
      const float pg = p_of(gid);
      CHECK(pg > 0.f && pg < 1.f);                            // incl. groups born from free slots during a sweep
      tot += bs.groupsize(gid);
    }
    CHECK(tot == M);
  }
  audit::dump();
  std::printf("test_mixture_state_gpu ok\n");
  return 0;
}
